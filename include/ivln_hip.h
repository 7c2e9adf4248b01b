/*
 * ivln_hip.h - C ABI of libivln_hip.so, the MI355X (gfx950) implementation of the IVLN-CE MapCMA
 * hot path.  Plain pointers and sizes only: every `void*`/typed pointer below is a DEVICE pointer
 * owned by the caller (contiguous, layout documented per call) unless marked "host"; `stream` is a
 * hipStream_t passed as void*.  All entry points return 0 on success or a negative IVLN_E_* code
 * (never throw, never allocate persistent memory outside *_create/_destroy handles).  Kernels are
 * enqueued on `stream` and the call returns immediately.
 *
 * The reference (jacobkrantz/IVLN-CE) is pure Python/torch and has no FFI: each entry point cites
 * the reference function whose arithmetic it replaces (paths relative to the reference root).  The
 * Python plugin classes in ivln-ce_amd/ bind these with ctypes.
 */
#ifndef IVLN_HIP_H
#define IVLN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IVLN_OK 0
#define IVLN_E_INVALID -1   /* bad argument */
#define IVLN_E_HIP -2       /* a HIP runtime call failed */
#define IVLN_E_KEYSPACE -3  /* mapper: keep-highest key exceeded the dense table capacity */
#define IVLN_E_CAPACITY -4  /* mapper: world cloud exceeded its capacity */
#define IVLN_E_UNSUPPORTED -5

const char* ivln_strerror(int code);
int ivln_version(void);

/* ------------------------------------------------------------------------------------------
 * Egocentric semantic mapper (replaces MappingModule.forward,
 * ivlnce_baselines/common/mapping_module/mapper.py:904-944 and everything it calls:
 * projector/core.py:117-230, mapper.py:236-266, 387-474, 477-648, 807-901).
 * ------------------------------------------------------------------------------------------ */
typedef struct ivln_mapper ivln_mapper;

/* B_max envs, depth H x W, vertical fov (radians), map size in metres and resolution
 * (obs_transforms.py:154-175 / setup_mapping_module.py:13-59).  world_capacity = max points in the
 * world cloud (0 -> default), table_cells = dense keep-highest table capacity (0 -> default). */
int ivln_mapper_create(int B_max, int H, int W, double vfov_rad, double height_m, double width_m,
                       double res_m, int64_t world_capacity, int64_t table_cells, ivln_mapper** out);
int ivln_mapper_destroy(ivln_mapper* m);
/* Forget the world cloud (new process / new eval). */
int ivln_mapper_reset(ivln_mapper* m, void* stream);

/* Launch width of the step's kernels: the local-cloud kernels on `local_blocks` workgroups; the world-cloud kernels on
 * at most `world_blocks` workgroups that take 16 points per thread, so that they cover few CUs while the cloud is
 * small and more as it grows (0, 0 = full width: one chunk of 1024 pixels per workgroup / up to 1024 workgroups, one
 * point per thread).  Results do not depend on it.  A mapper that runs BESIDE a latency-bound kernel chain on another
 * stream (the depth ResNet of the rollout step) should be narrow - e.g. 16 per env / 256: it then takes ~140 instead
 * of ~60 us at 4 envs but leaves the chip to the chain. */
int ivln_mapper_set_launch_width(ivln_mapper* m, int local_blocks, int world_blocks);

/* core.py:6-37 (_transform3D with elevation + pi, mapper.py:132-138) and mapper.py:38-48
 * (rotate_around_y_matrix(-heading)): pose f32 (B,3), orientation f64 (B,2) [elev, heading] ->
 * T f32 (B,4,4), rot f32 (B,3,3).  fp64 sin/cos on device, rounded to fp32. */
int ivln_mapper_frames(const float* pose, const double* orientation, int B, float* T, float* rot,
                       void* stream);

/* One mapper step.  depth f32 (B,H,W) in [0,1]; labels u8 (B,H,W) (semantic12 or RedNet argmax);
 * T f32 (B,4,4); pose f32 (B,3); rot f32 (B,3,3); not_done u8 (B) -> occupancy_map, semantic_map
 * u8 (B,rows,cols), both fully rewritten. */
int ivln_mapper_step(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T,
                     const float* pose, const float* rot, const uint8_t* not_done, int B,
                     uint8_t* occ_out, uint8_t* sem_out, void* stream);

/* The same step from the sensor pose itself: ivln_mapper_frames + ivln_mapper_step in six launches instead of seven
 * (the transforms are derived inside the first kernel).  pose f32 (B,3), orientation f64 (B,2) [elevation, heading] (mapper.py:132-138, obs_transforms.py:79-103);
 * T_out f32 (B,4,4) and rot_out f32 (B,3,3) receive what ivln_mapper_frames would have produced (bit-identical). */
int ivln_mapper_step_posed(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* pose,
                           const double* orientation, const uint8_t* not_done, int B, uint8_t* occ_out,
                           uint8_t* sem_out, float* T_out, float* rot_out, void* stream);

/* ivln_mapper_step_posed in two calls, so that the half that needs no labels can run beside the network that predicts them
 * (mapper.py:398-474: unprojection, filters and the keep-highest selection read depth and pose only; the labels enter at
 * :513-617).  _begin = camera transforms + local min / max + the keep-highest arg-max (2 launches), _finish = label
 * selection, world-cloud merge, raster (4 launches): the same six kernels in the same order, the same bits.  _begin may be
 * enqueued on another stream than _finish as long as _finish is ordered behind it.  _finish takes the T / rot that _begin
 * wrote (T_out, rot_out) and the same depth, pose, not_done and B; nothing on the host links the two calls (a captured step
 * records each half once and replays them many times), so a _finish without its _begin is the caller's error: it would
 * select from the previous step's arg-max table. */
int ivln_mapper_step_begin(ivln_mapper* m, const float* depth, const float* pose, const double* orientation,
                           const uint8_t* not_done, int B, uint8_t* occ_out, float* T_out, float* rot_out, void* stream);
int ivln_mapper_step_finish(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                            const float* rot, const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out, void* stream);

/* Known-map mode (mapper.py:851-881): begin = clear finished/paused envs; load = append the
 * pre-built cloud of env b (xyz f32 (n,3), sem u8 (n), device); raster = maps from the cloud. */
int ivln_mapper_known_begin(ivln_mapper* m, const uint8_t* not_done, int B, void* stream);
int ivln_mapper_load_known(ivln_mapper* m, int b, const float* xyz, const uint8_t* sem, int64_t n,
                           void* stream);
int ivln_mapper_known_raster(ivln_mapper* m, const float* pose, const float* rot, int B,
                             uint8_t* occ_out, uint8_t* sem_out, void* stream);

/* Synchronises `stream` and returns the sticky device status (IVLN_OK / IVLN_E_KEYSPACE /
 * IVLN_E_CAPACITY); *world_n (host, may be NULL) receives the world cloud size. */
int ivln_mapper_status(ivln_mapper* m, int64_t* world_n, void* stream);
/* Debug/test export of the world cloud (unordered): xyz f32 (max_n,3), meta u32 (max_n) =
 * batch<<8|label, rank i64 (max_n) = position key of the reference's ordering.  Synchronises. */
int ivln_mapper_world_export(ivln_mapper* m, float* xyz, uint32_t* meta, int64_t* rank,
                             int64_t max_n, int64_t* n_out, void* stream);


/* ------------------------------------------------------------------------------------------
 * fp32 MFMA implicit GEMM:  D[m][n] = sum_k A[m][k] * B[k][n]  (csrc/gemm_conv.hip).
 * One descriptor covers every GEMM-shaped op of the path.  Replaces the ATen/cuDNN kernels behind
 * nn.Conv2d (models/encoders/map_encoder.py:13-20, rednet.py:20-65, habitat-lab ResNetEncoder via
 * models/encoders/resnet_encoders.py:31-43), nn.ConvTranspose2d (rednet.py:117-150,330-340),
 * nn.Conv1d k=1 and nn.Linear (models/map_cma_policy.py:156-231) and their autograd backward.
 * ------------------------------------------------------------------------------------------ */
enum { IVLN_A_MK = 0,      /* A[m*lda + k]  (OIHW weights, Linear.weight)            */
       IVLN_A_KM = 1,      /* A[k*lda + m]                                            */
       IVLN_A_NCHW_P = 2   /* A[m = channel][k = pixel] of an NCHW tensor (wgrad dy)  */ };
enum { IVLN_B_CONV = 0,    /* im2col gather from NCHW via koff/kpos tables            */
       IVLN_B_CONV1X1 = 1, /* 1x1 conv, pad 0                                         */
       IVLN_B_KN = 2,      /* B[k*ldb + n]                                            */
       IVLN_B_NK = 3,      /* B[n*ldb + k]  (activations [rows][features])            */
       IVLN_B_IM2COL_T = 4,/* B[k = out pixel][n = (ci,kh,kw)] (conv weight gradient) */
       IVLN_B_CONVT = 5,   /* transposed-conv gather (koff = ci*Hin*Win)              */
       IVLN_B_CONV_K3 = 6, /* 3x3, dilation 1: tap indices by constant division, no tables */
       IVLN_B_CONV_K7 = 7, /* 7x7, dilation 1                                            */
       IVLN_B_CONV_K2 = 8  /* 2x2, dilation 1 (taps at input offsets 0..1 with pad 0 and Hout = Hin: the row / column
                              past the edge reads as zero - the window of a stride-2 3x3 transposed conv's parity
                              classes) */ };
enum { IVLN_D_NCHW = 0,    /* D[(img*Ctot + m)*HoWo + pp], n = img*HoWo + pp          */
       IVLN_D_DENSE = 1,   /* D[m*sDm + n*sDn]                                        */
       IVLN_D_NCHW_UP2 = 2,/* one output-parity class of a stride-2 transposed conv: pixel (ho,wo) of the
                              (Hout x Wout) class grid lands at (2*ho + sDm, 2*wo + sDn) of a
                              (2*Hout x 2*Wout) NCHW destination (sDm, sDn in {0,1} = row/col parity) */
       IVLN_D_NCHW_UP2X4 = 3 /* all four parity classes in ONE GEMM: rows m = 4*channel + cls, cls = 2*a + b;
                              pixel (ho,wo) of row m lands at (2*ho + a, 2*wo + b) of channel m / 4; the epilogue
                              parameters are indexed by the channel (rednet.py:210-216, 262-279).  Interleaved so
                              that an output tile holds whole 2x2 output blocks and leaves as 16-byte stores */ };

typedef struct ivln_gemm_desc {
    const float* A;
    const float* B;
    float* D;
    int M, N, K;
    int amode, bmode, dmode;
    int64_t lda, ldb;
    /* conv geometry (B_CONV / B_CONV1X1 / B_CONVT / B_IM2COL_T, A_NCHW_P, D_NCHW) */
    int Cin, Hin, Win, Hout, Wout, stride, pad, dil;
    const int32_t* koff; /* [K or N]: ci*Hin*Win + kh*dil*Win + kw*dil (B_CONVT: ci*Hin*Win) */
    const int32_t* kpos; /* [K or N]: kh << 16 | kw                                         */
    int HoWo;            /* Hout*Wout (1 for plain GEMMs)                                   */
    int Ctot;            /* channels of the NCHW destination (0 -> M); D may point at a slice */
    int64_t in_img_stride; /* floats between images of the NCHW input (0 -> Cin*Hin*Win)        */
    int64_t sDm, sDn;
    /* fused epilogue: v = acc*scale[m] + shift[m] (or + shift[m]); + residual[addr]; + D[addr]
     * when accumulate; ReLU.  residual is addressed like D. */
    const float* scale;
    const float* shift;
    const float* residual;
    int relu, accumulate;
    /* split-K: splits = 0 -> heuristic (needs ws), >= 1 forced.  ws holds splits*M*N floats. */
    int splits;
    float* ws;
    int64_t ws_floats;
    /* defer_epilogue: write the raw accumulators of every split to ws ([split][M][N]) and launch no
     * reduce/epilogue kernel - the consumer (ivln_groupnorm_f32, ivln_scale_shift_relu_avgpool2_f32)
     * fuses the slab reduction.  splits_used (host, optional) receives the split count chosen. */
    int defer_epilogue;
    int* splits_used;
    /* 0 = heuristic (direct conv / vector-load GEMM where eligible, else the scalar-gather implicit GEMM);
     * 1..5 force the scalar-gather kernel with block tile 64x64 / 32x128 / 128x32 / 128x128 / 64x128;
     * 6 insist on the LDS-patch direct conv / weight-gradient kernels (conv_direct.hip);
     * 7 insist on the float4-staged GEMM (gemm_vec.hip); 8 insist on the streaming short-K 1x1 conv
     * (conv1x1_stream.hip: K = 64 / 128 / 256, NCHW, whole 128-pixel strips); 9 insist on the split-bf16 direct conv
     * (conv_bf3.hip, needs A_split); 10 insist on its K-split-over-waves form for pixel-starved deep 3x3 convs
     * (k_conv_bf3_ks), 11 on the deep-K 1x1 form (k_conv1x1_bf3_ks).  6 ... 11 return IVLN_E_UNSUPPORTED when
     * the shape is not eligible (tuning, tests). */
    int tile_override;
    /* optional (stride-1 3x3 / 7x7 / 2x2 convs): the weights pre-arranged by ivln_conv_pack_weights_f32; when set and
     * the direct kernel is chosen, its weight staging becomes a linear float4 copy.  A must still be given.
     * 1x1 convs with K = 64 / 128 / 256 (KS = 1 in the pack call): the streaming kernel's per-lane register image. */
    const float* A_packed;
    /* 0 (default): workgroup ids are remapped so that each XCD (hardware places workgroup b on XCD b % 8, each
     * with its own 4 MB L2) works on one contiguous range of output tiles - neighbouring tiles share operand rows /
     * input halos through that L2 instead of re-fetching them from HBM.  1: identity mapping (A/B measurements).
     * Results are identical either way. */
    int no_xcd_remap;
    /* Image-grouped weights (conv modes with D_NCHW; 0 = off): images [g*grp_imgs, (g+1)*grp_imgs) use weight set g,
     * A + g*a_grp_stride (and A_packed + g*a_packed_grp_stride), and epilogue parameters scale / shift [g*M + m].
     * RedNet's RGB and depth encoders are the same ResNet-50 with different weights (rednet.py:190-222): stacked on the
     * image axis they run as ONE launch per layer with twice the output tiles (fewer split-K epilogues).  Output tiles
     * must not straddle a group: grp_imgs * HoWo has to be a multiple of the pixel tile (checked, else
     * IVLN_E_UNSUPPORTED). */
    int grp_imgs;
    int64_t a_grp_stride, a_packed_grp_stride;
    /* 1: keep the per-lane 4-byte stores of the MFMA layout for NCHW outputs instead of sending the tile through LDS
     * and writing 16 bytes per lane along the pixel index (csrc/gemm_vec.hip; A/B measurements and tests).  Results
     * are identical either way. */
    int no_wide_epilogue;
    /* Optional per-tile statistics of what the launch STORES (BatchNorm in train mode right behind the conv, map_encoder.py:
     * 13-20): when the direct conv kernel takes its wide NCHW epilogue (stride-1/2 3x3 / 7x7 / 2x2, splits == 1,
     * Wout % 4 == 0) every (pixel tile, output channel) leaves {count, mean, M2} = Welford partials of its <= 128 stored
     * values in stat_partials[(tile*M + m)*3 ..]; *stat_tiles (host) receives the number of pixel tiles, or 0 when the
     * launch went another way and wrote nothing (the caller then runs ivln_bn_train_stats_f32).  Merged in a fixed order
     * by ivln_bn_stats_from_partials_f32. */
    float* stat_partials;
    int* stat_tiles;
    /* optional (stride-1 same-size 3x3 / 7x7 convs into an NCHW destination, Wout a multiple of 4 and >= 8): the weights
     * split into three bf16 pieces per value and arranged per MFMA lane by ivln_conv_split_weights_f32
     * (ivln_conv_split_words(M, Cin, KS) 4-byte words per weight set, KS 1 / 3 / 7; a_split_grp_stride = words between the sets of an
     * image-grouped conv).  When set and the shape fills the chip, the conv runs on the bf16 MFMA pipe with BOTH operands
     * carried as three bf16 pieces (exact) and six of the nine piece products accumulated in fp32: the dropped ones are
     * below 2^-23 of a product, i.e. the result is as close to the exact convolution as the fp32 MFMA kernel's
     * (csrc/conv_bf3.hip).  tile_override 9 insists on this kernel (IVLN_E_UNSUPPORTED when not eligible), 20 + c on its tile
     * configuration c = 0 ... 6 (tuning: tools/conv_cfg_sweep.py);
     * IVLN_NO_SPLIT_BF16 in the environment keeps the fp32 MFMA kernels (A/B).  A must still be given. */
    const void* A_split;
    int64_t a_split_grp_stride;
    /* 1: the weight gradient of a 7x7 same-size conv (A_NCHW_P x B_IM2COL_T -> dense) may run on the same split-bf16
     * arithmetic (k_wgrad_bf3: both operands are activations, split while they are staged - nothing to pre-arrange);
     * tile_override 9 insists on it.  2: ... and the caller PROMISES that every value of B (the conv's input x) is exact in
     * bf16 - the map CNN's first layer reads one-hot map features (map_encoder.py:60-75) -: x is staged as one piece, half the
     * LDS, three workgroups per CU (32-channel layers on 64-wide maps; elsewhere the promise is not used).  A value that is
     * not exact turns the whole gradient into NaNs rather than into a wrong number. */
    int split_ok;
    /* optional (D_NCHW destinations, splits forced to 1): one int32 per image of the destination (N / HoWo of them); an
     * output tile ALL of whose images carry 0 is skipped - neither computed nor stored, the destination keeps what it
     * held.  Device memory, read by the kernel: the decision replays in a hipGraph.  User: the folded attention operands of
     * the instruction (map_cma_policy.py:293, 320-325), recomputed only for rows whose tokens changed
     * (ivln_embed_gates_cached_f32).  Only the float4-staged GEMM and the scalar-gather implicit GEMM honour it: the
     * dispatcher sends such a call to one of them. */
    const int32_t* img_run_flags;
    /* optional: a ResNet bottleneck's TAIL as one launch (rednet.py:20-65: conv2 3x3 + bn2 + ReLU, conv3 1x1 + bn3 + residual +
     * ReLU).  The descriptor describes the 3x3 conv (A / A_split / M = planes = 64 or 128, stride 1, scale / shift = the folded
     * bn2, relu = 1); fuse_A_split = the 1x1 conv's (fuse_M x planes) weights as ivln_conv_split_weights_f32(KS = 1) arranges
     * them (fuse_a_grp_stride words between the sets of an image-grouped pair), fuse_scale / fuse_shift its folded bn3
     * ([g * fuse_M + m] when grouped).  D / Ctot / residual then belong to the FINAL (fuse_M-channel) tensor; the
     * planes-channel intermediate lives in LDS only.  Needs Wout % 32 == 0, Hout % 4 == 0, fuse_M % 32 == 0; anything else
     * (or IVLN_BF3_FUSE=0 in the environment): IVLN_E_UNSUPPORTED, and the caller issues the two convs. */
    const void* fuse_A_split;
    int64_t fuse_a_grp_stride;
    const float* fuse_scale;
    const float* fuse_shift;
    int fuse_M;
    /* 1: `residual` is added AFTER the ReLU - D = relu(scale * acc + shift) + residual - the form of RedNet's decoder skips
     * (rednet.py:244-263: x = deconv(x) + agant(fuse), the 1x1 "agant" conv ending in a ReLU): the separate add launch goes.
     * Taken by the stride-1 1x1 split-bf16 kernels only (csrc/conv_bf3.hip: k_conv1x1_bf3_ks); any other route returns
     * IVLN_E_UNSUPPORTED and the caller issues conv + add. */
    int residual_after_relu;
    /* IVLN_B_CONV_K2 -> IVLN_D_NCHW_UP2X4 only: how many of the 16 window taps of the four stacked parity classes are real
     * (9 for a stride-2 3x3 transposed conv; 0 = all).  Arithmetic ignores it - the zero-padded taps multiply zeros - it only
     * makes ivln_conv_split_counters tally ALGORITHMIC FLOPs instead of executed ones. */
    int real_taps;
} ivln_gemm_desc;

int ivln_gemm_f32(const ivln_gemm_desc* desc, void* stream);
/* Weights (M, Cin, KS, KS) fp32 -> the split-bf16 image ivln_gemm_desc.A_split expects: out holds
 * ivln_conv_split_words(M, Cin, KS) 4-byte words (0: KS is not 1, 3 or 7). */
int64_t ivln_conv_split_words(int M, int Cin, int KS);
int ivln_conv_split_weights_f32(const float* W, int M, int Cin, int KS, void* out, void* stream);
/* RedNet's stems (rednet.py:201-210: conv1 3 -> 64 and conv1_d 1 -> 64, 7x7, stride 2, pad 3; forward rednet.py:190-199):
 * weights (M, Cin, 7, 7) fp32, Cin 1 or 3 -> the split-bf16 image with K laid out as kernel rows x 8 columns that
 * ivln_gemm_desc.A_split must hold for a 7x7 stride-2 conv (ivln_conv_stem_split_words(M, Cin) 4-byte words; 0: not a stem).
 * With it ivln_gemm_f32 runs such a conv - Hin = 2 Hout, Win = 2 Wout, Wout a multiple of 128, no image groups - on
 * k_conv7s2_bf3 (csrc/conv_bf3.hip), residual in front of or behind the ReLU (residual_after_relu: the depth stem's output added
 * to the RGB stem's); other shapes fall through to the fp32 kernels, which ignore A_split. */
int64_t ivln_conv_stem_split_words(int M, int Cin);
int ivln_conv_stem_split_weights_f32(const float* W, int M, int Cin, void* out, void* stream);
/* Tally of the convs ivln_gemm_f32 sent to the split-bf16 kernel since the last reset: algorithmic FLOPs (2 M N K) and
 * launches (host side, at enqueue; measurement aid of bench.py - a captured graph's replays are not counted). */
int ivln_conv_split_counters(double* flops, long long* launches, int reset);
/* ... by kernel form: out4 = launches of the tiled kernel, the 3x3 K-split-over-waves kernel, the 1x1 K-split-over-waves kernel and
 * the 1x1 wave-tile kernel (csrc/conv_bf3.hip). */
int ivln_conv_split_kinds(long long* out4, int reset);

/* Duration sink of the MFMA family's launches (ivln_gemm_f32 - its split-K reduction excluded -, ivln_gn_conv_f32,
 * ivln_nconv_f32): between _begin and _end every such launch carries a start / stop event of its own
 * (hipExtLaunchKernelGGL: the dispatch's begin and end timestamps, what rocprofv3 reports per kernel).  _end waits for
 * the launches, returns the SUM of their durations and their number; `dropped` (optional) = launches beyond
 * max_launches that went out untimed.  One user at a time, not for captured streams: it is how bench.py measures
 * `roofline.achieved` live. */
int ivln_family_timing_begin(int max_launches);
int ivln_family_timing_end(double* total_ms, int* launches, int* dropped);
/* The same pass broken down by kernel: after _end, one text line "<kernel> <launches> <ms>\n" per kernel name (the
 * template's name without its arguments, e.g. k_conv1x1_bf3_ks) into buf (NUL-terminated; returns the bytes needed
 * incl. the NUL when cap is too small - call again -, < 0 on error).  ivln_family_kernel_names() = the comma-separated
 * list of every kernel that launches through the sink - THE definition of "the MFMA family" for bench.py's FLOP hooks,
 * tools/pmc_traffic.py and the per-kernel MFMA-busy table (tools/kernel_family.py mirrors it; tests pin the mirror). */
int ivln_family_timing_report(char* buf, int cap);
const char* ivln_family_kernel_names(void);
/* Pre-arrangement of OIHW conv weights (M, Cin, KS, KS), KS in {3, 7}, Cin % (KS == 7 ? 2 : 8) == 0, into the
 * LDS image of the direct convolution kernel; `out` holds ivln_conv_packed_floats(M, Cin, KS) floats (0 = shape
 * not eligible).  Re-run whenever the weights change. */
int64_t ivln_conv_packed_floats(int M, int Cin, int KS);
int ivln_conv_pack_weights_f32(const float* W, int M, int Cin, int KS, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ second normalised operand) (+ residual) (+ ReLU) (+ MaxPool2d(3, 2, 1)) fused with the NEXT
 * bias-free convolution(s) (csrc/gn_conv.hip) - the other way round from ivln_conv_gn_f32.  One workgroup per
 * (image, group) reduces the `splits` slabs of the incoming [C][N*H*W] matrix (the split-K workspace of a deferred
 * ivln_gemm_f32, or the slabs a previous ivln_gn_conv_f32 wrote: slab s at x + s*slab_stride), normalises and
 * activates its C/groups channels, and multiplies them into conv A (k = 1 | 3) and optionally conv B (1x1): the next
 * convolution is split over K by GroupNorm group, every workgroup writes partial slab `group` of
 *   ya : [groups][Cout_a][N*Ho*Wo]        yb : [groups][Cout_b][N*Hb*Wb]
 * which the next ivln_gn_conv_f32 / ivln_groupnorm_f32 (splits = groups, slab_stride = Cout*N*Ho*Wo) reduces.
 * One launch per conv layer of habitat-lab's DD-PPO ResNetEncoder (models/encoders/resnet_encoders.py:31-43, 95):
 * conv -> GroupNorm -> ReLU -> conv ..., Bottleneck tails relu(GN(c3) + GN_ds(ds)) / relu(GN(c3) + identity) feeding
 * the next block's conv1 (A) and downsample conv (B).  act_out (N, C, H', W') receives the activation when a later
 * block needs it as its identity.  IVLN_E_UNSUPPORTED: (C/groups)*H*W > 8192, odd C/groups, k not in {1, 3}, tiles +
 * weights that do not fit 152 KB of LDS. */
typedef struct ivln_gn_conv_desc {
    const float* x;       /* slabs of [C][N*H*W] */
    int splits;
    int64_t slab_stride;
    const float* gamma;   /* (C) */
    const float* beta;
    const float* x2;      /* second operand: slabs of [C][N*H*W], or NULL */
    int splits2;
    int64_t slab_stride2;
    const float* gamma2;
    const float* beta2;
    const float* residual; /* (N, C, H, W) or NULL */
    int N, C, H, W, groups;
    float eps;
    int relu;
    int pool;             /* 1: MaxPool2d(3, stride 2, padding 1) after the activation */
    float* act_out;       /* (N, C, H', W') (after the pool) or NULL */
    const float* wa;      /* (Cout_a, C, ka, ka) or NULL */
    int Cout_a, ka, stride_a, pad_a;
    float* ya;
    const float* wb;      /* (Cout_b, C, 1, 1) or NULL */
    int Cout_b, stride_b;
    float* yb;
    /* Optional front stage (x == NULL then): the GroupNorm input is itself computed in the block, two conv layers per
     * launch.  x0 = slabs of the PREVIOUS layer's raw output [C0][N*H*W]; the block normalises ALL groups0 groups of
     * its image (gamma0 / beta0, ReLU), and runs the 1x1 conv w0 (C, C0, 1, 1) for its own C/groups output channels
     * over the full K = C0 - the (image, group) tile of x that the statistics above then see.  The Bottleneck's
     * GroupNorm -> ReLU -> conv3 -> GroupNorm -> (+ identity | + downsample) -> ReLU -> next conv1 in one launch. */
    const float* x0;
    int splits0;
    int64_t slab_stride0;
    int C0, groups0;
    const float* gamma0;
    const float* beta0;
    const float* w0;
} ivln_gn_conv_desc;
int ivln_gn_conv_f32(const ivln_gn_conv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole DD-PPO depth encoder (habitat-lab ResNetEncoder: avg_pool2d(2) -> GroupNorm ResNet-50 -> compression conv ->
 * GroupNorm(1) + ReLU; models/encoders/resnet_encoders.py:31-43, 95) for 1..8 images as ONE persistent launch
 * (csrc/depth_net.hip): a cluster of 32 workgroups per image walks a table of conv "ops", GroupNorm (+ residual /
 * downsample branch, ReLU, MaxPool, the input's avg-pool) applied on load, complete outputs + GroupNorm statistics
 * partials stored per op, a counter barrier per cluster between dependent ops.  The op table, the packed weights and
 * the parameter blob are built once per model by the host (ivln-ce_amd/depth_net.py documents the packing); all
 * offsets are in floats, `*_off` into the per-image arena unless stated otherwise.
 * ------------------------------------------------------------------------------------------ */
typedef struct ivln_depthnet_op {
    int kind;            /* 0: conv op; 1: final GroupNorm(1 group) + ReLU over `nslab` slabs -> out (workgroup 0) */
    int Cin, Cout, ks, stride, pad;
    int Hin, Win;        /* input map the conv sees (after the MaxPool when pool = 1, after the avg-pool when avg_in = 1) */
    int wout_shift;      /* output map is (1 << wout_shift) squared */
    int M;               /* rows of an output-channel tile: 16, or 8 (half-filled MFMA rows) */
    int WCT, WPT, P, KW; /* a workgroup's 8 waves = WCT channel tiles x WPT pixel-tile groups x KW K ranges; P pixel tiles (of 16) per wave */
    int kwg;             /* K split over workgroups (slabs of the output; 1 = none) */
    int n_ctg, n_ptg;    /* task grid: channel-tile groups x pixel groups; n_ctg * n_ptg * kwg <= 32 */
    int ksteps;          /* K / 4: (Cin / 4) * ks * ks, or 13 for the one-channel 7x7 stem (taps padded to 52) */
    int cs, wp;          /* LDS tile: channel stride and row pitch (Win + 2 pad) in floats */
    int src_off, nslab, slab_stride;           /* raw input [Cin][Hin*Win] (pool: [Cin][2 Hin * 2 Win]); slabs summed on load */
    int st_off, st_parts;                      /* its statistics partials [16][st_parts][4] = (count, mean, M2, -); st_parts = 0: no GroupNorm */
    int gamma_off, beta_off;                   /* into the parameter blob */
    int src2_off, st2_off, st2_parts, gamma2_off, beta2_off;   /* second normalised operand (downsample branch) or src2_off = -1 */
    int res_off;         /* identity activation added after the GroupNorm(s), or -1 */
    int relu, pool, avg_in;
    int act_out_off;     /* >= 0: the transformed input is also stored here (a later block's identity) */
    int dst_off, dst_slab_stride;              /* raw output [Cout][HWout] (slab kwg_i at + kwg_i * dst_slab_stride) */
    int st_out_off, st_out_parts;              /* statistics partials of the output [16][st_out_parts][4]; 0: none */
    int w_off;           /* packed weights of this op, into the weight blob */
    int barrier_before;  /* 1: the op reads what the previous ops of the cluster stored */
} ivln_depthnet_op;
/* ops_dev / ops_host: the same table in device and host memory.  depth (N, 2 Hin0, 2 Win0) raw images, image stride
 * depth_img_stride; arena: N per-image work areas of arena_stride floats; out: image stride out_img_stride.  sync_ws: 2048
 * bytes of device memory ZEROED ONCE by the caller (a launch that completes leaves the counters zero again).  Word 256 is
 * a sticky error flag set when a bounded spin times out (~0.2 s: some workgroup of the grid never became resident); words
 * 258-259 may hold the device address of a pinned host uint32 (ivln_host_device_ptr) that receives the same flag, so that
 * the host sees a time-out at its next stream synchronisation without a read-back (word 257: test hook).  Time-out path: the
 * launch winds down WITHOUT writing valid features, its counters stay where they were, and every later launch on that
 * workspace returns at once (it checks word 256 before touching `out`) until the caller has run ivln_depth_net_reset -
 * and computed the step again on the per-layer launches (ivln_nconv_f32 / ivln_gn_conv_f32).  IVLN_E_UNSUPPORTED: N > 8, or
 * the 256 workgroups of the launch cannot all be resident for this launch's LDS footprint on this device (partitioned /
 * masked device): the caller runs the per-layer launches. */
int ivln_depth_net_f32(const ivln_depthnet_op* ops_dev, const ivln_depthnet_op* ops_host, int n_ops, const float* weights,
                       const float* params, const float* depth, int64_t depth_img_stride, float* arena, int64_t arena_stride,
                       float* out, int64_t out_img_stride, int N, float eps, void* sync_ws, void* stream);
int ivln_depth_net_status(const void* sync_ws, void* stream);
int ivln_depth_net_reset(void* sync_ws, void* stream);
int ivln_host_device_ptr(void* host, void** dev);

/* ------------------------------------------------------------------------------------------
 * Convolution with GroupNorm applied to its INPUT on load and the GroupNorm statistics of its OUTPUT emitted as
 * partials (csrc/gn_conv.hip, k_nconv) - for the large feature maps of the depth ResNet's layer 1, where the
 * 16-slab scheme of ivln_gn_conv_f32 costs more bytes than it saves launches.  One launch per conv layer:
 *   in  = act( GN(x; stats, gamma, beta) [+ GN(x2; stats2, gamma2, beta2)] [+ residual] )      (stats == NULL: in = x)
 *   ya  = conv_a(in)   (k = 1 | 3, stride 1 | 2, pad (k-1)/2),   yb = conv_b(in)   (1x1, stride 1 | 2, optional)
 * A workgroup owns a strip of `rows_per_block` output rows of one image with its halo, over ALL channels, so the
 * outputs are complete (no slabs).  stats layout: [parts][N][groups][3] = (count, mean, M2) of one (image, group)
 * per producing workgroup; the consumer merges them in part order (Chan), i.e. the variance is the two-pass one.
 * act_out (N, C, H, W) optionally receives `in` (a later block's identity).  habitat-lab ResNetEncoder Bottleneck,
 * models/encoders/resnet_encoders.py:31-43, 95.  IVLN_E_UNSUPPORTED: W > 32*k ... see gn_conv.hip (strip + weights
 * + output tile must fit 152 KB of LDS, C and Cout multiples of 2 * groups). */
typedef struct ivln_nconv_desc {
    const float* x;        /* raw (pre-GroupNorm) input [C][N][H][W] - the one-slab layout of the deferred ivln_gemm_f32 /
                              ivln_gn_conv_f32 outputs - or, when stats == NULL, the activated input (N, C, H, W) */
    const float* stats;    /* [parts][N][groups][3] or NULL */
    int parts;
    const float* gamma;
    const float* beta;
    const float* x2;       /* second raw operand [C][N][H][W] with its own statistics, or NULL */
    const float* stats2;
    int parts2;
    const float* gamma2;
    const float* beta2;
    const float* residual; /* (N, C, H, W) or NULL */
    int N, C, H, W, groups;
    float eps;
    int relu;
    float* act_out;        /* (N, C, H, W) or NULL */
    const float* wa;       /* (Cout_a, C, ka, ka) */
    int Cout_a, ka, groups_a;
    float* ya;             /* raw output [Cout_a][N][Ho][Wo] */
    float* stats_a;        /* [strips][N][groups_a][3], strips = ceil(Ho / rows_per_block); stats_b has the same strips */
    const float* wb;       /* (Cout_b, C, 1, 1) or NULL */
    int Cout_b, groups_b;
    float* yb;
    float* stats_b;
    int rows_per_block;    /* 0 = default (64 output pixels per workgroup) */
    int stride_a, stride_b; /* 0 | 1 | 2; act_out and conv B need stride_a == 1 */
} ivln_nconv_desc;
int ivln_nconv_f32(const ivln_nconv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Non-GEMM forward kernels (csrc/nn_ops.hip).  All tensors fp32 NCHW unless noted.
 * ------------------------------------------------------------------------------------------ */
/* nn.GroupNorm (+ residual add) (+ ReLU): habitat-lab ddppo resnet Bottleneck / ResNetEncoder
 * compression (reference call site models/encoders/resnet_encoders.py:31-43,95).  *_img_stride = 0
 * -> C*HW.  save_mean/save_rstd (N*groups) optional.  The input may be the split-K workspace of the
 * producing conv (splits slabs of a [C][N*HW] matrix: x_chan_stride = N*HW, x_img_stride = HW,
 * slab_stride = C*N*HW): the slab reduction is fused into the normalisation. */
int ivln_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* residual,
                       float* y, int N, int C, int HW, int groups, float eps, int relu,
                       int64_t x_img_stride, int64_t x_chan_stride, int splits, int64_t slab_stride,
                       int64_t y_img_stride, int64_t r_img_stride, float* save_mean, float* save_rstd,
                       void* stream);
/* Same, plus a second group-normalised operand added before the ReLU: y = act(GN(x) + GN2(x2) [+ residual]).
 * x2 (raw tensor or split-K slabs, same C / HW / groups) is the downsample branch of a ResNet bottleneck
 * (habitat-lab ddppo resnet Bottleneck: `out = relu(convs(x) + downsample(x))`): its GroupNorm needs no launch
 * of its own. */
int ivln_groupnorm2_f32(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                        int N, int C, int HW, int groups, float eps, int relu, int64_t x_img_stride,
                        int64_t x_chan_stride, int splits, int64_t slab_stride, int64_t y_img_stride,
                        int64_t r_img_stride, float* save_mean, float* save_rstd, const float* x2, const float* gamma2,
                        const float* beta2, int64_t x2_img_stride, int64_t x2_chan_stride, int splits2,
                        int64_t slab_stride2, void* stream);
/* nn.BatchNorm2d eval folding / train-mode batch statistics (models/encoders/map_encoder.py:13-20;
 * quirk Q6: train mode also during rollouts).  Both produce per-channel scale/shift. */
int ivln_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, const float* conv_bias /* optional, folded into shift */,
                     float eps, int C, float* scale, float* shift, void* stream);
int ivln_bn_train_stats_f32(const float* x, int N, int C, int HW, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps,
                            float* scale, float* shift, float* save_mean, float* save_rstd,
                            float* ws /* scratch, >= 3*C floats (3*C*64 for full parallelism) */,
                            int64_t ws_floats, void* stream);

/* The same outputs from the per-tile partials a conv launch left behind (ivln_gemm_desc.stat_partials: [tiles][C][3] =
 * {count, mean, M2}), merged per channel in a fixed order with Chan's formula - the statistics pass over the conv's
 * output (two reads of the whole tensor) is not needed. */
int ivln_bn_stats_from_partials_f32(const float* partials, int tiles, int C, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                    float* shift, float* save_mean, float* save_rstd, void* stream);
/* CBRA tail: relu(x*scale+shift) then AvgPool2d(2) (map_encoder.py:16-19). */
int ivln_scale_shift_relu_avgpool2_f32(const float* x, const float* scale, const float* shift, float* y,
                                       int N, int C, int H, int W, int64_t img_stride, int64_t chan_stride,
                                       int splits, int64_t slab_stride, void* stream);
/* F.avg_pool2d / nn.MaxPool2d over (NC,H,W); mode 0 = max, 1 = avg. */
int ivln_pool2d_f32(const float* x, float* y, int NC, int H, int W, int k, int s, int p, int mode,
                    void* stream);
/* SemanticMapEncoder.generate_map_features (map_encoder.py:85-90): u8 maps -> f32 (B,1+classes,cells) */
int ivln_map_features_f32(const uint8_t* occ, const uint8_t* sem, float* y, int B, int cells, int classes,
                          void* stream);
/* InstructionEncoder front end (instruction_encoder.py:70-82): tokens i64 (B,L) -> emb (B*L,E),
 * lengths i32 (B). */
int ivln_embed_lengths(const int64_t* tokens, const float* table, int B, int L, int E, int V, float* emb,
                       int* lengths, void* stream);
/* Inference fold of the same front end with the bi-LSTM's input projections (instruction_encoder.py:70-94):
 * table (V, 2G) = embedding . [W_ih ; W_ih_reverse]^T + [b_ih ; b_ih_reverse] (built by the caller with
 * ivln_gemm_f32 whenever the weights change), row_nonzero u8 (V) = the embedding row has a non-zero element.
 * tokens i64 (B,L) -> gx_f, gx_r (B*L, G) = the two halves of the token's table row, lengths i32 (B). */
int ivln_embed_gates_f32(const int64_t* tokens, const float* table, const uint8_t* row_nonzero, int B, int L, int G, int V,
                         float* gx_f, float* gx_r, int* lengths, void* stream);
/* The same with a per-row cache: the reference re-encodes an episode's instruction at every step (map_cma_policy.py:293,
 * instruction_encoder.py:72-94); the encoding is a pure function of the tokens.  cache_tokens i64 (B,L) = the tokens each
 * row encoded last (caller-owned, persistent across steps; fill with -1 to invalidate, e.g. when the weights change), dirty
 * i32 (B) = output: 1 where the row's tokens differ from the cache (the row is then encoded and the cache updated), 0 where
 * they are equal (NOTHING of the row is written: gx_f / gx_r / lengths keep their values, which therefore have to be
 * persistent buffers too).  Both NULL = ivln_embed_gates_f32. */
int ivln_embed_gates_cached_f32(const int64_t* tokens, const float* table, const uint8_t* row_nonzero, int B, int L, int G, int V,
                                float* gx_f, float* gx_r, int* lengths, int64_t* cache_tokens, int* dirty, void* stream);
/* nn.LSTM(bidirectional) over packed sequences (instruction_encoder.py:84-94): gx_* = W_ih x + b_ih
 * for all (b,t) as (B*L, 4H); out (B, 2H, L), zero for t >= lengths[b].  H must be 128. */
int ivln_lstm_bidir_fwd_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                            const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                            float* out, float* save_gates, float* save_c, void* stream);
/* The same recurrence launched with 2B * spare blocks for its 2B (sequence, direction) items: a block takes the next item
 * when it STARTS (atomic ticket in *ticket, a zeroed u32 the caller owns; the last block re-arms it), blocks past 2B
 * leave.  For a replay beside a kernel that saturates some XCDs (ivln_depth_net_f32 with fewer than 8 images): the blocks on
 * the free XCDs do all the work instead of half of it waiting for the neighbour to end.  Same results as the plain entry
 * point for every item; spare 1..8; one launch in flight per ticket word. */
int ivln_lstm_bidir_fwd_spread_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                                   const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                                   float* out, float* save_gates, float* save_c, unsigned* ticket, int spare, void* stream);
/* ... and with the per-row cache of ivln_embed_gates_cached_f32: `dirty` i32 (B) or NULL; the (sequence, direction) items
 * of a row with dirty == 0 are not run and `out` keeps that row's values of the last step it was run. */
int ivln_lstm_bidir_fwd_cached_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                                   const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                                   float* out, float* save_gates, float* save_c, unsigned* ticket, int spare, const int* dirty,
                                   void* stream);
/* The two consumers of an encoder's feature map in the MapCMA head in ONE launch (models/map_cma_policy.py:156-171,
 * 180-185, 276-296): feat (rows, C, P) contiguous ->  kv (rows, Ckv, P) = nn.Conv1d(C, Ckv, 1)  and
 * lin[r*ld_lin + o] = act(nn.Linear(C*P, O) of the flattened row).  rows <= 8 and rows*C*P*4 B <= 150 KB of LDS,
 * else IVLN_E_UNSUPPORTED (the caller runs ivln_gemm_f32 + ivln_linear_skinny_f32). */
int ivln_kv_linear_f32(const float* feat, int rows, int C, int P, const float* w_kv, const float* b_kv, int Ckv, float* kv,
                       const float* w_lin, const float* b_lin, int O, int relu, float* lin, int64_t ld_lin, void* stream);
/* nn.Linear for few rows (rollout batch): y[r][o] = act(W[o].x[r] + b[o]). */
int ivln_linear_skinny_f32(const float* x, int64_t ldx, const float* W, const float* bias, float* y,
                           int64_t ldy, int rows, int K, int O, int relu, void* stream);
/* One masked GRU step (habitat-lab RNNStateEncoder over nn.GRU; map_cma_policy.py:314-318,346-353).
 * x (rows,I) or gi_pre (rows,3H) = W_ih x + b_ih precomputed; h_in rows with stride ldh; mask u8. */
int ivln_gru_step_f32(const float* x, int64_t ldx, int I, const float* gi_pre, int64_t ldgi,
                      const float* h_in, int64_t ldh, const uint8_t* mask, const float* w_ih,
                      const float* w_hh, const float* b_ih, const float* b_hh, float* h_out, int64_t ldo,
                      float* h_out2, int64_t ldo2, int rows, int H, float* save_r, float* save_z,
                      float* save_n, float* save_ghn, void* stream);
/* The masked GRU over a whole time-major (T*N rows) sequence batch in ONE call (habitat-lab RNNStateEncoder
 * seq_forward; BPTT forward of base_il_trainer.py:173-219): gi = W_ih x + b_ih for all rows (caller's GEMM), h0 (N, H)
 * row stride ld_h0, masks u8 (T*N) -> out (T*N, H) row stride ldo, state_out (N, H) = the last step; optional saves
 * r / z / n / gh_n (T*N, H) for ivln_cma_seq_bwd_f32.
 * sync_ws: 256 bytes of device memory owned by the caller (one per stream), or NULL.  The caller ZEROES the 256 bytes
 * once after allocating them (ivln_seq_sync_init, or any memset): every launch clears the step counters (bytes 0..191)
 * itself, but byte 192 is a STICKY error word that only a timed-out spin ever writes and nothing in the library
 * clears - ivln_seq_sync_status on a never-zeroed workspace reports a timeout that did not happen.  With a workspace
 * and a shape inside ivln_cma_seq_persistent_ok the whole sequence is ONE persistent launch (csrc/gru_seq.hip: W_hh
 * resident in registers over 64 workgroups of 256 threads - 32 of 512 with IVLN_SEQ_UPB=16 -, h_t exchanged through
 * `out` with write-through stores and one counter per step).  The single launch is taken only when the whole grid can
 * be resident at once (occupancy query x CU count >= grid: not on a 32-CU partition or with N so large that LDS admits
 * fewer workgroups than the grid) and `out` / h0 / dgh are 16-byte aligned (they are read with 16-byte buffer loads);
 * otherwise, and always without a workspace, T dependent per-step launches are enqueued.  Both paths compute the same
 * values (2e-7 apart: different reduction tree).  After a timeout the run is unrecoverable: outputs of that launch are
 * undefined and the word stays set. */
int ivln_cma_seq_fwd_f32(const float* gi, const float* h0, int64_t ld_h0, const uint8_t* masks, const float* w_hh,
                         const float* b_hh, float* out, int64_t ldo, float* state_out, int64_t ld_so, int T, int N,
                         int H, float* save_r, float* save_z, float* save_n, float* save_ghn, void* sync_ws,
                         void* stream);
/* 1 when the single-launch path serves (N, H) (backward != 0: the BPTT kernel's envelope). */
int ivln_cma_seq_persistent_ok(int N, int H, int backward);
/* Zeroes a 256-byte sync workspace on `stream` (counters and the sticky error word): call once per workspace. */
int ivln_seq_sync_init(void* sync_ws, void* stream);
/* Synchronises `stream` and returns IVLN_OK, or IVLN_E_HIP when a bounded spin of the last persistent launch that
 * used `sync_ws` timed out (its outputs are then undefined; the launch itself always terminates). */
int ivln_seq_sync_status(const void* sync_ws, void* stream);
/* MapCMANet._attn (map_cma_policy.py:266-274); k (rows,Ck,I), v (rows,Cv,I) channel-major. */
int ivln_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t k_img_stride, const float* v,
                      int64_t v_img_stride, const int* valid_len, float scale, int rows, int Ck, int Cv,
                      int I, float* out, int64_t ldo, float* save_attn, float* logits_ws /* rows*I */,
                      void* stream);
/* Same, with row_index i32 (rows): row r attends over key/value image row_index[r] (valid_len is per image).  Update
 * batches are time-major T*N rows whose instruction is the same at every timestep of a trajectory: the instruction
 * encoder runs once per UNIQUE token row and T rows share its output. */
int ivln_attn_fwd_idx_f32(const float* q, int64_t ldq, const float* k, int64_t k_img_stride, const float* v,
                          int64_t v_img_stride, const int* valid_len, float scale, int rows, int Ck, int Cv, int I,
                          float* out, int64_t ldo, float* save_attn, float* logits_ws, const int* row_index,
                          void* stream);
/* The same attention for a short key axis (I <= 32: the 4x4 depth / map feature grids of MapCMANet.forward,
 * map_cma_policy.py:330-343) and up to two key/value sets that share the query, in ONE launch; k1 == NULL ->
 * one set.  No mask, no saved probabilities (rollout head). */
int ivln_attn_small2_f32(const float* q, int64_t ldq, float scale, int rows, int I, const float* k0,
                         int64_t k0_img_stride, const float* v0, int64_t v0_img_stride, int Ck0, int Cv0, float* out0,
                         int64_t ldo0, const float* k1, int64_t k1_img_stride, const float* v1, int64_t v1_img_stride,
                         int Ck1, int Cv1, float* out1, int64_t ldo1, void* stream);
/* prev_action_embedding(((a+1)*mask).long()) (map_cma_policy.py:297-299), written to two slices. */
int ivln_prev_action_embed_f32(const int64_t* prev_actions, const uint8_t* mask, const float* table,
                               int rows, int E, int n_emb, float* out1, int64_t ld1, float* out2,
                               int64_t ld2, void* stream);
/* distribution.mode() (common/utils.py:168-169) and predicted_scores.argmax(1) (mapper.py:796-798). */
/* CategoricalNet + distribution.mode() of a deterministic step in one launch (models/policy.py:35-48,
 * common/utils.py:149-185): logits = W x + b (O <= 8 actions), action[r] = first arg-max; logits_out optional
 * (rows, O). */
int ivln_linear_argmax_f32(const float* x, int64_t ldx, const float* W, const float* bias, int rows, int K, int O,
                           int64_t* action, float* logits_out, void* stream);
/* The SAMPLED action of a DAgger collection step in one launch (models/policy.py:28-46 with deterministic=False;
 * dagger_trainer.py:416-427, 469-472): a ~ softmax(W x + b) by inverse CDF with the caller's uniform u_sample[r]
 * (p_o = exp(l_o - max); the first o whose running sum exceeds u * sum), then - when `expert` (rows, f64: the
 * shortest-path sensor as batched) is given - `where(u_beta[r] < beta, expert[r], a)` (u_beta NULL: no mixing) and 0
 * where expert[r] == -1.  A pure function of its inputs, so the step can be captured in a hipGraph. */
int ivln_linear_sample_f32(const float* x, int64_t ldx, const float* W, const float* bias, int rows, int K, int O,
                           const float* u_sample, const float* u_beta, float beta, const double* expert,
                           int64_t* action, float* logits_out, void* stream);
/* Tour-long memory slot of the Latent-CMA `tour_memory_variant` (models/latent_cma_policy.py:395-399, 433-439):
 * out[n] = mask[n] * (h ? max(mem[n], h[n]) : mem[n]) - the previous step's max-pool with the first GRU's new
 * state and this step's reset where a tour starts, in one launch; written to out1 and (optionally) out2, all
 * row-strided (N, H) views. */
int ivln_tour_memory_f32(const float* mem, int64_t ld_mem, const float* h, int64_t ld_h, const uint8_t* mask, int N,
                         int H, float* out1, int64_t ld1, float* out2, int64_t ld2, void* stream);
/* u8 NHWC -> f32 NCHW / div (TorchVisionResNet.forward, models/encoders/resnet_encoders.py:171-198: / 255) */
int ivln_rgb_to_nchw_f32(const uint8_t* rgb, int B, int H, int W, float div, float* out, void* stream);
/* F.adaptive_avg_pool2d (SpatialAvgPool -> 4x4, resnet_encoders.py:152-158); out_img_stride lets `out` be a
 * channel slice of a wider NCHW buffer (0 -> C*OH*OW) */
int ivln_adaptive_avgpool2d_f32(const float* x, int N, int C, int H, int W, int OH, int OW, float* out,
                                int64_t out_img_stride, void* stream);
int ivln_argmax_rows(const float* x, int rows, int C, int64_t* out, void* stream);
int ivln_argmax_channels_u8(const float* x, int N, int C, int HW, uint8_t* out, void* stream);
/* PredictSemantics input prep (mapper.py:715-736,788-793). */
int ivln_rgb_resize_normalize_f32(const uint8_t* rgb, int B, int Hi, int Wi, int Ho, int Wo, float* out,
                                  void* stream);
int ivln_affine_f32(const float* x, float* y, int64_t n, float sub, float div, void* stream);
int ivln_add_f32(const float* a, const float* b, float* y, int64_t n, int relu, void* stream);
/* n <= 8 contiguous device-to-device copies in one launch (GraphedRollout.load: the observation tensors of
 * an env step -> the captured input buffers of the step graph; replaces per-tensor Tensor.copy_ at
 * base_il_trainer.py:688-703's batch_obs hand-over). */
int ivln_copy_multi(const void* const* srcs, void* const* dsts, const int64_t* bytes, int n, void* stream);
/* dsts[j][i] += srcs[j][i] (counts[j] floats, n <= 64 contiguous tensors) in one launch: the parameter gradients of
 * one backward pass added onto the flat gradient bucket (`loss.backward()` accumulation, base_il_trainer.py:211). */
int ivln_add_multi_f32(const float* const* srcs, float* const* dsts, const int64_t* counts, int n, void* stream);
/* Up to 32 column sums out[j][c] = sum_r x[j][r*ld[j] + c] in two launches (the bias gradients of one DAgger update:
 * autograd's `grad_output.sum(0)` of every nn.Linear / GRU / LSTM gate matrix, base_il_trainer.py:211); partials and
 * sums in the fixed order of ivln_colsum_f32 (bit-identical to calling it per matrix).  The arrays are host arrays;
 * ws holds sum_j splits_j * cols_j floats (splits_j = min(128, ceil(rows_j / 256))). */
int ivln_colsum_multi_f32(const float* const* xs, const int64_t* lds, const int* rows, const int* cols,
                          float* const* outs, int n, float* ws, int64_t ws_floats, void* stream);
int ivln_copy2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols,
                    int broadcast_rows, void* stream);

/* RedNet forward as ONE C call (PredictSemantics.forward, mapper.py:781-800, over
 * RedNet.forward, mapping_module/rednet.py:190-263): the caller hands a packed table of the launches one forward is
 * made of - built once per (weights, batch shape) with every weight, BN fold, activation buffer and split-K workspace
 * pointer resolved - and the library walks it: no host code between the ~170 launches.  Table entries:
 *   IVLN_OP_GEMM       `gemm` as ivln_gemm_f32 takes it (every conv / transposed-conv class, fused BN + ReLU +
 *                      residual epilogues)
 *   IVLN_OP_ADD        dst = src0 + src1 (n floats, relu = i[0])               the encoder fusion / decoder skip adds
 *   IVLN_OP_POOL       ivln_pool2d_f32(src0, dst, NC = i[0], H, W, k, s, p, mode = i[6])
 *   IVLN_OP_RGB_NORM   ivln_rgb_resize_normalize_f32(src0 | the call's `rgb`, B = i[0], Hi, Wi, Ho, Wo, dst)
 *   IVLN_OP_AFFINE     dst = (src0 | the call's `depth` - f[0]) / f[1]   (n floats)
 *   IVLN_OP_ARGMAX_U8  ivln_argmax_channels_u8(src0, N = i[0], C, HW, dst | the call's `labels_out`)
 * A NULL src0 of RGB_NORM / AFFINE and a NULL dst of ARGMAX_U8 stand for the per-call arguments, so one table serves
 * every step.  Returns the first failing launch's status. */
enum { IVLN_OP_GEMM = 0, IVLN_OP_ADD = 1, IVLN_OP_POOL = 2, IVLN_OP_RGB_NORM = 3, IVLN_OP_AFFINE = 4, IVLN_OP_ARGMAX_U8 = 5 };
typedef struct ivln_rednet_op {
    int32_t kind;
    int32_t i[7];
    float f[2];
    int64_t n;
    const void* src0;
    const void* src1;
    void* dst;
    ivln_gemm_desc gemm;
} ivln_rednet_op;
int ivln_rednet_fwd(const ivln_rednet_op* table, int n_ops, const uint8_t* rgb, const float* depth, uint8_t* labels_out,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused recurrent / attention head of one rollout step (csrc/cma_step.hip): everything MapCMANet.forward does
 * after its encoders (ivlnce_baselines/models/map_cma_policy.py:305-353, `_attn` :266-274; two single-step
 * habitat-lab RNNStateEncoders):  GRU-1 -> state_q / text attention -> text_q / depth + map attention ->
 * second_state_compress -> GRU-2, as five dependent phase kernels enqueued by this ONE call (`mode` is reserved, pass 0;
 * a persistent single-launch form was measured slower and dropped, see csrc/cma_step.hip).  The caller supplies the
 * instruction-only folds (see the file header):
 *   Mq  (rows, H+1, L): rows 0..H-1 = W_q^T text_k, row H = b_q . text_k      (image stride Mq_img floats)
 *   TQb (rows, Hq, L) = W_tq txt + b_tq                                        (image stride TQb_img)
 * and the per-step operands: state_in (rows, d_out+m_out+E) = [ReLU(depth_linear) | ReLU(map_linear) | prev-action
 * embedding], h_in (rows, 2, H) with row stride ld_h, mask u8 (rows), txt (rows, Ct, L), lengths i32 (rows),
 * dkv (rows, Hq+d_out, P) / mkv (rows, Hq+m_out, P) = the dep_kv / map_kv projections (keys first).
 * Outputs: x2 (rows, x2w) = [state | text | dep' | map' | prev] (the prev slice is the caller's), h_out (rows, 2, H)
 * row stride ld_ho, feats (rows, H).  ws: ivln_cma_step_ws_floats() floats of scratch (128-byte aligned).
 * IVLN_E_UNSUPPORTED outside L <= 512, P <= 16, 64-aligned widths. */
typedef struct ivln_cma_step_desc {
    int rows, L, P, H, Hq, Ct, d_out, m_out, E, x2w;
    const float* state_in;
    const float* h_in;
    int64_t ld_h;
    const uint8_t* mask;
    const float *w_ih1, *w_hh1, *b_ih1, *b_hh1; /* state_encoder.rnn  (3H x (d_out+m_out+E), 3H x H) */
    const float* Mq;
    int64_t Mq_img;
    const int* lengths;
    const float* txt;
    const float* TQb;
    int64_t TQb_img;
    const float* dkv;
    const float* mkv;
    float scale;
    const float *w_c, *b_c;                     /* second_state_compress.0 (H x x2w) */
    const float *w_ih2, *w_hh2, *b_ih2, *b_hh2; /* second_state_encoder.rnn (3H x H twice) */
    float* x2;
    float* h_out;
    int64_t ld_ho;
    float* feats;
    float* ws;
} ivln_cma_step_desc;
int64_t ivln_cma_step_ws_floats(int rows, int L, int P, int H);
int ivln_cma_step_fwd(const ivln_cma_step_desc* d, int mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward / loss / optimizer kernels of the DAgger update (csrc/train_ops.hip): replace the
 * autograd backward of the modules above plus F.cross_entropy / AuxLosses / torch.optim.Adam in
 * BaseVLNCETrainer._update_agent (ivlnce_baselines/common/base_il_trainer.py:173-219).
 * ------------------------------------------------------------------------------------------ */
int ivln_relu_bwd_f32(const float* dy, const float* y, float* dx, int rows, int cols, int64_t ld_dy,
                      int64_t ld_y, int64_t ld_dx, void* stream);
int ivln_add2d_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int rows,
                   int cols, void* stream);
/* deterministic column sums (bias gradients); ws: scratch of >= 128*cols floats */
int ivln_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, float* ws,
                    int64_t ws_floats, void* stream);
int ivln_nchw_chansum_f32(const float* x, int N, int C, int HW, float* out, float* ws, int64_t ws_floats,
                          void* stream);
int ivln_transpose_f32(const float* x, float* y, int R, int C, void* stream);
/* W (O,I,k,k) -> (I,O,k,k) spatially flipped: conv dgrad = conv(dy, W', pad = k-1-pad) */
int ivln_weight_flip_transpose_f32(const float* w, float* wt, int O, int I, int KH, int KW, void* stream);
/* backward of ivln_attn_fwd_f32 (MapCMANet._attn, map_cma_policy.py:266-274) */
int ivln_attn_bwd_f32(const float* dout, int64_t ld_dout, const float* attn, const float* q, int64_t ldq,
                      const float* k, int64_t k_img_stride, const float* v, int64_t v_img_stride, float scale,
                      int rows, int Ck, int Cv, int I, float* dq, int64_t ld_dq, float* dk,
                      int64_t dk_img_stride, float* dv, int64_t dv_img_stride, void* stream);
/* the same with shared key/value images (ivln_attn_fwd_idx_f32): k / v are read through row_index, dk / dv are still
 * written per ROW; ivln_index_sum_f32 then folds them onto the images:
 *   dst[u][:] = sum_{r : index[r] == u} src[r][:]   (ascending r; M % 4 == 0, 16-byte aligned) */
int ivln_attn_bwd_idx_f32(const float* dout, int64_t ld_dout, const float* attn, const float* q, int64_t ldq,
                          const float* k, int64_t k_img_stride, const float* v, int64_t v_img_stride, float scale,
                          int rows, int Ck, int Cv, int I, float* dq, int64_t ld_dq, float* dk,
                          int64_t dk_img_stride, float* dv, int64_t dv_img_stride, const int* row_index, void* stream);
int ivln_index_sum_f32(const float* src, const int* index, int rows, int64_t M, int U, float* dst, void* stream);
/* one BPTT step of the masked GRU: gate gradients (element part) */
int ivln_gru_bwd_elem_f32(const float* dout, int64_t ld_dout, const float* dh_carry, const float* r,
                          const float* z, const float* n, const float* ghn, const float* h_prev, int64_t ldh,
                          const uint8_t* mask, int rows, int H, float* dgi, float* dgh, float* dhz,
                          float* hp_out, void* stream);
/* one full BPTT step of the masked GRU in one launch: dh_prev = (dgh_t . W_hh + dhz) * mask_t (whh_t = W_hh^T,
 * (H, 3H)) followed by the element part of step t-1 (ivln_gru_bwd_elem_f32 with that carry) on the same
 * hidden unit; dhz (rows, H) is read (step t) and overwritten (step t-1) in place.  r/z/n/ghn, h_prev,
 * mask_prev, dout_prev and the outputs are the step t-1 slices. */
int ivln_gru_bwd_step_f32(const float* dgh_t, int64_t ld_dgh, const float* whh_t, const uint8_t* mask_t,
                          const float* dout_prev, int64_t ld_dout, const float* r, const float* z, const float* n,
                          const float* ghn, const float* h_prev, int64_t ldh, const uint8_t* mask_prev, int rows, int H,
                          float* dhz, float* dgi_prev, float* dgh_prev, float* hp_prev, void* stream);
/* BPTT of ivln_cma_seq_fwd_f32 in one call (whh_t = W_hh^T (H, 3H)): d_out (T*N, H) row stride ld_dout, the forward's
 * saves and outputs -> dgi, dgh (T*N, 3H), hp = h_prev * mask (T*N, H); dhz (N, H) scratch of the per-step path.
 * sync_ws as in the forward (one persistent launch, dgh rows exchanged write-through) or NULL (T launches). */
int ivln_cma_seq_bwd_f32(const float* d_out, int64_t ld_dout, const float* r, const float* z, const float* n,
                         const float* ghn, const float* out, int64_t ld_out, const float* h0, int64_t ld_h0,
                         const uint8_t* masks, const float* whh_t, int T, int N, int H, float* dgi, float* dgh, float* hp,
                         float* dhz, void* sync_ws, void* stream);
/* y[r][o] = (W[o].x[r] + add[r][o]) * (rowmask[r] != 0)  (dh_prev of the GRU BPTT) */
int ivln_linear_skinny_ex_f32(const float* x, int64_t ldx, const float* W, const float* add, int64_t ld_add,
                              const uint8_t* rowmask, float* y, int64_t ldy, int rows, int K, int O,
                              void* stream);
/* BPTT of ivln_lstm_bidir_fwd_f32: dout (B,2H,L) -> dgx_* (B*L,4H), hprev_* (B*L,H) */
int ivln_lstm_bidir_bwd_f32(const float* dout, const float* out, const float* gates, const float* cs,
                            const float* whh_f, const float* whh_r, const int* lengths, int B, int L, int H,
                            float* dgx_f, float* dgx_r, float* hprev_f, float* hprev_r, void* stream);
/* backward of BatchNorm2d(train|eval) -> ReLU -> AvgPool2d(2) (CBRA, map_encoder.py:13-20).  ws: scratch of at least
 * 4 * C + 2 floats (the per-channel sums are accumulated and merged in double: 4 floats per (channel, split)). */
int ivln_cbra_bwd_f32(const float* dout, const float* y, const float* scale, const float* shift,
                      const float* mean, const float* rstd, int N, int C, int H, int W, int train,
                      float* dgamma, float* dbeta, float* dy, float* ws, int64_t ws_floats, void* stream);
int ivln_embedding_scatter_add_f32(const int64_t* tokens, const float* d, int rows, int E, int V,
                                   int padding_idx, float* grad, void* stream);
int ivln_prev_action_embed_bwd_f32(const int64_t* prev_actions, const uint8_t* mask, const float* d1,
                                   int64_t ld1, const float* d2, int64_t ld2, int rows, int E, int n_emb,
                                   float* grad, void* stream);
/* inflection-weighted cross entropy + its gradient (base_il_trainer.py:201-204); logits (T,N,A) */
int ivln_ce_iw_loss_f32(const float* logits, const int64_t* targets, const float* weights, int T, int N, int A,
                        float loss_scale, float* loss_out, float* dlogits, void* stream);
/* progress-monitor loss with the reference's (TN,)x(TN,1) broadcast (map_cma_policy.py:355-361) */
int ivln_pm_loss_fwd_f32(const float* pre, const float* progress, int n, float* hat, float* loss_matrix,
                         void* stream);
int ivln_pm_loss_bwd_f32(const float* dL, const float* hat, const float* progress, int n, float* dpre,
                         void* stream);
/* The same loss as the update consumes it (AuxLosses.reduce, aux_losses.py:22-29): the mean over the entries the (n,)
 * u8 mask selects (whole columns i) of L[j][i] = (tanh(pre_i) - progress_j)^2, without materialising the matrix.
 * out2[0] = mean, out2[1] = number of selected entries; hat / dsum (n) are saved for the backward, which returns
 * dpre = gout[0] * alpha * d(mean)/d(pre) (gout: the upstream gradient, a device scalar). */
int ivln_pm_masked_mean_fwd_f32(const float* pre, const float* progress, const uint8_t* mask, int n, float* hat,
                                float* dsum, float* out2, void* stream);
int ivln_pm_masked_mean_bwd_f32(const float* gout, const float* hat, const float* dsum, const uint8_t* mask,
                                const float* out2, int n, float alpha, float* dpre, void* stream);
/* torch.optim.Adam step on a flat fp32 bucket (base_il_trainer.py:78-94, 213-215); seg_of/seg_lr give
 * per-segment learning rates (SEMANTIC_MAP_ENCODER.custom_lr) or NULL; zero_grad clears the grads. */
int ivln_adam_step_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       const int* seg_of, const float* seg_lr, float beta1, float beta2, float eps, int step,
                       float grad_scale, int zero_grad, void* stream);
/* The same with a device-side guard: *guard != 0 (e.g. word 48 of a sequence-GRU sync workspace: a bounded spin of this
 * update's persistent launch timed out, its gradients are void) makes the launch do NOTHING - parameters, moments and
 * gradients keep their values - so that the host can run the update again after it has seen the error at its next
 * synchronisation.  guard NULL = ivln_adam_step_f32. */
int ivln_adam_step_guarded_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               const int* seg_of, const float* seg_lr, float beta1, float beta2, float eps, int step,
                               float grad_scale, int zero_grad, const void* guard, void* stream);

/* Host-side (CPU) windowed DTW, step pattern symmetric1 (dtw-python 1.3.0 semantics; call site
 * habitat_extensions/tour_ndtw.py:118-124).  a (n,dim), b (m,dim) HOST doubles; window (n,m) u8 or NULL. */
int ivln_dtw_symmetric1(const double* a, int n, const double* b, int m, int dim, const uint8_t* window,
                        double* distance_out);

#ifdef __cplusplus
}
#endif
#endif
