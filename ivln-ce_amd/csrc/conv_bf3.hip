// Direct stride-1 3x3 / 7x7 convolution with fp32 operands carried as THREE bf16 pieces each, on the bf16 MFMA pipe
// (gfx950: v_mfma_f32_32x32x16_bf16 issues 16x the FLOPs of v_mfma_f32_32x32x2_f32 per cycle).
//
//   x = x1 + x2 + x3: x1 = bf16(x) (round to nearest even), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); the remainders are
//   exact in fp32, |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|, and what the three pieces miss is below 2^-25 |x| - an fp32
//   significand is 24 bits, bf16 has fp32's exponent range (below 2^-110 the last piece runs out of exponent; infinities
//   and NaNs travel in x1 alone: a non-finite input never gives a finite output, but an infinity can come out as NaN).
//   a * b = (a1 + a2 + a3)(b1 + b2 + b3): of the nine piece products the kernel issues the six of relative size >= 2^-16 -
//   a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 - each exact in the MFMA (8 x 8 significand bits) and accumulated in fp32 like
//   the fp32 MFMA accumulates its products; the three it drops (a2b3, a3b2 <= 2^-24 |a b| each, a3b3) sum to less than
//   2^-23 |a b| - the size of fp32's own rounding of the product.  Measured against float64 the result is as close as the
//   fp32 MFMA kernel's (tests/test_gpu_kernels.py: both within the same bar, error tables in DESIGN.md section 3).
//   Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2: 192 instead of 512 pipe cycles per 16 channels x 1 tap.
//
// Users: the map CNN's four 7x7 convs and their input gradients (map_encoder.py:8-97 under base_il_trainer.py:173-219),
// RedNet's 3x3 convs (rednet.py:190-358).  Data flow per workgroup (512 threads = 8 waves, 2 per SIMD):
//   * tile = BM output channels x BN output pixels (IMGS images x PTH x PTW), waves WM x WN, wave tile (32 TM) x 64;
//   * the input patch of 16 channels is staged ONCE per chunk in LDS, already split, channel-last: a pixel is a 112-byte
//     record [piece][16 channels] (28 words: the 16 lanes of a ds_read_b128 phase hit 16 distinct bank quads), so an
//     operand fetch for tap (kh, kw) is `ds_read_b128 v, base offset:imm` with a per-lane base that never changes;
//   * the weights arrive split and in the MFMA's per-lane order from ivln_conv_split_weights_f32 and stream global ->
//     registers, DA taps ahead, through a buffer resource (loads through a plain pointer are sunk to their first use);
//   * the next chunk's patch values fly under the MFMA phase (registers), are split and written after the barrier;
//   * epilogue through LDS in slabs of 32 channels: 16-byte stores along the pixel index with the fused scale / shift /
//     residual / ReLU of ivln_gemm_f32 and the per-(128-pixel segment, channel) Welford partials of stat_partials.
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"
#include "residency.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int PIXB = 112;   // bytes per staged pixel: 3 pieces x 16 channels x 2 bytes + 16 of padding
constexpr int CB = 16;      // input channels per chunk = K of one MFMA

__host__ __device__ constexpr int bf3_taps_padded(int KS) { return KS == 7 ? 54 : (KS == 3 ? 9 : (KS == 2 ? 4 : 1)); }  // multiple of every prefetch depth used (3, 6 | 3, 9 | 2, 4)
// KS == 2: the 2 x 2 window of the stacked output-parity classes of a stride-2 3x3 transposed conv (IVLN_B_CONV_K2 ->
// IVLN_D_NCHW_UP2X4, rednet.py:152-181: taps at input offsets 0..1, pad 0, the row / column past the edge reads as zero; rows
// m = 4 * channel + class, a row's pixel (ho, wo) is output pixel (2 ho + a, 2 wo + b) of channel m / 4).
// Patch geometry of a (PTH x PTW) tile: rows ho0 - pad .. + PTH + KS - 2; columns on a grid of aligned 16-byte groups that
// starts bf3_gx0(KS) pixels left of the tile (odd kernels need the left halo's group, the 2 x 2 window does not).
__host__ __device__ constexpr int bf3_gx0(int KS) { return KS == 2 ? 0 : 4; }
__host__ __device__ constexpr int bf3_xoff(int KS) { return KS == 2 ? 0 : 4 - KS / 2; }  // patch column x = pixel x + XOFF of the group grid
__host__ __device__ constexpr int bf3_stage_chunks(int KS) { return KS == 1 ? 4 : 1; }  // 16-channel chunks staged per barrier pair (1x1: one tap per chunk)

// x -> the upper 16 bits of its three pieces (see the header): round-to-nearest-even at each step, remainders exact.
__device__ __forceinline__ uint32_t bf16_rne_bits(float v, bool& fin) {
    const uint32_t u = __float_as_uint(v);
    fin = (u & 0x7F800000u) != 0x7F800000u;
    uint32_t hb = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    if (fin && (hb & 0x7F800000u) == 0x7F800000u) hb = u & 0xFFFF0000u;  // (next to FLT_MAX: do not round a finite value to infinity)
    if (!fin) hb = (u & 0xFFFF0000u) | ((u & 0x007FFFFFu) ? 0x00400000u : 0u);  // infinity as it is; a NaN stays a NaN
    return hb;
}
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    bool fin, f2;
    const uint32_t hb = bf16_rne_bits(x, fin);
    const float r = fin ? __fsub_rn(x, __uint_as_float(hb)) : 0.f;  // (infinities and NaNs travel in the first piece alone)
    const uint32_t mb = bf16_rne_bits(r, f2);
    const float r2 = __fsub_rn(r, __uint_as_float(mb));
    const uint32_t lb = bf16_rne_bits(r2, f2);
    h = hb >> 16;
    m = mb >> 16;
    l = lb >> 16;
}

// Two values at once, on v_cvt_pk_bf16_f32 (round to nearest even, two floats -> one packed word: first value in the low half):
// 11 VALU operations per pair and piece set instead of ~40.  What the staging passes use; non-finite values: the first piece
// carries them, the remainders turn NaN (the header's "an infinity can come out as NaN"), and a finite value within 2^-9 of
// FLT_MAX rounds its first piece to infinity.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float v0, float v1, uint32_t& H, uint32_t& M, uint32_t& L) {
    H = cvt_pk_bf16(v0, v1);
    const float r0 = __fsub_rn(v0, __uint_as_float(H << 16)), r1 = __fsub_rn(v1, __uint_as_float(H & 0xFFFF0000u));
    M = cvt_pk_bf16(r0, r1);
    const float q0 = __fsub_rn(r0, __uint_as_float(M << 16)), q1 = __fsub_rn(r1, __uint_as_float(M & 0xFFFF0000u));
    L = cvt_pk_bf16(q0, q1);
}

#ifdef BF3_TIMING  // tools/conv_bf3_phases.py: per-workgroup phase sums (100 MHz wall clock): prologue, staging, MFMA, epilogue
__device__ unsigned long long g_bf3_stamp[8192 * 8];  // records of 8 words
#define BF3_T() (threadIdx.x == 0 ? wall_clock64() : 0ull)
#else
#define BF3_T() 0ull
#endif

// Behind a 16-byte buffer store issued straight from computed registers: four wait states, pinned in place, before anything
// may write the store's data registers again.  The store unit reads its data a few cycles after issue, 16 lanes at a time;
// the compiler pads for that only in the cases its hazard table lists, and on this part a VALU write right behind such a
// store (scalar channel offset in soffset) was seen to land first in lanes 48-63 - one register of one store stale, once in
// a few thousand workgroups, run-to-run different (tools/dbg_fuse.py: the fused bottleneck tail against the two launches).
#ifdef BF3_NO_STORE_GUARD  // (tools/check_store_hazard.py's self-test: the checker has to find these sites unguarded)
#define BF3_STORE_GUARD() do {} while (0)
#else
#define BF3_STORE_GUARD()                       \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("s_nop 3" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf3_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// ST = 2 (round 6): the stride-2 3x3 convs of RedNet's layer 2-4 entry blocks (rednet.py:84-110, pad 1, Hin = 2 Hout).  Output
// (ho, wo) reads input rows 2 ho - 1 .. 2 ho + 1: the patch is staged as FOUR PHASE PLANES - (row parity, column parity) of the
// input pixel, each (PTH + 1) x (PTW + 1) pixels per image - so that tap (kh, kw) is plane (kh != 1, kw != 1) at offset
// ((kh == 2), (kw == 2)): a compile-time LDS offset per tap, and a lane's base address depends on its output pixel only,
// exactly as in the stride-1 kernel.  Odd input rows 2 i - 1 are plane row i (i = 0 .. PTH), even rows 2 i plane row i.
template <int KS, int TM, int WM, int WN, int PTH, int PTW, int IMGS, int DA, bool FUSE = false, int ST = 1>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_conv_bf3(const ivln_gemm_desc p, const unsigned char* a_split, long long a_grp_bytes,
                                                           int tiles_w, int tiles_h, int nimg, int chunks_per_split) {
#ifdef BF3_TIMING
    const unsigned long long t_entry = threadIdx.x == 0 ? wall_clock64() : 0ull;
#endif
    constexpr int NTB = 64 * WM * WN, TN = 2;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static_assert(IMGS * PTH * PTW == BN && PTW % 4 == 0, "pixel tile");
    constexpr int KK = KS * KS, KKP = bf3_taps_padded(KS), CS = bf3_stage_chunks(KS);
    constexpr int RP = KKP * CS;  // taps (padded) between two barriers: a STAGE = CS chunks of 16 channels
    static_assert(RP % DA == 0 && KKP >= KK && (CS == 1 || KK == 1), "prefetch rotation closes over a stage");
    static_assert(ST == 1 || (ST == 2 && KS == 3 && !FUSE), "stride 2: the 3x3 kernel only");
    constexpr bool S2 = ST == 2;
    // stride 1: PH x PWR patch pixels per image.  stride 2: four planes of PH x PWR = (PTH + 1) x (PTW + 1) each
    constexpr int PH = S2 ? PTH + 1 : PTH + KS - 1, PWR = S2 ? PTW + 1 : PTW + KS - 1;
    constexpr int PLANE = PH * PWR, NPIX = IMGS * (S2 ? 4 : 1) * PLANE;
    constexpr int ITEMS = NPIX * (CB / 2) * CS;
    constexpr int LDT = BN + 4;
    constexpr bool HAS_LITE = KS == 7 && TM == 1;  // (a second copy of the unrolled tap loop: only where one-hot inputs occur)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // (uniform: the weight stream's buffer resource lives in SGPRs)
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int bx = bid.x;
    const int tw = bx % tiles_w, th = (bx / tiles_w) % tiles_h, ig = bx / (tiles_w * tiles_h);
    const int img0 = ig * IMGS, ho0 = th * PTH, wo0 = tw * PTW;
    const int m0 = bid.y * BM;
    const int nch = (p.Cin + CB - 1) / CB;
    const int HW = p.Hin * p.Win;
    const int grp = p.grp_imgs > 0 ? img0 / p.grp_imgs : 0;

    // Staging work of a thread.
    // 3x3 / 7x7 (round 5): an ITEM = one aligned 16-byte group of four pixels of a patch row x one channel pair - two
    // buffer_load_b128 (one per channel) through an SGPR descriptor with the chunk's channel base in the scalar offset.  W is
    // a multiple of 4 and the groups start at multiples of 4 pixels of the image row, so a group lies wholly inside the
    // row or wholly outside: what is outside (halo past the image, images past the batch, channels past Cin) gets an
    // out-of-range offset and the hardware returns zeros - no selects, no per-element validity.  Item -> lane: the pair
    // index fastest, then the group (the LDS writes of 32 lanes then hit 16 banks twice: free; a wave's loads touch whole
    // 128-byte lines).  Six loads per thread and chunk where the first version issued 32 four-byte loads with ~10 address
    // operations each, and - the loads being unconditional - the compiler knows how many are in flight at every tap: the
    // first version's `if (c + 1 < c_end) load_patch` made it wait for the NEXT chunk's patch (an HBM round trip) at the
    // first tap of every chunk (s_waitcnt vmcnt(13) with 50 loads in flight).
    // 1x1 (CS chunks per stage, no halo): the flat (pixel, pair) enumeration with offsets re-derived per stage.
    constexpr int XOFF = bf3_xoff(KS);               // patch column x is pixel x + XOFF of the aligned group grid
    // stride 2: input rows 2 ho0 - 1 .. 2 ho0 + 2 PTH - 1 (2 PTH + 1 of them), columns on the group grid that starts at 2 wo0 - 4
    constexpr int PROWS = S2 ? 2 * PTH + 1 : PH;     // INPUT rows staged per image
    constexpr int NG = S2 ? (2 * PTW + 3) / 4 + 1 : (XOFF + PWR + 3) / 4;  // 16-byte groups per input row
    constexpr bool UP = KS == 2;                     // rows = 4 * channel + parity class, stores into the (2 H x 2 W) output
    constexpr int ITEMS3 = IMGS * PROWS * NG * (CB / 2);
    constexpr int NI3 = KS == 1 ? 1 : (ITEMS3 + NTB - 1) / NTB;
    constexpr unsigned OOB = 0x80000000u;            // (>= num_records of the descriptor: the load returns zeros)
    static_assert(NTB % 8 == 0, "a thread keeps its channel pair over its items");
    unsigned ivo[NI3];
    int idst[NI3], imask[NI3];
    int idst1[S2 ? NI3 : 1];  // (stride 2: the odd-column plane's base; idst is the even-column plane's)
    const int qpair = t & 7;
    if constexpr (KS != 1 && !S2) {
#pragma unroll
        for (int j = 0; j < NI3; ++j) {
            const int idx = t + j * NTB, rest = idx >> 3;
            const int g = rest % NG, yy = rest / NG, il = yy / PH, y = yy - il * PH;
            const int hi = ho0 - p.pad + y, wi = wo0 - bf3_gx0(KS) + 4 * g, img = img0 + il;
            const bool ok = idx < ITEMS3 && img < nimg && (unsigned)hi < (unsigned)p.Hin && wi >= 0 && wi + 3 < p.Win;
            ivo[j] = ok ? (unsigned)(((int64_t)img * p.in_img_stride + (int64_t)(2 * qpair) * HW + hi * p.Win + wi) * 4) : OOB;
            idst[j] = ((il * PH + y) * PWR + 4 * g - XOFF) * PIXB + qpair * 4;  // pixel e of the group: + e * PIXB
            int m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) m |= (idx < ITEMS3 && (unsigned)(4 * g + e - XOFF) < (unsigned)PWR) ? (1 << e) : 0;
            imask[j] = m;
        }
    }
    if constexpr (S2) {
        // group g of input row y: pixels wi = 2 wo0 - 4 + 4 g + e.  e = 0, 2 are even columns 2 (wo0 + c): plane column
        // c = 2 g - 2 + e / 2 (needed for c < PTW); e = 1, 3 are odd columns 2 (wo0 + c) - 1: c = 2 g - 1 + e / 2 (c <= PTW).
        // Row y is input row 2 ho0 - 1 + y: y even = an odd row (plane row y / 2 <= PTH), y odd = an even row (plane row (y - 1) / 2).
#pragma unroll
        for (int j = 0; j < NI3; ++j) {
            const int idx = t + j * NTB, rest = idx >> 3;
            const int g = rest % NG, yy = rest / NG, il = yy / PROWS, y = yy - il * PROWS;
            const int hi = 2 * ho0 - 1 + y, wi = 2 * wo0 - 4 + 4 * g, img = img0 + il;
            const bool ok = idx < ITEMS3 && img < nimg && (unsigned)hi < (unsigned)p.Hin && wi >= 0 && wi + 3 < p.Win;
            ivo[j] = ok ? (unsigned)(((int64_t)img * p.in_img_stride + (int64_t)(2 * qpair) * HW + hi * p.Win + wi) * 4) : OOB;
            const int rp = (y & 1) ^ 1, prow = y >> 1;
            const int pbase = (il * 4 + rp * 2) * PLANE + prow * PWR;
            idst[j] = (pbase + 2 * g - 2) * PIXB + qpair * 4;           // even columns: e = 0 here, e = 2 one pixel on
            idst1[j] = (pbase + PLANE + 2 * g - 1) * PIXB + qpair * 4;  // odd columns:  e = 1 here, e = 3 one pixel on
            int m = 0;
            if (idx < ITEMS3) {
                m |= ((unsigned)(2 * g - 2) < (unsigned)PTW) ? 1 : 0;
                m |= ((unsigned)(2 * g - 1) <= (unsigned)PTW) ? 2 : 0;
                m |= ((unsigned)(2 * g - 1) < (unsigned)PTW) ? 4 : 0;
                m |= ((unsigned)(2 * g) <= (unsigned)PTW) ? 8 : 0;
            }
            imask[j] = m;
        }
    }
    const __amdgpu_buffer_rsrc_t rB = bf3_rsrc(p.B);
    // 1x1: the patch is the output tile itself (no halo, maybe strided); NTB / NPIX thread groups share a pixel set and take
    // every (NTB / NPIX)-th chunk of the stage, all 8 channel pairs of each
    constexpr int NPI = KS == 1 ? ITEMS / NTB : 1;
    constexpr int G1 = KS == 1 ? NTB / NPIX : 1;
    static_assert(KS != 1 || (NTB % NPIX == 0 && CS % G1 == 0), "1x1: whole thread groups per pixel set");
    int src1 = -1, dst1 = 0, g1 = 0;
    if constexpr (KS == 1) {
        const int pix = t % NPIX;
        g1 = t / NPIX;
        const int il = pix / (PH * PWR), rem = pix - il * (PH * PWR);
        const int y = rem / PWR, x = rem - y * PWR;
        const int hi = (ho0 + y) * p.stride, wi = (wo0 + x) * p.stride, img = img0 + il;
        const bool ok = img < nimg && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        src1 = ok ? (int)((int64_t)img * p.in_img_stride + hi * p.Win + wi) : -1;
        dst1 = pix * PIXB;
    }
    float r0[NPI], r1[NPI];
    v4i rv[NI3][2];
    auto load_patch = [&](int c) {
        const int cbase = c * (CB * CS) * HW;
        const int left = p.Cin - c * (CB * CS);  // channels this stage still has (ragged last chunk: the rest reads as zero)
        if constexpr (KS == 1) {
#pragma unroll
            for (int jj = 0; jj < CS / G1; ++jj)
#pragma unroll
                for (int q = 0; q < CB / 2; ++q) {
                    const int ch = (g1 + jj * G1) * CB + 2 * q;  // channel of the pair inside the stage
                    const bool ok0 = src1 >= 0 && ch < left, ok1 = src1 >= 0 && ch + 1 < left;
                    const int o = src1 + cbase + ch * HW;
                    r0[jj * (CB / 2) + q] = p.B[ok0 ? o : 0];
                    r1[jj * (CB / 2) + q] = p.B[ok1 ? o + HW : 0];
                }
        } else {
            const unsigned hw4 = (unsigned)HW * 4u;
            const bool ok0 = 2 * qpair < left, ok1 = 2 * qpair + 1 < left;
#pragma unroll
            for (int j = 0; j < NI3; ++j) {
                const unsigned v0 = ok0 ? ivo[j] : OOB, v1 = ok1 ? (ivo[j] | hw4 * 0u) + ((ivo[j] & OOB) ? 0u : hw4) : OOB;
                rv[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)v0, cbase * 4, 0);
                rv[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)v1, cbase * 4, 0);
            }
        }
    };
    auto stage = [&](int c) -> uint32_t {  // returns the OR of the lower pieces this thread wrote (0: its values were bf16-exact)
        uint32_t nz = 0;
        const int left = p.Cin - c * (CB * CS);
        if constexpr (KS == 1) {
#pragma unroll
            for (int jj = 0; jj < CS / G1; ++jj)
#pragma unroll
                for (int q = 0; q < CB / 2; ++q) {
                    const int ch = (g1 + jj * G1) * CB + 2 * q;
                    const float v0 = (src1 >= 0 && ch < left) ? r0[jj * (CB / 2) + q] : 0.f;
                    const float v1 = (src1 >= 0 && ch + 1 < left) ? r1[jj * (CB / 2) + q] : 0.f;
                    uint32_t H, M, L;
                    split3_pair(v0, v1, H, M, L);
                    nz |= M | L;
                    unsigned char* d = smem + (g1 + jj * G1) * (NPIX * PIXB) + dst1 + q * 4;
                    *reinterpret_cast<uint32_t*>(d) = H;
                    *reinterpret_cast<uint32_t*>(d + 32) = M;
                    *reinterpret_cast<uint32_t*>(d + 64) = L;
                }
        } else {
#pragma unroll
            for (int j = 0; j < NI3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t H, M, L;
                    split3_pair(__int_as_float(rv[j][0][e]), __int_as_float(rv[j][1][e]), H, M, L);
                    if ((imask[j] >> e) & 1) {  // (pixels of the group outside the patch - or items past the last - are not staged)
                        nz |= M | L;
                        unsigned char* d = S2 ? smem + ((e & 1) ? idst1[j] : idst[j]) + (e >> 1) * PIXB : smem + idst[j] + e * PIXB;
                        *reinterpret_cast<uint32_t*>(d) = H;
                        *reinterpret_cast<uint32_t*>(d + 32) = M;
                        *reinterpret_cast<uint32_t*>(d + 64) = L;
                    }
                }
        }
        return nz;
    };

    // per-lane operand bases
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nl = (wn * TN + tn) * 32 + l31;
        const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
        bbase[tn] = ((il * (S2 ? 4 : 1) * PH + ph) * PWR + pw) * PIXB + half * 16;  // (stride 2: plane (0, 0) of the image; the tap adds its plane)
    }
    // weights: [32-channel tile][chunk][tap (padded)][piece][lane] x 16 bytes
    __amdgpu_buffer_rsrc_t rA[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int mt = min(m0 / 32 + wm * TM + tm, (p.M + 31) / 32 - 1);  // (a channel tile past M reads the last one's weights; its rows are never stored)
        rA[tm] = bf3_rsrc(a_split + (int64_t)grp * a_grp_bytes + (int64_t)mt * nch * KKP * (3 * 1024));
    }
    const int nst = (nch + CS - 1) / CS;  // stages
    const int c_beg = bid.z * chunks_per_split, c_end = min(nst, c_beg + chunks_per_split);  // (split K: stages over blockIdx.z)
    const int steps = nch * KKP;
    auto load_a = [&](int tm, int s, int pl) -> v4i {
        const int sc = s < steps ? s : steps - 1;
        return __builtin_amdgcn_raw_buffer_load_b128(rA[tm], (sc * 3 + pl) * 1024 + lane * 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

    v4i abuf[DA][TM][3];
#pragma unroll
    for (int d = 0; d < DA; ++d)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) abuf[d][tm][pl] = load_a(tm, c_beg * RP + d, pl);

    unsigned long long tk0 = BF3_T(), t_stage = 0, t_mma = 0;
    (void)tk0, (void)t_stage, (void)t_mma;
#ifdef BF3_TIMING
    const unsigned long long cyc0 = clock64();
#endif
    load_patch(c_beg);
    for (int c = c_beg; c < c_end; ++c) {
        const unsigned long long ta = BF3_T();
        const uint32_t nz = stage(c);
        // A chunk whose staged values are all bf16-exact (one-hot map features: the map CNN's first layer) has zero second and
        // third pieces: the three products against them are exact zeros and are not issued - same bits, half the MFMAs and a
        // third of the operand reads.  The block-wide OR rides in the barrier the staging needs anyway.
        const bool lite = HAS_LITE ? __syncthreads_or((int)(nz != 0)) == 0 : (__syncthreads(), false);
        const unsigned long long tb = BF3_T();
        t_stage += tb - ta;
        // The next chunk's patch flies under the MFMA phase.  Unconditional (the last chunk re-loads itself, six loads nobody
        // reads): with a branch around it the compiler cannot count the loads in flight and waits for ALL of them at the first
        // tap.  3x3 / 7x7: issued PATCH_TAP taps into the phase, ~3 us of MFMA work before the staging pass that reads them -
        // the weight loads of the taps in between (waited for in order, a tap after tap) then never queue behind an HBM miss.
        constexpr int PATCH_TAP = KS == 1 ? -1 : (KK - (TM == 2 ? 4 : 8) > 0 ? KK - (TM == 2 ? 4 : 8) : 0);
        if constexpr (KS == 1) load_patch(c + 1 < c_end ? c + 1 : c);
        const int s0 = c * RP;
        auto taps = [&](auto lite_tag) {
            constexpr bool LITE = decltype(lite_tag)::value;
            constexpr int NPL = LITE ? 1 : 3;
            auto read_b = [&](int r, bf16x8 (&b)[TN][3]) {  // the tap's B fragments: one ds_read_b128 per (pixel tile, piece)
                const int kh = r / KS, kw = r - kh * KS;
                const int toff = KS == 1 ? r * (NPIX * PIXB)  // (1x1: tap r = chunk r of the stage)
                                 : (S2 ? ((((kh != 1) * 2 + (kw != 1)) * PLANE + (kh == 2) * PWR + (kw == 2)) * PIXB) : (kh * PWR + kw) * PIXB);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        b[tn][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4i*>(smem + bbase[tn] + toff + pl * 32));
            };
            bf16x8 bq[2][TN][3];  // this tap's fragments and the next tap's, read while this tap's MFMAs issue
            read_b(0, bq[0]);
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                const int slot = r % DA;
                if (r < KK * CS && (KS != 1 || c * CS + r < nch)) {  // (padding taps, and 1x1: chunks past the last one, issue nothing)
                    if (r + 1 < KK * CS) read_b(r + 1, bq[(r + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);  // (the reads stay AHEAD of this tap's MFMAs: left alone they sink to their use)
                    bf16x8 a[TM][3];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) a[tm][pl] = __builtin_bit_cast(bf16x8, abuf[slot][tm][pl]);
                    // smallest products first; consecutive MFMAs hit different accumulators
#define IVLN_BF3_PROD(PA, PB)                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)              \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][PA], bq[r & 1][tn][PB], acc[tm][tn], 0, 0, 0)
                    if constexpr (!LITE) {
                        IVLN_BF3_PROD(0, 2);
                        IVLN_BF3_PROD(1, 1);
                    }
                    IVLN_BF3_PROD(2, 0);
                    if constexpr (!LITE) IVLN_BF3_PROD(0, 1);
                    IVLN_BF3_PROD(1, 0);
                    IVLN_BF3_PROD(0, 0);
#undef IVLN_BF3_PROD
                }
                __builtin_amdgcn_sched_barrier(0);
                // the slot just used receives the weights of DA taps ahead (the rotation closes over the stage: RP % DA == 0)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) abuf[slot][tm][pl] = load_a(tm, s0 + r + DA, pl);
                if (r == PATCH_TAP) load_patch(min(c + 1, c_end - 1));  // (behind this tap's weight loads: every wait of this chunk is for loads older than these)
                __builtin_amdgcn_sched_barrier(0);  // (... and so do the weight loads: DA taps of flight time, not one)
            }
        };
        if constexpr (HAS_LITE) {
            if (lite) taps(std::true_type{});
            else taps(std::false_type{});
        } else {
            taps(std::false_type{});
        }
        __syncthreads();
        t_mma += BF3_T() - tb;
    }
    const unsigned long long tk1 = BF3_T();
    (void)tk1;

    // ---- FUSE: the bottleneck's 1x1 expansion behind this 3x3 conv, in the same launch (ivln_gemm_desc.fuse_*).  The workgroup
    // holds ALL of the conv's channels (M == BM) for its 128 pixels:
    //   1. y = ReLU(scale * acc + shift) goes to LDS as fp32 [channel][pixel of the tile] - over the patch, which is dead;
    //   2. every wave then runs the wave-tile 1x1 kernel's loop (k_conv1x1_bf3_ks<true>) on it for the 32-channel tiles
    //      wave, wave + 4, ... of the fuse_M outputs: a lane reads four consecutive pixels of a channel (one ds_read_b128 where
    //      that kernel issues a buffer load), builds the B fragments in registers, streams the 1x1 weights global -> registers;
    //   3. and stores its tile straight from the accumulators (register r of the four tiles = a float4 along the pixel index)
    //      with the folded bn3, the residual and the ReLU.
    // Same arithmetic, same order per accumulator as the two separate launches; the planes-channel tensor never leaves the CU
    // (written + read once per block before: 2 x 4 B x planes x pixels), one launch and one prologue / epilogue less. ----
    if constexpr (FUSE) {
        static_assert(KS == 3 && IMGS == 1 && PTW == 32 && (BN == 128 || BN == 64) && NTB == 256, "the fused tail is built for the 4 x 32 and 2 x 32 pixel tiles");
        constexpr int Q = BN / 32;  // consecutive pixels of a lane in the second stage = pixel tiles of its B fragments (4 | 2)
        constexpr int LDY = BN + 4;
        float* const Y = reinterpret_cast<float*>(smem);
        {
            const __amdgpu_buffer_rsrc_t rS = bf3_rsrc(p.scale), rH = bf3_rsrc(p.shift);
            const int me0 = grp * p.M + m0 + 4 * half;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = 32 * (wm * TM + tm) + (r & 3) + 8 * (r >> 2);  // (+ 4 half: in me0 / below)
                    const float sc = p.scale ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, me0 * 4, cl * 4, 0)) : 1.f;
                    const float sh = p.shift ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rH, me0 * 4, cl * 4, 0)) : 0.f;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        float v = acc[tm][tn][r];
                        v = p.scale ? fmaf(v, sc, sh) : v + sh;
                        if (p.relu) v = fmaxf(v, 0.f);
                        Y[(cl + 4 * half) * LDY + (wn * TN + tn) * 32 + l31] = v;
                    }
                }
        }
        __syncthreads();
        constexpr int NCH2 = BM / CB, DA2 = 2;
        const int ntile2 = p.fuse_M / 32;
        // this lane's pixels: tile-linear Q l31 .. + Q - 1 = row Q l31 / 32 of the tile, columns (Q l31) % 32 ..
        const int ph = (Q * l31) >> 5, pw = (Q * l31) & 31;
        const int ho = ho0 + ph, wo = wo0 + pw;
        const bool pix_ok = img0 < nimg && ho < p.Hout && wo < p.Wout;
        const unsigned char* const a2 = reinterpret_cast<const unsigned char*>(p.fuse_A_split) + (int64_t)grp * p.fuse_a_grp_stride * 4;
        const __amdgpu_buffer_rsrc_t rR = bf3_rsrc(p.residual), rS3 = bf3_rsrc(p.fuse_scale), rH3 = bf3_rsrc(p.fuse_shift), rD = bf3_rsrc(p.D);
        const bool has_res = p.residual != nullptr, has_sc = p.fuse_scale != nullptr, has_sh = p.fuse_shift != nullptr;
        const float* const yl = Y + (8 * half) * LDY + Q * l31;
        for (int mt = wave; mt < ntile2; mt += NTB / 64) {
            const __amdgpu_buffer_rsrc_t rA2 = bf3_rsrc(a2 + (int64_t)mt * NCH2 * 3072);
            auto load_a2 = [&](int c, v4i (&ab)[3]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ab[pl] = __builtin_amdgcn_raw_buffer_load_b128(rA2, (min(c, NCH2 - 1) * 3 + pl) * 1024 + lane * 16, 0, 0);
            };
            v4i ab[DA2][3];
#pragma unroll
            for (int d = 0; d < DA2; ++d) load_a2(d, ab[d]);
            // the tile's epilogue operands, requested before its K loop
            const int mch = mt * 32 + 4 * half;
            const int me3 = grp * p.fuse_M + mch;
            const unsigned off0 = (unsigned)((((int64_t)img0 * p.Ctot + mch) * p.HoWo + ho * p.Wout + wo) * 4);
            float rres[16][Q];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = pix_ok && has_res ? (int)off0 : (int)OOB, so = ((r & 3) + 8 * (r >> 2)) * p.HoWo * 4;
                if constexpr (Q == 4) {
                    const v4i t4 = __builtin_amdgcn_raw_buffer_load_b128(rR, ro, so, 0);
#pragma unroll
                    for (int e = 0; e < Q; ++e) rres[r][e] = __int_as_float(t4[e]);
                } else {
                    const v2i t2 = __builtin_amdgcn_raw_buffer_load_b64(rR, ro, so, 0);
#pragma unroll
                    for (int e = 0; e < Q; ++e) rres[r][e] = __int_as_float(t2[e]);
                }
            }
            f32x16 acc2[Q];
#pragma unroll
            for (int e = 0; e < Q; ++e)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[e][i] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH2; ++c) {
                v4i bq[Q][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x0[Q], x1[Q];
                    if constexpr (Q == 4) {
                        const v4i t0 = *reinterpret_cast<const v4i*>(yl + (c * CB + 2 * i) * LDY), t1 = *reinterpret_cast<const v4i*>(yl + (c * CB + 2 * i + 1) * LDY);
#pragma unroll
                        for (int e = 0; e < Q; ++e) x0[e] = __int_as_float(t0[e]), x1[e] = __int_as_float(t1[e]);
                    } else {
                        const v2i t0 = *reinterpret_cast<const v2i*>(yl + (c * CB + 2 * i) * LDY), t1 = *reinterpret_cast<const v2i*>(yl + (c * CB + 2 * i + 1) * LDY);
#pragma unroll
                        for (int e = 0; e < Q; ++e) x0[e] = __int_as_float(t0[e]), x1[e] = __int_as_float(t1[e]);
                    }
#pragma unroll
                    for (int e = 0; e < Q; ++e) {
                        uint32_t H, M, L;
                        split3_pair(x0[e], x1[e], H, M, L);
                        bq[e][0][i] = (int)H, bq[e][1][i] = (int)M, bq[e][2][i] = (int)L;
                    }
                }
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, ab[c % DA2][pl]);
                __builtin_amdgcn_sched_barrier(0);
                load_a2(c + DA2, ab[c % DA2]);
                __builtin_amdgcn_sched_barrier(0);
#define IVLN_BF3_PROD(PA, PB)                           \
    _Pragma("unroll") for (int e = 0; e < Q; ++e)       \
        acc2[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], __builtin_bit_cast(bf16x8, bq[e][PB]), acc2[e], 0, 0, 0)
                IVLN_BF3_PROD(0, 2);
                IVLN_BF3_PROD(1, 1);
                IVLN_BF3_PROD(2, 0);
                IVLN_BF3_PROD(0, 1);
                IVLN_BF3_PROD(1, 0);
                IVLN_BF3_PROD(0, 0);
#undef IVLN_BF3_PROD
            }
            // (the folded bn3 of the tile's channels: 8 x 16 bytes per lane, L2-resident after the first workgroups - asked for
            //  here, not before the K loop, because 32 more live registers there spill)
            v4i esc4[4], esh4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                esc4[g] = __builtin_amdgcn_raw_buffer_load_b128(rS3, has_sc ? me3 * 4 : (int)OOB, g * 32, 0);
                esh4[g] = __builtin_amdgcn_raw_buffer_load_b128(rH3, has_sh ? me3 * 4 : (int)OOB, g * 32, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = (r & 3) + 8 * (r >> 2);
                float v[Q];
#pragma unroll
                for (int e = 0; e < Q; ++e) v[e] = acc2[e][r];
                // STRAIGHT-LINE code between the stores (absent scale = 1, absent shift / residual = the zeros the out-of-range
                // loads returned): a 16-byte buffer store with a scalar offset reads its data registers a few cycles after
                // it issues, the compiler's hazard recognizer pads for that inside a basic block, and with uniform branches
                // here it let a VALU write follow the store across a block boundary - lanes 48-63 of one register of one store
                // stale, once in a few thousand workgroups (found by tools/dbg_fuse.py against the two-launch path).
                const float sc = has_sc ? __int_as_float(esc4[r >> 2][r & 3]) : 1.f, sh = __int_as_float(esh4[r >> 2][r & 3]);
#pragma unroll
                for (int e = 0; e < Q; ++e) v[e] = fmaxf(fmaf(v[e], sc, sh) + rres[r][e], 0.f);  // (bn3, residual, the block's closing ReLU)
                if constexpr (Q == 4) {
                    v4i o;
#pragma unroll
                    for (int e = 0; e < Q; ++e) o[e] = __float_as_int(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rD, pix_ok ? (int)off0 : (int)OOB, cs * p.HoWo * 4, 0);
                } else {
                    v2i o;
#pragma unroll
                    for (int e = 0; e < Q; ++e) o[e] = __float_as_int(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b64(o, rD, pix_ok ? (int)off0 : (int)OOB, cs * p.HoWo * 4, 0);
                }
                BF3_STORE_GUARD();
            }
        }
        return;
    }

    // ---- epilogue: slabs of 32 channels through LDS.  acc[r] -> channel (r&3) + 8*(r>>2) + 4*half, pixel l31 ----
    float* T = reinterpret_cast<float*>(smem);
    for (int slab = 0; slab < TM * WM; ++slab) {
        const int sw = slab / TM, stm = slab - sw * TM;
        if (wm == sw) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                if (tm == stm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            T[((r & 3) + 8 * (r >> 2) + 4 * half) * LDT + (wn * TN + tn) * 32 + l31] = acc[tm][tn][r];
        }
        __syncthreads();
        const int mbase = m0 + slab * 32;
        if constexpr (UP) {
            // the slab's 32 rows = 8 channels x 4 parity classes; rows 4 c + 2 a (b = 0) and + 1 (b = 1) x two horizontally
            // adjacent pixels of the tile = four consecutive floats of output row 2 ho + a: one 16-byte store per
            // (channel, a, pixel pair), whole 64-byte segments of the (2 H x 2 W) tensor (gemm_common.h: up2x4_wide_store)
            for (int idx = t; idx < 16 * (BN / 2); idx += NTB) {
                const int r = idx / (BN / 2), j = idx - r * (BN / 2);  // r = 2 * (channel of the slab) + a
                const int ml = 2 * r, nl = 2 * j;
                const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
                const int img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
                if (mbase + ml >= p.M || img >= nimg || ho >= p.Hout || wo >= p.Wout) continue;  // (M % 4 == 0, Wout % 4 == 0: in or out whole)
                const int co = (mbase + ml) >> 2, a = r & 1;
                const int64_t addr = (((int64_t)img * p.Ctot + co) * (2 * p.Hout) + 2 * ho + a) * (2 * p.Wout) + 2 * wo;
                const float2 t0 = *reinterpret_cast<const float2*>(T + ml * LDT + nl);
                const float2 t1 = *reinterpret_cast<const float2*>(T + (ml + 1) * LDT + nl);
                float4 v = make_float4(t0.x, t1.x, t0.y, t1.y);
                if (p.scale) {
                    const float sc = p.scale[co], sh = p.shift[co];
                    v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
                } else if (p.shift) {
                    const float sh = p.shift[co];
                    v.x += sh, v.y += sh, v.z += sh, v.w += sh;
                }
                if (p.residual) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.residual + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.accumulate) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.D + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4*>(p.D + addr) = v;
            }
            __syncthreads();
            continue;
        }
        for (int idx = t; idx < 32 * (BN / 4); idx += NTB) {
            const int ml = idx / (BN / 4), c4 = idx - ml * (BN / 4);
            const int m = mbase + ml;
            const int nl = 4 * c4;
            const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
            const int img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
            const bool ok = m < p.M && img < nimg && ho < p.Hout && wo < p.Wout;  // (Wout % 4 == 0: a quad is in or out whole)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && p.splits > 1) {  // (uniform) raw partial sums of this channel range: slab [split][M][N], reduced by k_splitk_epilogue
                v = *reinterpret_cast<const float4*>(T + ml * LDT + nl);
                *reinterpret_cast<float4*>(p.ws + ((int64_t)bid.z * p.M + m) * p.N + (int64_t)img * p.HoWo + ho * p.Wout + wo) = v;
            } else if (ok) {
                v = *reinterpret_cast<const float4*>(T + ml * LDT + nl);
                const int64_t addr = ((int64_t)img * p.Ctot + m) * p.HoWo + ho * p.Wout + wo;
                const int me = p.grp_imgs > 0 ? (img / p.grp_imgs) * p.M + m : m;
                if (p.scale) {
                    const float sc = p.scale[me], sh = p.shift[me];
                    v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
                } else if (p.shift) {
                    const float sh = p.shift[me];
                    v.x += sh, v.y += sh, v.z += sh, v.w += sh;
                }
                if (p.residual) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.residual + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.accumulate) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.D + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4*>(p.D + addr) = v;
            }
            if (p.stat_partials) {  // (uniform) {count, mean, M2} of the 128 pixels a half-wave just stored
                float cnt = ok ? 4.f : 0.f, sum = ok ? (v.x + v.y) + (v.z + v.w) : 0.f;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o), sum += __shfl_xor(sum, o);
                const float mean = cnt > 0.f ? sum / cnt : 0.f;
                const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                float q = ok ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
                if ((t & 31) == 0 && m < p.M) {
                    float* st = p.stat_partials + ((int64_t)(bx * (BN / 128) + c4 / 32) * p.M + m) * 3;
                    st[0] = cnt, st[1] = mean, st[2] = q;
                }
            }
        }
        __syncthreads();
    }
#ifdef BF3_TIMING
    if (threadIdx.x == 0) {
        const int b = (blockIdx.y * gridDim.x + blockIdx.x) & 8191;
        g_bf3_stamp[b * 8 + 0] = tk1 - tk0;
        g_bf3_stamp[b * 8 + 1] = t_stage;
        g_bf3_stamp[b * 8 + 2] = t_mma;
        g_bf3_stamp[b * 8 + 3] = clock64() - cyc0;  // shader cycles over the whole kernel: with [0] + epilogue wall time, the clock it ran at
        g_bf3_stamp[b * 8 + 4] = t_entry;
        g_bf3_stamp[b * 8 + 5] = tk0;
        __builtin_amdgcn_s_waitcnt(0);
        g_bf3_stamp[b * 8 + 6] = wall_clock64();
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The same arithmetic for the PIXEL-STARVED deep 3x3 convs (RedNet's layers 3-4 and the decoder's first stages,
// rednet.py:190-263: 128-512 channels on 8x8 ... 32x32 maps, N = 512 ... 8192 pixels, K = 1152 ... 4608): round 5.
// The tiled kernel above fills the chip there only by splitting K over blockIdx.z - two or three 16-channel chunks per
// workgroup between a prologue and a slab epilogue, then a reduction launch: 30 us + 6.5 us for 2.4 GFLOP.  Here K is split
// over the WAVES of a workgroup instead:
//   * a workgroup (8 waves) owns 32 output channels x 64 output pixels (a whole 8x8 image, 4 rows of a 16-wide or 2 rows of
//     a 32-wide one) over the WHOLE K; wave w takes the contiguous range [w, w + 1) * ceil(nch / 8) of the 16-channel chunks;
//   * everything a wave needs is wave-private: its chunk's input patch is loaded (aligned 16-byte buffer loads, zeros from
//     out-of-range offsets - the scheme of the tiled kernel), split into the three bf16 pieces and written to ITS region of
//     LDS, its weight stream comes global -> registers three taps ahead.  No workgroup barrier inside the K loop: LDS
//     accesses of one wave execute in program order, so the waves run free of each other and hide one another's latencies;
//   * at the end the eight partial tiles meet in LDS (a barrier, 8 x 8 KB, aliasing the patch regions), every thread sums
//     four outputs over the waves in a fixed order and applies the fused epilogue of ivln_gemm_f32: no slabs, no reduction
//     launch, the same bits on every run.
// L2 -> CU traffic is the price: every workgroup streams its 32 channels' weights for the whole K (885 KB at K = 4608).
// ------------------------------------------------------------------------------------------------------------------
// KS = 2: the same kernel for the stacked parity classes of a stride-2 3x3 transposed conv over few pixels (the decoder's first
// upsampling stages, rednet.py:152-181 at 8 x 8 and 16 x 16): rows = 4 * channel + class, a 2 x 2 window, and an epilogue that
// stores 2 x 2 output blocks (IVLN_D_NCHW_UP2X4).
template <int PTH, int PTW, int TN = 2, int KS = 3>
__global__ __launch_bounds__(512, 2) void k_conv_bf3_ks(const ivln_gemm_desc p, const unsigned char* a_split, long long a_grp_bytes,
                                                        int tiles_w, int tiles_h, int nimg) {
    // DA: weight taps in flight.  A tap is only 12 MFMAs here (384 pipe cycles, 768 with the SIMD's other wave): three taps
    // ahead were ~1 us of cover against an L2 round trip of 1-2 us under load - every tap waited (first version: 28 us for
    // 512 x 512 x 4608 with 13 us of MFMA issue).  A whole chunk ahead (9 taps, 108 registers) covers it.
    // TN = 1 (32 pixels per workgroup): the launches that fill less than half the chip with 64-pixel tiles (512 x 512 x 4608:
    // 128 workgroups) - twice the workgroups, half the MFMAs per wave, 20 % more halo per output.
    constexpr int NW = 8, KK = KS * KS, DA = KK;
    constexpr bool UP = KS == 2;
    static_assert(PTH * PTW == 32 * TN && (KS == 3 || KS == 2), "32 TN pixels per workgroup");
    constexpr int PH = PTH + KS - 1, PWR = PTW + KS - 1, NPIX = PH * PWR;
    constexpr int XOFF = bf3_xoff(KS), NG = (XOFF + PWR + 3) / 4;
    constexpr int ITEMS = PH * NG * (CB / 2), NI = (ITEMS + 63) / 64;
    constexpr int WREG = (NPIX * PIXB + 15) & ~15;  // bytes of a wave's patch region
    constexpr int LDT = 32 * TN + 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int bx = bid.x;
    const int tw = bx % tiles_w, th = (bx / tiles_w) % tiles_h, img0 = bx / (tiles_w * tiles_h);
    const int ho0 = th * PTH, wo0 = tw * PTW;
    const int m0 = bid.y * 32;
    const int nch = p.Cin / CB;
    const int HW = p.Hin * p.Win;
    const int grp = p.grp_imgs > 0 ? img0 / p.grp_imgs : 0;
    const int cpw = (nch + NW - 1) / NW;
    const int c0 = min(nch, wave * cpw), c1 = min(nch, c0 + cpw);
    unsigned char* const wsm = smem + wave * WREG;

    // staging items of a LANE (the wave stages its own patch): (patch row, aligned 16-byte group, channel pair)
    unsigned ivo[NI];
    int idst[NI], imask[NI];
    const int qpair = lane & 7;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int idx = lane + j * 64, rest = idx >> 3;
        const int g = rest % NG, y = rest / NG;
        const int hi = ho0 - (KS == 3 ? 1 : 0) + y, wi = wo0 - bf3_gx0(KS) + 4 * g;  // (pad 1 | the 2 x 2 window starts at the pixel itself)
        const bool ok = idx < ITEMS && img0 < nimg && (unsigned)hi < (unsigned)p.Hin && wi >= 0 && wi + 3 < p.Win;
        ivo[j] = ok ? (unsigned)(((int64_t)img0 * p.in_img_stride + (int64_t)(2 * qpair) * HW + hi * p.Win + wi) * 4) : OOB;
        idst[j] = (y * PWR + 4 * g - XOFF) * PIXB + qpair * 4;
        int m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) m |= (idx < ITEMS && (unsigned)(4 * g + e - XOFF) < (unsigned)PWR) ? (1 << e) : 0;
        imask[j] = m;
    }
    const __amdgpu_buffer_rsrc_t rB = bf3_rsrc(p.B);
    const unsigned hw4 = (unsigned)HW * 4u;
    v4i rv[NI][2];
    auto load_patch = [&](int c) {
        const int so = c * CB * HW * 4;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            rv[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)ivo[j], so, 0);
            rv[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)((ivo[j] & OOB) ? OOB : ivo[j] + hw4), so, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t H, M, L;
                split3_pair(__int_as_float(rv[j][0][e]), __int_as_float(rv[j][1][e]), H, M, L);
                if ((imask[j] >> e) & 1) {
                    unsigned char* d = wsm + idst[j] + e * PIXB;
                    *reinterpret_cast<uint32_t*>(d) = H;
                    *reinterpret_cast<uint32_t*>(d + 32) = M;
                    *reinterpret_cast<uint32_t*>(d + 64) = L;
                }
            }
    };
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nl = tn * 32 + l31, ph = nl / PTW, pw = nl % PTW;
        bbase[tn] = (ph * PWR + pw) * PIXB + half * 16;
    }
    const int mt = min(m0 / 32, (p.M + 31) / 32 - 1);
    const __amdgpu_buffer_rsrc_t rA = bf3_rsrc(a_split + (int64_t)grp * a_grp_bytes + (int64_t)mt * nch * KK * (3 * 1024));
    const int s_last = max(c1 * KK - 1, 0);
    auto load_a = [&](int s_, int pl) -> v4i {
        return __builtin_amdgcn_raw_buffer_load_b128(rA, (min(s_, s_last) * 3 + pl) * 1024 + lane * 16, 0, 0);
    };
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

#ifdef BF3_TIMING
    const unsigned long long ts0 = lane == 0 ? wall_clock64() : 0ull;
    unsigned long long ts1 = ts0, t_stage = 0;
#endif
    if (c0 < c1) {
        v4i abuf[DA][3];
#pragma unroll
        for (int d = 0; d < DA; ++d)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) abuf[d][pl] = load_a(c0 * KK + d, pl);
        load_patch(c0);
#ifdef BF3_TIMING
        __builtin_amdgcn_s_waitcnt(0);  // (timing build only: the prologue's round trip as a phase of its own)
        ts1 = lane == 0 ? wall_clock64() : 0ull;
#endif
        for (int c = c0; c < c1; ++c) {
#ifdef BF3_TIMING
            const unsigned long long tsa = lane == 0 ? wall_clock64() : 0ull;
#endif
            stage();
#ifdef BF3_TIMING
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            t_stage += (lane == 0 ? wall_clock64() : 0ull) - tsa;
#endif
            const int s0 = c * KK;
            auto read_b = [&](int r, bf16x8 (&b)[TN][3]) {
                const int kh = r / KS, kw = r - kh * KS, toff = (kh * PWR + kw) * PIXB;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[tn][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4i*>(wsm + bbase[tn] + toff + pl * 32));
            };
            bf16x8 bq[2][TN][3];
            read_b(0, bq[0]);
#pragma unroll
            for (int r = 0; r < KK; ++r) {
                const int slot = r % DA;
                if (r + 1 < KK) read_b(r + 1, bq[(r + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, abuf[slot][pl]);
#define IVLN_BF3_PROD(PA, PB)                            \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)    \
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], bq[r & 1][tn][PB], acc[tn], 0, 0, 0)
                IVLN_BF3_PROD(0, 2);
                IVLN_BF3_PROD(1, 1);
                IVLN_BF3_PROD(2, 0);
                IVLN_BF3_PROD(0, 1);
                IVLN_BF3_PROD(1, 0);
                IVLN_BF3_PROD(0, 0);
#undef IVLN_BF3_PROD
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) abuf[slot][pl] = load_a(s0 + r + DA, pl);
                if (r == (KK > 4 ? 1 : 0)) load_patch(min(c + 1, c1 - 1));  // (next chunk's patch: 7 | 3 taps of MFMA work before the staging pass reads it)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- the eight partial tiles meet in LDS: red[wave][32 channels][64 pixels (+4)] over the patch regions ----
#ifdef BF3_TIMING
    const unsigned long long ts2 = lane == 0 ? wall_clock64() : 0ull;
#endif
    __syncthreads();  // every wave is done reading its patch
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + tn * 32 + l31] = acc[tn][r];
    __syncthreads();
    if constexpr (UP) {
        // 8 channels x 2 rows of the 2 x 2 blocks x 16 TN pixel pairs: GEMM rows 4 c + 2 a (b = 0) and + 1 (b = 1) at two
        // horizontally adjacent pixels = four consecutive floats of output row 2 ho + a
        if (t < 16 * 16 * TN) {
            const int r = t / (16 * TN), j = t % (16 * TN);
            const int ml = 2 * r, nl = 2 * j;
            const int ph = nl / PTW, pw = nl % PTW;
            const int ho = ho0 + ph, wo = wo0 + pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < NW; ++w) {  // fixed order
                const float2 u0 = *reinterpret_cast<const float2*>(red + (w * 32 + ml) * LDT + nl);
                const float2 u1 = *reinterpret_cast<const float2*>(red + (w * 32 + ml + 1) * LDT + nl);
                v.x += u0.x, v.y += u1.x, v.z += u0.y, v.w += u1.y;
            }
            if (m0 + ml < p.M && img0 < nimg && ho < p.Hout && wo < p.Wout) {
                const int co = (m0 + ml) >> 2, a = r & 1;
                const int64_t addr = (((int64_t)img0 * p.Ctot + co) * (2 * p.Hout) + 2 * ho + a) * (2 * p.Wout) + 2 * wo;
                if (p.scale) {
                    const float sc = p.scale[co], sh = p.shift[co];
                    v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
                } else if (p.shift) {
                    const float sh = p.shift[co];
                    v.x += sh, v.y += sh, v.z += sh, v.w += sh;
                }
                if (p.residual) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.residual + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.accumulate) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.D + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4*>(p.D + addr) = v;
            }
        }
    } else if (t < 32 * 8 * TN) {
        const int ml = t / (8 * TN), c4 = t % (8 * TN);  // 32 channels x 8 TN pixel quads
        const int m = m0 + ml, nl = 4 * c4;
        const int ph = nl / PTW, pw = nl % PTW;
        const int ho = ho0 + ph, wo = wo0 + pw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {  // fixed order: the sum does not depend on the run
            const float4 u = *reinterpret_cast<const float4*>(red + (w * 32 + ml) * LDT + nl);
            v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
        }
        if (m < p.M && img0 < nimg && ho < p.Hout && wo < p.Wout) {
            const int64_t addr = ((int64_t)img0 * p.Ctot + m) * p.HoWo + ho * p.Wout + wo;
            const int me = p.grp_imgs > 0 ? (img0 / p.grp_imgs) * p.M + m : m;
            if (p.scale) {
                const float sc = p.scale[me], sh = p.shift[me];
                v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
            } else if (p.shift) {
                const float sh = p.shift[me];
                v.x += sh, v.y += sh, v.z += sh, v.w += sh;
            }
            if (p.residual) {
                const float4 rr = *reinterpret_cast<const float4*>(p.residual + addr);
                v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
            }
            if (p.accumulate) {
                const float4 rr = *reinterpret_cast<const float4*>(p.D + addr);
                v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
            }
            if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<float4*>(p.D + addr) = v;
        }
    }
#ifdef BF3_TIMING
    if (lane == 0) {  // one record per wave: prologue round trip, K loop (of which staging: [3] low half), reduction + epilogue, start
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long ts3 = wall_clock64();
        const int b = ((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) & 8191;
        g_bf3_stamp[b * 8 + 0] = ts1 - ts0;
        g_bf3_stamp[b * 8 + 1] = ts2 - ts1;
        g_bf3_stamp[b * 8 + 2] = ts3 - ts2;
        g_bf3_stamp[b * 8 + 3] = ts0;
        g_bf3_stamp[b * 8 + 4] = t_stage;
    }
#endif
}

template <int PTH, int PTW, int TN = 2, int KS = 3>
int launch_bf3_ks_tile(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a_split, int64_t grp_bytes, int nimg) {
    constexpr int NPIX = (PTH + KS - 1) * (PTW + KS - 1), WREG = (NPIX * PIXB + 15) & ~15, RED = 8 * 32 * (32 * TN + 4) * 4;
    constexpr size_t lds = (size_t)(8 * WREG > RED ? 8 * WREG : RED);
    static_assert(lds <= 160 * 1024, "patch regions do not fit");
    auto kern = k_conv_bf3_ks<PTH, PTW, TN, KS>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IVLN_E_HIP;
        attr_done = true;
    }
    const int tiles_w = d.Wout / PTW, tiles_h = d.Hout / PTH;
    dim3 grid(tiles_w * tiles_h * nimg, (d.M + 31) / 32, 1);
    IVLN_LAUNCH_FAMILY_NAMED("k_conv_bf3_ks", kern, grid, dim3(512), lds, s, d, a_split, (long long)grp_bytes, tiles_w, tiles_h, nimg);
    return IVLN_OK;
}

// Eligibility of the K-split-over-waves kernel: deep 3x3 convs over few pixels.  mode: 0 = heuristic, 1 = insist (tests, tuning)
int bf3_ks_launch(ivln_gemm_desc& d, hipStream_t s, int nimg, int mode, int tn_pin = 0) {
    if (d.Cin % CB != 0 || d.Cin < 8 * CB || d.stat_partials || d.splits > 1) return IVLN_E_UNSUPPORTED;
    if (d.bmode == BMODE_CONV_K2) {  // the stacked transposed-conv classes: 64-pixel tiles, no image groups
        if (d.grp_imgs > 0 || (d.Wout != 8 && d.Wout != 16 && d.Wout != 32) || d.Hout % (64 / d.Wout) != 0 || (d.in_img_stride & 3) ||
            (((uintptr_t)d.B) & 15) || (int64_t)nimg * d.in_img_stride * 4 >= (int64_t)1 << 31)
            return IVLN_E_UNSUPPORTED;
        const int64_t wgs2 = (int64_t)(d.N / 64) * ((d.M + 31) / 32);
        if (mode == 0 && (wgs2 > 2 * (int64_t)ivln_cu_count() || wgs2 < ivln_cu_count() / 4)) return IVLN_E_UNSUPPORTED;
        d.splits = 1;
        const unsigned char* a2 = (const unsigned char*)d.A_split;
        const int64_t gb2 = d.a_split_grp_stride * 4;
        if (d.Wout == 8) return launch_bf3_ks_tile<8, 8, 2, 2>(d, s, a2, gb2, nimg);
        if (d.Wout == 16) return launch_bf3_ks_tile<4, 16, 2, 2>(d, s, a2, gb2, nimg);
        return launch_bf3_ks_tile<2, 32, 2, 2>(d, s, a2, gb2, nimg);
    }
    if (d.Wout != 8 && d.Wout != 16 && d.Wout != 32) return IVLN_E_UNSUPPORTED;
    const int64_t wgs = (int64_t)(d.N / 64) * ((d.M + 31) / 32);
    // 32-pixel tiles where 64-pixel ones leave half of the CUs without a workgroup; IVLN_BF3_KS_TN = 1 | 2 pins one (tuning)
    constexpr int tn_env = 0;
    const bool small = tn_pin ? tn_pin == 1 : (tn_env ? tn_env == 1 : 2 * wgs <= ivln_cu_count());
    const int pth = (small ? 32 : 64) / d.Wout;
    if (d.Hout % pth != 0 || (d.in_img_stride & 3) || (((uintptr_t)d.B) & 15)) return IVLN_E_UNSUPPORTED;
    if ((int64_t)nimg * d.in_img_stride * 4 >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // byte offsets of the buffer loads
    if (d.grp_imgs > 0 && nimg % d.grp_imgs != 0) return IVLN_E_UNSUPPORTED;
    if (mode == 0 && (wgs > 2 * (int64_t)ivln_cu_count() || wgs < ivln_cu_count() / 4)) return IVLN_E_UNSUPPORTED;  // (more pixels: the tiled kernel fills the chip by itself)
    const int64_t gb = d.a_split_grp_stride * 4;
    d.splits = 1;
    const unsigned char* a = (const unsigned char*)d.A_split;
    if (small) {
        if (d.Wout == 8) return launch_bf3_ks_tile<4, 8, 1>(d, s, a, gb, nimg);
        if (d.Wout == 16) return launch_bf3_ks_tile<2, 16, 1>(d, s, a, gb, nimg);
        return launch_bf3_ks_tile<1, 32, 1>(d, s, a, gb, nimg);
    }
    if (d.Wout == 8) return launch_bf3_ks_tile<8, 8>(d, s, a, gb, nimg);
    if (d.Wout == 16) return launch_bf3_ks_tile<4, 16>(d, s, a, gb, nimg);
    return launch_bf3_ks_tile<2, 32>(d, s, a, gb, nimg);
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 convs with a deep K (RedNet's bottleneck reductions 1024 -> 256, 2048 -> 512, its 2048-channel expansions and skip
// convs: rednet.py:20-65, 244-248) on the same arithmetic, K split over the waves of a workgroup, NO LDS in the K loop:
//   * a workgroup (8 waves) owns 32 output channels x 128 consecutive pixels over the whole K, wave w the chunks
//     [w, w + 1) * ceil(nch / 8);
//   * a 1x1 conv needs no patch: the B fragment of v_mfma_f32_32x32x16_bf16 - column n, 8 consecutive k per lane - is
//     built IN REGISTERS.  Lane (l31, half) loads, for its 8 channels (8 half ..) of the chunk, the four consecutive pixels
//     4 l31 .. 4 l31 + 3 (one buffer_load_b128 per channel, 512 contiguous bytes per half-wave), splits channel pairs
//     into the three bf16 pieces (v_cvt_pk_bf16_f32 packs a pair into the word the fragment wants) and has the B fragments
//     of FOUR pixel tiles - tile e holds pixel 4 l31 + e in column l31 - without a single LDS access or barrier;
//   * weights: the split image of ivln_conv_split_weights_f32 (KS = 1), global -> registers four chunks ahead;
//   * the eight partial tiles meet in LDS at the end, summed in a fixed order, fused epilogue, 16-byte stores.
// An activation element is split once per 32-channel tile that reads it (176 VALU operations per lane and chunk beside 24
// MFMAs): the VALU work rides under the other wave's MFMAs.
// ------------------------------------------------------------------------------------------------------------------
// WT (wave tiles, short K: 64 ... 256 input channels - the bottleneck expansions 64 -> 256 ... 256 -> 1024 with their residual):
// no K split at all - each of the workgroup's 4 waves owns its own 32 x 128 tile over the whole K and finishes it alone
// (wave-private LDS for the transposing epilogue, no barrier anywhere); these launches are bound by their activation bytes.
template <bool WT>
__global__ __launch_bounds__(WT ? 256 : 512, 2) void k_conv1x1_bf3_ks(const ivln_gemm_desc p, const unsigned char* a_split, long long a_grp_bytes) {
    constexpr int NW = WT ? 4 : 8, TN = 4, DA = 4, BN = 128;
    constexpr int LDT = BN + 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    // grid = (channel tiles, pixel tiles): consecutive workgroup ids - one XCD's contiguous range after the remap - share their
    // PIXELS.  A 1x1 conv's activation tile (128 pixels x Cin x 4 bytes: 512 KB at 1024 channels) outweighs a channel tile's
    // weights (196 KB), and with the pixel tile as the fast index every XCD fetched every activation: 8 x the input bytes
    // over the fabric, which is what bounded the first version (256 x 4096 x 1024: 134 MB in 23 us).
    // (WT: the workgroup's four waves own four pixel tiles of ONE channel tile - they stream the same weights; four channel tiles of
    //  one pixel tile, i.e. shared activations, measured 38 us per RedNet step slower: profiles/r06_predsem_ab_wtshare.txt)
    const int n0 = WT ? (bid.y * NW + wave) * BN : bid.y * BN, m0 = bid.x * 32;
    const int nch = p.Cin / CB;
    const int HW = p.HoWo;
    const int cpw = WT ? nch : (nch + NW - 1) / NW;
    const int c0 = WT ? 0 : min(nch, wave * cpw), c1 = WT ? nch : min(nch, c0 + cpw);
    const int n_img0 = min(n0, p.N - 1) / HW;
    const int grp = p.grp_imgs > 0 ? n_img0 / p.grp_imgs : 0;

    const int nq = n0 + 4 * l31;  // this lane's four pixels (one image: HW is a multiple of 4)
    const int qimg = nq / HW, qpp = nq - qimg * HW;
    const unsigned hw4 = (unsigned)HW * 4u;
    const unsigned xvo = nq < p.N ? (unsigned)(((int64_t)qimg * p.in_img_stride + qpp + (int64_t)(8 * half) * HW) * 4) : OOB;
    const __amdgpu_buffer_rsrc_t rB = bf3_rsrc(p.B);
    auto load_x = [&](int c, v4i (&xb)[8]) {  // (the channel steps ride in the SCALAR offset, which the range check ignores: no VALU per load)
        const int so = c * CB * HW * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)xvo, so + j * (int)hw4, 0);
    };
    const int mt = min(m0 / 32, (p.M + 31) / 32 - 1);
    const __amdgpu_buffer_rsrc_t rA = bf3_rsrc(a_split + (int64_t)grp * a_grp_bytes + (int64_t)mt * nch * (3 * 1024));
    const int c_last = max(c1 - 1, 0);
    auto load_a = [&](int c, v4i (&ab)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ab[pl] = __builtin_amdgcn_raw_buffer_load_b128(rA, (min(c, c_last) * 3 + pl) * 1024 + lane * 16, 0, 0);
    };
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

#ifdef BF3_TIMING
    const unsigned long long ts0 = lane == 0 ? wall_clock64() : 0ull;
    unsigned long long ts1 = ts0, tc_split = 0, tc_load = 0, tc_mfma = 0;  // (shader cycles inside the K loop: waits + split | load issue | MFMA issue)
#endif
    if (c0 < c1 && (!WT || n0 < p.N)) {
        v4i xb[2][8], ab[DA][3];
#pragma unroll
        for (int d = 0; d < DA; ++d) load_a(c0 + d, ab[d]);
        load_x(c0, xb[0]);
        load_x(min(c0 + 1, c_last), xb[1]);
#ifdef BF3_TIMING
        __builtin_amdgcn_s_waitcnt(0);  // (timing build only: the prologue's round trip as a phase of its own)
        ts1 = lane == 0 ? wall_clock64() : 0ull;
#endif
        for (int cc = c0; cc < c1; cc += DA) {
#pragma unroll
            for (int k = 0; k < DA; ++k) {
                const int c = cc + k;
                v4i bq[TN][3];
                bf16x8 a[3];
#ifdef BF3_TIMING
                const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
                // (loads stay outside the branch: the compiler then knows how many are in flight at every wait)
                if (c < c1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            uint32_t H, M, L;
#ifdef BF3_PROBE_NO_SPLIT  // (tools/conv_bf3_ks_phases.py: what the K loop costs without its VALU work - wrong results)
                            H = xb[k & 1][2 * i][e], M = xb[k & 1][2 * i + 1][e], L = H ^ M;
#else
                            split3_pair(__int_as_float(xb[k & 1][2 * i][e]), __int_as_float(xb[k & 1][2 * i + 1][e]), H, M, L);
#endif
                            bq[e][0][i] = (int)H, bq[e][1][i] = (int)M, bq[e][2][i] = (int)L;
                        }
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, ab[k][pl]);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef BF3_TIMING
                const unsigned long long tl1 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
#endif
                load_x(min(c + 2, c_last), xb[k & 1]);  // two chunks ahead
                load_a(c + DA, ab[k]);                   // DA chunks ahead
                __builtin_amdgcn_sched_barrier(0);
#ifdef BF3_TIMING
                const unsigned long long tl2 = __builtin_readcyclecounter();
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (c < c1) {
#define IVLN_BF3_PROD(PA, PB)                            \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)    \
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], __builtin_bit_cast(bf16x8, bq[tn][PB]), acc[tn], 0, 0, 0)
#ifdef BF3_PROBE_NO_MFMA  // (... and without its matrix work: one product of the six)
                    IVLN_BF3_PROD(0, 0);
#else
                    IVLN_BF3_PROD(0, 2);
                    IVLN_BF3_PROD(1, 1);
                    IVLN_BF3_PROD(2, 0);
                    IVLN_BF3_PROD(0, 1);
                    IVLN_BF3_PROD(1, 0);
                    IVLN_BF3_PROD(0, 0);
#endif
#undef IVLN_BF3_PROD
                }
#ifdef BF3_TIMING
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long tl3 = __builtin_readcyclecounter();
                tc_split += tl1 - tl0, tc_load += tl2 - tl1, tc_mfma += tl3 - tl2;
#endif
            }
        }
    }
    // ---- epilogue.
    // WT: straight from the accumulators.  Tile e holds pixel 4 l31 + e in column l31, so register r of the four tiles IS the
    // float4 of channel (r & 3) + 8 (r >> 2) + 4 half at pixels 4 l31 .. + 3: one 16-byte store per register, 512 contiguous
    // bytes per half-wave, no LDS, no transposition.  The channel step of register r rides in the scalar offset of the
    // buffer accesses (the range check ignores it).  The operands (residual quads, scale / shift) are requested FIRST, all of
    // them: with one wave per SIMD every dependent round trip would otherwise sit exposed at the end of a 20 us kernel.
    // KS: the eight waves' partial tiles meet in LDS - red[wave][32 channels][4 tiles x 32 columns (+4)] -, summed in a fixed order. ----
#ifdef BF3_TIMING
    const unsigned long long ts2 = lane == 0 ? wall_clock64() : 0ull;
#endif
    // IVLN_D_NCHW_UP2X4 (the 2 x 2 stride-2 upsampling branches and the final deconv, rednet.py:239-245, 217-218: one tap per
    // parity class): rows 4 c + 2 a (b = 0) and + 1 (b = 1) are registers r and r + 1 of one lane, its four pixels of a class-grid
    // row are eight consecutive floats of output row 2 ho + a - two 16-byte stores per register pair, plain stores (no
    // scalar-offset buffer stores here: nothing for the store guard to do).
    const bool up = p.dmode == DMODE_NCHW_UP2X4;  // (uniform)
    if (WT && up) {
        const int n = n0 + 4 * l31;
        if (n < p.N) {
            const int img = n / HW, pp = n - img * HW;
            const int ho = pp / p.Wout, wo = pp - ho * p.Wout;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = m0 + 4 * half + (r & 3) + 8 * (r >> 2);  // row of b = 0; + 1 is b = 1 (M % 4 == 0: in or out together)
                if (m >= p.M) continue;
                const int co = m >> 2, a = (m >> 1) & 1;
                const int64_t addr = (((int64_t)img * p.Ctot + co) * (2 * p.Hout) + 2 * ho + a) * (2 * p.Wout) + 2 * wo;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[2 * e] = acc[e][r], v[2 * e + 1] = acc[e][r + 1];
                const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = p.scale ? fmaf(v[e], sc, sh) : v[e] + sh;
                if (p.residual) {
                    const float4 r0 = *reinterpret_cast<const float4*>(p.residual + addr), r1 = *reinterpret_cast<const float4*>(p.residual + addr + 4);
                    v[0] += r0.x, v[1] += r0.y, v[2] += r0.z, v[3] += r0.w, v[4] += r1.x, v[5] += r1.y, v[6] += r1.z, v[7] += r1.w;
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<float4*>(p.D + addr) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.D + addr + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    } else
    if constexpr (WT) {
        const int n = n0 + 4 * l31;
        const int img = min(n, p.N - 1) / HW, pp = n - img * HW;
        const int mrem = n < p.N ? p.M - m0 - 4 * half : 0;  // register r is inside the tensor iff its channel step is below this
        const int me0 = (p.grp_imgs > 0 ? (img / p.grp_imgs) * p.M : 0) + m0 + 4 * half;
        const unsigned off0 = (unsigned)(((img * p.Ctot + m0 + 4 * half) * HW + pp) * 4);
        const __amdgpu_buffer_rsrc_t rR = bf3_rsrc(p.residual), rS = bf3_rsrc(p.scale), rH = bf3_rsrc(p.shift), rD = bf3_rsrc(p.D);
        const bool has_res = p.residual != nullptr, has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
        const float relu_lo = p.relu ? 0.f : -__builtin_huge_valf();
        const bool res_post = p.residual_after_relu != 0;
        v4i rres[16];
        float esc[16], esh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = (r & 3) + 8 * (r >> 2);  // channel step of the register
            const bool ok = cs < mrem;
            esc[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, ok && has_sc ? me0 * 4 : (int)OOB, cs * 4, 0));
            esh[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rH, ok && has_sh ? me0 * 4 : (int)OOB, cs * 4, 0));
            rres[r] = __builtin_amdgcn_raw_buffer_load_b128(rR, ok && has_res ? (int)off0 : (int)OOB, cs * HW * 4, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = (r & 3) + 8 * (r >> 2);
            float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            // STRAIGHT-LINE code between the stores (absent scale = 1, absent shift / residual = the zeros the out-of-range loads
            // returned; `accumulate` is declined by the launcher): see the fused tail of k_conv_bf3 for what a uniform branch
            // behind a 16-byte buffer store with a scalar offset does to the store's data registers.
            const float sc = has_sc ? esc[r] : 1.f, sh = esh[r];
            v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
            // the residual in front of the ReLU (a bottleneck's identity) or behind it (ivln_gemm_desc.residual_after_relu: the
            // decoder's skip adds) - selects, no branch
            const float r0 = __int_as_float(rres[r][0]), r1 = __int_as_float(rres[r][1]), r2 = __int_as_float(rres[r][2]), r3 = __int_as_float(rres[r][3]);
            v.x += res_post ? 0.f : r0, v.y += res_post ? 0.f : r1, v.z += res_post ? 0.f : r2, v.w += res_post ? 0.f : r3;
            const float lo = relu_lo;  // 0 with ReLU, -inf without: fmaxf(x, -inf) == x
            v.x = fmaxf(v.x, lo), v.y = fmaxf(v.y, lo), v.z = fmaxf(v.z, lo), v.w = fmaxf(v.w, lo);
            v.x += res_post ? r0 : 0.f, v.y += res_post ? r1 : 0.f, v.z += res_post ? r2 : 0.f, v.w += res_post ? r3 : 0.f;
            v4i o;
            o[0] = __float_as_int(v.x), o[1] = __float_as_int(v.y), o[2] = __float_as_int(v.z), o[3] = __float_as_int(v.w);
            __builtin_amdgcn_raw_buffer_store_b128(o, rD, cs < mrem ? (int)off0 : (int)OOB, cs * HW * 4, 0);
            BF3_STORE_GUARD();
        }
    } else if (up) {
        // the eight waves' partial tiles meet in LDS as below; an item = (channel of the tile, a, pixel pair): rows 4 c + 2 a and + 1
        // at two horizontally adjacent class-grid pixels = four consecutive floats of output row 2 ho + a
        float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + tn * 32 + l31] = acc[tn][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = t + it * 512, r = item >> 6, j = item & 63;  // 16 (channel, a) x 64 pixel pairs
            const int ml = 2 * r, nl = 2 * j, n = n0 + nl;
            const int e0 = nl & 3, q = nl >> 2;  // tile e, column q holds pixel 4 q + e
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < NW; ++w) {  // fixed order
                const float* r0 = red + (w * 32 + ml) * LDT + q;
                const float* r1 = r0 + LDT;
                v.x += r0[e0 * 32], v.y += r1[e0 * 32], v.z += r0[(e0 + 1) * 32], v.w += r1[(e0 + 1) * 32];
            }
            if (m0 + ml < p.M && n < p.N) {
                const int img = n / HW, pp = n - img * HW;
                const int ho = pp / p.Wout, wo = pp - ho * p.Wout;
                const int co = (m0 + ml) >> 2, a = r & 1;
                const int64_t addr = (((int64_t)img * p.Ctot + co) * (2 * p.Hout) + 2 * ho + a) * (2 * p.Wout) + 2 * wo;
                if (p.scale) {
                    const float sc = p.scale[co], sh = p.shift[co];
                    v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
                } else if (p.shift) {
                    const float sh = p.shift[co];
                    v.x += sh, v.y += sh, v.z += sh, v.w += sh;
                }
                if (p.residual) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.residual + addr);
                    v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
                }
                if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4*>(p.D + addr) = v;
            }
        }
    } else {
    constexpr int ITEMS = 2;  // 32 channels x 32 pixel quads = 1024 items over the workgroup's 512 threads
    v4i rres[ITEMS];
    float esc[ITEMS], esh[ITEMS];
    int eoff[ITEMS];  // element offset of the item's quad in D / residual (< 2^29: checked by the launcher), -1 = outside
    const __amdgpu_buffer_rsrc_t rR = bf3_rsrc(p.residual), rS = bf3_rsrc(p.scale), rH = bf3_rsrc(p.shift);
    const bool has_res = p.residual != nullptr, has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
    auto request = [&](int it) {
        const int item = t + it * 512, ml = item >> 5, q = item & 31;
        const int m = m0 + ml, n = n0 + 4 * q;
        const bool ok = m < p.M && n < p.N;
        const int img = n / HW, pp = n - img * HW;
        const int off = (img * p.Ctot + m) * HW + pp;
        const int me = p.grp_imgs > 0 ? (img / p.grp_imgs) * p.M + m : m;
        eoff[it] = ok ? off : -1;
        esc[it] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, ok && has_sc ? me * 4 : (int)OOB, 0, 0));
        esh[it] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rH, ok && has_sh ? me * 4 : (int)OOB, 0, 0));
        rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rR, ok && has_res ? off * 4 : (int)OOB, 0, 0);
    };
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) request(it);
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + tn * 32 + l31] = acc[tn][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = t + it * 512, ml = item >> 5, q = item & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {  // fixed order; tile e, column q = pixel 4 q + e
            const float* rr = red + (w * 32 + ml) * LDT + q;
            v.x += rr[0], v.y += rr[32], v.z += rr[64], v.w += rr[96];
        }
        if (eoff[it] >= 0) {
            const int64_t addr = eoff[it];
            const float sc = esc[it], sh = esh[it];
            if (p.scale) v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
            else if (p.shift) v.x += sh, v.y += sh, v.z += sh, v.w += sh;
            // (absent residual: the out-of-range loads returned zeros, but adding them would turn a -0 into +0: the branch stays)
            const bool res_pre = p.residual && !p.residual_after_relu, res_post = p.residual && p.residual_after_relu;
            if (res_pre)
                v.x += __int_as_float(rres[it][0]), v.y += __int_as_float(rres[it][1]), v.z += __int_as_float(rres[it][2]),
                    v.w += __int_as_float(rres[it][3]);
            if (p.accumulate) {
                const float4 ra = *reinterpret_cast<const float4*>(p.D + addr);
                v.x += ra.x, v.y += ra.y, v.z += ra.z, v.w += ra.w;
            }
            if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            if (res_post)  // (ivln_gemm_desc.residual_after_relu: the decoder's skip adds)
                v.x += __int_as_float(rres[it][0]), v.y += __int_as_float(rres[it][1]), v.z += __int_as_float(rres[it][2]),
                    v.w += __int_as_float(rres[it][3]);
            *reinterpret_cast<float4*>(p.D + addr) = v;
        }
    }
    }
#ifdef BF3_TIMING
    if (lane == 0) {  // one record per wave: prologue round trip, K loop, epilogue (100 MHz wall clock), start offset of the wave
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long ts3 = wall_clock64();
        const int b = ((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) & 8191;
        g_bf3_stamp[b * 8 + 0] = ts1 - ts0;
        g_bf3_stamp[b * 8 + 1] = ts2 - ts1;
        g_bf3_stamp[b * 8 + 2] = ts3 - ts2;
        g_bf3_stamp[b * 8 + 3] = ts0;
        g_bf3_stamp[b * 8 + 4] = tc_split;
        g_bf3_stamp[b * 8 + 5] = tc_load;
        g_bf3_stamp[b * 8 + 6] = tc_mfma;
    }
#endif
}

// Eligibility of the 1x1 kernels in their two forms.  mode: 0 = heuristic, 1 = insist
int bf3_1x1_ks_launch(ivln_gemm_desc& d, hipStream_t s, int mode, int form_pin = -1) {
    if (d.stride != 1 || d.pad != 0 || d.Cin % CB != 0 || d.Cin < 4 * CB || d.stat_partials || d.splits > 1) return IVLN_E_UNSUPPORTED;
    if ((d.HoWo & 3) || (d.in_img_stride & 3) || ((((uintptr_t)d.B) | ((uintptr_t)d.D) | ((uintptr_t)d.residual)) & 15)) return IVLN_E_UNSUPPORTED;
    const int64_t nimg = d.N / d.HoWo;
    if (nimg * d.in_img_stride * 4 >= (int64_t)1 << 31 || nimg * d.Ctot * d.HoWo * 4 >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // byte offsets of the buffer loads
    if (d.grp_imgs > 0 && ((int64_t)d.grp_imgs * d.HoWo) % 128 != 0) return IVLN_E_UNSUPPORTED;  // a tile's pixels share one weight set
    const int nch = d.Cin / CB;
    // wave tiles (no K split) up to 16 chunks, K split over the 8 waves from 32 (in between - 272 ... 496 channels - the K-split
    // form with short slices); IVLN_BF3_1X1_FORM = ks | wt pins one (tuning)
    constexpr const char* form_env = nullptr;
    bool wt = nch <= 16;
    // (32 chunks over MANY pixels - 256 x 16384 x 512, layer 3's first reduction at 8 + 8 images - would be four rounds of one
    //  K-split workgroup per CU: the wave tiles take it, 43.6 against 51.5 us on the tiled 1x1 form and 52.1 K-split, tools/conv_cfg_sweep.py)
    constexpr int maxwg0 = 2;
    if (nch == 32 && (int64_t)((d.N + 127) / 128) * ((d.M + 31) / 32) > maxwg0 * (int64_t)ivln_cu_count()) wt = true;
    if (form_env) wt = form_env[0] == 'w';
    if (form_pin >= 0) wt = form_pin == 1;
    if (wt && nch > 64) wt = false;
    if (wt && d.accumulate) return IVLN_E_UNSUPPORTED;  // (the wave-tile epilogue is straight-line code: no D += form)
    const int64_t mtiles = (d.M + 31) / 32;
    const int64_t wgs = wt ? (int64_t)((d.N + 511) / 512) * mtiles : (int64_t)((d.N + 127) / 128) * mtiles;
    constexpr int maxwg_env = 2;  // tuning: rounds of one workgroup per CU
    if ((d.N + 127) / 128 > 65535) return IVLN_E_UNSUPPORTED;
    if (mode == 0) {
        // (measured inside RedNet: wins at 128 ... 512 workgroups - 256 x 4096 x 1024 35.7 -> 28 us, 256 x 2048 x 1024 28.4 -> 21.6 -,
        //  loses at 64 - 512 x 512 x 2048 - and at 1024 - 256 x 16384 x 512, four rounds of one workgroup per CU)
        if (!wt && (d.Cin < 32 * CB || wgs > maxwg_env * (int64_t)ivln_cu_count() || wgs < ivln_cu_count() / 2)) return IVLN_E_UNSUPPORTED;
        if (wt && wgs < ivln_cu_count()) return IVLN_E_UNSUPPORTED;  // (too few wave tiles to fill the chip: the tiled GEMMs split K)
    }
    d.splits = 1;
    const unsigned char* a = (const unsigned char*)d.A_split;
    const long long gb = (long long)(d.a_split_grp_stride * 4);
    if (wt) {
        constexpr size_t lds = 0;  // (the wave-tile form keeps everything in registers)
        dim3 grid((unsigned)mtiles, (unsigned)((d.N + 511) / 512), 1);
        IVLN_LAUNCH_FAMILY(k_conv1x1_bf3_ks<true>, grid, dim3(256), lds, s, d, a, gb);
    } else {
        constexpr size_t lds = (size_t)8 * 32 * (128 + 4) * 4;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute((const void*)k_conv1x1_bf3_ks<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IVLN_E_HIP;
            attr_done = true;
        }
        dim3 grid((unsigned)mtiles, (unsigned)((d.N + 127) / 128), 1);
        IVLN_LAUNCH_FAMILY(k_conv1x1_bf3_ks<false>, grid, dim3(512), lds, s, d, a, gb);
    }
    return IVLN_OK;
}

// OIHW fp32 weights -> [32-channel tile][16-channel chunk][tap, padded][piece][lane] x 8 bf16: lane (l31, half) of tile mt
// holds W[32 mt + l31][16 c + 8 half .. + 7][tap] - the A operand of v_mfma_f32_32x32x16_bf16 as one 16-byte load.
__global__ __launch_bounds__(256) void k_conv_bf3_pack(const float* __restrict__ W, int M, int Cin, int KS, uint16_t* __restrict__ out,
                                                       int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int KK = KS * KS, KKP = bf3_taps_padded(KS), nch = (Cin + CB - 1) / CB;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t q = idx >> 9;
    const int pl = (int)(q % 3);
    q /= 3;
    const int r = (int)(q % KKP);
    q /= KKP;
    const int c = (int)(q % nch);
    const int mt = (int)(q / nch);
    const int m = mt * 32 + (lane & 31), ci = c * CB + (lane >> 5) * 8 + e;
    const float v = (m < M && ci < Cin && r < KK) ? W[((int64_t)m * Cin + ci) * KK + r] : 0.f;
    uint32_t h, mm, l;
    split3(v, h, mm, l);
    out[idx] = (uint16_t)(pl == 0 ? h : (pl == 1 ? mm : l));
}

template <int KS, int TM, int WM, int WN, int PTH, int PTW, int IMGS, int DA, int ST = 1>
int launch_bf3(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a_split, int64_t grp_bytes, int nimg, int cps) {
    constexpr int NTB = 64 * WM * WN, BM = 32 * TM * WM, BN = 64 * WN;
    constexpr int PH = ST == 2 ? PTH + 1 : PTH + KS - 1, PWR = ST == 2 ? PTW + 1 : PTW + KS - 1;
    constexpr int NPIX = IMGS * (ST == 2 ? 4 : 1) * PH * PWR * bf3_stage_chunks(KS);  // (stride 2: four phase planes)
    constexpr size_t lds = (size_t)(NPIX * PIXB > 32 * (BN + 4) * 4 ? NPIX * PIXB : 32 * (BN + 4) * 4);
    if constexpr (lds > 160 * 1024) {
        return IVLN_E_UNSUPPORTED;  // (a stride-2 tile whose planes do not fit: the dispatcher picks another)
    } else {
    auto kern = k_conv_bf3<KS, TM, WM, WN, PTH, PTW, IMGS, DA, false, ST>;
    static bool attr_done = false;  // (idempotent; a race only repeats the call)
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IVLN_E_HIP;
        attr_done = true;
    }
    const int tiles_w = (d.Wout + PTW - 1) / PTW, tiles_h = (d.Hout + PTH - 1) / PTH;
    const int groups = (nimg + IMGS - 1) / IMGS;
    dim3 grid(tiles_w * tiles_h * groups, (d.M + BM - 1) / BM, d.splits);
    IVLN_LAUNCH_FAMILY_NAMED("k_conv_bf3", kern, grid, dim3(NTB), lds, s, d, a_split, (long long)grp_bytes, tiles_w, tiles_h, nimg, cps);
    return IVLN_OK;
    }
}

// The fused bottleneck tail (ivln_gemm_desc.fuse_*): the 64 x 128 / 128 x 128 tiles of the 3x3 kernel with the 1x1 expansion behind.
template <int TM, int WM, int WN, int DA>
int launch_bf3_fused(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a_split, int64_t grp_bytes, int nimg) {
    constexpr int KS = 3, PTW = 32, NTB = 64 * WM * WN, BM = 32 * TM * WM, BN = 64 * WN, PTH = BN / PTW;
    static_assert(NTB == 256, "four waves");
    constexpr int NPIX = (PTH + 2) * (PTW + 2);
    constexpr size_t ybytes = (size_t)BM * (BN + 4) * 4, lds = (size_t)NPIX * PIXB > ybytes ? (size_t)NPIX * PIXB : ybytes;
    auto kern = k_conv_bf3<KS, TM, WM, WN, PTH, PTW, 1, DA, true>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IVLN_E_HIP;
        attr_done = true;
    }
    const int tiles_w = d.Wout / PTW, tiles_h = d.Hout / PTH;
    dim3 grid(tiles_w * tiles_h * nimg, 1, 1);
    const int nch = d.Cin / CB;
    IVLN_LAUNCH_FAMILY_NAMED("k_conv_bf3", kern, grid, dim3(NTB), lds, s, d, a_split, (long long)grp_bytes, tiles_w, tiles_h, nimg, nch);
    return IVLN_OK;
}

// pixel tile of a block for its pixel count BN and the output width: rows x columns x images
struct Bf3Px { int pth, ptw, imgs; };
inline Bf3Px bf3_px(int BN, int Wout) {
    const int ptw = Wout > 16 ? 32 : (Wout > 8 ? 16 : 8);
    const int pth = ptw == 32 ? BN / 32 : (ptw == 16 ? (BN >= 256 ? 16 : 8) : 8);
    return {pth, ptw, BN / (ptw * pth)};
}

template <int KS, int TM, int WM, int WN, int DA, int ST = 1>
int launch_bf3_px(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a, int64_t gb, int nimg, int cps) {
    constexpr int BN = 64 * WN;
    if constexpr (ST == 2) {  // stride-2 3x3: the 128- and 256-pixel tiles
        static_assert(KS == 3 && (BN == 128 || BN == 256), "stride-2 tiles");
        if constexpr (BN == 256) {
            if (d.Wout > 16) return launch_bf3<KS, TM, WM, WN, 8, 32, 1, DA, 2>(d, s, a, gb, nimg, cps);
            if (d.Wout > 8) return launch_bf3<KS, TM, WM, WN, 16, 16, 1, DA, 2>(d, s, a, gb, nimg, cps);
            return launch_bf3<KS, TM, WM, WN, 8, 8, 4, DA, 2>(d, s, a, gb, nimg, cps);
        } else {
            if (d.Wout > 16) return launch_bf3<KS, TM, WM, WN, 4, 32, 1, DA, 2>(d, s, a, gb, nimg, cps);
            if (d.Wout > 8) return launch_bf3<KS, TM, WM, WN, 8, 16, 1, DA, 2>(d, s, a, gb, nimg, cps);
            return launch_bf3<KS, TM, WM, WN, 8, 8, 2, DA, 2>(d, s, a, gb, nimg, cps);
        }
    } else
    if constexpr (BN == 512 && KS == 1) {
        return IVLN_E_UNSUPPORTED;  // (four staged chunks of 512 pixels do not fit the LDS)
    } else if constexpr (BN == 512) {
        if (d.Wout > 16) return launch_bf3<KS, TM, WM, WN, 16, 32, 1, DA>(d, s, a, gb, nimg, cps);
        if (d.Wout > 8) return launch_bf3<KS, TM, WM, WN, 16, 16, 2, DA>(d, s, a, gb, nimg, cps);
        if constexpr (KS == 3) return launch_bf3<KS, TM, WM, WN, 8, 8, 8, DA>(d, s, a, gb, nimg, cps);
        return IVLN_E_UNSUPPORTED;
    } else if constexpr (BN == 256) {
        if (d.Wout > 16) return launch_bf3<KS, TM, WM, WN, 8, 32, 1, DA>(d, s, a, gb, nimg, cps);
        if (d.Wout > 8) return launch_bf3<KS, TM, WM, WN, 16, 16, 1, DA>(d, s, a, gb, nimg, cps);
        return launch_bf3<KS, TM, WM, WN, 8, 8, 4, DA>(d, s, a, gb, nimg, cps);
    } else {
        if (d.Wout > 16) return launch_bf3<KS, TM, WM, WN, 4, 32, 1, DA>(d, s, a, gb, nimg, cps);
        if (d.Wout > 8) return launch_bf3<KS, TM, WM, WN, 8, 16, 1, DA>(d, s, a, gb, nimg, cps);
        return launch_bf3<KS, TM, WM, WN, 8, 8, 2, DA>(d, s, a, gb, nimg, cps);
    }
}

// block tiles: {channels, pixels, waves}
constexpr int kBf3Cfgs = 7;
constexpr int kBf3BM[kBf3Cfgs] = {32, 64, 64, 128, 64, 128, 32};
constexpr int kBf3BN[kBf3Cfgs] = {512, 512, 256, 256, 128, 128, 256};

template <int KS>
int launch_bf3_ks(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a, int64_t gb, int nimg, int cfg, int cps) {
    // weight taps in flight: a tap is 24 MFMAs = 768 pipe cycles per wave (12 = 384 with one channel tile per wave), an L2 /
    // MALL round trip ~2000
    constexpr int DA2 = KS == 1 || KS == 2 ? 2 : 3, DA1 = KS == 1 || KS == 2 ? 4 : (KS == 7 ? 2 : 3);  // (6 / 9 taps ahead measured the same as 3 and cost 36-72 registers)
    if constexpr (KS == 2) {  // (the transposed-conv classes: 4 M rows per output channel - only the tiles their shapes take)
        switch (cfg) {
            case 2: return launch_bf3_px<KS, 1, 2, 4, DA1>(d, s, a, gb, nimg, cps);   // 64 x 256
            case 3: return launch_bf3_px<KS, 2, 2, 4, DA2>(d, s, a, gb, nimg, cps);   // 128 x 256
            case 4: return launch_bf3_px<KS, 1, 2, 2, DA1>(d, s, a, gb, nimg, cps);   // 64 x 128
            case 5: return launch_bf3_px<KS, 2, 2, 2, DA2>(d, s, a, gb, nimg, cps);   // 128 x 128
            default: return IVLN_E_UNSUPPORTED;
        }
    }
    switch (cfg) {
        case 0: return launch_bf3_px<KS, 1, 1, 8, DA1>(d, s, a, gb, nimg, cps);   // 32 x 512, 8 waves
        case 1: return launch_bf3_px<KS, 2, 1, 8, DA2>(d, s, a, gb, nimg, cps);   // 64 x 512
        case 2: return launch_bf3_px<KS, 1, 2, 4, DA1>(d, s, a, gb, nimg, cps);   // 64 x 256
        case 3: return launch_bf3_px<KS, 2, 2, 4, DA2>(d, s, a, gb, nimg, cps);   // 128 x 256
        case 4: return launch_bf3_px<KS, 1, 2, 2, DA1>(d, s, a, gb, nimg, cps);   // 64 x 128, 4 waves: several workgroups per CU
        case 5: return launch_bf3_px<KS, 2, 2, 2, DA2>(d, s, a, gb, nimg, cps);   // 128 x 128, 4 waves
        default: return launch_bf3_px<KS, 1, 1, 4, DA1>(d, s, a, gb, nimg, cps);  // 32 x 256, 4 waves: two workgroups per CU
    }
}


// stride-2 3x3 (k_conv_bf3<..., ST = 2>): the tiles whose four phase planes fit the LDS
int launch_bf3_s2(const ivln_gemm_desc& d, hipStream_t s, const unsigned char* a, int64_t gb, int nimg, int cfg, int cps) {
    switch (cfg) {
        case 2: return launch_bf3_px<3, 1, 2, 4, 3, 2>(d, s, a, gb, nimg, cps);   // 64 x 256
        case 3: return launch_bf3_px<3, 2, 2, 4, 3, 2>(d, s, a, gb, nimg, cps);   // 128 x 256
        case 4: return launch_bf3_px<3, 1, 2, 2, 3, 2>(d, s, a, gb, nimg, cps);   // 64 x 128
        case 5: return launch_bf3_px<3, 2, 2, 2, 3, 2>(d, s, a, gb, nimg, cps);   // 128 x 128
        case 6: return launch_bf3_px<3, 1, 1, 4, 3, 2>(d, s, a, gb, nimg, cps);   // 32 x 256
        default: return IVLN_E_UNSUPPORTED;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 7x7 same-size conv on the same arithmetic (map CNN, base_il_trainer.py:173-219):
//   dW[co][(ci,kh,kw)] = sum over pixels p of dy[co][p] * x[ci][p + (kh,kw) - pad]      M = Cout, N = Cin * 49, K = pixels.
// K of an MFMA = 16 consecutive pixels of an output row (W = 8: two rows of 8), a lane's 8 k-values = 8 consecutive pixels.
//   * A = dy: a STRIP of 128 pixels (2 / 4 / 8 rows of one image, or two 8x8 images) per (channel, piece) as a row of 256 + 16
//     bytes in LDS: one ds_read_b128 per fragment, 16 lanes on 16 distinct bank quads;
//   * B = x: the column's (kh, kw) moves the fragment by kw PIXELS = 2 kw bytes in a bf16 row, which ds_read_b128 cannot
//     address.  The patch of the strip ((rows + 6) x (W + 6) pixels of the <= 7 / 12 input channels the column tile touches)
//     is kept as TWO copies per piece, pixel pairs (2j, 2j+1) and (2j+1, 2j+2): every kw is 4-byte aligned in one of them and
//     a fragment is four ds_read_b32 from a per-lane base ((ci, kh, kw) of the lane's column) + an immediate (the k-step);
//   * strips are split over blockIdx.z (M x N is tiny, K is millions of pixels): raw slabs, reduced in fixed order by the
//     family's k_splitk_epilogue like the fp32 weight-gradient kernel's;
//   * the next strip's dy pairs and x triples fly under the MFMA phase in registers, are split and written after the barrier.
// ------------------------------------------------------------------------------------------------------------------
// XE (ivln_gemm_desc.split_ok == 2: the caller's promise that every x value is exact in bf16 - the first layer's one-hot map
// features): x is staged as ONE piece (its upper 16 bits; a value with lower bits set poisons the result with NaNs), the three
// products against its lower pieces do not exist, the x region is a third (43 KB of LDS with dy for a 32 x 256 tile: three
// workgroups per CU hide each other's strip loads and barriers) and the kernel fits three waves per SIMD.
template <int TM, int WM, int WN, int W, bool XE = false>
__global__ __launch_bounds__(64 * WM * WN, XE ? 3 : 1) void k_wgrad_bf3(const ivln_gemm_desc p, int nimg, int strips_total, int strips_per_split) {
    constexpr int NTB = 64 * WM * WN, TN = 2, KS = 7, KK = 49;
    constexpr int NPX = XE ? 1 : 3;                                // pieces of x in LDS
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int IMS = W == 8 ? 2 : 1, ROWS = 128 / (W * IMS);   // a strip: IMS images x ROWS rows x W columns = 128 pixels
    constexpr int PR = ROWS + KS - 1, PRT = IMS * PR, PWP = W + KS - 1;
    constexpr int XW = PWP / 2 + 1;                                // pixel-pair words a patch row needs in either copy
    constexpr int XP = (XW | 1);                                   // row pitch in words (odd)
    constexpr int NCIB = (BN + KK - 1) / KK + 1;                   // input channels a column tile can touch
    constexpr int DP = 272;                                        // bytes per (channel, piece) row of dy: 128 bf16 + 16
    constexpr int CHB = PRT * XP * 4, PLB = NCIB * CHB, CPYB = NPX * PLB;
    constexpr int DYB = 3 * BM * DP;                               // dy region in front of the x region
    constexpr int ND = BM * 64, NDI = (ND + NTB - 1) / NTB, NX = NCIB * PRT * XW, NXI = (NX + NTB - 1) / NTB;
    constexpr bool HAS_LITE = TM == 1 && !XE;  // (the 32-channel layer: the one whose x is the one-hot map features)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xS = smem + DYB;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int ci_lo = n0 / KK;
    const int spi = p.Hout / ROWS;  // strips per image (W = 8: one strip = two whole images)
    const int HWo = p.HoWo, HWi = p.Hin * p.Win;
    const float* __restrict__ dy = p.A;
    const float* __restrict__ x = p.B;

    // per-lane operand bases
    int abase[TM], bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = ((wm * TM + tm) * 32 + l31) * DP + half * 16;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = min(n0 + (wn * TN + tn) * 32 + l31, p.N - 1);  // (columns past N compute a duplicate; never stored)
        const int ci = n / KK, rem = n - ci * KK, kh = rem / KS, kw = rem - kh * KS, sft = kw & 1;
        bbase[tn] = sft * CPYB + (ci - ci_lo) * CHB + kh * (XP * 4) + ((kw - sft) >> 1) * 4 + (W >= 16 ? half * 16 : half * (XP * 4));
    }

    float2 dyv[NDI];
    float xv[NXI][3];
    // x items of this thread, derived once: offset of patch column 2j relative to the strip's (image 0, row 0) origin, the
    // LDS word, and what does not depend on the strip - channel and column validity (bits 0-2), image of the pair (bit 3),
    // patch row (bits 4..) - so that a strip costs an add and a row test per item, not three divisions
    int xrel[NXI], xdst[NXI], xinf[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int idx = t + i * NTB;
        const int cl = idx / (PRT * XW), rem = idx - cl * (PRT * XW);
        const int prow = rem / XW, j = rem - prow * XW;
        const int il = prow / PR, pr = prow - il * PR;
        const bool cok = idx < NX && ci_lo + cl < p.Cin;
        int m = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e) m |= (cok && (unsigned)(2 * j + e - p.pad) < (unsigned)p.Win) ? (1 << e) : 0;
        xinf[i] = m | (il << 3) | (pr << 4);
        xrel[i] = (int)((int64_t)il * p.in_img_stride + (int64_t)(ci_lo + cl) * HWi + (pr - p.pad) * p.Win + 2 * j - p.pad);
        xdst[i] = idx < NX ? cl * CHB + prow * (XP * 4) + j * 4 : -1;
    }
    auto strip_origin = [&](int st, int& img0, int& row0) {
        img0 = (st / spi) * IMS;
        row0 = (st - (st / spi) * spi) * ROWS;
    };
    auto load_strip = [&](int st) {
        int img0, row0;
        strip_origin(st, img0, row0);
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < NDI; ++i) {  // dy: (channel, pixel pair)
            const int idx = tt + i * NTB;
            const int co = idx >> 6, px = (idx & 63) * 2;
            const int il = px / (ROWS * W), q = px - il * (ROWS * W);
            const int img = img0 + il;
            const bool ok = idx < ND && m0 + co < p.M && img < nimg;
            dyv[i] = ok ? *reinterpret_cast<const float2*>(dy + ((int64_t)img * p.M + m0 + co) * HWo + row0 * W + q) : make_float2(0.f, 0.f);
        }
        const float* const xo = x + (int64_t)img0 * p.in_img_stride + row0 * p.Win;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {  // x: (channel, patch row, pixel-pair word): columns 2j, 2j+1, 2j+2 of the patch
            const int inf = xinf[i];
            const int hi = row0 + (inf >> 4) - p.pad;
            const bool rok = img0 + ((inf >> 3) & 1) < nimg && (unsigned)hi < (unsigned)p.Hin;
#pragma unroll
            for (int e = 0; e < 3; ++e) xv[i][e] = (rok && ((inf >> e) & 1)) ? xo[xrel[i] + e] : 0.f;
        }
    };
    auto stage_strip = [&]() -> uint32_t {  // returns the OR of the x values' lower pieces (0: the strip's patch is bf16-exact)
        uint32_t nzx = 0;
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < NDI; ++i) {
            const int idx = tt + i * NTB;
            const int co = idx >> 6, pr = idx & 63;
            uint32_t H, M, L;
            split3_pair(dyv[i].x, dyv[i].y, H, M, L);
            unsigned char* d = smem + co * DP + pr * 4;
            if (ND % NTB == 0 || idx < ND) {
                *reinterpret_cast<uint32_t*>(d) = H;
                *reinterpret_cast<uint32_t*>(d + BM * DP) = M;
                *reinterpret_cast<uint32_t*>(d + 2 * BM * DP) = L;
            }
        }
        if constexpr (XE) {
#pragma unroll
            for (int i = 0; i < NXI; ++i) {  // exact values: the upper halves ARE the first pieces, the lower halves must be zero
                const uint32_t u0 = __float_as_uint(xv[i][0]), u1 = __float_as_uint(xv[i][1]), u2 = __float_as_uint(xv[i][2]);
                nzx |= (u0 | u1 | u2) & 0xFFFFu;
                unsigned char* d = xS + xdst[i];
                if (xdst[i] >= 0) {
                    *reinterpret_cast<uint32_t*>(d) = (u0 >> 16) | (u1 & 0xFFFF0000u);         // pair (2j, 2j+1)
                    *reinterpret_cast<uint32_t*>(d + CPYB) = (u1 >> 16) | (u2 & 0xFFFF0000u);  // pair (2j+1, 2j+2)
                }
            }
            return nzx;
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            uint32_t pa0, pa1, pa2, pb0, pb1, pb2;
            split3_pair(xv[i][0], xv[i][1], pa0, pa1, pa2);  // pairs (2j, 2j+1)
            split3_pair(xv[i][1], xv[i][2], pb0, pb1, pb2);  // pairs (2j+1, 2j+2)
            nzx |= pa1 | pa2 | pb1 | pb2;
            unsigned char* d = xS + xdst[i];
            if (xdst[i] >= 0) {
                *reinterpret_cast<uint32_t*>(d) = pa0;
                *reinterpret_cast<uint32_t*>(d + PLB) = pa1;
                *reinterpret_cast<uint32_t*>(d + 2 * PLB) = pa2;
                *reinterpret_cast<uint32_t*>(d + CPYB) = pb0;
                *reinterpret_cast<uint32_t*>(d + CPYB + PLB) = pb1;
                *reinterpret_cast<uint32_t*>(d + CPYB + 2 * PLB) = pb2;
            }
        }
        return nzx;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

    const int s_beg = blockIdx.z * strips_per_split, s_end = min(strips_total, s_beg + strips_per_split);
    unsigned long long tk0 = BF3_T(), t_stage = 0, t_mma = 0;
    (void)tk0, (void)t_stage, (void)t_mma;
#ifdef BF3_TIMING
    const unsigned long long cyc0 = clock64();
#endif
    uint32_t inexact = 0;  // (XE: lower bits seen in x)
    if (s_beg < s_end) load_strip(s_beg);
    for (int st = s_beg; st < s_end; ++st) {
        const unsigned long long ta = BF3_T();
        const uint32_t nzx = stage_strip();
        if constexpr (XE) inexact |= nzx;
        // (a strip whose x patch is bf16-exact - the first layer's one-hot map features - has zero lower pieces: the products
        //  against them are not issued and their fragments not read: a third of the 4-byte LDS reads that bound this kernel)
        const bool lite = HAS_LITE ? __syncthreads_or((int)(nzx != 0)) == 0 : (__syncthreads(), false);
        const unsigned long long tb = BF3_T();
        t_stage += tb - ta;
        if (st + 1 < s_end) load_strip(st + 1);  // in flight under the MFMA phase
        auto ksteps = [&](auto lite_tag) {
            constexpr bool LITE = decltype(lite_tag)::value;
            constexpr int NPL = LITE ? 1 : 3;
            auto read_ab = [&](int ks, bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
                // the k-step's 16 pixels: W >= 16 a run of a row (the lane halves take its two octets), W = 8 two rows of an image
                const int aoff = ks * 32;
                const int boff = W >= 16 ? ((ks * 16) / W) * (XP * 4) + (((ks * 16) % W) >> 1) * 4
                                         : ((ks / 4) * PR + (ks % 4) * 2) * (XP * 4);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[tm][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4i*>(smem + pl * (BM * DP) + abase[tm] + aoff));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(xS + bbase[tn] + pl * PLB + boff);
                        v4i v;
                        v.x = (int)q[0], v.y = (int)q[1], v.z = (int)q[2], v.w = (int)q[3];
                        b[tn][pl] = __builtin_bit_cast(bf16x8, v);
                    }
            };
            bf16x8 aq[2][TM][3], bq[2][TN][3];
            read_ab(0, aq[0], bq[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) read_ab(ks + 1, aq[(ks + 1) & 1], bq[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#define IVLN_BF3_PROD(PA, PB)                                                                                       \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)              \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks & 1][tm][PA], bq[ks & 1][tn][PB], acc[tm][tn], 0, 0, 0)
                if constexpr (!LITE) {
                    IVLN_BF3_PROD(0, 2);
                    IVLN_BF3_PROD(1, 1);
                }
                IVLN_BF3_PROD(2, 0);
                if constexpr (!LITE) IVLN_BF3_PROD(0, 1);
                IVLN_BF3_PROD(1, 0);
                IVLN_BF3_PROD(0, 0);
#undef IVLN_BF3_PROD
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (XE) {
            ksteps(std::true_type{});
        } else if constexpr (HAS_LITE) {
            if (lite) ksteps(std::true_type{});
            else ksteps(std::false_type{});
        } else {
            ksteps(std::false_type{});
        }
        __syncthreads();
        t_mma += BF3_T() - tb;
    }
    // XE: a broken promise must not pass for a gradient - every output of a workgroup that saw an inexact x value becomes a NaN
    const bool poison = XE ? __syncthreads_or((int)(inexact != 0)) != 0 : false;
#ifdef BF3_TIMING
    if (threadIdx.x == 0) {
        const int b = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 8191;
        g_bf3_stamp[b * 8 + 0] = wall_clock64() - tk0;
        g_bf3_stamp[b * 8 + 1] = t_stage;
        g_bf3_stamp[b * 8 + 2] = t_mma;
        g_bf3_stamp[b * 8 + 3] = clock64() - cyc0;
    }
#endif

    // acc[r] -> channel (r&3) + 8*(r>>2) + 4*half, column l31 of the sub-tile: 32 consecutive columns per store instruction
    // always a raw slab, even a single one (the reduction launch applies the epilogue): ivln_gemm_f32's epilogue_store inlined
    // 64 times made the nest too large to unroll, and the accumulators, then indexed dynamically, lived in scratch
    float* const slab = p.ws + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TN + tn) * 32 + l31;
                const int m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M && n < p.N) slab[(int64_t)m * p.N + n] = poison ? __builtin_nanf("") : acc[tm][tn][r];
            }
}

template <int TM, int WM, int WN, int W, bool XE = false>
int launch_wgrad_bf3(const ivln_gemm_desc& d, hipStream_t s, int nimg, int strips, int sps) {
    constexpr int BM = 32 * TM * WM, BN = 64 * WN, IMS = W == 8 ? 2 : 1, ROWS = 128 / (W * IMS);
    constexpr int PRT = IMS * (ROWS + 6), XP = ((W + 6) / 2 + 1) | 1, NCIB = (BN + 48) / 49 + 1;
    constexpr size_t lds = (size_t)3 * BM * 272 + (size_t)(XE ? 2 : 6) * NCIB * PRT * XP * 4;
    static_assert(lds <= 160 * 1024, "strip does not fit");
    auto kern = k_wgrad_bf3<TM, WM, WN, W, XE>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IVLN_E_HIP;
        attr_done = true;
    }
    dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.splits);
    IVLN_LAUNCH_FAMILY_NAMED("k_wgrad_bf3", kern, grid, dim3(64 * WM * WN), lds, s, d, nimg, strips, sps);
    return IVLN_OK;
}

template <int TM, int WM, int WN>
int launch_wgrad_bf3_w(const ivln_gemm_desc& d, hipStream_t s, int nimg, int strips, int sps) {
    switch (d.Wout) {
        case 64: return launch_wgrad_bf3<TM, WM, WN, 64>(d, s, nimg, strips, sps);
        case 32: return launch_wgrad_bf3<TM, WM, WN, 32>(d, s, nimg, strips, sps);
        case 16: return launch_wgrad_bf3<TM, WM, WN, 16>(d, s, nimg, strips, sps);
        default: return launch_wgrad_bf3<TM, WM, WN, 8>(d, s, nimg, strips, sps);
    }
}

}  // namespace

extern "C" int64_t ivln_conv_split_words(int M, int Cin, int KS) {
    if ((KS != 1 && KS != 2 && KS != 3 && KS != 7) || M <= 0 || Cin <= 0) return 0;
    return (int64_t)((M + 31) / 32) * ((Cin + CB - 1) / CB) * bf3_taps_padded(KS) * (3 * 1024 / 4);
}

extern "C" int ivln_conv_split_weights_f32(const float* W, int M, int Cin, int KS, void* out, void* stream) {
    const int64_t words = ivln_conv_split_words(M, Cin, KS);
    if (!W || !out || words <= 0) return IVLN_E_INVALID;
    const int64_t total = words * 2;
    hipLaunchKernelGGL(k_conv_bf3_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, M, Cin, KS,
                       (uint16_t*)out, total);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

namespace {
// ------------------------------------------------------------------------------------------------------------------
// RedNet's stems (rednet.py:201-210, 190-199: conv1 3 -> 64 and conv1_d 1 -> 64; 7x7, stride 2, pad 3) on the same arithmetic.
// Three or one input channels leave the 16-channel chunks of the kernels above 81 / 94 % empty, so K is laid out the other
// way round: a K step of 16 = TWO kernel rows (channel c, row kh) x 8 columns (kw = 0 .. 6 and a zero) - 11 steps for the 21
// rows of the RGB stem, 4 for the 7 of the depth stem (one zero row each).  NO LDS and no barrier anywhere:
//   * a wave owns 32 output channels x 128 consecutive pixels of ONE output row; as in the 1x1 kernels tile e holds pixel
//     4 l31 + e in column l31, so register r of the four accumulators is a float4 of the output row;
//   * lane (l31, half) needs, for its kernel row of the step, input columns 2 (4 l31 + e) - 3 + kw: the 14 consecutive floats
//     8 l31 - 3 .. 8 l31 + 10 of input row 2 ho - 3 + kh, read as FOUR aligned 16-byte loads (8 l31 - 4 ..); rows and column
//     groups outside the image are out-of-range buffer offsets, i.e. zeros - the padding costs nothing;
//   * element pairs (2 t + 1, 2 t + 2), t = 0 .. 6, are split into the three bf16 pieces once (v_cvt_pk_bf16_f32 leaves the
//     word the fragment wants); tile e's B fragment is pairs e .. e + 3 - every register index static;
//   * weights: the image of ivln_conv_stem_split_weights_f32 - [32-channel tile][step][piece][lane] x 8 bf16, lane (l31, half)
//     holding W[32 mt + l31][row 2 s + half][kw 0 .. 6], 0 - global -> registers two steps ahead;
//   * epilogue: the 1x1 wave-tile kernel's (scale / shift, residual in front of or BEHIND the ReLU - the depth stem's output
//     added to the RGB stem's, rednet.py:196 -, 16-byte stores).
// ------------------------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256, 2) void k_conv7s2_bf3(const ivln_gemm_desc p, const unsigned char* a_split, int segs, int items) {
    constexpr int Q = CIN * 7, S = (Q + 1) / 2, TN = 4;
    constexpr unsigned OOB = 0x80000000u;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int item = bid.x * 4 + wave;  // (channel tile, row segment, output row, image), channel tile fastest: a workgroup shares its input rows
    if (item >= items) return;
    const int mtiles = (p.M + 31) / 32;
    const int mt = item % mtiles;
    int rest = item / mtiles;
    const int seg = rest % segs;
    rest /= segs;
    const int ho = rest % p.Hout, img = rest / p.Hout;
    const int m0 = mt * 32;
    const int H = p.Hin, W = p.Win, HW = p.HoWo;
    const int b = 256 * seg + 8 * l31;  // input column of the lane's first pixel (2 * (128 seg + 4 l31))
    const bool in0 = b >= 4, in3 = b + 8 < W;
    const __amdgpu_buffer_rsrc_t rB = bf3_rsrc(p.B);
    const __amdgpu_buffer_rsrc_t rA = bf3_rsrc(a_split + (int64_t)mt * S * 3072);
    const int64_t img_off = (int64_t)img * p.in_img_stride + b - 4;
    auto load_x = [&](int s, v4i (&xb)[4]) {
        const int q = 2 * s + half;  // this half's kernel row (channel c, row kh) of step s
        const int c = q / 7, kh = q - 7 * c;
        const int hi = 2 * ho - 3 + kh;
        const bool ok = q < Q && (unsigned)hi < (unsigned)H;
        const unsigned off = (unsigned)((img_off + ((int64_t)c * H + hi) * W) * 4);
        const unsigned o12 = ok ? off : OOB, o0 = ok && in0 ? off : OOB, o3 = ok && in3 ? off : OOB;
        xb[0] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)o0, 0, 0);
        xb[1] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)(o12 + 16u), 0, 0);
        xb[2] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)(o12 + 32u), 0, 0);
        xb[3] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)(o3 + 48u), 0, 0);
    };
    auto load_a = [&](int s, v4i (&ab)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ab[pl] = __builtin_amdgcn_raw_buffer_load_b128(rA, (s * 3 + pl) * 1024 + lane * 16, 0, 0);
    };
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;
    {
        v4i xb[3][4], ab[3][3];
        load_a(0, ab[0]);
        load_x(0, xb[0]);
        if (S > 1) load_a(1, ab[1]), load_x(1, xb[1]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s + 2 < S) load_a(s + 2, ab[(s + 2) % 3]), load_x(s + 2, xb[(s + 2) % 3]);  // two steps ahead
            uint32_t Hh[7], Mm[7], Ll[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i0 = 2 * u + 1, i1 = 2 * u + 2;
                split3_pair(__int_as_float(xb[s % 3][i0 >> 2][i0 & 3]), __int_as_float(xb[s % 3][i1 >> 2][i1 & 3]), Hh[u], Mm[u], Ll[u]);
            }
            v4i bq[TN][3];
#pragma unroll
            for (int e = 0; e < TN; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[e][0][i] = (int)Hh[e + i], bq[e][1][i] = (int)Mm[e + i], bq[e][2][i] = (int)Ll[e + i];
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, ab[s % 3][pl]);
#define IVLN_BF3_PROD(PA, PB)                            \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)    \
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], __builtin_bit_cast(bf16x8, bq[tn][PB]), acc[tn], 0, 0, 0)
            IVLN_BF3_PROD(0, 2);
            IVLN_BF3_PROD(1, 1);
            IVLN_BF3_PROD(2, 0);
            IVLN_BF3_PROD(0, 1);
            IVLN_BF3_PROD(1, 0);
            IVLN_BF3_PROD(0, 0);
#undef IVLN_BF3_PROD
        }
    }
    // ---- epilogue (the wave-tile 1x1 kernel's): register r of the four tiles = channel (r & 3) + 8 (r >> 2) + 4 half at pixels
    // 4 l31 .. + 3; operands requested first, straight-line code between the stores (BF3_STORE_GUARD) ----
    const int pp = ho * p.Wout + 128 * seg + 4 * l31;
    const int mrem = p.M - m0 - 4 * half;
    const int me0 = m0 + 4 * half;
    const unsigned off0 = (unsigned)(((img * p.Ctot + m0 + 4 * half) * HW + pp) * 4);
    const __amdgpu_buffer_rsrc_t rR = bf3_rsrc(p.residual), rS = bf3_rsrc(p.scale), rH = bf3_rsrc(p.shift), rD = bf3_rsrc(p.D);
    const bool has_res = p.residual != nullptr, has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
    const float relu_lo = p.relu ? 0.f : -__builtin_huge_valf();
    const bool res_post = p.residual_after_relu != 0;
    v4i rres[16];
    float esc[16], esh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cs = (r & 3) + 8 * (r >> 2);
        const bool ok = cs < mrem;
        esc[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, ok && has_sc ? me0 * 4 : (int)OOB, cs * 4, 0));
        esh[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rH, ok && has_sh ? me0 * 4 : (int)OOB, cs * 4, 0));
        rres[r] = __builtin_amdgcn_raw_buffer_load_b128(rR, ok && has_res ? (int)off0 : (int)OOB, cs * HW * 4, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cs = (r & 3) + 8 * (r >> 2);
        float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        const float sc = has_sc ? esc[r] : 1.f, sh = esh[r];
        v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
        const float r0 = __int_as_float(rres[r][0]), r1 = __int_as_float(rres[r][1]), r2 = __int_as_float(rres[r][2]), r3 = __int_as_float(rres[r][3]);
        v.x += res_post ? 0.f : r0, v.y += res_post ? 0.f : r1, v.z += res_post ? 0.f : r2, v.w += res_post ? 0.f : r3;
        const float lo = relu_lo;
        v.x = fmaxf(v.x, lo), v.y = fmaxf(v.y, lo), v.z = fmaxf(v.z, lo), v.w = fmaxf(v.w, lo);
        v.x += res_post ? r0 : 0.f, v.y += res_post ? r1 : 0.f, v.z += res_post ? r2 : 0.f, v.w += res_post ? r3 : 0.f;
        v4i o;
        o[0] = __float_as_int(v.x), o[1] = __float_as_int(v.y), o[2] = __float_as_int(v.z), o[3] = __float_as_int(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(o, rD, cs < mrem ? (int)off0 : (int)OOB, cs * HW * 4, 0);
        BF3_STORE_GUARD();
    }
}

// [32-channel tile][step][piece][lane] x 8 bf16 of the stem kernel above: lane (l31, half), element j = W[32 mt + l31][c][kh][j] with
// (c, kh) = kernel row 2 step + half, zero for j = 7 and rows past the last
__global__ __launch_bounds__(256) void k_conv7s2_pack(const float* __restrict__ W, int M, int Cin, uint16_t* __restrict__ out, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int S = (Cin * 7 + 1) / 2;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t q = idx >> 9;
    const int pl = (int)(q % 3);
    q /= 3;
    const int st = (int)(q % S);
    const int mt = (int)(q / S);
    const int m = mt * 32 + (lane & 31), row = 2 * st + (lane >> 5);
    const float v = (m < M && row < Cin * 7 && j < 7) ? W[((int64_t)m * Cin * 7 + row) * 7 + j] : 0.f;
    uint32_t h, mm, l;
    split3(v, h, mm, l);
    out[idx] = (uint16_t)(pl == 0 ? h : (pl == 1 ? mm : l));
}

}  // namespace

extern "C" int64_t ivln_conv_stem_split_words(int M, int Cin) {
    if (M <= 0 || (Cin != 1 && Cin != 3)) return 0;
    return (int64_t)((M + 31) / 32) * ((Cin * 7 + 1) / 2) * (3 * 1024 / 4);
}

extern "C" int ivln_conv_stem_split_weights_f32(const float* W, int M, int Cin, void* out, void* stream) {
    const int64_t words = ivln_conv_stem_split_words(M, Cin);
    if (!W || !out || words <= 0) return IVLN_E_INVALID;
    const int64_t total = words * 2;
    hipLaunchKernelGGL(k_conv7s2_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, M, Cin, (uint16_t*)out, total);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

namespace {
// The stems' launcher: 7x7, stride 2, pad 3, one or three input channels, output rows of whole 128-pixel segments
int bf3_stem_launch(ivln_gemm_desc& d, hipStream_t s) {
    if ((d.Cin != 1 && d.Cin != 3) || d.pad != 3 || d.dil != 1 || d.Hin != 2 * d.Hout || d.Win != 2 * d.Wout || d.Wout % 128 != 0 ||
        d.K != d.Cin * 49 || d.HoWo != d.Hout * d.Wout || d.N % d.HoWo != 0 || d.dmode != DMODE_NCHW || d.amode != AMODE_MK)
        return IVLN_E_UNSUPPORTED;
    if (d.grp_imgs > 0 || d.accumulate || d.defer_epilogue || d.splits > 1 || d.stat_partials || d.img_run_flags || d.fuse_A_split)
        return IVLN_E_UNSUPPORTED;
    if ((d.in_img_stride & 3) || ((((uintptr_t)d.B) | ((uintptr_t)d.D) | ((uintptr_t)d.residual)) & 15)) return IVLN_E_UNSUPPORTED;
    const int64_t nimg = d.N / d.HoWo;
    if (nimg * d.in_img_stride * 4 >= (int64_t)1 << 31 || nimg * d.Ctot * d.HoWo * 4 >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // byte offsets of the buffer accesses
    const int segs = d.Wout / 128;
    const int64_t items = nimg * d.Hout * segs * ((d.M + 31) / 32);
    if (items >= (int64_t)1 << 30) return IVLN_E_UNSUPPORTED;
    d.splits = 1;
    const dim3 grid((unsigned)((items + 3) / 4));
    const unsigned char* a = (const unsigned char*)d.A_split;
    if (d.Cin == 3) IVLN_LAUNCH_FAMILY(k_conv7s2_bf3<3>, grid, dim3(256), 0, s, d, a, segs, (int)items);
    else IVLN_LAUNCH_FAMILY(k_conv7s2_bf3<1>, grid, dim3(256), 0, s, d, a, segs, (int)items);
    return IVLN_OK;
}
}  // namespace

// Eligibility + tile choice.  IVLN_E_UNSUPPORTED sends the caller to the fp32 MFMA kernels.
// What went through this kernel since the last reset: algorithmic FLOPs (2 M N K of the convs it took) and launches -
// bench.py prices them against the bf16 peak / 6 instead of the fp32 MFMA peak.  Host-side tally, not thread-safe.
static double g_bf3_flops = 0.0;
static long long g_bf3_launches = 0;
static long long g_bf3_kind[4] = {0, 0, 0, 0};  // launches by form: tiled | 3x3 K-split over waves | 1x1 K-split over waves | 1x1 wave tiles
extern "C" int ivln_conv_split_counters(double* flops, long long* launches, int reset) {
    if (flops) *flops = g_bf3_flops;
    if (launches) *launches = g_bf3_launches;
    if (reset) g_bf3_flops = 0.0, g_bf3_launches = 0;
    return IVLN_OK;
}
/* launches of the split-bf16 conv kernels by form since the last reset: [0] tiled (k_conv_bf3), [1] 3x3 with K split over the
 * waves (k_conv_bf3_ks), [2] 1x1 with K split over the waves, [3] 1x1 wave tiles (k_conv1x1_bf3_ks) */
extern "C" int ivln_conv_split_kinds(long long* out4, int reset) {
    if (out4) for (int i = 0; i < 4; ++i) out4[i] = g_bf3_kind[i];
    if (reset) for (int i = 0; i < 4; ++i) g_bf3_kind[i] = 0;
    return IVLN_OK;
}

#ifdef BF3_TIMING
extern "C" int ivln_conv_bf3_stamps(unsigned long long* host, int n) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bf3_stamp), sizeof(unsigned long long) * n) != hipSuccess) return -1;
    void* dp = nullptr;  // (cleared for the next launch: a smaller grid leaves no stale entries behind)
    if (hipGetSymbolAddress(&dp, HIP_SYMBOL(g_bf3_stamp)) != hipSuccess) return -1;
    return hipMemset(dp, 0, sizeof(g_bf3_stamp)) == hipSuccess ? 0 : -1;
}
#endif

int ivln_conv_bf3_launch(ivln_gemm_desc& d, hipStream_t s, bool force) {
    constexpr bool disabled = false;  // A/B switch
    if (!d.A_split || (disabled && !force)) return IVLN_E_UNSUPPORTED;
    const int KS = d.bmode == BMODE_CONV1X1 ? 1 : conv_ks(d.bmode);
    if ((KS != 1 && KS != 2 && KS != 3 && KS != 7) || d.amode != AMODE_MK || d.dil != 1) return IVLN_E_UNSUPPORTED;
    if (KS == 7 && d.stride == 2) {  // RedNet's stems: A_split is the image of ivln_conv_stem_split_weights_f32 (the caller's contract)
        const int rc = bf3_stem_launch(d, s);
        if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches, ++g_bf3_kind[0];
        if (rc == IVLN_OK && d.stat_tiles) *d.stat_tiles = 0;
        return rc;
    }
    // (the 2 x 2 window only as the stacked transposed-conv classes; a 1x1 conv may store them too: the 2 x 2 stride-2 upsampling convs)
    const bool up1 = KS == 1 && d.dmode == DMODE_NCHW_UP2X4;
    if (!up1 && d.dmode != (KS == 2 ? DMODE_NCHW_UP2X4 : DMODE_NCHW)) return IVLN_E_UNSUPPORTED;
    if (up1) {  // stacked one-tap classes: the register-built stride-1 1x1 kernels with their 2 x 2-block epilogue, or nothing
        static const bool convt_off1 = getenv("IVLN_BF3_CONVT") && getenv("IVLN_BF3_CONVT")[0] == '0';
        if ((convt_off1 && !force) || (d.M & 3) || d.Ctot * 4 != d.M || d.grp_imgs > 0 || d.accumulate || d.residual_after_relu || d.defer_epilogue ||
            d.splits > 1 || d.stat_partials || d.img_run_flags || d.fuse_A_split || (d.Wout & 3) || d.stride != 1 || d.pad != 0 || d.Hout != d.Hin ||
            d.Wout != d.Win || d.K != d.Cin || d.HoWo != d.Hout * d.Wout || d.N % d.HoWo != 0)
            return IVLN_E_UNSUPPORTED;
        const int ovu = d.tile_override;
        const int rc = bf3_1x1_ks_launch(d, s, (force || ovu == 11 || ovu == 12 || ovu == 13) ? 1 : 0, ovu == 12 ? 0 : (ovu == 13 ? 1 : -1));
        if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches, ++g_bf3_kind[d.Cin / CB <= 16 ? 3 : 2];
        return rc;
    }
    // algorithmic FLOPs of a launch: the stacked classes multiply their common window's zero padding too (ivln_gemm_desc.real_taps)
    const double flops_of = 2.0 * d.M * (double)d.N * d.K * (KS == 2 && d.real_taps > 0 ? d.real_taps / 16.0 : 1.0);
    if (KS == 2) {
        if (d.stride != 1 || d.pad != 0 || d.Hout != d.Hin || d.Wout != d.Win || (d.M & 3) || d.grp_imgs > 0 || d.stat_partials || d.defer_epilogue ||
            d.splits > 1 || d.fuse_A_split || d.residual_after_relu || d.img_run_flags || d.K != d.Cin * 4 || d.HoWo != d.Hout * d.Wout ||
            d.N % d.HoWo != 0 || (d.Wout & 3) || d.Wout < 8 || (((uintptr_t)d.D | (uintptr_t)d.residual) & 15) || d.Ctot * 4 != d.M)
            return IVLN_E_UNSUPPORTED;
        const int nimg2 = d.N / d.HoWo;
        if ((int64_t)nimg2 * d.in_img_stride >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;
        static const bool convt_off = getenv("IVLN_BF3_CONVT") && getenv("IVLN_BF3_CONVT")[0] == '0';  // A/B switch
        if (convt_off && !force) return IVLN_E_UNSUPPORTED;
        const int ov2 = d.tile_override;
        if (ov2 < 20) {  // few pixels: K split over the waves
            const int rc = bf3_ks_launch(d, s, nimg2, (ov2 == 10 || ov2 == 15) ? 1 : 0);
            if (rc != IVLN_E_UNSUPPORTED || ov2 == 10 || ov2 == 15) {
                if (rc == IVLN_OK) g_bf3_flops += flops_of, ++g_bf3_launches, ++g_bf3_kind[1];
                return rc;
            }
        }
        const int CUS2 = ivln_cu_count();
        auto blocks2 = [&](int cfg) {
            const Bf3Px t = bf3_px(kBf3BN[cfg], d.Wout);
            return (int64_t)((d.Wout + t.ptw - 1) / t.ptw) * ((d.Hout + t.pth - 1) / t.pth) * ((nimg2 + t.imgs - 1) / t.imgs) *
                   ((d.M + kBf3BM[cfg] - 1) / kBf3BM[cfg]);
        };
        // the widest of the four tiles that gives every CU a workgroup (the classes cannot split K: no slabs in this store form)
        int cfg2 = blocks2(3) >= CUS2 ? 3 : (blocks2(2) >= CUS2 ? 2 : (blocks2(5) >= CUS2 ? 5 : 4));
        if (ov2 >= 20 && ov2 < 20 + kBf3Cfgs) cfg2 = ov2 - 20;
        if (!force && blocks2(cfg2) < CUS2 / 2) return IVLN_E_UNSUPPORTED;
        d.splits = 1;
        const int rc = launch_bf3_ks<2>(d, s, (const unsigned char*)d.A_split, d.a_split_grp_stride * 4, nimg2, cfg2, (d.Cin + CB - 1) / CB);
        if (rc == IVLN_OK) g_bf3_flops += flops_of, ++g_bf3_launches, ++g_bf3_kind[0];
        return rc;
    }
    // deep-K 1x1 convs: K split over the waves of a workgroup, fragments built in registers (k_conv1x1_bf3_ks);
    // IVLN_BF3_1X1_KS=0 | 1 = never | wherever eligible, tile_override 11 insists
    static const int ks1_env = getenv("IVLN_BF3_1X1_KS") ? atoi(getenv("IVLN_BF3_1X1_KS")) : -1;
    // (tuning, tools/conv_cfg_sweep.py: tile_override 12 / 13 insist on the K-split / wave-tile form, 14 / 15 on 32- / 64-pixel tiles
    //  of the 3x3 K-split kernel, 20 + c on tile c of the tiled kernel)
    const int ov = d.tile_override;
    const bool post = d.residual_after_relu != 0;  // (only the stride-1 1x1 kernels below know this epilogue form: they are insisted on)
    const bool ins1 = ov == 11 || ov == 12 || ov == 13 || post, ins3 = ov == 10 || ov == 14 || ov == 15;
    if (KS == 1 && ov < 20 && (ks1_env != 0 || ins1) && d.splits <= 1 && !d.defer_epilogue && d.HoWo == d.Hout * d.Wout &&
        d.K == d.Cin && d.N % d.HoWo == 0 && d.Hout == d.Hin && d.Wout == d.Win) {
        const int rc = bf3_1x1_ks_launch(d, s, (ks1_env == 1 || ins1) ? 1 : 0, ov == 12 ? 0 : (ov == 13 ? 1 : -1));
        if (rc != IVLN_E_UNSUPPORTED || ins1) {
            if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches, ++g_bf3_kind[ov == 13 ? 3 : (ov == 12 ? 2 : (d.Cin / CB <= 16 ? 3 : 2))];
            if (rc == IVLN_OK && d.stat_tiles) *d.stat_tiles = 0;
            return rc;
        }
    }
    if (ins1 || (ins3 && KS != 3)) return IVLN_E_UNSUPPORTED;
    if (KS == 1) {  // 1x1, stride 1 or 2, no padding
        if ((d.stride != 1 && d.stride != 2) || d.pad != 0 || d.Hout != (d.Hin - 1) / d.stride + 1 || d.Wout != (d.Win - 1) / d.stride + 1 || d.M < 64)
            return IVLN_E_UNSUPPORTED;
    } else if (KS == 3 && d.stride == 2) {  // stride-2 3x3 (pad 1, even input): the tiled kernel's phase-plane staging
        static const bool s2_off = getenv("IVLN_BF3_S2") && getenv("IVLN_BF3_S2")[0] == '0';  // A/B switch
        if ((s2_off && !force) || d.pad != 1 || d.Hin != 2 * d.Hout || d.Win != 2 * d.Wout || d.Cin % CB != 0 || d.fuse_A_split || d.stat_partials ||
            d.img_run_flags)
            return IVLN_E_UNSUPPORTED;
    } else if (d.stride != 1 || d.pad != KS / 2 || d.Hout != d.Hin || d.Wout != d.Win) {
        return IVLN_E_UNSUPPORTED;
    }
    const bool s2 = KS == 3 && d.stride == 2;
    if (d.K != d.Cin * KS * KS || d.HoWo != d.Hout * d.Wout || d.N % d.HoWo != 0) return IVLN_E_UNSUPPORTED;
    if (d.defer_epilogue || d.splits > 1 || (d.Wout & 3) || d.Wout < 8 || (((uintptr_t)d.D | (uintptr_t)d.residual | (uintptr_t)d.ws) & 15))
        return IVLN_E_UNSUPPORTED;
    const int nimg = d.N / d.HoWo;
    if ((int64_t)nimg * d.in_img_stride >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // 32-bit patch offsets
    if (d.fuse_A_split) {  // the bottleneck tail: this 3x3 conv + the 1x1 expansion behind it in one launch (k_conv_bf3<..., FUSE>)
        static const bool fuse_off = getenv("IVLN_BF3_FUSE") && getenv("IVLN_BF3_FUSE")[0] == '0';  // A/B switch
        if (fuse_off || KS != 3 || (d.M != 64 && d.M != 128) || d.Cin % CB != 0 || d.Wout % 32 != 0 || d.Hout % 4 != 0 || d.fuse_M <= 0 ||
            d.fuse_M % 32 != 0 || d.stat_partials || d.accumulate || !d.relu || d.img_run_flags)
            return IVLN_E_UNSUPPORTED;
        if ((int64_t)nimg * d.Ctot * d.HoWo * 4 >= (int64_t)1 << 31 || (d.grp_imgs > 0 && nimg % d.grp_imgs != 0)) return IVLN_E_UNSUPPORTED;
        d.splits = 1;
        const unsigned char* a = (const unsigned char*)d.A_split;
        const int64_t gb = d.a_split_grp_stride * 4;
        // tiles: 64 or 128 channels x 128 pixels (4 x 32); 128 channels x 64 pixels (2 x 32, four waves of 32 channels) where the
        // 128-pixel grid would leave CUs without a workgroup (layer 2 at 8 + 8 images: 128 workgroups); IVLN_BF3_FUSE_PX pins (tuning)
        constexpr int px_env = 0;
        const int64_t wg128 = (int64_t)nimg * (d.Wout / 32) * (d.Hout / 4);
        const bool px64 = d.M == 128 && d.Hout % 2 == 0 && (px_env ? px_env == 64 : wg128 < ivln_cu_count());
        const int rc = d.M == 64 ? launch_bf3_fused<1, 2, 2, 3>(d, s, a, gb, nimg)
                       : (px64 ? launch_bf3_fused<1, 4, 1, 3>(d, s, a, gb, nimg) : launch_bf3_fused<2, 2, 2, 3>(d, s, a, gb, nimg));
        if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K + 2.0 * d.fuse_M * (double)d.N * d.M, ++g_bf3_launches, ++g_bf3_kind[0];
        if (rc == IVLN_OK && d.stat_tiles) *d.stat_tiles = 0;
        return rc;
    }
    // pixel-starved deep 3x3 convs: K split over the waves of a workgroup, no slabs (k_conv_bf3_ks); IVLN_BF3_KS=0 | 1 = never | wherever eligible
    static const int ks_env = getenv("IVLN_BF3_KS") ? atoi(getenv("IVLN_BF3_KS")) : -1;
    if (KS == 3 && !s2 && ov < 20 && (ks_env != 0 || ins3) && d.splits <= 1) {
        const int rc = bf3_ks_launch(d, s, nimg, (ks_env == 1 || ins3) ? 1 : 0, ov == 14 ? 1 : (ov == 15 ? 2 : 0));
        if (rc != IVLN_E_UNSUPPORTED || ins3) {
            if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches, ++g_bf3_kind[1];
            if (rc == IVLN_OK && d.stat_tiles) *d.stat_tiles = 0;
            return rc;
        }
    }
    const int nch = ((d.Cin + CB - 1) / CB + bf3_stage_chunks(KS) - 1) / bf3_stage_chunks(KS);  // stages: what blockIdx.z can split
    auto tiles_of = [&](int cfg) {
        const Bf3Px t = bf3_px(kBf3BN[cfg], d.Wout);
        return (int64_t)((d.Wout + t.ptw - 1) / t.ptw) * ((d.Hout + t.pth - 1) / t.pth) * ((nimg + t.imgs - 1) / t.imgs);
    };
    auto blocks_of = [&](int cfg) { return tiles_of(cfg) * ((d.M + kBf3BM[cfg] - 1) / kBf3BM[cfg]); };
    // fills the chip: at most a quarter of the last round of 256 x slots workgroups empty
    const int CUS = ivln_cu_count();  // (256 on MI355X; a partition mode or another part gets its own count)
    auto fills = [&](int64_t nb, int slots) {
        const int64_t round = (int64_t)CUS * slots, rounds = (nb + round - 1) / round;
        return nb * 4 >= rounds * round * 3;
    };
    constexpr int cfg_env = -1;  // tuning
    constexpr int split_env = 0;
    // the widest tile that fills the chip on its own; else the 4-wave tiles (two or three workgroups per CU) with the channel
    // chunks split over blockIdx.z (raw slabs reduced by k_splitk_epilogue, like the fp32 kernels)
    int cfg = -1, splits = 1;
    const bool big_ok = !(KS == 7 && d.Wout <= 8) && !s2;  // (7x7 on 8x8 maps: eight images of 14 x 14 patch pixels do not fit; stride 2: 128- / 256-pixel tiles only)
    // (4-wave tiles, measured on RedNet's 3x3 shapes at 8 + 8 stacked images against the fp32 kernels: 128 x 128 wins from
    //  2 M outputs - 57 vs 64 us on 128 x 16384, 50 vs 67 on 256 x 4096 and 512 x 1024 -, 64 x 128 below - 41 vs 46 on
    //  128 x 8192, 37.5 vs 44 on 256 x 2048 and 512 x 512; 64-channel convs that need them lose - 42 vs 38 us on 64 x 32768)
    constexpr bool nosplit4 = true;  // A/B: =0 restores the split 64 x 128 tiles
    constexpr int cfg32_env = -1;  // tuning: 0 | 6
    if (d.M <= 32) cfg = cfg32_env >= 0 ? cfg32_env : 6;
    else if (d.M <= 64)
        cfg = (big_ok && KS != 1 && fills(blocks_of(1), 1)) ? 1
              : (fills(blocks_of(2), 1) ? 2 : ((force || (nosplit4 && blocks_of(4) >= CUS)) ? 4 : -1));
    // (64 x 32768, the decoder's 64-channel convs at 64 x 64: the 64 x 128 tile UNSPLIT - one workgroup per CU - 24.9 us against the
    //  fp32 kernel's 37.2 and 35.8 with its channel chunks split three ways + the reduction launch: tools/conv_cfg_sweep.py)
    // (round 5, aligned-group staging: the 64 x 128 tile now beats 128 x 128 wherever the latter leaves CUs without a workgroup -
    //  128 x 16384: 42.7 vs 50.9 us, 256 x 4096: 43.2 vs 45.3)
    else cfg = fills(blocks_of(3), 1) ? 3 : (fills(blocks_of(2), 1) ? 2 : (((int64_t)d.M * d.N >= (1 << 21) && blocks_of(5) >= CUS) ? 5 : 4));
    if (cfg < 0) return IVLN_E_UNSUPPORTED;
    if (cfg_env >= 0 && cfg_env < kBf3Cfgs && !(cfg_env == 0 && d.M > 32)) cfg = cfg_env;
    if (d.tile_override >= 20 && d.tile_override < 20 + kBf3Cfgs) cfg = d.tile_override - 20;  // (tuning: tools/conv_cfg_sweep.py pins a tile)
    if ((cfg == 0 && !big_ok) || (KS == 1 && (cfg <= 1 || cfg == 6))) return IVLN_E_UNSUPPORTED;
    const int64_t nb = blocks_of(cfg);
    // a pixel tile is rows x columns of ONE image (or whole small images): a Conv1d-shaped input (H = 1: the policy's k / v
    // projections, 384 x 8192 x 192 in an update) would fill 1 row of 16 - 427 us for 1.2 GFLOP, found in round 5 after the
    // 1x1 eligibility had been widened.  Tilings that use less than 15 % of their pixels go back to the fp32 GEMMs (at 21 % -
    // 256 x 40960 x 256 as rows of 80 - this kernel still wins: 66.7 against 80.3 us).
    if (!force && (double)d.N < 0.15 * (double)tiles_of(cfg) * kBf3BN[cfg]) return IVLN_E_UNSUPPORTED;
    if (cfg >= 4) {
        const int slots = cfg == 4 ? 3 : 2;
        const bool may_split = d.splits == 0 && d.ws && nch >= 2 && !d.stat_partials;
        if (may_split && nb < (int64_t)CUS * slots && !(cfg == 4 && nosplit4 && nb >= CUS)) {  // (128 x 16384: 45.0 unsplit, 53.0 split two ways)
            const int64_t want = (int64_t)CUS * slots;
            splits = (int)((want + nb - 1) / nb);
            if (splits > nch) splits = nch;
            if (splits > 16) splits = 16;
            const int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
        if (split_env > 0 && may_split) {
            splits = split_env > nch ? nch : split_env;
            const int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);  // (the slabs have to fit the workspace whatever the tuning knob says)
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
        if (!force && nb * splits < CUS * 3 / 4) return IVLN_E_UNSUPPORTED;  // pixel- and channel-starved: the implicit GEMM splits K deeper
    } else if (!force && !fills(nb, cfg == 6 ? 2 : 1)) {
        return IVLN_E_UNSUPPORTED;
    }
    const int cps = (nch + splits - 1) / splits;
    splits = (nch + cps - 1) / cps;
    const Bf3Px t = bf3_px(kBf3BN[cfg], d.Wout);
    if (d.grp_imgs > 0 && (d.grp_imgs % t.imgs != 0 || nimg % d.grp_imgs != 0)) return IVLN_E_UNSUPPORTED;  // a tile's images share one weight set
    d.splits = splits;
    if (splits > 1) d.stat_partials = nullptr;
    const int64_t tiles = tiles_of(cfg);
    const int BN = kBf3BN[cfg];
    const int64_t gb = d.a_split_grp_stride * 4;
    const int rc = KS == 7   ? launch_bf3_ks<7>(d, s, (const unsigned char*)d.A_split, gb, nimg, cfg, cps)
                   : s2      ? launch_bf3_s2(d, s, (const unsigned char*)d.A_split, gb, nimg, cfg, cps)
                   : KS == 3 ? launch_bf3_ks<3>(d, s, (const unsigned char*)d.A_split, gb, nimg, cfg, cps)
                             : launch_bf3_ks<1>(d, s, (const unsigned char*)d.A_split, gb, nimg, cfg, cps);
    if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches, ++g_bf3_kind[0];
    if (rc == IVLN_OK && d.stat_tiles) *d.stat_tiles = d.stat_partials ? (int)(tiles * (BN / 128)) : 0;
    return rc;
}

// The 7x7 weight gradient on the split-bf16 arithmetic (k_wgrad_bf3).  IVLN_E_UNSUPPORTED -> the fp32 MFMA weight-gradient kernel.
int ivln_wgrad_bf3_launch(ivln_gemm_desc& d, hipStream_t s, bool force) {
    constexpr bool disabled = false;  // A/B switches
    if ((!d.split_ok && !force) || (disabled && !force)) return IVLN_E_UNSUPPORTED;
    if (d.amode != AMODE_NCHW_P || d.bmode != BMODE_IM2COL_T || d.dmode != DMODE_DENSE || d.stride != 1 || d.dil != 1 || d.Cin <= 0 ||
        d.N != d.Cin * 49 || d.defer_epilogue || d.pad != 3)
        return IVLN_E_UNSUPPORTED;
    if (d.HoWo != d.Hout * d.Wout || d.K % d.HoWo != 0 || d.Hout != d.Hin || d.Wout != d.Win) return IVLN_E_UNSUPPORTED;
    if (d.Wout != 64 && d.Wout != 32 && d.Wout != 16 && d.Wout != 8) return IVLN_E_UNSUPPORTED;
    const int ims = d.Wout == 8 ? 2 : 1, rows = 128 / (d.Wout * ims);
    if (d.Hout % rows != 0 || (d.Wout == 8 && d.Hout != 8) || (((uintptr_t)d.A) & 7)) return IVLN_E_UNSUPPORTED;
    const int nimg = d.K / d.HoWo;
    if ((int64_t)nimg * d.in_img_stride >= (int64_t)1 << 31 || (int64_t)nimg * d.M * d.HoWo >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;
    const int strips = ((nimg + ims - 1) / ims) * (d.Hout / rows);
    // tile: 32 x 512, 64 x 512 or 128 x 256 (channels x columns); strips over blockIdx.z until a workgroup per CU
    // (32-channel outputs: 14 x 49 = 686 columns are two tiles of 384 - six waves - with 11 % of the columns idle; two tiles of
    //  512 left 33 % idle)
    constexpr int l1_env = 0;  // tuning: 6 | 8
    const bool six = d.M <= 32 && (l1_env ? l1_env == 6 : (d.N + 383) / 384 * 384 < (d.N + 511) / 512 * 512);
    // x promised exact in bf16 (split_ok = 2): the one-piece form, as 32 x 256 tiles of four waves - 43 KB of LDS, 3 workgroups per CU
    // (9.09 against 9.02 ms per update with 32 x 384 tiles of six waves, two per CU; 9.44 with the three-piece form)
    const bool xe = d.split_ok == 2 && d.M <= 32 && d.Wout == 64;
    const int BM = d.M <= 32 ? 32 : (d.M <= 64 ? 64 : 128), BN = xe ? 256 : (d.M <= 32 ? (six ? 384 : 512) : (d.M <= 64 ? 512 : 256));
    const int64_t blocks = (int64_t)((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM);
    if (!d.ws || d.ws_floats < (int64_t)d.M * d.N) return IVLN_E_UNSUPPORTED;  // (the kernel always leaves raw slabs)
    int splits = 1;
    if (d.splits == 0) {
        if (d.ws) {
            constexpr int want_env = 0;  // tuning
            // one workgroup per CU (LDS): as many splits as keep the grid inside whole rounds of 256 (13 column tiles x 20
            // splits = 260 workgroups ran a second round for four of them)
            const int64_t want = want_env > 0 ? want_env : (xe ? 3 : 1) * (int64_t)ivln_cu_count();
            splits = (int)(want / blocks);
            if (splits < 1) splits = 1;
            if (splits > strips) splits = strips;
            const int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
    } else {
        splits = d.splits > strips ? strips : d.splits;
        if (splits > 1 && (!d.ws || d.ws_floats < (int64_t)splits * d.M * d.N)) return IVLN_E_INVALID;
    }
    if (!force && blocks * splits < ivln_cu_count() / 2) return IVLN_E_UNSUPPORTED;  // (a rollout-sized batch: nothing to win)
    const int sps = (strips + splits - 1) / splits;
    splits = (strips + sps - 1) / sps;
    d.splits = splits;
    int rc;
    if (xe) rc = launch_wgrad_bf3<1, 1, 4, 64, true>(d, s, nimg, strips, sps);
    else if (d.M <= 32 && six) rc = launch_wgrad_bf3_w<1, 1, 6>(d, s, nimg, strips, sps);
    else if (d.M <= 32) rc = launch_wgrad_bf3_w<1, 1, 8>(d, s, nimg, strips, sps);
    else if (d.M <= 64) rc = launch_wgrad_bf3_w<2, 1, 8>(d, s, nimg, strips, sps);
    else rc = launch_wgrad_bf3_w<2, 2, 4>(d, s, nimg, strips, sps);
    if (rc == IVLN_OK) g_bf3_flops += 2.0 * d.M * (double)d.N * d.K, ++g_bf3_launches;
    return rc;
}
