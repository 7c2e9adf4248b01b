// The whole DD-PPO depth encoder of a rollout batch as ONE persistent launch (ivln_depth_net_f32): avg_pool2d(2) ->
// 7x7 stem -> GroupNorm + ReLU + MaxPool -> 16 GroupNorm bottlenecks -> 3x3 compression conv -> GroupNorm(1) + ReLU
// (habitat-lab ResNetEncoder; call site ivlnce_baselines/models/encoders/resnet_encoders.py:31-43, 95; restated in
// oracle/habitat_ext_ref.py:37-175).
//
// Why: at 4-8 images the encoder is 53 DEPENDENT conv layers of 5-120 MFLOP per image; as launches (k_nconv / k_gn_conv
// chain, round 2-3) a layer costs 8-13 us, and the round-3 stand-in showed that a grid-wide persistent form loses to the
// launches because 256 workgroups have to exchange slabs across XCDs.  But the images are independent all the way
// through (GroupNorm is per image), so nothing has to be grid-wide: a CLUSTER of 32 workgroups owns one image, and with
// cluster = blockIdx % 8 its workgroups sit on one XCD (observed dispatch order: workgroup b -> XCC (b + 7) % 8) and
// exchange through that XCD's L2.  Measured (tools/cluster_bench.hip, profiles/r04_cluster_bench.txt): counter barrier
// + 32 KB read / 4 KB written per workgroup = 2.6 us per layer with plain stores on one XCD, 2.9 us write-through.
//
// Per conv layer ("op"), every workgroup of the cluster:
//   1. prefetches the first 32 k-steps of ITS slice of the packed weights into registers (independent of the barrier),
//   2. waits on the cluster counter (all 32 workgroups have stored the previous layer),
//   3. merges the GroupNorm statistics partials its producers left (count, mean, M2 per (group, part): Chan's merge
//      written as the two-pass formula it equals) into a per-channel scale / shift table, loads its input rows (L1-
//      bypassing loads: another CU wrote them), applies GroupNorm (+ the downsample branch's GroupNorm | + the identity)
//      (+ ReLU) (+ the stem's MaxPool | the input's avg_pool2d) ON LOAD into an LDS tile with zero halo,
//   4. multiplies on the matrix cores: wave = (16-row output-channel tile, P pixel tiles of 16, a K range);
//      v_mfma_f32_16x16x4_f32, A (weights) streamed global -> registers in the packed per-lane order, B (activations)
//      gathered from the LDS tile; K ranges of a tile are summed through LDS,
//   5. stores the COMPLETE raw outputs of its (channel tile group, pixel group) plus the statistics partials of the
//      GroupNorm groups it covers, and arrives at the counter.
// No slabs (the output of a layer is written once, read 1-8 times from L2), no GroupNorm launches, no launch gaps.
// The compression conv (K = 9216) is split over 4 workgroups per tile (4 slabs), reduced by the final GroupNorm op.
//
// Placement independence: the counters are agent-scope atomics, every cross-workgroup load bypasses L1 (sc1), stores are
// write-through (sc1) until the cluster has PROVED it sits on one XCD (every workgroup ORs 1 << HW_REG_XCC_ID into a
// cluster word before its first arrival; after the first barrier a one-bit mask switches that cluster to plain stores).
// All 32 * N workgroups must be resident together (512 threads, ~120 KB LDS: one per CU); a spin that does not finish
// in ~0.2 s sets a sticky error word and the launch winds down.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/ivln_hip.h"
#include "family_timing.h"
#include "residency.h"

namespace {

constexpr int NT = 512;   // threads per workgroup
constexpr int CL = 32;    // workgroups per image
constexpr unsigned SPIN_MAX = 1u << 22;
#ifndef DEPTH_NET_MIN_BLOCKS
#define DEPTH_NET_MIN_BLOCKS 3  // minimum waves per SIMD the register allocation must allow (3 -> at most 170 VGPRs)
#endif
#ifndef DEPTH_NET_LOCAL_ATOMICS
#define DEPTH_NET_LOCAL_ATOMICS 1
#endif
constexpr bool LOCAL_ATOMICS = DEPTH_NET_LOCAL_ATOMICS != 0;  // A/B switch of the XCD-local arrival (cluster_arrive)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef ivln_depthnet_op Op;

// sync workspace (uint32 words): cluster c owns words [32 c, 32 c + 32): +0 arrivals, +1 exits, +2 XCC mask; word 256 = sticky error;
// word 257 = test hook (non-zero: workgroup 1 of cluster 0 leaves at entry without ever arriving - what a workgroup that is not
// resident looks like to the others); words 258-259 = device address of a HOST-visible uint32 (pinned memory) that also
// receives the error, or 0: the host then learns of a time-out at its next stream synchronisation without a read-back
constexpr int SY_ERR = 256, SY_TEST = 257, SY_HOST = 258;

struct fdiv {  // a / b for 0 <= a < 2^20 (see gn_conv.hip)
    float r;
    __device__ __forceinline__ explicit fdiv(int b) : r(__builtin_amdgcn_rcpf((float)b)) {}
    __device__ __forceinline__ int operator()(int a) const { return (int)(((float)a + 0.5f) * r); }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// L1-bypassing loads (aux 16 = sc1): data another workgroup of the cluster wrote during this launch
__device__ __forceinline__ float4 ld4_x(__amdgpu_buffer_rsrc_t r, int float_off) {
    const v4i x = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 16);
    return make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
}
// plain (cached) 16-byte / 4-byte loads through a resource: weights, parameters
__device__ __forceinline__ float4 ld4_w(__amdgpu_buffer_rsrc_t r, int float_off) {
    const v4i x = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 0);
    return make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
}
__device__ __forceinline__ float ld1_w(__amdgpu_buffer_rsrc_t r, int float_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, float_off * 4, 0, 0));
}
__device__ __forceinline__ float ld1_x(__amdgpu_buffer_rsrc_t r, int float_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, float_off * 4, 0, 16));
}
__device__ __forceinline__ void st4_x(__amdgpu_buffer_rsrc_t r, int float_off, float4 v, bool plain) {
    const v4i x = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
    if (plain) __builtin_amdgcn_raw_buffer_store_b128(x, r, float_off * 4, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(x, r, float_off * 4, 0, 16);
}
__device__ __forceinline__ void st1_x(__amdgpu_buffer_rsrc_t r, int float_off, float v, bool plain) {
    if (plain) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, float_off * 4, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, float_off * 4, 0, 16);
}

__device__ __forceinline__ float wave_sum(float v) {  // total in every lane (DPP rows, then two cross-row steps)
#define IVLN_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    IVLN_DPP_ADD(0xB1);
    IVLN_DPP_ADD(0x4E);
    IVLN_DPP_ADD(0x141);
    IVLN_DPP_ADD(0x140);
#undef IVLN_DPP_ADD
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float half_sum32(float v) {  // sum over the 32 lanes of a half wave, in each of them
#define IVLN_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    IVLN_DPP_ADD(0xB1);
    IVLN_DPP_ADD(0x4E);
    IVLN_DPP_ADD(0x141);
    IVLN_DPP_ADD(0x140);
#undef IVLN_DPP_ADD
    v += __shfl_xor(v, 16);
    return v;
}

#ifdef DEPTH_NET_TIMING  // tools/depth_net_phases.py: per-op phase stamps (100 MHz wall clock) of cluster 0's workgroups
__device__ unsigned long long g_dn_stamp[CL * 64 * 12];
#define DN_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (cluster == 0 && threadIdx.x == 0 && oi < 64) g_dn_stamp[(rank * 64 + oi) * 12 + (k)] = wall_clock64(); \
    } while (0)
#else
#define DN_STAMP(k)
#endif

struct Lds {  // float offsets into the dynamic LDS block (host-computed maxima over the program)
    int tab, btab, tile, otile, scratch;
};

// cluster barrier: wait until `target` arrivals.  false: timed out (sticky error set)
__device__ __forceinline__ bool cluster_wait(unsigned* sy, int cluster, unsigned target, int* s_flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(&sy[cluster * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_MAX || ((spins & 1023) == 0 && __hip_atomic_load(&sy[SY_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&sy[SY_ERR], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned* const host_flag = reinterpret_cast<unsigned*>(((unsigned long long)sy[SY_HOST + 1] << 32) | sy[SY_HOST]);
                if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}
// `local`: the cluster has proved to sit on one XCD - the arrival is then an atomic of that XCD's L2 (workgroup scope: no
// sc1, executed in the L2 every poller's L1-bypassing load is served from) instead of a memory-side agent-scope one
__device__ __forceinline__ void cluster_arrive(unsigned* sy, int cluster, bool local) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have left (write-through) / reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (local) __hip_atomic_fetch_add(&sy[cluster * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&sy[cluster * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One wave's share of the matrix product of an op: acc[p] += A(co tile, k range) x B(pixel tile p, k range).
//   A: packed weights, chunk c of this wave = 4 k-steps = one float4 per lane (M = 8: lanes with (lane & 15) >= 8 contribute 0),
//      eight chunks deep in registers; refills are unconditional loads on clamped chunk indices
//   B: the LDS tile at lane_base[p] + btab[chunk][kq][u], the per-op table of k-step -> tile offsets the workgroup built
//      while staging (one 16-byte LDS read per chunk and lane; k-steps past the range repeat the last valid offset and
//      meet zero weights).  One generic body for 1x1 / 3x3 / the 7x7 stem: the kernel is instruction-fetch bound when
//      every kernel size and every chunk has code of its own (first version: 57 KB of code, 140 instructions per chunk).
constexpr int MAXCH = 18;  // chunks (of 4 k-steps) per wave and op: 72 k-steps (the 3x3 convs of layer 4, the compression conv)
// Register diet (round 4, measured): 8 weight chunks + 4 input float4s in flight cost 218 VGPRs - two waves per SIMD then
// leave 76 registers per lane, and the first kernel of the neighbouring graph that needs more (the map CNN's convs: 137-189)
// waits for this launch to END.  4 + 2 in flight under __launch_bounds__(512, 3) are 150 VGPRs without a spill: 212 left.
#ifndef DEPTH_NET_ADEPTH
#define DEPTH_NET_ADEPTH 4
#endif
#ifndef DEPTH_NET_SB
#define DEPTH_NET_SB 2
#endif
constexpr int ADEPTH = DEPTH_NET_ADEPTH;  // weight chunks in flight per wave
constexpr int SB = DEPTH_NET_SB;          // input float4s in flight per thread while staging

template <int P>
struct MmaState {
    float4 abuf[MAXCH + ADEPTH];
    int4 off[MAXCH + 2];
    float b[MAXCH + 1][4][P];
};

// chunk C of the wave's K range, then chunk C + 1 ... (template recursion: every buffer index is a compile-time constant, so
// the buffers live in registers under their own names - a `#pragma unroll` loop with an early exit stayed rolled and put
// them in scratch memory; a rolled loop with rotating registers got its refills serialised behind s_waitcnt vmcnt(0))
template <int P, int C>
__device__ __forceinline__ void mma_chunk(MmaState<P>& S, __amdgpu_buffer_rsrc_t rW, int a_off0, int a_stride, int nch, const float* tile,
                                          const int4* bt, const int* lane_base, f32x4* acc, float a_mask) {
    if constexpr (C < MAXCH) {
        if (C >= nch) return;  // (wave-uniform)
        // issue: the weight chunk ADEPTH ahead, the offsets two chunks ahead, the activations of the next chunk
        S.abuf[C + ADEPTH] = ld4_w(rW, a_off0 + min(C + ADEPTH, nch - 1) * a_stride);
        S.off[C + 2] = bt[4 * min(C + 2, nch - 1)];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            S.b[C + 1][0][p] = tile[lane_base[p] + S.off[C + 1].x];
            S.b[C + 1][1][p] = tile[lane_base[p] + S.off[C + 1].y];
            S.b[C + 1][2][p] = tile[lane_base[p] + S.off[C + 1].z];
            S.b[C + 1][3][p] = tile[lane_base[p] + S.off[C + 1].w];
        }
        __builtin_amdgcn_sched_barrier(0);  // (otherwise every LDS read sinks to its MFMA: one LDS latency per k-step)
        const float av[4] = {S.abuf[C].x * a_mask, S.abuf[C].y * a_mask, S.abuf[C].z * a_mask, S.abuf[C].w * a_mask};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], S.b[C][e][p], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<P, C + 1>(S, rW, a_off0, a_stride, nch, tile, bt, lane_base, acc, a_mask);
    }
}

template <int P>
__device__ __forceinline__ void wave_mma(__amdgpu_buffer_rsrc_t rW, int a_off0, int a_stride, int nch, const float* tile,
                                         const int* btab_w, const int* lane_base, f32x4* acc, const float4* apre, float a_mask) {
    const int kq = (threadIdx.x & 63) >> 4;
    const int4* bt = reinterpret_cast<const int4*>(btab_w) + kq;  // chunk c: bt[4 c]
    MmaState<P> S;
#pragma unroll
    for (int u = 0; u < ADEPTH; ++u) S.abuf[u] = apre[u];
    S.off[0] = bt[0];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        S.b[0][0][p] = tile[lane_base[p] + S.off[0].x];
        S.b[0][1][p] = tile[lane_base[p] + S.off[0].y];
        S.b[0][2][p] = tile[lane_base[p] + S.off[0].z];
        S.b[0][3][p] = tile[lane_base[p] + S.off[0].w];
    }
    S.off[1] = bt[4 * min(1, nch - 1)];
    mma_chunk<P, 0>(S, rW, a_off0, a_stride, nch, tile, bt, lane_base, acc, a_mask);
}

__global__ __launch_bounds__(NT, DEPTH_NET_MIN_BLOCKS) void k_depth_net(const Op* __restrict__ ops, int n_ops, const float* wts, const float* prm,
                                                  const float* __restrict__ depth,
                                                  int64_t depth_img_stride, float* __restrict__ arena, int64_t arena_stride,
                                                  float* __restrict__ out, int64_t out_img_stride, int N, float eps,
                                                  unsigned* __restrict__ sy, const Lds L, int allow_plain) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cluster = blockIdx.x & 7, rank = blockIdx.x >> 3;
    if (cluster >= N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: scalar registers)
    int* s_flag = reinterpret_cast<int*>(smem);        // [0]: barrier verdict
    float* red = smem + 8;                              // 16 floats of block-reduction scratch
    float* tab = smem + L.tab;                          // scale | shift | scale2 | shift2, CMAX each
    int* btab = reinterpret_cast<int*>(smem + L.btab);  // k-step -> tile offset, [K range of a wave][chunk][kq][4]
    float* tile = smem + L.tile;
    float* otile = smem + L.otile;
    float* scratch = smem + L.scratch;
    float* A = arena + (int64_t)cluster * arena_stride;
    const __amdgpu_buffer_rsrc_t rA = rsrc(A);
    // Weights and affine parameters are read through buffer resources too, NOT through `const __restrict__` pointers: loads
    // from memory the compiler knows to be read-only and unaliased are free to SINK to their first use - past the cluster
    // barrier they are issued in front of, which is the whole point of issuing them there (measured: the first version's
    // MFMA phase waited a full L2 round trip per chunk; tools/depth_net_phases.py).
    const __amdgpu_buffer_rsrc_t rW = rsrc(wts), rP = rsrc(prm);
    const float* dimg = depth + (int64_t)cluster * depth_img_stride;
    // A launch whose sync words carry the sticky error of an EARLIER launch must not run: that launch left its arrival
    // counters wherever the time-out found them, every barrier here would pass at once and `out` would receive garbage without
    // any error.  Every workgroup leaves before it touches anything; the host clears the words (ivln_depth_net_reset).
    if (tid == 0) {
        int bad = __hip_atomic_load(&sy[SY_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (!bad && sy[SY_TEST] != 0 && cluster == 0 && rank == 1) bad = 2;  // (test hook: this workgroup "is not resident")
        *s_flag = bad;
    }
    __syncthreads();
    if (*s_flag) return;
    __syncthreads();  // (s_flag is reused by the barrier verdicts)
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&sy[cluster * 32 + 2], 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned arrivals = 0;  // per workgroup so far
    bool plain = false;     // stores may stay in this XCD's L2 (the whole cluster sits on it)
    bool alive = true;

    for (int oi = 0; oi < n_ops && alive; ++oi) {
        const Op op = ops[oi];
        const int n_tasks = op.n_ctg * op.n_ptg * op.kwg;
        const bool has_task = rank < n_tasks;
        // task -> (channel-tile group, pixel group, K slice of the workgroup)
        const int kwg_i = rank % op.kwg, tq = rank / op.kwg;
        const int ptg = tq % op.n_ptg, ctg = tq / op.n_ptg;
        // wave -> (channel tile, pixel-tile group, K range)
        const int wct = wave % op.WCT, wq = wave / op.WCT, wpt = wq % op.WPT, kw = wq / op.WPT;
        const int KWT = op.KW * op.kwg, kwt = kwg_i * op.KW + kw;
        const int per = (op.ksteps + KWT - 1) / KWT, cpk = (per + 3) >> 2;
        const int kbeg = min(kwt * per, op.ksteps), kend = min(kbeg + per, op.ksteps);
        const int nch = (kend - kbeg + 3) >> 2;
        const int ctile = ctg * op.WCT + wct;
        const int a_chunk0 = (ctile * KWT + kwt) * cpk;  // chunk index from the op's weight base
        const bool a_lane = op.M == 16 || (lane & 15) < 8;
        DN_STAMP(0);
        // ---- 1. weight prefetch: independent of what the other workgroups are still storing ----
        float4 abuf[ADEPTH];
        const int a_stride = op.M == 16 ? 256 : 128;  // floats per chunk
        // (M = 8: the lanes of accumulator rows 8..15 read a valid address too and are multiplied by 0)
        const int a_off0 = op.w_off + a_chunk0 * a_stride + (op.M == 16 ? lane * 4 : ((lane & 7) + 8 * (lane >> 4)) * 4);
        const float a_mask = a_lane ? 1.f : 0.f;
        if (op.kind == 0 && has_task && nch > 0) {
#pragma unroll
            for (int u = 0; u < ADEPTH; ++u) abuf[u] = ld4_w(rW, a_off0 + min(u, nch - 1) * a_stride);
        }
        // ... and the GroupNorm affine parameters of the channels this thread will write into the scale / shift table
        float gab[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, beb[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (op.kind == 0 && has_task && op.st_parts > 0) {
            const int cpg = op.Cin >> 4, g = tid >> 5, part = tid & 31;
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ccn = part + 32 * h;
                    if (ccn < cpg && (which == 0 || op.src2_off >= 0)) {
                        gab[which][h] = ld1_w(rP, (which ? op.gamma2_off : op.gamma_off) + g * cpg + ccn);
                        beb[which][h] = ld1_w(rP, (which ? op.beta2_off : op.beta_off) + g * cpg + ccn);
                    }
                }
        }
        const int Cin = op.Cin, Win = op.Win, Hin = op.Hin, Wp = op.wp, CS = op.cs, pad = op.pad;
        const int PG = 16 * op.WPT * op.P;          // output pixels of this task
        const int rows_out = PG >> op.wout_shift;   // whole output rows
        const int oy0 = ptg * rows_out;
        // a 1x1 stride-2 conv only ever reads every second row / column: the tile holds the sub-sampled map
        const bool sub = op.ks == 1 && op.stride == 2;
        const int s_eff = sub ? 1 : op.stride;
        const int Rs = (rows_out - 1) * s_eff + op.ks, iy0 = oy0 * s_eff - pad;
        const int HWin = Hin * Win;
        // a workgroup that owns a K slice (kwg > 1) stages the channels of that slice only
        int c_lo = 0, c_n = Cin;
        if (op.kwg > 1) {
            const int KK = op.ks * op.ks, kb_wg = min(kwg_i * op.KW * per, op.ksteps), ke_wg = min(kb_wg + op.KW * per, op.ksteps);
            c_lo = 4 * (kb_wg / KK);
            c_n = 4 * ((ke_wg + KK - 1) / KK) - c_lo;
        }
        const int CM = (L.btab - L.tab) >> 2;  // table stride
        const bool gn = op.st_parts > 0, two = op.src2_off >= 0, resid = op.res_off >= 0;
        const bool general = !op.avg_in && !op.pool;
        const int W4 = Win >> 2, per_c = Rs * W4, total = general ? c_n * per_c : 0;
        const fdiv by_pc(per_c), by_w4(W4);
        const bool wr_act = op.act_out_off >= 0 && ctg == 0 && kwg_i == 0;
        float4 xv[SB], yv[SB];  // yv: the second operand - the downsample branch's raw output OR the identity (never both)
        const int off2 = two ? op.src2_off : op.res_off;
        auto load_batch = [&](int i0, bool first_op, bool second_op) {  // four elements per thread, all their loads in flight together
#pragma unroll
            for (int e = 0; e < SB; ++e) {
                const int i = min(i0 + e * NT, total - 1);
                const int cl = by_pc(i), c = c_lo + cl, rem = i - cl * per_c, r = by_w4(rem), x4 = rem - r * W4;
                const int iy = sub ? 2 * (oy0 + r) : iy0 + r;
                if (first_op) xv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (second_op) yv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)iy < (unsigned)Hin) {
                    const int o = c * HWin + iy * Win + 4 * x4;
                    if (first_op) {
                        xv[e] = ld4_x(rA, op.src_off + o);
                        for (int z = 1; z < op.nslab; ++z) {
                            const float4 w = ld4_x(rA, op.src_off + z * op.slab_stride + o);
                            xv[e].x += w.x, xv[e].y += w.y, xv[e].z += w.z, xv[e].w += w.w;
                        }
                    }
                    if (second_op && (two || resid)) yv[e] = ld4_x(rA, off2 + o);
                }
            }
        };
        // the second operand of a block tail (downsample branch or identity) was stored at least one barrier ago: its
        // first batch is loaded in front of the barrier wait, like the weights
        if (op.kind == 0 && has_task && total > 0 && (two || resid)) load_batch(tid, false, true);
        // ---- 2. the cluster has stored everything this op reads ----
        if (op.barrier_before) {
            if (!cluster_wait(sy, cluster, arrivals * CL, s_flag)) {
                alive = false;
                break;
            }
            if (!plain && arrivals == 1) {  // first barrier passed: every workgroup of the cluster has reported its XCC
                const unsigned m = __hip_atomic_load(&sy[cluster * 32 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                plain = allow_plain && __builtin_popcount(m) == 1;
            }
        }
        DN_STAMP(1);
        if (op.kind == 1) {
            // ---- final GroupNorm(1 group) + ReLU over the `nslab` slabs of the compression conv: workgroup 0 ----
            if (rank == 0) {
                const int nel = op.Cin * op.Hin * op.Win;  // 128 * 16
                float v[4];
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = tid + e * NT;
                    v[e] = 0.f;
                    if (i < nel) {
                        for (int z = 0; z < op.nslab; ++z) v[e] += ld1_x(rA, op.src_off + z * op.slab_stride + i);
                        s += v[e];
                    }
                }
                s = wave_sum(s);
                if (lane == 0) red[wave] = s;
                __syncthreads();
                float tot = 0.f;
                for (int w8 = 0; w8 < NT / 64; ++w8) tot += red[w8];
                const float mean = tot / (float)nel;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tid + e * NT < nel) q += (v[e] - mean) * (v[e] - mean);
                q = wave_sum(q);
                __syncthreads();
                if (lane == 0) red[wave] = q;
                __syncthreads();
                tot = 0.f;
                for (int w8 = 0; w8 < NT / 64; ++w8) tot += red[w8];
                const float rstd = rsqrtf(tot / (float)nel + eps);
                const int HW = op.Hin * op.Win;
                const fdiv by_hw(HW);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = tid + e * NT;
                    if (i < nel) {
                        const int c = by_hw(i);
                        const float ga = ld1_w(rP, op.gamma_off + c) * rstd, be = ld1_w(rP, op.beta_off + c) - mean * ga;
                        out[(int64_t)cluster * out_img_stride + i] = fmaxf(fmaf(v[e], ga, be), 0.f);
                    }
                }
            }
            continue;
        }
        if (has_task) {
            // ---- 3. input rows -> LDS tile [channels][Rs][Wp] (zero halo), transformed on load.  Every global load of
            // the phase is issued first - the statistics partials, then the first batch of input float4s - so that the
            // scale / shift table is computed under the inputs' flight ----
            const int sg = tid >> 5, spart = tid & 31;
            float sn[2] = {0.f, 0.f}, sm[2] = {0.f, 0.f}, sM[2] = {0.f, 0.f};
            if (gn) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int parts = which ? op.st2_parts : op.st_parts, soff = which ? op.st2_off : op.st_off;
                    if ((which == 0 || two) && spart < parts) {  // one 16-byte record (count, mean, M2, -) per (group, part)
                        const float4 rec = ld4_x(rA, soff + (sg * parts + spart) * 4);
                        sn[which] = rec.x, sm[which] = rec.y, sM[which] = rec.z;
                    }
                }
            }
            if (total > 0) load_batch(tid, true, false);
            // ---- scale / shift tables: 16 groups x up to 32 parts, one half wave per group (Chan's merge as the two-pass formula) ----
            if (gn) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    if (which == 0 || two) {
                        const float n = sn[which], m = sm[which];
                        const float cnt = half_sum32(n), mean = half_sum32(n * m) / cnt;
                        const float dd = m - mean;
                        const float rstd = rsqrtf(half_sum32(fmaf(n * dd, dd, sM[which])) / cnt + eps);
                        const int cpg = Cin >> 4;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ccn = spart + 32 * h;
                            if (ccn < cpg) {
                                const int c = sg * cpg + ccn;
                                const float sc = gab[which][h] * rstd;
                                tab[(2 * which) * CM + c] = sc;
                                tab[(2 * which + 1) * CM + c] = beb[which][h] - mean * sc;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            // ---- k-step -> tile offset table of this workgroup's K ranges (pure ALU; the barrier below publishes it) ----
            {
                const int KK = op.ks * op.ks, q_lo = c_lo >> 2, n_ent = op.KW * cpk * 16;
                for (int e = tid; e < n_ent; e += NT) {
                    const int u = e & 3, kq_ = (e >> 2) & 3, ch = (e >> 4) % cpk, kw_ = (e >> 4) / cpk;
                    const int kb = min((kwg_i * op.KW + kw_) * per, op.ksteps), ke = min(kb + per, op.ksteps);
                    int off = 0;
                    if (ke > kb) {
                        const int k = min(kb + 4 * ch + u, ke - 1);
                        if (op.ks == 7) {
                            const int tap = min(4 * k + kq_, KK - 1), dy = tap / 7;
                            off = dy * Wp + (tap - 7 * dy);
                        } else {
                            const int q = k / KK, t = k - q * KK, dy = t / op.ks;
                            off = (4 * (q - q_lo) + kq_) * CS + dy * Wp + (t - dy * op.ks);
                        }
                    }
                    btab[e] = off;
                }
            }
            DN_STAMP(2);
            if (op.avg_in) {
                // op 0: F.avg_pool2d(depth, 2) of the raw 2 Hin x 2 Win image, rows iy0 .. iy0 + Rs
                const int Wr = 2 * Win;
                for (int i = tid; i < Rs * Wp; i += NT) {
                    const int r = i / Wp, xp = i - r * Wp, x = xp - pad, iy = iy0 + r;
                    float v = 0.f;
                    if ((unsigned)iy < (unsigned)Hin && (unsigned)x < (unsigned)Win) {
                        const float2 a = *reinterpret_cast<const float2*>(dimg + (int64_t)(2 * iy) * Wr + 2 * x);
                        const float2 b = *reinterpret_cast<const float2*>(dimg + (int64_t)(2 * iy + 1) * Wr + 2 * x);
                        v = (((a.x + a.y) + b.x) + b.y) * 0.25f;
                    }
                    tile[i] = v;
                }
            } else if (op.pool) {
                // MaxPool2d(3, 2, 1) of relu(GN(raw)) on load: raw map 2 Hin x 2 Win; ks == 1 here (one staged row per output row)
                const int Wr = 2 * Win, Hr = 2 * Hin;
                const fdiv by_row(Rs * Win), by_w(Win);
                for (int i = tid; i < Cin * Rs * Win; i += NT) {
                    const int c = by_row(i), rem = i - c * Rs * Win, r = by_w(rem), x = rem - r * Win, iy = iy0 + r;
                    const float sc = tab[c], sh = tab[CM + c];
                    float v9[9];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) {  // nine independent loads (clamped addresses), then the maximum of the valid ones
                            const int hh = min(max(2 * iy - 1 + a, 0), Hr - 1), ww = min(max(2 * x - 1 + b, 0), Wr - 1);
                            v9[a * 3 + b] = ld1_x(rA, op.src_off + c * Hr * Wr + hh * Wr + ww);
                        }
                    float m = -INFINITY;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            const int hh = 2 * iy - 1 + a, ww = 2 * x - 1 + b;
                            if ((unsigned)hh < (unsigned)Hr && (unsigned)ww < (unsigned)Wr) {
                                float v = fmaf(v9[a * 3 + b], sc, sh);
                                if (op.relu) v = fmaxf(v, 0.f);
                                m = fmaxf(m, v);
                            }
                        }
                    tile[c * CS + r * Wp + pad + x] = m;
                }
            } else {
                for (int i0 = tid; i0 < total; i0 += NT * SB) {
                    if (i0 != tid) load_batch(i0, true, true);
#pragma unroll
                    for (int e = 0; e < SB; ++e) {
                        if (i0 + e * NT < total) {
                            const int i = i0 + e * NT;  // (indices recomputed: cheaper than 12 live registers)
                            const int cl = by_pc(i), c = c_lo + cl, rem = i - cl * per_c, r = by_w4(rem), x4 = rem - r * W4;
                            const int iy = sub ? 2 * (oy0 + r) : iy0 + r;
                            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                            if ((unsigned)iy < (unsigned)Hin) {
                                a = xv[e];
                                if (gn) {
                                    const float sc = tab[c], sh = tab[CM + c];
                                    a.x = fmaf(a.x, sc, sh), a.y = fmaf(a.y, sc, sh), a.z = fmaf(a.z, sc, sh), a.w = fmaf(a.w, sc, sh);
                                    if (two) {
                                        const float s2 = tab[2 * CM + c], h2 = tab[3 * CM + c];
                                        a.x += fmaf(yv[e].x, s2, h2), a.y += fmaf(yv[e].y, s2, h2);
                                        a.z += fmaf(yv[e].z, s2, h2), a.w += fmaf(yv[e].w, s2, h2);
                                    }
                                }
                                if (resid) a.x += yv[e].x, a.y += yv[e].y, a.z += yv[e].z, a.w += yv[e].w;
                                if (op.relu) a.x = fmaxf(a.x, 0.f), a.y = fmaxf(a.y, 0.f), a.z = fmaxf(a.z, 0.f), a.w = fmaxf(a.w, 0.f);
                                if (wr_act) st4_x(rA, op.act_out_off + c * HWin + iy * Win + 4 * x4, a, plain);
                            }
                            if (sub) {  // columns 4 x4 and 4 x4 + 2 of the input row = outputs 2 x4, 2 x4 + 1
                                float* tq = tile + (c - c_lo) * CS + r * Wp + 2 * x4;
                                tq[0] = a.x, tq[1] = a.z;
                                continue;
                            }
                            float* tp = tile + (c - c_lo) * CS + r * Wp + pad + 4 * x4;
                            if (pad == 0) {
                                *reinterpret_cast<float4*>(tp) = a;  // (CS, Wp multiples of 4: 16-byte aligned)
                            } else {
                                tp[0] = a.x, tp[1] = a.y, tp[2] = a.z, tp[3] = a.w;
                                if (x4 == 0)
                                    for (int z = 1; z <= pad; ++z) tp[-z] = 0.f;
                                if (x4 == W4 - 1)
                                    for (int z = 0; z < pad; ++z) tp[4 + z] = 0.f;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            DN_STAMP(3);
            // ---- 4. matrix product ----
            const int kq = lane >> 4, j = lane & 15;
            int lane_base[2];
            f32x4 acc[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int qpx = (wpt * op.P + p) * 16 + j;
                const int oyl = qpx >> op.wout_shift, ox = qpx & ((1 << op.wout_shift) - 1);
                lane_base[p] = oyl * s_eff * Wp + ox * s_eff;
                acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            DN_STAMP(8);
            if (nch > 0) {
                const int* btab_w = btab + kw * cpk * 16;
                if (op.P == 2) wave_mma<2>(rW, a_off0, a_stride, nch, tile, btab_w, lane_base, acc, abuf, a_mask);
                else wave_mma<1>(rW, a_off0, a_stride, nch, tile, btab_w, lane_base, acc, abuf, a_mask);
            }
            DN_STAMP(9);
            __syncthreads();  // every wave is done with the input tile: the output tile and the K-range scratch reuse its LDS
            DN_STAMP(4);
            // ---- 5. complete output tile in LDS: otile[WCT * M rows][PG + 4] ----
            const int M = op.M, OP_ = PG + 4, rows_t = op.WCT * M;
            const bool row_lane = M == 16 || kq < 2;  // (M = 8: accumulator rows 8..15 are padding)
            if (op.KW == 1) {
                if (row_lane)
                    for (int p = 0; p < op.P; ++p)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            otile[(wct * M + 4 * kq + r) * OP_ + (wpt * op.P + p) * 16 + j] = acc[p][r];
                __syncthreads();
            } else {
                for (int p = 0; p < op.P; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scratch[((wave * op.P + p) * 4 + r) * 64 + lane] = acc[p][r];
                __syncthreads();
                const fdiv by_pg(PG), by_m(M), by_p(op.P);
                for (int e = tid; e < rows_t * PG; e += NT) {
                    const int row = by_pg(e), px = e - row * PG, ct = by_m(row), i = row - ct * M;
                    const int pxt = px >> 4, jj = px & 15, wp_ = by_p(pxt), p = pxt - wp_ * op.P;
                    const int ln = jj + 16 * (i >> 2), r = i & 3;
                    float s = 0.f;
                    for (int k2 = 0; k2 < op.KW; ++k2)
                        s += scratch[((((k2 * op.WPT + wp_) * op.WCT + ct) * op.P + p) * 4 + r) * 64 + ln];
                    otile[row * OP_ + px] = s;
                }
                __syncthreads();
            }
            DN_STAMP(5);
            // raw outputs: whole pixel rows, 16 bytes per lane
            const int HWo = 1 << (2 * op.wout_shift), co0 = ctg * rows_t, px0 = ptg * PG;
            {
                const int PG4 = PG >> 2;
                const fdiv by_pg4(PG4);
                const int dbase = op.dst_off + kwg_i * op.dst_slab_stride;
                for (int e = tid; e < rows_t * PG4; e += NT) {
                    const int row = by_pg4(e), x4 = e - row * PG4;
                    if (co0 + row < op.Cout)
                        st4_x(rA, dbase + (co0 + row) * HWo + px0 + 4 * x4, *reinterpret_cast<const float4*>(&otile[row * OP_ + 4 * x4]), plain);
                }
            }
            // statistics partials of the GroupNorm groups this tile covers: one wave per local group, one shifted pass
            if (op.st_out_parts > 0) {
                const int cpo = op.Cout >> 4;                      // channels per output group
                const int rows_lg = min(cpo, rows_t), n_lg = rows_t / rows_lg;
                const int cparts = max(1, cpo / rows_t);
                const int part = ptg * cparts + (cpo > rows_t ? (co0 % cpo) / rows_t : 0);
                const int nel = rows_lg * PG;
                const fdiv by_pg(PG);
                for (int lg = wave; lg < n_lg; lg += NT / 64) {
                    const float* og = otile + lg * rows_lg * OP_;
                    const float pilot = og[0];
                    float s1 = 0.f, s2 = 0.f;
                    for (int i = lane; i < nel; i += 64) {
                        const int rl = by_pg(i);
                        const float dd = og[rl * OP_ + (i - rl * PG)] - pilot;
                        s1 += dd;
                        s2 = fmaf(dd, dd, s2);
                    }
                    s1 = wave_sum(s1);
                    s2 = wave_sum(s2);
                    if (lane == 0) {
                        const int g = co0 / cpo + (cpo > rows_t ? 0 : lg);
                        st4_x(rA, op.st_out_off + (g * op.st_out_parts + part) * 4,
                              make_float4((float)nel, pilot + s1 / (float)nel, fmaxf(s2 - s1 * s1 / (float)nel, 0.f), 0.f), plain);
                    }
                }
            }
        }
        // ---- arrive when the next op (or the end of the program) has to see these stores ----
        DN_STAMP(6);
        if (oi + 1 < n_ops && ops[oi + 1].barrier_before) {
            cluster_arrive(sy, cluster, plain && LOCAL_ATOMICS);
            ++arrivals;
            DN_STAMP(7);
        } else {
            __syncthreads();  // (LDS is reused by the next op)
        }
    }
    // ---- exit: the last workgroup of the cluster to leave resets the cluster's words for the next launch ----
    if (tid == 0 && alive) {
        const unsigned old = __hip_atomic_fetch_add(&sy[cluster * 32 + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == CL - 1) {
            __hip_atomic_store(&sy[cluster * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy[cluster * 32 + 2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy[cluster * 32 + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" {

int ivln_depth_net_f32(const ivln_depthnet_op* ops_dev, const ivln_depthnet_op* ops_host, int n_ops, const float* weights,
                       const float* params, const float* depth, int64_t depth_img_stride, float* arena, int64_t arena_stride,
                       float* out, int64_t out_img_stride, int N, float eps, void* sync_ws, void* stream) {
    if (!ops_dev || !ops_host || n_ops <= 0 || !weights || !params || !depth || !arena || !out || !sync_ws) return IVLN_E_INVALID;
    if (N < 1 || N > 8) return IVLN_E_UNSUPPORTED;  // one image per XCD-sized cluster
    // LDS layout from the program's maxima
    int cmax = 0, tile = 0, otile = 0, scr = 0, btab_max = 0;
    for (int i = 0; i < n_ops; ++i) {
        const ivln_depthnet_op& o = ops_host[i];
        if (o.kind != 0) continue;
        if (o.WCT * o.WPT * o.KW != NT / 64 || (o.P != 1 && o.P != 2) || (o.M != 16 && o.M != 8) || o.n_ctg * o.n_ptg * o.kwg > CL ||
            (o.ks != 1 && o.ks != 3 && o.ks != 7) || (o.Win & 3))
            return IVLN_E_INVALID;
        const int PG = 16 * o.WPT * o.P, rows_out = PG >> o.wout_shift;
        if (rows_out < 1 || (rows_out << o.wout_shift) != PG) return IVLN_E_INVALID;
        const bool sub = o.ks == 1 && o.stride == 2;
        const int Rs = (rows_out - 1) * (sub ? 1 : o.stride) + o.ks;
        int c_n = o.Cin;
        if (o.kwg > 1) {  // a K-slice workgroup stages its own channels only
            const int KWT = o.KW * o.kwg, per = (o.ksteps + KWT - 1) / KWT, KK = o.ks * o.ks;
            c_n = 4 * ((o.KW * per + KK - 1) / KK + 1);
            if (c_n > o.Cin) c_n = o.Cin;
        }
        const int need = o.ks == 7 ? Rs * o.wp : c_n * o.cs;
        if (o.cs < Rs * o.wp || (sub && (o.act_out_off >= 0 || o.pool || o.wp != (1 << o.wout_shift)))) return IVLN_E_INVALID;
        if (o.st_parts > 32 || o.st2_parts > 32 || o.st_out_parts > 32 || (o.src2_off >= 0 && o.res_off >= 0)) return IVLN_E_INVALID;
        cmax = o.Cin > cmax ? o.Cin : cmax;
        tile = need > tile ? need : tile;
        const int ot = o.WCT * o.M * (PG + 4);
        otile = ot > otile ? ot : otile;
        {
            const int KWT = o.KW * o.kwg, per = (o.ksteps + KWT - 1) / KWT, cpk = (per + 3) >> 2;
            if (cpk > MAXCH) return IVLN_E_UNSUPPORTED;
            const int bt = o.KW * cpk * 16;
            btab_max = bt > btab_max ? bt : btab_max;
        }
        const int sc = o.KW > 1 ? (NT / 64) * o.P * 256 : 0;
        scr = sc > scr ? sc : scr;
    }
    Lds L;
    L.tab = 32;
    cmax = (cmax + 3) & ~3;
    L.btab = L.tab + 4 * cmax;
    L.tile = L.btab + btab_max;
    L.otile = L.tile;  // the output tile and the K-range scratch reuse the input tile's LDS (a barrier separates the phases)
    L.scratch = L.otile + ((otile + 3) & ~3);
    const int body = tile > ((otile + 3) & ~3) + scr ? tile : ((otile + 3) & ~3) + scr;
    const size_t lds = sizeof(float) * (size_t)(L.tile + body + 8);
    if (lds > 156 * 1024) return IVLN_E_UNSUPPORTED;
    static std::mutex mu;
    static bool attr_done = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_done) {
            if (hipFuncSetAttribute((const void*)k_depth_net, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
                return IVLN_E_HIP;
            attr_done = true;
        }
    }
    // residency with THIS launch's LDS footprint on THIS device (csrc/residency.h; the first version asked once, for
    // 120 KB, on whichever device was current)
    const int resident = ivln_resident_blocks((const void*)k_depth_net, NT, lds);
    if (resident < 8 * CL) return IVLN_E_UNSUPPORTED;
    // (the clusters spin on each other's arrivals: all of them resident, or none)
    constexpr int allow_plain = 1;  // A/B switch: write-through stores only (519 -> 545 us)
#ifdef DEPTH_NET_TIMING
    hipLaunchKernelGGL(k_depth_net, dim3(8 * CL), dim3(NT), lds, (hipStream_t)stream, ops_dev, n_ops, weights, params, depth,
                       depth_img_stride, arena, arena_stride, out, out_img_stride, N, eps, (unsigned*)sync_ws, L, allow_plain);
#else
    IVLN_LAUNCH_FAMILY(k_depth_net, dim3(8 * CL), dim3(NT), lds, (hipStream_t)stream, ops_dev, n_ops, weights, params, depth,
                       depth_img_stride, arena, arena_stride, out, out_img_stride, N, eps, (unsigned*)sync_ws, L, allow_plain);
#endif
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

#ifdef DEPTH_NET_TIMING
int ivln_depth_net_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dn_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

/* Clears the counters and the sticky error of a sync workspace (words 0..257; the host-flag address in words 258-259 stays)
 * on `stream`: after a time-out, before the workspace is used again. */
int ivln_depth_net_reset(void* sync_ws, void* stream) {
    if (!sync_ws) return IVLN_E_INVALID;
    return hipMemsetAsync(sync_ws, 0, sizeof(unsigned) * SY_HOST, (hipStream_t)stream) == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

/* Device-side address of a pinned host allocation (hipHostGetDevicePointer): what words 258-259 of a sync workspace hold. */
int ivln_host_device_ptr(void* host, void** dev) {
    if (!host || !dev) return IVLN_E_INVALID;
    return hipHostGetDevicePointer(dev, host, 0) == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

/* Synchronises `stream` and reads the sticky error word of a depth-net sync workspace. */
int ivln_depth_net_status(const void* sync_ws, void* stream) {
    unsigned err = 0;
    if (hipMemcpyAsync(&err, (const unsigned*)sync_ws + SY_ERR, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return IVLN_E_HIP;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return IVLN_E_HIP;
    return err ? IVLN_E_HIP : IVLN_OK;
}

}  // extern "C"
