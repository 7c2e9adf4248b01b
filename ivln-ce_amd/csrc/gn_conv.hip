// GroupNorm (+ second normalised operand / residual) (+ ReLU) (+ MaxPool) fused with the NEXT convolution(s), for the
// DD-PPO depth ResNet at rollout batch sizes (habitat-lab ResNetEncoder: conv -> GroupNorm(16, C) -> ReLU chains and
// Bottleneck tails relu(convs(x) + downsample(x)) / relu(convs(x) + x); call site
// ivlnce_baselines/models/encoders/resnet_encoders.py:31-43, 95; restated in oracle/habitat_ext_ref.py:37-175).
//
// The encoder's chain at 4-8 envs is launch-bound: 105 dependent launches of ~6 us (conv with split-K slabs, then a
// GroupNorm launch that reduces them).  Fusing the conv INTO ITS OWN GroupNorm (csrc/conv_gn.hip: workgroup = (image,
// group) with the whole K reduction inside) lost - the block then has to stream the image's whole input.  This
// kernel fuses the other way round: a workgroup still owns one (image, group) of GroupNorm i - it reduces the slabs,
// normalises, activates - and then multiplies ITS cpg activated channels into the NEXT convolution:
//     y_{i+1}[co][p] = sum_g  sum_{c in group g, taps} W[co][c][tap] a_i[c][p + tap]
// i.e. the next conv is split over K by GroupNorm group, each block writes one partial slab (all Cout, all pixels of
// its image, its cpg*k*k slice of K), and the NEXT GroupNorm kernel reduces the 16 slabs exactly like the split-K
// slabs it reduces today.  K per block is tiny (cpg*k*k = 8..576), the activated tile sits in LDS, weights are the
// block's wave-uniform [co][cpg*k*k] slices: one launch per conv layer, 52 instead of 105 for the encoder.
// The bottleneck's first conv and its downsample conv share the producing kernel (conv A and conv B).
//
// Bound: L2 -> CU bandwidth for the slab reduction (16 x the tile) and the slab write (Cout x pixels per block), then
// fp32 VALU for the partial conv (0.26 - 0.6 MMAC per block).  Not MFMA-shaped: K per block is 8-72 in the layers
// that matter.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <set>
#include "../../include/ivln_hip.h"
#include "family_timing.h"

namespace {

#ifndef GN_CONV_THREADS
#define GN_CONV_THREADS 512
#endif
constexpr int GT = GN_CONV_THREADS;  // threads per block
// k_nconv's statistics merge gives threads [0, 256) to the first operand and [256, 512) to the second, and the loader /
// MFMA split counts whole waves: other block sizes are not a tuning knob without touching those
static_assert(GT >= 512 && GT % 256 == 0, "GN_CONV_THREADS must be a multiple of 256 and at least 512");
constexpr int NW = GT / 64;      // waves per block
constexpr int ZW = 2 * NW;       // index of the zero word in the reduction scratch
constexpr int TILE_MAX = 8192;   // cpg * H * W floats held in LDS
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr size_t kLdsFloats = 38 * 1024;  // 152 KB of the CU's 160 KB

typedef ivln_gn_conv_desc Desc;

#ifdef GN_CONV_TIMING  // tools/gn_conv_phases.py: per-block phase stamps (100 MHz wall clock)
__device__ unsigned long long g_stamp[4096 * 8];
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        __syncthreads();                                                                                 \
        if (threadIdx.x == 0) g_stamp[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(k)
#endif

// wave64 sum on the DPP cross-lane path (6 VALU ops; __shfl_xor is an LDS permute per step): total in lane 63
__device__ __forceinline__ float wave_sum(float v) {
#define IVLN_DPP_ADD(ctrl, row_mask)                                                                                   \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    IVLN_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
    IVLN_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
    IVLN_DPP_ADD(0x141, 0xf);  // row_half_mirror
    IVLN_DPP_ADD(0x140, 0xf);  // row_mirror: every lane of a row of 16 holds the row's sum
    IVLN_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
    IVLN_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
#undef IVLN_DPP_ADD
    return v;
}
// block-wide sums of two values at once (red: 32 floats of LDS)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) {
        red[w] = a;
        red[NW + w] = b;
    }
    __syncthreads();
    a = 0.f;
    b = 0.f;
#pragma unroll
    for (int i = 0; i < GT / 64; ++i) {
        a += red[i];
        b += red[NW + i];
    }
}

// a / b for 0 <= a < 2^20 without the ~40-instruction integer division: (a + 0.5) / b is at least 0.5 / b away from an
// integer, far more than the float error
struct fdiv {
    float r;
    __device__ __forceinline__ explicit fdiv(int b) : r(__builtin_amdgcn_rcpf((float)b)) {}
    __device__ __forceinline__ int operator()(int a) const { return (int)(((float)a + 0.5f) * r); }
};

// launch geometry derived on the host (no divisions by launch constants in the kernel)
struct Geo {
    int cpg, Hc, Wc;           // channels per group; the convs' input size (after the pool)
    int Ho_a, Wo_a, Ho_b, Wo_b;
    int per_a, per_b;          // output channels per blockIdx.z
    int wl_floats, wthr, part_floats;
    int fthr, w0thr;           // front stage: threads that load its slabs / its weight slice + affine parameters
};

template <int V> struct vecf;
template <> struct vecf<4> { typedef float4 type; };
template <> struct vecf<1> { typedef float type; };
__device__ __forceinline__ void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float& a, const float& b) { a += b; }
__device__ __forceinline__ float vsum(const float4& a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float vsum(const float& a) { return a; }

// `splits` slabs at p (stride slab_stride) summed in slab order; the 16-slab case (the output of a previous gn_conv)
// keeps all its loads in flight at once
template <int V>
__device__ __forceinline__ typename vecf<V>::type slab_sum(const float* __restrict__ p, int splits, int64_t slab_stride) {
    typedef typename vecf<V>::type T;
    T v = *reinterpret_cast<const T*>(p);
    int z = 1;
    if (splits == 16) {
        T w[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) w[q] = *reinterpret_cast<const T*>(p + (int64_t)(q + 1) * slab_stride);
#pragma unroll
        for (int q = 0; q < 15; ++q) vadd(v, w[q]);
        return v;
    }
    for (; z + 3 < splits; z += 4) {
        const T w0 = *reinterpret_cast<const T*>(p + (int64_t)z * slab_stride);
        const T w1 = *reinterpret_cast<const T*>(p + (int64_t)(z + 1) * slab_stride);
        const T w2 = *reinterpret_cast<const T*>(p + (int64_t)(z + 2) * slab_stride);
        const T w3 = *reinterpret_cast<const T*>(p + (int64_t)(z + 3) * slab_stride);
        vadd(v, w0); vadd(v, w1); vadd(v, w2); vadd(v, w3);
    }
    for (; z < splits; ++z) vadd(v, *reinterpret_cast<const T*>(p + (int64_t)z * slab_stride));
    return v;
}

// One pass over the block's (image, group) tile by threads [0, nthr): the slabs of x (and of the second operand x2)
// summed into LDS, the residual copied next to them - every global load of the GroupNorm front-end is issued in
// this one loop, so the tile costs one memory round trip.  s / s2: this thread's partial sums of x / x2.
template <int V, bool FRONT>
__device__ __forceinline__ void load_tiles(const Desc& D, int n, int c0, int cpg, int HW, int nthr, float* tile, float* tile2,
                                           float* tres, float& s, float& s2) {
    typedef typename vecf<V>::type T;
    const int nel = cpg * HW;
    const int64_t NHW = (int64_t)D.N * HW;
    const fdiv by_hw(HW);
    for (int i = threadIdx.x * V; i < nel; i += nthr * V) {
        const int cl = by_hw(i), pp = i - cl * HW;
        const int64_t o = (int64_t)(c0 + cl) * NHW + (int64_t)n * HW + pp;
        T r;
        if (D.residual) r = *reinterpret_cast<const T*>(D.residual + ((int64_t)n * D.C + c0) * HW + i);
        if (!FRONT) {  // (with a front stage the tile is computed in the block)
            const T v = slab_sum<V>(D.x + o, D.splits, D.slab_stride);
            *reinterpret_cast<T*>(&tile[i]) = v;
            s += vsum(v);
        }
        if (D.x2) {
            const T v2 = slab_sum<V>(D.x2 + o, D.splits2, D.slab_stride2);
            *reinterpret_cast<T*>(&tile2[i]) = v2;
            s2 += vsum(v2);
        }
        if (D.residual) *reinterpret_cast<T*>(&tres[i]) = r;
    }
}

// Front stage, part 1: the slabs of ALL C0 channels of image n summed into act0[C0][HW] (LDS) by threads [t0, t0+nthr)
template <int V>
__device__ __forceinline__ void load_front(const Desc& D, int n, int HW, int t0, int nthr, float* act0) {
    typedef typename vecf<V>::type T;
    const int tid = (int)threadIdx.x - t0;
    if (tid < 0 || tid >= nthr) return;
    const int nel = D.C0 * HW;
    const int64_t NHW = (int64_t)D.N * HW;
    const fdiv by_hw(HW);
    for (int i = tid * V; i < nel; i += nthr * V) {
        const int c = by_hw(i), pp = i - c * HW;
        *reinterpret_cast<T*>(&act0[i]) = slab_sum<V>(D.x0 + (int64_t)c * NHW + (int64_t)n * HW + pp, D.splits0, D.slab_stride0);
    }
}
// Front stage, part 2: GroupNorm (+ ReLU) of every group of act0 in place, one wave per group at a time (wave-level
// two-pass statistics, no block barrier inside); gb0 = gamma0 | beta0 staged in LDS.
__device__ __forceinline__ void norm_front(const Desc& D, int HW, float* act0, const float* gb0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpg0 = D.C0 / D.groups0, nel = cpg0 * HW;
    const fdiv by_hw(HW);
    for (int g0 = wave; g0 < D.groups0; g0 += GT / 64) {
        float* t = act0 + g0 * nel;
        float s = 0.f;
        for (int i = lane; i < nel; i += 64) s += t[i];
        const float mean = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum(s)), 63)) / (float)nel;
        float q = 0.f;
        for (int i = lane; i < nel; i += 64) {
            const float d = t[i] - mean;
            q += d * d;
        }
        const float var = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum(q)), 63)) / (float)nel;
        const float rstd = rsqrtf(var + D.eps);
        for (int i = lane; i < nel; i += 64) {
            const int c = g0 * cpg0 + by_hw(i);
            const float ga = gb0[c] * rstd, be = gb0[D.C0 + c] - mean * ga;
            t[i] = fmaxf(fmaf(t[i], ga, be), 0.f);
        }
    }
}

// This block's weight slice W[co0 .. co0+nrow)[koff .. koff+Kb) (rows of K floats in global) -> wl[nrow][Kbp] (LDS) by
// threads [t0, t0+nthr), float4 when the rows allow it.
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, int K, int koff, int Kb, int Kbp, int co0, int nrow,
                                              float* wl, int t0, int nthr) {
    const int tid = (int)threadIdx.x - t0;
    if (tid < 0 || tid >= nthr) return;
    if (((Kb | K) & 3) == 0) {  // koff = group * Kb is then a multiple of 4 too
        const int q = Kb >> 2, tot = nrow * q;
        const fdiv by_q(q);
#pragma unroll 4
        for (int e = tid; e < tot; e += nthr) {
            const int r = by_q(e), c = e - r * q;
            *reinterpret_cast<float4*>(wl + r * Kbp + 4 * c) =
                *reinterpret_cast<const float4*>(w + (int64_t)(co0 + r) * K + koff + 4 * c);
        }
    } else {
        const int tot = nrow * Kb;
        const fdiv by_kb(Kb);
#pragma unroll 4
        for (int e = tid; e < tot; e += nthr) {
            const int r = by_kb(e), c = e - r * Kb;
            wl[r * Kbp + c] = w[(int64_t)(co0 + r) * K + koff + c];
        }
    }
}
__device__ __forceinline__ int padded_row(int Kb) { return ((Kb + 3) & ~3) + 4; }  // +4: neighbouring rows on different banks

// Partial convolution of the block's activated tile act[cpg][H*W] (LDS) into slab `g` of y ([groups][Cout][N*HoWo]) on
// the matrix cores: out[co][px] = sum_k W[co][k] A[k][px] with k = (tap, channel of the group), 32 co x 32 px tiles of
// v_mfma_f32_32x32x2_f32 (the two k slots = a channel pair), one tile per wave.  A operand: the staged weight slice
// wl[row][Kbp] (LDS), B operand: the tile gathered through per-lane tap addresses (zero padding and pixels past the
// map read the block's zero word with channel stride 0 - no conditional load).  When a block has fewer than 8 tiles
// its waves split the channel pairs of a tile between them and the partial tiles are summed through LDS (`part`).
template <int KSZ>
__device__ __forceinline__ void partial_conv(const float* act, int cpg, int H, int W, const float* __restrict__ w, int C,
                                             int c0, int co_beg, int co_end, int stride, int pad, int Ho, int Wo,
                                             float* __restrict__ yslab, int64_t NHWo, int n, float* wl, int wl_floats,
                                             bool prestaged, int zero_off, float* part, int pad_h = -1) {
    if (pad_h < 0) pad_h = pad;  // (k_nconv's strips carry their halo rows: vertical padding 0, horizontal as the conv's)
    constexpr int KK = KSZ * KSZ;
    const int HWo = Ho * Wo, HW = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int K = C * KK, Kb = cpg * KK, Kbp = padded_row(Kb), hc = cpg >> 1;
    const int ptl = (HWo + 31) >> 5;  // pixel tiles
    int rows_max = co_end - co_beg;  // prestaged: the whole slice is resident, one chunk
    if (!prestaged) {
        rows_max = fdiv(Kbp)(wl_floats);
        if (rows_max > 32) rows_max &= ~31;
    }
    if (rows_max <= 0) return;  // (block-uniform: this blockIdx.z has no channels of this conv)
    const fdiv by_ptl(ptl), by_wo(Wo);
    for (int cb0 = co_beg; cb0 < co_end; cb0 += rows_max) {
        const int nrow = min(rows_max, co_end - cb0);
        if (!prestaged) {  // (prestaged: the whole slice was loaded next to the tile, before the block's barriers)
            __syncthreads();  // the tile is complete / the previous chunk has been consumed
            stage_weights(w, K, c0 * KK, Kb, Kbp, cb0, nrow, wl, 0, GT);
            __syncthreads();
        }
        const int T = ((nrow + 31) >> 5) * ptl;
        int ksl = (T >= 8 || !part) ? 0 : (T >= 4 ? 1 : (T >= 2 ? 2 : 3));  // log2 of the waves per tile
        while ((1 << ksl) > hc) --ksl;
        const int KS = 1 << ksl;
        for (int item = wave; item < T * KS; item += NW) {
            const int tile = item >> ksl, ks = item - (tile << ksl);
            const int ci = by_ptl(tile), pi = tile - ci * ptl;
            const int p = pi * 32 + l31;
            const bool live = p < HWo;
            const int oh = live ? by_wo(p) : 0, ow = live ? p - oh * Wo : 0;
            const int cp0 = (ks * hc) >> ksl, cp1 = ((ks + 1) * hc) >> ksl;
            const float* wrow = wl + min(ci * 32 + l31, nrow - 1) * Kbp + half * KK;  // rows past the slice repeat its last row
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int ih = oh * stride - pad_h + t / KSZ, iw = ow * stride - pad + t % KSZ;
                const bool in = live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                const float* ap = act + (in ? ih * W + iw + half * HW : zero_off);
                const int cstr = in ? 2 * HW : 0;
                int cp = cp0;
                for (; cp + 4 <= cp1; cp += 4) {  // four channel pairs at a time: eight LDS reads in flight, then four MFMAs
                    float a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = wrow[2 * (cp + u) * KK + t];
                        b[u] = ap[(cp + u) * cstr];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks every read to its MFMA: one LDS latency per step)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (; cp < cp1; ++cp)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[2 * cp * KK + t], ap[cp * cstr], acc, 0, 0, 0);
            }
            // acc[r] -> row (r & 3) + 8 * (r >> 2) + 4 * half, column l31
#ifdef GN_CONV_TIMING
            if (item < 8 && threadIdx.x == (unsigned)wave * 64 && wave == 0) g_stamp[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 6] = wall_clock64();
#endif
            if (KS == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (live && row < nrow) yslab[(int64_t)(cb0 + row) * NHWo + (int64_t)n * HWo + p] = acc[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) part[item * 1024 + r * 64 + lane] = acc[r];
            }
        }
        if (KS > 1) {  // (block-uniform)
            __syncthreads();
            for (int idx = threadIdx.x; idx < T * 1024; idx += GT) {
                const int tile = idx >> 10, e = idx & 1023, r = e >> 6, ln = e & 63;
                float v = part[(tile * KS) * 1024 + e];
                for (int ks = 1; ks < KS; ++ks) v += part[(tile * KS + ks) * 1024 + e];
                const int ci = by_ptl(tile), pi = tile - ci * ptl;
                const int row = ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), p = pi * 32 + (ln & 31);
                if (p < HWo && row < nrow) yslab[(int64_t)(cb0 + row) * NHWo + (int64_t)n * HWo + p] = v;
            }
            __syncthreads();
        }
    }
}

template <int KSZ, bool FRONT>
__global__ __launch_bounds__(GT) void k_gn_conv(const Desc D, const Geo G) {
    const int wthr = G.wthr;  // the last `wthr` threads load this block's weight slices beside the tile (0: chunked)
    const int prestage = wthr > 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = D.H * D.W;
    const int cpg = G.cpg, nel = cpg * HW, nel4 = (nel + 3) & ~3;
    float* red = smem;        // 64: block reductions; red[ZW] stays 0 (the convs' padding word)
    float* gb = smem + 64;    // gamma, beta, gamma2, beta2 of this group's channels
    float* tile = gb + 4 * ((cpg + 3) & ~3);  // cpg*HW
    float* p = tile + nel4;
    float* tile2 = p;         // second operand
    if (D.x2) p += nel4;
    float* tres = p;          // residual
    if (D.residual) p += nel4;
    const int Hc = G.Hc, Wc = G.Wc;  // the convs' input size
    float* pooled = p;
    if (D.pool) p += (cpg * Hc * Wc + 3) & ~3;
    float* part = G.part_floats ? p : nullptr;  // partial tiles of waves that share a tile
    p += G.part_floats;
    // front stage (two conv layers per launch): all channels of the previous layer, its affine parameters, the rows of
    // its 1x1 conv that produce this block's group
    float* act0 = p;
    float* gb0 = p;
    float* w0l = p;
    const int kbp0 = FRONT ? padded_row(D.C0) : 0;
    if (FRONT) {
        gb0 = act0 + ((D.C0 * HW + 3) & ~3);
        w0l = gb0 + ((2 * D.C0 + 3) & ~3);
        p = w0l + cpg * kbp0;
    }
    float* wl = p;            // staged weights: conv A, then (prestage) conv B behind it
    // grid (group, image, slice of the output channels): the slices of one (image, group) share an XCD's L2
    const int g = blockIdx.x, n = blockIdx.y, sy = blockIdx.z;
    const int c0 = g * cpg;
    const int a_beg = sy * G.per_a, a_end = min(D.Cout_a, a_beg + G.per_a), b_beg = sy * G.per_b, b_end = min(D.Cout_b, b_beg + G.per_b);
    const int kbp_a = padded_row(cpg * KSZ * KSZ), kbp_b = padded_row(cpg);
    float* wl_b = wl + (prestage && D.wa ? max(a_end - a_beg, 0) * kbp_a : 0);

    STAMP(0);
    // every global load of the kernel up front, by role: threads [0, t1) the (image, group) tile(s), [t1, t2) the front
    // stage's slabs, [t2, t3) its weight rows and affine parameters, [t3, GT) this block's slices of the next conv(s)
    float s = 0.f, s2 = 0.f;
    const int t3 = GT - wthr, t2 = t3 - G.w0thr, t1 = t2 - G.fthr;
    const int tid = threadIdx.x;
    if (tid >= GT - cpg) {  // affine parameters: fetched with everything else, read from LDS after the statistics
        const int c = tid - (GT - cpg), cp = (cpg + 3) & ~3;
        gb[c] = D.gamma[c0 + c];
        gb[cp + c] = D.beta[c0 + c];
        if (D.x2) {
            gb[2 * cp + c] = D.gamma2[c0 + c];
            gb[3 * cp + c] = D.beta2[c0 + c];
        }
    }
    if (tid == 0) red[ZW] = 0.f;
    if (tid < t1) {
        if ((HW & 3) == 0) load_tiles<4, FRONT>(D, n, c0, cpg, HW, t1, tile, tile2, tres, s, s2);
        else load_tiles<1, FRONT>(D, n, c0, cpg, HW, t1, tile, tile2, tres, s, s2);
    } else if (FRONT && tid < t2) {
        if ((HW & 3) == 0) load_front<4>(D, n, HW, t1, t2 - t1, act0);
        else load_front<1>(D, n, HW, t1, t2 - t1, act0);
    } else if (FRONT && tid < t3) {
        stage_weights(D.w0, D.C0, 0, D.C0, kbp0, c0, cpg, w0l, t2, t3 - t2);
        for (int c = tid - t2; c < D.C0; c += t3 - t2) {
            gb0[c] = D.gamma0[c];
            gb0[D.C0 + c] = D.beta0[c];
        }
    } else {
        if (D.wa) stage_weights(D.wa, D.C * KSZ * KSZ, c0 * KSZ * KSZ, cpg * KSZ * KSZ, kbp_a, a_beg, a_end - a_beg, wl, t3, GT - t3);
        if (D.wb) stage_weights(D.wb, D.C, c0, cpg, kbp_b, b_beg, b_end - b_beg, wl_b, t3, GT - t3);
    }
    if (FRONT) {
        STAMP(1);
        // GroupNorm + ReLU of the whole previous layer (every block of the image repeats it: 16-64 KB of LDS work), then
        // this group's rows of the 1x1 conv over the full K = C0 -> the tile the statistics below see
        __syncthreads();
        norm_front(D, HW, act0, gb0);
        __syncthreads();
        partial_conv<1>(act0, D.C0, D.H, D.W, D.w0, D.C0, 0, c0, c0 + cpg, 1, 0, D.H, D.W, tile - (int64_t)c0 * HW, HW, 0, w0l,
                        cpg * kbp0, true, (int)(red + ZW - act0), part);
        __syncthreads();
        for (int i = tid; i < nel; i += GT) s += tile[i];
        STAMP(7);
    } else {
        STAMP(1);
    }
    // statistics of both operands: two block reductions (means, then centred squares)
    block_sum2(s, s2, red);
    const float mean = s / (float)nel, mean2 = s2 / (float)nel;
    float q = 0.f, q2 = 0.f;
    for (int i = threadIdx.x; i < nel; i += GT) {
        const float d = tile[i] - mean;
        q += d * d;
        if (D.x2) {
            const float d2 = tile2[i] - mean2;
            q2 += d2 * d2;
        }
    }
    block_sum2(q, q2, red);
    const float rstd = rsqrtf(q / (float)nel + D.eps), rstd2 = rsqrtf(q2 / (float)nel + D.eps);
    // normalise (+ second operand) (+ residual) (+ ReLU), in place
    const fdiv by_hw(HW);
    for (int i = threadIdx.x; i < nel; i += GT) {
        const int c = by_hw(i), cp = (cpg + 3) & ~3;
        const float ga = gb[c] * rstd, be = gb[cp + c] - mean * ga;
        float v = fmaf(tile[i], ga, be);
        if (D.x2) {
            const float g2 = gb[2 * cp + c] * rstd2, b2 = gb[3 * cp + c] - mean2 * g2;
            v += fmaf(tile2[i], g2, b2);
        }
        if (D.residual) v += tres[i];
        if (D.relu) v = fmaxf(v, 0.f);
        tile[i] = v;
    }
    __syncthreads();
    // MaxPool2d(3, stride 2, padding 1) of the activated tile (the stem)
    const float* act = tile;
    if (D.pool) {
        const fdiv by_hwc(Hc * Wc), by_wc(Wc);
        for (int i = threadIdx.x; i < cpg * Hc * Wc; i += GT) {
            const int c = by_hwc(i), pp = i - c * Hc * Wc, ho = by_wc(pp), wo = pp - ho * Wc;
            float m = -INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int h = ho * 2 - 1 + a, ww = wo * 2 - 1 + b;
                    if ((unsigned)h < (unsigned)D.H && (unsigned)ww < (unsigned)D.W) m = fmaxf(m, tile[c * HW + h * D.W + ww]);
                }
            pooled[i] = m;
        }
        __syncthreads();
        act = pooled;
    }
    STAMP(2);
    if (D.act_out && sy == 0) {
        float* ap = D.act_out + ((int64_t)n * D.C + c0) * Hc * Wc;
        for (int i = threadIdx.x; i < cpg * Hc * Wc; i += GT) ap[i] = act[i];
    }
    STAMP(3);
    // the next convolution(s): this block's slice of K, output channels [beg, end) of this blockIdx.z
    if (D.wa) {
        const int64_t NHWo = (int64_t)D.N * G.Ho_a * G.Wo_a;
        partial_conv<KSZ>(act, cpg, Hc, Wc, D.wa, D.C, c0, a_beg, a_end, D.stride_a, D.pad_a, G.Ho_a, G.Wo_a,
                          D.ya + (int64_t)g * D.Cout_a * NHWo, NHWo, n, wl, G.wl_floats, prestage != 0, (int)(red + ZW - act), part);
    }
    STAMP(4);
    if (D.wb) {
        const int64_t NHWo = (int64_t)D.N * G.Ho_b * G.Wo_b;
        partial_conv<1>(act, cpg, Hc, Wc, D.wb, D.C, c0, b_beg, b_end, D.stride_b, 0, G.Ho_b, G.Wo_b,
                        D.yb + (int64_t)g * D.Cout_b * NHWo, NHWo, n, wl_b, G.wl_floats, prestage != 0, (int)(red + ZW - act), part);
    }
    STAMP(5);
}


// ------------------------------------------------------------------------------------------------------------------
// k_nconv: convolution with GroupNorm applied to its input ON LOAD and the GroupNorm statistics of its output emitted as
// per-workgroup partials (ivln_nconv_f32).  For the large maps of layer 1: a workgroup owns a strip of output rows of
// one image (with its halo rows) over ALL input channels, so its outputs are complete - no slabs - and the launch
// chain is still one launch per conv layer.  The statistics the consumer needs are the (count, mean, M2) partials its
// producer left per (strip, image, group), merged in strip order with Chan's formula (a two-pass variance).
// ------------------------------------------------------------------------------------------------------------------
typedef ivln_nconv_desc NDesc;
struct NGeo {
    int RS, strips, Hs;      // output rows (of conv A) per workgroup, strips per image, staged input rows
    int part_floats, wthr;   // partial-tile scratch of partial_conv; threads that stage the weights
    int kbp_a, kbp_b;
    int sa, sb;              // strides
    int Ho_a, Wo_a, Ho_b, Wo_b, RSb;  // output sizes; conv B's output rows per workgroup
    int per_a, per_b;        // output channels per blockIdx.z
};
constexpr int NC_E4 = 6;     // float4 of the strip per loader thread (registers)

// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float row_sum16(float v) {
#define IVLN_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    IVLN_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    IVLN_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    IVLN_DPP_ADD(0x141);  // row_half_mirror
    IVLN_DPP_ADD(0x140);  // row_mirror
#undef IVLN_DPP_ADD
    return v;
}

// (count, mean, M2) partials of one (image, group) -> mean, rstd.  Two loops in part order, no division inside:
// mean = sum(n_p mean_p) / n, M2 = sum(M2_p + n_p (mean_p - mean)^2) - Chan's merge written as the two-pass formula
// it is equal to.  st: this image's [parts][groups][3] in LDS.
__device__ __forceinline__ void merge_stats(const float* st, int parts, int groups, int g, float eps, float& mean, float& rstd) {
    float cnt = 0.f, sm = 0.f;
    for (int p = 0; p < parts; ++p) {
        const float* q = st + (p * groups + g) * 3;
        cnt += q[0];
        sm = fmaf(q[0], q[1], sm);
    }
    const float m = sm / cnt;
    float M2 = 0.f;
    for (int p = 0; p < parts; ++p) {
        const float* q = st + (p * groups + g) * 3;
        const float d = q[1] - m;
        M2 += fmaf(q[0] * d, d, q[2]);
    }
    mean = m;
    rstd = rsqrtf(M2 / cnt + eps);
}

template <int KSZ>
__global__ __launch_bounds__(GT) void k_nconv(const NDesc D, const NGeo G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PH = KSZ / 2;  // halo rows above / below, horizontal padding
    const int C = D.C, W = D.W, HW = D.H * D.W, Cp = (C + 3) & ~3;
    const int strip = blockIdx.x, n = blockIdx.y, mz = blockIdx.z;
    const int r0 = strip * G.RS, rows = min(G.RS, G.Ho_a - r0);  // output rows of conv A
    const int rin0 = r0 * G.sa - PH;                             // first staged input row
    const int a_beg = mz * G.per_a, a_end = min(D.Cout_a, a_beg + G.per_a), na = max(a_end - a_beg, 0);
    const int b_beg = mz * G.per_b, b_end = min(D.Cout_b, b_beg + G.per_b), nb = D.wb ? max(b_end - b_beg, 0) : 0;
    const int rb0 = r0 * G.sa / max(G.sb, 1), rows_b = D.wb ? min(G.RSb, G.Ho_b - rb0) : 0;
    const int SW = G.Hs * W;     // floats of one channel of the staged strip
    float* red = smem;           // 64 (zero word at ZW)
    float* pst = red + 64;       // this image's statistics partials [parts][groups][3] of x, then of x2
    const int pfl = D.stats ? ((D.parts * D.groups * 3 + 3) & ~3) : 0, pfl2 = D.x2 ? ((D.parts2 * D.groups * 3 + 3) & ~3) : 0;
    float* gst = pst + pfl + pfl2;  // mean, rstd per group of x, then of x2
    float* gab = gst + 4 * ((D.groups + 3) & ~3);  // gamma, beta, gamma2, beta2
    float* tab = gab + 4 * Cp;   // scale, shift, scale2, shift2 per channel
    float* in = tab + 4 * Cp;    // [C][Hs*W] transformed input strip
    float* outa = in + ((C * SW + 3) & ~3);
    const int ta = G.RS * G.Wo_a, tb = G.RSb * G.Wo_b;  // floats per output channel of the LDS output tiles
    float* outb = outa + ((G.per_a * ta + 3) & ~3);
    float* part = outb + (D.wb ? ((G.per_b * tb + 3) & ~3) : 0);
    float* wl = part + G.part_floats;
    float* wl_b = wl + G.per_a * G.kbp_a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nload = GT - G.wthr;

    STAMP(0);
    // ---- every global load up front: statistics partials, the raw strip (registers), weights, affine ----
    if (tid == 0) red[ZW] = 0.f;
    if (D.stats) {  // [parts][N][groups][3] -> this image's rows, all in flight at once
        const int row = D.groups * 3;
        const fdiv by_row(row);
        for (int i = tid; i < D.parts * row; i += GT) {
            const int pp = by_row(i);
            pst[i] = D.stats[((int64_t)pp * D.N + n) * row + (i - pp * row)];
        }
        if (D.x2)
            for (int i = tid; i < D.parts2 * row; i += GT) {
                const int pp = by_row(i);
                pst[pfl + i] = D.stats2[((int64_t)pp * D.N + n) * row + (i - pp * row)];
            }
    }
    float4 xr[NC_E4], x2r[NC_E4], rr[NC_E4];
    const int w4 = W >> 2, nv = C * G.Hs * w4;  // float4 of the strip
    if (tid < nload) {
        const fdiv by_w4(w4), by_hs(G.Hs);
#pragma unroll
        for (int e = 0; e < NC_E4; ++e) {
            const int v = tid + e * nload;
            xr[e] = x2r[e] = rr[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < nv) {
                const int ch = by_w4(v), q = v - ch * w4, c = by_hs(ch), hs = ch - c * G.Hs;  // v = (c*Hs + hs)*w4 + q
                const int row = rin0 + hs;
                if ((unsigned)row < (unsigned)D.H) {
                    // raw conv outputs are channel-major over the batch ([C][N][H][W], the slab layout of ivln_gemm_f32 /
                    // ivln_gn_conv_f32 with one slab); activations (plain input, residual, act_out) are NCHW
                    const int64_t o = ((int64_t)n * C + c) * HW + (int64_t)row * W + 4 * q;
                    const int64_t oc = ((int64_t)c * D.N + n) * HW + (int64_t)row * W + 4 * q;
                    xr[e] = *reinterpret_cast<const float4*>(D.x + (D.stats ? oc : o));
                    if (D.x2) x2r[e] = *reinterpret_cast<const float4*>(D.x2 + oc);
                    if (D.residual) rr[e] = *reinterpret_cast<const float4*>(D.residual + o);
                }
            }
        }
    } else {
        const int t0 = nload, nt = GT - nload;
        if (D.stats)
            for (int c = tid - t0; c < C; c += nt) {
                gab[c] = D.gamma[c];
                gab[Cp + c] = D.beta[c];
                if (D.x2) {
                    gab[2 * Cp + c] = D.gamma2[c];
                    gab[3 * Cp + c] = D.beta2[c];
                }
            }
        stage_weights(D.wa, C * KSZ * KSZ, 0, C * KSZ * KSZ, G.kbp_a, a_beg, na, wl, t0, nt);
        if (nb) stage_weights(D.wb, C, 0, C, G.kbp_b, b_beg, nb, wl_b, t0, nt);
    }
    __syncthreads();
    STAMP(1);
    // ---- per-channel scale / shift of this image ----
    if (D.stats) {
        // merge the partials: a DPP row of 16 lanes per group (lane = part) when there are at most 16 parts - no loop -,
        // threads [0, 256) for x and [256, 512) for x2; otherwise one thread per group walks its parts.  Every lane of
        // the row ends up with the group's mean / rstd, so the row writes its channels' scale / shift straight away.
        const int half_t = tid & 255, which = tid >> 8, cpg = C / D.groups;
        if (which == 0 || D.x2) {
            const float* ps = which ? pst + pfl : pst;
            const int np_ = which ? D.parts2 : D.parts;
            float* sc_o = tab + which * 2 * Cp;
            const float* ga_i = gab + which * 2 * Cp;
            if (np_ <= 16 && D.groups <= 16) {
                const int g = half_t >> 4, pp = half_t & 15;
                const bool have = g < D.groups && pp < np_;
                const float* q = ps + (pp * D.groups + (g < D.groups ? g : 0)) * 3;
                const float nb_ = have ? q[0] : 0.f, mb = have ? q[1] : 0.f, Mb = have ? q[2] : 0.f;
                const float cnt = row_sum16(nb_), mean = row_sum16(nb_ * mb) / cnt;
                const float dd = mb - mean;
                const float rstd = rsqrtf(row_sum16(fmaf(nb_ * dd, dd, Mb)) / cnt + D.eps);
                if (g < D.groups)
                    for (int cc = pp; cc < cpg; cc += 16) {
                        const int c = g * cpg + cc;
                        const float sc = ga_i[c] * rstd;
                        sc_o[c] = sc;
                        sc_o[Cp + c] = ga_i[Cp + c] - mean * sc;
                    }
            } else if (half_t < D.groups) {
                float mean, rstd;
                merge_stats(ps, np_, D.groups, half_t, D.eps, mean, rstd);
                for (int cc = 0; cc < cpg; ++cc) {
                    const int c = half_t * cpg + cc;
                    const float sc = ga_i[c] * rstd;
                    sc_o[c] = sc;
                    sc_o[Cp + c] = ga_i[Cp + c] - mean * sc;
                }
            }
        }
        __syncthreads();
    }
    // ---- transform the strip into LDS (rows outside the image stay 0), hand out the activation ----
    if (tid < nload) {
        const fdiv by_w4(w4), by_hs(G.Hs);
#pragma unroll
        for (int e = 0; e < NC_E4; ++e) {
            const int v = tid + e * nload;
            if (v < nv) {
                const int ch = by_w4(v), q = v - ch * w4, c = by_hs(ch), hs = ch - c * G.Hs;
                const int row = rin0 + hs;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)row < (unsigned)D.H) {
                    a = xr[e];
                    if (D.stats) {
                        const float sc = tab[c], sh = tab[Cp + c];
                        a.x = fmaf(a.x, sc, sh); a.y = fmaf(a.y, sc, sh); a.z = fmaf(a.z, sc, sh); a.w = fmaf(a.w, sc, sh);
                        if (D.x2) {
                            const float s2 = tab[2 * Cp + c], h2 = tab[3 * Cp + c];
                            a.x += fmaf(x2r[e].x, s2, h2); a.y += fmaf(x2r[e].y, s2, h2);
                            a.z += fmaf(x2r[e].z, s2, h2); a.w += fmaf(x2r[e].w, s2, h2);
                        }
                    }
                    if (D.residual) { a.x += rr[e].x; a.y += rr[e].y; a.z += rr[e].z; a.w += rr[e].w; }
                    if (D.relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                    if (D.act_out && mz == 0 && hs >= PH && hs < PH + rows)  // (stride 1 only: host)
                        *reinterpret_cast<float4*>(D.act_out + ((int64_t)n * C + c) * HW + (int64_t)row * W + 4 * q) = a;
                }
                *reinterpret_cast<float4*>(&in[(c * G.Hs + hs) * W + 4 * q]) = a;
            }
        }
    }
    __syncthreads();
    STAMP(2);
    // ---- the convolution(s) over the full K: complete output tiles in LDS ----
    const int zoff = (int)(red + ZW - in);
    // (yslab offsets: partial_conv addresses rows by their global index cb0 + row)
    if (na)
        partial_conv<KSZ>(in, C, G.Hs, W, D.wa, C, 0, a_beg, a_end, G.sa, PH, rows, G.Wo_a, outa - (int64_t)a_beg * ta, ta, 0, wl,
                          na * G.kbp_a, true, zoff, G.part_floats ? part : nullptr, 0);
    if (nb) {
        // conv B is 1x1 (stride sb): it reads the strip's own rows only
        partial_conv<1>(in + PH * W, C, G.Hs, W, D.wb, C, 0, b_beg, b_end, G.sb, 0, rows_b, G.Wo_b, outb - (int64_t)b_beg * tb, tb, 0,
                        wl_b, nb * G.kbp_b, true, (int)(red + ZW - (in + PH * W)), G.part_floats ? part : nullptr, 0);
    }
    __syncthreads();
    STAMP(3);
    // ---- outputs to memory (whole rows) and the statistics partials of this strip ----
    for (int pass = 0; pass < (D.wb ? 2 : 1); ++pass) {
        const float* o = pass ? outb : outa;
        float* y = pass ? D.yb : D.ya;
        const int Co = pass ? nb : na, cbeg = pass ? b_beg : a_beg, ng = pass ? D.groups_b : D.groups_a;
        const int tw = pass ? tb : ta, npx = pass ? rows_b * G.Wo_b : rows * G.Wo_a, npx4 = npx >> 2;
        const int HWo = pass ? G.Ho_b * G.Wo_b : G.Ho_a * G.Wo_a, row0 = pass ? rb0 * G.Wo_b : r0 * G.Wo_a;
        const int Ctot = pass ? D.Cout_b : D.Cout_a;
        float* so = pass ? D.stats_b : D.stats_a;
        if (Co <= 0) continue;
        if (npx <= 0) {  // (a strip below conv B's last output row: an empty partial)
            if (so && tid < Co / (Ctot / ng) * 3) so[((int64_t)(strip * D.N + n) * ng + cbeg / (Ctot / ng)) * 3 + tid] = 0.f;
            continue;
        }
        const fdiv by_px4(npx4);
        for (int v = tid; v < Co * npx4; v += GT) {
            const int co = by_px4(v), q = v - co * npx4;
            *reinterpret_cast<float4*>(y + ((int64_t)(cbeg + co) * D.N + n) * HWo + row0 + 4 * q) =
                *reinterpret_cast<const float4*>(&o[co * tw + 4 * q]);
        }
        if (so) {
            const int cpo = Ctot / ng, nel = cpo * npx, g0 = cbeg / cpo;
            for (int g = wave; g < Co / cpo; g += NW) {  // one wave per group, ONE pass over its cpo x npx values, shifted by
                // the group's first value so that sum-of-squares minus squared-sum does not cancel
                const float* og = o + g * cpo * tw;
                const float pilot = og[0];
                float s1 = 0.f, s2 = 0.f;
                const fdiv by_npx(npx);
                for (int i = lane; i < nel; i += 64) {
                    const int cl = by_npx(i);
                    const float dd = og[cl * tw + (i - cl * npx)] - pilot;
                    s1 += dd;
                    s2 = fmaf(dd, dd, s2);
                }
                s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum(s1)), 63));
                s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum(s2)), 63));
                const float mean = pilot + s1 / (float)nel;
                const float M2 = fmaxf(s2 - s1 * s1 / (float)nel, 0.f);
                if (lane == 0) {
                    float* dst = so + ((int64_t)(strip * D.N + n) * ng + g0 + g) * 3;
                    dst[0] = (float)nel;
                    dst[1] = mean;
                    dst[2] = M2;
                }
            }
        }
    }
    STAMP(4);
}

typedef void (*gn_conv_fn)(const Desc, const Geo);

}  // namespace

extern "C" {

int ivln_gn_conv_f32(const ivln_gn_conv_desc* d, void* stream) {
    if (!d || (!d->x && !d->x0) || !d->gamma || !d->beta || d->N <= 0 || d->C <= 0 || d->groups <= 0 || d->C % d->groups) return IVLN_E_INVALID;
    if ((d->x && d->splits < 1) || (d->x2 && (d->splits2 < 1 || !d->gamma2 || !d->beta2))) return IVLN_E_INVALID;
    if (!d->wa && !d->wb && !d->act_out) return IVLN_E_INVALID;
    if (d->x0 && (d->x || d->pool || d->splits0 < 1 || d->C0 <= 0 || d->groups0 <= 0 || d->C0 % d->groups0 || (d->C0 & 1) || !d->gamma0 ||
                  !d->beta0 || !d->w0))
        return IVLN_E_INVALID;
    if (d->wa && (!d->ya || d->Cout_a <= 0 || (d->ka != 1 && d->ka != 3) || d->stride_a < 1)) return IVLN_E_UNSUPPORTED;
    if (d->wb && (!d->yb || d->Cout_b <= 0 || d->stride_b < 1)) return IVLN_E_INVALID;
    const int cpg = d->C / d->groups, HW = d->H * d->W, nel = cpg * HW;
    if (nel > TILE_MAX || cpg > GT / 2) return IVLN_E_UNSUPPORTED;
    int H = d->H, W = d->W;
    if (d->pool) {
        H = (H + 2 - 3) / 2 + 1;
        W = (W + 2 - 3) / 2 + 1;
    }
    Geo G = {};
    G.cpg = cpg;
    G.Hc = H;
    G.Wc = W;
    int HWo_a = 0, HWo_b = 0;
    if (d->wa) {
        G.Ho_a = (H + 2 * d->pad_a - d->ka) / d->stride_a + 1;
        G.Wo_a = (W + 2 * d->pad_a - d->ka) / d->stride_a + 1;
        if (G.Ho_a <= 0 || G.Wo_a <= 0) return IVLN_E_INVALID;
        HWo_a = G.Ho_a * G.Wo_a;
    }
    if (d->wb) {
        G.Ho_b = (H - 1) / d->stride_b + 1;
        G.Wo_b = (W - 1) / d->stride_b + 1;
        HWo_b = G.Ho_b * G.Wo_b;
    }
    if ((d->wa || d->wb) && (cpg & 1)) return IVLN_E_UNSUPPORTED;  // the MFMA's two k slots take a channel pair
    const bool k3 = d->wa && d->ka == 3;
    gn_conv_fn fn = d->x0 ? (k3 ? k_gn_conv<3, true> : k_gn_conv<1, true>) : (k3 ? k_gn_conv<3, false> : k_gn_conv<1, false>);
    // output channels over blockIdx.y so that ~256 blocks are in flight (every block repeats the cheap GroupNorm)
    int S = 1;
    const int blocks = d->N * d->groups;
    const int cmin = d->wa ? (d->wb && d->Cout_b < d->Cout_a ? d->Cout_b : d->Cout_a) : (d->wb ? d->Cout_b : 1);
    constexpr int s_cap = 4;
    while (S < s_cap && blocks * S * 2 <= 256 && cmin / (S * 2) >= 16) S *= 2;
    // LDS: reduction scratch + tile (+ second operand) (+ residual) (+ pooled tile) + the staged weight slices.  When
    // both convs' slices fit they are loaded up front beside the tile (prestage); otherwise each conv streams its
    // slice through the remaining space in chunks of rows.
    const size_t nel4 = (size_t)((nel + 3) & ~3);
    // waves of a block with fewer than 8 output tiles (32 channels x 32 pixels) share tiles: 8 partial tiles in LDS
    const int tiles_a = d->wa ? (((d->Cout_a + S - 1) / S + 31) / 32) * ((HWo_a + 31) / 32) : 8;
    const int tiles_b = d->wb ? (((d->Cout_b + S - 1) / S + 31) / 32) * ((HWo_b + 31) / 32) : 8;
    size_t part = (tiles_a < 8 || tiles_b < 8) ? 8 * 1024 : 0;
    size_t fixed = 64 + 4 * (size_t)((cpg + 3) & ~3) + nel4 * (1 + (d->x2 ? 1 : 0) + (d->residual ? 1 : 0)) +
                   (d->pool ? (size_t)((cpg * H * W + 3) & ~3) : 0);
    size_t front = 0;
    if (d->x0) {  // all channels of the previous layer + its affine parameters + this group's rows of its 1x1 conv
        front = (size_t)((d->C0 * HW + 3) & ~3) + (size_t)((2 * d->C0 + 3) & ~3) + (size_t)cpg * (((d->C0 + 3) & ~3) + 4);
        if (((cpg + 31) / 32) * ((HW + 31) / 32) < 8) part = 8 * 1024;
        fixed += front;
    }
    if (fixed + part + 2048 > kLdsFloats) part = 0;
    fixed += part;
    if (fixed + 64 > kLdsFloats) return IVLN_E_UNSUPPORTED;
    const size_t kbp_a = d->wa ? (size_t)((cpg * d->ka * d->ka + 3) & ~3) + 4 : 0, kbp_b = d->wb ? (size_t)((cpg + 3) & ~3) + 4 : 0;
    const size_t need_a = d->wa ? (size_t)((d->Cout_a + S - 1) / S) * kbp_a : 0, need_b = d->wb ? (size_t)((d->Cout_b + S - 1) / S) * kbp_b : 0;
    const size_t cap = kLdsFloats - fixed;
    const int prestage = need_a + need_b <= cap;
    const size_t wl = prestage ? need_a + need_b : cap;
    if (wl < kbp_a || wl < kbp_b) return IVLN_E_UNSUPPORTED;
    const size_t bytes = sizeof(float) * (fixed + wl);
    {
        static std::mutex mu;
        static std::set<const void*> raised;
        std::lock_guard<std::mutex> lk(mu);
        if (!raised.count((const void*)fn)) {
            if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsFloats * sizeof(float))) !=
                hipSuccess)
                return IVLN_E_HIP;
            raised.insert((const void*)fn);
        }
    }
    // the block's 8 waves take load roles in proportion to the bytes: tile slabs | front-stage slabs | front-stage weights |
    // the next convs' weight slices (none when those are streamed in chunks)
    {
        const double tb = (double)nel * ((d->x ? d->splits : 0) + (d->x2 ? d->splits2 : 0) + (d->residual ? 1 : 0));
        const double fb = d->x0 ? (double)d->C0 * HW * d->splits0 : 0.0;
        const double w0b = d->x0 ? (double)cpg * d->C0 + 2.0 * d->C0 : 0.0;
        const double wb = prestage ? (double)(need_a + need_b) : 0.0;
        const double tot = tb + fb + w0b + wb;
        int nw[4];
        const double by[4] = {tb, fb, w0b, wb};
        int used = 0;
        for (int i = 0; i < 4; ++i) {
            nw[i] = by[i] > 0 ? (int)((double)NW * by[i] / tot + 0.5) : 0;
            if (by[i] > 0 && nw[i] < 1) nw[i] = 1;
            used += nw[i];
        }
        while (used > NW) {  // take from the largest
            int m = 0;
            for (int i = 1; i < 4; ++i) if (nw[i] > nw[m]) m = i;
            --nw[m];
            --used;
        }
        int big = tb >= fb ? 0 : 1;  // spare waves go to the larger slab role
        if (by[big] <= 0) big = 3;
        nw[big] += NW - used;
        G.wthr = 64 * nw[3];
        G.w0thr = 64 * nw[2];
        G.fthr = 64 * nw[1];
        if (by[big] <= 0) return IVLN_E_INVALID;
    }
    G.per_a = d->wa ? (d->Cout_a + S - 1) / S : 0;
    G.per_b = d->wb ? (d->Cout_b + S - 1) / S : 0;
    G.wl_floats = (int)wl;
    G.part_floats = (int)part;
    if (d->N > 65535) return IVLN_E_UNSUPPORTED;
    IVLN_LAUNCH_FAMILY_NAMED("k_gn_conv", fn, dim3(d->groups, d->N, S), dim3(GT), bytes, (hipStream_t)stream, *d, G);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_nconv_f32(const ivln_nconv_desc* d, void* stream) {
    if (!d || !d->x || !d->wa || !d->ya || d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->Cout_a <= 0) return IVLN_E_INVALID;
    if (d->stats && (!d->gamma || !d->beta || d->parts < 1 || d->groups <= 0 || d->C % d->groups)) return IVLN_E_INVALID;
    if (d->x2 && (!d->stats || !d->stats2 || !d->gamma2 || !d->beta2 || d->parts2 < 1)) return IVLN_E_INVALID;
    if (d->stats_a && (d->groups_a <= 0 || d->Cout_a % d->groups_a)) return IVLN_E_INVALID;
    if (d->wb && (!d->yb || d->Cout_b <= 0 || (d->stats_b && (d->groups_b <= 0 || d->Cout_b % d->groups_b)))) return IVLN_E_INVALID;
    const int sa = d->stride_a > 0 ? d->stride_a : 1, sb = d->stride_b > 0 ? d->stride_b : 1;
    if ((d->ka != 1 && d->ka != 3) || (d->W & 3) || (d->C & 1) || d->groups > 32 || d->N > 65535 || sa > 2 || sb > 2) return IVLN_E_UNSUPPORTED;
    if (d->act_out && sa != 1) return IVLN_E_UNSUPPORTED;
    if (d->wb && sa != 1) return IVLN_E_UNSUPPORTED;  // conv B shares conv A's input strip: only beside a stride-1 conv A
    NGeo G = {};
    const int ph = d->ka / 2;
    G.sa = sa;
    G.sb = sb;
    G.Ho_a = (d->H + 2 * ph - d->ka) / sa + 1;
    G.Wo_a = (d->W + 2 * ph - d->ka) / sa + 1;
    G.Ho_b = d->wb ? (d->H - 1) / sb + 1 : 0;
    G.Wo_b = d->wb ? (d->W - 1) / sb + 1 : 0;
    if ((G.Wo_a & 3) || (d->wb && (G.Wo_b & 3))) return IVLN_E_UNSUPPORTED;
    const int Cp = (d->C + 3) & ~3;
    const int cpo_a = d->stats_a ? d->Cout_a / d->groups_a : 1, cpo_b = (d->wb && d->stats_b) ? d->Cout_b / d->groups_b : 1;
    const size_t pfl = (d->stats ? (size_t)((d->parts * d->groups * 3 + 3) & ~3) : 0) + (d->x2 ? (size_t)((d->parts2 * d->groups * 3 + 3) & ~3) : 0);
    G.kbp_a = (((d->C * d->ka * d->ka) + 3) & ~3) + 4;
    G.kbp_b = d->wb ? ((d->C + 3) & ~3) + 4 : 0;
    int mz = 1;
    size_t fl = 0, strip = 0;
    // rows per workgroup: as asked (the consumer counts on the strips it was told), else 64 output pixels, halved until
    // the strip (registers of the loader threads, LDS) and the weight slice fit
    int rs = d->rows_per_block > 0 ? d->rows_per_block : (G.Wo_a >= 64 ? 1 : 64 / G.Wo_a);
    for (bool ok = false; !ok; rs /= 2) {
        if (rs < 1) return IVLN_E_UNSUPPORTED;
        G.RS = rs > G.Ho_a ? G.Ho_a : rs;
        if (d->wb && (G.RS % sb)) G.RS = (G.RS + sb - 1) / sb * sb;  // conv B's rows start on its stride
        G.RSb = d->wb ? G.RS / sb : 0;
        G.strips = (G.Ho_a + G.RS - 1) / G.RS;
        G.Hs = (G.RS - 1) * sa + d->ka;
        strip = (size_t)d->C * G.Hs * d->W;
        const size_t base = 64 + pfl + 4 * (size_t)((d->groups + 3) & ~3) + 8 * (size_t)Cp + ((strip + 3) & ~3);
        // output channels over blockIdx.z until weights + output tiles fit (slices keep whole GroupNorm groups and whole
        // 32-row MFMA tiles), and further while the grid is small
        for (mz = 1;; mz *= 2) {
            G.per_a = (d->Cout_a + mz - 1) / mz;
            G.per_b = d->wb ? (d->Cout_b + mz - 1) / mz : 0;
            const bool whole = (mz == 1) || (G.per_a % cpo_a == 0 && G.per_a % 32 == 0 && d->Cout_a % G.per_a == 0 &&
                                             (!d->wb || (G.per_b % cpo_b == 0 && G.per_b % 32 == 0 && d->Cout_b % G.per_b == 0)));
            if (!whole) break;
            const int tiles_a = ((G.per_a + 31) / 32) * ((G.RS * G.Wo_a + 31) / 32);
            const int tiles_b = d->wb ? ((G.per_b + 31) / 32) * ((G.RSb * G.Wo_b + 31) / 32) : NW;
            G.part_floats = (tiles_a < NW || tiles_b < NW) ? 8 * 1024 : 0;
            fl = base + (((size_t)G.per_a * G.RS * G.Wo_a + 3) & ~3) + (d->wb ? (((size_t)G.per_b * G.RSb * G.Wo_b + 3) & ~3) : 0) +
                 (size_t)G.per_a * G.kbp_a + (d->wb ? (size_t)G.per_b * G.kbp_b : 0);
            if (fl + G.part_floats > kLdsFloats && fl <= kLdsFloats) G.part_floats = 0;
            const bool fits = fl + G.part_floats <= kLdsFloats;
            const bool more = (int64_t)G.strips * d->N * mz * 2 <= 256 && G.per_a >= 64 && G.per_a % 64 == 0 && (!d->wb || G.per_b % 64 == 0);
            if (fits && !more) {
                ok = true;
                break;
            }
            if (mz >= 16) {
                ok = fits;
                break;
            }
        }
        // loader threads keep their share of the strip in registers (NC_E4 float4 each)
        if (ok && (size_t)(GT - 64) * NC_E4 * 4 < strip) ok = false;
        if (!ok && d->rows_per_block > 0) return IVLN_E_UNSUPPORTED;
    }
    fl += G.part_floats;
    const size_t wfl = (size_t)G.per_a * G.kbp_a + (d->wb ? (size_t)G.per_b * G.kbp_b : 0);
    int nw = (int)((double)NW * wfl / (double)(wfl + strip * (1 + (d->x2 ? 1 : 0) + (d->residual ? 1 : 0))) + 0.5);
    nw = nw < 1 ? 1 : (nw > NW - 2 ? NW - 2 : nw);
    while (nw > 1 && (size_t)(GT - 64 * nw) * NC_E4 * 4 < strip) --nw;
    if ((size_t)(GT - 64 * nw) * NC_E4 * 4 < strip) return IVLN_E_UNSUPPORTED;
    G.wthr = 64 * nw;
    typedef void (*nconv_fn)(const NDesc, const NGeo);
    nconv_fn fn = d->ka == 3 ? k_nconv<3> : k_nconv<1>;
    {
        static std::mutex mu;
        static std::set<const void*> raised;
        std::lock_guard<std::mutex> lk(mu);
        if (!raised.count((const void*)fn)) {
            if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsFloats * sizeof(float))) !=
                hipSuccess)
                return IVLN_E_HIP;
            raised.insert((const void*)fn);
        }
    }
    IVLN_LAUNCH_FAMILY_NAMED("k_nconv", fn, dim3(G.strips, d->N, mz), dim3(GT), fl * sizeof(float), (hipStream_t)stream, *d, G);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

#ifdef GN_CONV_TIMING
int ivln_gn_conv_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
