// Non-GEMM forward kernels of the MapCMA hot path for gfx950 (all fp32, NCHW):
// GroupNorm(+residual,+ReLU), BatchNorm statistics / folding, pooling, one-hot map features,
// embedding + lengths, bidirectional LSTM recurrence, masked GRU step, skinny (B<=8 rows)
// linear, cross-modal attention, arg-max heads.  Each entry point cites the reference op it
// replaces.  These are HBM/L2-bound or latency-bound element/reduction kernels: 64-wide wave
// reductions via DPP shuffles, LDS for per-block staging, coalesced NCHW row accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/ivln_hip.h"
#include "gru_seq.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block-wide sum for blockDim.x <= 1024 (red: >= 16 floats of LDS)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int i = 0; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// GroupNorm (+ residual) (+ ReLU): one 512-thread block per (image, group), float4 accesses.
// The input is either an NCHW tensor (chan_stride = HW) or the split-K workspace of the producing
// conv, `splits` slabs of a [C][N*HW] matrix (chan_stride = N*HW, img_stride = HW): the deterministic
// slab reduction is fused here, so the conv's raw output is never materialised and no reduce kernel
// is launched.  Two-pass mean / biased variance over values cached in LDS (re-read when they do not
// fit), then y = (x-mean)*rstd*gamma[c] + beta[c] (+ residual) (ReLU).
// ------------------------------------------------------------------------------------------
constexpr int GN_CACHE = 8192;
constexpr int GN_THREADS = 512;

__device__ __forceinline__ float4 gn_load4(const float* __restrict__ xp, int i, int HW, int64_t chan_stride,
                                           int splits, int64_t slab_stride) {
    int cl = i / HW, pp = i - cl * HW;
    const float* p = xp + (int64_t)cl * chan_stride + pp;
    float4 v = *reinterpret_cast<const float4*>(p);
    int z = 1;
    for (; z + 3 < splits; z += 4) {  // 4 independent loads in flight
        float4 w0 = *reinterpret_cast<const float4*>(p + (int64_t)z * slab_stride);
        float4 w1 = *reinterpret_cast<const float4*>(p + (int64_t)(z + 1) * slab_stride);
        float4 w2 = *reinterpret_cast<const float4*>(p + (int64_t)(z + 2) * slab_stride);
        float4 w3 = *reinterpret_cast<const float4*>(p + (int64_t)(z + 3) * slab_stride);
        v.x += (w0.x + w1.x) + (w2.x + w3.x);
        v.y += (w0.y + w1.y) + (w2.y + w3.y);
        v.z += (w0.z + w1.z) + (w2.z + w3.z);
        v.w += (w0.w + w1.w) + (w2.w + w3.w);
    }
    for (; z < splits; ++z) {
        float4 w = *reinterpret_cast<const float4*>(p + (int64_t)z * slab_stride);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}
// slabs z0, z0+zs, z0+2zs, ... of one float4 (cooperative reduction: several threads per element)
__device__ __forceinline__ float4 gn_load4_strided(const float* __restrict__ xp, int i, int HW, int64_t chan_stride,
                                                   int splits, int64_t slab_stride, int z0, int zs) {
    int cl = i / HW, pp = i - cl * HW;
    const float* p = xp + (int64_t)cl * chan_stride + pp;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = z0;
    for (; z + zs < splits; z += 2 * zs) {
        float4 w0 = *reinterpret_cast<const float4*>(p + (int64_t)z * slab_stride);
        float4 w1 = *reinterpret_cast<const float4*>(p + (int64_t)(z + zs) * slab_stride);
        v.x += w0.x + w1.x; v.y += w0.y + w1.y; v.z += w0.z + w1.z; v.w += w0.w + w1.w;
    }
    if (z < splits) {
        float4 w = *reinterpret_cast<const float4*>(p + (int64_t)z * slab_stride);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}
__device__ __forceinline__ float gn_load1(const float* __restrict__ xp, int i, int HW, int64_t chan_stride,
                                          int splits, int64_t slab_stride) {
    int cl = i / HW, pp = i - cl * HW;
    const float* p = xp + (int64_t)cl * chan_stride + pp;
    float v = *p;
    for (int z = 1; z < splits; ++z) v += p[(int64_t)z * slab_stride];
    return v;
}

__global__ __launch_bounds__(GN_THREADS) void k_groupnorm(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, int C, int HW, int groups, float eps, int relu,
    int64_t x_img_stride, int64_t x_chan_stride, int splits, int64_t slab_stride, int64_t y_img_stride,
    int64_t r_img_stride, float* __restrict__ save_mean, float* __restrict__ save_rstd,
    // optional second group-normalised operand added before the ReLU (the bottleneck's downsample branch:
    // y = relu(GN(conv3) + GN_ds(conv_ds)), same channel/group partition): raw slabs of the downsample conv
    const float* __restrict__ x2, const float* __restrict__ gamma2, const float* __restrict__ beta2,
    int64_t x2_img_stride, int64_t x2_chan_stride, int splits2, int64_t slab_stride2) {
    __shared__ __attribute__((aligned(16))) float cache[GN_CACHE];
    __shared__ float red[16];
    const int img = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const int n = cpg * HW;
    const float* xp = x + (int64_t)img * x_img_stride + (int64_t)g * cpg * x_chan_stride;
    const bool cached = n <= GN_CACHE;
    const bool vec = (HW & 3) == 0;
    // affine parameters of this thread's first float4 are fetched now, under the data loads, instead
    // of as a dependent load after the two reductions
    float g_first = 0.f, b_first = 0.f;
    if (vec && (int)threadIdx.x * 4 < n) {
        const int c0 = g * cpg + ((int)threadIdx.x * 4) / HW;
        g_first = gamma[c0];
        b_first = beta[c0];
    }
    float s = 0.f;
    const int n4 = n >> 2;
    if (vec && splits > 1 && n4 * 2 <= GN_THREADS && 2 * n <= GN_CACHE) {
        // small group, many slabs (late ResNet layers: n = 512..2048, up to 64 slabs): several
        // threads share one float4, each summing a strided subset of the slabs; partials meet in LDS
        const int max_slots = (GN_CACHE - n) / n;  // staging room behind the final image (>= 1: 2n <= GN_CACHE)
        const int nz = min(min(GN_THREADS / n4, splits), max_slots + 1);
        const int e = threadIdx.x % n4, zp = threadIdx.x / n4;
        float* part = cache + n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (zp < nz) v = gn_load4_strided(xp, e * 4, HW, x_chan_stride, splits, slab_stride, zp, nz);
        if (zp < nz) {  // slot 0 -> cache, slots 1.. -> part
            if (zp == 0) *reinterpret_cast<float4*>(&cache[e * 4]) = v;
            else *reinterpret_cast<float4*>(&part[(zp - 1) * n + e * 4]) = v;
        }
        __syncthreads();
        if (zp == 0) {
            float4 t = *reinterpret_cast<float4*>(&cache[e * 4]);
            for (int q2 = 0; q2 < nz - 1; ++q2) {
                float4 w = *reinterpret_cast<float4*>(&part[q2 * n + e * 4]);
                t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
            }
            *reinterpret_cast<float4*>(&cache[e * 4]) = t;
            s = (t.x + t.y) + (t.z + t.w);
        }
        __syncthreads();
    } else if (vec) {
        for (int i = threadIdx.x * 4; i < n; i += GN_THREADS * 4) {
            float4 v = gn_load4(xp, i, HW, x_chan_stride, splits, slab_stride);
            if (cached) *reinterpret_cast<float4*>(&cache[i]) = v;
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += GN_THREADS) {
            float v = gn_load1(xp, i, HW, x_chan_stride, splits, slab_stride);
            if (cached) cache[i] = v;
            s += v;
        }
    }
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    if (cached) {
        for (int i = threadIdx.x; i < n; i += GN_THREADS) {
            float d = cache[i] - mean;
            q += d * d;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += GN_THREADS) {
            float d = gn_load1(xp, i, HW, x_chan_stride, splits, slab_stride) - mean;
            q += d * d;
        }
    }
    const float var = block_sum(q, red) / (float)n;
    const float rstd = rsqrtf(var + eps);
    if (threadIdx.x == 0 && save_mean) {
        save_mean[blockIdx.x] = mean;
        save_rstd[blockIdx.x] = rstd;
    }
    // statistics of the second operand (two passes straight from its slabs: it is small)
    const float* x2p = x2 ? x2 + (int64_t)img * x2_img_stride + (int64_t)g * cpg * x2_chan_stride : nullptr;
    float mean2 = 0.f, rstd2 = 0.f;
    if (x2p) {
        float s2 = 0.f;
        if (vec) {
            for (int i = threadIdx.x * 4; i < n; i += GN_THREADS * 4) {
                const float4 v = gn_load4(x2p, i, HW, x2_chan_stride, splits2, slab_stride2);
                s2 += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = threadIdx.x; i < n; i += GN_THREADS) s2 += gn_load1(x2p, i, HW, x2_chan_stride, splits2, slab_stride2);
        }
        mean2 = block_sum(s2, red) / (float)n;
        float q2 = 0.f;
        if (vec) {
            for (int i = threadIdx.x * 4; i < n; i += GN_THREADS * 4) {
                const float4 v = gn_load4(x2p, i, HW, x2_chan_stride, splits2, slab_stride2);
                const float d0 = v.x - mean2, d1 = v.y - mean2, d2 = v.z - mean2, d3 = v.w - mean2;
                q2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        } else {
            for (int i = threadIdx.x; i < n; i += GN_THREADS) {
                const float d = gn_load1(x2p, i, HW, x2_chan_stride, splits2, slab_stride2) - mean2;
                q2 += d * d;
            }
        }
        rstd2 = rsqrtf(block_sum(q2, red) / (float)n + eps);
    }
    float* yp = y + (int64_t)img * y_img_stride + (int64_t)g * n;
    const float* rp = residual ? residual + (int64_t)img * r_img_stride + (int64_t)g * n : nullptr;
    if (vec) {
        for (int i = threadIdx.x * 4; i < n; i += GN_THREADS * 4) {
            int c = g * cpg + i / HW;
            float4 v = cached ? *reinterpret_cast<const float4*>(&cache[i])
                              : gn_load4(xp, i, HW, x_chan_stride, splits, slab_stride);
            const bool first = i == (int)threadIdx.x * 4;
            const float ga = (first ? g_first : gamma[c]) * rstd, be = (first ? b_first : beta[c]) - mean * ga;
            v.x = fmaf(v.x, ga, be); v.y = fmaf(v.y, ga, be); v.z = fmaf(v.z, ga, be); v.w = fmaf(v.w, ga, be);
            if (rp) {
                float4 r = *reinterpret_cast<const float4*>(&rp[i]);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (x2p) {
                const float4 r = gn_load4(x2p, i, HW, x2_chan_stride, splits2, slab_stride2);
                const float g2 = gamma2[c] * rstd2, b2 = beta2[c] - mean2 * g2;
                v.x += fmaf(r.x, g2, b2); v.y += fmaf(r.y, g2, b2); v.z += fmaf(r.z, g2, b2); v.w += fmaf(r.w, g2, b2);
            }
            if (relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(&yp[i]) = v;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += GN_THREADS) {
            int c = g * cpg + i / HW;
            float v = cached ? cache[i] : gn_load1(xp, i, HW, x_chan_stride, splits, slab_stride);
            v = (v - mean) * rstd * gamma[c] + beta[c];
            if (rp) v += rp[i];
            if (x2p)
                v += (gn_load1(x2p, i, HW, x2_chan_stride, splits2, slab_stride2) - mean2) * rstd2 * gamma2[c] + beta2[c];
            if (relu) v = fmaxf(v, 0.f);
            yp[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm2d.  Eval: scale = g/sqrt(rv+eps), shift = b - rm*scale.  Train: batch statistics per
// channel over (N,H,W) (two-pass), running stats updated with momentum / unbiased variance exactly
// like torch.nn.BatchNorm2d, scale/shift from the batch statistics.
// ------------------------------------------------------------------------------------------
__global__ void k_bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta,
                          const float* __restrict__ rmean, const float* __restrict__ rvar,
                          const float* __restrict__ conv_bias, float eps, int C, float* __restrict__ scale,
                          float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    float m = rmean[c] - (conv_bias ? conv_bias[c] : 0.f);  // BN(conv + bias) = conv*sc + (beta - (rm - bias)*sc)
    shift[c] = beta[c] - m * sc;
}

// Train-mode batch statistics in two launches: grid (C, S) blocks each reduce a slice of the images
// with a local two-pass (count, mean, M2); one thread per channel then merges the S partials in a
// fixed order with Chan's parallel-variance formula (deterministic, no cancellation), produces
// scale/shift and updates the running statistics.  (One block per channel took 7.8 ms on the
// (512,32,64,64) update batch.)
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ x, int N, int C, int HW,
                                                          int imgs_per_split, float* __restrict__ partial) {
    __shared__ float red[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int i0 = sp * imgs_per_split, i1 = min(N, i0 + imgs_per_split);
    const int64_t cnt = (int64_t)max(0, i1 - i0) * HW;
    const bool vec = (HW & 3) == 0 && ((uintptr_t)x & 15) == 0;  // 16-byte loads, 4 partial sums per thread
    float s = 0.f;
    if (vec) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int img = i0; img < i1; ++img) {
            const float* xp = x + ((int64_t)img * C + c) * HW;
            for (int i = threadIdx.x * 4; i < HW; i += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(xp + i);
                a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
            }
        }
        s = (a.x + a.y) + (a.z + a.w);
    } else {
        for (int img = i0; img < i1; ++img) {
            const float* xp = x + ((int64_t)img * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
        }
    }
    const float mean = cnt > 0 ? block_sum(s, red) / (float)cnt : 0.f;
    float q = 0.f;
    if (vec) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int img = i0; img < i1; ++img) {
            const float* xp = x + ((int64_t)img * C + c) * HW;
            for (int i = threadIdx.x * 4; i < HW; i += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(xp + i);
                const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                a.x += d0 * d0, a.y += d1 * d1, a.z += d2 * d2, a.w += d3 * d3;
            }
        }
        q = (a.x + a.y) + (a.z + a.w);
    } else {
        for (int img = i0; img < i1; ++img) {
            const float* xp = x + ((int64_t)img * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                float d = xp[i] - mean;
                q += d * d;
            }
        }
    }
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((int64_t)c * S + sp) * 3;
        o[0] = (float)cnt;
        o[1] = mean;
        o[2] = q;
    }
}

__global__ void k_bn_stats_final(const float* __restrict__ partial, int S, int C, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ rmean,
                                 float* __restrict__ rvar, float momentum, float eps, float* __restrict__ scale,
                                 float* __restrict__ shift, float* __restrict__ save_mean,
                                 float* __restrict__ save_rstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int sp = 0; sp < S; ++sp) {
        const float* o = partial + ((int64_t)c * S + sp) * 3;
        const float nb = o[0];
        if (nb <= 0.f) continue;
        const float delta = o[1] - mean;
        const float nn = n + nb;
        mean += delta * (nb / nn);
        m2 += o[2] + delta * delta * (n * nb / nn);
        n = nn;
    }
    const float var = m2 / n;
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) {
        save_mean[c] = mean;
        save_rstd[c] = rstd;
    }
    if (rmean) {
        const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}

// The same finish from the per-tile partials a conv launch left in its epilogue (ivln_gemm_desc.stat_partials:
// [tiles][C][3]): one block per channel; thread t folds tiles t, t+1024, ... in order, then the 1024 running triples are
// folded pairwise in LDS (a fixed tree: the result does not depend on the launch).
__global__ __launch_bounds__(1024) void k_bn_stats_from_tiles(const float* __restrict__ partial, int tiles, int C,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float momentum, float eps,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ save_mean,
                                                             float* __restrict__ save_rstd) {
    __shared__ float sn[1024], sm[1024], sq[1024];
    const int c = blockIdx.x, t = threadIdx.x;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int i = t; i < tiles; i += 1024) {
        const float* o = partial + ((int64_t)i * C + c) * 3;
        const float nb = o[0];
        if (nb <= 0.f) continue;
        const float delta = o[1] - mean, nn = n + nb;
        mean += delta * (nb / nn);
        m2 += o[2] + delta * delta * (n * nb / nn);
        n = nn;
    }
    sn[t] = n, sm[t] = mean, sq[t] = m2;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (t < w) {
            const float na = sn[t], nb = sn[t + w];
            if (nb > 0.f) {
                const float delta = sm[t + w] - sm[t], nn = na + nb;
                sm[t] += delta * (nb / nn);
                sq[t] += sq[t + w] + delta * delta * (na * nb / nn);
                sn[t] = nn;
            }
        }
        __syncthreads();
    }
    if (t != 0) return;
    n = sn[0], mean = sm[0], m2 = sq[0];
    const float var = n > 0.f ? m2 / n : 0.f;
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) {
        save_mean[c] = mean;
        save_rstd[c] = rstd;
    }
    if (rmean) {
        const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}

// y[n,c,ho,wo] = mean over the 2x2 window of relu(x*scale[c] + shift[c])   (CBRA tail:
// BatchNorm2d -> ReLU -> AvgPool2d(2), models/encoders/map_encoder.py:13-20).  x is NCHW
// (chan_stride = H*W, img_stride = C*H*W) or the producing conv's split-K slabs ([C][N*H*W]:
// chan_stride = N*H*W, img_stride = H*W) whose reduction is fused here.
// lpo (1,2,4,8,16) lanes share one output and split the slabs between them: the tail layers at rollout
// batch have few outputs (8192) but 16-64 slabs each - one thread per output was a 224-load serial chain.
__global__ __launch_bounds__(256) void k_scale_shift_relu_avgpool2(const float* __restrict__ x,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   float* __restrict__ y, int N, int C, int H,
                                                                   int W, int64_t img_stride, int64_t chan_stride,
                                                                   int splits, int64_t slab_stride, int lpo) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int sub = (int)(gid % lpo);
    int64_t idx = gid / lpo;
    const bool live = idx < (int64_t)N * C * Ho * Wo;
    if (!live) idx = 0;  // keep the whole wave in the shuffles
    int wo = (int)(idx % Wo);
    int ho = (int)((idx / Wo) % Ho);
    int nc = (int)(idx / ((int64_t)Wo * Ho));
    int c = nc % C, n = nc / C;
    const float* xp = x + (int64_t)n * img_stride + (int64_t)c * chan_stride + (int64_t)(2 * ho) * W + 2 * wo;
    float2 t = make_float2(0.f, 0.f), u = make_float2(0.f, 0.f);
    for (int z = sub; z < splits; z += lpo) {
        float2 t2 = *reinterpret_cast<const float2*>(xp + (int64_t)z * slab_stride);
        float2 u2 = *reinterpret_cast<const float2*>(xp + (int64_t)z * slab_stride + W);
        t.x += t2.x; t.y += t2.y; u.x += u2.x; u.y += u2.y;
    }
    for (int off = lpo >> 1; off > 0; off >>= 1) {
        t.x += __shfl_xor(t.x, off, 64);
        t.y += __shfl_xor(t.y, off, 64);
        u.x += __shfl_xor(u.x, off, 64);
        u.y += __shfl_xor(u.y, off, 64);
    }
    if (!live || sub != 0) return;
    const float sc = scale[c], sh = shift[c];
    float a = fmaxf(fmaf(t.x, sc, sh), 0.f);
    float b = fmaxf(fmaf(t.y, sc, sh), 0.f);
    float c2 = fmaxf(fmaf(u.x, sc, sh), 0.f);
    float d = fmaxf(fmaf(u.y, sc, sh), 0.f);
    y[idx] = (((a + b) + c2) + d) * 0.25f;
}

// generic 2-D pooling over (NC,H,W): mode 0 = max (pad value -inf), 1 = avg (count_include_pad)
__global__ __launch_bounds__(256) void k_pool2d(const float* __restrict__ x, float* __restrict__ y, int NC, int H,
                                                int W, int Ho, int Wo, int k, int s, int p, int mode) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)NC * Ho * Wo) return;
    int wo = (int)(idx % Wo);
    int ho = (int)((idx / Wo) % Ho);
    int nc = (int)(idx / ((int64_t)Wo * Ho));
    const float* xp = x + (int64_t)nc * H * W;
    float acc = mode == 0 ? -INFINITY : 0.f;
    for (int i = 0; i < k; ++i) {
        int h = ho * s - p + i;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int j = 0; j < k; ++j) {
            int w = wo * s - p + j;
            if ((unsigned)w >= (unsigned)W) continue;
            float v = xp[(int64_t)h * W + w];
            acc = mode == 0 ? fmaxf(acc, v) : acc + v;
        }
    }
    if (mode == 1) acc = acc / (float)(k * k);
    y[idx] = acc;
}

// A 1 x 1 window at stride 2 (mode is irrelevant: one value per output) - the gather in front of a stride-2 1x1 conv
// (ops.conv2d: RedNet's downsample branches, rednet.py:226-232): four consecutive outputs of a row per thread = every second value
// of two 16-byte loads, one 16-byte store.  (The generic kernel reads one dword per output at an 8-byte stride.)
__global__ __launch_bounds__(256) void k_subsample2_x4(const float* __restrict__ x, float* __restrict__ y, int64_t total4, int H, int W,
                                                      int Ho, int Wo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int W4 = Wo >> 2;
    const int j = (int)(idx % W4);
    const int ho = (int)((idx / W4) % Ho);
    const int64_t nc = idx / ((int64_t)W4 * Ho);
    const float4* r = reinterpret_cast<const float4*>(x + (nc * H + 2 * ho) * W + 8 * j);
    const float4 a = r[0], b = r[1];
    *reinterpret_cast<float4*>(y + (nc * Ho + ho) * Wo + 4 * j) = make_float4(a.x, a.z, b.x, b.z);
}

// MaxPool2d(3, 2, 1) (RedNet's stem, rednet.py:195-197), four consecutive outputs of a row per thread: their windows cover input
// columns 8 j - 1 .. 8 j + 7 of three rows - one 4-byte load + two 16-byte loads per row, every lane's loads contiguous with its
// neighbours' - and leave as one 16-byte store.  (The generic kernel reads nine scattered dwords per output: 36 us for the
// stem's 16 x 64 x 128 x 128 map, 2.3 TB/s.)  max() over the same nine values in any order: the same bits.
__global__ __launch_bounds__(256) void k_maxpool3s2p1_x4(const float* __restrict__ x, float* __restrict__ y, int64_t total4, int H, int W,
                                                        int Ho, int Wo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int W4 = Wo >> 2;
    const int j = (int)(idx % W4);
    const int ho = (int)((idx / W4) % Ho);
    const int64_t nc = idx / ((int64_t)W4 * Ho);
    const float* xp = x + nc * H * W;
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int h = 2 * ho - 1 + i;
        if ((unsigned)h < (unsigned)H) {
            const float* r = xp + (int64_t)h * W + 8 * j;
            const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 4);
            const float l = j > 0 ? r[-1] : -INFINITY;
            m[0] = fmaxf(m[0], l), m[1] = fmaxf(m[1], a.x), m[2] = fmaxf(m[2], a.y), m[3] = fmaxf(m[3], a.z), m[4] = fmaxf(m[4], a.w);
            m[5] = fmaxf(m[5], b.x), m[6] = fmaxf(m[6], b.y), m[7] = fmaxf(m[7], b.z), m[8] = fmaxf(m[8], b.w);
        }
    }
    float4 o;
    o.x = fmaxf(fmaxf(m[0], m[1]), m[2]), o.y = fmaxf(fmaxf(m[2], m[3]), m[4]);
    o.z = fmaxf(fmaxf(m[4], m[5]), m[6]), o.w = fmaxf(fmaxf(m[6], m[7]), m[8]);
    *reinterpret_cast<float4*>(y + (nc * Ho + ho) * Wo + 4 * j) = o;
}

// occupancy (1 ch) ++ one_hot(semantic, classes) -> f32 (B, 1+classes, cells)
// (models/encoders/map_encoder.py:85-90)
__global__ __launch_bounds__(256) void k_map_features(const uint8_t* __restrict__ occ,
                                                      const uint8_t* __restrict__ sem, float* __restrict__ y,
                                                      int B, int cells, int classes) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)B * (1 + classes) * cells;
    if (idx >= total) return;
    int p = (int)(idx % cells);
    int c = (int)((idx / cells) % (1 + classes));
    int b = (int)(idx / ((int64_t)cells * (1 + classes)));
    float v;
    if (c == 0) v = (float)occ[(int64_t)b * cells + p];
    else v = sem[(int64_t)b * cells + p] == (uint8_t)(c - 1) ? 1.f : 0.f;
    y[idx] = v;
}

// four cells per thread: one 4-byte load of the labels, one 16-byte store (cells % 4 == 0, aligned tensors); the update
// batch writes 117 MB of features here - 1.9 TB/s with 4-byte stores
__global__ __launch_bounds__(256) void k_map_features4(const uint8_t* __restrict__ occ, const uint8_t* __restrict__ sem,
                                                       float* __restrict__ y, int B, int cells, int classes) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cq = cells >> 2;
    const int64_t total = (int64_t)B * (1 + classes) * cq;
    if (q >= total) return;
    const int p4 = (int)(q % cq);
    const int c = (int)((q / cq) % (1 + classes));
    const int b = (int)(q / ((int64_t)cq * (1 + classes)));
    const uchar4 s4 = *reinterpret_cast<const uchar4*>((c == 0 ? occ : sem) + (int64_t)b * cells + 4 * p4);
    float4 v;
    if (c == 0) {
        v = make_float4((float)s4.x, (float)s4.y, (float)s4.z, (float)s4.w);
    } else {
        const uint8_t k = (uint8_t)(c - 1);
        v = make_float4(s4.x == k ? 1.f : 0.f, s4.y == k ? 1.f : 0.f, s4.z == k ? 1.f : 0.f, s4.w == k ? 1.f : 0.f);
    }
    *reinterpret_cast<float4*>(y + 4 * q) = v;
}

// ------------------------------------------------------------------------------------------
// Instruction encoder front end (models/encoders/instruction_encoder.py:70-82): embedding gather
// and lengths = number of tokens whose embedding row has any non-zero entry.
// out: emb (B*L, E) row-major; lengths (B) int32.
// ------------------------------------------------------------------------------------------
// One block per sequence, 16 waves, a wave per token: lanes run along the embedding row (coalesced 200-byte
// rows; one thread per token walked its row serially - 50 dependent strided loads).
__global__ __launch_bounds__(1024) void k_embed_lengths(const int64_t* __restrict__ tokens,
                                                        const float* __restrict__ table, int L, int E, int V,
                                                        float* __restrict__ emb, int* __restrict__ lengths) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int local = 0;
    for (int t = wave; t < L; t += nw) {
        int64_t tok = tokens[(int64_t)b * L + t];
        if (tok < 0 || tok >= V) tok = 0;
        const float* row = table + tok * E;
        float* o = emb + ((int64_t)b * L + t) * E;
        bool nz = false;
        for (int e = lane; e < E; e += 64) {
            const float v = row[e];
            o[e] = v;
            nz |= (v != 0.0f);
        }
        local += __any(nz) ? 1 : 0;
    }
    if (lane == 0) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) lengths[b] = cnt;
}

// Inference fold of the instruction front end: the embedding lookup followed by the two W_ih projections of the
// bi-LSTM is a lookup in table[v] = embedding[v] . [W_ih ; W_ih_reverse]^T + [b_ih ; b_ih_reverse] (V x 2G, built
// once per weight version by the MFMA GEMM).  One block per sequence, a wave per token: the first G floats of the
// token's row go to gx_f, the next G to gx_r; lengths counts the tokens whose EMBEDDING row has a non-zero element
// (row_nonzero[v], the reference's `(instruction != 0).sum(2) != 0` quirk, instruction_encoder.py:70-78).
//
// Per-episode cache (`cache_tokens` + `dirty`, both or neither): the reference re-encodes the same instruction at every step
// of an episode (map_cma_policy.py:293).  The encoding is a pure function of the tokens, so the block first compares its
// row with the tokens it encoded last time: equal -> dirty[b] = 0 and NOTHING is written (gx, lengths and - in the
// launches that follow and read dirty[b] - the bi-LSTM's output and the folded attention operands keep last step's
// values); different (a new episode, a changed instruction, a compacted batch row, an invalidated cache: tokens < 0) ->
// the row is copied into the cache, dirty[b] = 1 and the row is encoded.  Decided on the device, so it replays in hipGraphs.
__global__ __launch_bounds__(1024) void k_embed_gates(const int64_t* __restrict__ tokens, const float* __restrict__ table,
                                                      const uint8_t* __restrict__ row_nonzero, int L, int G, int V,
                                                      float* __restrict__ gx_f, float* __restrict__ gx_r,
                                                      int* __restrict__ lengths, int64_t* __restrict__ cache_tokens,
                                                      int* __restrict__ dirty) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (cache_tokens) {
        int diff = 0;
        for (int t = threadIdx.x; t < L; t += blockDim.x) diff |= tokens[(int64_t)b * L + t] != cache_tokens[(int64_t)b * L + t];
        diff = __syncthreads_or(diff);
        if (threadIdx.x == 0) dirty[b] = diff ? 1 : 0;
        if (!diff) return;
        for (int t = threadIdx.x; t < L; t += blockDim.x) cache_tokens[(int64_t)b * L + t] = tokens[(int64_t)b * L + t];
    }
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int local = 0;
    for (int t = wave; t < L; t += nw) {
        int64_t tok = tokens[(int64_t)b * L + t];
        if (tok < 0 || tok >= V) tok = 0;
        const float4* row = reinterpret_cast<const float4*>(table + tok * 2 * G);
        float4* of = reinterpret_cast<float4*>(gx_f + ((int64_t)b * L + t) * G);
        float4* orv = reinterpret_cast<float4*>(gx_r + ((int64_t)b * L + t) * G);
        const int q = G >> 2;
        for (int e = lane; e < q; e += 64) {
            of[e] = row[e];
            orv[e] = row[q + e];
        }
        local += row_nonzero[tok] ? 1 : 0;
    }
    if (lane == 0) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) lengths[b] = cnt;
}

// ------------------------------------------------------------------------------------------
// Bidirectional LSTM recurrence (nn.LSTM over a packed sequence, instruction_encoder.py:84-94).
// grid (B, 2): one block per (sequence, direction); 4H = 512 threads, thread g owns gate row g of
// W_hh (H=128 weights in registers); h and c live in LDS.  gx = W_ih x + b_ih precomputed by the
// GEMM for all (b,t).  Gate order i,f,g,o.  out: (B, 2H, L) channel-major, zero for t >= len.
// Optional saves for BPTT: gates (B,2,L,4H) post-activation, cs (B,2,L,H).
// ------------------------------------------------------------------------------------------
// DPP quad permutation of a float (ctrl = p0 | p1<<2 | p2<<4 | p3<<6: lane i of each quad reads lane p_i)
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Barrier that orders LDS traffic only.  __syncthreads() also releases GLOBAL stores, i.e. the compiler
// puts s_waitcnt vmcnt(0) in front of it: in a per-timestep loop that also writes its outputs to HBM
// every step then waits for the store acknowledgement.  Nothing in those loops reads global data written
// by the block, so the recurrent kernels order only their LDS traffic.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

typedef float v2f __attribute__((ext_vector_type(2)));

template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_bidir(const float* __restrict__ gx_f,
                                                      const float* __restrict__ gx_r,
                                                      const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_r,
                                                      const float* __restrict__ bhh_f,
                                                      const float* __restrict__ bhh_r,
                                                      const int* __restrict__ lengths, int L,
                                                      float* __restrict__ out, float* __restrict__ save_gates,
                                                      float* __restrict__ save_c, int B, unsigned* __restrict__ ticket,
                                                      const int* __restrict__ dirty) {
    constexpr int G = 4 * H;
    // Which (sequence, direction) this block runs: its index, or - with `ticket` - the order in which the blocks START.
    // The launcher then over-subscribes the grid (2B * spare blocks for 2B items): beside a kernel that fills some XCDs
    // (the persistent depth encoder; this kernel's 340 registers per SIMD lane do not fit next to it) the blocks the
    // dispatcher handed to the free XCDs start first and take all the work, the others start when the neighbour ends and
    // leave at once.  The block that draws the last ticket re-arms the counter for the next launch.
    __shared__ int s_item;
    int item = blockIdx.x;
    if (ticket) {
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(ticket, 1u);
            if (t == gridDim.x - 1) atomicExch(ticket, 0u);
            s_item = (int)t;
        }
        __syncthreads();
        item = s_item;
        if (item >= 2 * B) return;
    }
    if (dirty && !dirty[item % B]) return;  // (per-episode cache, k_embed_gates: this row's output of last step stands)
    // Quad j (threads 4j..4j+3) owns hidden unit j: lane q multiplies the 4 gate rows {i,f,g,o} of unit j
    // with ITS quarter of h (a 4 x H/4 block of W_hh = H weights in registers), the quad adds the partial
    // sums by DPP, lane q activates gate q, the quad exchanges the four activations by DPP and every lane
    // updates c/h redundantly (c lives in a register).  Per timestep: H/16 ds_read_b128 per thread (every
    // thread reading all of h saturated the LDS port), no LDS round trip for gates or cell state and ONE
    // barrier (h for the next step).
    constexpr int HQ = H / 4, HQP = HQ + 4;  // +4 words per quarter: the 4 quarters hit different banks
    __shared__ __attribute__((aligned(16))) float hs[2][4 * HQP];  // double-buffered: one barrier per step
    const int b = item % B, dir = item / B, tid = threadIdx.x;
    const int q = tid & 3, j = tid >> 2;
    const int g = q * H + j;  // this lane's gate row (PyTorch order i,f,g,o)
    const float* gx = (dir == 0 ? gx_f : gx_r) + (int64_t)b * L * G;
    const float* whh = (dir == 0 ? whh_f : whh_r);
    const float bias = (dir == 0 ? bhh_f : bhh_r)[g];
    v2f w[4][HQ / 2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < HQ / 2; ++k) {
            const float* wp = whh + (int64_t)(r * H + j) * H + q * HQ + 2 * k;
            w[r][k] = v2f{wp[0], wp[1]};
        }
    const int hslot = (j / HQ) * HQP + j % HQ;
    if (q == 0) hs[0][hslot] = 0.f;
    float c = 0.f;
    lds_barrier();
    int len = lengths[b];
    if (len > L) len = L;
    // gx (this lane's gate input, one float per timestep) is prefetched FOUR steps ahead in a rotating
    // register queue: a timestep is ~0.4 us of work but a fresh HBM row costs ~2 us, so a one-step
    // prefetch left every step waiting on memory.
    auto gx_at = [&](int s) -> float {
        return s < len ? gx[(int64_t)(dir == 0 ? s : len - 1 - s) * G + g] : 0.f;
    };
    auto step = [&](int s, float gxv) {
        const int t = dir == 0 ? s : len - 1 - s;
        const float* hcur = hs[s & 1];
        v2f p[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < HQ; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&hcur[q * HQP + k]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __builtin_elementwise_fma(w[r][k / 2], v2f{hv.x, hv.y}, p[r]);
                p[r] = __builtin_elementwise_fma(w[r][k / 2 + 1], v2f{hv.z, hv.w}, p[r]);
            }
        }
        float ps[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = p[r].x + p[r].y;
            v += quad_perm<0xB1>(v);  // lanes 1,0,3,2
            v += quad_perm<0x4E>(v);  // lanes 2,3,0,1
            ps[r] = v;
        }
        const float acc = gxv + bias + (q == 0 ? ps[0] : (q == 1 ? ps[1] : (q == 2 ? ps[2] : ps[3])));
        // sigmoid / tanh through one fast exp each (|err| ~1e-7): tanh(x) = 2*sigmoid(2x) - 1
        // (v_rcp_f32 is 1 ulp; an IEEE division is a ~10-instruction sequence on the per-step critical path)
        const float e = __expf(q == 2 ? -2.f * acc : -acc);
        const float rc = __builtin_amdgcn_rcpf(1.f + e);
        const float a = q == 2 ? 2.f * rc - 1.f : rc;
        if (save_gates) save_gates[(((int64_t)b * 2 + dir) * L + t) * G + g] = a;
        const float ai = quad_perm<0x00>(a), af = quad_perm<0x55>(a), ag = quad_perm<0xAA>(a), ao = quad_perm<0xFF>(a);
        c = af * c + ai * ag;
        const float h = ao * (2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * c)) - 1.f);
        if (q == 0) {
            hs[(s + 1) & 1][hslot] = h;
            out[((int64_t)b * 2 * H + dir * H + j) * L + t] = h;
            if (save_c) save_c[(((int64_t)b * 2 + dir) * L + t) * H + j] = c;
        }
        lds_barrier();
    };
    float g0 = gx_at(0), g1 = gx_at(1), g2 = gx_at(2), g3 = gx_at(3);
    for (int s = 0; s < len; s += 4) {
        step(s, g0);
        g0 = gx_at(s + 4);
        if (s + 1 >= len) break;
        step(s + 1, g1);
        g1 = gx_at(s + 5);
        if (s + 2 >= len) break;
        step(s + 2, g2);
        g2 = gx_at(s + 6);
        if (s + 3 >= len) break;
        step(s + 3, g3);
        g3 = gx_at(s + 7);
    }
    if (q == 0)
        for (int t = len; t < L; ++t) out[((int64_t)b * 2 * H + dir * H + j) * L + t] = 0.f;
}

// ------------------------------------------------------------------------------------------
// Skinny linear: y[n][o] = act(W[o].x[n] + b[o]) for few rows n (rollout batch).  One 256-thread
// block per output row: the 4 waves split K (float4 loads, every lane busy even at K = 3072),
// partial sums meet in LDS.  Rows processed 8 at a time.
// (nn.Linear at models/map_cma_policy.py:156-171,218-224, common/utils.py:176-185)
// ------------------------------------------------------------------------------------------
constexpr int SK_ROWS = 8;

__global__ __launch_bounds__(256) void k_linear_skinny(const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ y, int64_t ldy, int rows, int K, int O,
                                                       int relu) {
    __shared__ float part[4][SK_ROWS];
    const int o = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* wr = W + (int64_t)o * K;
    for (int r0 = 0; r0 < rows; r0 += SK_ROWS) {
        float acc[SK_ROWS];
#pragma unroll
        for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
        if ((K & 3) == 0) {
            for (int k = threadIdx.x * 4; k < K; k += 1024) {
                float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
                for (int r = 0; r < SK_ROWS; ++r) {
                    if (r0 + r < rows) {
                        float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + r) * ldx + k);
                        acc[r] = fmaf(wv.x, xv.x, acc[r]);
                        acc[r] = fmaf(wv.y, xv.y, acc[r]);
                        acc[r] = fmaf(wv.z, xv.z, acc[r]);
                        acc[r] = fmaf(wv.w, xv.w, acc[r]);
                    }
                }
            }
        } else {
            for (int k = threadIdx.x; k < K; k += 256) {
                float wv = wr[k];
#pragma unroll
                for (int r = 0; r < SK_ROWS; ++r)
                    if (r0 + r < rows) acc[r] = fmaf(wv, x[(int64_t)(r0 + r) * ldx + k], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < SK_ROWS; ++r) {
            float v = wave_sum(acc[r]);
            if (lane == 0) part[wave][r] = v;
        }
        __syncthreads();
        if (threadIdx.x < SK_ROWS && r0 + (int)threadIdx.x < rows) {
            const int r = threadIdx.x;
            float v = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
            if (bias) v += bias[o];
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)(r0 + r) * ldy + o] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The two consumers of an encoder's (rows, C, P) feature map in the MapCMA head in ONE launch
// (models/map_cma_policy.py:276-296): the k/v projection nn.Conv1d(C, Ckv, 1) over the P positions and
// nn.Flatten -> nn.Linear(C*P, O) -> ReLU.  Every block stages the feature map (<= 8 rows) once in LDS; the first
// kv_blocks blocks compute 16 k/v channels each (a wave owns 4 channels: lane = (row, position), one LDS read feeds
// four FMAs, weights are wave-uniform), the rest 4 linear outputs each (a wave per output, K split over the lanes).
// ------------------------------------------------------------------------------------------
constexpr int KVL_CO = 4, KVL_ROWS = 8;

__global__ __launch_bounds__(256) void k_kv_linear(const float* __restrict__ feat, int rows, int C, int P,
                                                   const float* __restrict__ Wkv, const float* __restrict__ bkv, int Ckv,
                                                   float* __restrict__ kv, const float* __restrict__ Wl,
                                                   const float* __restrict__ bl, int O, int relu, float* __restrict__ lin,
                                                   int64_t ldl, int kv_blocks) {
    extern __shared__ __attribute__((aligned(16))) float kvl_fs[];
    const int CP = C * P, st = CP + 16;  // +16 words: the rows of a (row, position) wave land on different banks
    for (int i = threadIdx.x * 4; i < rows * CP; i += 1024) {
        const int r = i / CP, k = i - r * CP;
        *reinterpret_cast<float4*>(&kvl_fs[r * st + k]) = *reinterpret_cast<const float4*>(&feat[i]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x < kv_blocks) {
        const int co0 = (blockIdx.x * 4 + wave) * KVL_CO;
        if (co0 >= Ckv) return;
        const int nco = min(KVL_CO, Ckv - co0);
        for (int q = lane; q < rows * P; q += 64) {
            const int r = q / P, px = q - r * P;
            const float* f = kvl_fs + r * st + px;
            float acc[KVL_CO];
#pragma unroll
            for (int j = 0; j < KVL_CO; ++j) acc[j] = (bkv && j < nco) ? bkv[co0 + j] : 0.f;
#pragma unroll 8
            for (int c = 0; c < C; ++c) {
                const float x = f[c * P];
#pragma unroll
                for (int j = 0; j < KVL_CO; ++j) acc[j] = fmaf(Wkv[(int64_t)(co0 + min(j, nco - 1)) * C + c], x, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < KVL_CO; ++j)
                if (j < nco) kv[((int64_t)r * Ckv + co0 + j) * P + px] = acc[j];
        }
    } else {
        const int o = ((int)blockIdx.x - kv_blocks) * 4 + wave;
        if (o >= O) return;
        float acc[KVL_ROWS];
#pragma unroll
        for (int r = 0; r < KVL_ROWS; ++r) acc[r] = 0.f;
        const float* wr = Wl + (int64_t)o * CP;
        for (int k = lane * 4; k < CP; k += 256) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
            for (int r = 0; r < KVL_ROWS; ++r) {
                if (r < rows) {
                    const float4 xv = *reinterpret_cast<const float4*>(&kvl_fs[r * st + k]);
                    acc[r] = fmaf(wv.x, xv.x, acc[r]);
                    acc[r] = fmaf(wv.y, xv.y, acc[r]);
                    acc[r] = fmaf(wv.z, xv.z, acc[r]);
                    acc[r] = fmaf(wv.w, xv.w, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < KVL_ROWS; ++r) {
            if (r < rows) {
                float v = wave_sum(acc[r]);
                if (lane == 0) {
                    if (bl) v += bl[o];
                    if (relu) v = fmaxf(v, 0.f);
                    lin[(int64_t)r * ldl + o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Masked GRU step (habitat-lab RNNStateEncoder single_forward / one step of seq_forward wrapping
// nn.GRU; call sites models/map_cma_policy.py:314-318,346-353).  One block per hidden unit j; the six
// weight rows W_ih[{r,z,n}][j], W_hh[{r,z,n}][j] are dotted with 8 (or 4) rows at a time:
//   gi = W_ih x + b_ih  (or precomputed gi when x == nullptr);  gh = W_hh (h*mask) + b_hh
//   r = s(gi_r+gh_r)  z = s(gi_z+gh_z)  n = tanh(gi_n + r*gh_n)  h' = (1-z)*n + z*h
// Optional saves for BPTT: (rows, H) each of r, z, n, gh_n.
// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_gru_step(const float* __restrict__ x, int64_t ldx, int I,
                                                  const float* __restrict__ gi_pre, int64_t ldgi,
                                                  const float* __restrict__ h_in, int64_t ldh,
                                                  const uint8_t* __restrict__ mask,
                                                  const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                  const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                  float* __restrict__ h_out, int64_t ldo,
                                                  float* __restrict__ h_out2, int64_t ldo2, int rows, int H,
                                                  float* __restrict__ save_r, float* __restrict__ save_z,
                                                  float* __restrict__ save_n, float* __restrict__ save_ghn) {
    // LPR lanes share one row: each lane owns every LPR-th float4 of K, so a (row, unit) needs one
    // log2(LPR)-step shuffle reduction of 6 values and no LDS / barrier (the previous wave-splits-K form
    // spent most of its 13 us in 48 full-wave reductions per block)
    constexpr int RPB = 256 / LPR;  // rows per pass
    const int j = blockIdx.x;
    const int l = threadIdx.x % LPR, rr = threadIdx.x / LPR;
    for (int r0 = 0; r0 < rows; r0 += RPB) {
        const int row = r0 + rr;
        const bool row_ok = row < rows;
        const int rowc = row_ok ? row : 0;
        float ai[3] = {0.f, 0.f, 0.f}, ah[3] = {0.f, 0.f, 0.f};
        if (x) {
            const float* xr = x + (int64_t)rowc * ldx;
            for (int k = l * 4; k < I; k += LPR * 4) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 wv = *reinterpret_cast<const float4*>(w_ih + ((int64_t)g * H + j) * I + k);
                    ai[g] = fmaf(wv.x, xv.x, ai[g]);
                    ai[g] = fmaf(wv.y, xv.y, ai[g]);
                    ai[g] = fmaf(wv.z, xv.z, ai[g]);
                    ai[g] = fmaf(wv.w, xv.w, ai[g]);
                }
            }
        }
        const float mk = mask ? (mask[rowc] ? 1.f : 0.f) : 1.f;
        const float* hr = h_in + (int64_t)rowc * ldh;
        for (int k = l * 4; k < H; k += LPR * 4) {
            float4 hv = *reinterpret_cast<const float4*>(hr + k);
            hv.x *= mk, hv.y *= mk, hv.z *= mk, hv.w *= mk;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 wv = *reinterpret_cast<const float4*>(w_hh + ((int64_t)g * H + j) * H + k);
                ah[g] = fmaf(wv.x, hv.x, ah[g]);
                ah[g] = fmaf(wv.y, hv.y, ah[g]);
                ah[g] = fmaf(wv.z, hv.z, ah[g]);
                ah[g] = fmaf(wv.w, hv.w, ah[g]);
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (x) ai[g] += __shfl_xor(ai[g], off, 64);
                ah[g] += __shfl_xor(ah[g], off, 64);
            }
        }
        if (l == 0 && row_ok) {
            float gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[g] = x ? ai[g] + b_ih[g * H + j] : gi_pre[(int64_t)row * ldgi + g * H + j];
                gh[g] = ah[g] + b_hh[g * H + j];
            }
            float hp = h_in[(int64_t)row * ldh + j] * mk;
            float rg = sigmoidf_(gi[0] + gh[0]);
            float zg = sigmoidf_(gi[1] + gh[1]);
            float ng = tanhf(gi[2] + rg * gh[2]);
            float hn = (1.f - zg) * ng + zg * hp;
            h_out[(int64_t)row * ldo + j] = hn;
            if (h_out2) h_out2[(int64_t)row * ldo2 + j] = hn;
            if (save_r) {
                save_r[(int64_t)row * H + j] = rg;
                save_z[(int64_t)row * H + j] = zg;
                save_n[(int64_t)row * H + j] = ng;
                save_ghn[(int64_t)row * H + j] = gh[2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Cross-modal attention (MapCMANet._attn, models/map_cma_policy.py:266-274): per row n
//   logits[i] = sum_c q[n][c] k[n][c][i];  masked i: logits - 1e8;  attn = softmax(logits*scale)
//   out[n][c'] = sum_i attn[i] v[n][c'][i]
// k: (N, Ck, I) and v: (N, Cv, I) channel-major with image strides; valid_len[n] (or null):
// positions >= valid_len are the masked (all-zero) text positions.
// Two launches so that a 4-row rollout batch still spreads over the chip (one block per row was
// ~80 us of serialised L2 round trips): k_attn_logits grid (rows, I/32) - 32 positions x 8 channel
// parts per block, 8 loads in flight per thread; k_attn_out grid (rows, Cv/16) - every block
// redoes the (tiny) softmax over I, then 16 channels x 16 position parts.
// ------------------------------------------------------------------------------------------
constexpr int ATT_MAX_I = 512;

__global__ __launch_bounds__(256) void k_attn_logits(const float* __restrict__ q, int64_t ldq,
                                                     const float* __restrict__ k, int64_t k_img_stride,
                                                     const int* __restrict__ valid_len, float scale, int Ck, int I,
                                                     float* __restrict__ logits, const int* __restrict__ row_index) {
    __shared__ float qs[1024];
    __shared__ float pl[8][33];
    const int n = blockIdx.x;
    // row_index: several rows attend over the SAME key/value image (update batches: the instruction of a
    // trajectory is encoded once, not once per timestep); valid_len is per image
    const int img = row_index ? row_index[n] : n;
    for (int c = threadIdx.x; c < Ck; c += 256) qs[c] = q[(int64_t)n * ldq + c];
    __syncthreads();
    const float* kp = k + (int64_t)img * k_img_stride;
    const int ti = threadIdx.x & 31, tp = threadIdx.x >> 5;  // position, channel part (8 parts)
    const int i = blockIdx.y * 32 + ti;
    float acc = 0.f;
    if (i < I) {
        int c = tp;
        for (; c + 56 < Ck; c += 64) {
            float kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kv[u] = kp[(int64_t)(c + 8 * u) * I + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(qs[c + 8 * u], kv[u], acc);
        }
        for (; c < Ck; c += 8) acc = fmaf(qs[c], kp[(int64_t)c * I + i], acc);
    }
    pl[tp][ti] = acc;
    __syncthreads();
    if (tp == 0 && i < I) {
        float s = ((pl[0][ti] + pl[1][ti]) + (pl[2][ti] + pl[3][ti])) + ((pl[4][ti] + pl[5][ti]) + (pl[6][ti] + pl[7][ti]));
        const int vl = valid_len ? valid_len[img] : I;
        if (i >= vl) s = s - 1e8f;
        logits[(int64_t)n * I + i] = s * scale;
    }
}

__global__ __launch_bounds__(256) void k_attn_out(const float* __restrict__ logits, const float* __restrict__ v,
                                                  int64_t v_img_stride, int Cv, int I, float* __restrict__ out,
                                                  int64_t ldo, float* __restrict__ save_attn,
                                                  const int* __restrict__ row_index) {
    __shared__ float ps[ATT_MAX_I];
    __shared__ float pl[16][17];
    __shared__ float red[16];
    const int n = blockIdx.x;
    const int img = row_index ? row_index[n] : n;
    float lmax = -INFINITY;
    for (int i = threadIdx.x; i < I; i += 256) {
        float l = logits[(int64_t)n * I + i];
        ps[i] = l;
        lmax = fmaxf(lmax, l);
    }
    lmax = block_max(lmax, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < I; i += 256) {
        float e = expf(ps[i] - lmax);
        ps[i] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int i = threadIdx.x; i < I; i += 256) {
        float a = ps[i] * inv;
        ps[i] = a;
        if (save_attn && blockIdx.y == 0) save_attn[(int64_t)n * I + i] = a;
    }
    __syncthreads();
    const int tc = threadIdx.x >> 4, tpart = threadIdx.x & 15;  // channel within the chunk, position part
    const int c = blockIdx.y * 16 + tc;
    float acc = 0.f;
    if (c < Cv) {
        const float* vp = v + (int64_t)img * v_img_stride + (int64_t)c * I;
        int i = tpart;
        for (; i + 48 < I; i += 64) {
            float v0 = vp[i], v1 = vp[i + 16], v2 = vp[i + 32], v3 = vp[i + 48];
            acc = fmaf(ps[i], v0, acc);
            acc = fmaf(ps[i + 16], v1, acc);
            acc = fmaf(ps[i + 32], v2, acc);
            acc = fmaf(ps[i + 48], v3, acc);
        }
        for (; i < I; i += 16) acc = fmaf(ps[i], vp[i], acc);
    }
    pl[tc][tpart] = acc;
    __syncthreads();
    if (tpart == 0 && c < Cv) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += pl[tc][u];
        out[(int64_t)n * ldo + c] = s;
    }
}

// Attention over a SHORT key axis (I <= 32: the 4x4 depth / map feature grids) in one launch, for up to two
// key/value sets sharing the query (blockIdx.z): logits (32 positions x 8 channel parts), softmax over the
// <= 32 positions by one wave, then 16 output channels x 16 position parts per block - every block redoes the
// (tiny) logits.  Replaces k_attn_logits + k_attn_out per set: 4 launches -> 1 in the rollout head.
struct AttnSet {
    const float* k;
    const float* v;
    float* out;
    int64_t k_img_stride, v_img_stride, ldo;
    int Ck, Cv;
};
__global__ __launch_bounds__(256) void k_attn_small(const float* __restrict__ q, int64_t ldq, float scale, int I,
                                                    const AttnSet s0, const AttnSet s1) {
    __shared__ float qs[1024];
    __shared__ float pl[16][33];
    __shared__ float ps[32];
    const AttnSet& S = blockIdx.z == 0 ? s0 : s1;
    if ((int)blockIdx.y * 16 >= S.Cv) return;  // the two sets may have different output widths
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < S.Ck; c += 256) qs[c] = q[(int64_t)n * ldq + c];
    __syncthreads();
    {
        const float* kp = S.k + (int64_t)n * S.k_img_stride;
        const int ti = threadIdx.x & 31, tp = threadIdx.x >> 5;
        float acc = 0.f;
        if (ti < I) {
            int c = tp;
            for (; c + 56 < S.Ck; c += 64) {
                float kv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kv[u] = kp[(int64_t)(c + 8 * u) * I + ti];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf(qs[c + 8 * u], kv[u], acc);
            }
            for (; c < S.Ck; c += 8) acc = fmaf(qs[c], kp[(int64_t)c * I + ti], acc);
        }
        pl[tp][ti] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // one wave: same summation order as k_attn_logits, softmax over <= 32 values
        const int ti = threadIdx.x;
        float l = -INFINITY;
        if (ti < I)
            l = (((pl[0][ti] + pl[1][ti]) + (pl[2][ti] + pl[3][ti])) + ((pl[4][ti] + pl[5][ti]) + (pl[6][ti] + pl[7][ti]))) * scale;
        const float mx = wave_max(l);
        const float e = ti < I ? expf(l - mx) : 0.f;
        const float sum = wave_sum(e);
        if (ti < I) ps[ti] = e * (1.f / sum);
    }
    __syncthreads();
    const int tc = threadIdx.x >> 4, tpart = threadIdx.x & 15;
    const int c = blockIdx.y * 16 + tc;
    float acc = 0.f;
    if (c < S.Cv) {
        const float* vp = S.v + (int64_t)n * S.v_img_stride + (int64_t)c * I;
        for (int i = tpart; i < I; i += 16) acc = fmaf(ps[i], vp[i], acc);
    }
    __syncthreads();
    pl[tc][tpart] = acc;
    __syncthreads();
    if (tpart == 0 && c < S.Cv) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += pl[tc][u];
        S.out[(int64_t)n * S.ldo + c] = t;
    }
}

// prev-action embedding: idx = (long)((float(a)+1) * mask) (map_cma_policy.py:297-299)
__global__ void k_prev_action_embed(const int64_t* __restrict__ prev_actions, const uint8_t* __restrict__ mask,
                                    const float* __restrict__ table, int rows, int E, int n_emb,
                                    float* __restrict__ out1, int64_t ld1, float* __restrict__ out2,
                                    int64_t ld2) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * E) return;
    int r = idx / E, e = idx % E;
    int64_t a = (int64_t)(((float)prev_actions[r] + 1.f) * (float)(mask[r] ? 1 : 0));
    if (a < 0) a = 0;
    if (a >= n_emb) a = n_emb - 1;
    float v = table[a * E + e];
    out1[(int64_t)r * ld1 + e] = v;
    if (out2) out2[(int64_t)r * ld2 + e] = v;
}

// row-wise argmax of logits (CustomFixedCategorical.mode: probs.argmax(-1), utils.py:168-169)
__global__ void k_argmax_rows(const float* __restrict__ x, int rows, int C, int64_t* __restrict__ out) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int best = 0;
    float bv = x[(int64_t)r * C];
    for (int c = 1; c < C; ++c) {
        float v = x[(int64_t)r * C + c];
        if (v > bv) {
            bv = v;
            best = c;
        }
    }
    out[r] = best;
}

// Action head of a deterministic rollout step in one launch: wave w computes the logits of actions w and w+4
// (lanes split K, one shuffle reduction per row), then the first arg-max per row (distribution.mode(),
// common/utils.py:149-185).  One block, O <= 8 actions; rows processed 8 at a time.
//
// SAMPLE: the sampled step of a DAgger collection instead (models/policy.py:28-46 with deterministic=False, then
// dagger_trainer.py:416-427, 469-472): the action is drawn from softmax(logits) by inverse CDF with the caller's
// uniform u_sample[r] - p_o = exp(l_o - max), the first o whose running sum exceeds u * sum - then mixed with the
// expert, `where(u_beta[r] < beta, expert[r], a)`, and replaced by 0 where the expert says -1 (the episode is being
// skipped).  Host-supplied uniforms make the step a pure function of its inputs: capturable in a hipGraph and
// bit-identical between replay and eager launches.  u_beta == nullptr: no mixing (plain sampling).
template <bool SAMPLE>
__global__ __launch_bounds__(256) void k_linear_argmax(const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       int rows, int K, int O, int64_t* __restrict__ action,
                                                       float* __restrict__ logits_out,
                                                       const float* __restrict__ u_sample,
                                                       const float* __restrict__ u_beta, float beta,
                                                       const double* __restrict__ expert) {
    __shared__ float lg[SK_ROWS][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = 0; r0 < rows; r0 += SK_ROWS) {
        for (int o = wave; o < O; o += 4) {
            const float* wr = W + (int64_t)o * K;
            float acc[SK_ROWS];
#pragma unroll
            for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
                for (int r = 0; r < SK_ROWS; ++r) {
                    if (r0 + r < rows) {
                        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + r) * ldx + k);
                        acc[r] = fmaf(wv.x, xv.x, acc[r]);
                        acc[r] = fmaf(wv.y, xv.y, acc[r]);
                        acc[r] = fmaf(wv.z, xv.z, acc[r]);
                        acc[r] = fmaf(wv.w, xv.w, acc[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < SK_ROWS; ++r) {
                float v = wave_sum(acc[r]);
                if (lane == 0 && r0 + r < rows) {
                    if (bias) v += bias[o];
                    lg[r][o] = v;
                    if (logits_out) logits_out[(int64_t)(r0 + r) * O + o] = v;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < SK_ROWS && r0 + (int)threadIdx.x < rows) {
            const int r = threadIdx.x;
            int best = 0;
            float bv = lg[r][0];
            for (int o = 1; o < O; ++o)
                if (lg[r][o] > bv) {
                    bv = lg[r][o];
                    best = o;
                }
            if (SAMPLE) {
                float p[8], total = 0.f;
                for (int o = 0; o < O; ++o) {
                    p[o] = expf(lg[r][o] - bv);
                    total += p[o];
                }
                const float target = u_sample[r0 + r] * total;
                float run = 0.f;
                best = O - 1;
                for (int o = 0; o < O; ++o) {
                    run += p[o];
                    if (run > target) {
                        best = o;
                        break;
                    }
                }
                if (expert) {
                    const long long e = (long long)expert[r0 + r];
                    if (u_beta && u_beta[r0 + r] < beta) best = (int)e;
                    if (e == -1) best = 0;
                }
            }
            action[r0 + r] = best;
        }
        __syncthreads();
    }
}

// u8 NHWC image -> f32 NCHW / div (TorchVisionResNet.forward: permute(0,3,1,2) and an IEEE division by 255,
// resnet_encoders.py:171-198 with normalize_visual_inputs False)
__global__ __launch_bounds__(256) void k_rgb_to_nchw(const uint8_t* __restrict__ rgb, int B, int H, int W, float div,
                                                     float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B*H*W pixels
    if (idx >= (int64_t)B * H * W) return;
    const int64_t b = idx / ((int64_t)H * W), p = idx - b * H * W;
    const uint8_t* px = rgb + idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(b * 3 + c) * H * W + p] = (float)px[c] / div;
}

// F.adaptive_avg_pool2d: window [floor(i*H/OH), ceil((i+1)*H/OH)) per output row / column
// (SpatialAvgPool to 4x4, resnet_encoders.py:152-158); out may be a channel slice of a wider NCHW buffer
__global__ __launch_bounds__(256) void k_adaptive_avgpool2d(const float* __restrict__ x, int N, int C, int H, int W,
                                                            int OH, int OW, float* __restrict__ out,
                                                            int64_t out_img_stride) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * C * OH * OW) return;
    const int ow = (int)(idx % OW), oh = (int)((idx / OW) % OH);
    const int64_t nc = idx / ((int64_t)OW * OH);
    const int n = (int)(nc / C), c = (int)(nc % C);
    const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
    const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
    const float* xp = x + nc * H * W;
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) s += xp[h * W + w];
    out[(int64_t)n * out_img_stride + ((int64_t)c * OH + oh) * OW + ow] = s / (float)((h1 - h0) * (w1 - w0));
}

// channel argmax over NCHW logits -> u8 labels (predicted_scores.argmax(1), mapper.py:796-798)
__global__ __launch_bounds__(256) void k_argmax_channels_u8(const float* __restrict__ x, int N, int C, int HW,
                                                            uint8_t* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * HW) return;
    int img = (int)(idx / HW), pp = (int)(idx % HW);
    const float* xp = x + (int64_t)img * C * HW + pp;
    int best = 0;
    float bv = xp[0];
    for (int c = 1; c < C; ++c) {
        float v = xp[(int64_t)c * HW];
        if (v > bv) {
            bv = v;
            best = c;
        }
    }
    out[idx] = (uint8_t)best;
}

// RedNet input prep (mapper.py:715-736, 788-793): rgb u8 NHWC -> /255 -> bilinear resize
// (align_corners=False) to (Ho,Wo) -> (x-mean)/std, NCHW f32; depth -> (d-0.213)/0.285.
__global__ __launch_bounds__(256) void k_rgb_resize_normalize(const uint8_t* __restrict__ rgb, int B, int Hi,
                                                              int Wi, int Ho, int Wo, float* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * 3 * Ho * Wo) return;
    int wo = (int)(idx % Wo);
    int ho = (int)((idx / Wo) % Ho);
    int c = (int)((idx / ((int64_t)Wo * Ho)) % 3);
    int b = (int)(idx / ((int64_t)Wo * Ho * 3));
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    // torch upsample_bilinear2d, align_corners=False: src = (dst+0.5)*scale - 0.5, clamped at 0
    float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    float fh = fmaxf(((float)ho + 0.5f) * sh - 0.5f, 0.f);
    float fw = fmaxf(((float)wo + 0.5f) * sw - 0.5f, 0.f);
    int h0 = (int)fh, w0 = (int)fw;
    int h1 = h0 + (h0 < Hi - 1 ? 1 : 0), w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
    float lh = fh - (float)h0, lw = fw - (float)w0;
    auto px = [&](int h, int w) { return (float)rgb[(((int64_t)b * Hi + h) * Wi + w) * 3 + c] / 255.0f; };
    float v = (1.f - lh) * ((1.f - lw) * px(h0, w0) + lw * px(h0, w1)) + lh * ((1.f - lw) * px(h1, w0) + lw * px(h1, w1));
    out[idx] = (v - mean[c]) / stdv[c];
}

// ... frames that already have the network's size (the 256 x 256 RGB-D of the rollout): the bilinear weights are exactly 1 and 0,
// so a value is (u8 / 255 - mean) / std - the same operations in the same order, i.e. the general kernel's bits.  A thread
// takes four pixels of a row: three 4-byte loads, one 16-byte store per channel plane.
__global__ __launch_bounds__(256) void k_rgb_normalize_x4(const uint8_t* __restrict__ rgb, int64_t quads, int HW, float* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int64_t pix = q * 4, b = pix / HW, pp = pix - b * HW;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb + pix * 3);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    const uint8_t by[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24), (uint8_t)w1, (uint8_t)(w1 >> 8),
                            (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24), (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((float)by[3 * e + c] / 255.0f - mean[c]) / stdv[c];
        *reinterpret_cast<float4*>(out + (b * 3 + c) * HW + pp) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void k_affine(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                float sub, float div) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) y[idx] = (x[idx] - sub) / div;
}

// y = relu?(a + b) elementwise (RedNet fuse adds, rednet.py:190-222)
__global__ __launch_bounds__(256) void k_add(const float* __restrict__ a, const float* __restrict__ b,
                                             float* __restrict__ y, int64_t n, int relu) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) {
        float v = a[idx] + b[idx];
        y[idx] = relu ? fmaxf(v, 0.f) : v;
    }
}

// copy a (rows x cols) f32 block between strided buffers (concat plumbing without torch launches)
__global__ __launch_bounds__(256) void k_copy2d(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                int64_t ldd, int rows, int cols, int bcast_rows) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * cols) return;
    int r = (int)(idx / cols), c = (int)(idx % cols);
    dst[(int64_t)r * ldd + c] = src[(bcast_rows ? 0 : (int64_t)r * lds_) + c];
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// Several contiguous device buffers copied by ONE launch (the observation tensors of an env step into the
// step graph's captured input buffers: six hipMemcpyAsync nodes were ~30 us of the 1.15 ms step).
struct CopyJobs {
    const uint8_t* src[8];
    uint8_t* dst[8];
    int64_t bytes[8];
    int first_block[9];  // blocks [first_block[j], first_block[j+1]) work on job j
    int n;
};
constexpr int COPY_CHUNK = 256 * 16 * 4;  // bytes per block: 4 x uint4 per thread

__global__ __launch_bounds__(256) void k_copy_multi(const CopyJobs J) {
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const int64_t off = (int64_t)((int)blockIdx.x - J.first_block[j]) * COPY_CHUNK;
    const int64_t end = min(J.bytes[j], off + COPY_CHUNK);
    const uint8_t* s = J.src[j];
    uint8_t* d = J.dst[j];
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const int64_t vend = off + ((end - off) & ~(int64_t)15);
        for (int64_t i = off + (int64_t)threadIdx.x * 16; i < vend; i += 256 * 16)
            *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
        for (int64_t i = vend + threadIdx.x; i < end; i += 256) d[i] = s[i];
    } else {
        for (int64_t i = off + threadIdx.x; i < end; i += 256) d[i] = s[i];
    }
}

// dst[j][i] += src[j][i] for up to 64 float tensors in one launch: the hand-off of one backward pass's parameter
// gradients to the flat gradient bucket (autograd's own accumulation is one elementwise launch per parameter: 54 per
// MapCMA update, profiles/r02_update_T64N8_kernel_stats.csv).
struct AddJobs {
    const float* src[64];
    float* dst[64];
    int64_t n[64];
    int first_block[65];
    int count;
};
constexpr int ADD_CHUNK = 256 * 4 * 4;  // floats per block: 4 x float4 per thread

__global__ __launch_bounds__(256) void k_add_multi(const AddJobs J) {
    int j = 0;
    while (j + 1 < J.count && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const int64_t off = (int64_t)((int)blockIdx.x - J.first_block[j]) * ADD_CHUNK;
    const int64_t end = min(J.n[j], off + ADD_CHUNK);
    const float* s = J.src[j];
    float* d = J.dst[j];
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const int64_t vend = off + ((end - off) & ~(int64_t)3);
        for (int64_t i = off + (int64_t)threadIdx.x * 4; i < vend; i += 256 * 4) {
            const float4 a = *reinterpret_cast<const float4*>(s + i);
            float4 b = *reinterpret_cast<float4*>(d + i);
            b.x += a.x, b.y += a.y, b.z += a.z, b.w += a.w;
            *reinterpret_cast<float4*>(d + i) = b;
        }
        for (int64_t i = vend + threadIdx.x; i < end; i += 256) d[i] += s[i];
    } else {
        for (int64_t i = off + threadIdx.x; i < end; i += 256) d[i] += s[i];
    }
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP)

extern "C" {

int ivln_groupnorm2_f32(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                        int N, int C, int HW, int groups, float eps, int relu, int64_t x_img_stride,
                        int64_t x_chan_stride, int splits, int64_t slab_stride, int64_t y_img_stride,
                        int64_t r_img_stride, float* save_mean, float* save_rstd, const float* x2, const float* gamma2,
                        const float* beta2, int64_t x2_img_stride, int64_t x2_chan_stride, int splits2,
                        int64_t slab_stride2, void* stream) {
    if (N <= 0 || C <= 0 || groups <= 0 || C % groups) return IVLN_E_INVALID;
    if (x2 && (!gamma2 || !beta2)) return IVLN_E_INVALID;
    if (x_chan_stride <= 0) x_chan_stride = HW;
    if (x_img_stride <= 0) x_img_stride = (int64_t)C * HW;
    if (y_img_stride <= 0) y_img_stride = (int64_t)C * HW;
    if (r_img_stride <= 0) r_img_stride = (int64_t)C * HW;
    if (x2_chan_stride <= 0) x2_chan_stride = HW;
    if (x2_img_stride <= 0) x2_img_stride = (int64_t)C * HW;
    if (splits < 1) splits = 1;
    if (splits2 < 1) splits2 = 1;
    hipLaunchKernelGGL(k_groupnorm, dim3(N * groups), dim3(GN_THREADS), 0, (hipStream_t)stream, x, gamma, beta,
                       residual, y, C, HW, groups, eps, relu, x_img_stride, x_chan_stride, splits, slab_stride,
                       y_img_stride, r_img_stride, save_mean, save_rstd, x2, gamma2, beta2, x2_img_stride,
                       x2_chan_stride, splits2, slab_stride2);
    return LAUNCH_OK();
}

int ivln_groupnorm_f32(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                       int N, int C, int HW, int groups, float eps, int relu, int64_t x_img_stride,
                       int64_t x_chan_stride, int splits, int64_t slab_stride, int64_t y_img_stride,
                       int64_t r_img_stride, float* save_mean, float* save_rstd, void* stream) {
    return ivln_groupnorm2_f32(x, gamma, beta, residual, y, N, C, HW, groups, eps, relu, x_img_stride, x_chan_stride,
                               splits, slab_stride, y_img_stride, r_img_stride, save_mean, save_rstd, nullptr, nullptr,
                               nullptr, 0, 0, 1, 0, stream);
}

int ivln_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream) {
    hipLaunchKernelGGL(k_bn_fold, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, conv_bias, eps, C, scale, shift);
    return LAUNCH_OK();
}

int ivln_bn_train_stats_f32(const float* x, int N, int C, int HW, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* scale,
                            float* shift, float* save_mean, float* save_rstd, float* ws, int64_t ws_floats,
                            void* stream) {
    if (!ws || ws_floats < (int64_t)3 * C) return IVLN_E_INVALID;
    int S = (int)(((int64_t)N * HW + 16383) / 16384);  // >= 16k elements per block
    if (S > N) S = N;
    if (S > 64) S = 64;
    if ((int64_t)S * C * 3 > ws_floats) S = (int)(ws_floats / ((int64_t)C * 3));
    if (S < 1) S = 1;
    const int ips = (N + S - 1) / S;
    S = (N + ips - 1) / ips;
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(C, S), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, ips, ws);
    hipLaunchKernelGGL(k_bn_stats_final, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, S, C, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_rstd);
    return LAUNCH_OK();
}

int ivln_bn_stats_from_partials_f32(const float* partials, int tiles, int C, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                    float* shift, float* save_mean, float* save_rstd, void* stream) {
    if (!partials || tiles <= 0 || C <= 0 || !gamma || !beta || !scale || !shift) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_bn_stats_from_tiles, dim3(C), dim3(1024), 0, (hipStream_t)stream, partials, tiles, C, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_rstd);
    return LAUNCH_OK();
}

int ivln_scale_shift_relu_avgpool2_f32(const float* x, const float* scale, const float* shift, float* y, int N,
                                       int C, int H, int W, int64_t img_stride, int64_t chan_stride, int splits,
                                       int64_t slab_stride, void* stream) {
    if ((W & 1) || (H & 1)) return IVLN_E_INVALID;
    if (chan_stride <= 0) chan_stride = (int64_t)H * W;
    if (img_stride <= 0) img_stride = (int64_t)C * H * W;
    if (splits < 1) splits = 1;
    int64_t total = (int64_t)N * C * (H / 2) * (W / 2);
    int lpo = 1;  // lanes per output: split the slab sum until ~64K threads are in flight
    while (lpo < 16 && lpo * 2 <= splits && total * lpo < 65536) lpo *= 2;
    hipLaunchKernelGGL(k_scale_shift_relu_avgpool2, dim3(nblk(total * lpo)), dim3(256), 0, (hipStream_t)stream, x,
                       scale, shift, y, N, C, H, W, img_stride, chan_stride, splits, slab_stride, lpo);
    return LAUNCH_OK();
}

int ivln_pool2d_f32(const float* x, float* y, int NC, int H, int W, int k, int s, int p, int mode, void* stream) {
    int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    int64_t total = (int64_t)NC * Ho * Wo;
    if (mode == 0 && k == 3 && s == 2 && p == 1 && (W & 7) == 0 && (Wo & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        hipLaunchKernelGGL(k_maxpool3s2p1_x4, dim3(nblk(total / 4)), dim3(256), 0, (hipStream_t)stream, x, y, total / 4, H, W, Ho, Wo);
        return LAUNCH_OK();
    }
    if (k == 1 && s == 2 && p == 0 && (W & 7) == 0 && (H & 1) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        hipLaunchKernelGGL(k_subsample2_x4, dim3(nblk(total / 4)), dim3(256), 0, (hipStream_t)stream, x, y, total / 4, H, W, Ho, Wo);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(k_pool2d, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, Ho, Wo, k, s,
                       p, mode);
    return LAUNCH_OK();
}

int ivln_map_features_f32(const uint8_t* occ, const uint8_t* sem, float* y, int B, int cells, int classes,
                          void* stream) {
    int64_t total = (int64_t)B * (1 + classes) * cells;
    if ((cells & 3) == 0 && (((uintptr_t)occ | (uintptr_t)sem) & 3) == 0 && ((uintptr_t)y & 15) == 0)
        hipLaunchKernelGGL(k_map_features4, dim3(nblk(total / 4)), dim3(256), 0, (hipStream_t)stream, occ, sem, y, B, cells,
                           classes);
    else
        hipLaunchKernelGGL(k_map_features, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, occ, sem, y, B, cells,
                           classes);
    return LAUNCH_OK();
}

int ivln_embed_lengths(const int64_t* tokens, const float* table, int B, int L, int E, int V, float* emb,
                       int* lengths, void* stream) {
    hipLaunchKernelGGL(k_embed_lengths, dim3(B), dim3(1024), 0, (hipStream_t)stream, tokens, table, L, E, V, emb,
                       lengths);
    return LAUNCH_OK();
}

int ivln_embed_gates_f32(const int64_t* tokens, const float* table, const uint8_t* row_nonzero, int B, int L, int G, int V,
                         float* gx_f, float* gx_r, int* lengths, void* stream) {
    return ivln_embed_gates_cached_f32(tokens, table, row_nonzero, B, L, G, V, gx_f, gx_r, lengths, nullptr, nullptr, stream);
}

int ivln_embed_gates_cached_f32(const int64_t* tokens, const float* table, const uint8_t* row_nonzero, int B, int L, int G, int V,
                                float* gx_f, float* gx_r, int* lengths, int64_t* cache_tokens, int* dirty, void* stream) {
    if (!tokens || !table || !row_nonzero || !gx_f || !gx_r || !lengths || B <= 0 || L <= 0 || G <= 0 || (G & 3) || V <= 0)
        return IVLN_E_INVALID;
    if ((cache_tokens == nullptr) != (dirty == nullptr)) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_embed_gates, dim3(B), dim3(1024), 0, (hipStream_t)stream, tokens, table, row_nonzero, L, G, V, gx_f,
                       gx_r, lengths, cache_tokens, dirty);
    return LAUNCH_OK();
}

int ivln_lstm_bidir_fwd_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                            const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                            float* out, float* save_gates, float* save_c, void* stream) {
    return ivln_lstm_bidir_fwd_cached_f32(gx_f, gx_r, whh_f, whh_r, bhh_f, bhh_r, lengths, B, L, H, out, save_gates, save_c,
                                          nullptr, 1, nullptr, stream);
}

int ivln_lstm_bidir_fwd_spread_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                                   const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                                   float* out, float* save_gates, float* save_c, unsigned* ticket, int spare, void* stream) {
    return ivln_lstm_bidir_fwd_cached_f32(gx_f, gx_r, whh_f, whh_r, bhh_f, bhh_r, lengths, B, L, H, out, save_gates, save_c,
                                          ticket, spare, nullptr, stream);
}

int ivln_lstm_bidir_fwd_cached_f32(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                                   const float* bhh_f, const float* bhh_r, const int* lengths, int B, int L, int H,
                                   float* out, float* save_gates, float* save_c, unsigned* ticket, int spare, const int* dirty,
                                   void* stream) {
    if (H != 128) return IVLN_E_UNSUPPORTED;
    if (B <= 0 || spare < 1 || spare > 8 || (spare > 1 && !ticket)) return IVLN_E_INVALID;
    hipLaunchKernelGGL((k_lstm_bidir<128>), dim3(2 * B * (ticket ? spare : 1)), dim3(512), 0, (hipStream_t)stream, gx_f, gx_r,
                       whh_f, whh_r, bhh_f, bhh_r, lengths, L, out, save_gates, save_c, B, ticket, dirty);
    return LAUNCH_OK();
}

int ivln_kv_linear_f32(const float* feat, int rows, int C, int P, const float* w_kv, const float* b_kv, int Ckv, float* kv,
                       const float* w_lin, const float* b_lin, int O, int relu, float* lin, int64_t ld_lin, void* stream) {
    if (!feat || !w_kv || !kv || !w_lin || !lin || rows <= 0 || C <= 0 || P <= 0 || Ckv <= 0 || O <= 0) return IVLN_E_INVALID;
    if (rows > KVL_ROWS || ((C * P) & 3)) return IVLN_E_UNSUPPORTED;
    const size_t bytes = sizeof(float) * (size_t)rows * (C * P + 16);
    if (bytes > 150 * 1024) return IVLN_E_UNSUPPORTED;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void*)k_kv_linear, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return IVLN_E_HIP;
        raised = true;
    }
    const int kv_blocks = (Ckv + 4 * KVL_CO - 1) / (4 * KVL_CO), lin_blocks = (O + 3) / 4;
    hipLaunchKernelGGL(k_kv_linear, dim3(kv_blocks + lin_blocks), dim3(256), bytes, (hipStream_t)stream, feat, rows, C, P, w_kv,
                       b_kv, Ckv, kv, w_lin, b_lin, O, relu, lin, ld_lin, kv_blocks);
    return LAUNCH_OK();
}

int ivln_linear_skinny_f32(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t ldy,
                           int rows, int K, int O, int relu, void* stream) {
    if (rows <= 0 || K <= 0 || O <= 0) return IVLN_E_INVALID;
    if ((K & 3) == 0 && (ldx & 3)) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_linear_skinny, dim3(O), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, y, ldy,
                       rows, K, O, relu);
    return LAUNCH_OK();
}

int ivln_gru_step_f32(const float* x, int64_t ldx, int I, const float* gi_pre, int64_t ldgi, const float* h_in,
                      int64_t ldh, const uint8_t* mask, const float* w_ih, const float* w_hh, const float* b_ih,
                      const float* b_hh, float* h_out, int64_t ldo, float* h_out2, int64_t ldo2, int rows, int H,
                      float* save_r, float* save_z, float* save_n, float* save_ghn, void* stream) {
    if (rows <= 0 || (H & 3) || (x && (I & 3)) || (ldh & 3) || (x && (ldx & 3))) return IVLN_E_INVALID;
    if (rows <= 4)
        hipLaunchKernelGGL(k_gru_step<64>, dim3(H), dim3(256), 0, (hipStream_t)stream, x, ldx, I, gi_pre, ldgi,
                           h_in, ldh, mask, w_ih, w_hh, b_ih, b_hh, h_out, ldo, h_out2, ldo2, rows, H, save_r, save_z,
                           save_n, save_ghn);
    else
        hipLaunchKernelGGL(k_gru_step<32>, dim3(H), dim3(256), 0, (hipStream_t)stream, x, ldx, I, gi_pre, ldgi,
                           h_in, ldh, mask, w_ih, w_hh, b_ih, b_hh, h_out, ldo, h_out2, ldo2, rows, H, save_r, save_z,
                           save_n, save_ghn);
    return LAUNCH_OK();
}

/* GRU over a whole time-major sequence batch in ONE call: T dependent k_gru_step launches enqueued from C.  The
 * per-timestep Python -> ctypes round trip (~15 us) was longer than the 6.5 us kernel, so the 128 forward steps of
 * an update left the GPU idle for ~1 ms (profiles/r02_update_kernel_stats.csv); enqueued back to back they are
 * GPU-bound.  gi = W_ih x + b_ih for all T*N rows (one GEMM, done by the caller). */
int ivln_cma_seq_fwd_f32(const float* gi, const float* h0, int64_t ld_h0, const uint8_t* masks, const float* w_hh,
                         const float* b_hh, float* out, int64_t ldo, float* state_out, int64_t ld_so, int T, int N,
                         int H, float* save_r, float* save_z, float* save_n, float* save_ghn, void* sync_ws,
                         void* stream) {
    if (!gi || !h0 || !masks || !w_hh || !b_hh || !out || T <= 0 || N <= 0 || (H & 3) || (ld_h0 & 3) || (ldo & 3))
        return IVLN_E_INVALID;
    if (sync_ws && T > 1 && ivln_cma_seq_persistent_ok(N, H, 0)) {   // one persistent launch (gru_seq.hip)
        const int rc = ivln_gru_seq_fwd_persistent(gi, h0, ld_h0, masks, w_hh, b_hh, out, ldo, state_out, ld_so, T, N,
                                                   save_r, save_z, save_n, save_ghn, sync_ws, stream);
        if (rc != IVLN_E_UNSUPPORTED) return rc;
    }
    for (int t = 0; t < T; ++t) {
        const int64_t r0 = (int64_t)t * N;
        const float* h_in = t == 0 ? h0 : out + (r0 - N) * ldo;
        const int64_t ldh = t == 0 ? ld_h0 : ldo;
        float* so = (t == T - 1) ? state_out : nullptr;
        float* sr = save_r ? save_r + r0 * H : nullptr;
        float* sz = save_r ? save_z + r0 * H : nullptr;
        float* sn = save_r ? save_n + r0 * H : nullptr;
        float* sg = save_r ? save_ghn + r0 * H : nullptr;
        if (N <= 4)
            hipLaunchKernelGGL(k_gru_step<64>, dim3(H), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (int64_t)0,
                               0, gi + r0 * 3 * H, (int64_t)3 * H, h_in, ldh, masks + r0, (const float*)nullptr, w_hh,
                               (const float*)nullptr, b_hh, out + r0 * ldo, ldo, so, ld_so, N, H, sr, sz, sn, sg);
        else
            hipLaunchKernelGGL(k_gru_step<32>, dim3(H), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (int64_t)0,
                               0, gi + r0 * 3 * H, (int64_t)3 * H, h_in, ldh, masks + r0, (const float*)nullptr, w_hh,
                               (const float*)nullptr, b_hh, out + r0 * ldo, ldo, so, ld_so, N, H, sr, sz, sn, sg);
    }
    return LAUNCH_OK();
}

int ivln_attn_fwd_idx_f32(const float* q, int64_t ldq, const float* k, int64_t k_img_stride, const float* v,
                          int64_t v_img_stride, const int* valid_len, float scale, int rows, int Ck, int Cv, int I,
                          float* out, int64_t ldo, float* save_attn, float* logits_ws, const int* row_index,
                          void* stream) {
    if (I > ATT_MAX_I || Ck > 1024 || !logits_ws) return IVLN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_attn_logits, dim3(rows, (I + 31) / 32), dim3(256), 0, (hipStream_t)stream, q, ldq, k,
                       k_img_stride, valid_len, scale, Ck, I, logits_ws, row_index);
    hipLaunchKernelGGL(k_attn_out, dim3(rows, (Cv + 15) / 16), dim3(256), 0, (hipStream_t)stream, logits_ws, v,
                       v_img_stride, Cv, I, out, ldo, save_attn, row_index);
    return LAUNCH_OK();
}

int ivln_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t k_img_stride, const float* v,
                      int64_t v_img_stride, const int* valid_len, float scale, int rows, int Ck, int Cv, int I,
                      float* out, int64_t ldo, float* save_attn, float* logits_ws, void* stream) {
    return ivln_attn_fwd_idx_f32(q, ldq, k, k_img_stride, v, v_img_stride, valid_len, scale, rows, Ck, Cv, I, out, ldo,
                                 save_attn, logits_ws, nullptr, stream);
}

int ivln_attn_small2_f32(const float* q, int64_t ldq, float scale, int rows, int I, const float* k0,
                         int64_t k0_img_stride, const float* v0, int64_t v0_img_stride, int Ck0, int Cv0, float* out0,
                         int64_t ldo0, const float* k1, int64_t k1_img_stride, const float* v1, int64_t v1_img_stride,
                         int Ck1, int Cv1, float* out1, int64_t ldo1, void* stream) {
    if (I > 32 || I <= 0 || Ck0 > 1024 || Ck1 > 1024 || rows <= 0) return IVLN_E_UNSUPPORTED;
    AttnSet a{k0, v0, out0, k0_img_stride, v0_img_stride, ldo0, Ck0, Cv0};
    AttnSet b{k1, v1, out1, k1_img_stride, v1_img_stride, ldo1, Ck1, Cv1};
    const int sets = k1 ? 2 : 1;
    const int cv = sets == 2 && Cv1 > Cv0 ? Cv1 : Cv0;
    hipLaunchKernelGGL(k_attn_small, dim3(rows, (cv + 15) / 16, sets), dim3(256), 0, (hipStream_t)stream, q, ldq, scale,
                       I, a, b);
    return LAUNCH_OK();
}

int ivln_prev_action_embed_f32(const int64_t* prev_actions, const uint8_t* mask, const float* table, int rows, int E,
                               int n_emb, float* out1, int64_t ld1, float* out2, int64_t ld2, void* stream) {
    hipLaunchKernelGGL(k_prev_action_embed, dim3((rows * E + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       prev_actions, mask, table, rows, E, n_emb, out1, ld1, out2, ld2);
    return LAUNCH_OK();
}

__global__ void k_tour_memory(const float* __restrict__ mem, int64_t ld_mem, const float* __restrict__ h, int64_t ld_h,
                              const uint8_t* __restrict__ mask, int N, int H, float* __restrict__ out1, int64_t ld1,
                              float* __restrict__ out2, int64_t ld2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * H) return;
    const int n = i / H, c = i - n * H;
    float v = mem[(int64_t)n * ld_mem + c];
    if (h) v = fmaxf(v, h[(int64_t)n * ld_h + c]);
    v = mask && !mask[n] ? 0.f : v;
    out1[(int64_t)n * ld1 + c] = v;
    if (out2) out2[(int64_t)n * ld2 + c] = v;
}

int ivln_tour_memory_f32(const float* mem, int64_t ld_mem, const float* h, int64_t ld_h, const uint8_t* mask, int N,
                         int H, float* out1, int64_t ld1, float* out2, int64_t ld2, void* stream) {
    if (N <= 0 || H <= 0 || !mem || !out1) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_tour_memory, dim3((N * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, mem, ld_mem, h, ld_h,
                       mask, N, H, out1, ld1, out2, ld2);
    return LAUNCH_OK();
}

int ivln_argmax_rows(const float* x, int rows, int C, int64_t* out, void* stream) {
    hipLaunchKernelGGL(k_argmax_rows, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, rows, C, out);
    return LAUNCH_OK();
}

int ivln_linear_argmax_f32(const float* x, int64_t ldx, const float* W, const float* bias, int rows, int K, int O,
                           int64_t* action, float* logits_out, void* stream) {
    if (O > 8 || O <= 0 || (K & 3) || (ldx & 3) || rows <= 0) return IVLN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_linear_argmax<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, rows, K, O,
                       action, logits_out, (const float*)nullptr, (const float*)nullptr, 0.f, (const double*)nullptr);
    return LAUNCH_OK();
}

int ivln_linear_sample_f32(const float* x, int64_t ldx, const float* W, const float* bias, int rows, int K, int O,
                           const float* u_sample, const float* u_beta, float beta, const double* expert,
                           int64_t* action, float* logits_out, void* stream) {
    if (O > 8 || O <= 0 || (K & 3) || (ldx & 3) || rows <= 0) return IVLN_E_UNSUPPORTED;
    if (!u_sample || (u_beta && !expert)) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_linear_argmax<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, rows, K, O,
                       action, logits_out, u_sample, u_beta, beta, expert);
    return LAUNCH_OK();
}

int ivln_rgb_to_nchw_f32(const uint8_t* rgb, int B, int H, int W, float div, float* out, void* stream) {
    if (B <= 0 || div == 0.f) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_rgb_to_nchw, dim3(nblk((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, rgb, B, H, W,
                       div, out);
    return LAUNCH_OK();
}

int ivln_adaptive_avgpool2d_f32(const float* x, int N, int C, int H, int W, int OH, int OW, float* out,
                                int64_t out_img_stride, void* stream) {
    if (N <= 0 || OH <= 0 || OW <= 0) return IVLN_E_INVALID;
    if (out_img_stride <= 0) out_img_stride = (int64_t)C * OH * OW;
    hipLaunchKernelGGL(k_adaptive_avgpool2d, dim3(nblk((int64_t)N * C * OH * OW)), dim3(256), 0, (hipStream_t)stream, x,
                       N, C, H, W, OH, OW, out, out_img_stride);
    return LAUNCH_OK();
}

int ivln_argmax_channels_u8(const float* x, int N, int C, int HW, uint8_t* out, void* stream) {
    hipLaunchKernelGGL(k_argmax_channels_u8, dim3(nblk((int64_t)N * HW)), dim3(256), 0, (hipStream_t)stream, x, N, C,
                       HW, out);
    return LAUNCH_OK();
}

int ivln_rgb_resize_normalize_f32(const uint8_t* rgb, int B, int Hi, int Wi, int Ho, int Wo, float* out,
                                  void* stream) {
    if (Hi == Ho && Wi == Wo && ((int64_t)Ho * Wo) % 4 == 0 && !((uintptr_t)rgb & 3) && !((uintptr_t)out & 15)) {
        const int64_t quads = (int64_t)B * Ho * Wo / 4;
        hipLaunchKernelGGL(k_rgb_normalize_x4, dim3(nblk(quads)), dim3(256), 0, (hipStream_t)stream, rgb, quads, Ho * Wo, out);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(k_rgb_resize_normalize, dim3(nblk((int64_t)B * 3 * Ho * Wo)), dim3(256), 0,
                       (hipStream_t)stream, rgb, B, Hi, Wi, Ho, Wo, out);
    return LAUNCH_OK();
}

int ivln_affine_f32(const float* x, float* y, int64_t n, float sub, float div, void* stream) {
    hipLaunchKernelGGL(k_affine, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, sub, div);
    return LAUNCH_OK();
}

int ivln_add_f32(const float* a, const float* b, float* y, int64_t n, int relu, void* stream) {
    hipLaunchKernelGGL(k_add, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, relu);
    return LAUNCH_OK();
}

int ivln_copy2d_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols,
                    int broadcast_rows, void* stream) {
    hipLaunchKernelGGL(k_copy2d, dim3(nblk((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       dst, ld_dst, rows, cols, broadcast_rows);
    return LAUNCH_OK();
}

int ivln_copy_multi(const void* const* srcs, void* const* dsts, const int64_t* bytes, int n, void* stream) {
    if (n < 0 || n > 8) return IVLN_E_INVALID;
    CopyJobs J;
    int blocks = 0, m = 0;
    for (int j = 0; j < n; ++j) {
        if (bytes[j] <= 0) continue;
        if (!srcs[j] || !dsts[j]) return IVLN_E_INVALID;
        J.src[m] = (const uint8_t*)srcs[j];
        J.dst[m] = (uint8_t*)dsts[j];
        J.bytes[m] = bytes[j];
        J.first_block[m] = blocks;
        blocks += (int)((bytes[j] + COPY_CHUNK - 1) / COPY_CHUNK);
        ++m;
    }
    if (m == 0) return IVLN_OK;
    J.first_block[m] = blocks;
    J.n = m;
    hipLaunchKernelGGL(k_copy_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    return LAUNCH_OK();
}

int ivln_rednet_fwd(const ivln_rednet_op* table, int n_ops, const uint8_t* rgb, const float* depth, uint8_t* labels_out,
                    void* stream) {
    if (!table || n_ops <= 0) return IVLN_E_INVALID;
    for (int k = 0; k < n_ops; ++k) {
        const ivln_rednet_op& op = table[k];
        int rc = IVLN_E_INVALID;
        switch (op.kind) {
            case IVLN_OP_GEMM:
                rc = ivln_gemm_f32(&op.gemm, stream);
                break;
            case IVLN_OP_ADD:
                rc = ivln_add_f32((const float*)op.src0, (const float*)op.src1, (float*)op.dst, op.n, op.i[0], stream);
                break;
            case IVLN_OP_POOL:
                rc = ivln_pool2d_f32((const float*)op.src0, (float*)op.dst, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4],
                                     op.i[5], op.i[6], stream);
                break;
            case IVLN_OP_RGB_NORM: {
                const uint8_t* src = op.src0 ? (const uint8_t*)op.src0 : rgb;
                if (src)
                    rc = ivln_rgb_resize_normalize_f32(src, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], (float*)op.dst,
                                                       stream);
                break;
            }
            case IVLN_OP_AFFINE: {
                const float* src = op.src0 ? (const float*)op.src0 : depth;
                if (src) rc = ivln_affine_f32(src, (float*)op.dst, op.n, op.f[0], op.f[1], stream);
                break;
            }
            case IVLN_OP_ARGMAX_U8: {
                uint8_t* dst = op.dst ? (uint8_t*)op.dst : labels_out;
                if (dst) rc = ivln_argmax_channels_u8((const float*)op.src0, op.i[0], op.i[1], op.i[2], dst, stream);
                break;
            }
            default:
                break;
        }
        if (rc != IVLN_OK) return rc;
    }
    return IVLN_OK;
}

int ivln_add_multi_f32(const float* const* srcs, float* const* dsts, const int64_t* counts, int n, void* stream) {
    if (n < 0 || n > 64) return IVLN_E_INVALID;
    AddJobs J;
    int blocks = 0, m = 0;
    for (int j = 0; j < n; ++j) {
        if (counts[j] <= 0) continue;
        if (!srcs[j] || !dsts[j]) return IVLN_E_INVALID;
        J.src[m] = srcs[j];
        J.dst[m] = dsts[j];
        J.n[m] = counts[j];
        J.first_block[m] = blocks;
        blocks += (int)((counts[j] + ADD_CHUNK - 1) / ADD_CHUNK);
        ++m;
    }
    if (m == 0) return IVLN_OK;
    J.first_block[m] = blocks;
    J.count = m;
    hipLaunchKernelGGL(k_add_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    return LAUNCH_OK();
}

}  // extern "C"
