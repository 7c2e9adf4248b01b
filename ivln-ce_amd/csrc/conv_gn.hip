// Convolution + GroupNorm (+ second normalised operand / residual) (+ ReLU) in ONE launch, for the DD-PPO depth
// ResNet at rollout batch sizes (habitat-lab ResNetEncoder: conv -> GroupNorm(16, C) -> ReLU triples, call site
// ivlnce_baselines/models/encoders/resnet_encoders.py:31-43, 95; restated in oracle/habitat_ext_ref.py:37-175).
//
// Why a second conv kernel family: at 4-8 envs every conv of that encoder is ~4-9 MMAC per image - microseconds of
// arithmetic - and the step was bounded by the LAUNCH CHAIN (conv, then a GroupNorm launch that reduces the conv's
// split-K slabs: 105 dependent launches, 50 of them GroupNorm, 324 us per step).  GroupNorm statistics couple
// exactly the cpg = C/groups channels of one group over all pixels of one image, so the decomposition here is
//     workgroup = (image, group):  a [cpg x HWo] output tile with the FULL reduction K = Cin*k*k inside the block
// - K is split over the block's waves (partial tiles meet in LDS), the finished tile stays in LDS, statistics are the
// usual two passes (mean, then biased variance) and the normalised / activated tile is written once.  No split-K
// slabs, no separate GroupNorm launch, no workspace traffic.  The bottleneck's last conv can also run its
// downsample branch (second conv + its own GroupNorm, added before the ReLU) in the same block.
//
// Arithmetic is VALU, not MFMA, on purpose: cpg is 2..64 (2-8 in the layers that matter), an MFMA tile's M = 32 / 16
// would idle 50-94 % of the matrix pipe, and fp32 MFMA runs at the fp32 VALU rate anyway (MI355X_MICROARCH.md).
// Weights of the group are wave-uniform operands (scalar loads) when a wave's lanes are 64 pixels of one K slice.
// Bound: L2 -> CU bandwidth (every group's block reads its image's whole input, 64-512 KB) and per-block latency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/ivln_hip.h"
#include "family_timing.h"

namespace {

constexpr int CG_THREADS = 512;

struct ConvSrc {            // one convolution feeding the tile
    const float* x;         // input  (N, Cin, Hin, Win), image stride x_img
    const float* w;         // weights (Cout, Cin, ks, ks)
    const float* gamma;     // GroupNorm affine of THIS conv's output (Cout)
    const float* beta;
    int64_t x_img;
    int Cin, Hin, Win, stride, pad;
};

struct ConvGnArgs {
    ConvSrc a, b;           // b.x == nullptr: single conv
    const float* residual;  // (N, Cout, Hout, Wout) added after the normalisation, or nullptr
    float* y;
    int64_t y_img, r_img;
    int Cout, Hout, Wout, groups;
    int px_log2;            // PX = 1 << px_log2 lanes along pixels (power of two, <= 512)
    float eps;
    int relu;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red16) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red16[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CG_THREADS / 64; ++i) s += red16[i];
    return s;
}

// ---- weights of the block's group -> LDS ([CPG][K], the group's rows are contiguous in OIHW) ----
__device__ __forceinline__ void stage_weights(const float* __restrict__ wg, float* wl, int n) {
    if ((n & 3) == 0 && (((uintptr_t)wg) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < n; i += CG_THREADS * 4)
            *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(wg + i);
    } else {
        for (int i = threadIdx.x; i < n; i += CG_THREADS) wl[i] = wg[i];
    }
}

// Accumulate this thread's K slice of conv `S` for its PPT pixels x CPG channels; weights come from LDS (`wl`,
// [CPG][K]: broadcast 16-byte reads), input values are fetched a whole chunk at a time (32-36 loads in flight per
// thread) with the next chunk's loads issued before the current chunk's FMAs: the first version's load -> use ->
// load chain made these kernels pure latency (10-31 us each).
template <int KSZ, int CPG, int PPT>
__device__ __forceinline__ void conv_slice(const ConvSrc& S, int n, const float* wl, int Hout, int Wout, int PX, int pl,
                                           int ks, int nks, float (&acc)[PPT][CPG]) {
    const int HWo = Hout * Wout;
    const int K = S.Cin * KSZ * KSZ;
    const int cpk = (S.Cin + nks - 1) / nks;  // input channels per K slice
    const int c0 = ks * cpk, c1 = min(S.Cin, c0 + cpk);
    const float* __restrict__ xin = S.x + (int64_t)n * S.x_img;
    const int HWi = S.Hin * S.Win;
#pragma unroll
    for (int i = 0; i < PPT; ++i)
#pragma unroll
        for (int c = 0; c < CPG; ++c) acc[i][c] = 0.f;
    int oh[PPT], ow[PPT];
    bool live[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = pl + i * PX;
        live[i] = p < HWo;
        const int pc = live[i] ? p : 0;
        oh[i] = pc / Wout;
        ow[i] = pc - oh[i] * Wout;
    }
    if constexpr (KSZ == 1) {
        // channels per chunk: up to 32 loads in flight per thread (x2: the prefetched chunk), fewer beside wide accumulators
        constexpr int CH = (PPT == 1 ? 32 : (PPT == 2 ? 16 : 4)) / (CPG >= 32 ? 4 : (CPG >= 16 ? 2 : 1));
        int off[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) off[i] = live[i] ? (oh[i] * S.stride) * S.Win + ow[i] * S.stride : -1;
        auto fetch = [&](float (&xv)[CH][PPT], int ci) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < PPT; ++i)
                    xv[u][i] = (off[i] >= 0 && ci + u < c1) ? xin[(int64_t)(ci + u) * HWi + off[i]] : 0.f;
        };
        auto fmas = [&](const float (&xv)[CH][PPT], int ci) {
            if (ci + CH <= c1 && ((ci | K) & 3) == 0) {
#pragma unroll
                for (int u = 0; u < CH; u += 4)
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        const float4 wv = *reinterpret_cast<const float4*>(wl + c * K + ci + u);
#pragma unroll
                        for (int i = 0; i < PPT; ++i) {
                            acc[i][c] = fmaf(wv.x, xv[u][i], acc[i][c]);
                            acc[i][c] = fmaf(wv.y, xv[u + 1][i], acc[i][c]);
                            acc[i][c] = fmaf(wv.z, xv[u + 2][i], acc[i][c]);
                            acc[i][c] = fmaf(wv.w, xv[u + 3][i], acc[i][c]);
                        }
                    }
            } else {
                for (int u = 0; u < CH && ci + u < c1; ++u)
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        const float wv = wl[c * K + ci + u];
#pragma unroll
                        for (int i = 0; i < PPT; ++i) acc[i][c] = fmaf(wv, xv[u][i], acc[i][c]);
                    }
            }
        };
        float xa[CH][PPT], xb[CH][PPT];
        fetch(xa, c0);
        for (int ci = c0; ci < c1; ci += 2 * CH) {
            if (ci + CH < c1) fetch(xb, ci + CH);
            fmas(xa, ci);
            if (ci + CH < c1) {
                if (ci + 2 * CH < c1) fetch(xa, ci + 2 * CH);
                fmas(xb, ci + CH);
            }
        }
    } else if constexpr (KSZ == 3) {
        constexpr int CH = PPT == 1 ? 4 : 2;  // CH * 9 * PPT = 36 loads in flight
        int base[PPT];
        unsigned okm[PPT];  // bit t: tap t of this pixel is inside the image
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int ih0 = oh[i] * S.stride - S.pad, iw0 = ow[i] * S.stride - S.pad;
            base[i] = ih0 * S.Win + iw0;
            okm[i] = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ih = ih0 + t / 3, iw = iw0 + t % 3;
                if (live[i] && (unsigned)ih < (unsigned)S.Hin && (unsigned)iw < (unsigned)S.Win) okm[i] |= 1u << t;
            }
        }
        auto fetch = [&](float (&xv)[CH][9][PPT], int ci) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < PPT; ++i)
                        xv[u][t][i] = (((okm[i] >> t) & 1u) && ci + u < c1)
                                          ? xin[(int64_t)(ci + u) * HWi + base[i] + (t / 3) * S.Win + (t % 3)] : 0.f;
        };
        auto fmas = [&](const float (&xv)[CH][9][PPT], int ci) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (ci + u < c1) {
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        const float* wr = wl + c * K + (ci + u) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const float wv = wr[t];
#pragma unroll
                            for (int i = 0; i < PPT; ++i) acc[i][c] = fmaf(wv, xv[u][t][i], acc[i][c]);
                        }
                    }
                }
            }
        };
        float xa[CH][9][PPT], xb[CH][9][PPT];
        fetch(xa, c0);
        for (int ci = c0; ci < c1; ci += 2 * CH) {
            if (ci + CH < c1) fetch(xb, ci + CH);
            fmas(xa, ci);
            if (ci + CH < c1) {
                if (ci + 2 * CH < c1) fetch(xa, ci + 2 * CH);
                fmas(xb, ci + CH);
            }
        }
    } else {
        for (int ci = c0; ci < c1; ++ci) {
            const float* __restrict__ xc = xin + (int64_t)ci * HWi;
            for (int kh = 0; kh < KSZ; ++kh) {  // one tap row (KSZ * PPT loads) at a time
                float xv[KSZ][PPT];
#pragma unroll
                for (int kw = 0; kw < KSZ; ++kw)
#pragma unroll
                    for (int i = 0; i < PPT; ++i) {
                        const int ih = oh[i] * S.stride - S.pad + kh, iw = ow[i] * S.stride - S.pad + kw;
                        const bool ok = live[i] && (unsigned)ih < (unsigned)S.Hin && (unsigned)iw < (unsigned)S.Win;
                        xv[kw][i] = ok ? xc[ih * S.Win + iw] : 0.f;
                    }
#pragma unroll
                for (int kw = 0; kw < KSZ; ++kw)
#pragma unroll
                    for (int c = 0; c < CPG; ++c) {
                        const float wv = wl[c * K + (ci * KSZ + kh) * KSZ + kw];
#pragma unroll
                        for (int i = 0; i < PPT; ++i) acc[i][c] = fmaf(wv, xv[kw][i], acc[i][c]);
                    }
            }
        }
    }
}

constexpr int EPT_MAX = 16;  // tile elements per thread held in registers through statistics and normalisation

// Conv `S` -> this thread's EPT elements (e = tid + j * 512) of the finished [CPG x HWo] tile in `val`, and the
// tile's GroupNorm (mean, rstd).  K slices meet in LDS (`red`); statistics are two passes over REGISTERS.
template <int KSZ, int CPG, int PPT, bool UNI>
__device__ __forceinline__ void conv_tile_stats(const ConvSrc& S, const ConvGnArgs& A, int n, int g, float* wl, float* red,
                                                float* red16, float (&val)[EPT_MAX], float& mean, float& rstd) {
    const int PX = 1 << A.px_log2;
    const int HWo = A.Hout * A.Wout;
    const int nks = CG_THREADS >> A.px_log2;  // K slices
    const int tid = threadIdx.x;
    const int pl = tid & (PX - 1);
    int ks = tid >> A.px_log2;
    if (UNI) ks = __builtin_amdgcn_readfirstlane(ks);
    const int K = S.Cin * KSZ * KSZ;
    stage_weights(S.w + (int64_t)g * CPG * K, wl, CPG * K);
    __syncthreads();
    float acc[PPT][CPG];
    conv_slice<KSZ, CPG, PPT>(S, n, wl, A.Hout, A.Wout, PX, pl, ks, nks, acc);
    const int n_el = CPG * HWo;
    // partial (or, with one K slice, final) tiles -> LDS as red[slice][c][pixel]
    int sid = ks, nsl = nks;
    bool writer = true;
    if (!UNI) {  // several slices inside one wave: fold them by shuffles first, one partial per wave
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            float v = acc[0][c];
            for (int o = 32; o >= PX; o >>= 1) v += __shfl_xor(v, o);
            acc[0][c] = v;
        }
        sid = tid >> 6;
        nsl = CG_THREADS / 64;
        writer = (tid & 63) < PX;
    }
    if (writer) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = pl + i * PX;
            if (p < HWo) {
#pragma unroll
                for (int c = 0; c < CPG; ++c) red[(sid * CPG + c) * HWo + p] = acc[i][c];
            }
        }
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPT_MAX; ++j) {
        const int e = tid + j * CG_THREADS;
        float v = 0.f;
        if (e < n_el) {
            for (int sl = 0; sl < nsl; ++sl) v += red[sl * n_el + e];
        }
        val[j] = v;
        s += v;
    }
    mean = block_sum(s, red16) / (float)n_el;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < EPT_MAX; ++j) {
        const float d = (tid + j * CG_THREADS < n_el) ? val[j] - mean : 0.f;
        q += d * d;
    }
    rstd = rsqrtf(block_sum(q, red16) / (float)n_el + A.eps);
}

template <int KSZ, int CPG, int PPT, bool UNI, bool DUAL>
__global__ __launch_bounds__(CG_THREADS) void k_conv_gn(const ConvGnArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HWo = A.Hout * A.Wout;
    const int n = blockIdx.x / A.groups, g = blockIdx.x % A.groups;
    float* red16 = smem;       // 16 floats
    float* wl = smem + 16;     // CPG * max(K, K2) floats (16-byte aligned: 16 floats in front)
    const int Ka = A.a.Cin * KSZ * KSZ, Kb = DUAL ? A.b.Cin : 0;
    float* red = wl + ((CPG * max(Ka, Kb) + 3) & ~3);
    float va[EPT_MAX], vb[DUAL ? EPT_MAX : 1];
    float meanA, rstdA, meanB = 0.f, rstdB = 0.f;
    conv_tile_stats<KSZ, CPG, PPT, UNI>(A.a, A, n, g, wl, red, red16, va, meanA, rstdA);
    if constexpr (DUAL) {
        __syncthreads();  // everyone is done with wl / red of the first conv
        conv_tile_stats<1, CPG, PPT, UNI>(A.b, A, n, g, wl, red, red16, vb, meanB, rstdB);
    }
    float* yp = A.y + (int64_t)n * A.y_img + (int64_t)g * CPG * HWo;
    const float* rp = A.residual ? A.residual + (int64_t)n * A.r_img + (int64_t)g * CPG * HWo : nullptr;
#pragma unroll
    for (int j = 0; j < EPT_MAX; ++j) {
        const int e = threadIdx.x + j * CG_THREADS;
        if (e < CPG * HWo) {
            const int c = g * CPG + e / HWo;
            const float ga = A.a.gamma[c] * rstdA, be = A.a.beta[c] - meanA * ga;
            float v = fmaf(va[j], ga, be);
            if constexpr (DUAL) {
                const float g2 = A.b.gamma[c] * rstdB, b2 = A.b.beta[c] - meanB * g2;
                v += fmaf(vb[j], g2, b2);
            }
            if (rp) v += rp[e];
            if (A.relu) v = fmaxf(v, 0.f);
            yp[e] = v;
        }
    }
}

typedef void (*conv_gn_fn)(const ConvGnArgs);

// Instantiated envelope (register budget: PPT * CPG accumulators + two chunks of input values, 256 VGPRs at 512
// threads):  pixels/thread 1: any cpg;  2 (HWo 513..1024): cpg <= 8;  8 (HWo <= 4096): cpg 2, the 7x7 stem only.
// The two-conv (downsample) form exists for 1x1 main convs with cpg >= 8: the bottleneck's last conv.
template <int KSZ, int CPG, bool DUAL>
conv_gn_fn pick_ppt(int ppt, bool uni) {
    if (ppt == 1) return uni ? k_conv_gn<KSZ, CPG, 1, true, DUAL> : k_conv_gn<KSZ, CPG, 1, false, DUAL>;
    if constexpr (CPG <= 8 && KSZ != 7) {
        if (ppt == 2 && uni) return k_conv_gn<KSZ, CPG, 2, true, DUAL>;
    }
    if constexpr (CPG <= 2 && KSZ == 7 && !DUAL) {
        if (ppt <= 8 && uni) return k_conv_gn<KSZ, CPG, 8, true, DUAL>;
    }
    return nullptr;
}

template <int KSZ>
conv_gn_fn pick_cpg(int cpg, int ppt, bool uni, bool dual) {
    if (dual) {
        if constexpr (KSZ == 1) {
            switch (cpg) {
                case 8: return pick_ppt<1, 8, true>(ppt, uni);
                case 16: return pick_ppt<1, 16, true>(ppt, uni);
                case 32: return pick_ppt<1, 32, true>(ppt, uni);
                case 64: return pick_ppt<1, 64, true>(ppt, uni);
                default: return nullptr;
            }
        }
        return nullptr;
    }
    switch (cpg) {
        case 2: return pick_ppt<KSZ, 2, false>(ppt, uni);
        case 4: if constexpr (KSZ != 7) return pick_ppt<KSZ, 4, false>(ppt, uni); else return nullptr;
        case 8: if constexpr (KSZ != 7) return pick_ppt<KSZ, 8, false>(ppt, uni); else return nullptr;
        case 16: if constexpr (KSZ != 7) return pick_ppt<KSZ, 16, false>(ppt, uni); else return nullptr;
        case 32: if constexpr (KSZ == 1) return pick_ppt<KSZ, 32, false>(ppt, uni); else return nullptr;
        case 64: if constexpr (KSZ == 1) return pick_ppt<KSZ, 64, false>(ppt, uni); else return nullptr;
        default: return nullptr;
    }
}

}  // namespace

extern "C" {

int ivln_conv_gn_f32(const ivln_conv_gn_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->gamma || !d->beta || !d->y) return IVLN_E_INVALID;
    if (d->N <= 0 || d->groups <= 0 || d->Cout % d->groups || d->Cin <= 0) return IVLN_E_INVALID;
    if (d->ksize != 1 && d->ksize != 3 && d->ksize != 7) return IVLN_E_UNSUPPORTED;
    const int Hout = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wout = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
    const int HWo = Hout * Wout;
    if (HWo <= 0 || HWo > 4096) return IVLN_E_UNSUPPORTED;
    const int cpg = d->Cout / d->groups;
    int px_log2 = 0;
    while ((1 << px_log2) < HWo && px_log2 < 9) ++px_log2;
    const int PX = 1 << px_log2;
    const int ppt_need = (HWo + PX - 1) / PX;
    const int ppt = ppt_need == 1 ? 1 : (ppt_need == 2 ? 2 : 8);
    if (ppt_need > 8) return IVLN_E_UNSUPPORTED;
    const bool uni = PX >= 64;
    if (d->x2) {
        if (!d->w2 || !d->gamma2 || !d->beta2 || d->Cin2 <= 0) return IVLN_E_INVALID;
        const int H2 = (d->Hin2 - 1) / d->stride2 + 1, W2 = (d->Win2 - 1) / d->stride2 + 1;  // 1x1, pad 0
        if (H2 != Hout || W2 != Wout) return IVLN_E_INVALID;
    }
    const bool dual = d->x2 != nullptr;
    conv_gn_fn fn = d->ksize == 1 ? pick_cpg<1>(cpg, ppt, uni, dual)
                                  : d->ksize == 3 ? pick_cpg<3>(cpg, ppt, uni, dual) : pick_cpg<7>(cpg, ppt, uni, dual);
    if (!fn) return IVLN_E_UNSUPPORTED;
    const int nks = CG_THREADS / PX;
    const int nsl = uni ? nks : CG_THREADS / 64;
    if ((size_t)cpg * HWo > (size_t)16 * CG_THREADS) return IVLN_E_UNSUPPORTED;  // EPT_MAX elements per thread
    const int Ka = d->Cin * d->ksize * d->ksize, Kb = d->x2 ? d->Cin2 : 0;
    const size_t wfl = ((size_t)cpg * (Ka > Kb ? Ka : Kb) + 3) & ~(size_t)3;
    const size_t floats = 16 + wfl + (size_t)nsl * cpg * HWo;
    const size_t bytes = floats * sizeof(float);
    if (bytes > 160 * 1024) return IVLN_E_UNSUPPORTED;
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return IVLN_E_HIP;
    }
    ConvGnArgs A;
    A.a = ConvSrc{d->x, d->w, d->gamma, d->beta, d->x_img_stride > 0 ? d->x_img_stride : (int64_t)d->Cin * d->Hin * d->Win,
                  d->Cin, d->Hin, d->Win, d->stride, d->pad};
    A.b = ConvSrc{d->x2, d->w2, d->gamma2, d->beta2,
                  d->x2_img_stride > 0 ? d->x2_img_stride : (int64_t)d->Cin2 * d->Hin2 * d->Win2, d->Cin2, d->Hin2, d->Win2,
                  d->stride2, 0};
    A.residual = d->residual;
    A.y = d->y;
    A.y_img = d->y_img_stride > 0 ? d->y_img_stride : (int64_t)d->Cout * HWo;
    A.r_img = d->r_img_stride > 0 ? d->r_img_stride : (int64_t)d->Cout * HWo;
    A.Cout = d->Cout; A.Hout = Hout; A.Wout = Wout; A.groups = d->groups;
    A.px_log2 = px_log2;
    A.eps = d->eps;
    A.relu = d->relu;
    IVLN_LAUNCH_FAMILY(fn, dim3(d->N * d->groups), dim3(CG_THREADS), bytes, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

}  // extern "C"
