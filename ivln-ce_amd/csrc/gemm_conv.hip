// fp32 MFMA implicit-GEMM for gfx950: every GEMM-shaped op of the MapCMA hot path runs through
// this one kernel family - conv2d forward (7x7/3x3/1x1, any stride/pad), transposed conv (RedNet
// decoder), conv weight-gradient, linear forward / input-gradient / weight-gradient and Conv1d k=1.
//
//   D[m][n] = sum_k A[m][k] * B[k][n]        m = output channel, n = output pixel (or row)
//
// * v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain over k; 157 TFLOP/s peak): the
//   reference runs fp32 and parity is stated against fp32, so no reduced-precision path.
// * NCHW activations and OIHW weights are consumed in place: K is ordered (ci,kh,kw) exactly like the
//   OIHW flattening, and the im2col operand is gathered straight from NCHW through two small
//   per-layer tables (koff = ci*Hin*Win + kh*Win + kw, kpos = kh<<16|kw) - no layout conversion pass.
// * Block = 256 threads = 4 waves, one 32x32 accumulator tile per wave; wave grid WMxWN in
//   {2x2, 1x4, 4x1} -> block tile 64x64 / 32x128 / 128x32 picked per shape; BK = 16 staged through LDS
//   (k-major, +1 padded rows -> conflict-free ds_read_b32 operand fetches, <=2-way on stores) with
//   register prefetch of the next K tile while the MFMAs of the current one run.
// * split-K over blockIdx.z for the K-heavy / pixel-starved tail layers (4x4 spatial, K up to 9216):
//   partial slabs are reduced in a fixed order (deterministic) by k_splitk_epilogue.
// * Epilogue fused: per-channel scale/shift (folded BatchNorm or bias), residual add, ReLU.
#include <stdlib.h>

#include <vector>
#include <string>
#include <string.h>
#include <stdio.h>

#include "gemm_common.h"

namespace {

template <int WM, int WN, int TM, int TN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void k_gemm(const ivln_gemm_desc p) {
    // each wave owns TM x TN accumulator tiles of 32x32 (register tiling: TM+TN LDS operand reads feed
    // TM*TN MFMAs per k-pair, and the per-tile staging cost is amortised over 4x the MFMA work at 2x2)
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
    __shared__ float smem[BK * LDA_S + BK * LDB_S];
    float* As = smem;
    float* Bs = smem + BK * LDA_S;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    if (ivln_tile_skipped(p, n0, BN)) return;  // (optional per-image run flags: nothing to do for this tile)

    // K range of this split
    const int nk = (p.K + BK - 1) / BK;
    const int tps = (nk + p.splits - 1) / p.splits;
    const int kbeg = bid.z * tps * BK;
    const int kend = min(p.K, kbeg + tps * BK);

    // ---- per-thread load coordinates ----
    constexpr bool A_KFAST = (AMODE != AMODE_KM);
    constexpr bool B_KFAST = (BMODE == BMODE_NK || BMODE == BMODE_IM2COL_T);
    const int a_k = A_KFAST ? (t & 15) : (t / BM);
    const int a_m = A_KFAST ? (t >> 4) : (t % BM);
    const int b_k = B_KFAST ? (t & 15) : (t / BN);
    const int b_n = B_KFAST ? (t >> 4) : (t % BN);
    constexpr int A_STEP = A_KFAST ? 16 : 256 / BM;  // step of m (k-fast) or of k (m-fast) per element
    constexpr int B_STEP = B_KFAST ? 16 : 256 / BN;

    // conv-style B operand: this thread's output pixel is fixed (n-fast modes)
    int64_t pix_base = 0;
    int hi0 = 0, wi0 = 0;
    bool n_ok = false;
    if constexpr (BMODE == BMODE_CONV || BMODE == BMODE_CONV1X1 || BMODE == BMODE_CONVT || conv_ks(BMODE) != 0) {
        int n = n0 + b_n;
        n_ok = n < p.N;
        int nn = n_ok ? n : 0;
        int img = nn / p.HoWo;
        int pp = nn - img * p.HoWo;
        int ho = pp / p.Wout, wo = pp - ho * p.Wout;
        pix_base = (int64_t)img * p.in_img_stride;
        if constexpr (BMODE == BMODE_CONVT) {
            hi0 = ho + p.pad;
            wi0 = wo + p.pad;
        } else {
            hi0 = ho * p.stride - p.pad;
            wi0 = wo * p.stride - p.pad;
        }
    }
    // wgrad B operand: this thread's (ci,kh,kw) columns are fixed
    int w_koff[EB], w_kpos[EB];
    if constexpr (BMODE == BMODE_IM2COL_T) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int n = n0 + b_n + e * B_STEP;
            bool ok = n < p.N;
            w_koff[e] = ok ? p.koff[n] : 0;
            w_kpos[e] = ok ? p.kpos[n] : -1;
        }
    }

    float ra0[EA], rb0[EB], ra1[EA], rb1[EB];
    uint32_t ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;  // validity of each loaded element (bit e), applied by stage()
    static_assert(EA <= 32 && EB <= 32, "validity masks are 32 bits");
    const float* __restrict__ Ag = p.A + (int64_t)tile_group(p, n0) * p.a_grp_stride;  // this tile's weight set

    // Branch-free tile loads: every lane ALWAYS issues its load from a clamped (valid) address and the
    // out-of-range / padding case is a select - made in stage(), two compute phases later, from a validity bit
    // kept per element.  With the loads unconditional hipcc can count them and emits partial s_waitcnt vmcnt(N)
    // for the older register set instead of vmcnt(0); with no instruction that depends on the loaded value before
    // stage() it does not park a wait in front of the MFMA phase the loads are meant to fly under.
    auto load_tile = [&](int k0, float (&ra)[EA], float (&rb)[EB], uint32_t& ma, uint32_t& mb) {
        ma = 0, mb = 0;
        // ---------------- A ----------------
        if constexpr (AMODE == AMODE_MK) {
            const int k = k0 + a_k;
            const bool kok = k < kend;
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int m = m0 + a_m + e * A_STEP;
                const bool ok = kok && m < p.M;
                const float v = Ag[ok ? (int64_t)m * p.lda + k : 0];
                ra[e] = v, ma |= (uint32_t)ok << e;
            }
        } else if constexpr (AMODE == AMODE_KM) {
            const int m = m0 + a_m;
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int k = k0 + a_k + e * A_STEP;
                const bool ok = m < p.M && k < kend;
                const float v = p.A[ok ? (int64_t)k * p.lda + m : 0];
                ra[e] = v, ma |= (uint32_t)ok << e;
            }
        } else {  // AMODE_NCHW_P: A[m = channel][k = pixel] of an NCHW gradient tensor
            const int k = k0 + a_k;
            const bool kok = k < kend;
            const int kk = kok ? k : 0;
            const int img = kk / p.HoWo;
            const int pp = kk - img * p.HoWo;
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int m = m0 + a_m + e * A_STEP;
                const bool ok = kok && m < p.M;
                const float v = p.A[ok ? ((int64_t)img * p.M + m) * p.HoWo + pp : 0];
                ra[e] = v, ma |= (uint32_t)ok << e;
            }
        }
        // ---------------- B ----------------
        if constexpr (BMODE == BMODE_CONV) {
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = k0 + b_k + e * B_STEP;
                const bool kok = n_ok && k < kend;
                const int kc = kok ? k : 0;
                const int kp = p.kpos[kc];
                const int hi = hi0 + (kp >> 16) * p.dil, wi = wi0 + (kp & 0xFFFF) * p.dil;
                const bool ok = kok && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
                const float v = p.B[ok ? pix_base + p.koff[kc] + (int64_t)hi0 * p.Win + wi0 : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else if constexpr (conv_ks(BMODE) != 0) {
            // (ci,kh,kw) from constant divisions: no dependent table load on the per-tile critical path
            constexpr int KS = conv_ks(BMODE);
            const int HWin = p.Hin * p.Win;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = k0 + b_k + e * B_STEP;
                const int ci = k / (KS * KS), r = k - ci * (KS * KS);
                const int kh = r / KS, kw = r - kh * KS;
                const int hi = hi0 + kh, wi = wi0 + kw;
                const bool ok = n_ok && k < kend && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
                const float v = p.B[ok ? pix_base + (int64_t)ci * HWin + (int64_t)hi * p.Win + wi : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else if constexpr (BMODE == BMODE_CONV1X1) {
            const int64_t HWin = (int64_t)p.Hin * p.Win;
            const int64_t pbase = pix_base + (int64_t)hi0 * p.Win + wi0;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = k0 + b_k + e * B_STEP;
                const bool ok = n_ok && k < kend;
                const float v = p.B[ok ? pbase + (int64_t)k * HWin : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else if constexpr (BMODE == BMODE_CONVT) {
            // y[oh] += x[(oh + pad - kh)/stride] * w[kh] when divisible (nn.ConvTranspose2d)
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = k0 + b_k + e * B_STEP;
                const bool kok = n_ok && k < kend;
                const int kc = kok ? k : 0;
                const int kp = p.kpos[kc];
                const int th = hi0 - (kp >> 16), tw = wi0 - (kp & 0xFFFF);
                const int hi = th / p.stride, wi = tw / p.stride;
                const bool ok = kok && th >= 0 && tw >= 0 && hi * p.stride == th && wi * p.stride == tw &&
                                hi < p.Hin && wi < p.Win;
                const float v = p.B[ok ? pix_base + p.koff[kc] + (int64_t)hi * p.Win + wi : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else if constexpr (BMODE == BMODE_KN) {
            const int n = n0 + b_n;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = k0 + b_k + e * B_STEP;
                const bool ok = n < p.N && k < kend;
                const float v = p.B[ok ? (int64_t)k * p.ldb + n : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else if constexpr (BMODE == BMODE_NK) {
            const int k = k0 + b_k;
            const bool kok = k < kend;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int n = n0 + b_n + e * B_STEP;
                const bool ok = kok && n < p.N;
                const float v = p.B[ok ? (int64_t)n * p.ldb + k : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        } else {  // BMODE_IM2COL_T: B[k = output pixel][n = (ci,kh,kw)] gathered from the NCHW input
            const int k = k0 + b_k;
            const bool kok = k < kend;
            const int kk = kok ? k : 0;
            const int img = kk / p.HoWo;
            const int pp = kk - img * p.HoWo;
            const int ho = pp / p.Wout, wo = pp - ho * p.Wout;
            const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
            const int64_t base = (int64_t)img * p.in_img_stride + (int64_t)h0 * p.Win + w0;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int hi = h0 + (w_kpos[e] >> 16), wi = w0 + (w_kpos[e] & 0xFFFF);
                const bool ok = kok && w_kpos[e] >= 0 && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
                const float v = p.B[ok ? base + w_koff[e] : 0];
                rb[e] = v, mb |= (uint32_t)ok << e;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

    // Software pipeline with TWO K tiles in flight in two register sets.  The loop is unrolled by two so
    // that the sets swap roles without register copies: a copy (ra0 = ra1) made hipcc wait vmcnt(0) at
    // the loop top, i.e. the loads issued one iteration earlier had only ONE MFMA phase (~512 cycles) to
    // land and every tile stalled on L2/MALL latency (MFMA pipe 47 % busy on a 1024x4096x1024 GEMM).
    auto stage = [&](float (&ra)[EA], float (&rb)[EB], uint32_t ma, uint32_t mb) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const float v = (ma >> e) & 1 ? ra[e] : 0.f;
            if constexpr (A_KFAST) As[a_k * LDA_S + a_m + e * A_STEP] = v;
            else As[(a_k + e * A_STEP) * LDA_S + a_m] = v;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const float v = (mb >> e) & 1 ? rb[e] : 0.f;
            if constexpr (B_KFAST) Bs[b_k * LDB_S + b_n + e * B_STEP] = v;
            else Bs[(b_k + e * B_STEP) * LDB_S + b_n] = v;
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                a[tm] = As[(2 * kk + (lane >> 5)) * LDA_S + (wm * TM + tm) * 32 + (lane & 31)];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                b[tn] = Bs[(2 * kk + (lane >> 5)) * LDB_S + (wn * TN + tn) * 32 + (lane & 31)];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    if (kbeg < kend) {
        load_tile(kbeg, ra0, rb0, ma0, mb0);
        load_tile(kbeg + BK, ra1, rb1, ma1, mb1);  // past-the-end tiles load clamped addresses and select zeros
        for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
            stage(ra0, rb0, ma0, mb0);
            __syncthreads();
            load_tile(k0 + 2 * BK, ra0, rb0, ma0, mb0);
            compute();
            __syncthreads();
            if (k0 + BK >= kend) break;
            stage(ra1, rb1, ma1, mb1);
            __syncthreads();
            load_tile(k0 + 3 * BK, ra1, rb1, ma1, mb1);
            compute();
            __syncthreads();
        }
    }

    // ---- epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31 ----
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            epilogue_tile(p, m0 + (wm * TM + tm) * 32, n0 + (wn * TN + tn) * 32 + (lane & 31), lane >> 5, acc[tm][tn],
                          bid.z);
}

__global__ __launch_bounds__(256) void k_splitk_epilogue(const ivln_gemm_desc p) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * p.N) return;
    int m = (int)(idx / p.N), n = (int)(idx - (int64_t)m * p.N);
    float v = 0.f;
    for (int z = 0; z < p.splits; ++z) v += p.ws[((int64_t)z * p.M + m) * p.N + n];
    epilogue_store(p, m, n, v);
}

// The same reduction four columns at a time - NCHW destinations whose images hold a multiple of four pixels, dense ones
// with unit column stride (weight gradients) -: 16-byte loads from every slab and one 16-byte store.
__global__ __launch_bounds__(256) void k_splitk_epilogue4(const ivln_gemm_desc p) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N >> 2;
    if (idx >= (int64_t)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < p.splits; ++z) {  // fixed order: the sum does not depend on the launch
        const float4 w = *reinterpret_cast<const float4*>(p.ws + ((int64_t)z * p.M + m) * p.N + n);
        v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    int64_t addr;
    int me = m;
    if (p.dmode == DMODE_NCHW) {
        const int img = n / p.HoWo, pp = n - img * p.HoWo;
        addr = ((int64_t)img * p.Ctot + m) * p.HoWo + pp;
        if (p.grp_imgs > 0) me = (img / p.grp_imgs) * p.M + m;
    } else {  // DMODE_DENSE with unit column stride (weight gradients: D[m][n])
        addr = (int64_t)m * p.sDm + n;
    }
    if (p.scale) {
        const float sc = p.scale[me], sh = p.shift[me];
        v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
    } else if (p.shift) {
        const float sh = p.shift[me];
        v.x += sh, v.y += sh, v.z += sh, v.w += sh;
    }
    if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + addr);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
    }
    if (p.accumulate) {
        const float4 r = *reinterpret_cast<const float4*>(p.D + addr);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
    }
    if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<float4*>(p.D + addr) = v;
}

// (64 float4 columns x 4 slab groups per block: weight gradients come as up to 512 slabs of a SMALL matrix - 22 K floats
//  for the map CNN's first layer - so a thread per element left 21 blocks walking 382 slabs each; the four group sums are
//  added in group order through LDS)
__global__ __launch_bounds__(256) void k_splitk_epilogue_flat4(const ivln_gemm_desc p) {
    __shared__ float4 part[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 64 + col, MN = (int64_t)p.M * p.N;
    const bool ok = q * 4 < MN;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        for (int z = grp; z < p.splits; z += 4) {
            const float4 w = *reinterpret_cast<const float4*>(p.ws + (int64_t)z * MN + q * 4);
            v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
        }
    }
    part[grp][col] = v;
    __syncthreads();
    if (grp != 0 || !ok) return;
#pragma unroll
    for (int g = 1; g < 4; ++g) {
        const float4 w = part[g][col];
        v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + q * 4);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
    }
    if (p.accumulate) {
        const float4 r = *reinterpret_cast<const float4*>(p.D + q * 4);
        v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
    }
    if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<float4*>(p.D + q * 4) = v;
}

// reduce the split-K slabs of d into D with the fused epilogue
void launch_splitk_epilogue(const ivln_gemm_desc& d, hipStream_t s) {
    // (a 32 x 32 LDS-transposing form for dense outputs with unit ROW stride - Linear activations - was measured: 10.4 us
    //  per call against 7 us for the element-per-thread form on the update's shapes; not kept)
    // fully contiguous dense outputs without per-row parameters (weight gradients): the matrix as one flat vector, so
    // that N need not be a multiple of four (the map CNN's first layer: N = 14 x 49 = 686)
    if (d.dmode == DMODE_DENSE && d.sDn == 1 && d.sDm == d.N && (((int64_t)d.M * d.N) & 3) == 0 && !d.scale && !d.shift &&
        !d.no_wide_epilogue && (((uintptr_t)d.D | (uintptr_t)d.residual | (uintptr_t)d.ws) & 15) == 0) {
        const int64_t total4 = (int64_t)d.M * d.N / 4;
        hipLaunchKernelGGL(k_splitk_epilogue_flat4, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, d);
        return;
    }
    const bool lay = (d.dmode == DMODE_NCHW && (d.HoWo & 3) == 0) || (d.dmode == DMODE_DENSE && d.sDn == 1 && (d.sDm & 3) == 0);
    const bool vec4 = lay && (d.N & 3) == 0 && !d.no_wide_epilogue &&
                      (((uintptr_t)d.D | (uintptr_t)d.residual | (uintptr_t)d.ws) & 15) == 0;
    if (vec4) {
        const int64_t total = (int64_t)d.M * (d.N >> 2);
        hipLaunchKernelGGL(k_splitk_epilogue4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
    } else {
        const int64_t total = (int64_t)d.M * d.N;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
    }
}

template <int WM, int WN, int TM, int TN>
int launch_tile(const ivln_gemm_desc& d, hipStream_t s) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.splits);
    dim3 block(256);
#define IVLN_CASE(AM, BMD)                                                              \
    if (d.amode == AM && d.bmode == BMD) {                                              \
        IVLN_LAUNCH_FAMILY((k_gemm<WM, WN, TM, TN, AM, BMD>), grid, block, 0, s, d);            \
        return IVLN_OK;                                                                 \
    }
    IVLN_CASE(AMODE_MK, BMODE_CONV)
    IVLN_CASE(AMODE_MK, BMODE_CONV1X1)
    IVLN_CASE(AMODE_MK, BMODE_CONV_K3)
    IVLN_CASE(AMODE_MK, BMODE_CONV_K7)
    IVLN_CASE(AMODE_MK, BMODE_CONV_K2)
    IVLN_CASE(AMODE_MK, BMODE_CONVT)
    IVLN_CASE(AMODE_MK, BMODE_NK)
    IVLN_CASE(AMODE_MK, BMODE_KN)
    IVLN_CASE(AMODE_KM, BMODE_NK)
    IVLN_CASE(AMODE_KM, BMODE_KN)
    IVLN_CASE(AMODE_NCHW_P, BMODE_IM2COL_T)
#undef IVLN_CASE
    return IVLN_E_UNSUPPORTED;
}

}  // namespace

// ---- duration sink of the family's launches (family_timing.h).  One user at a time (bench.py's instrumented pass). ----
namespace {
// every kernel that launches through IVLN_LAUNCH_FAMILY: the definition of "the MFMA family" (ivln_family_kernel_names)
const char* const kFamilyKernels[] = {"k_gemm",     "k_gemm_vec",    "k_conv_direct",    "k_wgrad_direct", "k_conv1x1_stream", "k_conv_bf3",
                                      "k_conv_bf3_ks", "k_conv1x1_bf3_ks", "k_conv7s2_bf3", "k_wgrad_bf3",    "k_gn_conv",        "k_nconv",
                                      "k_depth_net"};
constexpr int kFamilyCount = (int)(sizeof(kFamilyKernels) / sizeof(kFamilyKernels[0]));
std::vector<hipEvent_t> g_timing_events;
std::vector<int> g_timing_kernel;  // per timed launch: index into kFamilyKernels, kFamilyCount = a name outside the list
int g_timing_used = -1;            // -1: not armed
int g_timing_dropped = 0;
double g_report_ms[kFamilyCount + 1];
int g_report_n[kFamilyCount + 1];
bool g_report_valid = false;

int family_index(const char* site) {  // "(k_conv_direct<3, 32, ...>)" / "k_depth_net" -> index of its template name
    while (*site == '(' || *site == ' ') ++site;
    size_t n = 0;
    while (site[n] && site[n] != '<' && site[n] != ')' && site[n] != ' ') ++n;
    for (int i = 0; i < kFamilyCount; ++i)
        if (strlen(kFamilyKernels[i]) == n && !strncmp(kFamilyKernels[i], site, n)) return i;
    return kFamilyCount;
}
}  // namespace

bool ivln_family_timing_next(hipEvent_t* start, hipEvent_t* stop, const char* kernel) {
    if (g_timing_used < 0) return false;
    if ((size_t)g_timing_used + 2 > g_timing_events.size()) {
        ++g_timing_dropped;
        return false;
    }
    *start = g_timing_events[g_timing_used];
    *stop = g_timing_events[g_timing_used + 1];
    g_timing_kernel[g_timing_used / 2] = family_index(kernel);
    g_timing_used += 2;
    return true;
}

extern "C" const char* ivln_family_kernel_names(void) {
    static std::string names;
    if (names.empty())
        for (int i = 0; i < kFamilyCount; ++i) names += std::string(i ? "," : "") + kFamilyKernels[i];
    return names.c_str();
}

extern "C" int ivln_family_timing_begin(int max_launches) {
    if (max_launches <= 0 || g_timing_used >= 0) return IVLN_E_INVALID;
    while (g_timing_events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return IVLN_E_HIP;
        g_timing_events.push_back(e);
    }
    g_timing_kernel.assign(g_timing_events.size() / 2, kFamilyCount);
    g_timing_used = 0;
    g_timing_dropped = 0;
    g_report_valid = false;
    return IVLN_OK;
}

extern "C" int ivln_family_timing_end(double* total_ms, int* launches, int* dropped) {
    if (g_timing_used < 0 || !total_ms || !launches) return IVLN_E_INVALID;
    const int used = g_timing_used;
    g_timing_used = -1;
    double sum = 0.0;
    for (int k = 0; k <= kFamilyCount; ++k) g_report_ms[k] = 0.0, g_report_n[k] = 0;
    for (int i = 0; i < used; i += 2) {
        if (hipEventSynchronize(g_timing_events[i + 1]) != hipSuccess) return IVLN_E_HIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_timing_events[i], g_timing_events[i + 1]) != hipSuccess) return IVLN_E_HIP;
        sum += ms;
        g_report_ms[g_timing_kernel[i / 2]] += ms;
        g_report_n[g_timing_kernel[i / 2]] += 1;
    }
    g_report_valid = true;
    *total_ms = sum;
    *launches = used / 2;
    if (dropped) *dropped = g_timing_dropped;
    return IVLN_OK;
}

extern "C" int ivln_family_timing_report(char* buf, int cap) {
    if (!g_report_valid) return IVLN_E_INVALID;
    std::string out;
    char line[128];
    for (int k = 0; k <= kFamilyCount; ++k) {
        if (!g_report_n[k]) continue;
        snprintf(line, sizeof line, "%s %d %.6f\n", k < kFamilyCount ? kFamilyKernels[k] : "?", g_report_n[k], g_report_ms[k]);
        out += line;
    }
    const int need = (int)out.size() + 1;
    if (!buf || cap < need) return need;
    memcpy(buf, out.c_str(), (size_t)need);
    return 0;
}

extern "C" int ivln_gemm_f32(const ivln_gemm_desc* desc, void* stream) {
    if (!desc || !desc->A || !desc->B || !desc->D || desc->M <= 0 || desc->N <= 0 || desc->K <= 0)
        return IVLN_E_INVALID;
    ivln_gemm_desc d = *desc;
    hipStream_t s = (hipStream_t)stream;
    static const bool no_xcd_env = getenv("IVLN_NO_XCD_REMAP") != nullptr;  // A/B switch
    if (no_xcd_env) d.no_xcd_remap = 1;
    if (d.HoWo <= 0) d.HoWo = 1;
    if (d.dil <= 0) d.dil = 1;
    if (d.stat_tiles) *d.stat_tiles = 0;  // (only the direct conv's wide epilogue produces statistics and says so)
    if (d.dmode == DMODE_NCHW_UP2X4 && (d.M & 3)) return IVLN_E_INVALID;
    if (d.Ctot <= 0) d.Ctot = d.dmode == DMODE_NCHW_UP2X4 ? d.M / 4 : d.M;
    if (d.in_img_stride <= 0) d.in_img_stride = (int64_t)d.Cin * d.Hin * d.Win;
    if (d.grp_imgs > 0) {  // image-grouped weights: forward convs into an NCHW destination only
        if (d.amode != AMODE_MK || d.dmode != DMODE_NCHW || d.N % ((int64_t)d.grp_imgs * d.HoWo) != 0) return IVLN_E_INVALID;
        if (d.a_grp_stride <= 0) d.a_grp_stride = (int64_t)d.M * d.lda;
    }
    if (d.img_run_flags) {  // per-image run flags: the two GEMM kernels below honour them, one slab, no deferred epilogue
        if (d.dmode != DMODE_NCHW || d.defer_epilogue || d.N % d.HoWo != 0 || d.tile_override == 6 || d.tile_override == 8 ||
            d.tile_override == 9)
            return IVLN_E_INVALID;
        d.splits = 1;
        d.A_split = nullptr;
        if (d.tile_override == 0) d.tile_override = ivln_gemm_vec_eligible(d) ? 7 : 1;
    }
    // residual behind the ReLU: only the stride-1 1x1 split-bf16 kernels have that epilogue form
    if (d.residual_after_relu) {
        if (!d.A_split || !d.residual || d.defer_epilogue || d.dmode != DMODE_NCHW || d.fuse_A_split || d.accumulate) return IVLN_E_UNSUPPORTED;
        const int rc = ivln_conv_bf3_launch(d, s, true);
        if (rc != IVLN_OK) return rc;
        if (d.splits_used) *d.splits_used = 1;
        return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
    }
    // a fused bottleneck tail (fuse_*): only conv_bf3.hip runs it; anything else would silently drop the 1x1 half
    if (d.fuse_A_split) {
        if (!d.A_split || d.defer_epilogue || d.dmode != DMODE_NCHW) return IVLN_E_UNSUPPORTED;
        const int rc = ivln_conv_bf3_launch(d, s, true);
        if (rc != IVLN_OK) return rc;
        if (d.splits_used) *d.splits_used = 1;
        return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
    }
    // stride-1 3x3 / 7x7 with split-bf16 weights on hand: the bf16-MFMA direct conv (conv_bf3.hip); 9 insists on it
    if ((d.tile_override == 0 || d.tile_override >= 9) && d.A_split) {
        const int splits_asked = d.splits;
        float* const stats_asked = d.stat_partials;
        const int rc = ivln_conv_bf3_launch(d, s, d.tile_override >= 9);
        if (rc == IVLN_OK) {
            if (d.splits_used) *d.splits_used = d.splits;
            if (d.splits > 1) launch_splitk_epilogue(d, s);
            return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
        }
        d.splits = splits_asked;  // (not taken: the fp32 kernels decide for themselves)
        d.stat_partials = stats_asked;
        if (rc != IVLN_E_UNSUPPORTED || d.tile_override >= 9) return rc;
    } else if ((d.tile_override == 0 || d.tile_override == 9) && d.bmode == BMODE_IM2COL_T) {
        const int splits_asked = d.splits;
        const int rc = ivln_wgrad_bf3_launch(d, s, d.tile_override == 9);
        if (rc == IVLN_OK) {
            if (d.splits_used) *d.splits_used = d.splits;
            launch_splitk_epilogue(d, s);  // (the kernel leaves raw slabs even when there is one)
            return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
        }
        d.splits = splits_asked;
        if (rc != IVLN_E_UNSUPPORTED || d.tile_override == 9) return rc;
    } else if (d.tile_override >= 9) {
        return IVLN_E_UNSUPPORTED;
    }
    // stride-1 3x3 / 7x7: LDS-staged direct convolution (conv_direct.hip); tile_override 1..5 pins the
    // implicit GEMM tiles, 6 insists on the direct kernel
    if (d.tile_override == 0 || d.tile_override == 6) {
        int rc = ivln_conv_direct_launch(d, s);
        if (rc == IVLN_E_UNSUPPORTED) rc = ivln_wgrad_direct_launch(d, s);
        if (rc == IVLN_OK) {
            if (d.splits_used) *d.splits_used = d.splits;
            if (d.splits > 1 && !d.defer_epilogue) {
                launch_splitk_epilogue(d, s);
            }
            return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
        }
        if (rc != IVLN_E_UNSUPPORTED || d.tile_override == 6) return rc;
    }
    // short-K 1x1 convs over many pixels: weights in registers, activations streamed (conv1x1_stream.hip)
    if ((d.tile_override == 0 || d.tile_override == 8) && d.splits <= 1) {  // (tile_override 8 insists on it: tests)
        const int rc = ivln_conv1x1_stream_launch(d, s, d.tile_override == 8);
        if (rc == IVLN_OK) {
            if (d.splits_used) *d.splits_used = 1;
            return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
        }
        if (rc != IVLN_E_UNSUPPORTED || d.tile_override == 8) return rc;
    }
    // tile shape: channel-starved -> 32x128, pixel-starved -> 128x32; otherwise 64x128 when that still
    // gives >= 2 blocks per CU (512 blocks), else 64x64
    int tile = 0;
    if (d.M <= 32) tile = 1;
    else if (d.N <= 32) tile = 2;
    else {
        auto nblocks = [&](int bm, int bn) { return (int64_t)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
        // (the 128x128 register-tiled variant measured slower than 64x128 on every conv / GEMM shape of
        //  the path - it runs at one block per CU - and is only reachable through tile_override)
        if (nblocks(64, 128) >= 512) tile = 4;
    }
    if (d.tile_override > 0 && d.tile_override <= 5) tile = d.tile_override - 1;
    constexpr int vec_tile_env = -1;  // tuning
    // contiguous operand modes (1x1 conv, linear): float4-staged kernel with 32-deep K tiles
    // (gemm_vec.hip); tile_override 7 insists on it, 1..5 pin the scalar-gather kernel
    const bool vec = (d.tile_override == 0 || d.tile_override == 7) && ivln_gemm_vec_eligible(d);
    if (d.tile_override == 7 && !vec) return IVLN_E_UNSUPPORTED;
    const int bk = vec ? 32 : BK;
    // the float4-staged kernel is fastest at 64x64 (~100 VGPRs: four waves per SIMD; 64x128 and 128x128
    // measured 90 / 79 vs 94 TFLOP/s on 1024x4096x1024 and lose more on the small 1x1 convs)
    if (vec && tile != 1 && tile != 2) tile = vec_tile_env >= 0 ? vec_tile_env : 0;
    if (d.grp_imgs > 0) {  // an output tile must lie inside one weight group
        const int64_t gp = (int64_t)d.grp_imgs * d.HoWo;
        const int bn = tile == 1 ? 128 : (tile == 2 ? 32 : (tile == 3 || tile == 4 ? 128 : 64));
        if (gp % bn != 0) {
            if (gp % 64 == 0) tile = 0;
            else if (gp % 32 == 0) tile = 2;
            else return IVLN_E_UNSUPPORTED;
        }
    }
    const int BM = tile == 1 ? 32 : (tile == 2 || tile == 3 ? 128 : 64);
    const int BN = tile == 1 ? 128 : (tile == 2 ? 32 : (tile == 3 || tile == 4 ? 128 : 64));
    // split-K when the output grid cannot fill the chip and K is deep
    int64_t blocks = (int64_t)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    int nk = (d.K + bk - 1) / bk;
    int splits = 1;
    if (d.defer_epilogue && (!d.ws || d.ws_floats < (int64_t)d.M * d.N)) return IVLN_E_INVALID;
    if (d.splits == 0) {
        // the consumer reduces for free when deferred
        const int min_tiles = d.defer_epilogue ? 2 : 4;
        constexpr int want_env = 0;  // tuning
        // deferred: one block per CU measured as fast as four (3.50-3.54 K env-steps/s for 192..2048) at a
        // quarter of the slab bytes the consumer has to read back
        constexpr int want_nd_env = 0;
        const int64_t want = want_env > 0 ? want_env
                             : (d.defer_epilogue ? 256 : (want_nd_env > 0 ? want_nd_env : 512));
        constexpr int below_env = 0;  // tuning
        const int64_t below = (below_env > 0 && !d.defer_epilogue) ? below_env : 256;
        if (d.ws && blocks < below && nk >= 2 * min_tiles) {
            splits = (int)((want + blocks - 1) / blocks);
            if (splits > nk / min_tiles) splits = nk / min_tiles;
            // weight gradients reduce over millions of pixels with a tiny M x N: allow deep splits there
            // (deep K with a small M x N - the LSTM / Conv1d weight gradients of the update, K = 40 960 rows: 16 splits left
            //  128 blocks on 256 CUs, 167 us for 2 GFLOP)
            constexpr int deep_env = 64;  // tuning
            const int max_splits = d.defer_epilogue ? (blocks <= 8 ? 64 : (blocks <= 32 ? 32 : 16))
                                                    : (nk >= 4096 ? 256 : (nk >= 512 ? deep_env : 16));
            if (splits > max_splits) splits = max_splits;
            int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
    } else {
        splits = d.splits;
        if (splits > 1 && (!d.ws || d.ws_floats < (int64_t)splits * d.M * d.N)) return IVLN_E_INVALID;
    }
    // drop empty trailing splits
    int tps = (nk + splits - 1) / splits;
    splits = (nk + tps - 1) / tps;
    d.splits = splits;
    int rc;
    if (vec) rc = ivln_gemm_vec_launch(d, s, tile);
    else switch (tile) {
        case 1: rc = launch_tile<1, 4, 1, 1>(d, s); break;
        case 2: rc = launch_tile<4, 1, 1, 1>(d, s); break;
        case 3: rc = launch_tile<2, 2, 2, 2>(d, s); break;
        case 4: rc = launch_tile<2, 2, 1, 2>(d, s); break;
        default: rc = launch_tile<2, 2, 1, 1>(d, s); break;
    }
    if (rc != IVLN_OK) return rc;
    if (d.splits_used) *d.splits_used = splits;
    if (splits > 1 && !d.defer_epilogue) {
        launch_splitk_epilogue(d, s);
    }
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}
