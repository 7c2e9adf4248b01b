// Shared pieces of the MFMA GEMM / direct-conv kernels: operand-mode enums and the fused epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ivln_hip.h"
#include "family_timing.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { AMODE_MK = 0, AMODE_KM = 1, AMODE_NCHW_P = 2 };
enum { BMODE_CONV = 0, BMODE_CONV1X1 = 1, BMODE_KN = 2, BMODE_NK = 3, BMODE_IM2COL_T = 4, BMODE_CONVT = 5,
       BMODE_CONV_K3 = 6, BMODE_CONV_K7 = 7,  // 3x3 / 7x7, dilation 1: (ci,kh,kw) by constant division, no tables
       BMODE_CONV_K2 = 8 };                   // 2x2 likewise (the stacked parity classes of a stride-2 3x3 transposed conv)
__host__ __device__ constexpr int conv_ks(int bmode) {
    return bmode == BMODE_CONV_K7 ? 7 : (bmode == BMODE_CONV_K3 ? 3 : (bmode == BMODE_CONV_K2 ? 2 : 0));
}
// input channels per chunk of the direct kernels (even: channel pairs fill the two k slots of the MFMA)
__host__ __device__ constexpr int conv_direct_ci(int KS) { return KS == 7 ? 2 : (KS == 2 ? 16 : 8); }
enum { DMODE_NCHW = 0, DMODE_DENSE = 1, DMODE_NCHW_UP2 = 2, DMODE_NCHW_UP2X4 = 3 };

// ivln_gemm_desc.img_run_flags: true when every image the tile's columns [n0, n0 + BN) belong to carries 0 (uniform over the
// workgroup; called before the first barrier)
__device__ __forceinline__ bool ivln_tile_skipped(const ivln_gemm_desc& p, int n0, int BN) {
    if (!p.img_run_flags) return false;
    const int last = min(p.N, n0 + BN) - 1;
    if (last < n0) return true;
    for (int i = n0 / p.HoWo; i <= last / p.HoWo; ++i)
        if (p.img_run_flags[i]) return false;
    return true;
}

constexpr int BK = 16;

// XCD-aware workgroup id (guide technique T1).  The dispatcher places workgroup b on XCD b % 8 and every XCD has
// a private 4 MB L2, so with the identity mapping neighbouring tiles - which share operand rows, or the halo of a
// convolution patch - sit on eight different L2s and each re-fetches the shared bytes from HBM.  The remap hands
// XCD x the x-th contiguous eighth of the tile ids (bijective for any grid size); tile -> result is unchanged.
struct BlockId {
    int x, y, z;
};
__device__ __forceinline__ BlockId xcd_block_id(int disabled) {
    BlockId b{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int gx = gridDim.x, gy = gridDim.y;
    const int nwg = gx * gy * (int)gridDim.z;
    if (disabled || nwg < 16) return b;
    const int orig = (b.z * gy + b.y) * gx + b.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int t = id / gx;
    b.x = id - t * gx;
    b.z = t / gy;
    b.y = t - b.z * gy;
    return b;
}

// weight set of the output tile whose first column is n0 (image-grouped convs, ivln_gemm_desc.grp_imgs)
__device__ __forceinline__ int tile_group(const ivln_gemm_desc& p, int n0) {
    return p.grp_imgs > 0 ? (n0 / p.HoWo) / p.grp_imgs : 0;
}

__device__ __forceinline__ void epilogue_store(const ivln_gemm_desc& p, int m, int n, float v) {
    int64_t addr;
    int me = m;  // index of the epilogue parameters: weight set g of image-grouped convs keeps them at [g*M + m]
    if (p.dmode == DMODE_NCHW) {
        int img = n / p.HoWo;
        int pp = n - img * p.HoWo;
        addr = ((int64_t)img * p.Ctot + m) * p.HoWo + pp;
        if (p.grp_imgs > 0) me = (img / p.grp_imgs) * p.M + m;
    } else if (p.dmode == DMODE_NCHW_UP2) {
        int img = n / p.HoWo;
        int pp = n - img * p.HoWo;
        int ho = pp / p.Wout, wo = pp - ho * p.Wout;
        addr = (((int64_t)img * p.Ctot + m) * (2 * p.Hout) + 2 * ho + (int)p.sDm) * (2 * p.Wout) + 2 * wo + (int)p.sDn;
    } else if (p.dmode == DMODE_NCHW_UP2X4) {  // four output-parity classes interleaved along M: m = 4*channel + cls
        const int cls = m & 3;
        me = m >> 2;
        int img = n / p.HoWo;
        int pp = n - img * p.HoWo;
        int ho = pp / p.Wout, wo = pp - ho * p.Wout;
        addr = (((int64_t)img * p.Ctot + me) * (2 * p.Hout) + 2 * ho + (cls >> 1)) * (2 * p.Wout) + 2 * wo + (cls & 1);
    } else {
        addr = (int64_t)m * p.sDm + (int64_t)n * p.sDn;
    }
    if (p.scale) v = fmaf(v, p.scale[me], p.shift[me]);
    else if (p.shift) v += p.shift[me];
    if (p.residual) v += p.residual[addr];
    if (p.accumulate) v += p.D[addr];
    if (p.relu) v = fmaxf(v, 0.f);
    p.D[addr] = v;
}

// One 32x32 accumulator tile -> D.  Lane (l31, half) holds column n = n_base + l31 and, in registers
// 4q..4q+3, the FOUR CONSECUTIVE rows m = m_base + 8q + 4*half + {0..3}.  When D is dense with unit row
// stride (nn.Linear outputs y[row][feature]: m is the contiguous index) those four go out as one 16-byte
// store instead of four 4-byte stores to the same 64-byte granule (the scalar form wrote every granule
// of a (102400 x 512) LSTM input projection 16 times over: 527 MB of WRITE_SIZE for a 210 MB tensor).
__device__ __forceinline__ void epilogue_tile(const ivln_gemm_desc& p, int m_base, int n, int half, const f32x16& acc,
                                              int split_z) {
    const bool to_ws = p.splits > 1 || p.defer_epilogue;
    const bool vec4 = !to_ws && p.dmode == DMODE_DENSE && p.sDm == 1 && (p.sDn & 3) == 0 && (p.M & 3) == 0 &&
                      (((uintptr_t)p.D | (uintptr_t)p.residual) & 15) == 0;
    if (n >= p.N) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m_base + 8 * q + 4 * half;
        if (vec4) {
            if (m >= p.M) continue;  // M % 4 == 0: the four rows are in or out together
            const int64_t addr = (int64_t)m + (int64_t)n * p.sDn;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[4 * q + i];
                if (p.scale) v[i] = fmaf(v[i], p.scale[m + i], p.shift[m + i]);
                else if (p.shift) v[i] += p.shift[m + i];
            }
            if (p.residual) {
                const float4 r = *reinterpret_cast<const float4*>(p.residual + addr);
                v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
            }
            if (p.accumulate) {
                const float4 r = *reinterpret_cast<const float4*>(p.D + addr);
                v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
            }
            if (p.relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            *reinterpret_cast<float4*>(p.D + addr) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (m + i < p.M) {
                    if (to_ws) p.ws[((int64_t)split_z * p.M + m + i) * p.N + n] = acc[4 * q + i];
                    else epilogue_store(p, m + i, n, acc[4 * q + i]);
                }
            }
        }
    }
}

// DMODE_NCHW_UP2X4 tile -> D through LDS.  Row m = 4*channel + 2*a + b of the GEMM holds output pixel
// (2*ho + a, 2*wo + b), so the four rows of a channel x two horizontally adjacent input pixels are 2 x 4 consecutive
// output floats: T ([BM][LDC] in LDS, rows m - m0, columns = the tile's pixels in an order where 2j and 2j+1 are
// horizontal neighbours) leaves as one 16-byte store per (channel, a, pixel pair) - whole 64-byte segments instead of
// the stride-2 4-byte stores of the MFMA layout, which touch every segment of the output twice.
// pix(nl, img, ho, wo) -> false when column nl of the tile is outside the problem.
template <int BM, int BN, int LDC, typename Pix>
__device__ __forceinline__ void up2x4_wide_store(const ivln_gemm_desc& p, const float* T, int m0, Pix pix) {
    constexpr int PAIRS = BN / 2;
    for (int f = threadIdx.x; f < (BM / 2) * PAIRS; f += 256) {
        const int j = f % PAIRS, r = f / PAIRS;   // r = 2*(channel of the tile) + a
        const int ml = 2 * r;                     // row of b = 0; b = 1 is the next one
        if (m0 + ml >= p.M) continue;             // M % 4 == 0: both rows are in or out together
        int img, ho, wo;
        if (!pix(2 * j, img, ho, wo)) continue;
        const int co = (m0 + ml) >> 2, a = r & 1;
        const int64_t addr = (((int64_t)img * p.Ctot + co) * (2 * p.Hout) + 2 * ho + a) * (2 * p.Wout) + 2 * wo;
        const float2 t0 = *reinterpret_cast<const float2*>(T + ml * LDC + 2 * j);
        const float2 t1 = *reinterpret_cast<const float2*>(T + (ml + 1) * LDC + 2 * j);
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
}
// DMODE_NCHW tile -> D through LDS: T ([BM][LDC], rows m - m0, columns = the tile's pixels in an order where 4j .. 4j+3
// are four horizontally adjacent pixels) leaves as one 16-byte store per (channel, pixel quad) instead of the MFMA
// layout's 4-byte stores (a lane holds four ROWS of one column).  pix(nl, img, pp) -> false when column nl of the
// tile is outside the problem; pp = pixel index inside the image.
template <int BM, int BN, int LDC, typename Pix>
__device__ __forceinline__ void nchw_wide_store(const ivln_gemm_desc& p, const float* T, int m0, Pix pix,
                                                float* stats = nullptr) {
    static_assert(BN == 128 && (BM * (BN / 4)) % 256 == 0, "a row of the tile = the 32 lanes of a half-wave");
    for (int idx = threadIdx.x; idx < BM * (BN / 4); idx += 256) {
        const int ml = idx / (BN / 4), c4 = idx - ml * (BN / 4);
        const int m = m0 + ml;
        int img = 0, pp = 0;
        const bool ok = m < p.M && pix(4 * c4, img, pp);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            v = *reinterpret_cast<const float4*>(T + ml * LDC + c4 * 4);
            const int64_t addr = ((int64_t)img * p.Ctot + m) * p.HoWo + pp;
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
        if (stats) {  // (uniform) {count, mean, M2} of the row's stored values: two half-wave reductions, like the two passes
                      // of k_bn_stats_partial, on values that are in registers anyway
            float cnt = ok ? 4.f : 0.f, sum = ok ? (v.x + v.y) + (v.z + v.w) : 0.f;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o), sum += __shfl_xor(sum, o);
            const float mean = cnt > 0.f ? sum / cnt : 0.f;
            const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
            float q = ok ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if ((threadIdx.x & 31) == 0 && m < p.M) {
                stats[m * 3 + 0] = cnt;
                stats[m * 3 + 1] = mean;
                stats[m * 3 + 2] = q;
            }
        }
    }
}

// (uniform) may this launch use up2x4_wide_store: even class-grid width so that pixel pairs share a row and every
// 16-byte store is aligned
__device__ __forceinline__ bool up2x4_wide_ok(const ivln_gemm_desc& p) {
    return p.dmode == DMODE_NCHW_UP2X4 && p.splits == 1 && !p.defer_epilogue && (p.Wout & 1) == 0 &&
           (((uintptr_t)p.D | (uintptr_t)p.residual) & 15) == 0 && !p.no_wide_epilogue;
}

}  // namespace

// conv_direct.hip: stride-1 3x3 / 7x7 convolution with the input patch and a weight slice staged in LDS.
// Returns IVLN_E_UNSUPPORTED when the shape is not eligible (the caller then uses the implicit GEMM).
int ivln_conv_direct_launch(ivln_gemm_desc& d, hipStream_t s);
// same file: direct weight gradient (A = dy NCHW, B = x gathered) of those convolutions
int ivln_wgrad_direct_launch(ivln_gemm_desc& d, hipStream_t s);
// gemm_vec.hip: float4-staged GEMM for contiguous operand modes (1x1 conv, linear); K tile of 32
bool ivln_gemm_vec_eligible(const ivln_gemm_desc& d);
int ivln_gemm_vec_launch(const ivln_gemm_desc& d, hipStream_t s, int tile);
// conv_bf3.hip: stride-1 3x3 / 7x7 conv with both operands as three bf16 pieces on the bf16 MFMA pipe (needs d.A_split)
int ivln_conv_bf3_launch(ivln_gemm_desc& d, hipStream_t s, bool force);
// same file: 7x7 weight gradient on the same arithmetic (d.split_ok)
int ivln_wgrad_bf3_launch(ivln_gemm_desc& d, hipStream_t s, bool force);
// conv1x1_stream.hip: short-K (64 / 128 / 256) 1x1 convs over many pixels, weights in registers; IVLN_E_UNSUPPORTED otherwise
int ivln_conv1x1_stream_launch(const ivln_gemm_desc& d, hipStream_t s, bool force);
