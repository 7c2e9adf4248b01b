// Vector-load fp32 MFMA GEMM for the operand modes whose global layout is contiguous along one GEMM index
// (1x1 convolutions over NCHW, nn.Linear forward / input-gradient / weight-gradient, Conv1d k=1):
//
//   D[m][n] = sum_k A[m][k] B[k][n]     A: [m][k] (k contiguous) or [k][m];  B: [k][n] (n contiguous) or [n][k]
//
// against k_gemm (gemm_conv.hip), which stages 16-deep K tiles with one scalar load + address computation
// per element:
//   * every global load is a float4 along the operand's contiguous index and goes to LDS as one
//     ds_write_b128; K tiles are 32 deep (half the barriers, twice the MFMA work per staged tile);
//   * an operand that is k-contiguous keeps its [row][k] order in LDS (row stride 36 words: conflict-free
//     b128) and feeds the MFMA with ds_read_b128 - the two k slots of v_mfma_f32_32x32x2_f32 take
//     k = j and k = 16 + j, so one lane's 16 operand values of a tile are 4 consecutive float4;
//   * an operand that is row-contiguous is stored [k][row] and read with ds_read_b32 base + immediate.
//   Either way the MFMA loop has no address arithmetic.
// Tiles, split-K slabs, the deferred mode and the fused epilogue are those of ivln_gemm_f32.
#include <stdlib.h>

#include "gemm_common.h"

namespace {

constexpr int BKV = 32;
enum { LAY_K = 0, LAY_R = 1 };

// BS2 (B = a 1x1 conv read at stride 2, RedNet's down-sampling shortcuts): n and n + 1 are input columns 2*wo and
// 2*wo + 2, so a 16-byte load yields TWO operand elements (.x and .z) instead of four - twice the loads per tile, the
// same staging and MFMA loop, in place of the scalar-gather GEMM these convs ran on (42 -> ~70 TFLOP/s).
template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, bool BS2 = false>
__global__ __launch_bounds__(256, 2) void k_gemm_vec(const ivln_gemm_desc p) {
    static_assert(!BS2 || BLAY == LAY_R, "stride-2 pixels are the n-contiguous layout");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int LDK = BKV + 4;  // [row][k] layout row stride (words)
    constexpr int A_WORDS = ALAY == LAY_K ? BM * LDK : BKV * BM;
    constexpr int B_WORDS = BLAY == LAY_K ? BN * LDK : BKV * BN;
    constexpr int EA = BM * BKV / 1024, EB = BN * BKV / 1024 * (BS2 ? 2 : 1);  // 16-byte loads per thread per tile
    // one LDS block: the operand tiles during the K loop, the output tile during the wide epilogue
    constexpr int LDC = BN + 4;
    constexpr bool WIDE_FITS = BM * LDC <= A_WORDS + B_WORDS;
    __shared__ __attribute__((aligned(16))) float smem[A_WORDS + B_WORDS];
    float* const As = smem;
    float* const Bs = smem + A_WORDS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int m0 = bid.y * BM, n0 = bid.x * BN;
    if (ivln_tile_skipped(p, n0, BN)) return;  // (optional per-image run flags: nothing to do for this tile)
    const int nk = (p.K + BKV - 1) / BKV;
    const int tps = (nk + p.splits - 1) / p.splits;
    const int kbeg = bid.z * tps * BKV;
    const int kend = min(p.K, kbeg + tps * BKV);

    // ---- per-thread staging coordinates (tile-invariant) ----
    // [row][k] operands: 8 float4 per row, thread -> (row = t/8 + 32 e, kq = t%8)
    // [k][row] operands: R4 = rows/4 float4 per k, thread -> (k = t/R4 + (256/R4) e, rq = t%R4)
    constexpr int AR4 = BM / 4, BR4 = BS2 ? BN / 2 : BN / 4, BNV = BS2 ? 2 : 4;  // (BNV: columns one B load yields)
    int64_t a_off[EA], b_off[EB];
    bool a_ok[EA], b_ok[EB];
    int a_k[EA], b_k[EB];  // k offset of the element inside the tile
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        if constexpr (ALAY == LAY_K) {
            const int m = m0 + (t >> 3) + 32 * e;
            a_k[e] = (t & 7) * 4;
            a_ok[e] = m < p.M;
            a_off[e] = a_ok[e] ? (int64_t)m * p.lda + a_k[e] : 0;
        } else {
            const int m = m0 + (t % AR4) * 4;
            a_k[e] = t / AR4 + (256 / AR4) * e;
            a_ok[e] = m < p.M;
            a_off[e] = a_ok[e] ? (int64_t)a_k[e] * p.lda + m : 0;
        }
    }
    const bool bconv = p.bmode == BMODE_CONV1X1;
    const int64_t brow = BS2 ? (int64_t)p.Hin * p.Win : (bconv ? (int64_t)p.HoWo : p.ldb);  // stride of k for [k][n] operands
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        if constexpr (BLAY == LAY_K) {
            const int n = n0 + (t >> 3) + 32 * e;
            b_k[e] = (t & 7) * 4;
            b_ok[e] = n < p.N;
            b_off[e] = b_ok[e] ? (int64_t)n * p.ldb + b_k[e] : 0;
        } else {
            const int n = n0 + (t % BR4) * BNV;
            b_k[e] = t / BR4 + (256 / BR4) * e;
            b_ok[e] = n < p.N;
            int64_t base = 0;
            if (b_ok[e]) {
                if constexpr (BS2) {
                    const int img = n / p.HoWo, pp = n - img * p.HoWo;
                    const int ho = pp / p.Wout, wo = pp - ho * p.Wout;
                    base = (int64_t)img * p.in_img_stride + (int64_t)(2 * ho) * p.Win + 2 * wo;
                } else if (bconv) {
                    const int img = n / p.HoWo;
                    base = (int64_t)img * p.in_img_stride + (n - img * p.HoWo);
                } else {
                    base = n;
                }
            }
            b_off[e] = base + (int64_t)b_k[e] * brow;
        }
    }
    const int64_t a_kstep = ALAY == LAY_K ? 1 : p.lda;
    const float* __restrict__ Ag = p.A + (int64_t)tile_group(p, n0) * p.a_grp_stride;  // this tile's weight set
    const int64_t b_kstep = BLAY == LAY_K ? 1 : brow;

    float4 ra0[EA], rb0[EB], ra1[EA], rb1[EB];
    // Tile loads are unconditional from a clamped (valid) address and carry NO dependent instruction: the zero for
    // an out-of-range row / k is selected when the registers are staged, two compute phases later.  (With the select
    // next to the load hipcc parks an s_waitcnt vmcnt(0) in front of the MFMA phase that follows - every tile then
    // waits out the full latency of the loads it has just issued.)
    auto load_tile = [&](int k0, float4 (&ra)[EA], float4 (&rb)[EB]) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const bool ok = a_ok[e] && k0 + a_k[e] < kend;
            ra[e] = *reinterpret_cast<const float4*>(Ag + (ok ? a_off[e] + (int64_t)k0 * a_kstep : 0));
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const bool ok = b_ok[e] && k0 + b_k[e] < kend;
            rb[e] = *reinterpret_cast<const float4*>(p.B + (ok ? b_off[e] + (int64_t)k0 * b_kstep : 0));
        }
    };
    // (component-wise selects: `ok ? v : zero4` on float4 lvalues became a pointer select through scratch)
    auto stage = [&](int k0, float4 (&ra)[EA], float4 (&rb)[EB]) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const bool ok = a_ok[e] && k0 + a_k[e] < kend;
            const float4 v = make_float4(ok ? ra[e].x : 0.f, ok ? ra[e].y : 0.f, ok ? ra[e].z : 0.f, ok ? ra[e].w : 0.f);
            if constexpr (ALAY == LAY_K) *reinterpret_cast<float4*>(&As[((t >> 3) + 32 * e) * LDK + (t & 7) * 4]) = v;
            else *reinterpret_cast<float4*>(&As[(t / AR4 + (256 / AR4) * e) * BM + (t % AR4) * 4]) = v;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const bool ok = b_ok[e] && k0 + b_k[e] < kend;
            const float4 v = make_float4(ok ? rb[e].x : 0.f, ok ? rb[e].y : 0.f, ok ? rb[e].z : 0.f, ok ? rb[e].w : 0.f);
            if constexpr (BLAY == LAY_K) *reinterpret_cast<float4*>(&Bs[((t >> 3) + 32 * e) * LDK + (t & 7) * 4]) = v;
            else if constexpr (BS2) *reinterpret_cast<float2*>(&Bs[(t / BR4 + (256 / BR4) * e) * BN + (t % BR4) * 2]) = make_float2(v.x, v.z);
            else *reinterpret_cast<float4*>(&Bs[(t / BR4 + (256 / BR4) * e) * BN + (t % BR4) * 4]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

    // k slot `half` of MFMA step j holds k = 16*half + j; operands are fetched 8 steps at a time (two
    // float4 per k-contiguous operand) to keep the 128x128 tile under 256 VGPRs
    auto compute = [&]() {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int ml = (wm * TM + tm) * 32 + l31;
                if constexpr (ALAY == LAY_K) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float4 v = *reinterpret_cast<const float4*>(&As[ml * LDK + half * (BKV / 2) + 8 * c + 4 * i]);
                        a[tm][4 * i] = v.x, a[tm][4 * i + 1] = v.y, a[tm][4 * i + 2] = v.z, a[tm][4 * i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[tm][j] = As[(half * (BKV / 2) + 8 * c + j) * BM + ml];
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nl = (wn * TN + tn) * 32 + l31;
                if constexpr (BLAY == LAY_K) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float4 v = *reinterpret_cast<const float4*>(&Bs[nl * LDK + half * (BKV / 2) + 8 * c + 4 * i]);
                        b[tn][4 * i] = v.x, b[tn][4 * i + 1] = v.y, b[tn][4 * i + 2] = v.z, b[tn][4 * i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[tn][j] = Bs[(half * (BKV / 2) + 8 * c + j) * BN + nl];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
        }
    };

    if (kbeg < kend) {
        load_tile(kbeg, ra0, rb0);
        load_tile(kbeg + BKV, ra1, rb1);
        for (int k0 = kbeg; k0 < kend; k0 += 2 * BKV) {
            stage(k0, ra0, rb0);
            __syncthreads();
            load_tile(k0 + 2 * BKV, ra0, rb0);
            compute();
            __syncthreads();
            if (k0 + BKV >= kend) break;
            stage(k0 + BKV, ra1, rb1);
            __syncthreads();
            load_tile(k0 + 3 * BKV, ra1, rb1);
            compute();
            __syncthreads();
        }
    }

    // ---- wide epilogue (NCHW outputs written directly): the MFMA leaves a lane with 4 consecutive ROWS of one column,
    // i.e. 4-byte stores of which a wave instruction covers two 128-byte segments - the store path, not HBM, bounds
    // the short-K 1x1 convolutions that way (64 -> 256 channels at 64x64: 67 MB of output in 84 us = 0.8 TB/s).  The
    // tile goes through LDS once and leaves as 16 bytes per lane along the pixel index: a wave instruction writes
    // four 256-byte row segments, a quarter of the store instructions, and the residual comes in as float4 too.
    if constexpr (WIDE_FITS) {
        const bool wide = p.dmode == DMODE_NCHW && p.splits == 1 && !p.defer_epilogue && (p.HoWo & 3) == 0 &&
                          (((uintptr_t)p.D | (uintptr_t)p.residual) & 15) == 0 && !p.no_wide_epilogue;
        const bool wide_up = up2x4_wide_ok(p);   // the four parity classes of a stride-2 transposed conv (gemm_common.h)
        if (wide || wide_up) {   // (uniform: every thread takes the same side)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        smem[((wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + (wn * TN + tn) * 32 + l31] =
                            acc[tm][tn][r];
            __syncthreads();
            if (wide_up) {
                up2x4_wide_store<BM, BN, LDC>(p, smem, m0, [&](int nl, int& img, int& ho, int& wo) {
                    const int n = n0 + nl;
                    if (n >= p.N) return false;
                    img = n / p.HoWo;
                    const int pp = n - img * p.HoWo;
                    ho = pp / p.Wout, wo = pp - ho * p.Wout;
                    return true;
                });
                return;
            }
            for (int idx = t; idx < BM * (BN / 4); idx += 256) {
                const int ml = idx / (BN / 4), c4 = idx - ml * (BN / 4);
                const int m = m0 + ml, n = n0 + c4 * 4;
                if (m >= p.M || n >= p.N) continue;   // N % 4 == 0 here (HoWo % 4 == 0): a float4 is in or out whole
                float4 v = *reinterpret_cast<const float4*>(&smem[ml * LDC + c4 * 4]);
                const int img = n / p.HoWo, pp = n - img * p.HoWo;
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
            return;
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            epilogue_tile(p, m0 + (wm * TM + tm) * 32, n0 + (wn * TN + tn) * 32 + l31, half, acc[tm][tn], bid.z);
}

template <int WM, int WN, int TM, int TN>
void launch_vec(const ivln_gemm_desc& d, hipStream_t s, int alay, int blay) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.splits);
#define IVLN_VCASE(AL, BL)                                                                          \
    if (alay == AL && blay == BL) {                                                                 \
        IVLN_LAUNCH_FAMILY((k_gemm_vec<WM, WN, TM, TN, AL, BL>), grid, dim3(256), 0, s, d);          \
        return;                                                                                     \
    }
    if (d.bmode == BMODE_CONV1X1 && d.stride == 2) {
        if constexpr (TM * TN == 1) {  // (eligibility keeps the stride-2 form to the one-accumulator tiles)
            IVLN_LAUNCH_FAMILY((k_gemm_vec<WM, WN, TM, TN, LAY_K, LAY_R, true>), grid, dim3(256), 0, s, d);
        }
        return;
    }
    IVLN_VCASE(LAY_K, LAY_R)
    IVLN_VCASE(LAY_K, LAY_K)
    IVLN_VCASE(LAY_R, LAY_R)
    IVLN_VCASE(LAY_R, LAY_K)
#undef IVLN_VCASE
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

int ivln_gemm_vec_tile_k() { return BKV; }

// Eligible: every float4 the kernel forms must be 16-byte aligned and must not straddle a row / image.
bool ivln_gemm_vec_eligible(const ivln_gemm_desc& d) {
    constexpr bool disabled = false;
    if (disabled || !al16(d.A) || !al16(d.B)) return false;
    if (d.amode == AMODE_MK) {
        if ((d.K & 3) || (d.lda & 3)) return false;
    } else if (d.amode == AMODE_KM) {
        if ((d.M & 3) || (d.lda & 3)) return false;
    } else {
        return false;
    }
    constexpr bool no_s2 = false;  // A/B switch
    if (d.bmode == BMODE_CONV1X1 && d.stride == 2) {
        // 16-byte loads at input column 2*wo (wo even): rows of 2*Wout columns, every row / plane / image 16-byte aligned
        if (no_s2 || d.amode != AMODE_MK || d.pad != 0 || d.Win != 2 * d.Wout || d.Hin < 2 * d.Hout - 1 || (d.Wout & 1) ||
            d.HoWo != d.Hout * d.Wout || (d.Win & 3) || (((int64_t)d.Hin * d.Win) & 3) || (d.in_img_stride & 3))
            return false;
    } else if (d.bmode == BMODE_CONV1X1) {
        if (d.stride != 1 || d.pad != 0 || d.Hin * d.Win != d.HoWo || (d.HoWo & 3) || (d.in_img_stride & 3)) return false;
    } else if (d.bmode == BMODE_KN) {
        if ((d.N & 3) || (d.ldb & 3)) return false;
    } else if (d.bmode == BMODE_NK) {
        if ((d.K & 3) || (d.ldb & 3)) return false;
    } else {
        return false;
    }
    return true;
}

// tile: 0 = 64x64, 1 = 32x128, 2 = 128x32, 3 = 128x128, 4 = 64x128 (numbering of ivln_gemm_f32)
int ivln_gemm_vec_launch(const ivln_gemm_desc& d, hipStream_t s, int tile) {
    const int alay = d.amode == AMODE_MK ? LAY_K : LAY_R;
    const int blay = d.bmode == BMODE_NK ? LAY_K : LAY_R;
    if (d.bmode == BMODE_CONV1X1 && d.stride == 2 && (tile == 3 || tile == 4)) tile = 0;  // built for the one-accumulator tiles
    switch (tile) {
        case 1: launch_vec<1, 4, 1, 1>(d, s, alay, blay); break;
        case 2: launch_vec<4, 1, 1, 1>(d, s, alay, blay); break;
        case 3: launch_vec<2, 2, 2, 2>(d, s, alay, blay); break;
        case 4: launch_vec<2, 2, 1, 2>(d, s, alay, blay); break;
        case 0: launch_vec<2, 2, 1, 1>(d, s, alay, blay); break;
        default: return IVLN_E_UNSUPPORTED;
    }
    return IVLN_OK;
}
