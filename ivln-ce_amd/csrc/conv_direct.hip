// Direct stride-1 3x3 / 7x7 / 2x2 convolution on the fp32 MFMA pipe (gfx950): the map CNN of the MapCMA policy
// (4 x 7x7, forward and input-gradient), the 3x3 convs of the DD-PPO ResNet and of RedNet, and RedNet's stride-2 3x3
// transposed convs as ONE 2x2-window conv over their four stacked output-parity classes (ops.convt_s2_stack).
//
// The implicit GEMM in gemm_conv.hip gathers every B element (one input pixel per (ci,kh,kw)) with its own
// address computation and bounds test; on 7x7 that VALU work competes with the MFMA issue slots.  Here a
// block owns 128 output pixels (IMGS images x PTH x PTW) x BM output channels and walks the input
// channels in chunks of CI:
//   * the (PTH+KS-1) x (PTW+KS-1) input patch of the chunk is staged ONCE in LDS (zero padded) and every
//     (kh,kw) tap re-reads it from there: KS*KS-fold reuse of each global load;
//   * the weight slice [BM][CI*KS*KS] is staged k-major next to it;
//   * the two k slots of v_mfma_f32_32x32x2_f32 (lane halves) take the two input channels of a pair, so
//     both operand fetches are `ds_read_b32 v, base offset:imm` with a per-lane base that never changes
//     and an immediate that encodes (channel pair, kh, kw): the MFMA loop carries no VALU work at all.
// K is therefore accumulated in (channel pair, kh, kw, channel parity) order instead of OIHW order; fp32
// sums differ from the implicit GEMM in the last bits (tests state the tolerance).
// Epilogue, split over channel chunks (blockIdx.z) and deferred raw slabs are the ones of ivln_gemm_f32.
#include <stdlib.h>

#include "gemm_common.h"

namespace {

// PACKED: the weights arrive pre-arranged by ivln_conv_pack_weights_f32 - per (64- or 32-channel block, chunk)
// the exact LDS image [k slot order][BM + 4] - so staging the weight slice is a linear copy of float4s
// (7 loads + 7 ds_write_b128 per thread instead of 25 scalar loads with their address arithmetic).
// (launch bounds: the packed 3x3 / 2x2 variants are held to 128 registers - 94 used, accumulators in VGPRs - so that a
//  SIMD keeps four or five waves instead of three: RedNet's 64-channel 128x128 convs launch 1024 blocks, exactly four
//  per CU, which at three resident blocks ran a second, quarter-full round)
// S = 2 (3x3 only: RedNet's down-sampling convs): the patch spans (2*PTH + 1) x (2*PTW + 1) input pixels and is staged
// with its even columns in front of its odd ones, so that the 32 lanes of an operand fetch - consecutive OUTPUT pixels,
// i.e. every second input column - still read consecutive LDS words.
template <int KS, int PTW, int PTH, int IMGS, int WM, bool PACKED, int S = 1>
__global__ __launch_bounds__(256, (PACKED && KS != 7 && S == 1) ? 4 : 1) void k_conv_direct(const ivln_gemm_desc p, int tiles_w, int tiles_h, int nimg,
                                                     int chunks_per_split) {
    static_assert(S == 1 || (S == 2 && KS != 2), "stride 2 is built for 3x3 and 7x7");
    constexpr int CI = conv_direct_ci(KS);  // input channels per chunk (even: channel pairs fill the k slots)
    constexpr int KK = KS * KS;
    constexpr int KC = CI * KK;   // k extent of a chunk
    constexpr int NQ = KC / 2;    // MFMA steps per chunk
    constexpr int WN = 4 / WM, TN = WM == 2 ? 2 : 1;
    constexpr int BM = 32 * WM, BN = 32 * WN * TN;
    static_assert(IMGS * PTH * PTW == BN, "pixel tile must hold 128 outputs");
    constexpr int PH = (PTH - 1) * S + KS, PWR = (PTW - 1) * S + KS;  // input rows / columns under the tile
    constexpr int PWH = (PWR + 1) / 2;                                // S = 2: even columns of a row (the odd ones follow)
    constexpr int PW = S == 2 ? 2 * PWH : PWR;                        // row stride of the staged patch
    constexpr int PLANE = PH * PW;
    constexpr int PATCH = IMGS * CI * PLANE;
    constexpr int LDA = PACKED ? BM + 4 : BM + 1;
    constexpr int NA = PACKED ? (KC * LDA / 4 + 255) / 256 * 4 : (BM * KC + 255) / 256, NP = (PATCH + 255) / 256;
    // KS == 2 serves the stacked parity classes of a stride-2 transposed conv (DMODE_NCHW_UP2X4): its output tile
    // goes through the same LDS block once more to leave as 16-byte stores (up2x4_wide_store, gemm_common.h)
    constexpr int LDC = BN + 4;
    constexpr int OPER = (KC * LDA + 3) / 4 * 4 + PATCH;
    // ... and so do NCHW output tiles whose rows hold whole pixel quads (nchw_wide_store): 16-byte stores along the row
    constexpr bool WIDE_NCHW = PTW % 4 == 0;
    constexpr int SMEM = ((KS == 2 || WIDE_NCHW) && BM * LDC > OPER) ? BM * LDC : OPER;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const As = smem;
    float* const Ps = smem + (KC * LDA + 3) / 4 * 4;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int bx = bid.x;
    const int tw = bx % tiles_w, th = (bx / tiles_w) % tiles_h, ig = bx / (tiles_w * tiles_h);
    const int img0 = ig * IMGS, ho0 = th * PTH, wo0 = tw * PTW;
    const int m0 = bid.y * BM;
    const int nch = (p.Cin + CI - 1) / CI;  // (a ragged last chunk - the 1- and 3-channel stems - reads its missing channels as zero)
    const int c_beg = bid.z * chunks_per_split;
    const int c_end = min(nch, c_beg + chunks_per_split);
    const int HW = p.Hin * p.Win;
    const int grp = p.grp_imgs > 0 ? img0 / p.grp_imgs : 0;  // weight set of this tile's images (image-grouped convs)
    const float* __restrict__ Ag = p.A + (int64_t)grp * p.a_grp_stride;

    // patch elements of this thread: chunk-invariant source offsets (-1 = zero padding / no such image)
    int poff[NP];
    uint32_t tail = 0;  // bit i: element i's channel exists in the LAST chunk too (all ones unless Cin % CI != 0)
    static_assert(NP <= 32, "one validity bit per patch element of a thread");
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = t + i * 256;
        const int il = idx / (CI * PLANE), rem = idx - il * (CI * PLANE);
        const int ci = rem / PLANE, rem2 = rem - ci * PLANE;
        tail |= (uint32_t)((nch - 1) * CI + ci < p.Cin) << i;
        const int y = rem2 / PW, xs = rem2 - y * PW;
        const int x = S == 2 ? (xs < PWH ? 2 * xs : 2 * (xs - PWH) + 1) : xs;  // staged column -> input column
        const int hi = ho0 * S - p.pad + y, wi = wo0 * S - p.pad + x, img = img0 + il;
        const bool ok = idx < PATCH && x < PWR && img < nimg && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        poff[i] = ok ? (int)((int64_t)img * p.in_img_stride + (int64_t)ci * HW + hi * p.Win + wi) : -1;
    }

    float ra[NA], rp[NP];
    auto load_chunk = [&](int c) {
        if constexpr (PACKED) {
            const float* wp = p.A_packed + (int64_t)grp * p.a_packed_grp_stride + ((int64_t)bid.y * nch + c) * (KC * LDA);
#pragma unroll
            for (int i = 0; i < NA / 4; ++i) {
                const int f = t + i * 256;
                const bool ok = f < KC * LDA / 4;
                const float4 v = *reinterpret_cast<const float4*>(wp + (ok ? f * 4 : 0));
                ra[4 * i] = v.x, ra[4 * i + 1] = v.y, ra[4 * i + 2] = v.z, ra[4 * i + 3] = v.w;
            }
        } else {
            const int kbase = c * KC;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = t + i * 256;
                const int co = idx / KC, kk = idx - co * KC;
                const bool ok = idx < BM * KC && m0 + co < p.M && kbase + kk < p.K;
                ra[i] = Ag[ok ? (int64_t)(m0 + co) * p.lda + kbase + kk : 0];
            }
        }
        const int cbase = c * CI * HW;
        // (no select here: the zero of a padding element is chosen in stage(), after the MFMA phase these loads fly
        //  under - a select next to the load makes hipcc wait for the load before that phase starts)
        const uint32_t live = c == nch - 1 ? tail : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NP; ++i) rp[i] = p.B[(poff[i] >= 0 && ((live >> i) & 1)) ? poff[i] + cbase : 0];
    };
    auto stage = [&](int c) {
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < NA / 4; ++i) {
                const int f = t + i * 256;
                if (f < KC * LDA / 4)
                    *reinterpret_cast<float4*>(&As[f * 4]) = make_float4(ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = t + i * 256;
                const int co = idx / KC, kk = idx - co * KC;
                const int ci = kk / KK, r = kk - ci * KK;
                if (idx < BM * KC)
                    As[((((ci >> 1) * KK + r) << 1) + (ci & 1)) * LDA + co] = (m0 + co < p.M && c * KC + kk < p.K) ? ra[i] : 0.f;
            }
        }
        const uint32_t live = c == nch - 1 ? tail : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = t + i * 256;
            if (idx < PATCH) Ps[idx] = (poff[i] >= 0 && ((live >> i) & 1)) ? rp[i] : 0.f;
        }
    };

    // per-lane operand bases: k slot (lane half) = channel parity inside the pair
    const int abase = half * LDA + wm * 32 + l31;
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nl = (wn * TN + tn) * 32 + l31;
        const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
        bbase[tn] = il * CI * PLANE + half * PLANE + ph * S * PW + pw;
    }

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    if (c_beg < c_end) {
        load_chunk(c_beg);
        for (int c = c_beg; c < c_end; ++c) {
            stage(c);
            __syncthreads();
            if (c + 1 < c_end) load_chunk(c + 1);  // in flight under the MFMA phase
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cp = q / KK, r = q - cp * KK, kh = r / KS, kw = r - kh * KS;
                const float a = As[abase + q * 2 * LDA];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float b = Ps[bbase[tn] + cp * 2 * PLANE + kh * PW + (S == 2 ? (kw & 1) * PWH + (kw >> 1) : kw)];
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tn], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: acc[r] -> channel (r&3) + 8*(r>>2) + 4*half, pixel l31 of the sub-tile ----
    if constexpr (WIDE_NCHW) {
        const bool wide = p.dmode == DMODE_NCHW && p.splits == 1 && !p.defer_epilogue && (p.Wout & 3) == 0 &&
                          (((uintptr_t)p.D | (uintptr_t)p.residual) & 15) == 0 && !p.no_wide_epilogue;
        if (wide) {   // (uniform; the K loop ended on a barrier, the operand tiles are dead)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + (wn * TN + tn) * 32 + l31] = acc[tn][r];
            __syncthreads();
            nchw_wide_store<BM, BN, LDC>(p, smem, m0, [&](int nl, int& img, int& pp) {
                const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
                img = img0 + il;
                const int ho = ho0 + ph, wo = wo0 + pw;
                pp = ho * p.Wout + wo;
                return img < nimg && ho < p.Hout && wo < p.Wout;   // (Wout % 4 == 0: a quad is in or out whole)
            }, p.stat_partials ? p.stat_partials + (int64_t)bx * p.M * 3 : nullptr);
            return;
        }
    }
    if constexpr (KS == 2) {
        if (up2x4_wide_ok(p)) {   // (uniform; the K loop ended on a barrier, the operand tiles are dead)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + (wn * TN + tn) * 32 + l31] = acc[tn][r];
            __syncthreads();
            up2x4_wide_store<BM, BN, LDC>(p, smem, m0, [&](int nl, int& img, int& ho, int& wo) {
                const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
                img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
                return img < nimg && ho < p.Hout && wo < p.Wout;
            });
            return;
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nl = (wn * TN + tn) * 32 + l31;
        const int il = nl / (PTH * PTW), ph = (nl / PTW) % PTH, pw = nl % PTW;
        const int img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
        const bool pix_ok = img < nimg && ho < p.Hout && wo < p.Wout;
        const int n = img * p.HoWo + ho * p.Wout + wo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (pix_ok && m < p.M) {
                if (p.splits > 1 || p.defer_epilogue) p.ws[((int64_t)bid.z * p.M + m) * p.N + n] = acc[tn][r];
                else epilogue_store(p, m, n, acc[tn][r]);
            }
        }
    }
}

// OIHW weights -> the direct kernel's LDS image, one [KC][BM + 4] tile per (channel block, chunk):
// row kp = ((ci>>1)*KS*KS + r)*2 + (ci&1) of chunk c holds W[m0 .. m0+BM-1][(c*CI + ci), r]; pad columns and
// channels past M are zero.
__global__ __launch_bounds__(256) void k_conv_pack_weights(const float* __restrict__ W, int M, int Cin, int KS, int BM,
                                                           float* __restrict__ out) {
    const int KK = KS * KS, CI = conv_direct_ci(KS), KC = CI * KK, LDA = BM + 4;
    const int nch = (Cin + CI - 1) / CI;
    const int64_t total = (int64_t)((M + BM - 1) / BM) * nch * KC * LDA;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % LDA);
    const int kp = (int)((idx / LDA) % KC);
    const int c = (int)((idx / ((int64_t)LDA * KC)) % nch);
    const int mb = (int)(idx / ((int64_t)LDA * KC * nch));
    const int q = kp >> 1, par = kp & 1;
    const int cp = q / KK, r = q - cp * KK;
    const int ci = c * CI + cp * 2 + par;
    const int m = mb * BM + col;
    out[idx] = (col < BM && m < M && ci < Cin) ? W[((int64_t)m * Cin + ci) * KK + r] : 0.f;
}

template <int KS, int PTW, int PTH, int IMGS, int S = 1>
void launch_wm(const ivln_gemm_desc& d, hipStream_t s, int nimg, int cps) {
    const int tiles_w = (d.Wout + PTW - 1) / PTW, tiles_h = (d.Hout + PTH - 1) / PTH;
    const int groups = (nimg + IMGS - 1) / IMGS;
    if (d.M <= 32) {
        dim3 grid(tiles_w * tiles_h * groups, (d.M + 31) / 32, d.splits);
        if (d.A_packed)
            IVLN_LAUNCH_FAMILY((k_conv_direct<KS, PTW, PTH, IMGS, 1, true, S>), grid, dim3(256), 0, s, d, tiles_w, tiles_h, nimg, cps);
        else
            IVLN_LAUNCH_FAMILY((k_conv_direct<KS, PTW, PTH, IMGS, 1, false, S>), grid, dim3(256), 0, s, d, tiles_w, tiles_h, nimg, cps);
    } else {
        dim3 grid(tiles_w * tiles_h * groups, (d.M + 63) / 64, d.splits);
        if (d.A_packed)
            IVLN_LAUNCH_FAMILY((k_conv_direct<KS, PTW, PTH, IMGS, 2, true, S>), grid, dim3(256), 0, s, d, tiles_w, tiles_h, nimg, cps);
        else
            IVLN_LAUNCH_FAMILY((k_conv_direct<KS, PTW, PTH, IMGS, 2, false, S>), grid, dim3(256), 0, s, d, tiles_w, tiles_h, nimg, cps);
    }
}

template <int KS, int S = 1>
void launch_ks(const ivln_gemm_desc& d, hipStream_t s, int nimg, int cps) {
    if (d.Wout > 16) launch_wm<KS, 32, 4, 1, S>(d, s, nimg, cps);
    else if (d.Wout > 8) launch_wm<KS, 16, 8, 1, S>(d, s, nimg, cps);
    else if (d.Wout > 4) launch_wm<KS, 8, 8, 2, S>(d, s, nimg, cps);
    else launch_wm<KS, 4, 4, 8, S>(d, s, nimg, cps);
}


// ------------------------------------------------------------------------------------------------
// Direct weight gradient of the same convolutions:
//   dW[co][ci][kh][kw] = sum over (img,ho,wo) dy[img][co][ho][wo] * x[img][ci][ho+kh-pad][wo+kw-pad]
// GEMM view: M = co, N = (ci,kh,kw), K = output pixels.  A block owns BM channels x 128 columns n and walks
// pixel tiles (IMGS x PTH x PTW = 128 pixels, the K extent of one iteration):
//   * dy tile staged pixel-major in LDS, x patch of the <= NCI input channels its 128 columns touch staged
//     once per pixel tile (each x element feeds up to KS*KS columns);
//   * the two k slots of the MFMA take two horizontally adjacent pixels, so again both operand fetches are
//     ds_read with a loop-invariant per-lane base ((ci,kh,kw) of the lane's column) + immediate (pixel).
// Pixel tiles are split over blockIdx.z (hundreds of splits: M x N is tiny, K is millions of pixels);
// slabs are reduced in fixed order by k_splitk_epilogue.
// ------------------------------------------------------------------------------------------------
// VEC (Wout % 4 == 0, dy 16-byte aligned): dy is fetched four horizontally adjacent pixels at a time and kept
// channel-major in LDS ([co][128 px + 1]: the +1 makes both the scalar stores - lanes = 4 channels x 8 pixel
// quads - and the operand reads - lanes = channels - conflict-free); a quarter of the loads and of the
// address arithmetic of the scalar path, which stores pixel-major.
template <int KS, int PTW, int PTH, int IMGS, int WM, bool VEC>
__global__ __launch_bounds__(256) void k_wgrad_direct(const ivln_gemm_desc p, int tiles_w, int tiles_h, int nimg,
                                                      int ntiles, int tiles_per_split) {
    constexpr int KK = KS * KS;
    constexpr int WN = 4 / WM, TN = WM == 2 ? 2 : 1;
    constexpr int BM = 32 * WM, BN = 32 * WN * TN;
    constexpr int NPX = IMGS * PTH * PTW;
    static_assert(NPX == 128 && BN == 128, "128 pixels per tile, 128 columns per block");
    constexpr int NCI = (BN + KK - 2) / KK + 1;  // input channels a 128-column tile can touch
    constexpr int PH = PTH + KS - 1, PW = PTW + KS - 1, PLANE = PH * PW;
    constexpr int PATCH = IMGS * NCI * PLANE;
    constexpr int LDA = BM + 1;    // pixel-major layout: Ds[px][co]
    constexpr int LDP = NPX + 1;   // channel-major layout (VEC): Ds[co][px]
    constexpr int NA = BM * NPX / 256, NP = (PATCH + 255) / 256;
    constexpr int NQ = NPX / 2;
    __shared__ float Ds[VEC ? BM * LDP : NPX * LDA];
    __shared__ float Ps[PATCH];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const BlockId bid = xcd_block_id(p.no_xcd_remap);
    const int n0 = bid.x * BN, m0 = bid.y * BM;
    const int c_lo = n0 / KK;
    const int HW = p.Hin * p.Win;
    const int t_beg = bid.z * tiles_per_split;
    const int t_end = min(ntiles, t_beg + tiles_per_split);

    float ra[NA], rp[NP];
    // validity of what the last load_tile fetched, one bit per load: the loads are unconditional from clamped addresses
    // and the zero of an out-of-range element is chosen in stage(), after the MFMA phase the loads fly under (a select
    // next to the load makes hipcc wait for the load before that phase starts)
    uint32_t oka = 0, okp = 0;
    static_assert((VEC ? NA / 4 : NA) <= 32 && NP <= 32, "validity masks are 32 bits");
    auto load_tile = [&](int tile) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, ig = tile / (tiles_w * tiles_h);
        const int img0 = ig * IMGS, ho0 = th * PTH, wo0 = tw * PTW;
        oka = 0, okp = 0;
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 4; ++i) {
                const int f = t + i * 256;  // (pixel-quad high, channel, pixel-quad low): 8 quads = 128 B per row
                const int co = (f >> 3) % BM, p4 = (f & 7) + 8 * (f / (8 * BM)), px = p4 * 4;
                const int il = px / (PTH * PTW), ph = (px / PTW) % PTH, pw = px % PTW;
                const int img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
                const bool ok = m0 + co < p.M && img < nimg && ho < p.Hout && wo < p.Wout;
                const float4 v = *reinterpret_cast<const float4*>(
                    p.A + (ok ? ((int64_t)img * p.M + m0 + co) * p.HoWo + ho * p.Wout + wo : 0));
                ra[4 * i] = v.x, ra[4 * i + 1] = v.y, ra[4 * i + 2] = v.z, ra[4 * i + 3] = v.w;
                oka |= (uint32_t)ok << i;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = t + i * 256;  // (co, pixel): pixel fastest -> coalesced rows of dy
                const int co = idx / NPX, px = idx - co * NPX;
                const int il = px / (PTH * PTW), ph = (px / PTW) % PTH, pw = px % PTW;
                const int img = img0 + il, ho = ho0 + ph, wo = wo0 + pw;
                const bool ok = m0 + co < p.M && img < nimg && ho < p.Hout && wo < p.Wout;
                ra[i] = p.A[ok ? ((int64_t)img * p.M + m0 + co) * p.HoWo + ho * p.Wout + wo : 0];
                oka |= (uint32_t)ok << i;
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = t + i * 256;
            const int il = idx / (NCI * PLANE), rem = idx - il * (NCI * PLANE);
            const int cl = rem / PLANE, rem2 = rem - cl * PLANE;
            const int y = rem2 / PW, x = rem2 - y * PW;
            const int img = img0 + il, ci = c_lo + cl, hi = ho0 - p.pad + y, wi = wo0 - p.pad + x;
            const bool ok = idx < PATCH && img < nimg && ci < p.Cin && (unsigned)hi < (unsigned)p.Hin &&
                            (unsigned)wi < (unsigned)p.Win;
            rp[i] = p.B[ok ? (int64_t)img * p.in_img_stride + (int64_t)ci * HW + hi * p.Win + wi : 0];
            okp |= (uint32_t)ok << i;
        }
    };
    auto stage = [&]() {
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 4; ++i) {
                const int f = t + i * 256;
                const int co = (f >> 3) % BM, px = ((f & 7) + 8 * (f / (8 * BM))) * 4;
                const bool ok = (oka >> i) & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) Ds[co * LDP + px + k] = ok ? ra[4 * i + k] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = t + i * 256;
                const int co = idx / NPX, px = idx - co * NPX;
                Ds[px * LDA + co] = (oka >> i) & 1 ? ra[i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = t + i * 256;
            if (idx < PATCH) Ps[idx] = (okp >> i) & 1 ? rp[i] : 0.f;
        }
    };

    // per-lane operand bases: k slot (lane half) = the odd pixel of a horizontal pair
    const int abase = VEC ? (wm * 32 + l31) * LDP + half : half * LDA + wm * 32 + l31;
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int n = n0 + (wn * TN + tn) * 32 + l31;
        if (n >= p.N) n = p.N - 1;  // masked at the store; keeps the LDS address in range
        const int ci = n / KK, r = n - ci * KK, kh = r / KS, kw = r - kh * KS;
        bbase[tn] = (ci - c_lo) * PLANE + kh * PW + kw + half;
    }

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    if (t_beg < t_end) {
        load_tile(t_beg);
        for (int tile = t_beg; tile < t_end; ++tile) {
            stage();
            __syncthreads();
            if (tile + 1 < t_end) load_tile(tile + 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int px = 2 * q;  // even pixel of the pair; the odd one is +1 in the same row
                const int il = px / (PTH * PTW), ph = (px / PTW) % PTH, pw = px % PTW;
                const float a = Ds[abase + (VEC ? px : px * LDA)];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float b = Ps[bbase[tn] + il * NCI * PLANE + ph * PW + pw];
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tn], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < p.M && n < p.N) {
                if (p.splits > 1) p.ws[((int64_t)bid.z * p.M + m) * p.N + n] = acc[tn][r];
                else epilogue_store(p, m, n, acc[tn][r]);
            }
        }
    }
}

template <int KS, int PTW, int PTH, int IMGS>
void launch_wgrad_wm(const ivln_gemm_desc& d, hipStream_t s, int nimg, int ntiles, int tps) {
    const int tiles_w = (d.Wout + PTW - 1) / PTW, tiles_h = (d.Hout + PTH - 1) / PTH;
    constexpr bool novec = false;  // A/B switch
    const bool vec = !novec && PTW >= 4 && (d.Wout & 3) == 0 && ((uintptr_t)d.A & 15) == 0 && (d.HoWo & 3) == 0;
    if (d.M <= 32) {
        dim3 grid((d.N + 127) / 128, (d.M + 31) / 32, d.splits);
        if (vec)
            IVLN_LAUNCH_FAMILY((k_wgrad_direct<KS, PTW, PTH, IMGS, 1, true>), grid, dim3(256), 0, s, d, tiles_w, tiles_h,
                               nimg, ntiles, tps);
        else
            IVLN_LAUNCH_FAMILY((k_wgrad_direct<KS, PTW, PTH, IMGS, 1, false>), grid, dim3(256), 0, s, d, tiles_w, tiles_h,
                               nimg, ntiles, tps);
    } else {
        dim3 grid((d.N + 127) / 128, (d.M + 63) / 64, d.splits);
        if (vec)
            IVLN_LAUNCH_FAMILY((k_wgrad_direct<KS, PTW, PTH, IMGS, 2, true>), grid, dim3(256), 0, s, d, tiles_w, tiles_h,
                               nimg, ntiles, tps);
        else
            IVLN_LAUNCH_FAMILY((k_wgrad_direct<KS, PTW, PTH, IMGS, 2, false>), grid, dim3(256), 0, s, d, tiles_w, tiles_h,
                               nimg, ntiles, tps);
    }
}

template <int KS>
void launch_wgrad_ks(const ivln_gemm_desc& d, hipStream_t s, int nimg, int ntiles, int tps) {
    if (d.Wout > 16) launch_wgrad_wm<KS, 32, 4, 1>(d, s, nimg, ntiles, tps);
    else if (d.Wout > 8) launch_wgrad_wm<KS, 16, 8, 1>(d, s, nimg, ntiles, tps);
    else if (d.Wout > 4) launch_wgrad_wm<KS, 8, 8, 2>(d, s, nimg, ntiles, tps);
    else launch_wgrad_wm<KS, 4, 4, 8>(d, s, nimg, ntiles, tps);
}

}  // namespace

int ivln_conv_direct_launch(ivln_gemm_desc& d, hipStream_t s) {
    constexpr bool disabled = false;
    const int KS = conv_ks(d.bmode);
    constexpr bool no_s2 = false;  // A/B switch
    if (disabled || KS == 0 || d.amode != AMODE_MK || d.dil != 1) return IVLN_E_UNSUPPORTED;
    if (d.stride != 1 && !(d.stride == 2 && KS != 2 && !no_s2)) return IVLN_E_UNSUPPORTED;
    const int CI = conv_direct_ci(KS);
    // (channel counts that do not fill the last chunk: only the stride-2 7x7 stems - 1 and 3 input channels)
    if ((d.Cin % CI != 0 && !(KS == 7 && d.stride == 2)) || d.K != d.Cin * KS * KS || d.HoWo != d.Hout * d.Wout ||
        d.N % d.HoWo != 0)
        return IVLN_E_UNSUPPORTED;
    const int nimg = d.N / d.HoWo;
    if ((int64_t)nimg * d.in_img_stride >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // 32-bit patch offsets
    const int PTW = d.Wout > 16 ? 32 : (d.Wout > 8 ? 16 : (d.Wout > 4 ? 8 : 4));
    const int PTH = PTW == 32 ? 4 : (PTW == 4 ? 4 : 8);
    const int IMGS = 128 / (PTW * PTH);
    if (d.grp_imgs > 0 && (d.grp_imgs % IMGS != 0 || nimg % d.grp_imgs != 0)) return IVLN_E_UNSUPPORTED;  // a tile's images share one weight set
    const int64_t tiles = (int64_t)((d.Wout + PTW - 1) / PTW) * ((d.Hout + PTH - 1) / PTH) * ((nimg + IMGS - 1) / IMGS);
    const int BM = d.M <= 32 ? 32 : 64;
    const int64_t blocks = tiles * ((d.M + BM - 1) / BM);
    const int nch = (d.Cin + CI - 1) / CI;
    // pixel-starved shapes (rollout batch, 4x4 / 8x8 tails): the implicit GEMM splits K far deeper than the
    // 16 channel-chunk splits available here and measured faster below these grid sizes
    if (d.tile_override == 0 && (KS == 7 ? blocks < 16 : blocks * (nch < 16 ? nch : 16) < 128)) return IVLN_E_UNSUPPORTED;
    int splits = 1;
    if (d.defer_epilogue && (!d.ws || d.ws_floats < (int64_t)d.M * d.N)) return IVLN_E_INVALID;
    if (d.splits == 0) {
        // split the channel chunks over blockIdx.z until the grid covers the chip
        constexpr int want_env = 0;  // tuning
        constexpr int below_env = 0;
        constexpr int defer_env = 0;  // tuning
        const int64_t want = (want_env > 0 && !d.defer_epilogue) ? want_env
                             : (d.defer_epilogue ? (defer_env > 0 ? defer_env : 256) : 512);
        if (d.ws && blocks < (below_env > 0 && !d.defer_epilogue ? below_env : 256) && nch >= 2) {
            splits = (int)((want + blocks - 1) / blocks);
            if (splits > nch) splits = nch;
            if (splits > 16) splits = 16;
            const int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
    } else {
        splits = d.splits > nch ? nch : d.splits;
        if (splits > 1 && (!d.ws || d.ws_floats < (int64_t)splits * d.M * d.N)) return IVLN_E_INVALID;
    }
    const int cps = (nch + splits - 1) / splits;
    splits = (nch + cps - 1) / cps;
    d.splits = splits;
    // per-tile statistics ride in the wide NCHW epilogue only (same test as in the kernel)
    const bool wide = d.dmode == DMODE_NCHW && splits == 1 && !d.defer_epilogue && (d.Wout & 3) == 0 &&
                      (((uintptr_t)d.D | (uintptr_t)d.residual) & 15) == 0 && !d.no_wide_epilogue;
    if (!wide) d.stat_partials = nullptr;
    if (d.stat_tiles) *d.stat_tiles = d.stat_partials ? (int)tiles : 0;
    if (KS == 7 && d.stride == 2) launch_ks<7, 2>(d, s, nimg, cps);
    else if (KS == 7) launch_ks<7>(d, s, nimg, cps);
    else if (KS == 2) launch_ks<2>(d, s, nimg, cps);
    else if (d.stride == 2) launch_ks<3, 2>(d, s, nimg, cps);
    else launch_ks<3>(d, s, nimg, cps);
    return IVLN_OK;
}

int ivln_wgrad_direct_launch(ivln_gemm_desc& d, hipStream_t s) {
    constexpr bool disabled = false;
    if (disabled || d.amode != AMODE_NCHW_P || d.bmode != BMODE_IM2COL_T || d.stride != 1 || d.dil != 1 ||
        d.Cin <= 0 || d.N % d.Cin != 0 || d.defer_epilogue)
        return IVLN_E_UNSUPPORTED;
    const int KK = d.N / d.Cin;
    const int KS = KK == 49 ? 7 : (KK == 9 ? 3 : 0);
    if (KS == 0 || d.HoWo != d.Hout * d.Wout || d.K % d.HoWo != 0) return IVLN_E_UNSUPPORTED;
    // the kernel reads x[ho + kh - pad]: only the same-geometry case the forward conv produced
    if (d.Hout != d.Hin + 2 * d.pad - KS + 1 || d.Wout != d.Win + 2 * d.pad - KS + 1) return IVLN_E_UNSUPPORTED;
    const int nimg = d.K / d.HoWo;
    const int PTW = d.Wout > 16 ? 32 : (d.Wout > 8 ? 16 : (d.Wout > 4 ? 8 : 4));
    const int PTH = PTW == 32 ? 4 : (PTW == 4 ? 4 : 8);
    const int IMGS = 128 / (PTW * PTH);
    const int ntiles = ((d.Wout + PTW - 1) / PTW) * ((d.Hout + PTH - 1) / PTH) * ((nimg + IMGS - 1) / IMGS);
    const int BM = d.M <= 32 ? 32 : 64;
    const int64_t blocks = (int64_t)((d.N + 127) / 128) * ((d.M + BM - 1) / BM);
    int splits = 1;
    if (d.splits == 0) {
        if (d.ws && ntiles >= 2) {
            constexpr int want_env = 0;  // tuning
            // 8-16 blocks per CU: shorter blocks even out the tail of the one long wave of work (measured on the map
            // CNN's four layers at 512 images, 1024 / 2048 / 4096 wanted blocks: layer 1 - six column tiles - 1152 /
            // 1086 / 1009 us, layer 2 1165 / 1122 / 1135 us, layers 3-4 unchanged)
            const int64_t want = want_env > 0 ? want_env : (blocks <= 8 ? 4096 : 2048);
            splits = (int)((want + blocks - 1) / blocks);
            if (splits > ntiles) splits = ntiles;
            if (splits > 512) splits = 512;
            const int64_t cap = d.ws_floats / ((int64_t)d.M * d.N);
            if (splits > cap) splits = (int)cap;
            if (splits < 1) splits = 1;
        }
    } else {
        splits = d.splits > ntiles ? ntiles : d.splits;
        if (splits > 1 && (!d.ws || d.ws_floats < (int64_t)splits * d.M * d.N)) return IVLN_E_INVALID;
    }
    const int tps = (ntiles + splits - 1) / splits;
    splits = (ntiles + tps - 1) / tps;
    d.splits = splits;
    if (KS == 7) launch_wgrad_ks<7>(d, s, nimg, ntiles, tps);
    else launch_wgrad_ks<3>(d, s, nimg, ntiles, tps);
    return IVLN_OK;
}

// 1x1 convs with K = 64 / 128 / 256 (conv1x1_stream.hip): the per-lane register image of the weights,
// out[((mt * K/2 + q) * 64) + lane] = W[min(32 mt + (lane & 31), M - 1)][2 q + (lane >> 5)] - one coalesced 256-byte row per MFMA step
__global__ __launch_bounds__(256) void k_conv1x1_pack_weights(const float* __restrict__ W, int M, int K, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int NQ = K / 2;
    const int64_t total = (int64_t)((M + 31) / 32) * NQ * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), q = (int)((idx >> 6) % NQ), mt = (int)((idx >> 6) / NQ);
    const int m = min(32 * mt + (lane & 31), M - 1);
    out[idx] = W[(int64_t)m * K + 2 * q + (lane >> 5)];
}

extern "C" int64_t ivln_conv_packed_floats(int M, int Cin, int KS) {
    if (KS == 1) return (M > 0 && (Cin == 64 || Cin == 128 || Cin == 256)) ? (int64_t)((M + 31) / 32) * 32 * Cin : 0;
    if ((KS != 2 && KS != 3 && KS != 7) || M <= 0) return 0;
    const int CI = conv_direct_ci(KS), BM = M <= 32 ? 32 : 64;
    if (Cin % CI && KS != 7) return 0;  // (ragged channel chunks: the 7x7 stems only)
    return (int64_t)((M + BM - 1) / BM) * ((Cin + CI - 1) / CI) * (CI * KS * KS) * (BM + 4);
}

extern "C" int ivln_conv_pack_weights_f32(const float* W, int M, int Cin, int KS, float* out, void* stream) {
    const int64_t total = ivln_conv_packed_floats(M, Cin, KS);
    if (!W || !out || total <= 0) return IVLN_E_INVALID;
    if (KS == 1) {
        hipLaunchKernelGGL(k_conv1x1_pack_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, M, Cin, out);
        return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
    }
    hipLaunchKernelGGL(k_conv_pack_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, M,
                       Cin, KS, M <= 32 ? 32 : 64, out);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}
