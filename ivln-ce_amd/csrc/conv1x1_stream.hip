// 1x1 convolution with SHORT K (64 / 128 / 256 input channels) over many pixels - RedNet's bottleneck expansions and
// lateral convs at 32x32 .. 128x128 (mapping_module/rednet.py:190-263: conv1 / conv3 of the Bottlenecks, the `agant`
// layers, the decoder's 1x1s) - as a weights-in-registers streaming kernel.
//
// Why not k_gemm_vec: with K = 64 a 64x64 tile is 0.5 MFLOP behind 32 KB of operand staging and 16 KB of output - the
// block is all prologue and epilogue (256 x 65536 x 64 ran at 44 TFLOP/s = 1.4 TB/s of output, r03 profile).  Here
//   * a wave keeps ITS slice of the weights in registers for its whole life (MT row tiles of 32 x K: 32 - 128 VGPRs),
//   * the activations stream global -> registers as 16-byte loads - lane (l31, half) reads channel 2 q + half, pixels
//     4 l31 .. 4 l31 + 3 - and the FOUR components feed four MFMAs whose pixel tiles are the four interleaved
//     quarter-strips {4 j + t}: no LDS, no barrier, fully coalesced 512-byte rows,
//   * so accumulator (row r, column l31) of quarter t is pixel 4 l31 + t and the epilogue leaves as 16-byte stores of
//     four consecutive pixels (scale / shift = folded BatchNorm or bias, residual, ReLU as in ivln_gemm_f32).
// A block = 4 waves = WAVES_M row groups x (4 / WAVES_M) strips of 128 pixels.  Bound: MFMA (4 MT MFMAs per 16-byte
// load and lane); algorithmic bytes per launch: (K + M) * N * 4.
#include <stdlib.h>

#include "gemm_common.h"
#include "family_timing.h"

namespace {

template <int K, int MT, int WAVES_M>
__global__ __launch_bounds__(256) void k_conv1x1_stream(const ivln_gemm_desc p) {
    constexpr int NQ = K / 2;             // MFMA steps (two channels per step)
    constexpr int PXW = 4 / WAVES_M;      // strips of 128 pixels per block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave % WAVES_M, wsx = wave / WAVES_M;
    const int64_t n0 = ((int64_t)blockIdx.x * PXW + wsx) * 128;
    if (n0 >= p.N) return;  // (no barriers in this kernel)
    const int img = (int)(n0 / p.HoWo), pp0 = (int)(n0 - (int64_t)img * p.HoWo);
    const int m_base = (blockIdx.y * WAVES_M + wm) * 32 * MT;
    if (m_base >= p.M) return;
    const int grp = p.grp_imgs > 0 ? img / p.grp_imgs : 0;
    const float* __restrict__ Ag = p.A + (int64_t)grp * p.a_grp_stride;
    // ---- this wave's weights: a[mt][q] = W[m_base + 32 mt + l31][2 q + half] (rows past M repeat the last row; masked at the store) ----
    float a[MT][NQ];
    if (p.A_packed) {  // the per-lane register image (ivln_conv_pack_weights_f32, KS = 1): one coalesced 256-byte row per step
        const float* __restrict__ ap = p.A_packed + (int64_t)grp * p.a_packed_grp_stride + (int64_t)(m_base >> 5) * NQ * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < NQ; ++q) a[mt][q] = ap[(mt * NQ + q) * 64];
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* row = Ag + (int64_t)min(m_base + 32 * mt + l31, p.M - 1) * p.lda;
#pragma unroll
            for (int j = 0; j < K / 4; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(row + 4 * j);
                a[mt][2 * j] = half ? w.y : w.x;
                a[mt][2 * j + 1] = half ? w.w : w.z;
            }
        }
    }
    // (pin the weights in registers: loads from read-only unaliased memory are otherwise free to be re-issued next to every use)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(a[mt][q]));
    // ---- stream the activations: channel 2 q + half, pixels pp0 + 4 l31 .. + 3 ----
    const float* __restrict__ bp = p.B + (int64_t)img * p.in_img_stride + (int64_t)half * p.HoWo + pp0 + 4 * l31;
    f32x16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;
    constexpr int DEPTH = K >= 128 ? 12 : 6;  // loads in flight per lane (4 MT MFMAs = 256 MT cycles of cover each)
    float4 b[NQ + DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) b[q] = *reinterpret_cast<const float4*>(bp + (int64_t)(2 * q) * p.HoWo);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + DEPTH < NQ) b[q + DEPTH] = *reinterpret_cast<const float4*>(bp + (int64_t)(2 * (q + DEPTH)) * p.HoWo);
        __builtin_amdgcn_sched_barrier(0);  // (the load stays AHEAD of this step's MFMAs: left alone, the scheduler sinks it to its
                                            //  use DEPTH steps later and the wave waits out a full memory round trip per step)
        const float bv[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][q], bv[t], acc[mt][t], 0, 0, 0);
    }
    // ---- epilogue: row (r & 3) + 8 (r >> 2) + 4 half of tile mt, pixels pp0 + 4 l31 + {0..3} = the four quarter tiles.
    // Eight rows at a time: their residual (and accumulate) loads are all issued before the first is used - one memory
    // round trip per batch instead of one per row (the first version spent more time here than in the MFMA loop) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float4 res[8], old[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i, m = min(m_base + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1);
                const int64_t addr = ((int64_t)img * p.Ctot + m) * p.HoWo + pp0 + 4 * l31;
                if (p.residual) res[i] = *reinterpret_cast<const float4*>(p.residual + addr);
                if (p.accumulate) old[i] = *reinterpret_cast<const float4*>(p.D + addr);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i, m = m_base + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) {
                    const int me = grp * p.M + m;  // (grp = 0 without image groups)
                    float4 v = make_float4(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]);
                    if (p.scale) {
                        const float sc = p.scale[me], sh = p.shift[me];
                        v.x = fmaf(v.x, sc, sh), v.y = fmaf(v.y, sc, sh), v.z = fmaf(v.z, sc, sh), v.w = fmaf(v.w, sc, sh);
                    } else if (p.shift) {
                        const float sh = p.shift[me];
                        v.x += sh, v.y += sh, v.z += sh, v.w += sh;
                    }
                    if (p.residual) v.x += res[i].x, v.y += res[i].y, v.z += res[i].z, v.w += res[i].w;
                    if (p.accumulate) v.x += old[i].x, v.y += old[i].y, v.z += old[i].z, v.w += old[i].w;
                    if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                    *reinterpret_cast<float4*>(p.D + ((int64_t)img * p.Ctot + m) * p.HoWo + pp0 + 4 * l31) = v;
                }
            }
        }
}

template <int K, int MT>
void launch_k(const ivln_gemm_desc& d, hipStream_t s) {
    const int rows = 32 * MT;  // per wave
    const int wavesm = d.M <= rows ? 1 : (d.M <= 2 * rows ? 2 : 4);
    const int64_t strips = d.N / 128;
    dim3 grid((unsigned)((strips + (4 / wavesm) - 1) / (4 / wavesm)), (unsigned)((d.M + rows * wavesm - 1) / (rows * wavesm)));
    if (wavesm == 1) IVLN_LAUNCH_FAMILY((k_conv1x1_stream<K, MT, 1>), grid, dim3(256), 0, s, d);
    else if (wavesm == 2) IVLN_LAUNCH_FAMILY((k_conv1x1_stream<K, MT, 2>), grid, dim3(256), 0, s, d);
    else IVLN_LAUNCH_FAMILY((k_conv1x1_stream<K, MT, 4>), grid, dim3(256), 0, s, d);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// IVLN_OK when launched, IVLN_E_UNSUPPORTED when the shape is not this kernel's (the caller goes on to k_gemm_vec).
int ivln_conv1x1_stream_launch(const ivln_gemm_desc& d, hipStream_t s, bool force) {
    constexpr bool disabled = false;  // A/B switch
    constexpr int min_n = 4096;
    if ((disabled && !force) || d.amode != AMODE_MK || d.bmode != BMODE_CONV1X1 || d.dmode != DMODE_NCHW) return IVLN_E_UNSUPPORTED;
    // Measured on RedNet's shapes at 8 frames (profiles/r04_rednet_B8_gemm_shapes.txt; us, this kernel vs k_gemm_vec):
    // 1024 x 4096 x 256: 31.9 vs 34.0, 512 x 16384 x 128: 38.0 vs 38.8, 64 x 65536 x 256: 29.7 vs 31.2 - and 256 x 65536 x 64:
    // 54.6 vs 48.8 (that shape moves 151 MB with its residual: HBM-bound either way, and the tiled kernel overlaps its
    // stores better).  K = 64 therefore stays with the tiled kernel unless IVLN_CONV1X1_STREAM_K64 is set.
    constexpr bool k64 = false;
    if (d.K != 128 && d.K != 256 && !(d.K == 64 && (k64 || force))) return IVLN_E_UNSUPPORTED;
    if (d.stride != 1 || d.pad != 0 || d.Hin * d.Win != d.HoWo || d.defer_epilogue || d.splits > 1 || d.stat_partials) return IVLN_E_UNSUPPORTED;
    if ((d.HoWo & 127) || d.N % d.HoWo != 0 || (d.N < min_n && !force) || (d.lda & 3) || (d.in_img_stride & 3)) return IVLN_E_UNSUPPORTED;
    if (!al16(d.A) || !al16(d.B) || !al16(d.D) || !al16(d.residual) || (d.grp_imgs > 0 && (d.a_grp_stride & 3))) return IVLN_E_UNSUPPORTED;
    // enough blocks to cover the chip at one or two per CU, else the tiled kernel's finer grid wins
    const int rows = d.K == 64 ? 64 : 32, wavesm = d.M <= rows ? 1 : (d.M <= 2 * rows ? 2 : 4);
    const int64_t blocks = ((d.N / 128 + (4 / wavesm) - 1) / (4 / wavesm)) * ((d.M + rows * wavesm - 1) / (rows * wavesm));
    if (blocks < 192 && !force) return IVLN_E_UNSUPPORTED;
    if (d.K == 64) launch_k<64, 2>(d, s);
    else if (d.K == 128) launch_k<128, 1>(d, s);
    else launch_k<256, 1>(d, s);
    return IVLN_OK;
}
