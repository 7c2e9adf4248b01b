// Backward / loss / optimizer kernels of the DAgger update (base_il_trainer.py:173-219) for gfx950.
// GEMM-shaped gradients (dX, dW of every conv / linear) run through gemm_conv.hip; this file holds the
// element / reduction / recurrence kernels around them: ReLU mask, deterministic column and channel
// sums (bias grads), attention backward, GRU BPTT step, bidirectional LSTM BPTT, BatchNorm(train)
// +ReLU+AvgPool backward, conv weight flip for dgrad, embedding scatter, inflection-weighted
// cross-entropy (+ gradient), progress-monitor loss, fused flat-bucket Adam.
#include <hip/hip_runtime.h>
#include <stdlib.h>
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
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                  float* __restrict__ dx, int rows, int cols, int64_t ld_dy,
                                                  int64_t ld_y, int64_t ld_dx) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * cols) return;
    int r = (int)(idx / cols), c = (int)(idx % cols);
    dx[(int64_t)r * ld_dx + c] = y[(int64_t)r * ld_y + c] > 0.f ? dy[(int64_t)r * ld_dy + c] : 0.f;
}

// out = a + b (strided rows)
__global__ __launch_bounds__(256) void k_add2d(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                               int64_t ldb, float* __restrict__ y, int64_t ldy, int rows,
                                               int cols) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * cols) return;
    int r = (int)(idx / cols), c = (int)(idx % cols);
    y[(int64_t)r * ldy + c] = a[(int64_t)r * lda + c] + b[(int64_t)r * ldb + c];
}

// Deterministic column sums of a (rows, cols) matrix: stage 1 partial[split][c], stage 2 fixed-order sum.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int64_t ld, int rows, int cols,
                                                        int rows_per_split, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_split;
    const int r1 = min(rows, r0 + rows_per_split);
    float s = 0.f;
    if (c < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four loads in flight per thread
        int r = r0 + sub;
        for (; r + 12 < r1; r += 16) {
            a0 += x[(int64_t)r * ld + c];
            a1 += x[(int64_t)(r + 4) * ld + c];
            a2 += x[(int64_t)(r + 8) * ld + c];
            a3 += x[(int64_t)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) a0 += x[(int64_t)r * ld + c];
        s = (a0 + a1) + (a2 + a3);
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < cols)
        partial[(int64_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void k_colsum_final(const float* __restrict__ partial, int splits, int cols, float* __restrict__ out,
                               int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * cols + c];
    out[c] = accumulate ? out[c] + s : s;
}

// Up to 32 column sums in TWO launches (ivln_colsum_multi_f32): the bias gradients of the update's Linear layers and
// GRU / LSTM gates used to be two launches each (17 x 2 per update, 0.19 ms of dependent launches).  Stage 1 block ->
// job by its first-block table; the arithmetic (four row sub-lanes, fixed-order partials and final) is k_colsum_*'s.
struct ColsumJobs {
    const float* x[32];
    float* out[32];
    int64_t ld[32];
    int rows[32], cols[32], splits[32], rps[32];
    int first_block[33];   // stage 1: blocks [first_block[j], first_block[j+1]) = job j, (col tile, split) row-major
    int64_t ws_off[32];    // partials of job j: ws[ws_off[j] + split*cols + c]
    int first_col[33];     // stage 2: columns [first_col[j], first_col[j+1]) of the concatenated column axis
    int n;
};
__global__ __launch_bounds__(256) void k_colsum_multi_partial(const ColsumJobs J, float* __restrict__ ws) {
    __shared__ float red[4][64];
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const int local = blockIdx.x - J.first_block[j];
    const int tiles = (J.cols[j] + 63) / 64;
    const int tile = local % tiles, split = local / tiles;
    const float* __restrict__ x = J.x[j];
    const int64_t ld = J.ld[j];
    const int cols = J.cols[j], rows = J.rows[j];
    const int c = tile * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int r0 = split * J.rps[j], r1 = min(rows, r0 + J.rps[j]);
    float s = 0.f;
    if (c < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = r0 + sub;
        for (; r + 12 < r1; r += 16) {
            a0 += x[(int64_t)r * ld + c];
            a1 += x[(int64_t)(r + 4) * ld + c];
            a2 += x[(int64_t)(r + 8) * ld + c];
            a3 += x[(int64_t)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) a0 += x[(int64_t)r * ld + c];
        s = (a0 + a1) + (a2 + a3);
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < cols)
        ws[J.ws_off[j] + (int64_t)split * cols + c] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_colsum_multi_final(const ColsumJobs J, const float* __restrict__ ws) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= J.first_col[J.n]) return;
    int j = 0;
    while (j + 1 < J.n && g >= J.first_col[j + 1]) ++j;
    const int c = g - J.first_col[j], cols = J.cols[j];
    float s = 0.f;
    for (int z = 0; z < J.splits[j]; ++z) s += ws[J.ws_off[j] + (int64_t)z * cols + c];
    J.out[j][c] = s;
}

// per-channel sum over (N, HW) of an NCHW tensor (conv bias grad): grid (C, S) partials over image
// slices, then a fixed-order final sum (deterministic)
__global__ __launch_bounds__(256) void k_nchw_chansum_partial(const float* __restrict__ x, int N, int C, int HW,
                                                              int imgs_per_split, float* __restrict__ partial) {
    __shared__ float red[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int i0 = sp * imgs_per_split, i1 = min(N, i0 + imgs_per_split);
    float s = 0.f;
    if ((HW & 3) == 0 && ((uintptr_t)x & 15) == 0) {  // 16-byte loads, four independent partial sums
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
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(int64_t)c * S + sp] = s;
}
// one 64-lane wave per channel: lane l adds partials l, l + 64, ... in order, then a fixed shuffle tree - the result does not
// depend on the launch (a single thread walking all S partials took 11-30 us once the passes ran 256 splits per channel)
__global__ __launch_bounds__(64) void k_chan_final(const float* __restrict__ partial, int S, int C, int K,
                                                   float* __restrict__ out0, float* __restrict__ out1) {
    const int c = blockIdx.x, l = threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b2 = 0.f;
    for (int sp = l; sp < S; sp += 64) {
        a += partial[((int64_t)c * S + sp) * K];
        if (K > 1) b2 += partial[((int64_t)c * S + sp) * K + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b2 += __shfl_xor(b2, o);
    }
    if (l == 0) {
        out0[c] = a;
        if (K > 1) out1[c] = b2;
    }
}

// (R, C) -> (C, R)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
    __shared__ float tile[16][17];
    int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (by + ty < R && bx + tx < C) tile[ty][tx] = x[(int64_t)(by + ty) * C + bx + tx];
    __syncthreads();
    if (bx + ty < C && by + tx < R) y[(int64_t)(bx + ty) * R + by + tx] = tile[tx][ty];
}

// W (O,I,k,k) -> W' (I,O,k,k) with both spatial axes flipped: dgrad of a stride-1 conv is
// conv(dy, W', pad = k-1-pad)
__global__ __launch_bounds__(256) void k_weight_flip_transpose(const float* __restrict__ w, float* __restrict__ wt,
                                                               int O, int I, int KH, int KW) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)O * I * KH * KW;
    if (idx >= total) return;
    int kw = (int)(idx % KW);
    int kh = (int)((idx / KW) % KH);
    int o = (int)((idx / ((int64_t)KW * KH)) % O);
    int i = (int)(idx / ((int64_t)KW * KH * O));
    wt[idx] = w[(((int64_t)o * I + i) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
}

// ------------------------------------------------------------------------------------------
// Attention backward (forward: k_attn in nn_ops.hip).  One block per row.
//   dv[c][i] = a[i]*dout[c];  da[i] = sum_c dout[c] v[c][i];  dl[i] = a[i]*(da[i] - sum_j a[j]da[j])*scale
//   dq[c] = sum_i dl[i] k[c][i];  dk[c][i] = dl[i] q[c]
// dk / dv are written (not accumulated) with image strides; masked positions have a == 0.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attn_bwd(const float* __restrict__ dout, int64_t ld_dout,
                                                  const float* __restrict__ attn, const float* __restrict__ q,
                                                  int64_t ldq, const float* __restrict__ k, int64_t k_img_stride,
                                                  const float* __restrict__ v, int64_t v_img_stride, float scale,
                                                  int Ck, int Cv, int I, float* __restrict__ dq, int64_t ld_dq,
                                                  float* __restrict__ dk, int64_t dk_img_stride,
                                                  float* __restrict__ dv, int64_t dv_img_stride,
                                                  const int* __restrict__ row_index) {
    __shared__ float ds[1024];   // dout row, later q row
    __shared__ float as[512];    // attn
    __shared__ float dl[512];    // dlogits
    __shared__ float red[16];
    const int n = blockIdx.x;
    const int img = row_index ? row_index[n] : n;  // rows sharing one key/value image (forward: k_attn_logits)
    for (int c = threadIdx.x; c < Cv; c += 256) ds[c] = dout[(int64_t)n * ld_dout + c];
    for (int i = threadIdx.x; i < I; i += 256) as[i] = attn[(int64_t)n * I + i];
    __syncthreads();
    const float* vp = v + (int64_t)img * v_img_stride;
    float* dvp = dv + (int64_t)n * dv_img_stride;
    float dot = 0.f;
    for (int i = threadIdx.x; i < I; i += 256) {
        float da = 0.f;
        for (int c = 0; c < Cv; ++c) da = fmaf(ds[c], vp[(int64_t)c * I + i], da);
        dl[i] = da;
        dot += as[i] * da;
    }
    // dv (coalesced over i)
    for (int64_t e = threadIdx.x; e < (int64_t)Cv * I; e += 256) {
        int c = (int)(e / I), i = (int)(e % I);
        dvp[e] = as[i] * ds[c];
    }
    dot = block_sum(dot, red);
    for (int i = threadIdx.x; i < I; i += 256) dl[i] = as[i] * (dl[i] - dot) * scale;
    __syncthreads();
    for (int c = threadIdx.x; c < Ck; c += 256) ds[c] = q[(int64_t)n * ldq + c];
    __syncthreads();
    const float* kp = k + (int64_t)img * k_img_stride;
    float* dkp = dk + (int64_t)n * dk_img_stride;
    for (int c = threadIdx.x; c < Ck; c += 256) {
        float acc = 0.f;
        for (int i = 0; i < I; ++i) acc = fmaf(dl[i], kp[(int64_t)c * I + i], acc);
        dq[(int64_t)n * ld_dq + c] = acc;
    }
    for (int64_t e = threadIdx.x; e < (int64_t)Ck * I; e += 256) {
        int c = (int)(e / I), i = (int)(e % I);
        dkp[e] = dl[i] * ds[c];
    }
}

// ------------------------------------------------------------------------------------------
// GRU BPTT, element part of one step (rows = N sequences of step t):
//   dh = dout + dh_carry;  dn = dh(1-z); dz = dh(hp - n); dhz = dh z
//   dn_pre = dn(1-n^2); dz_pre = dz z(1-z); dr_pre = dn_pre ghn r(1-r)
//   dgi = [dr_pre, dz_pre, dn_pre];  dgh = [dr_pre, dz_pre, dn_pre r];  hp = h_prev*mask
// The matvec dh_prev = dgh . W_hh then runs through k_linear_skinny_ex with (+dhz)*mask epilogue.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gru_bwd_elem(const float* __restrict__ dout, int64_t ld_dout,
                                                      const float* __restrict__ dh_carry,
                                                      const float* __restrict__ r, const float* __restrict__ z,
                                                      const float* __restrict__ n, const float* __restrict__ ghn,
                                                      const float* __restrict__ h_prev, int64_t ldh,
                                                      const uint8_t* __restrict__ mask, int rows, int H,
                                                      float* __restrict__ dgi, float* __restrict__ dgh,
                                                      float* __restrict__ dhz, float* __restrict__ hp_out) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    int row = idx / H, j = idx % H;
    float dh = dout[(int64_t)row * ld_dout + j] + (dh_carry ? dh_carry[idx] : 0.f);
    float mk = mask[row] ? 1.f : 0.f;
    float hp = h_prev[(int64_t)row * ldh + j] * mk;
    float rg = r[idx], zg = z[idx], ng = n[idx], gh = ghn[idx];
    float dn = dh * (1.f - zg);
    float dz = dh * (hp - ng);
    float dn_pre = dn * (1.f - ng * ng);
    float dz_pre = dz * zg * (1.f - zg);
    float dr_pre = dn_pre * gh * rg * (1.f - rg);
    int64_t o = (int64_t)row * 3 * H + j;
    dgi[o] = dr_pre;
    dgi[o + H] = dz_pre;
    dgi[o + 2 * H] = dn_pre;
    dgh[o] = dr_pre;
    dgh[o + H] = dz_pre;
    dgh[o + 2 * H] = dn_pre * rg;
    dhz[idx] = dh * zg;
    hp_out[idx] = hp;
}

// y[r][o] = (W[o].x[r] + add[r][o]) * (rowmask[r] ? 1 : 0)   (skinny rows; see k_linear_skinny)
// One block per output o, 32 lanes per row (8 rows per pass): each lane owns every 32nd float4 of K and a
// (row, o) costs one 5-step shuffle reduction - no LDS, no barrier.
__global__ __launch_bounds__(256) void k_linear_skinny_ex(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ add, int64_t ld_add,
                                                          const uint8_t* __restrict__ rowmask,
                                                          float* __restrict__ y, int64_t ldy, int rows, int K,
                                                          int O) {
    const int o = blockIdx.x;
    const int l = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const float* wr = W + (int64_t)o * K;
    for (int r0 = 0; r0 < rows; r0 += 8) {
        const int row = r0 + rr;
        const bool row_ok = row < rows;
        const float* xr = x + (int64_t)(row_ok ? row : 0) * ldx;
        float a0 = 0.f, a1 = 0.f;
        for (int k = l * 4; k < K; k += 128) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + k);
            const float4 xv = *reinterpret_cast<const float4*>(xr + k);
            a0 = fmaf(wv.x, xv.x, a0);
            a1 = fmaf(wv.y, xv.y, a1);
            a0 = fmaf(wv.z, xv.z, a0);
            a1 = fmaf(wv.w, xv.w, a1);
        }
        float v = a0 + a1;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (l == 0 && row_ok) {
            if (add) v += add[(int64_t)row * ld_add + o];
            if (rowmask) v = rowmask[row] ? v : 0.f;
            y[(int64_t)row * ldy + o] = v;
        }
    }
}

// One BPTT step of the masked GRU in ONE launch: block j first finishes step t for hidden unit j,
//   dh_prev[row][j] = (W_hh^T[j] . dgh_t[row] + dhz[row][j]) * mask_t[row]        (k_linear_skinny_ex)
// and, since the element part of step t-1 for unit j needs nothing but that value, runs it right away
// (k_gru_bwd_elem on column j).  Halves the launches of the BPTT chain (2 x 126 per GRU per update).
__global__ __launch_bounds__(256) void k_gru_bwd_step(
    const float* __restrict__ dgh_t, int64_t ld_dgh, const float* __restrict__ Wt, const uint8_t* __restrict__ mask_t,
    const float* __restrict__ dout_p, int64_t ld_dout, const float* __restrict__ r, const float* __restrict__ z,
    const float* __restrict__ n, const float* __restrict__ ghn, const float* __restrict__ h_pp, int64_t ldh,
    const uint8_t* __restrict__ mask_p, int rows, int H, float* __restrict__ dhz, float* __restrict__ dgi_p,
    float* __restrict__ dgh_p, float* __restrict__ hp_p) {
    const int j = blockIdx.x;
    const int l = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const int K = 3 * H;
    const float* wr = Wt + (int64_t)j * K;
    for (int r0 = 0; r0 < rows; r0 += 8) {
        const int row = r0 + rr;
        const bool row_ok = row < rows;
        const int rowc = row_ok ? row : 0;
        const float* xr = dgh_t + (int64_t)rowc * ld_dgh;
        // the element part's inputs do not depend on the matvec: fetch them first, under its loads
        const int idx = rowc * H + j;
        const float e_dout = dout_p[(int64_t)rowc * ld_dout + j], e_dhz = dhz[idx];
        const float e_h = h_pp[(int64_t)rowc * ldh + j];
        const float rg = r[idx], zg = z[idx], ng = n[idx], gh = ghn[idx];
        const bool e_mt = mask_t[rowc] != 0, e_mp = mask_p[rowc] != 0;
        float a0 = 0.f, a1 = 0.f;
        for (int k = l * 4; k < K; k += 128) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + k);
            const float4 xv = *reinterpret_cast<const float4*>(xr + k);
            a0 = fmaf(wv.x, xv.x, a0);
            a1 = fmaf(wv.y, xv.y, a1);
            a0 = fmaf(wv.z, xv.z, a0);
            a1 = fmaf(wv.w, xv.w, a1);
        }
        float v = a0 + a1;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (l == 0 && row_ok) {
            v = e_mt ? v + e_dhz : 0.f;  // dh carried into step t-1
            // ---- element part of step t-1 (k_gru_bwd_elem) ----
            const float dh = e_dout + v;
            const float hp = e_mp ? e_h : 0.f;
            const float dn = dh * (1.f - zg);
            const float dz = dh * (hp - ng);
            const float dn_pre = dn * (1.f - ng * ng);
            const float dz_pre = dz * zg * (1.f - zg);
            const float dr_pre = dn_pre * gh * rg * (1.f - rg);
            const int64_t o = (int64_t)row * 3 * H + j;
            dgi_p[o] = dr_pre;
            dgi_p[o + H] = dz_pre;
            dgi_p[o + 2 * H] = dn_pre;
            dgh_p[o] = dr_pre;
            dgh_p[o + H] = dz_pre;
            dgh_p[o + 2 * H] = dn_pre * rg;
            dhz[idx] = dh * zg;
            hp_p[idx] = hp;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Bidirectional LSTM BPTT (forward: k_lstm_bidir).  grid (B, 2); 4H = 512 threads.  Thread
// (k = tid % H, part = tid / H) keeps W_hh[part*H .. part*H+H-1][k] (a column slice) in registers so
// dh_prev[k] = sum_g W_hh[g][k] dgate[g] is 4 partial dots of H terms + an LDS reduction.
// dout: (B, 2H, L) gradient of the channel-major outputs.  Writes dgx (B*L, 4H) per direction
// (pre-activation gate grads, zero for t >= len) and hprev (B*L, H) per direction (h_{t-1} in
// processing order) for the dW_hh / dW_ih GEMMs.
// ------------------------------------------------------------------------------------------
// LDS-only barrier (see nn_ops.hip): __syncthreads() would wait for the per-step global stores.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_bidir_bwd(const float* __restrict__ dout,
                                                          const float* __restrict__ out,
                                                          const float* __restrict__ gates,
                                                          const float* __restrict__ cs,
                                                          const float* __restrict__ whh_f,
                                                          const float* __restrict__ whh_r,
                                                          const int* __restrict__ lengths, int L,
                                                          float* __restrict__ dgx_f, float* __restrict__ dgx_r,
                                                          float* __restrict__ hprev_f,
                                                          float* __restrict__ hprev_r) {
    constexpr int G = 4 * H;
    // Matvec dh_prev[k] = sum_g W_hh[g][k] * dgate[g] (G = 4H terms): thread (ko = tid/16, ig = tid%16) owns the
    // 4 outputs k = 4ko..4ko+3 and the 32 gate rows g = 32ig..32ig+31 (H weights in registers), reads ONLY its
    // 32 gate gradients from LDS (8 ds_read_b128; every thread reading all 4H saturated the LDS port), and the
    // 16 lanes that share an output add their partial sums by shuffles.  The element part's inputs (saved gates,
    // cell states, dout) are fetched one timestep ahead.  Two barriers per timestep.
    constexpr int GS = 36;  // LDS stride of a 32-gate group: +4 words -> the 16 groups spread over the banks
    __shared__ __attribute__((aligned(16))) float dg[16 * GS];
    __shared__ float dhc[H];
    static_assert(H == 128, "mapping below assumes 4H = 512 threads");
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int ig = tid & 15, ko = tid >> 4;
    const float* whh = dir == 0 ? whh_f : whh_r;
    float* dgx = (dir == 0 ? dgx_f : dgx_r) + (int64_t)b * L * G;
    float* hprev = (dir == 0 ? hprev_f : hprev_r) + (int64_t)b * L * H;
    float w[4][32];
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(whh + (int64_t)(ig * 32 + g) * H + ko * 4);
        w[0][g] = v.x, w[1][g] = v.y, w[2][g] = v.z, w[3][g] = v.w;
    }
    if (tid < H) dhc[tid] = 0.f;
    int len = lengths[b];
    if (len > L) len = L;
    {
        const int k = tid % H, pq = tid / H;
        for (int t = len + pq; t < L; t += 4) {  // padded positions carry no gradient
#pragma unroll 4
            for (int g = k; g < G; g += H) dgx[(int64_t)t * G + g] = 0.f;
            hprev[(int64_t)t * H + k] = 0.f;
        }
    }
    lds_barrier();
    const float* gt = gates + ((int64_t)b * 2 + dir) * L * G;
    const float* ct = cs + ((int64_t)b * 2 + dir) * L * H;
    const int j = tid;  // element part: threads 0..H-1 own hidden unit j
    const int64_t orow = ((int64_t)b * 2 * H + dir * H + (j < H ? j : 0)) * L;
    float dcc = 0.f;    // dc carried to the previous timestep (register: only thread j touches it)
    // inputs of a timestep: gates i,f,g,o, c, c_prev, h_prev, dout
    float n_i = 0.f, n_f = 0.f, n_g = 0.f, n_o = 0.f, n_c = 0.f, n_cp = 0.f, n_hp = 0.f, n_do = 0.f;
    auto fetch = [&](int s) {
        if (j < H && s >= 0) {
            const int t = dir == 0 ? s : len - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
            n_i = gt[(int64_t)t * G + j], n_f = gt[(int64_t)t * G + H + j];
            n_g = gt[(int64_t)t * G + 2 * H + j], n_o = gt[(int64_t)t * G + 3 * H + j];
            n_c = ct[(int64_t)t * H + j];
            n_cp = s > 0 ? ct[(int64_t)tp * H + j] : 0.f;
            n_hp = s > 0 ? out[orow + tp] : 0.f;
            n_do = dout[orow + t];
        }
    };
    fetch(len - 1);
    for (int s = len - 1; s >= 0; --s) {
        const int t = dir == 0 ? s : len - 1 - s;  // time index of processing step s
        const float ig_ = n_i, fg = n_f, gg = n_g, og = n_o, c = n_c, cp = n_cp, hp = n_hp, dov = n_do;
        fetch(s - 1);  // next timestep's inputs, in flight under this one
        if (j < H) {
            const float e2 = __expf(-2.f * c);
            const float tc = 2.f * __builtin_amdgcn_rcpf(1.f + e2) - 1.f;  // tanh(c)
            const float dh = dov + dhc[j];
            const float d_o = dh * tc;
            const float dc = dh * og * (1.f - tc * tc) + dcc;
            const float di = dc * gg, df = dc * cp, dgg = dc * ig_;
            dcc = dc * fg;
            const float a0 = di * ig_ * (1.f - ig_), a1 = df * fg * (1.f - fg), a2 = dgg * (1.f - gg * gg),
                        a3 = d_o * og * (1.f - og);
            // gate row q*H + j lives in group (q*H + j) / 32, slot (q*H + j) % 32
            dg[((0 * H + j) >> 5) * GS + (j & 31)] = a0;
            dg[((1 * H + j) >> 5) * GS + (j & 31)] = a1;
            dg[((2 * H + j) >> 5) * GS + (j & 31)] = a2;
            dg[((3 * H + j) >> 5) * GS + (j & 31)] = a3;
            dgx[(int64_t)t * G + j] = a0;
            dgx[(int64_t)t * G + H + j] = a1;
            dgx[(int64_t)t * G + 2 * H + j] = a2;
            dgx[(int64_t)t * G + 3 * H + j] = a3;
            hprev[(int64_t)t * H + j] = hp;
        }
        lds_barrier();
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int g = 0; g < 32; g += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&dg[ig * GS + g]);
            p0 = fmaf(w[0][g], v.x, p0), p1 = fmaf(w[1][g], v.x, p1), p2 = fmaf(w[2][g], v.x, p2), p3 = fmaf(w[3][g], v.x, p3);
            p0 = fmaf(w[0][g + 1], v.y, p0), p1 = fmaf(w[1][g + 1], v.y, p1), p2 = fmaf(w[2][g + 1], v.y, p2),
            p3 = fmaf(w[3][g + 1], v.y, p3);
            p0 = fmaf(w[0][g + 2], v.z, p0), p1 = fmaf(w[1][g + 2], v.z, p1), p2 = fmaf(w[2][g + 2], v.z, p2),
            p3 = fmaf(w[3][g + 2], v.z, p3);
            p0 = fmaf(w[0][g + 3], v.w, p0), p1 = fmaf(w[1][g + 3], v.w, p1), p2 = fmaf(w[2][g + 3], v.w, p2),
            p3 = fmaf(w[3][g + 3], v.w, p3);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {  // the 16 lanes of an output group are consecutive
            p0 += __shfl_xor(p0, off, 64);
            p1 += __shfl_xor(p1, off, 64);
            p2 += __shfl_xor(p2, off, 64);
            p3 += __shfl_xor(p3, off, 64);
        }
        if (ig == 0) {  // dhc of this step was consumed before the first barrier: safe to overwrite
            dhc[ko * 4] = p0;
            dhc[ko * 4 + 1] = p1;
            dhc[ko * 4 + 2] = p2;
            dhc[ko * 4 + 3] = p3;
        }
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// CBRA backward (map_encoder.py:13-20): out = avgpool2(relu(y*scale + shift)), BatchNorm in train
// mode (scale = gamma*rstd, shift = beta - mean*scale) or eval mode.
// stats: per channel S1 = sum dz, S2 = sum dz*xhat with dz = dout/4 * (z>0), xhat = (y-mean)*rstd.
// apply: train: dy = gamma*rstd*(dz - S1/M - xhat*S2/M); eval: dy = dz*scale.
// dgamma = S2 (train) / sum dz*(y-rm)*rsqrt(rv+eps) (eval, same formula with mean=rm, rstd=that);
// dbeta = S1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cbra_bwd_stats(const float* __restrict__ dout, const float* __restrict__ y,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, int N, int C, int H, int W,
                                                        int imgs_per_split, double* __restrict__ partial) {
    // The two channel sums are accumulated in DOUBLE.  They become the per-channel means that BatchNorm's backward
    // subtracts from every element, so an error e in one of them is the SAME for all N*H*W elements of the channel
    // and adds up coherently in the conv's weight gradient (sum over pixels of x * e): with float accumulators the
    // weight gradients of the map CNN were 2e-3 of their largest element away from a float64 run of the reference
    // arithmetic (the reference's own fp32 autograd: 4e-3), with double 1e-4 (tests/test_gpu_train.py).
    __shared__ double red[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int i0 = sp * imgs_per_split, i1 = min(N, i0 + imgs_per_split);
    const int Ho = H / 2, Wo = W / 2, HW = H * W;
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    double s1 = 0.0, s2 = 0.0;
    if ((W & 3) == 0 && (((uintptr_t)y | (uintptr_t)dout) & 15) == 0) {
        // four consecutive pixels of a row per thread: one float4 of y, one float2 of the pooled gradient
        double t1 = 0.0, t2 = 0.0;
        for (int img = i0; img < i1; ++img) {
            const float* yp = y + ((int64_t)img * C + c) * HW;
            const float* dp = dout + ((int64_t)img * C + c) * Ho * Wo;
            for (int i = threadIdx.x * 4; i < HW; i += 1024) {
                const int h = i / W, w = i - h * W;
                const float4 yv = *reinterpret_cast<const float4*>(yp + i);
                const float2 dv = *reinterpret_cast<const float2*>(dp + (h >> 1) * Wo + (w >> 1));
                const float d0 = fmaf(yv.x, sc, sh) > 0.f ? 0.25f * dv.x : 0.f;
                const float d1 = fmaf(yv.y, sc, sh) > 0.f ? 0.25f * dv.x : 0.f;
                const float d2 = fmaf(yv.z, sc, sh) > 0.f ? 0.25f * dv.y : 0.f;
                const float d3 = fmaf(yv.w, sc, sh) > 0.f ? 0.25f * dv.y : 0.f;
                s1 += (double)d0 + (double)d1;
                t1 += (double)d2 + (double)d3;
                s2 += (double)(d0 * ((yv.x - mu) * rs)) + (double)(d1 * ((yv.y - mu) * rs));
                t2 += (double)(d2 * ((yv.z - mu) * rs)) + (double)(d3 * ((yv.w - mu) * rs));
            }
        }
        s1 += t1;
        s2 += t2;
    } else {
        for (int img = i0; img < i1; ++img) {
            const float* yp = y + ((int64_t)img * C + c) * HW;
            const float* dp = dout + ((int64_t)img * C + c) * Ho * Wo;
            for (int i = threadIdx.x; i < HW; i += 256) {
                int h = i / W, w = i - h * W;
                float yv = yp[i];
                float dz = fmaf(yv, sc, sh) > 0.f ? 0.25f * dp[(h >> 1) * Wo + (w >> 1)] : 0.f;
                s1 += (double)dz;
                s2 += (double)(dz * ((yv - mu) * rs));
            }
        }
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * S + sp) * 2] = s1;      // -> dbeta
        partial[((int64_t)c * S + sp) * 2 + 1] = s2;  // -> dgamma
    }
}

// one 64-lane wave per channel: the S double partials of both sums in a fixed order -> dbeta, dgamma (float)
__global__ __launch_bounds__(64) void k_cbra_bwd_final(const double* __restrict__ partial, int S, int C, float* __restrict__ dbeta,
                                                       float* __restrict__ dgamma) {
    const int c = blockIdx.x, l = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int sp = l; sp < S; sp += 64) {
        a += partial[((int64_t)c * S + sp) * 2];
        b += partial[((int64_t)c * S + sp) * 2 + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if (l == 0) {
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
    }
}

// four pixels of a row per thread (W % 4 == 0, 16-byte aligned tensors)
__global__ __launch_bounds__(256) void k_cbra_bwd_apply4(const float* __restrict__ dout, const float* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ dgamma,
                                                         const float* __restrict__ dbeta, int N, int C, int H, int W,
                                                         int train, float* __restrict__ dy) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int HW = H * W, Ho = H / 2, Wo = W / 2;
    if (idx >= (int64_t)N * C * HW) return;
    const int i = (int)(idx % HW);
    const int nc = (int)(idx / HW);
    const int c = nc % C;
    const int h = i / W, w = i - h * W;
    const float4 yv = *reinterpret_cast<const float4*>(y + idx);
    const float2 dv = *reinterpret_cast<const float2*>(dout + (int64_t)nc * Ho * Wo + (h >> 1) * Wo + (w >> 1));
    const float sc = scale[c], sh = shift[c];
    float yy[4] = {yv.x, yv.y, yv.z, yv.w};
    float dd[4] = {dv.x, dv.x, dv.y, dv.y};
    float o[4];
    const float M = (float)N * (float)HW;
    const float mu = train ? mean[c] : 0.f, rs = train ? rstd[c] : 0.f;
    const float b_m = train ? dbeta[c] / M : 0.f, g_m = train ? dgamma[c] / M : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dz = fmaf(yy[k], sc, sh) > 0.f ? 0.25f * dd[k] : 0.f;
        if (train) {
            const float xhat = (yy[k] - mu) * rs;
            o[k] = sc * (dz - b_m - xhat * g_m);
        } else {
            o[k] = dz * sc;
        }
    }
    *reinterpret_cast<float4*>(dy + idx) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void k_cbra_bwd_apply(const float* __restrict__ dout, const float* __restrict__ y,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta, int N, int C, int H, int W,
                                                        int train, float* __restrict__ dy) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W, Ho = H / 2, Wo = W / 2;
    if (idx >= (int64_t)N * C * HW) return;
    int i = (int)(idx % HW);
    int nc = (int)(idx / HW);
    int c = nc % C;
    int h = i / W, w = i - h * W;
    float yv = y[idx];
    float sc = scale[c];
    float dz = fmaf(yv, sc, shift[c]) > 0.f ? 0.25f * dout[(int64_t)nc * Ho * Wo + (h >> 1) * Wo + (w >> 1)] : 0.f;
    float v;
    if (train) {
        float M = (float)N * (float)HW;
        float xhat = (yv - mean[c]) * rstd[c];
        v = sc * (dz - dbeta[c] / M - xhat * dgamma[c] / M);
    } else {
        v = dz * sc;
    }
    dy[idx] = v;
}

// embedding gradient: table_grad[token[r]] += d[r] (atomic; padding row skipped)
__global__ __launch_bounds__(256) void k_embedding_scatter_add(const int64_t* __restrict__ tokens,
                                                               const float* __restrict__ d, int rows, int E, int V,
                                                               int padding_idx, float* __restrict__ grad) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * E) return;
    int r = (int)(idx / E), e = (int)(idx % E);
    int64_t tok = tokens[r];
    if (tok < 0 || tok >= V || tok == padding_idx) return;
    atomicAdd(&grad[tok * E + e], d[idx]);
}

// prev-action embedding gradient, deterministic: block per embedding row a; thread (part, e) sums every
// 8th batch row that selected a, the 8 parts meet in LDS in a fixed order.  (One thread per (a,e) walking
// all rows serially took 0.18 ms on 512 rows.)  Requires E <= 32.
__global__ __launch_bounds__(256) void k_prev_action_embed_bwd(const int64_t* __restrict__ prev_actions,
                                                               const uint8_t* __restrict__ mask,
                                                               const float* __restrict__ d1, int64_t ld1,
                                                               const float* __restrict__ d2, int64_t ld2, int rows,
                                                               int E, int n_emb, float* __restrict__ grad) {
    __shared__ float part[8][32];
    const int a = blockIdx.x, e = threadIdx.x & 31, pt = threadIdx.x >> 5;
    float s = 0.f;
    if (e < E) {
        for (int r = pt; r < rows; r += 8) {
            int64_t ar = (int64_t)(((float)prev_actions[r] + 1.f) * (float)(mask[r] ? 1 : 0));
            if (ar < 0) ar = 0;
            if (ar >= n_emb) ar = n_emb - 1;
            if (ar == a) s += d1[(int64_t)r * ld1 + e] + (d2 ? d2[(int64_t)r * ld2 + e] : 0.f);
        }
    }
    part[pt][e] = s;
    __syncthreads();
    if (pt == 0 && e < E) {
        float t = part[0][e];
        for (int i = 1; i < 8; ++i) t += part[i][e];
        grad[a * E + e] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Inflection-weighted cross-entropy (base_il_trainer.py:201-204):
//   ce[t,n] = -log_softmax(logits[t,n])[target];  loss = mean_n( sum_t w ce / sum_t w )
// One block; also writes dlogits = d loss / d logits * loss_scale.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ce_iw_loss(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ targets,
                                                     const float* __restrict__ weights, int T, int N, int A,
                                                     float loss_scale, float* __restrict__ loss_out,
                                                     float* __restrict__ dlogits) {
    // one wave per trajectory n, lanes over the timesteps (a thread per n walked its T steps serially:
    // 0.11 ms for 8 x 64); every reduction is a fixed tree, so the result is deterministic
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float total = 0.f;
    for (int n = wave; n < N; n += nw) {
        float wsum = 0.f;
        for (int t = lane; t < T; t += 64) wsum += weights[t * N + n];
        wsum = wave_sum(wsum);
        float acc = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float* l = logits + ((int64_t)t * N + n) * A;
            float mx = l[0];
            for (int a = 1; a < A; ++a) mx = fmaxf(mx, l[a]);
            float se = 0.f;
            for (int a = 0; a < A; ++a) se += expf(l[a] - mx);
            float lse = mx + logf(se);
            int tg = (int)targets[t * N + n];
            float w = weights[t * N + n];
            acc += w * (lse - l[tg]);
            float coef = loss_scale * w / (wsum * (float)N);
            for (int a = 0; a < A; ++a)
                dlogits[((int64_t)t * N + n) * A + a] = coef * (expf(l[a] - lse) - (a == tg ? 1.f : 0.f));
        }
        acc = wave_sum(acc);
        total += acc / wsum;
    }
    // every lane of a wave holds the same `total`; combine the waves in a fixed order
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; ++i) s += red[i];
        loss_out[0] = s / (float)N;
    }
}

// ------------------------------------------------------------------------------------------
// Progress monitor with the reference's (TN,) x (TN,1) broadcast (quirk Q7, map_cma_policy.py:355-361):
//   hat[i] = tanh(pre[i]);  L[j][i] = (hat[i] - p[j])^2   (TN x TN)
// fwd writes L; bwd: dpre[i] = (sum_j dL[j][i] * 2 (hat[i] - p[j])) * (1 - hat[i]^2).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pm_loss_fwd(const float* __restrict__ pre, const float* __restrict__ p,
                                                     int n, float* __restrict__ hat, float* __restrict__ Lm) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    int j = (int)(idx / n), i = (int)(idx % n);
    float h = tanhf(pre[i]);
    if (j == 0) hat[i] = h;
    float d = h - p[j];
    Lm[idx] = d * d;
}
// dpre[i] = (1 - hat_i^2) * sum_j dL[j][i] * 2 (hat_i - p_j).  A block owns 16 columns; its 16 row slices walk j = s,
// s + 16, ... and are summed in slice order through LDS (deterministic).  (One thread per column looping over all n
// rows left the (512, 512) case on two workgroups: 118 us of serial load latency per update.)
__global__ __launch_bounds__(256) void k_pm_loss_bwd(const float* __restrict__ dL, const float* __restrict__ hat,
                                                     const float* __restrict__ p, int n, float* __restrict__ dpre) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float s = 0.f;
    if (i < n) {
        const float h = hat[i];
        for (int j = sl; j < n; j += 16) s += dL[(int64_t)j * n + i] * 2.f * (h - p[j]);
    }
    part[sl][c] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][c];
        const float h = hat[i];
        dpre[i] = t * (1.f - h * h);
    }
}

// The progress monitor's aux loss as the update step consumes it (quirk Q7 + AuxLosses.reduce; map_cma_policy.py:355-366,
// aux_losses.py:22-29): mean over the SELECTED entries of L[j][i] = (tanh(pre_i) - p_j)^2, where the (n,) mask selects
// whole columns i.  One workgroup; the (n, n) matrix is never written: thread i walks the p_j (LDS) for its columns,
// keeping sum_j (h_i - p_j)^2 for the loss and sum_j (h_i - p_j) for the gradient, then a fixed-order block sum.
//   out2[0] = mean, out2[1] = number of selected entries (n * #masked).  An empty selection gives 0 / 0 = NaN, like
//   torch's mean of an empty tensor.
__global__ __launch_bounds__(1024) void k_pm_masked_mean_fwd(const float* __restrict__ pre, const float* __restrict__ p,
                                                             const uint8_t* __restrict__ mask, int n,
                                                             float* __restrict__ hat, float* __restrict__ dsum,
                                                             float* __restrict__ out2) {
    extern __shared__ float s_p[];   // n progress values, then 32 reduction slots
    float* red = s_p + n;
    for (int j = threadIdx.x; j < n; j += 1024) s_p[j] = p[j];
    __syncthreads();
    float tot = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float h = tanhf(pre[i]);
        float sq = 0.f, d1 = 0.f;
        for (int j = 0; j < n; ++j) {
            const float d = h - s_p[j];
            sq = fmaf(d, d, sq);
            d1 += d;
        }
        hat[i] = h;
        dsum[i] = d1;
        if (mask[i]) {
            tot += sq;
            cnt += (float)n;
        }
    }
    // block sums in a fixed order: wave shuffles, then the 16 wave totals by thread 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = tot;
        red[16 + (threadIdx.x >> 6)] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, c = 0.f;
        for (int w = 0; w < 16; ++w) {
            t += red[w];
            c += red[16 + w];
        }
        out2[0] = t / c;
        out2[1] = c;
    }
}
// d(mean)/d(pre_i) = mask_i * 2 * sum_j (h_i - p_j) * (1 - h_i^2) / count, times the upstream gradient (a device scalar)
__global__ __launch_bounds__(256) void k_pm_masked_mean_bwd(const float* __restrict__ gout, const float* __restrict__ hat,
                                                            const float* __restrict__ dsum,
                                                            const uint8_t* __restrict__ mask,
                                                            const float* __restrict__ out2, int n, float alpha,
                                                            float* __restrict__ dpre) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float h = hat[i];
    dpre[i] = mask[i] ? gout[0] * alpha * 2.f * dsum[i] * (1.f - h * h) / out2[1] : 0.f;
}

// ------------------------------------------------------------------------------------------
// Adam on a flat fp32 bucket (torch.optim.Adam defaults semantics: bias-corrected, eps outside sqrt
// of the corrected second moment, no weight decay / amsgrad; base_il_trainer.py:78-94, 213-215).
// lr may differ per segment: lr_per_elem == nullptr -> scalar lr.  grad_scale folds the 1/world of
// the data-parallel mean.  Grads are zeroed in the same pass (optimizer.zero_grad()).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam_flat(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, const int* __restrict__ seg_of, const float* seg_lr,
                                                   float b1, float b2, float eps, float bc1, float bc2,
                                                   float grad_scale, int zero_grad, const unsigned* __restrict__ guard) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // `guard`: a device word that is non-zero when the gradients in `g` are void (the persistent sequence GRU of this update
    // timed out: ivln_seq_sync_status' sticky word).  The step is then skipped ON THE DEVICE - parameters, moments and
    // gradients stay as they are - and the host, which learns of it after its next synchronisation, runs the update again.
    if (guard && *guard != 0u) return;
    float gi = g[i] * grad_scale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float l = seg_of ? seg_lr[seg_of[i]] : lr;
    float step = l / bc1;
    float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = p[i] - step * (mi / denom);
    if (zero_grad) g[i] = 0.f;
}

// dst[u][m] = sum over rows with index[row] == u of src[row][m] in a fixed order: the per-row gradients of a key/value
// image shared by several rows (k_attn_bwd with row_index) folded onto the image.  A block = 32 float4 columns x 8 row
// chunks: every thread scans one eighth of the rows (one thread walking all 512 rows was a chain of 64 dependent-issue
// loads, 48 us per call), the eight partial sums are added in chunk order through LDS.
__global__ __launch_bounds__(256) void k_index_sum(const float* __restrict__ src, const int* __restrict__ index, int rows,
                                                   int64_t M4, int U, float* __restrict__ dst) {
    __shared__ float4 part[8][32];
    const int col = threadIdx.x & 31, chunk = threadIdx.x >> 5, u = blockIdx.y;
    const int64_t m = (int64_t)blockIdx.x * 32 + col;
    const int rpc = (rows + 7) / 8, r0 = chunk * rpc, r1 = min(rows, r0 + rpc);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M4) {
        for (int r = r0; r < r1; ++r) {
            if (index[r] == u) {
                const float4 v = reinterpret_cast<const float4*>(src)[(int64_t)r * M4 + m];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    part[chunk][col] = acc;
    __syncthreads();
    if (chunk == 0 && m < M4) {
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            const float4 v = part[c][col];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(dst)[(int64_t)u * M4 + m] = acc;
    }
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP)

extern "C" {

int ivln_relu_bwd_f32(const float* dy, const float* y, float* dx, int rows, int cols, int64_t ld_dy, int64_t ld_y,
                      int64_t ld_dx, void* stream) {
    hipLaunchKernelGGL(k_relu_bwd, dim3(nblk((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, dy, y, dx,
                       rows, cols, ld_dy, ld_y, ld_dx);
    return LAUNCH_OK();
}

int ivln_add2d_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int rows, int cols,
                   void* stream) {
    hipLaunchKernelGGL(k_add2d, dim3(nblk((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                       y, ldy, rows, cols);
    return LAUNCH_OK();
}

int ivln_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, float* ws,
                    int64_t ws_floats, void* stream) {
    if (rows <= 0 || cols <= 0) return IVLN_E_INVALID;
    int splits = (rows + 255) / 256;
    if (splits > 128) splits = 128;
    if ((int64_t)splits * cols > ws_floats) splits = (int)(ws_floats / cols);
    if (splits < 1) return IVLN_E_INVALID;
    int rps = (rows + splits - 1) / splits;
    splits = (rows + rps - 1) / rps;
    hipLaunchKernelGGL(k_colsum_partial, dim3((cols + 63) / 64, splits), dim3(256), 0, (hipStream_t)stream, x, ld, rows,
                       cols, rps, ws);
    hipLaunchKernelGGL(k_colsum_final, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, splits, cols,
                       out, accumulate);
    return LAUNCH_OK();
}

int ivln_colsum_multi_f32(const float* const* xs, const int64_t* lds, const int* rows, const int* cols, float* const* outs,
                          int n, float* ws, int64_t ws_floats, void* stream) {
    if (!xs || !lds || !rows || !cols || !outs || !ws || n <= 0 || n > 32) return IVLN_E_INVALID;
    ColsumJobs J;
    J.n = n;
    int blocks = 0, colsum = 0;
    int64_t off = 0;
    for (int j = 0; j < n; ++j) {
        if (!xs[j] || !outs[j] || rows[j] <= 0 || cols[j] <= 0) return IVLN_E_INVALID;
        int splits = (rows[j] + 255) / 256;  // (the split rule of ivln_colsum_f32: same partials, same sums)
        if (splits > 128) splits = 128;
        const int rps = (rows[j] + splits - 1) / splits;
        splits = (rows[j] + rps - 1) / rps;
        J.x[j] = xs[j], J.out[j] = outs[j], J.ld[j] = lds[j], J.rows[j] = rows[j], J.cols[j] = cols[j];
        J.splits[j] = splits, J.rps[j] = rps, J.first_block[j] = blocks, J.first_col[j] = colsum, J.ws_off[j] = off;
        blocks += ((cols[j] + 63) / 64) * splits;
        colsum += cols[j];
        off += (int64_t)splits * cols[j];
    }
    if (off > ws_floats) return IVLN_E_INVALID;
    J.first_block[n] = blocks, J.first_col[n] = colsum;
    hipLaunchKernelGGL(k_colsum_multi_partial, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J, ws);
    hipLaunchKernelGGL(k_colsum_multi_final, dim3((colsum + 255) / 256), dim3(256), 0, (hipStream_t)stream, J, ws);
    return LAUNCH_OK();
}

static int chan_splits(int N, int HW, int C, int K, int64_t ws_floats) {
    // >= 4 K elements per block, up to 256 blocks per channel (16 K / 64 left the BatchNorm backward statistics at 1.5-1.9
    // TB/s on the 268 MB layer: 32 x 64 blocks of 32 serial trips each; 90 -> 47 us per launch with 256)
    constexpr int gran = 4096;  // tuning
    constexpr int cap = 256;
    int S = (int)(((int64_t)N * HW + gran - 1) / gran);
    if (S > N) S = N;
    if (S > cap) S = cap;
    if ((int64_t)S * C * K > ws_floats) S = (int)(ws_floats / ((int64_t)C * K));
    return S < 1 ? 1 : S;
}

int ivln_nchw_chansum_f32(const float* x, int N, int C, int HW, float* out, float* ws, int64_t ws_floats,
                          void* stream) {
    if (!ws || ws_floats < C) return IVLN_E_INVALID;
    // Short rows (the Conv1d(k=1) bias gradients of the update: HW = 16 positions or an 80-token axis): a block per
    // channel would walk N images with HW/4 lanes busy (99 us for 12 MB at T*N = 512 rows).  The tensor is an
    // (N, C*HW) row-major matrix: column sums by the two-stage colsum kernels into the workspace tail, then each
    // channel adds its HW columns (fixed order throughout).
    const int64_t cols = (int64_t)C * HW;
    const int64_t splits_max = (N + 255) / 256 < 128 ? (N + 255) / 256 : 128;   // what ivln_colsum_f32 will use
    if (HW <= 128 && cols <= 1 << 20 && ws_floats >= (splits_max + 1) * cols) {
        float* col = ws + splits_max * cols;   // colsum's partials live in ws[0, splits * cols)
        const int rc = ivln_colsum_f32(x, cols, N, (int)cols, col, 0, ws, splits_max * cols, stream);
        if (rc != IVLN_OK) return rc;
        hipLaunchKernelGGL(k_chan_final, dim3(C), dim3(64), 0, (hipStream_t)stream, col, HW, C, 1, out,
                           (float*)nullptr);
        return LAUNCH_OK();
    }
    int S = chan_splits(N, HW, C, 1, ws_floats);
    const int ips = (N + S - 1) / S;
    S = (N + ips - 1) / ips;
    hipLaunchKernelGGL(k_nchw_chansum_partial, dim3(C, S), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, ips, ws);
    hipLaunchKernelGGL(k_chan_final, dim3(C), dim3(64), 0, (hipStream_t)stream, ws, S, C, 1, out,
                       (float*)nullptr);
    return LAUNCH_OK();
}

int ivln_transpose_f32(const float* x, float* y, int R, int C, void* stream) {
    hipLaunchKernelGGL(k_transpose, dim3((C + 15) / 16, (R + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, y, R, C);
    return LAUNCH_OK();
}

int ivln_weight_flip_transpose_f32(const float* w, float* wt, int O, int I, int KH, int KW, void* stream) {
    hipLaunchKernelGGL(k_weight_flip_transpose, dim3(nblk((int64_t)O * I * KH * KW)), dim3(256), 0,
                       (hipStream_t)stream, w, wt, O, I, KH, KW);
    return LAUNCH_OK();
}

int ivln_attn_bwd_idx_f32(const float* dout, int64_t ld_dout, const float* attn, const float* q, int64_t ldq,
                      const float* k, int64_t k_img_stride, const float* v, int64_t v_img_stride, float scale,
                      int rows, int Ck, int Cv, int I, float* dq, int64_t ld_dq, float* dk, int64_t dk_img_stride,
                      float* dv, int64_t dv_img_stride, const int* row_index, void* stream) {
    if (I > 512 || Ck > 1024 || Cv > 1024) return IVLN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_attn_bwd, dim3(rows), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, attn, q, ldq, k,
                       k_img_stride, v, v_img_stride, scale, Ck, Cv, I, dq, ld_dq, dk, dk_img_stride, dv,
                       dv_img_stride, row_index);
    return LAUNCH_OK();
}

int ivln_attn_bwd_f32(const float* dout, int64_t ld_dout, const float* attn, const float* q, int64_t ldq,
                      const float* k, int64_t k_img_stride, const float* v, int64_t v_img_stride, float scale,
                      int rows, int Ck, int Cv, int I, float* dq, int64_t ld_dq, float* dk, int64_t dk_img_stride,
                      float* dv, int64_t dv_img_stride, void* stream) {
    return ivln_attn_bwd_idx_f32(dout, ld_dout, attn, q, ldq, k, k_img_stride, v, v_img_stride, scale, rows, Ck, Cv, I, dq, ld_dq,
                                 dk, dk_img_stride, dv, dv_img_stride, nullptr, stream);
}

int ivln_gru_bwd_elem_f32(const float* dout, int64_t ld_dout, const float* dh_carry, const float* r, const float* z,
                          const float* n, const float* ghn, const float* h_prev, int64_t ldh, const uint8_t* mask,
                          int rows, int H, float* dgi, float* dgh, float* dhz, float* hp_out, void* stream) {
    hipLaunchKernelGGL(k_gru_bwd_elem, dim3(nblk((int64_t)rows * H)), dim3(256), 0, (hipStream_t)stream, dout, ld_dout,
                       dh_carry, r, z, n, ghn, h_prev, ldh, mask, rows, H, dgi, dgh, dhz, hp_out);
    return LAUNCH_OK();
}

int ivln_gru_bwd_step_f32(const float* dgh_t, int64_t ld_dgh, const float* whh_t, const uint8_t* mask_t,
                          const float* dout_prev, int64_t ld_dout, const float* r, const float* z, const float* n,
                          const float* ghn, const float* h_prev, int64_t ldh, const uint8_t* mask_prev, int rows, int H,
                          float* dhz, float* dgi_prev, float* dgh_prev, float* hp_prev, void* stream) {
    if ((H & 3) || (ld_dgh & 3) || rows <= 0) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_gru_bwd_step, dim3(H), dim3(256), 0, (hipStream_t)stream, dgh_t, ld_dgh, whh_t, mask_t,
                       dout_prev, ld_dout, r, z, n, ghn, h_prev, ldh, mask_prev, rows, H, dhz, dgi_prev, dgh_prev,
                       hp_prev);
    return LAUNCH_OK();
}

/* BPTT of ivln_cma_seq_fwd_f32 in one call: the element part of step T-1, then T-1 fused (carry of step t + element
 * part of step t-1) launches, enqueued from C (see the forward).  whh_t = W_hh^T (H, 3H).  Outputs dgi / dgh
 * (T*N, 3H), hp = h_prev * mask (T*N, H); dhz (N, H) is scratch. */
int ivln_cma_seq_bwd_f32(const float* d_out, int64_t ld_dout, const float* r, const float* z, const float* n,
                         const float* ghn, const float* out, int64_t ld_out, const float* h0, int64_t ld_h0,
                         const uint8_t* masks, const float* whh_t, int T, int N, int H, float* dgi, float* dgh, float* hp,
                         float* dhz, void* sync_ws, void* stream) {
    if (!d_out || !r || !out || !h0 || !masks || !whh_t || !dgi || !dgh || !hp || !dhz || T <= 0 || N <= 0 || (H & 3))
        return IVLN_E_INVALID;
    if (sync_ws && T > 1 && ivln_cma_seq_persistent_ok(N, H, 1)) {   // one persistent launch (gru_seq.hip)
        const int rc = ivln_gru_seq_bwd_persistent(d_out, ld_dout, r, z, n, ghn, out, ld_out, h0, ld_h0, masks, whh_t, T, N,
                                                   dgi, dgh, hp, sync_ws, stream);
        if (rc != IVLN_E_UNSUPPORTED) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    auto hprev = [&](int t, int64_t& ld) -> const float* {  // hidden state entering step t
        ld = t == 0 ? ld_h0 : ld_out;
        return t == 0 ? h0 : out + (int64_t)(t - 1) * N * ld_out;
    };
    {
        const int64_t r0 = (int64_t)(T - 1) * N;
        int64_t ld;
        const float* hpv = hprev(T - 1, ld);
        hipLaunchKernelGGL(k_gru_bwd_elem, dim3(nblk((int64_t)N * H)), dim3(256), 0, s, d_out + r0 * ld_dout, ld_dout,
                           (const float*)nullptr, r + r0 * H, z + r0 * H, n + r0 * H, ghn + r0 * H, hpv, ld, masks + r0, N,
                           H, dgi + r0 * 3 * H, dgh + r0 * 3 * H, dhz, hp + r0 * H);
    }
    for (int t = T - 1; t > 0; --t) {
        const int64_t rt = (int64_t)t * N, rp = (int64_t)(t - 1) * N;
        int64_t ld;
        const float* hpp = hprev(t - 1, ld);
        hipLaunchKernelGGL(k_gru_bwd_step, dim3(H), dim3(256), 0, s, dgh + rt * 3 * H, (int64_t)3 * H, whh_t, masks + rt,
                           d_out + rp * ld_dout, ld_dout, r + rp * H, z + rp * H, n + rp * H, ghn + rp * H, hpp, ld,
                           masks + rp, N, H, dhz, dgi + rp * 3 * H, dgh + rp * 3 * H, hp + rp * H);
    }
    return LAUNCH_OK();
}

int ivln_linear_skinny_ex_f32(const float* x, int64_t ldx, const float* W, const float* add, int64_t ld_add,
                              const uint8_t* rowmask, float* y, int64_t ldy, int rows, int K, int O, void* stream) {
    if ((K & 3) || (ldx & 3)) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_linear_skinny_ex, dim3(O), dim3(256), 0, (hipStream_t)stream, x, ldx, W, add,
                       ld_add, rowmask, y, ldy, rows, K, O);
    return LAUNCH_OK();
}

int ivln_lstm_bidir_bwd_f32(const float* dout, const float* out, const float* gates, const float* cs,
                            const float* whh_f, const float* whh_r, const int* lengths, int B, int L, int H,
                            float* dgx_f, float* dgx_r, float* hprev_f, float* hprev_r, void* stream) {
    if (H != 128) return IVLN_E_UNSUPPORTED;
    hipLaunchKernelGGL((k_lstm_bidir_bwd<128>), dim3(B, 2), dim3(512), 0, (hipStream_t)stream, dout, out, gates, cs,
                       whh_f, whh_r, lengths, L, dgx_f, dgx_r, hprev_f, hprev_r);
    return LAUNCH_OK();
}

int ivln_cbra_bwd_f32(const float* dout, const float* y, const float* scale, const float* shift, const float* mean,
                      const float* rstd, int N, int C, int H, int W, int train, float* dgamma, float* dbeta,
                      float* dy, float* ws, int64_t ws_floats, void* stream) {
    if (!ws || ws_floats < (int64_t)4 * C + 2 || ((uintptr_t)ws & 3)) return IVLN_E_INVALID;
    double* wsd = reinterpret_cast<double*>(((uintptr_t)ws + 7) & ~(uintptr_t)7);  // partials are doubles: 4 floats per (channel, split)
    int S = chan_splits(N, H * W, C, 4, ws_floats - 2);
    const int ips = (N + S - 1) / S;
    S = (N + ips - 1) / ips;
    hipLaunchKernelGGL(k_cbra_bwd_stats, dim3(C, S), dim3(256), 0, (hipStream_t)stream, dout, y, scale, shift, mean,
                       rstd, N, C, H, W, ips, wsd);
    hipLaunchKernelGGL(k_cbra_bwd_final, dim3(C), dim3(64), 0, (hipStream_t)stream, wsd, S, C, dbeta, dgamma);
    if ((W & 3) == 0 && (((uintptr_t)y | (uintptr_t)dout | (uintptr_t)dy) & 15) == 0)
        hipLaunchKernelGGL(k_cbra_bwd_apply4, dim3(nblk((int64_t)N * C * H * W / 4)), dim3(256), 0, (hipStream_t)stream,
                           dout, y, scale, shift, mean, rstd, dgamma, dbeta, N, C, H, W, train, dy);
    else
        hipLaunchKernelGGL(k_cbra_bwd_apply, dim3(nblk((int64_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                           dout, y, scale, shift, mean, rstd, dgamma, dbeta, N, C, H, W, train, dy);
    return LAUNCH_OK();
}

int ivln_embedding_scatter_add_f32(const int64_t* tokens, const float* d, int rows, int E, int V, int padding_idx,
                                   float* grad, void* stream) {
    hipLaunchKernelGGL(k_embedding_scatter_add, dim3(nblk((int64_t)rows * E)), dim3(256), 0, (hipStream_t)stream,
                       tokens, d, rows, E, V, padding_idx, grad);
    return LAUNCH_OK();
}

int ivln_prev_action_embed_bwd_f32(const int64_t* prev_actions, const uint8_t* mask, const float* d1, int64_t ld1,
                                   const float* d2, int64_t ld2, int rows, int E, int n_emb, float* grad,
                                   void* stream) {
    if (E > 32 || n_emb <= 0) return IVLN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_prev_action_embed_bwd, dim3(n_emb), dim3(256), 0, (hipStream_t)stream, prev_actions, mask,
                       d1, ld1, d2, ld2, rows, E, n_emb, grad);
    return LAUNCH_OK();
}

int ivln_ce_iw_loss_f32(const float* logits, const int64_t* targets, const float* weights, int T, int N, int A,
                        float loss_scale, float* loss_out, float* dlogits, void* stream) {
    const int waves = N < 16 ? (N < 1 ? 1 : N) : 16;
    hipLaunchKernelGGL(k_ce_iw_loss, dim3(1), dim3(64 * waves), 0, (hipStream_t)stream, logits, targets, weights, T,
                       N, A, loss_scale, loss_out, dlogits);
    return LAUNCH_OK();
}

int ivln_pm_loss_fwd_f32(const float* pre, const float* progress, int n, float* hat, float* loss_matrix,
                         void* stream) {
    hipLaunchKernelGGL(k_pm_loss_fwd, dim3(nblk((int64_t)n * n)), dim3(256), 0, (hipStream_t)stream, pre, progress, n,
                       hat, loss_matrix);
    return LAUNCH_OK();
}

int ivln_pm_loss_bwd_f32(const float* dL, const float* hat, const float* progress, int n, float* dpre,
                         void* stream) {
    hipLaunchKernelGGL(k_pm_loss_bwd, dim3((n + 15) / 16), dim3(256), 0, (hipStream_t)stream, dL, hat, progress, n,
                       dpre);
    return LAUNCH_OK();
}

int ivln_pm_masked_mean_fwd_f32(const float* pre, const float* progress, const uint8_t* mask, int n, float* hat,
                                float* dsum, float* out2, void* stream) {
    if (!pre || !progress || !mask || !hat || !dsum || !out2 || n <= 0 || n > 15000) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_pm_masked_mean_fwd, dim3(1), dim3(1024), (size_t)(n + 32) * sizeof(float), (hipStream_t)stream,
                       pre, progress, mask, n, hat, dsum, out2);
    return LAUNCH_OK();
}

int ivln_pm_masked_mean_bwd_f32(const float* gout, const float* hat, const float* dsum, const uint8_t* mask,
                                const float* out2, int n, float alpha, float* dpre, void* stream) {
    if (!gout || !hat || !dsum || !mask || !out2 || !dpre || n <= 0) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_pm_masked_mean_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gout, hat, dsum,
                       mask, out2, n, alpha, dpre);
    return LAUNCH_OK();
}

int ivln_adam_step_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       const int* seg_of, const float* seg_lr, float beta1, float beta2, float eps, int step,
                       float grad_scale, int zero_grad, void* stream) {
    return ivln_adam_step_guarded_f32(params, grads, exp_avg, exp_avg_sq, n, lr, seg_of, seg_lr, beta1, beta2, eps, step, grad_scale,
                                      zero_grad, nullptr, stream);
}

int ivln_adam_step_guarded_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               const int* seg_of, const float* seg_lr, float beta1, float beta2, float eps, int step,
                               float grad_scale, int zero_grad, const void* guard, void* stream) {
    if (n <= 0 || step < 1) return IVLN_E_INVALID;
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(k_adam_flat, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, lr, seg_of, seg_lr, beta1, beta2, eps, bc1, bc2, grad_scale, zero_grad, (const unsigned*)guard);
    return LAUNCH_OK();
}

int ivln_index_sum_f32(const float* src, const int* index, int rows, int64_t M, int U, float* dst, void* stream) {
    if (!src || !index || !dst || rows <= 0 || U <= 0 || M <= 0 || (M & 3)) return IVLN_E_INVALID;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_index_sum, dim3((unsigned)((M / 4 + 31) / 32), U), dim3(256), 0, (hipStream_t)stream, src, index,
                       rows, M / 4, U, dst);
    return LAUNCH_OK();
}

}  // extern "C"
