// Shared pieces of the MFMA GEMM / direct-conv kernels: operand-mode enums and the fused epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ivln_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { AMODE_MK = 0, AMODE_KM = 1, AMODE_NCHW_P = 2 };
enum { BMODE_CONV = 0, BMODE_CONV1X1 = 1, BMODE_KN = 2, BMODE_NK = 3, BMODE_IM2COL_T = 4, BMODE_CONVT = 5,
       BMODE_CONV_K3 = 6, BMODE_CONV_K7 = 7 };  // 3x3 / 7x7, dilation 1: (ci,kh,kw) by constant division, no tables
enum { DMODE_NCHW = 0, DMODE_DENSE = 1, DMODE_NCHW_UP2 = 2 };

constexpr int BK = 16;

__device__ __forceinline__ void epilogue_store(const ivln_gemm_desc& p, int m, int n, float v) {
    int64_t addr;
    if (p.dmode == DMODE_NCHW) {
        int img = n / p.HoWo;
        int pp = n - img * p.HoWo;
        addr = ((int64_t)img * p.Ctot + m) * p.HoWo + pp;
    } else if (p.dmode == DMODE_NCHW_UP2) {
        int img = n / p.HoWo;
        int pp = n - img * p.HoWo;
        int ho = pp / p.Wout, wo = pp - ho * p.Wout;
        addr = (((int64_t)img * p.Ctot + m) * (2 * p.Hout) + 2 * ho + (int)p.sDm) * (2 * p.Wout) + 2 * wo + (int)p.sDn;
    } else {
        addr = (int64_t)m * p.sDm + (int64_t)n * p.sDn;
    }
    if (p.scale) v = fmaf(v, p.scale[m], p.shift[m]);
    else if (p.shift) v += p.shift[m];
    if (p.residual) v += p.residual[addr];
    if (p.accumulate) v += p.D[addr];
    if (p.relu) v = fmaxf(v, 0.f);
    p.D[addr] = v;
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
