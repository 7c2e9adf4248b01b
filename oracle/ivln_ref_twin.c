/*
 * TEST INFRASTRUCTURE ONLY - CPU twin of libivln_hip.so: the SAME C-ABI symbols as include/ivln_hip.h (SURVEY.md
 * section 8b, last column) with HOST pointers instead of device pointers, so that a test can run one `ivln_*` entry
 * point in both libraries on the same bytes and diff the results symbol for symbol.  Only tests/ load it
 * (oracle/libivln_ref.so); the product never does.
 *
 *   ivln_mapper_*      thin wrappers over the C restatement of the reference mapper (mapper_ref.c, pinned to the
 *                      reference's own MappingModule through tests/golden/mapper_*.npz and known_map.npz)
 *   ivln_gemm_f32      naive loops over the descriptor's operand modes: D[m][n] = sum_k A[m][k] B[k][n] as an fmaf
 *                      chain in k order + the fused epilogue (scale / shift, residual, accumulate, ReLU), image-grouped
 *                      weights included.  Restates what nn.Conv2d / nn.Linear compute at the call sites cited in
 *                      include/ivln_hip.h; pinned by tests/test_oracle_twin.py against torch.nn.functional.
 *   ivln_groupnorm_f32 two-pass GroupNorm (+ residual, + ReLU) over NCHW or split-K slabs
 *   ivln_strerror / ivln_version
 * Entry points of the device library that have no twin return IVLN_E_UNSUPPORTED here only if somebody adds a stub;
 * this file exports exactly the list above (tests/test_oracle_twin.py checks the signatures against the header).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ivln_hip.h"

/* ---- mapper_ref.c ---- */
typedef struct MapperRef MapperRef;
MapperRef *mapper_ref_create(int H, int W, double vfov_rad, double height_m, double width_m, double res_m);
void mapper_ref_destroy(MapperRef *m);
void mapper_ref_reset(MapperRef *m);
void mapper_ref_frames(int B, const float *pose, const double *orient, float *T, float *rot);
int mapper_ref_step(MapperRef *m, int B, const float *depth, const uint8_t *labels, const float *T, const float *pose,
                    const float *rot, const uint8_t *not_done, uint8_t *occ, uint8_t *sem);
void mapper_ref_clear_done(MapperRef *m, int B, const uint8_t *not_done);
int mapper_ref_raster(MapperRef *m, int B, const float *pose, const float *rot, uint8_t *occ, uint8_t *sem);
int mapper_ref_load_known(MapperRef *m, int b, const float *xyz, const uint8_t *semv, int64_t n);
int64_t mapper_ref_world_size(const MapperRef *m);
void mapper_ref_world_get(const MapperRef *m, float *xyz, int32_t *b, uint8_t *sem);

struct ivln_mapper {
    MapperRef *ref;
    int B_max;
};

const char *ivln_strerror(int code) {
    switch (code) {
        case IVLN_OK: return "ok";
        case IVLN_E_INVALID: return "invalid argument";
        case IVLN_E_HIP: return "HIP runtime error";
        case IVLN_E_KEYSPACE: return "mapper keep-highest key exceeds dense table capacity";
        case IVLN_E_CAPACITY: return "mapper world cloud capacity exceeded";
        case IVLN_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

int ivln_version(void) { return 1; }

int ivln_mapper_create(int B_max, int H, int W, double vfov_rad, double height_m, double width_m, double res_m,
                       int64_t world_capacity, int64_t table_cells, ivln_mapper **out) {
    (void)world_capacity;
    (void)table_cells;
    if (!out || B_max <= 0 || H <= 0 || W <= 0 || res_m <= 0) return IVLN_E_INVALID;
    ivln_mapper *m = (ivln_mapper *)calloc(1, sizeof(ivln_mapper));
    if (!m) return IVLN_E_INVALID;
    m->ref = mapper_ref_create(H, W, vfov_rad, height_m, width_m, res_m);
    m->B_max = B_max;
    *out = m;
    return IVLN_OK;
}

int ivln_mapper_destroy(ivln_mapper *m) {
    if (m) {
        mapper_ref_destroy(m->ref);
        free(m);
    }
    return IVLN_OK;
}

int ivln_mapper_set_launch_width(ivln_mapper *m, int local_blocks, int world_blocks) {  /* a launch hint: nothing on the host */
    return (!m || local_blocks < 0 || world_blocks < 0) ? IVLN_E_INVALID : IVLN_OK;
}

int ivln_mapper_reset(ivln_mapper *m, void *stream) {
    (void)stream;
    if (!m) return IVLN_E_INVALID;
    mapper_ref_reset(m->ref);
    return IVLN_OK;
}

int ivln_mapper_frames(const float *pose, const double *orientation, int B, float *T, float *rot, void *stream) {
    (void)stream;
    if (B <= 0) return IVLN_E_INVALID;
    mapper_ref_frames(B, pose, orientation, T, rot);
    return IVLN_OK;
}

int ivln_mapper_step(ivln_mapper *m, const float *depth, const uint8_t *labels, const float *T, const float *pose,
                     const float *rot, const uint8_t *not_done, int B, uint8_t *occ_out, uint8_t *sem_out, void *stream) {
    (void)stream;
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    return mapper_ref_step(m->ref, B, depth, labels, T, pose, rot, not_done, occ_out, sem_out) == 0 ? IVLN_OK : IVLN_E_INVALID;
}

int ivln_mapper_known_begin(ivln_mapper *m, const uint8_t *not_done, int B, void *stream) {
    (void)stream;
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    mapper_ref_clear_done(m->ref, B, not_done);
    return IVLN_OK;
}

int ivln_mapper_load_known(ivln_mapper *m, int b, const float *xyz, const uint8_t *sem, int64_t n, void *stream) {
    (void)stream;
    if (!m || b < 0 || b >= m->B_max || n < 0) return IVLN_E_INVALID;
    return mapper_ref_load_known(m->ref, b, xyz, sem, n) == 0 ? IVLN_OK : IVLN_E_INVALID;
}

int ivln_mapper_known_raster(ivln_mapper *m, const float *pose, const float *rot, int B, uint8_t *occ_out,
                             uint8_t *sem_out, void *stream) {
    (void)stream;
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    return mapper_ref_raster(m->ref, B, pose, rot, occ_out, sem_out) == 0 ? IVLN_OK : IVLN_E_INVALID;
}

int ivln_mapper_status(ivln_mapper *m, int64_t *world_n, void *stream) {
    (void)stream;
    if (!m) return IVLN_E_INVALID;
    if (world_n) *world_n = mapper_ref_world_size(m->ref);
    return IVLN_OK;
}

/* The CPU cloud is already in the reference's order: rank = position. */
int ivln_mapper_world_export(ivln_mapper *m, float *xyz, uint32_t *meta, int64_t *rank, int64_t max_n, int64_t *n_out,
                             void *stream) {
    (void)stream;
    if (!m || !n_out) return IVLN_E_INVALID;
    int64_t n = mapper_ref_world_size(m->ref);
    *n_out = n;
    if (n > max_n || n == 0) return IVLN_OK;
    int32_t *b = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t *s = (uint8_t *)malloc((size_t)n);
    mapper_ref_world_get(m->ref, xyz, b, s);
    for (int64_t i = 0; i < n; ++i) {
        meta[i] = ((uint32_t)b[i] << 8) | s[i];
        rank[i] = i;
    }
    free(b);
    free(s);
    return IVLN_OK;
}

/* ---------------------------------------------------------------------------------------------------------
 * ivln_gemm_f32 on the host
 * --------------------------------------------------------------------------------------------------------- */
static float a_at(const ivln_gemm_desc *d, const float *A, int m, int k) {
    switch (d->amode) {
        case IVLN_A_MK: return A[(int64_t)m * d->lda + k];
        case IVLN_A_KM: return A[(int64_t)k * d->lda + m];
        default: {  /* IVLN_A_NCHW_P: A[m = channel][k = pixel] of an NCHW gradient tensor */
            int img = k / d->HoWo, pp = k - img * d->HoWo;
            return A[((int64_t)img * d->M + m) * d->HoWo + pp];
        }
    }
}

/* B[k][n]; conv modes gather from the NCHW input with zero padding */
static float b_at(const ivln_gemm_desc *d, int k, int n) {
    const float *B = d->B;
    switch (d->bmode) {
        case IVLN_B_KN: return B[(int64_t)k * d->ldb + n];
        case IVLN_B_NK: return B[(int64_t)n * d->ldb + k];
        case IVLN_B_CONV1X1: {
            int img = n / d->HoWo, pp = n - img * d->HoWo, ho = pp / d->Wout, wo = pp - ho * d->Wout;
            return B[(int64_t)img * d->in_img_stride + (int64_t)k * d->Hin * d->Win + (ho * d->stride) * d->Win + wo * d->stride];
        }
        case IVLN_B_CONV:
        case IVLN_B_CONV_K3:
        case IVLN_B_CONV_K7: {
            int kk = d->K / d->Cin; /* taps per channel */
            int ks = kk == 9 ? 3 : (kk == 49 ? 7 : 0), kh, kw, ci = k / kk, t = k - ci * kk;
            if (d->bmode == IVLN_B_CONV) {
                int32_t pos = d->kpos[k];
                kh = pos >> 16;
                kw = pos & 0xFFFF;
            } else {
                kh = t / ks;
                kw = t - kh * ks;
            }
            int img = n / d->HoWo, pp = n - img * d->HoWo, ho = pp / d->Wout, wo = pp - ho * d->Wout;
            int hi = ho * d->stride - d->pad + kh * d->dil, wi = wo * d->stride - d->pad + kw * d->dil;
            if (hi < 0 || hi >= d->Hin || wi < 0 || wi >= d->Win) return 0.f;
            return B[(int64_t)img * d->in_img_stride + (int64_t)ci * d->Hin * d->Win + hi * d->Win + wi];
        }
        default: return NAN; /* IVLN_B_IM2COL_T / IVLN_B_CONVT: not twinned */
    }
}

int ivln_gemm_f32(const ivln_gemm_desc *desc, void *stream) {
    (void)stream;
    if (!desc || !desc->A || !desc->B || !desc->D || desc->M <= 0 || desc->N <= 0 || desc->K <= 0) return IVLN_E_INVALID;
    ivln_gemm_desc d = *desc;
    if (d.bmode == IVLN_B_IM2COL_T || d.bmode == IVLN_B_CONVT || d.defer_epilogue) return IVLN_E_UNSUPPORTED;
    if (d.HoWo <= 0) d.HoWo = 1;
    if (d.dil <= 0) d.dil = 1;
    if (d.Ctot <= 0) d.Ctot = d.dmode == IVLN_D_NCHW_UP2X4 ? d.M / 4 : d.M;
    if (d.in_img_stride <= 0) d.in_img_stride = (int64_t)d.Cin * d.Hin * d.Win;
    if (d.grp_imgs > 0 && d.a_grp_stride <= 0) d.a_grp_stride = (int64_t)d.M * d.lda;
    for (int n = 0; n < d.N; ++n) {
        int img = n / d.HoWo, pp = n - img * d.HoWo;
        int grp = d.grp_imgs > 0 ? img / d.grp_imgs : 0;
        const float *A = d.A + (int64_t)grp * d.a_grp_stride;
        for (int m = 0; m < d.M; ++m) {
            float acc = 0.f;
            for (int k = 0; k < d.K; ++k) acc = fmaf(a_at(&d, A, m, k), b_at(&d, k, n), acc);
            int64_t addr;
            if (d.dmode == IVLN_D_NCHW) addr = ((int64_t)img * d.Ctot + m) * d.HoWo + pp;
            else if (d.dmode == IVLN_D_NCHW_UP2) {
                int ho = pp / d.Wout, wo = pp - ho * d.Wout;
                addr = (((int64_t)img * d.Ctot + m) * (2 * d.Hout) + 2 * ho + (int)d.sDm) * (2 * d.Wout) + 2 * wo + (int)d.sDn;
            } else if (d.dmode == IVLN_D_NCHW_UP2X4) {
                int cq = d.M / 4, cls = m / cq, ho = pp / d.Wout, wo = pp - ho * d.Wout;
                addr = (((int64_t)img * d.Ctot + (m - cls * cq)) * (2 * d.Hout) + 2 * ho + (cls >> 1)) * (2 * d.Wout) + 2 * wo + (cls & 1);
            } else addr = (int64_t)m * d.sDm + (int64_t)n * d.sDn;
            int me = (d.dmode == IVLN_D_NCHW && d.grp_imgs > 0) ? grp * d.M + m : (d.dmode == IVLN_D_NCHW_UP2X4 ? m % (d.M / 4) : m);
            float v = acc;
            if (d.scale) v = fmaf(v, d.scale[me], d.shift[me]);
            else if (d.shift) v += d.shift[me];
            if (d.residual) v += d.residual[addr];
            if (d.accumulate) v += d.D[addr];
            if (d.relu) v = v > 0.f ? v : 0.f;
            d.D[addr] = v;
        }
    }
    if (d.splits_used) *d.splits_used = 1;
    return IVLN_OK;
}

int ivln_groupnorm_f32(const float *x, const float *gamma, const float *beta, const float *residual, float *y, int N, int C,
                       int HW, int groups, float eps, int relu, int64_t x_img_stride, int64_t x_chan_stride, int splits,
                       int64_t slab_stride, int64_t y_img_stride, int64_t r_img_stride, float *save_mean, float *save_rstd,
                       void *stream) {
    (void)stream;
    if (N <= 0 || C <= 0 || groups <= 0 || C % groups) return IVLN_E_INVALID;
    if (x_chan_stride <= 0) x_chan_stride = HW;
    if (x_img_stride <= 0) x_img_stride = (int64_t)C * HW;
    if (y_img_stride <= 0) y_img_stride = (int64_t)C * HW;
    if (r_img_stride <= 0) r_img_stride = (int64_t)C * HW;
    if (splits < 1) splits = 1;
    const int cpg = C / groups, n = cpg * HW;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int img = 0; img < N; ++img)
        for (int g = 0; g < groups; ++g) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) {
                int cl = i / HW, pp = i - cl * HW;
                const float *p = x + (int64_t)img * x_img_stride + (int64_t)(g * cpg + cl) * x_chan_stride + pp;
                float v = 0.f;
                for (int z = 0; z < splits; ++z) v += p[(int64_t)z * slab_stride];
                tmp[i] = v;
                s += v;
            }
            const float mean = (float)(s / n);
            double q = 0.0;
            for (int i = 0; i < n; ++i) q += (double)(tmp[i] - mean) * (tmp[i] - mean);
            const float rstd = 1.0f / sqrtf((float)(q / n) + eps);
            if (save_mean) {
                save_mean[img * groups + g] = mean;
                save_rstd[img * groups + g] = rstd;
            }
            for (int i = 0; i < n; ++i) {
                int c = g * cpg + i / HW;
                float v = (tmp[i] - mean) * rstd * gamma[c] + beta[c];
                if (residual) v += residual[(int64_t)img * r_img_stride + (int64_t)g * n + i];
                if (relu) v = v > 0.f ? v : 0.f;
                y[(int64_t)img * y_img_stride + (int64_t)g * n + i] = v;
            }
        }
    free(tmp);
    return IVLN_OK;
}
