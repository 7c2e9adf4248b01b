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
 *   ivln_gn_conv_f32   GroupNorm (+ second operand, + residual, + ReLU, + MaxPool) and the next conv(s) as per-group
 *                      partial slabs (no front stage: IVLN_E_UNSUPPORTED)
 *   ivln_nconv_f32     GroupNorm-on-load conv with the output's (count, mean, M2) partials per strip
 *   ivln_kv_linear_f32 Conv1d(k = 1) projection + Flatten-Linear(-ReLU) of one feature map
 *   ivln_cma_step_fwd  the recurrent / attention head of one rollout step from the folded operands
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

int ivln_mapper_step_posed(ivln_mapper *m, const float *depth, const uint8_t *labels, const float *pose,
                           const double *orientation, const uint8_t *not_done, int B, uint8_t *occ_out, uint8_t *sem_out,
                           float *T_out, float *rot_out, void *stream) {
    if (!orientation || !T_out || !rot_out) return IVLN_E_INVALID;
    int rc = ivln_mapper_frames(pose, orientation, B, T_out, rot_out, stream);
    return rc != IVLN_OK ? rc : ivln_mapper_step(m, depth, labels, T_out, pose, rot_out, not_done, B, occ_out, sem_out, stream);
}

/* (the host twin has nothing to overlap: _begin derives the frames, _finish runs the whole step from them) */
int ivln_mapper_step_begin(ivln_mapper *m, const float *depth, const float *pose, const double *orientation,
                           const uint8_t *not_done, int B, uint8_t *occ_out, float *T_out, float *rot_out, void *stream) {
    if (!m || B <= 0 || B > m->B_max || !depth || !pose || !orientation || !T_out || !rot_out || !not_done || !occ_out) return IVLN_E_INVALID;
    return ivln_mapper_frames(pose, orientation, B, T_out, rot_out, stream);
}

int ivln_mapper_step_finish(ivln_mapper *m, const float *depth, const uint8_t *labels, const float *T, const float *pose,
                            const float *rot, const uint8_t *not_done, int B, uint8_t *occ_out, uint8_t *sem_out, void *stream) {
    if (!depth || !labels || !T || !pose || !rot || !not_done || !occ_out || !sem_out) return IVLN_E_INVALID;
    return ivln_mapper_step(m, depth, labels, T, pose, rot, not_done, B, occ_out, sem_out, stream);
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
        case IVLN_B_CONV_K7:
        case IVLN_B_CONV_K2: {
            int kk = d->K / d->Cin; /* taps per channel */
            int ks = kk == 9 ? 3 : (kk == 49 ? 7 : (kk == 4 ? 2 : 0)), kh, kw, ci = k / kk, t = k - ci * kk;
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
    if (d.residual_after_relu) return IVLN_E_UNSUPPORTED; /* (the decoder-skip epilogue form: no twin, callers issue conv + add) */
    if (d.fuse_A_split) return IVLN_E_UNSUPPORTED; /* the fused bottleneck tail reads device-packed split weights: no twin (callers issue the two convs) */
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
                int cls = m & 3, ho = pp / d.Wout, wo = pp - ho * d.Wout;
                addr = (((int64_t)img * d.Ctot + (m >> 2)) * (2 * d.Hout) + 2 * ho + (cls >> 1)) * (2 * d.Wout) + 2 * wo + (cls & 1);
            } else addr = (int64_t)m * d.sDm + (int64_t)n * d.sDn;
            int me = (d.dmode == IVLN_D_NCHW && d.grp_imgs > 0) ? grp * d.M + m : (d.dmode == IVLN_D_NCHW_UP2X4 ? m >> 2 : m);
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

/* =====================================================================================================================
 * Twins of the round-2 entry points.  Plain loops from the semantics stated in include/ivln_hip.h; pinned by
 * tests/test_oracle_twin.py against torch (F.group_norm / F.conv2d / F.max_pool2d / nn.GRUCell arithmetic / softmax)
 * and run against the device library symbol for symbol in tests/test_gpu_twin.py.
 * ===================================================================================================================== */

/* GroupNorm over one (image, group) tile held as `n` floats: two-pass mean / variance, y = (v - mean) * rstd * gamma + beta */
static void gn_tile(const float *v, int n, float eps, float *mean_out, float *rstd_out) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    const float mean = (float)(s / n);
    for (int i = 0; i < n; ++i) q += (double)(v[i] - mean) * (v[i] - mean);
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf((float)(q / n) + eps);
}

/* y[co][oh][ow] += sum_{c in [c0, c1)} sum_taps w[co][c][kh][kw] * in[c][oh*s + kh - pad][ow*s + kw - pad] for ONE image;
 * in: (C, H, W), w: (Cout, C, k, k), y: Cout x Ho*Wo with row stride ldy */
static void conv_image(const float *in, int C, int H, int W, const float *w, int Cout, int k, int s, int pad, int c0, int c1,
                       float *y, int64_t ldy, int Ho, int Wo) {
    for (int co = 0; co < Cout; ++co)
        for (int oh = 0; oh < Ho; ++oh)
            for (int ow = 0; ow < Wo; ++ow) {
                float acc = 0.f;
                for (int c = c0; c < c1; ++c)
                    for (int kh = 0; kh < k; ++kh) {
                        const int ih = oh * s + kh - pad;
                        if (ih < 0 || ih >= H) continue;
                        for (int kw = 0; kw < k; ++kw) {
                            const int iw = ow * s + kw - pad;
                            if (iw < 0 || iw >= W) continue;
                            acc = fmaf(w[(((int64_t)co * C + c) * k + kh) * k + kw], in[((int64_t)c * H + ih) * W + iw], acc);
                        }
                    }
                y[(int64_t)co * ldy + oh * Wo + ow] = acc;
            }
}

int ivln_gn_conv_f32(const ivln_gn_conv_desc *d, void *stream) {
    (void)stream;
    if (!d || d->x0 || !d->x) return IVLN_E_UNSUPPORTED; /* the opt-in front stage has no twin */
    const int N = d->N, C = d->C, H = d->H, W = d->W, G = d->groups;
    if (N <= 0 || C <= 0 || G <= 0 || C % G) return IVLN_E_INVALID;
    const int cpg = C / G, HW = H * W, n = cpg * HW;
    const int64_t M = (int64_t)N * HW;
    const int Hp = d->pool ? (H + 2 - 3) / 2 + 1 : H, Wp = d->pool ? (W + 2 - 3) / 2 + 1 : W;
    float *act = (float *)malloc(sizeof(float) * (size_t)N * C * Hp * Wp);
    float *tile = (float *)malloc(sizeof(float) * (size_t)n), *tile2 = (float *)malloc(sizeof(float) * (size_t)n);
    float *full = (float *)malloc(sizeof(float) * (size_t)C * HW);
    for (int img = 0; img < N; ++img) {
        for (int g = 0; g < G; ++g) {
            for (int op = 0; op < 2; ++op) {
                const float *x = op ? d->x2 : d->x;
                if (!x) continue;
                const int splits = op ? d->splits2 : d->splits;
                const int64_t ss = op ? d->slab_stride2 : d->slab_stride;
                const float *ga = op ? d->gamma2 : d->gamma, *be = op ? d->beta2 : d->beta;
                float *t = op ? tile2 : tile;
                for (int i = 0; i < n; ++i) {
                    const int c = g * cpg + i / HW, p = i % HW;
                    float v = 0.f;
                    for (int z = 0; z < (splits < 1 ? 1 : splits); ++z) v += x[(int64_t)z * ss + (int64_t)c * M + (int64_t)img * HW + p];
                    t[i] = v;
                }
                float mean, rstd;
                gn_tile(t, n, d->eps, &mean, &rstd);
                for (int i = 0; i < n; ++i) {
                    const int c = g * cpg + i / HW;
                    t[i] = (t[i] - mean) * rstd * ga[c] + be[c];
                }
            }
            for (int i = 0; i < n; ++i) {
                const int c = g * cpg + i / HW, p = i % HW;
                float v = tile[i] + (d->x2 ? tile2[i] : 0.f);
                if (d->residual) v += d->residual[((int64_t)img * C + c) * HW + p];
                if (d->relu) v = v > 0.f ? v : 0.f;
                full[(int64_t)c * HW + p] = v;
            }
        }
        for (int c = 0; c < C; ++c)
            for (int oh = 0; oh < Hp; ++oh)
                for (int ow = 0; ow < Wp; ++ow) {
                    float v;
                    if (d->pool) { /* MaxPool2d(3, stride 2, padding 1): padding never wins */
                        v = -INFINITY;
                        for (int kh = 0; kh < 3; ++kh)
                            for (int kw = 0; kw < 3; ++kw) {
                                const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                                const float u = full[(int64_t)c * HW + ih * W + iw];
                                v = u > v ? u : v;
                            }
                    } else {
                        v = full[(int64_t)c * HW + oh * W + ow];
                    }
                    act[(((int64_t)img * C + c) * Hp + oh) * Wp + ow] = v;
                }
    }
    if (d->act_out) memcpy(d->act_out, act, sizeof(float) * (size_t)N * C * Hp * Wp);
    for (int which = 0; which < 2; ++which) {
        const float *w = which ? d->wb : d->wa;
        float *y = which ? d->yb : d->ya;
        if (!w) continue;
        const int Cout = which ? d->Cout_b : d->Cout_a, k = which ? 1 : d->ka, s = which ? d->stride_b : d->stride_a;
        const int pad = which ? 0 : d->pad_a;
        const int Ho = (Hp + 2 * pad - k) / s + 1, Wo = (Wp + 2 * pad - k) / s + 1;
        const int64_t Mo = (int64_t)N * Ho * Wo;
        for (int g = 0; g < G; ++g) /* the convolution split over K by GroupNorm group: partial slab g */
            for (int img = 0; img < N; ++img)
                conv_image(act + (int64_t)img * C * Hp * Wp, C, Hp, Wp, w, Cout, k, s, pad, g * cpg, (g + 1) * cpg,
                           y + (int64_t)g * Cout * Mo + (int64_t)img * Ho * Wo, Mo, Ho, Wo);
    }
    free(act);
    free(tile);
    free(tile2);
    free(full);
    return IVLN_OK;
}

/* merge the (count, mean, M2) partials of one (image, group) in part order */
static void merge_parts(const float *stats, int parts, int N, int groups, int img, int g, float eps, float *mean_out, float *rstd_out) {
    double cnt = 0.0, sum = 0.0;
    for (int p = 0; p < parts; ++p) {
        const float *s = stats + (((int64_t)p * N + img) * groups + g) * 3;
        cnt += s[0];
        sum += (double)s[0] * s[1];
    }
    const double mean = sum / cnt;
    double m2 = 0.0;
    for (int p = 0; p < parts; ++p) {
        const float *s = stats + (((int64_t)p * N + img) * groups + g) * 3;
        m2 += s[2] + (double)s[0] * (s[1] - mean) * (s[1] - mean);
    }
    *mean_out = (float)mean;
    *rstd_out = 1.0f / sqrtf((float)(m2 / cnt) + eps);
}

int ivln_nconv_f32(const ivln_nconv_desc *d, void *stream) {
    (void)stream;
    if (!d || !d->x || !d->wa || !d->ya) return IVLN_E_INVALID;
    const int N = d->N, C = d->C, H = d->H, W = d->W, G = d->groups, HW = H * W;
    const int sa = d->stride_a > 0 ? d->stride_a : 1, sb = d->stride_b > 0 ? d->stride_b : 1;
    if ((d->ka != 1 && d->ka != 3) || sa > 2 || sb > 2) return IVLN_E_UNSUPPORTED;
    if ((d->act_out || d->wb) && sa != 1) return IVLN_E_UNSUPPORTED;
    float *in = (float *)malloc(sizeof(float) * (size_t)N * C * HW);
    for (int img = 0; img < N; ++img)
        for (int c = 0; c < C; ++c) {
            float m1 = 0.f, r1 = 1.f, m2 = 0.f, r2 = 1.f;
            if (d->stats) {
                const int g = c / (C / G);
                merge_parts(d->stats, d->parts, N, G, img, g, d->eps, &m1, &r1);
                if (d->x2) merge_parts(d->stats2, d->parts2, N, G, img, g, d->eps, &m2, &r2);
            }
            for (int p = 0; p < HW; ++p) {
                float v;
                if (d->stats) {
                    v = (d->x[((int64_t)c * N + img) * HW + p] - m1) * r1 * d->gamma[c] + d->beta[c];
                    if (d->x2) v += (d->x2[((int64_t)c * N + img) * HW + p] - m2) * r2 * d->gamma2[c] + d->beta2[c];
                } else {
                    v = d->x[((int64_t)img * C + c) * HW + p];
                }
                if (d->residual) v += d->residual[((int64_t)img * C + c) * HW + p];
                if (d->relu) v = v > 0.f ? v : 0.f;
                in[((int64_t)img * C + c) * HW + p] = v;
            }
        }
    if (d->act_out) memcpy(d->act_out, in, sizeof(float) * (size_t)N * C * HW);
    const int ph = d->ka / 2;
    const int Ho_a = (H + 2 * ph - d->ka) / sa + 1, Wo_a = (W + 2 * ph - d->ka) / sa + 1;
    int RS = d->rows_per_block > 0 ? d->rows_per_block : (Wo_a >= 64 ? 1 : 64 / Wo_a);
    if (RS > Ho_a) RS = Ho_a;
    if (d->wb && (RS % sb)) RS = (RS + sb - 1) / sb * sb;
    const int strips = (Ho_a + RS - 1) / RS;
    for (int which = 0; which < 2; ++which) {
        const float *w = which ? d->wb : d->wa;
        if (!w) continue;
        float *y = which ? d->yb : d->ya, *st = which ? d->stats_b : d->stats_a;
        const int Cout = which ? d->Cout_b : d->Cout_a, k = which ? 1 : d->ka, s = which ? sb : sa, pad = which ? 0 : ph;
        const int go = which ? d->groups_b : d->groups_a;
        const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
        const int rs = which ? RS / sb : RS; /* conv B's rows of a strip start on its stride */
        float *tmp = (float *)malloc(sizeof(float) * (size_t)Cout * Ho * Wo);
        for (int img = 0; img < N; ++img) {
            conv_image(in + (int64_t)img * C * HW, C, H, W, w, Cout, k, s, pad, 0, C, tmp, (int64_t)Ho * Wo, Ho, Wo);
            for (int co = 0; co < Cout; ++co)
                memcpy(y + ((int64_t)co * N + img) * Ho * Wo, tmp + (int64_t)co * Ho * Wo, sizeof(float) * (size_t)Ho * Wo);
            if (!st) continue;
            const int cpo = Cout / go;
            for (int sp = 0; sp < strips; ++sp) {
                const int r0 = sp * rs, r1 = (r0 + rs < Ho) ? r0 + rs : Ho;
                for (int g = 0; g < go; ++g) {
                    double sum = 0.0, q = 0.0;
                    int64_t cnt = 0;
                    for (int co = g * cpo; co < (g + 1) * cpo; ++co)
                        for (int r = r0; r < r1; ++r)
                            for (int x = 0; x < Wo; ++x) {
                                sum += tmp[((int64_t)co * Ho + r) * Wo + x];
                                ++cnt;
                            }
                    const double mean = cnt ? sum / (double)cnt : 0.0;
                    for (int co = g * cpo; co < (g + 1) * cpo; ++co)
                        for (int r = r0; r < r1; ++r)
                            for (int x = 0; x < Wo; ++x) {
                                const double dv = tmp[((int64_t)co * Ho + r) * Wo + x] - mean;
                                q += dv * dv;
                            }
                    float *o = st + (((int64_t)sp * N + img) * go + g) * 3;
                    o[0] = (float)cnt, o[1] = (float)mean, o[2] = (float)q;
                }
            }
        }
        free(tmp);
    }
    free(in);
    return IVLN_OK;
}

int ivln_kv_linear_f32(const float *feat, int rows, int C, int P, const float *w_kv, const float *b_kv, int Ckv, float *kv,
                       const float *w_lin, const float *b_lin, int O, int relu, float *lin, int64_t ld_lin, void *stream) {
    (void)stream;
    if (!feat || rows <= 0 || C <= 0 || P <= 0) return IVLN_E_INVALID;
    for (int r = 0; r < rows; ++r) {
        const float *f = feat + (int64_t)r * C * P;
        if (w_kv && kv)
            for (int o = 0; o < Ckv; ++o)
                for (int p = 0; p < P; ++p) {
                    float acc = 0.f;
                    for (int c = 0; c < C; ++c) acc = fmaf(w_kv[(int64_t)o * C + c], f[(int64_t)c * P + p], acc);
                    kv[((int64_t)r * Ckv + o) * P + p] = acc + (b_kv ? b_kv[o] : 0.f);
                }
        if (w_lin && lin)
            for (int o = 0; o < O; ++o) {
                float acc = 0.f;
                for (int i = 0; i < C * P; ++i) acc = fmaf(w_lin[(int64_t)o * C * P + i], f[i], acc);
                acc += b_lin ? b_lin[o] : 0.f;
                lin[(int64_t)r * ld_lin + o] = (relu && acc < 0.f) ? 0.f : acc;
            }
    }
    return IVLN_OK;
}

/* one masked GRU step of one row (PyTorch gate order r, z, n) */
static void gru_row(const float *x, int I, const float *h_in, float mk, const float *w_ih, const float *w_hh, const float *b_ih,
                    const float *b_hh, int H, float *h_out) {
    for (int j = 0; j < H; ++j) {
        float gi[3], gh[3];
        for (int g = 0; g < 3; ++g) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < I; ++k) a = fmaf(w_ih[((int64_t)g * H + j) * I + k], x[k], a);
            for (int k = 0; k < H; ++k) b = fmaf(w_hh[((int64_t)g * H + j) * H + k], h_in[k] * mk, b);
            gi[g] = a + b_ih[g * H + j];
            gh[g] = b + b_hh[g * H + j];
        }
        const float r = 1.f / (1.f + expf(-(gi[0] + gh[0]))), z = 1.f / (1.f + expf(-(gi[1] + gh[1])));
        const float nn = tanhf(gi[2] + r * gh[2]);
        h_out[j] = (1.f - z) * nn + z * (h_in[j] * mk);
    }
}

/* softmax(logits * scale) . v over n positions (map_cma_policy.py:266-274); v: (Cv, n) channel-major */
static void attend(const float *logits, int n, float scale, const float *v, int Cv, float *out, float *attn) {
    float mx = -INFINITY, den = 0.f;
    for (int i = 0; i < n; ++i) mx = logits[i] * scale > mx ? logits[i] * scale : mx;
    for (int i = 0; i < n; ++i) {
        attn[i] = expf(logits[i] * scale - mx);
        den += attn[i];
    }
    for (int i = 0; i < n; ++i) attn[i] /= den;
    for (int c = 0; c < Cv; ++c) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc = fmaf(attn[i], v[(int64_t)c * n + i], acc);
        out[c] = acc;
    }
}

int64_t ivln_cma_step_ws_floats(int rows, int L, int P, int H) {
    (void)rows, (void)L, (void)P, (void)H;
    return 0; /* the host loops need no scratch */
}

int ivln_cma_step_fwd(const ivln_cma_step_desc *d, int mode, void *stream) {
    (void)mode, (void)stream;
    if (!d || d->rows <= 0) return IVLN_E_INVALID;
    const int H = d->H, Hq = d->Hq, L = d->L, P = d->P, Ct = d->Ct, I1 = d->d_out + d->m_out + d->E;
    const int o_txt = H, o_dep = H + Ct, o_map = H + Ct + d->d_out;
    float *logits = (float *)malloc(sizeof(float) * (size_t)(L > P ? L : P)), *attn = (float *)malloc(sizeof(float) * (size_t)(L > P ? L : P));
    float *q2 = (float *)malloc(sizeof(float) * (size_t)Hq), *c2 = (float *)malloc(sizeof(float) * (size_t)H);
    for (int r = 0; r < d->rows; ++r) {
        const float mk = d->mask[r] ? 1.f : 0.f;
        float *x2 = d->x2 + (int64_t)r * d->x2w;
        const float *h1 = d->h_in + (int64_t)r * d->ld_h, *h2 = h1 + H;
        float *ho = d->h_out + (int64_t)r * d->ld_ho;
        /* GRU-1 -> state (also x2[0:H]) */
        gru_row(d->state_in + (int64_t)r * I1, I1, h1, mk, d->w_ih1, d->w_hh1, d->b_ih1, d->b_hh1, H, ho);
        memcpy(x2, ho, sizeof(float) * (size_t)H);
        /* text attention: logits straight from the state through the folded Mq; PAD positions pushed down by 1e8 */
        const float *Mq = d->Mq + (int64_t)r * d->Mq_img;
        for (int i = 0; i < L; ++i) {
            float a = Mq[(int64_t)H * L + i];
            for (int h = 0; h < H; ++h) a = fmaf(ho[h], Mq[(int64_t)h * L + i], a);
            logits[i] = i >= d->lengths[r] ? a - 1e8f : a;
        }
        attend(logits, L, d->scale, d->txt + (int64_t)r * Ct * L, Ct, x2 + o_txt, attn);
        /* the query of the two short attentions through the folded TQb: q2 = sum_i a_i TQb[:, i] */
        const float *TQb = d->TQb + (int64_t)r * d->TQb_img;
        for (int c = 0; c < Hq; ++c) {
            float a = 0.f;
            for (int i = 0; i < L; ++i) a = fmaf(attn[i], TQb[(int64_t)c * L + i], a);
            q2[c] = a;
        }
        for (int which = 0; which < 2; ++which) {
            const int Cv = which ? d->m_out : d->d_out;
            const float *kv = (which ? d->mkv : d->dkv) + (int64_t)r * (Hq + Cv) * P;
            for (int p = 0; p < P; ++p) {
                float a = 0.f;
                for (int c = 0; c < Hq; ++c) a = fmaf(q2[c], kv[(int64_t)c * P + p], a);
                logits[p] = a;
            }
            attend(logits, P, d->scale, kv + (int64_t)Hq * P, Cv, x2 + (which ? o_map : o_dep), attn);
        }
        /* second_state_compress + GRU-2 (x2's prev-action slice is the caller's) */
        for (int j = 0; j < H; ++j) {
            float a = 0.f;
            for (int k = 0; k < d->x2w; ++k) a = fmaf(d->w_c[(int64_t)j * d->x2w + k], x2[k], a);
            a += d->b_c[j];
            c2[j] = a > 0.f ? a : 0.f;
        }
        gru_row(c2, H, h2, mk, d->w_ih2, d->w_hh2, d->b_ih2, d->b_hh2, H, ho + H);
        memcpy(d->feats + (int64_t)r * H, ho + H, sizeof(float) * (size_t)H);
    }
    free(logits);
    free(attn);
    free(q2);
    free(c2);
    return IVLN_OK;
}
