/*
 * TEST INFRASTRUCTURE ONLY - plain-C, single-threaded restatement of the reference's egocentric
 * semantic mapper.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product path (ivln-ce_amd/) never does.
 *
 * Follows, function by function (paths relative to /root/reference):
 *   frames        ivlnce_baselines/common/mapping_module/projector/core.py:6-37 (_transform3D),
 *                 mapper.py:38-48 (rotate_around_y_matrix), mapper.py:132-138 (elevation + pi)
 *   unproject     core.py:70-115 (intrinsics, x_scale/y_scale), core.py:117-147 (point_cloud),
 *                 core.py:149-171 (bmm == fmaf chain k=0..3, verified bit-exact vs goldens),
 *                 mapper.py:381-384 (depth * 10)
 *   filters       mapper.py:236-253, 415-424 (0.01<d<0.99 ; h-1.0<y<h+0.5)
 *   keep_highest  mapper.py:428-474 incl. the colliding hash of :468-469 (quirk Q1) and
 *                 torch_scatter.scatter_max first-max-wins (quirk Q4, parity unpinned for ties)
 *   world cloud   mapper.py:297-333, 825-848
 *   raster        mapper.py:884-901 (h-1.25<y<h+0.75), :255-266 (shift origin; un-fused fp32
 *                 (a*x + b*y) + c*z), :101-114, 513-531 (discretise), :569-571 (zero-fill then
 *                 last-writer-wins store, quirk Q3), :611 (drop label 0 for the semantic map)
 *
 * Pinned by tests/test_oracle_mapper.py against tests/golden/mapper_*.npz, which were produced by
 * running the reference's own MappingModule (tests/golden/gen_mapper_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  -ffp-contract=off matters:
 * every multiply/add below must round exactly where written.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float x, y, z;
    int32_t b;
    uint8_t sem;
} Pt;

typedef struct {
    int H, W, rows, cols;
    float res, half_h, half_w; /* map resolution, height_meters/2, width_meters/2 */
    float *xs, *ys;            /* x_scale[W], y_scale[H] */
    Pt *world;
    int64_t nworld, cap;
    Pt *tmp;
    int64_t tmpcap;
} MapperRef;

static void ensure(Pt **p, int64_t *cap, int64_t n) {
    if (n > *cap) {
        int64_t c = *cap ? *cap : 1024;
        while (c < n) c *= 2;
        *p = (Pt *)realloc(*p, (size_t)c * sizeof(Pt));
        *cap = c;
    }
}

MapperRef *mapper_ref_create(int H, int W, double vfov_rad, double height_m, double width_m, double res_m) {
    MapperRef *m = (MapperRef *)calloc(1, sizeof(MapperRef));
    m->H = H;
    m->W = W;
    m->rows = (int)ceil(height_m / res_m);
    m->cols = (int)ceil(width_m / res_m);
    m->res = (float)res_m;
    m->half_h = (float)(height_m / 2);
    m->half_w = (float)(width_m / 2);
    /* core.py:70-77: python doubles, then torch.Tensor() rounds to fp32 */
    double hfov = (double)W / (double)H * vfov_rad;
    float fx = (float)((double)W / (2.0 * tan(hfov / 2.0)));
    float fy = (float)((double)H / (2.0 * tan(vfov_rad / 2.0)));
    float cx = (float)(W / 2.0), cy = (float)(H / 2.0);
    m->xs = (float *)malloc(sizeof(float) * W);
    m->ys = (float *)malloc(sizeof(float) * H);
    for (int u = 0; u < W; ++u) { /* core.py:105 */
        float t = (float)u + 0.5f;
        t = t - cx;
        m->xs[u] = t / fx;
    }
    for (int v = 0; v < H; ++v) {
        float t = (float)v + 0.5f;
        t = t - cy;
        m->ys[v] = t / fy;
    }
    return m;
}

void mapper_ref_destroy(MapperRef *m) {
    if (!m) return;
    free(m->xs);
    free(m->ys);
    free(m->world);
    free(m->tmp);
    free(m);
}

void mapper_ref_reset(MapperRef *m) { m->nworld = 0; }

/* core.py:6-37 + mapper.py:38-48: trig in fp64 (orientation is float64, quirk Q9), rounded into
 * fp32 matrices.  T: B x 16 row-major, rot: B x 9 row-major for angle = -heading. */
void mapper_ref_frames(int B, const float *pose, const double *orient, float *T, float *rot) {
    for (int b = 0; b < B; ++b) {
        double elev = orient[2 * b + 0] + 3.141592653589793; /* torch.pi */
        double head = orient[2 * b + 1];
        double cx = cos(elev), sx = sin(elev), cy = cos(head), sy = sin(head);
        float *t = T + 16 * b;
        t[0] = (float)cy;  t[1] = (float)(sx * sy);  t[2] = (float)(cx * sy);  t[3] = pose[3 * b + 0];
        t[4] = 0.f;        t[5] = (float)cx;         t[6] = (float)(-sx);      t[7] = pose[3 * b + 1];
        t[8] = (float)(-sy); t[9] = (float)(cy * sx); t[10] = (float)(cy * cx); t[11] = pose[3 * b + 2];
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
        double a = -head;
        float *r = rot + 9 * b;
        r[0] = (float)cos(a); r[1] = 0.f; r[2] = (float)sin(a);
        r[3] = 0.f;           r[4] = 1.f; r[5] = 0.f;
        r[6] = (float)(-sin(a)); r[7] = 0.f; r[8] = (float)cos(a);
    }
}

typedef struct {
    int64_t key;
    int64_t idx;
} KI;

static int cmp_ki(const void *a, const void *b) {
    const KI *x = (const KI *)a, *y = (const KI *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* mapper.py:428-474.  In place; returns new count; output ordered by ascending key. */
static int64_t keep_highest(Pt *p, int64_t n, float half_res, Pt **scratch, int64_t *scap) {
    if (n <= 0) return n;
    int64_t *r = (int64_t *)malloc(sizeof(int64_t) * n), *c = (int64_t *)malloc(sizeof(int64_t) * n);
    int64_t rmin = INT64_MAX, cmin = INT64_MAX;
    for (int64_t i = 0; i < n; ++i) {
        r[i] = (int64_t)rintf(p[i].z / half_res); /* :464, round-half-even */
        c[i] = (int64_t)rintf(p[i].x / half_res);
        if (r[i] < rmin) rmin = r[i];
        if (c[i] < cmin) cmin = c[i];
    }
    int64_t R = 0, C = 0;
    for (int64_t i = 0; i < n; ++i) {
        r[i] -= rmin;
        c[i] -= cmin;
        if (r[i] > R) R = r[i];
        if (c[i] > C) C = c[i];
    }
    KI *ki = (KI *)malloc(sizeof(KI) * n);
    for (int64_t i = 0; i < n; ++i) {
        ki[i].key = (int64_t)p[i].b * (R * C) + r[i] * C + c[i]; /* :469, max not max+1 (Q1) */
        ki[i].idx = i;
    }
    qsort(ki, (size_t)n, sizeof(KI), cmp_ki);
    ensure(scratch, scap, n);
    Pt *out = *scratch;
    int64_t m = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i, best = ki[i].idx;
        for (; j < n && ki[j].key == ki[i].key; ++j)
            if (p[ki[j].idx].y > p[best].y) best = ki[j].idx; /* strict >: first max wins (Q4) */
        out[m++] = p[best];
        i = j;
    }
    memcpy(p, out, sizeof(Pt) * (size_t)m);
    free(r);
    free(c);
    free(ki);
    return m;
}

/* clear_completed_episode_data, mapper.py:310-326 */
void mapper_ref_clear_done(MapperRef *m, int B, const uint8_t *not_done) {
    int64_t k = 0;
    for (int64_t i = 0; i < m->nworld; ++i) {
        int b = m->world[i].b;
        if (b >= B) continue;          /* paused envs, :315-318 */
        if (not_done[b] == 0) continue; /* finished episodes, :320-326 */
        m->world[k++] = m->world[i];
    }
    m->nworld = k;
}

int mapper_ref_raster(MapperRef *m, int B, const float *pose, const float *rot, uint8_t *occ, uint8_t *sem);

/* One MappingModule.forward (mapper.py:921-944).  depth: B*H*W fp32 in [0,1]; labels: B*H*W u8;
 * T: B*16; pose: B*3; rot: B*9 (rotation for -heading); not_done: B u8.  occ/sem: B*rows*cols. */
int mapper_ref_step(MapperRef *m, int B, const float *depth, const uint8_t *labels, const float *T,
                    const float *pose, const float *rot, const uint8_t *not_done, uint8_t *occ,
                    uint8_t *sem) {
    const int H = m->H, W = m->W;
    mapper_ref_clear_done(m, B, not_done);
    /* --- GenerateSemanticPointCloud, mapper.py:398-425 --- */
    int64_t np_max = (int64_t)B * H * W;
    Pt *loc = (Pt *)malloc(sizeof(Pt) * (size_t)np_max);
    int64_t nl = 0;
    for (int b = 0; b < B; ++b) {
        const float *t = T + 16 * b;
        float h = pose[3 * b + 1];
        float hlo = h - 1.0f, hhi = h + 0.5f;
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                int64_t pix = ((int64_t)b * H + v) * W + u;
                float d = depth[pix];
                if (!(d > 0.01f && d < 0.99f)) continue;
                float z = d * 10.0f;
                float x = z * m->xs[u];
                float y = z * m->ys[v];
                float w[3];
                for (int r = 0; r < 3; ++r) { /* core.py:171 bmm == fma chain over k */
                    float acc = t[4 * r + 0] * x;
                    acc = fmaf(t[4 * r + 1], y, acc);
                    acc = fmaf(t[4 * r + 2], z, acc);
                    acc = fmaf(t[4 * r + 3], 1.0f, acc);
                    w[r] = acc - 0.0f; /* world_shift_origin = 0, core.py:214 */
                }
                if (!(w[1] > hlo && w[1] < hhi)) continue;
                Pt q = {w[0], w[1], w[2], b, labels[pix]};
                loc[nl++] = q;
            }
    }
    float half_res = (float)((double)m->res / 2); /* python: map_resolution_meters / 2 */
    nl = keep_highest(loc, nl, half_res, &m->tmp, &m->tmpcap);
    /* --- concatenate + keep_highest on world, mapper.py:844-847 --- */
    ensure(&m->world, &m->cap, m->nworld + nl);
    memcpy(m->world + m->nworld, loc, sizeof(Pt) * (size_t)nl);
    m->nworld += nl;
    free(loc);
    m->nworld = keep_highest(m->world, m->nworld, half_res, &m->tmp, &m->tmpcap);
    return mapper_ref_raster(m, B, pose, rot, occ, sem);
}

/* FilterPointCloudByRobotHeight + DenseMap.update, mapper.py:884-901, 555-571 */
int mapper_ref_raster(MapperRef *m, int B, const float *pose, const float *rot, uint8_t *occ, uint8_t *sem) {
    size_t cells = (size_t)B * m->rows * m->cols;
    memset(occ, 0, cells);
    memset(sem, 0, cells);
    for (int64_t i = 0; i < m->nworld; ++i) {
        Pt q = m->world[i];
        int b = q.b;
        float h = pose[3 * b + 1];
        if (!(q.y > h - 1.25f && q.y < h + 0.75f)) continue;
        float x = q.x + (-pose[3 * b + 0]);
        float y = q.y + (-pose[3 * b + 1]);
        float z = q.z + (-pose[3 * b + 2]);
        const float *r = rot + 9 * b;
        float xr = (r[0] * x + r[1] * y) + r[2] * z; /* mapper.py:261, un-fused */
        float zr = (r[6] * x + r[7] * y) + r[8] * z;
        int64_t row = (int64_t)rintf((zr + m->half_h) / m->res);
        int64_t col = (int64_t)rintf((xr + m->half_w) / m->res);
        if (row < 0 || row >= m->rows || col < 0 || col >= m->cols) continue;
        size_t o = ((size_t)b * m->rows + (size_t)row) * m->cols + (size_t)col;
        occ[o] = 1;
        if (q.sem != 0) sem[o] = q.sem; /* :611 + last writer wins in cloud order (Q3) */
    }
    return 0;
}

/* Known-map mode (mapper.py:283-294, 851-881): append a pre-built cloud for env b. */
int mapper_ref_load_known(MapperRef *m, int b, const float *xyz, const uint8_t *semv, int64_t n) {
    ensure(&m->world, &m->cap, m->nworld + n);
    for (int64_t i = 0; i < n; ++i) {
        Pt q = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], b, semv[i]};
        m->world[m->nworld++] = q;
    }
    return 0;
}

/* Known-map step = mapper_ref_clear_done, then mapper_ref_load_known for each finished env
 * (mapper.py:871-879), then mapper_ref_raster. */

int64_t mapper_ref_world_size(const MapperRef *m) { return m->nworld; }

void mapper_ref_world_get(const MapperRef *m, float *xyz, int32_t *b, uint8_t *semv) {
    for (int64_t i = 0; i < m->nworld; ++i) {
        xyz[3 * i] = m->world[i].x;
        xyz[3 * i + 1] = m->world[i].y;
        xyz[3 * i + 2] = m->world[i].z;
        b[i] = m->world[i].b;
        semv[i] = m->world[i].sem;
    }
}
