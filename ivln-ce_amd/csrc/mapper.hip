// Egocentric semantic mapper for gfx950 - bit-exact with the reference's MappingModule
// (ivlnce_baselines/common/mapping_module/mapper.py:904-944) on identical depth/label inputs.
//
// The reference runs ~150 tiny torch ops per step, two torch_scatter.scatter_max calls, sorts the
// cloud by a (colliding) cell hash and relies on last-writer-wins index_put_.  Here one step is 7
// launches and the cloud is an UNORDERED bag: every decision the reference takes by position in its
// sorted cloud is taken by a 64-bit `rank` carried with each point (= its key in the last
// keep-highest), so append order does not matter and no sort is needed:
//   keep-highest arg-max  -> atomicMax of (orderable height, ~index) in a dense per-key table
//   first-max-wins ties   -> min rank among points that attain the max height
//   last-writer-wins map  -> atomicMax of (rank<<8 | label) per 10 cm map cell
// Float recipe (verified against goldens through oracle/mapper_ref.c): camera->world is an fmaf
// chain over k=0..3, the ego rotation is un-fused (a*x + b*y) + c*z, divisions are IEEE, rounding is
// half-to-even.  Compile with -ffp-contract=off; every op below is an explicit __f*_rn.
//
// HBM traffic per step is ~1 MB/env (launch-latency bound, not bandwidth bound) - see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <new>
#include "../../include/ivln_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr uint64_t kLocalRankBit = 1ull << 62;

struct Scalars {       // device-side scalars of one mapper (reset by k_finalize for the next step)
    int mmL[4];        // local cloud: rmin, cmin, rmax, cmax
    int mmW[4];        // world cloud
    unsigned cnt_src;  // points in the source world buffer
    unsigned cnt_old;  // snapshot of cnt_src before this step's appends
    unsigned cnt_dst;  // points appended to the destination world buffer
    int err;           // sticky IVLN_E_*
};

struct Pt {  // 16-byte world point
    float x, y, z;
    uint32_t meta;  // batch << 8 | label
};

__device__ __forceinline__ uint32_t ord_f32(float h) {
    h = __fadd_rn(h, 0.0f);  // -0 -> +0 so that -0 == +0 ties like the reference's `>` compare
    uint32_t u = __float_as_uint(h);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int cell_index(float v, float half_res) {
    return (int)rintf(__fdiv_rn(v, half_res));  // mapper.py:464 (v / (res/2)).round().long()
}

// Global min/max of the cell indices without atomics: every block writes ONE partial
// {rmin,cmin,rmax,cmax} (wave shuffles + LDS), and the next kernel's blocks each fold the (few hundred
// to ~1300) partials at start-up.  (The first version issued 4 same-address atomics per wave:
// ~16k serialised L2 atomics = 100-146 us for what is a ~5 us kernel.)
__device__ __forceinline__ void block_minmax_store(bool valid, int r, int c, int* __restrict__ dst4) {
    __shared__ int sh[4][4];
    int rmin = valid ? r : INT32_MAX, cmin = valid ? c : INT32_MAX;
    int rmax = valid ? r : INT32_MIN, cmax = valid ? c : INT32_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, o));
        cmin = min(cmin, __shfl_xor(cmin, o));
        rmax = max(rmax, __shfl_xor(rmax, o));
        cmax = max(cmax, __shfl_xor(cmax, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[w][0] = rmin; sh[w][1] = cmin; sh[w][2] = rmax; sh[w][3] = cmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dst4[0] = min(min(sh[0][0], sh[1][0]), min(sh[2][0], sh[3][0]));
        dst4[1] = min(min(sh[0][1], sh[1][1]), min(sh[2][1], sh[3][1]));
        dst4[2] = max(max(sh[0][2], sh[1][2]), max(sh[2][2], sh[3][2]));
        dst4[3] = max(max(sh[0][3], sh[1][3]), max(sh[2][3], sh[3][3]));
    }
}

// Fold n partials into mm4 (shared memory of the calling block); block 0 also publishes them.
__device__ __forceinline__ void block_minmax_fold(const int* __restrict__ partials, int n, int* mm4,
                                                  int* __restrict__ publish) {
    __shared__ int sh[4][4];
    int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int4 v = *reinterpret_cast<const int4*>(partials + 4 * i);
        rmin = min(rmin, v.x); cmin = min(cmin, v.y); rmax = max(rmax, v.z); cmax = max(cmax, v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, o));
        cmin = min(cmin, __shfl_xor(cmin, o));
        rmax = max(rmax, __shfl_xor(rmax, o));
        cmax = max(cmax, __shfl_xor(cmax, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[w][0] = rmin; sh[w][1] = cmin; sh[w][2] = rmax; sh[w][3] = cmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm4[0] = min(min(sh[0][0], sh[1][0]), min(sh[2][0], sh[3][0]));
        mm4[1] = min(min(sh[0][1], sh[1][1]), min(sh[2][1], sh[3][1]));
        mm4[2] = max(max(sh[0][2], sh[1][2]), max(sh[2][2], sh[3][2]));
        mm4[3] = max(max(sh[0][3], sh[1][3]), max(sh[2][3], sh[3][3]));
        if (publish && blockIdx.x == 0) {
            publish[0] = mm4[0]; publish[1] = mm4[1]; publish[2] = mm4[2]; publish[3] = mm4[3];
        }
    }
    __syncthreads();
}

// Block-aggregated append: ONE atomicAdd per block call (wave ballots + a 4-entry LDS prefix);
// returns the slot for threads with pred, 0xFFFFFFFF otherwise / when full.  Every thread of the block
// must call it.  (Per-wave atomics on the single counter serialised ~4k same-address atomics in
// k_world_select: 59 us.)
__device__ __forceinline__ unsigned wave_append(bool pred, unsigned* counter, unsigned capacity, int* err) {
    __shared__ unsigned wcnt[4];
    __shared__ unsigned bbase;
    const unsigned long long mask = __ballot(pred);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wcnt[w] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        bbase = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    unsigned off = bbase;
    for (int i = 0; i < w; ++i) off += wcnt[i];
    const unsigned slot = off + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // wcnt / bbase are reused by the next call
    if (!pred) return 0xFFFFFFFFu;
    if (slot >= capacity) {
        *err = IVLN_E_CAPACITY;
        return 0xFFFFFFFFu;
    }
    return slot;
}

__device__ __forceinline__ int64_t make_key(int b, int r, int c, const int* mm) {
    int64_t R = (int64_t)mm[2] - mm[0], C = (int64_t)mm[3] - mm[1];
    return (int64_t)b * (R * C) + (int64_t)(r - mm[0]) * C + (int64_t)(c - mm[1]);  // mapper.py:469
}

// ---- A: unproject + filter local pixels, local min/max; old world min/max; zero occupancy ----
__global__ __launch_bounds__(kThreads) void k_local_unproject(
    const float* __restrict__ depth, const float* __restrict__ T, const float* __restrict__ pose,
    const uint8_t* __restrict__ not_done, const float* __restrict__ xs, const float* __restrict__ ys,
    int B, int H, int W, float half_res, float4* __restrict__ rec, const Pt* __restrict__ wsrc,
    Scalars* sc, uint8_t* __restrict__ occ, int map_cells, int pix_blocks, int* __restrict__ bmmL,
    int* __restrict__ bmmW) {
    if ((int)blockIdx.x < pix_blocks) {
        int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
        int64_t total = (int64_t)B * H * W;
        bool valid = false;
        int r = 0, c = 0;
        if (pix < total) {
            int u = (int)(pix % W);
            int v = (int)((pix / W) % H);
            int b = (int)(pix / ((int64_t)W * H));
            float d = depth[pix];
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d > 0.01f && d < 0.99f) {  // mapper.py:416-418
                float z = __fmul_rn(d, 10.0f);  // mapper.py:381-384
                float x = __fmul_rn(z, xs[u]);  // core.py:137-139
                float y = __fmul_rn(z, ys[v]);
                const float* t = T + 16 * b;
                float w[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {  // core.py:171 bmm == fma chain over k
                    float acc = __fmul_rn(t[4 * k + 0], x);
                    acc = __fmaf_rn(t[4 * k + 1], y, acc);
                    acc = __fmaf_rn(t[4 * k + 2], z, acc);
                    acc = __fmaf_rn(t[4 * k + 3], 1.0f, acc);
                    w[k] = __fsub_rn(acc, 0.0f);  // world_shift_origin == 0 (core.py:214)
                }
                float h = pose[3 * b + 1];
                if (w[1] > __fsub_rn(h, 1.0f) && w[1] < __fadd_rn(h, 0.5f)) {  // mapper.py:420-424
                    valid = true;
                    out = make_float4(w[0], w[1], w[2], 1.0f);
                    r = cell_index(w[2], half_res);
                    c = cell_index(w[0], half_res);
                }
            }
            rec[pix] = out;
        }
        block_minmax_store(valid, r, c, bmmL + 4 * blockIdx.x);
        // zero the occupancy output (DenseMap.update_map fill_(0), mapper.py:570)
        for (int64_t i = pix; i < map_cells; i += (int64_t)pix_blocks * kThreads) occ[i] = 0;
    } else {
        // old world points that survive clear_completed_episode_data (mapper.py:310-326)
        unsigned n = sc->cnt_src;
        int nb = gridDim.x - pix_blocks;
        int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
        for (unsigned i = ((unsigned)blockIdx.x - pix_blocks) * kThreads + threadIdx.x; i < n; i += nb * kThreads) {
            Pt p = wsrc[i];
            int b = (int)(p.meta >> 8);
            if (b < B && not_done[b] != 0) {
                int r = cell_index(p.z, half_res), c = cell_index(p.x, half_res);
                rmin = min(rmin, r); cmin = min(cmin, c); rmax = max(rmax, r); cmax = max(cmax, c);
            }
        }
        // encode the per-thread box as two "points" for the block reduction
        const bool any = rmin != INT32_MAX;
        __shared__ int shw[4][4];
        int v0 = rmin, v1 = cmin, v2 = rmax, v3 = cmax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v0 = min(v0, __shfl_xor(v0, o));
            v1 = min(v1, __shfl_xor(v1, o));
            v2 = max(v2, __shfl_xor(v2, o));
            v3 = max(v3, __shfl_xor(v3, o));
        }
        (void)any;
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            shw[w][0] = v0; shw[w][1] = v1; shw[w][2] = v2; shw[w][3] = v3;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int* dst = bmmW + 4 * ((int)blockIdx.x - pix_blocks);
            dst[0] = min(min(shw[0][0], shw[1][0]), min(shw[2][0], shw[3][0]));
            dst[1] = min(min(shw[0][1], shw[1][1]), min(shw[2][1], shw[3][1]));
            dst[2] = max(max(shw[0][2], shw[1][2]), max(shw[2][2], shw[3][2]));
            dst[3] = max(max(shw[0][3], shw[1][3]), max(shw[2][3], shw[3][3]));
        }
        if (blockIdx.x == (unsigned)pix_blocks && threadIdx.x == 0) {
            sc->cnt_old = n;
            sc->cnt_dst = 0;
        }
    }
}

// ---- B: local arg-max of height per key (scatter_max, mapper.py:471-472) ----
__global__ __launch_bounds__(kThreads) void k_local_argmax(const float4* __restrict__ rec, int B, int H, int W,
                                                           float half_res, Scalars* sc,
                                                           unsigned long long* __restrict__ tab64,
                                                           int64_t table_cells, const int* __restrict__ bmmL,
                                                           int n_partials) {
    __shared__ int mm[4];
    block_minmax_fold(bmmL, n_partials, mm, sc->mmL);
    int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (pix >= (int64_t)B * H * W) return;
    float4 p = rec[pix];
    if (p.w == 0.f) return;
    int b = (int)(pix / ((int64_t)W * H));
    int64_t key = make_key(b, cell_index(p.z, half_res), cell_index(p.x, half_res), mm);
    if (key < 0 || key >= table_cells) {
        sc->err = IVLN_E_KEYSPACE;
        return;
    }
    unsigned long long packed = ((unsigned long long)ord_f32(p.y) << 32) | (0xFFFFFFFFull - (unsigned long long)pix);
    atomicMax(&tab64[key], packed);
}

// ---- C: local survivors -> append to the world source buffer (mapper.py:444, 844) ----
__global__ __launch_bounds__(kThreads) void k_local_select(const float4* __restrict__ rec,
                                                           const uint8_t* __restrict__ labels, int B, int H,
                                                           int W, float half_res, Scalars* sc,
                                                           unsigned long long* __restrict__ tab64,
                                                           int64_t table_cells, Pt* __restrict__ wsrc,
                                                           int64_t* __restrict__ rsrc, unsigned capacity,
                                                           int* __restrict__ bmmW_local) {
    int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    bool win = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t key = 0;
    int b = 0, r = 0, c = 0;
    if (pix < (int64_t)B * H * W) {
        p = rec[pix];
        if (p.w != 0.f) {
            b = (int)(pix / ((int64_t)W * H));
            r = cell_index(p.z, half_res);
            c = cell_index(p.x, half_res);
            key = make_key(b, r, c, sc->mmL);
            if (key >= 0 && key < table_cells) {
                unsigned long long packed =
                    ((unsigned long long)ord_f32(p.y) << 32) | (0xFFFFFFFFull - (unsigned long long)pix);
                win = (tab64[key] == packed);
            }
        }
    }
    unsigned slot = wave_append(win, &sc->cnt_src, capacity, &sc->err);
    bool stored = win && slot != 0xFFFFFFFFu;
    if (win) tab64[key] = 0ull;  // leave the table clean for the world phase
    if (stored) {
        Pt q;
        q.x = p.x; q.y = p.y; q.z = p.z;
        q.meta = ((uint32_t)b << 8) | labels[pix];
        wsrc[slot] = q;
        rsrc[slot] = (int64_t)(kLocalRankBit | (uint64_t)key);
    }
    block_minmax_store(stored, r, c, bmmW_local + 4 * blockIdx.x);
}

__device__ __forceinline__ bool world_alive(const Pt& p, unsigned i, unsigned cnt_old, int B,
                                            const uint8_t* not_done) {
    int b = (int)(p.meta >> 8);
    if (i >= cnt_old) return true;  // appended this step
    return b < B && not_done[b] != 0;
}

// ---- D: world phase A - max height per key ----
__global__ __launch_bounds__(kThreads) void k_world_max(const Pt* __restrict__ wsrc, int B,
                                                        const uint8_t* __restrict__ not_done, float half_res,
                                                        Scalars* sc, unsigned* __restrict__ tab32,
                                                        int64_t table_cells, unsigned capacity,
                                                        const int* __restrict__ bmmW, int n_partials) {
    __shared__ int mm[4];
    block_minmax_fold(bmmW, n_partials, mm, sc->mmW);
    unsigned n = min(sc->cnt_src, capacity);
    unsigned cnt_old = sc->cnt_old;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pt p = wsrc[i];
        if (!world_alive(p, i, cnt_old, B, not_done)) continue;
        int64_t key = make_key((int)(p.meta >> 8), cell_index(p.z, half_res), cell_index(p.x, half_res), mm);
        if (key < 0 || key >= table_cells) {
            sc->err = IVLN_E_KEYSPACE;
            continue;
        }
        atomicMax(&tab32[key], ord_f32(p.y));
    }
}

// ---- E: world phase B - first (lowest rank) among the points attaining the max ----
__global__ __launch_bounds__(kThreads) void k_world_first(const Pt* __restrict__ wsrc,
                                                          const int64_t* __restrict__ rsrc, int B,
                                                          const uint8_t* __restrict__ not_done, float half_res,
                                                          Scalars* sc, const unsigned* __restrict__ tab32,
                                                          unsigned long long* __restrict__ tab64,
                                                          int64_t table_cells, unsigned capacity) {
    unsigned n = min(sc->cnt_src, capacity);
    unsigned cnt_old = sc->cnt_old;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pt p = wsrc[i];
        if (!world_alive(p, i, cnt_old, B, not_done)) continue;
        int64_t key = make_key((int)(p.meta >> 8), cell_index(p.z, half_res), cell_index(p.x, half_res), sc->mmW);
        if (key < 0 || key >= table_cells) continue;
        if (tab32[key] == ord_f32(p.y)) atomicMax(&tab64[key], ~(unsigned long long)rsrc[i]);
    }
}

// Raster one world point into the egocentric maps (mapper.py:884-901, 255-266, 513-531, 569-571).
__device__ __forceinline__ void raster_point(const Pt& p, uint64_t rank, const float* __restrict__ pose,
                                             const float* __restrict__ rot, int rows, int cols, float res,
                                             float half_h, float half_w, uint8_t* __restrict__ occ,
                                             unsigned long long* __restrict__ cell) {
    int b = (int)(p.meta >> 8);
    float h = pose[3 * b + 1];
    if (!(p.y > __fsub_rn(h, 1.25f) && p.y < __fadd_rn(h, 0.75f))) return;
    float x = __fadd_rn(p.x, -pose[3 * b + 0]);
    float y = __fadd_rn(p.y, -pose[3 * b + 1]);
    float z = __fadd_rn(p.z, -pose[3 * b + 2]);
    const float* r = rot + 9 * b;
    float xr = __fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z));
    float zr = __fadd_rn(__fadd_rn(__fmul_rn(r[6], x), __fmul_rn(r[7], y)), __fmul_rn(r[8], z));
    float fr = rintf(__fdiv_rn(__fadd_rn(zr, half_h), res));
    float fc = rintf(__fdiv_rn(__fadd_rn(xr, half_w), res));
    if (!(fr >= 0.f && fr < (float)rows && fc >= 0.f && fc < (float)cols)) return;
    int o = (b * rows + (int)fr) * cols + (int)fc;
    occ[o] = 1;
    uint32_t label = p.meta & 0xFFu;
    if (label != 0) {
        const unsigned long long v = ((unsigned long long)(rank + 1) << 8) | label;
        if (v > __hip_atomic_load(&cell[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cell[o], v);
    }
}

// ---- F: world survivors -> destination buffer + raster ----
__global__ __launch_bounds__(kThreads) void k_world_select(
    const Pt* __restrict__ wsrc, const int64_t* __restrict__ rsrc, int B, const uint8_t* __restrict__ not_done,
    float half_res, Scalars* sc, unsigned* __restrict__ tab32, unsigned long long* __restrict__ tab64,
    int64_t table_cells, Pt* __restrict__ wdst, int64_t* __restrict__ rdst, unsigned capacity,
    const float* __restrict__ pose, const float* __restrict__ rot, int rows, int cols, float res, float half_h,
    float half_w, uint8_t* __restrict__ occ, unsigned long long* __restrict__ cell) {
    unsigned n = min(sc->cnt_src, capacity);
    unsigned cnt_old = sc->cnt_old;
    unsigned iters = (n + gridDim.x * kThreads - 1) / (gridDim.x * kThreads);
    for (unsigned it = 0; it < iters; ++it) {
        unsigned i = (it * gridDim.x + blockIdx.x) * kThreads + threadIdx.x;
        bool win = false;
        Pt p;
        p.x = p.y = p.z = 0.f;
        p.meta = 0;
        int64_t key = 0;
        if (i < n) {
            p = wsrc[i];
            if (world_alive(p, i, cnt_old, B, not_done)) {
                key = make_key((int)(p.meta >> 8), cell_index(p.z, half_res), cell_index(p.x, half_res), sc->mmW);
                if (key >= 0 && key < table_cells)
                    win = (tab32[key] == ord_f32(p.y)) && (tab64[key] == ~(unsigned long long)rsrc[i]);
            }
        }
        unsigned slot = wave_append(win, &sc->cnt_dst, capacity, &sc->err);
        if (win) {
            tab32[key] = 0u;
            tab64[key] = 0ull;
            if (slot != 0xFFFFFFFFu) {
                wdst[slot] = p;
                rdst[slot] = key;
            }
            raster_point(p, (uint64_t)key, pose, rot, rows, cols, res, half_h, half_w, occ, cell);
        }
    }
}

// ---- G: semantic map from the per-cell winners; copy the surviving cloud back into the source
// buffer (fixed pointers -> the whole step can be captured in a hipGraph and replayed); reset scalars ----
__global__ __launch_bounds__(kThreads) void k_finalize(unsigned long long* __restrict__ cell,
                                                       uint8_t* __restrict__ sem, int map_cells, Scalars* sc,
                                                       int swap, unsigned capacity, const Pt* __restrict__ wdst,
                                                       const int64_t* __restrict__ rdst, Pt* __restrict__ wsrc,
                                                       int64_t* __restrict__ rsrc) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < map_cells) {
        unsigned long long v = cell[i];
        sem[i] = (uint8_t)(v & 0xFFull);
        cell[i] = 0ull;
    }
    if (swap) {
        unsigned n = min(sc->cnt_dst, capacity);  // cnt_dst is zeroed by the NEXT step's first kernel
        for (unsigned j = i; j < n; j += gridDim.x * kThreads) {
            wsrc[j] = wdst[j];
            rsrc[j] = rdst[j];
        }
        if (i == 0) sc->cnt_src = n;
    }
    if (i == 0) {
        sc->mmL[0] = sc->mmL[1] = sc->mmW[0] = sc->mmW[1] = INT32_MAX;
        sc->mmL[2] = sc->mmL[3] = sc->mmW[2] = sc->mmW[3] = INT32_MIN;
        sc->cnt_old = 0;
    }
}

// ---- known-map mode ----
__global__ __launch_bounds__(kThreads) void k_known_clear(const Pt* __restrict__ wsrc,
                                                          const int64_t* __restrict__ rsrc, int B,
                                                          const uint8_t* __restrict__ not_done, Scalars* sc,
                                                          Pt* __restrict__ wdst, int64_t* __restrict__ rdst,
                                                          unsigned capacity) {
    unsigned n = min(sc->cnt_src, capacity);
    unsigned iters = (n + gridDim.x * kThreads - 1) / (gridDim.x * kThreads);
    for (unsigned it = 0; it < iters; ++it) {
        unsigned i = (it * gridDim.x + blockIdx.x) * kThreads + threadIdx.x;
        bool keep = false;
        Pt p;
        p.x = p.y = p.z = 0.f;
        p.meta = 0;
        if (i < n) {
            p = wsrc[i];
            int b = (int)(p.meta >> 8);
            keep = b < B && not_done[b] != 0;
        }
        unsigned slot = wave_append(keep, &sc->cnt_dst, capacity, &sc->err);
        if (keep && slot != 0xFFFFFFFFu) {
            wdst[slot] = p;
            rdst[slot] = rsrc[i];
        }
    }
}

__global__ void k_swap_counts(Scalars* sc, unsigned capacity) {
    sc->cnt_src = min(sc->cnt_dst, capacity);
    sc->cnt_dst = 0;
}

__global__ __launch_bounds__(kThreads) void k_known_load(const float* __restrict__ xyz,
                                                         const uint8_t* __restrict__ semv, int64_t n, int b,
                                                         int64_t rank_base, Scalars* sc, Pt* __restrict__ wsrc,
                                                         int64_t* __restrict__ rsrc, unsigned capacity) {
    // sequential ranks keep the file order (mapper.py:283-294: order of the npz arrays)
    unsigned start = sc->cnt_src;  // same value for every block: counts are bumped by k_known_commit
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        uint64_t slot = (uint64_t)start + (uint64_t)i;
        if (slot >= capacity) {
            sc->err = IVLN_E_CAPACITY;
            continue;
        }
        Pt q;
        q.x = xyz[3 * i]; q.y = xyz[3 * i + 1]; q.z = xyz[3 * i + 2];
        q.meta = ((uint32_t)b << 8) | semv[i];
        wsrc[slot] = q;
        rsrc[slot] = rank_base + i;
    }
}

__global__ void k_known_commit(Scalars* sc, int64_t n, unsigned capacity) {
    uint64_t v = (uint64_t)sc->cnt_src + (uint64_t)n;
    sc->cnt_src = (unsigned)(v > capacity ? capacity : v);
}

__global__ __launch_bounds__(kThreads) void k_known_raster(const Pt* __restrict__ wsrc,
                                                           const int64_t* __restrict__ rsrc, Scalars* sc,
                                                           const float* __restrict__ pose,
                                                           const float* __restrict__ rot, int B, int rows,
                                                           int cols, float res, float half_h, float half_w,
                                                           uint8_t* __restrict__ occ,
                                                           unsigned long long* __restrict__ cell) {
    unsigned n = sc->cnt_src;  // clamped by k_known_commit / k_swap_counts
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        Pt p = wsrc[i];
        if ((int)(p.meta >> 8) >= B) continue;
        raster_point(p, (uint64_t)rsrc[i], pose, rot, rows, cols, res, half_h, half_w, occ, cell);
    }
}

__global__ void k_zero_u8(uint8_t* p, int n) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ void k_frames(const float* __restrict__ pose, const double* __restrict__ orient, int B,
                         float* __restrict__ T, float* __restrict__ rot) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double elev = orient[2 * b + 0] + 3.141592653589793;  // mapper.py:135 elevation + torch.pi
    double head = orient[2 * b + 1];
    double cx = cos(elev), sx = sin(elev), cy = cos(head), sy = sin(head);
    float* t = T + 16 * b;  // core.py:20-36
    t[0] = (float)cy;    t[1] = (float)(sx * sy); t[2] = (float)(cx * sy);  t[3] = pose[3 * b + 0];
    t[4] = 0.f;          t[5] = (float)cx;        t[6] = (float)(-sx);      t[7] = pose[3 * b + 1];
    t[8] = (float)(-sy); t[9] = (float)(cy * sx); t[10] = (float)(cy * cx); t[11] = pose[3 * b + 2];
    t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
    double a = -head;  // mapper.py:266 rotate_around_y(-origin.heading)
    float* r = rot + 9 * b;
    r[0] = (float)cos(a);    r[1] = 0.f; r[2] = (float)sin(a);
    r[3] = 0.f;              r[4] = 1.f; r[5] = 0.f;
    r[6] = (float)(-sin(a)); r[7] = 0.f; r[8] = (float)cos(a);
}

}  // namespace

struct ivln_mapper {
    int B_max, H, W, rows, cols;
    float res, half_h, half_w, half_res;
    int64_t capacity, table_cells;
    float *xs, *ys;
    float4* rec;
    Pt* wbuf[2];
    int64_t* rbuf[2];
    int cur;  // index of the source buffer
    unsigned long long* tab64;
    unsigned* tab32;
    unsigned long long* cell;
    Scalars* sc;
    int* bmmL;  // per-block min/max partials (local cloud)
    int* bmmW;  // per-block partials (old world blocks, then local-select blocks)
    int64_t known_rank;
};

#define HIPCHK(x)                          \
    do {                                   \
        if ((x) != hipSuccess) return IVLN_E_HIP; \
    } while (0)

extern "C" {

const char* ivln_strerror(int code) {
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

static int init_scalars(ivln_mapper* m, hipStream_t s) {
    Scalars h;
    h.mmL[0] = h.mmL[1] = h.mmW[0] = h.mmW[1] = INT32_MAX;
    h.mmL[2] = h.mmL[3] = h.mmW[2] = h.mmW[3] = INT32_MIN;
    h.cnt_src = h.cnt_old = h.cnt_dst = 0;
    h.err = 0;
    HIPCHK(hipMemcpyAsync(m->sc, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return IVLN_OK;
}

int ivln_mapper_create(int B_max, int H, int W, double vfov_rad, double height_m, double width_m, double res_m,
                       int64_t world_capacity, int64_t table_cells, ivln_mapper** out) {
    if (!out || B_max <= 0 || H <= 0 || W <= 0 || res_m <= 0) return IVLN_E_INVALID;
    ivln_mapper* m = new (std::nothrow) ivln_mapper();
    if (!m) return IVLN_E_INVALID;
    m->B_max = B_max; m->H = H; m->W = W;
    m->rows = (int)ceil(height_m / res_m);  // mapper.py:97-99
    m->cols = (int)ceil(width_m / res_m);
    m->res = (float)res_m;
    m->half_h = (float)(height_m / 2);
    m->half_w = (float)(width_m / 2);
    m->half_res = (float)(res_m / 2);
    m->capacity = world_capacity > 0 ? world_capacity : (int64_t)B_max * (1 << 20);
    if (m->capacity > 0xFFFFFF00ll) m->capacity = 0xFFFFFF00ll;
    m->table_cells = table_cells > 0 ? table_cells : (int64_t)(16 << 20);
    m->cur = 0;
    m->known_rank = 0;
    // core.py:70-115: intrinsics in python doubles -> fp32; (u + 0.5 - cx) / fx in fp32
    double hfov = (double)W / (double)H * vfov_rad;
    float fx = (float)((double)W / (2.0 * tan(hfov / 2.0)));
    float fy = (float)((double)H / (2.0 * tan(vfov_rad / 2.0)));
    float cx = (float)(W / 2.0), cy = (float)(H / 2.0);
    float* hx = new float[W];
    float* hy = new float[H];
    for (int u = 0; u < W; ++u) {
        volatile float t = (float)u + 0.5f;
        t = t - cx;
        hx[u] = t / fx;
    }
    for (int v = 0; v < H; ++v) {
        volatile float t = (float)v + 0.5f;
        t = t - cy;
        hy[v] = t / fy;
    }
    int cells = B_max * m->rows * m->cols;
    bool ok = hipMalloc(&m->xs, sizeof(float) * W) == hipSuccess && hipMalloc(&m->ys, sizeof(float) * H) == hipSuccess &&
              hipMalloc(&m->rec, sizeof(float4) * (size_t)B_max * H * W) == hipSuccess &&
              hipMalloc(&m->wbuf[0], sizeof(Pt) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->wbuf[1], sizeof(Pt) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->rbuf[0], sizeof(int64_t) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->rbuf[1], sizeof(int64_t) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->tab64, sizeof(unsigned long long) * (size_t)m->table_cells) == hipSuccess &&
              hipMalloc(&m->tab32, sizeof(unsigned) * (size_t)m->table_cells) == hipSuccess &&
              hipMalloc(&m->cell, sizeof(unsigned long long) * (size_t)cells) == hipSuccess &&
              hipMalloc(&m->sc, sizeof(Scalars)) == hipSuccess &&
              hipMalloc(&m->bmmL, sizeof(int) * 4 * (((size_t)B_max * H * W + kThreads - 1) / kThreads + 1)) == hipSuccess &&
              hipMalloc(&m->bmmW, sizeof(int) * 4 * (256 + ((size_t)B_max * H * W + kThreads - 1) / kThreads + 1)) == hipSuccess;
    if (ok) {
        ok = hipMemcpy(m->xs, hx, sizeof(float) * W, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(m->ys, hy, sizeof(float) * H, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(m->tab64, 0, sizeof(unsigned long long) * (size_t)m->table_cells) == hipSuccess &&
             hipMemset(m->tab32, 0, sizeof(unsigned) * (size_t)m->table_cells) == hipSuccess &&
             hipMemset(m->cell, 0, sizeof(unsigned long long) * (size_t)cells) == hipSuccess &&
             init_scalars(m, nullptr) == IVLN_OK;
    }
    delete[] hx;
    delete[] hy;
    if (!ok) {
        ivln_mapper_destroy(m);
        return IVLN_E_HIP;
    }
    *out = m;
    return IVLN_OK;
}

int ivln_mapper_destroy(ivln_mapper* m) {
    if (!m) return IVLN_OK;
    (void)hipFree(m->xs); (void)hipFree(m->ys); (void)hipFree(m->rec);
    (void)hipFree(m->wbuf[0]); (void)hipFree(m->wbuf[1]); (void)hipFree(m->rbuf[0]); (void)hipFree(m->rbuf[1]);
    (void)hipFree(m->tab64); (void)hipFree(m->tab32); (void)hipFree(m->cell); (void)hipFree(m->sc); (void)hipFree(m->bmmL); (void)hipFree(m->bmmW);
    delete m;
    return IVLN_OK;
}

int ivln_mapper_reset(ivln_mapper* m, void* stream) {
    if (!m) return IVLN_E_INVALID;
    m->known_rank = 0;
    return init_scalars(m, (hipStream_t)stream);
}

int ivln_mapper_frames(const float* pose, const double* orientation, int B, float* T, float* rot, void* stream) {
    if (B <= 0) return IVLN_E_INVALID;
    hipLaunchKernelGGL(k_frames, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, orientation, B, T, rot);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_step(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                     const float* rot, const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out,
                     void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * m->H * m->W;
    const int pix_blocks = (int)((npix + kThreads - 1) / kThreads);
    const int world_blocks = 256;
    const int map_cells = B * m->rows * m->cols;
    Pt* wsrc = m->wbuf[m->cur];
    Pt* wdst = m->wbuf[m->cur ^ 1];
    int64_t* rsrc = m->rbuf[m->cur];
    int64_t* rdst = m->rbuf[m->cur ^ 1];
    hipLaunchKernelGGL(k_local_unproject, dim3(pix_blocks + world_blocks), dim3(kThreads), 0, s, depth, T, pose,
                       not_done, m->xs, m->ys, B, m->H, m->W, m->half_res, m->rec, wsrc, m->sc, occ_out, map_cells,
                       pix_blocks, m->bmmL, m->bmmW);
    hipLaunchKernelGGL(k_local_argmax, dim3(pix_blocks), dim3(kThreads), 0, s, m->rec, B, m->H, m->W, m->half_res,
                       m->sc, m->tab64, m->table_cells, m->bmmL, pix_blocks);
    hipLaunchKernelGGL(k_local_select, dim3(pix_blocks), dim3(kThreads), 0, s, m->rec, labels, B, m->H, m->W,
                       m->half_res, m->sc, m->tab64, m->table_cells, wsrc, rsrc, (unsigned)m->capacity,
                       m->bmmW + 4 * world_blocks);
    hipLaunchKernelGGL(k_world_max, dim3(world_blocks), dim3(kThreads), 0, s, wsrc, B, not_done, m->half_res, m->sc,
                       m->tab32, m->table_cells, (unsigned)m->capacity, m->bmmW, world_blocks + pix_blocks);
    hipLaunchKernelGGL(k_world_first, dim3(world_blocks), dim3(kThreads), 0, s, wsrc, rsrc, B, not_done, m->half_res,
                       m->sc, m->tab32, m->tab64, m->table_cells, (unsigned)m->capacity);
    hipLaunchKernelGGL(k_world_select, dim3(world_blocks), dim3(kThreads), 0, s, wsrc, rsrc, B, not_done, m->half_res,
                       m->sc, m->tab32, m->tab64, m->table_cells, wdst, rdst, (unsigned)m->capacity, pose, rot,
                       m->rows, m->cols, m->res, m->half_h, m->half_w, occ_out, m->cell);
    int fin_blocks = (map_cells + kThreads - 1) / kThreads;
    if (fin_blocks < 256) fin_blocks = 256;
    hipLaunchKernelGGL(k_finalize, dim3(fin_blocks), dim3(kThreads), 0, s, m->cell, sem_out, map_cells, m->sc, 1,
                       (unsigned)m->capacity, wdst, rdst, wsrc, rsrc);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_known_begin(ivln_mapper* m, const uint8_t* not_done, int B, void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_known_clear, dim3(256), dim3(kThreads), 0, s, m->wbuf[m->cur], m->rbuf[m->cur], B, not_done,
                       m->sc, m->wbuf[m->cur ^ 1], m->rbuf[m->cur ^ 1], (unsigned)m->capacity);
    hipLaunchKernelGGL(k_swap_counts, dim3(1), dim3(1), 0, s, m->sc, (unsigned)m->capacity);
    m->cur ^= 1;
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_load_known(ivln_mapper* m, int b, const float* xyz, const uint8_t* sem, int64_t n, void* stream) {
    if (!m || b < 0 || b >= m->B_max || n < 0) return IVLN_E_INVALID;
    if (n == 0) return IVLN_OK;
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_known_load, dim3(blocks), dim3(kThreads), 0, s, xyz, sem, n, b, m->known_rank, m->sc,
                       m->wbuf[m->cur], m->rbuf[m->cur], (unsigned)m->capacity);
    hipLaunchKernelGGL(k_known_commit, dim3(1), dim3(1), 0, s, m->sc, n, (unsigned)m->capacity);
    m->known_rank += n;
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_known_raster(ivln_mapper* m, const float* pose, const float* rot, int B, uint8_t* occ_out,
                             uint8_t* sem_out, void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int map_cells = B * m->rows * m->cols;
    hipLaunchKernelGGL(k_zero_u8, dim3((map_cells + kThreads - 1) / kThreads), dim3(kThreads), 0, s, occ_out, map_cells);
    hipLaunchKernelGGL(k_known_raster, dim3(256), dim3(kThreads), 0, s, m->wbuf[m->cur], m->rbuf[m->cur], m->sc, pose,
                       rot, B, m->rows, m->cols, m->res, m->half_h, m->half_w, occ_out, m->cell);
    hipLaunchKernelGGL(k_finalize, dim3((map_cells + kThreads - 1) / kThreads), dim3(kThreads), 0, s, m->cell,
                       sem_out, map_cells, m->sc, 0, (unsigned)m->capacity, nullptr, nullptr, nullptr, nullptr);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_status(ivln_mapper* m, int64_t* world_n, void* stream) {
    if (!m) return IVLN_E_INVALID;
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, m->sc, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (world_n) *world_n = (int64_t)h.cnt_src;
    return h.err;
}

int ivln_mapper_world_export(ivln_mapper* m, float* xyz, uint32_t* meta, int64_t* rank, int64_t max_n,
                             int64_t* n_out, void* stream) {
    if (!m || !n_out) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, m->sc, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    int64_t n = (int64_t)h.cnt_src;
    *n_out = n;
    if (n > max_n) n = max_n;
    if (n > 0) {
        // strided device->device copies out of the 16-byte point records
        HIPCHK(hipMemcpy2DAsync(xyz, 12, m->wbuf[m->cur], sizeof(Pt), 12, (size_t)n, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync(meta, 4, (const char*)m->wbuf[m->cur] + 12, sizeof(Pt), 4, (size_t)n,
                                hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(rank, m->rbuf[m->cur], sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return IVLN_OK;
}

}  // extern "C"
