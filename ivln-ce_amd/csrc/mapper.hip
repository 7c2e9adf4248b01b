// Egocentric semantic mapper for gfx950 - bit-exact with the reference's MappingModule
// (ivlnce_baselines/common/mapping_module/mapper.py:904-944) on identical depth/label inputs.
//
// The reference runs ~150 tiny torch ops per step, two torch_scatter.scatter_max calls, sorts the
// cloud by a (colliding) cell hash and relies on last-writer-wins index_put_.  Here one step is 6
// launches and the cloud is an UNORDERED bag: every decision the reference takes by position in its
// sorted cloud is taken by a `rank` carried with each point (= its key in the last keep-highest), so
// append order does not matter and no sort is needed:
//   keep-highest arg-max  -> ONE 64-bit atomicMax of (orderable height, ~index | ~rank) in a dense per-key
//                            table: highest point, and among equal heights the first in cloud order
//   last-writer-wins map  -> atomicMax of (rank<<8 | label) per 10 cm map cell
// HBM traffic (the thing to minimise - this is byte work): the local-cloud kernels re-derive
// a pixel's point from its 4-byte depth instead of passing 16-byte records between them; the world cloud is
// read twice (keep-highest, select) and written once per step, its two buffers swap by a device-side index
// (no copy-back), and the bounding box of the surviving OLD cloud comes from per-env boxes the previous
// select left behind (no third pass).  Round 1's 7-launch version moved 54 MB per 4-env step; see
// profiles/r02_rollout_pmc_traffic.json for this one.
// Float recipe (verified against goldens through oracle/mapper_ref.c): camera->world is an fmaf
// chain over k=0..3, the ego rotation is un-fused (a*x + b*y) + c*z, divisions are IEEE, rounding is
// half-to-even.  Compile with -ffp-contract=off; every op below is an explicit __f*_rn.
//
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <new>
#include "../../include/ivln_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPPT = 4;                       // pixels per thread in the local-cloud kernels
constexpr uint32_t kLocalRankBit = 1u << 31;  // rank of a point that entered the cloud this step = bit | local key
constexpr int kFrameEnvs = 64;                // envs a workgroup derives camera transforms for (B_max <= 64)

struct Scalars {       // device-side scalars of one mapper
    int mmL[4];        // local cloud: rmin, cmin, rmax, cmax   (published by k_local_argmax)
    int mmW[4];        // world cloud                           (published by k_world_max)
    int mmWold[4];     // surviving OLD world points            (published by k_local_argmax from the per-env boxes)
    unsigned cnt[2];   // points in world buffer 0 / 1
    unsigned cnt_old;  // snapshot of cnt[cur] before this step's appends
    int cur;           // index of the source buffer (the other one receives this step's survivors)
    int n_bbox;        // blocks of per-env bounding-box partials written by the last k_world_select (0: none)
    int n_bbox_envs;   // envs per block in those partials (the batch size of that step)
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

// {rmin,cmin,rmax,cmax} of a block's threads -> dst4 (wave shuffles + LDS; no atomics: the first version issued 4
// same-address atomics per wave, ~16k serialised L2 atomics = 100-146 us for what is a ~5 us kernel)
__device__ __forceinline__ void block_minmax_store(int rmin, int cmin, int rmax, int cmax, int* __restrict__ dst4) {
    __shared__ int sh[4][4];
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
    __syncthreads();
}

// Fold n partial boxes (and an optional extra box) into mm4 (LDS of the calling block); block 0 also publishes.
__device__ __forceinline__ void block_minmax_fold(const int* __restrict__ partials, int n, const int* extra4, int* mm4,
                                                  int* __restrict__ publish) {
    __shared__ int sh[4][4];
    int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int4 v = *reinterpret_cast<const int4*>(partials + 4 * i);
        rmin = min(rmin, v.x); cmin = min(cmin, v.y); rmax = max(rmax, v.z); cmax = max(cmax, v.w);
    }
    if (extra4 && threadIdx.x == 0) {
        rmin = min(rmin, extra4[0]); cmin = min(cmin, extra4[1]); rmax = max(rmax, extra4[2]); cmax = max(cmax, extra4[3]);
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
// must call it.  (Per-wave atomics on the single counter serialised ~4k same-address atomics: 59 us.)
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

// Slot of a key in the dense arg-max table.  The KEY (above) is the reference's flattened cell number - with its
// multipliers rows.max() / cols.max() (not max + 1: cell (r, C) and cell (r + 1, 0) share a key, quirk Q2) - and it is
// the rank that orders the cloud and breaks ties: it stays what it was.  The table SLOT only has to be a bijection of
// the key, and slot = key makes every probe of a point whose neighbours in the cloud differ in r a separate 64-byte
// line (consecutive points come from consecutive pixels: a wall runs along r as often as along c).  Slots therefore
// tile the key plane (q = key / C, kc = key % C) 4 x 4 = 16 slots = 128 bytes, so that neighbours in either direction
// share lines (counted HBM bytes of the mapper: profiles/README.md, round 3).  -1: outside the table
// (IVLN_E_KEYSPACE; keys are ranks as well and must fit the same 31 bits).
__device__ __forceinline__ int64_t table_slot(int b, int r, int c, const int* mm, int64_t table_cells) {
    const int64_t key = make_key(b, r, c, mm);
    if (key < 0 || key >= table_cells) return -1;
    const int64_t C = (int64_t)mm[3] - mm[1];
    if (C <= 0) return key;  // (one column of cells: every key is a row number already)
    const uint32_t q = (uint32_t)key / (uint32_t)C, kc = (uint32_t)key - q * (uint32_t)C;
    const int64_t Ct = (C + 3) >> 2;
    const int64_t slot = ((((int64_t)(q >> 2)) * Ct + (kc >> 2)) << 4) | ((q & 3) << 2) | (kc & 3);
    return slot < table_cells ? slot : -1;
}

// Unproject + filter one pixel (GenerateSemanticPointCloud.forward, mapper.py:387-425; core.py:137-171).  The three
// local-cloud kernels each RE-DERIVE the point from the depth value (a dozen flops) instead of passing a float4
// record per pixel between them: 4 bytes read per pixel and kernel instead of 16 written + 2 x 16 read.
struct Cam {
    const float* depth; const float* T; const float* pose; const float* xs; const float* ys;
    int B, H, W;
    float half_res;
};
__device__ __forceinline__ bool unproject(const Cam& cm, int64_t pix, float (&w)[3], int& b, int& r, int& c) {
    const int u = (int)(pix % cm.W);
    const int v = (int)((pix / cm.W) % cm.H);
    b = (int)(pix / ((int64_t)cm.W * cm.H));
    const float d = cm.depth[pix];
    if (!(d > 0.01f && d < 0.99f)) return false;  // mapper.py:416-418
    const float z = __fmul_rn(d, 10.0f);          // mapper.py:381-384
    const float x = __fmul_rn(z, cm.xs[u]);       // core.py:137-139
    const float y = __fmul_rn(z, cm.ys[v]);
    const float* t = cm.T + 16 * b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // core.py:171 bmm == fma chain over k
        float acc = __fmul_rn(t[4 * k + 0], x);
        acc = __fmaf_rn(t[4 * k + 1], y, acc);
        acc = __fmaf_rn(t[4 * k + 2], z, acc);
        acc = __fmaf_rn(t[4 * k + 3], 1.0f, acc);
        w[k] = __fsub_rn(acc, 0.0f);  // world_shift_origin == 0 (core.py:214)
    }
    const float h = cm.pose[3 * b + 1];
    if (!(w[1] > __fsub_rn(h, 1.0f) && w[1] < __fadd_rn(h, 0.5f))) return false;  // mapper.py:420-424
    r = cell_index(w[2], cm.half_res);
    c = cell_index(w[0], cm.half_res);
    return true;
}

// Camera-to-world transform of env b (core.py:20-36 with elevation + pi, mapper.py:135): fp64 sin / cos, rounded to fp32.
__device__ __forceinline__ void frame_T(const float* __restrict__ pose, const double* __restrict__ orient, int b, float* t) {
    const double elev = orient[2 * b + 0] + 3.141592653589793;
    const double head = orient[2 * b + 1];
    const double cx = cos(elev), sx = sin(elev), cy = cos(head), sy = sin(head);
    t[0] = (float)cy;    t[1] = (float)(sx * sy); t[2] = (float)(cx * sy);  t[3] = pose[3 * b + 0];
    t[4] = 0.f;          t[5] = (float)cx;        t[6] = (float)(-sx);      t[7] = pose[3 * b + 1];
    t[8] = (float)(-sy); t[9] = (float)(cy * sx); t[10] = (float)(cy * cx); t[11] = pose[3 * b + 2];
    t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
}
// mapper.py:266 rotate_around_y(-origin.heading)
__device__ __forceinline__ void frame_rot(const double* __restrict__ orient, int b, float* r) {
    const double a = -orient[2 * b + 1];
    r[0] = (float)cos(a);    r[1] = 0.f; r[2] = (float)sin(a);
    r[3] = 0.f;              r[4] = 1.f; r[5] = 0.f;
    r[6] = (float)(-sin(a)); r[7] = 0.f; r[8] = (float)cos(a);
}

// ---- A: local min/max of the cell indices; zero the occupancy output.  FRAMES: the step was given (pose,
// orientation) instead of ready transforms - every workgroup derives the B camera transforms it unprojects with into
// LDS (B <= 64 threads, a few fp64 sin / cos each) and workgroup 0 also leaves T and rot in global memory for the
// kernels behind it: the former k_frames launch in front of the step is gone. ----
template <bool FRAMES>
__global__ __launch_bounds__(kThreads) void k_local_minmax(Cam cm, uint8_t* __restrict__ occ, int map_cells,
                                                           int* __restrict__ bmmL, const double* __restrict__ orient,
                                                           float* __restrict__ T_out, float* __restrict__ rot_out) {
    if constexpr (FRAMES) {
        __shared__ float Ts[kFrameEnvs * 16];
        if ((int)threadIdx.x < cm.B) {
            frame_T(cm.pose, orient, threadIdx.x, Ts + 16 * threadIdx.x);
            if (blockIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) T_out[16 * threadIdx.x + k] = Ts[16 * threadIdx.x + k];
                frame_rot(orient, threadIdx.x, rot_out + 9 * threadIdx.x);
            }
        }
        __syncthreads();
        cm.T = Ts;
    }
    const int64_t total = (int64_t)cm.B * cm.H * cm.W;
    int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
    const int64_t nchunks = (total + kThreads * kPPT - 1) / (kThreads * kPPT);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {  // (a narrow grid walks several chunks)
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            const int64_t pix = (chunk * kPPT + i) * kThreads + threadIdx.x;
            float w[3];
            int b, r, c;
            if (pix < total && unproject(cm, pix, w, b, r, c)) {
                rmin = min(rmin, r); cmin = min(cmin, c); rmax = max(rmax, r); cmax = max(cmax, c);
            }
        }
    }
    block_minmax_store(rmin, cmin, rmax, cmax, bmmL + 4 * blockIdx.x);
    // zero the occupancy output (DenseMap.update_map fill_(0), mapper.py:570)
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < map_cells; i += (int64_t)gridDim.x * kThreads) occ[i] = 0;
}

// ---- B: local arg-max of height per key (scatter_max, mapper.py:471-472).  Block 0 also derives the box of the
// surviving OLD world points from the per-env boxes the last k_world_select left (clear_completed_episode_data,
// mapper.py:310-326: envs with not_done == 0 and rows >= B drop out) - no pass over the old cloud. ----
__global__ __launch_bounds__(kThreads) void k_local_argmax(const Cam cm, Scalars* sc, unsigned long long* __restrict__ tab64,
                                                           int64_t table_cells, const int* __restrict__ bmmL, int n_partials,
                                                           const int* __restrict__ bbox, int bbox_B,
                                                           const uint8_t* __restrict__ not_done) {
    __shared__ int mm[4];
    block_minmax_fold(bmmL, n_partials, nullptr, mm, sc->mmL);
    if (blockIdx.x == 0) {
        const int nb = sc->n_bbox, ne = min(sc->n_bbox_envs, cm.B);  // envs >= B are paused: their points drop out
        int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
        for (int i = threadIdx.x; i < nb * ne; i += kThreads) {
            const int blk = i / ne, b = i - blk * ne;
            if (not_done[b] != 0) {
                const int4 v = *reinterpret_cast<const int4*>(bbox + 4 * ((int64_t)blk * bbox_B + b));
                rmin = min(rmin, v.x); cmin = min(cmin, v.y); rmax = max(rmax, v.z); cmax = max(cmax, v.w);
            }
        }
        block_minmax_store(rmin, cmin, rmax, cmax, sc->mmWold);
        if (threadIdx.x == 0) {
            sc->cnt_old = sc->cnt[sc->cur];
            sc->cnt[sc->cur ^ 1] = 0;
        }
    }
    const int64_t total = (int64_t)cm.B * cm.H * cm.W;
    const int64_t nchunks = (total + kThreads * kPPT - 1) / (kThreads * kPPT);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            const int64_t pix = (chunk * kPPT + i) * kThreads + threadIdx.x;
            float w[3];
            int b, r, c;
            if (pix < total && unproject(cm, pix, w, b, r, c)) {
                const int64_t slot = table_slot(b, r, c, mm, table_cells);
                if (slot < 0) {
                    sc->err = IVLN_E_KEYSPACE;
                    continue;
                }
                const unsigned long long packed =
                    ((unsigned long long)ord_f32(w[1]) << 32) | (0xFFFFFFFFull - (unsigned long long)pix);
                atomicMax(&tab64[slot], packed);
            }
        }
    }
}

// ---- C: local survivors -> appended to the world source buffer (mapper.py:444, 844) ----
__global__ __launch_bounds__(kThreads) void k_local_select(const Cam cm, const uint8_t* __restrict__ labels, Scalars* sc,
                                                           unsigned long long* __restrict__ tab64, int64_t table_cells,
                                                           Pt* __restrict__ w0, Pt* __restrict__ w1,
                                                           int64_t* __restrict__ r0, int64_t* __restrict__ r1,
                                                           unsigned capacity, int* __restrict__ bmmA) {
    const int cur = sc->cur;
    Pt* wsrc = cur ? w1 : w0;
    int64_t* rsrc = cur ? r1 : r0;
    const int64_t total = (int64_t)cm.B * cm.H * cm.W;
    int rmin = INT32_MAX, cmin = INT32_MAX, rmax = INT32_MIN, cmax = INT32_MIN;
    const int64_t nchunks = (total + kThreads * kPPT - 1) / (kThreads * kPPT);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
    for (int i = 0; i < kPPT; ++i) {
        const int64_t pix = (chunk * kPPT + i) * kThreads + threadIdx.x;
        float w[3] = {0.f, 0.f, 0.f};
        int b = 0, r = 0, c = 0;
        int64_t key = 0, tslot = -1;
        bool win = false;
        if (pix < total && unproject(cm, pix, w, b, r, c)) {
            key = make_key(b, r, c, sc->mmL);
            tslot = table_slot(b, r, c, sc->mmL, table_cells);
            if (tslot >= 0) {
                const unsigned long long packed =
                    ((unsigned long long)ord_f32(w[1]) << 32) | (0xFFFFFFFFull - (unsigned long long)pix);
                win = (tab64[tslot] == packed);
            }
        }
        const unsigned slot = wave_append(win, &sc->cnt[cur], capacity, &sc->err);
        if (win) tab64[tslot] = 0ull;  // leave the table clean for the world phase
        if (win && slot != 0xFFFFFFFFu) {
            Pt q;
            q.x = w[0]; q.y = w[1]; q.z = w[2];
            q.meta = ((uint32_t)b << 8) | labels[pix];
            wsrc[slot] = q;
            rsrc[slot] = (int64_t)(kLocalRankBit | (uint32_t)key);
            rmin = min(rmin, r); cmin = min(cmin, c); rmax = max(rmax, r); cmax = max(cmax, c);
        }
    }
    block_minmax_store(rmin, cmin, rmax, cmax, bmmA + 4 * blockIdx.x);
}

// workgroups of a world-cloud kernel that take part: enough for `wpt` points per thread, at most the grid
__device__ __forceinline__ unsigned working_blocks(unsigned n, int wpt) {
    const unsigned per = (unsigned)kThreads * (unsigned)(wpt > 0 ? wpt : 1);
    const unsigned need = (n + per - 1) / per;
    return need < 1u ? 1u : (need < gridDim.x ? need : gridDim.x);
}

__device__ __forceinline__ bool world_alive(const Pt& p, unsigned i, unsigned cnt_old, int B, const uint8_t* not_done) {
    const int b = (int)(p.meta >> 8);
    if (i >= cnt_old) return true;  // appended this step
    return b < B && not_done[b] != 0;
}

// ---- D: world keep-highest in ONE pass: per key the max of (orderable height, ~rank) - highest point, and among
// equal heights the lowest rank = the first in the reference's cloud order (torch_scatter.scatter_max on the CPU keeps
// the first maximum; old world points, ordered by their previous key, precede this step's local ones).  Ranks fit 32
// bits (key < table_cells <= 2^31, bit 31 = entered this step), so one 64-bit atomicMax replaces the former
// max-height pass + first-rank pass. ----
__global__ __launch_bounds__(kThreads) void k_world_max(const Pt* __restrict__ w0, const Pt* __restrict__ w1,
                                                        const int64_t* __restrict__ r0, const int64_t* __restrict__ r1,
                                                        int B, const uint8_t* __restrict__ not_done, float half_res,
                                                        Scalars* sc, unsigned long long* __restrict__ tab64,
                                                        int64_t table_cells, unsigned capacity,
                                                        const int* __restrict__ bmmA, int n_partials, int wpt) {
    __shared__ int mm[4];
    block_minmax_fold(bmmA, n_partials, sc->mmWold, mm, sc->mmW);
    const int cur = sc->cur;
    const Pt* wsrc = cur ? w1 : w0;
    const int64_t* rsrc = cur ? r1 : r0;
    const unsigned n = min(sc->cnt[cur], capacity);
    const unsigned cnt_old = sc->cnt_old;
    const unsigned nbw = working_blocks(n, wpt);  // (a narrow launch grows with the cloud: wpt points per thread)
    if (blockIdx.x >= nbw) return;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += nbw * kThreads) {
        const Pt p = wsrc[i];
        if (!world_alive(p, i, cnt_old, B, not_done)) continue;
        const int64_t slot = table_slot((int)(p.meta >> 8), cell_index(p.z, half_res), cell_index(p.x, half_res), mm, table_cells);
        if (slot < 0) {
            sc->err = IVLN_E_KEYSPACE;
            continue;
        }
        const unsigned long long packed = ((unsigned long long)ord_f32(p.y) << 32) | (unsigned long long)(~(uint32_t)rsrc[i]);
        atomicMax(&tab64[slot], packed);
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

__device__ __forceinline__ void finalize_scalars(Scalars* sc, int swap, unsigned capacity, int n_bbox, int n_bbox_envs) {
    if (swap) {
        const int cur = sc->cur;
        sc->cnt[cur ^ 1] = min(sc->cnt[cur ^ 1], capacity);
        sc->cnt[cur] = 0;
        sc->cur = cur ^ 1;
        sc->n_bbox = n_bbox;
        sc->n_bbox_envs = n_bbox_envs;
    }
    sc->mmL[0] = sc->mmL[1] = sc->mmW[0] = sc->mmW[1] = sc->mmWold[0] = sc->mmWold[1] = INT32_MAX;
    sc->mmL[2] = sc->mmL[3] = sc->mmW[2] = sc->mmW[3] = sc->mmWold[2] = sc->mmWold[3] = INT32_MIN;
    sc->cnt_old = 0;
}

// ---- E: world survivors -> the other buffer + raster + per-env bounding boxes for the next step.
// (Folding F into this kernel - the last workgroup to finish converts the cells - was built and measured in round 3:
//  the release fence every one of the 1024 workgroups then needs in front of its ticket, an L2 write-back each, took
//  the kernel from 48 us to 123 us (69 us on an empty cloud); a launch of its own costs F 4.8 us.  profiles/README.md) ----
constexpr int kBoxEnvs = 64;  // envs whose box a block tracks in LDS (B_max <= 64)
__global__ __launch_bounds__(kThreads) void k_world_select(
    Pt* w0, Pt* w1, int64_t* r0, int64_t* r1, int B, const uint8_t* __restrict__ not_done, float half_res, Scalars* sc, unsigned long long* __restrict__ tab64,
    int64_t table_cells, unsigned capacity, const float* __restrict__ pose, const float* __restrict__ rot, int rows,
    int cols, float res, float half_h, float half_w, uint8_t* __restrict__ occ, unsigned long long* __restrict__ cell,
    int* __restrict__ bbox, int bbox_B, int wpt) {
    __shared__ int box[kBoxEnvs][4];
    for (int i = threadIdx.x; i < B * 4; i += kThreads) box[i >> 2][i & 3] = (i & 2) ? INT32_MIN : INT32_MAX;
    __syncthreads();
    const int cur = sc->cur;
    const Pt* wsrc = cur ? w1 : w0;
    const int64_t* rsrc = cur ? r1 : r0;
    Pt* wdst = cur ? w0 : w1;  // the OTHER buffer
    int64_t* rdst = cur ? r0 : r1;
    const unsigned n = min(sc->cnt[cur], capacity);
    const unsigned cnt_old = sc->cnt_old;
    const unsigned nbw = working_blocks(n, wpt);
    const unsigned iters = blockIdx.x < nbw ? (n + nbw * kThreads - 1) / (nbw * kThreads) : 0;  // (idle blocks leave empty boxes)
    for (unsigned it = 0; it < iters; ++it) {
        const unsigned i = (it * nbw + blockIdx.x) * kThreads + threadIdx.x;
        bool win = false;
        Pt p;
        p.x = p.y = p.z = 0.f;
        p.meta = 0;
        int64_t key = 0, tslot = -1;
        int r = 0, c = 0;
        if (i < n) {
            p = wsrc[i];
            if (world_alive(p, i, cnt_old, B, not_done)) {
                r = cell_index(p.z, half_res);
                c = cell_index(p.x, half_res);
                key = make_key((int)(p.meta >> 8), r, c, sc->mmW);
                tslot = table_slot((int)(p.meta >> 8), r, c, sc->mmW, table_cells);
                if (tslot >= 0) {
                    const unsigned long long packed =
                        ((unsigned long long)ord_f32(p.y) << 32) | (unsigned long long)(~(uint32_t)rsrc[i]);
                    win = (tab64[tslot] == packed);
                }
            }
        }
        const unsigned slot = wave_append(win, &sc->cnt[cur ^ 1], capacity, &sc->err);
        if (win) {
            tab64[tslot] = 0ull;
            if (slot != 0xFFFFFFFFu) {
                wdst[slot] = p;
                rdst[slot] = key;
                const int b = (int)(p.meta >> 8);
                if (b < B) {  // filtered LDS atomics: a box is touched a handful of times per block
                    if (r < box[b][0]) atomicMin(&box[b][0], r);
                    if (c < box[b][1]) atomicMin(&box[b][1], c);
                    if (r > box[b][2]) atomicMax(&box[b][2], r);
                    if (c > box[b][3]) atomicMax(&box[b][3], c);
                }
            }
            raster_point(p, (uint64_t)key, pose, rot, rows, cols, res, half_h, half_w, occ, cell);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * 4; i += kThreads) bbox[(int64_t)blockIdx.x * bbox_B * 4 + i] = box[i >> 2][i & 3];
}

// ---- F: semantic map from the per-cell winners; swap the buffers (a device-side index: the launch sequence has
// fixed pointers and can be captured in a hipGraph; the former copy of the survivors back into the source buffer -
// 2 x 24 bytes per world point and step - is gone); reset the scalars ----
__global__ __launch_bounds__(kThreads) void k_finalize(unsigned long long* __restrict__ cell, uint8_t* __restrict__ sem,
                                                       int map_cells, Scalars* sc, int swap, unsigned capacity,
                                                       int n_bbox, int n_bbox_envs) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < map_cells) {
        unsigned long long v = cell[i];
        sem[i] = (uint8_t)(v & 0xFFull);
        cell[i] = 0ull;
    }
    if (i == 0) finalize_scalars(sc, swap, capacity, n_bbox, n_bbox_envs);
}

// ---- known-map mode ----
__global__ __launch_bounds__(kThreads) void k_known_clear(Pt* w0, Pt* w1, int64_t* r0, int64_t* r1, int B,
                                                          const uint8_t* __restrict__ not_done, Scalars* sc,
                                                          unsigned capacity) {
    const int cur = sc->cur;
    const Pt* wsrc = cur ? w1 : w0;
    const int64_t* rsrc = cur ? r1 : r0;
    Pt* wdst = cur ? w0 : w1;
    int64_t* rdst = cur ? r0 : r1;
    unsigned n = min(sc->cnt[cur], capacity);
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
        unsigned slot = wave_append(keep, &sc->cnt[cur ^ 1], capacity, &sc->err);
        if (keep && slot != 0xFFFFFFFFu) {
            wdst[slot] = p;
            rdst[slot] = rsrc[i];
        }
    }
}

__global__ void k_swap_counts(Scalars* sc, unsigned capacity) {
    const int cur = sc->cur;
    sc->cnt[cur ^ 1] = min(sc->cnt[cur ^ 1], capacity);
    sc->cnt[cur] = 0;
    sc->cur = cur ^ 1;
}

__global__ __launch_bounds__(kThreads) void k_known_load(const float* __restrict__ xyz,
                                                         const uint8_t* __restrict__ semv, int64_t n, int b,
                                                         int64_t rank_base, Scalars* sc, Pt* w0, Pt* w1, int64_t* r0,
                                                         int64_t* r1, unsigned capacity) {
    // sequential ranks keep the file order (mapper.py:283-294: order of the npz arrays)
    const int cur = sc->cur;
    Pt* wsrc = cur ? w1 : w0;
    int64_t* rsrc = cur ? r1 : r0;
    unsigned start = sc->cnt[cur];  // same value for every block: counts are bumped by k_known_commit
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
    uint64_t v = (uint64_t)sc->cnt[sc->cur] + (uint64_t)n;
    sc->cnt[sc->cur] = (unsigned)(v > capacity ? capacity : v);
}

__global__ __launch_bounds__(kThreads) void k_known_raster(const Pt* w0, const Pt* w1, const int64_t* r0, const int64_t* r1,
                                                           Scalars* sc, const float* __restrict__ pose,
                                                           const float* __restrict__ rot, int B, int rows,
                                                           int cols, float res, float half_h, float half_w,
                                                           uint8_t* __restrict__ occ,
                                                           unsigned long long* __restrict__ cell) {
    const int cur = sc->cur;
    const Pt* wsrc = cur ? w1 : w0;
    const int64_t* rsrc = cur ? r1 : r0;
    unsigned n = sc->cnt[cur];  // clamped by k_known_commit / k_swap_counts
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
    frame_T(pose, orient, b, T + 16 * b);
    frame_rot(orient, b, rot + 9 * b);
}

}  // namespace

struct ivln_mapper {
    int B_max, H, W, rows, cols;
    float res, half_h, half_w, half_res;
    int64_t capacity, table_cells;
    float *xs, *ys;
    Pt* wbuf[2];
    int64_t* rbuf[2];
    unsigned long long* tab64;  // dense per-key arg-max table, left zeroed by the winners
    unsigned long long* cell;   // per map cell: (rank + 1) << 8 | label of the last writer
    Scalars* sc;
    int* bmmL;   // per-block min/max partials of the local cloud (k_local_minmax)
    int* bmmA;   // per-block partials of the points appended this step (k_local_select)
    int* bbox;   // per-block, per-env boxes of the surviving world (k_world_select), read by the NEXT step
    int n_blocks_local;
    int local_blocks, world_blocks;  // launch width of the local- / world-cloud kernels (0: full width)
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
    h.mmL[0] = h.mmL[1] = h.mmW[0] = h.mmW[1] = h.mmWold[0] = h.mmWold[1] = INT32_MAX;
    h.mmL[2] = h.mmL[3] = h.mmW[2] = h.mmW[3] = h.mmWold[2] = h.mmWold[3] = INT32_MIN;
    h.cnt[0] = h.cnt[1] = h.cnt_old = 0;
    h.cur = 0;
    h.n_bbox = 0;
    h.n_bbox_envs = 0;
    h.err = 0;
    HIPCHK(hipMemcpyAsync(m->sc, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return IVLN_OK;
}

constexpr int kWorldBlocks = 1024;  // grid of the world-cloud kernels (grid-stride over the cloud; one point per thread up to 262 k points)

int ivln_mapper_create(int B_max, int H, int W, double vfov_rad, double height_m, double width_m, double res_m,
                       int64_t world_capacity, int64_t table_cells, ivln_mapper** out) {
    if (!out || B_max <= 0 || B_max > kBoxEnvs || H <= 0 || W <= 0 || res_m <= 0) return IVLN_E_INVALID;
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
    if (m->table_cells > (1ll << 31)) m->table_cells = 1ll << 31;  // ranks = keys must fit 31 bits (k_world_max)
    m->known_rank = 0;
    m->local_blocks = m->world_blocks = 0;
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
    m->n_blocks_local = (int)(((int64_t)B_max * H * W + kThreads * kPPT - 1) / (kThreads * kPPT));
    bool ok = hipMalloc(&m->xs, sizeof(float) * W) == hipSuccess && hipMalloc(&m->ys, sizeof(float) * H) == hipSuccess &&
              hipMalloc(&m->wbuf[0], sizeof(Pt) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->wbuf[1], sizeof(Pt) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->rbuf[0], sizeof(int64_t) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->rbuf[1], sizeof(int64_t) * (size_t)m->capacity) == hipSuccess &&
              hipMalloc(&m->tab64, sizeof(unsigned long long) * (size_t)m->table_cells) == hipSuccess &&
              hipMalloc(&m->cell, sizeof(unsigned long long) * (size_t)cells) == hipSuccess &&
              hipMalloc(&m->sc, sizeof(Scalars)) == hipSuccess &&
              hipMalloc(&m->bmmL, sizeof(int) * 4 * (size_t)(m->n_blocks_local + 1)) == hipSuccess &&
              hipMalloc(&m->bmmA, sizeof(int) * 4 * (size_t)(m->n_blocks_local + 1)) == hipSuccess &&
              hipMalloc(&m->bbox, sizeof(int) * 4 * (size_t)kWorldBlocks * B_max) == hipSuccess;
    if (ok) {
        ok = hipMemcpy(m->xs, hx, sizeof(float) * W, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(m->ys, hy, sizeof(float) * H, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(m->tab64, 0, sizeof(unsigned long long) * (size_t)m->table_cells) == hipSuccess &&
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
    (void)hipFree(m->xs); (void)hipFree(m->ys);
    (void)hipFree(m->wbuf[0]); (void)hipFree(m->wbuf[1]); (void)hipFree(m->rbuf[0]); (void)hipFree(m->rbuf[1]);
    (void)hipFree(m->tab64); (void)hipFree(m->cell); (void)hipFree(m->sc); (void)hipFree(m->bmmL); (void)hipFree(m->bmmA);
    (void)hipFree(m->bbox);
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

// The label-free half of a step: camera transforms (posed entry), local min / max, the keep-highest arg-max of the local
// cloud (csrc header: kernels 1-2).  Reads depth and the sensor pose only - with predicted semantics it can run beside the
// network that produces the labels (ivln_mapper_step_begin).
static int mapper_begin(ivln_mapper* m, const float* depth, const float* T, const float* pose, const float* rot,
                        const double* orientation, float* T_out, float* rot_out, const uint8_t* not_done, int B,
                        uint8_t* occ_out, void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * m->H * m->W;
    int lb = (int)((npix + kThreads * kPPT - 1) / (kThreads * kPPT));  // local-cloud chunks: 4 pixels / thread
    if (m->local_blocks > 0 && m->local_blocks < lb) lb = m->local_blocks;  // narrow launch: a block walks several chunks
    const int map_cells = B * m->rows * m->cols;
    if (orientation) {  // posed entry: transforms derived inside the first kernel
        const Cam c0{depth, nullptr, pose, m->xs, m->ys, B, m->H, m->W, m->half_res};
        hipLaunchKernelGGL(k_local_minmax<true>, dim3(lb), dim3(kThreads), 0, s, c0, occ_out, map_cells, m->bmmL,
                           orientation, T_out, rot_out);
        T = T_out;
        rot = rot_out;
    }
    const Cam cm{depth, T, pose, m->xs, m->ys, B, m->H, m->W, m->half_res};
    if (!orientation)
        hipLaunchKernelGGL(k_local_minmax<false>, dim3(lb), dim3(kThreads), 0, s, cm, occ_out, map_cells, m->bmmL,
                           (const double*)nullptr, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_local_argmax, dim3(lb), dim3(kThreads), 0, s, cm, m->sc, m->tab64, m->table_cells, m->bmmL, lb,
                       m->bbox, m->B_max, not_done);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

// The rest of the step begun by mapper_begin (same batch, same T / pose / rot, same launch width): label the surviving local
// points, merge them into the world cloud, raster.  No host-side state links the two halves - a captured step records each
// half once and replays them many times -, the kernels' own state (arg-max table, block partials) does.
static int mapper_finish(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                         const float* rot, const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out, void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * m->H * m->W;
    int lb = (int)((npix + kThreads * kPPT - 1) / (kThreads * kPPT));
    if (m->local_blocks > 0 && m->local_blocks < lb) lb = m->local_blocks;  // (as in mapper_begin)
    const int map_cells = B * m->rows * m->cols;
    const unsigned cap = (unsigned)m->capacity;
    const Cam cm{depth, T, pose, m->xs, m->ys, B, m->H, m->W, m->half_res};
    hipLaunchKernelGGL(k_local_select, dim3(lb), dim3(kThreads), 0, s, cm, labels, m->sc, m->tab64, m->table_cells,
                       m->wbuf[0], m->wbuf[1], m->rbuf[0], m->rbuf[1], cap, m->bmmA);
    // world-cloud kernels: full width = up to 1024 workgroups, one point per thread; a narrow launch (set_launch_width) =
    // at most world_blocks workgroups that take 16 points per thread, so few CUs while the cloud is small
    const int wb = (m->world_blocks > 0 && m->world_blocks < kWorldBlocks) ? m->world_blocks : kWorldBlocks;
    const int wpt = m->world_blocks > 0 ? 16 : 1;
    hipLaunchKernelGGL(k_world_max, dim3(wb), dim3(kThreads), 0, s, m->wbuf[0], m->wbuf[1], m->rbuf[0],
                       m->rbuf[1], B, not_done, m->half_res, m->sc, m->tab64, m->table_cells, cap, m->bmmA, lb, wpt);
    hipLaunchKernelGGL(k_world_select, dim3(wb), dim3(kThreads), 0, s, m->wbuf[0], m->wbuf[1], m->rbuf[0],
                       m->rbuf[1], B, not_done, m->half_res, m->sc, m->tab64, m->table_cells, cap, pose, rot, m->rows,
                       m->cols, m->res, m->half_h, m->half_w, occ_out, m->cell, m->bbox, m->B_max, wpt);
    const int fin_blocks = (map_cells + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_finalize, dim3(fin_blocks), dim3(kThreads), 0, s, m->cell, sem_out, map_cells, m->sc, 1, cap,
                       wb, B);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

static int mapper_step(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                       const float* rot, const double* orientation, float* T_out, float* rot_out,
                       const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out, void* stream) {
    const int rc = mapper_begin(m, depth, T, pose, rot, orientation, T_out, rot_out, not_done, B, occ_out, stream);
    if (rc != IVLN_OK) return rc;
    return mapper_finish(m, depth, labels, orientation ? T_out : T, pose, orientation ? rot_out : rot, not_done, B, occ_out, sem_out, stream);
}

int ivln_mapper_step_begin(ivln_mapper* m, const float* depth, const float* pose, const double* orientation,
                           const uint8_t* not_done, int B, uint8_t* occ_out, float* T_out, float* rot_out, void* stream) {
    if (!depth || !pose || !orientation || !T_out || !rot_out || !not_done || !occ_out) return IVLN_E_INVALID;
    return mapper_begin(m, depth, nullptr, pose, nullptr, orientation, T_out, rot_out, not_done, B, occ_out, stream);
}

int ivln_mapper_step_finish(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                            const float* rot, const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out, void* stream) {
    if (!depth || !labels || !T || !pose || !rot || !not_done || !occ_out || !sem_out) return IVLN_E_INVALID;
    return mapper_finish(m, depth, labels, T, pose, rot, not_done, B, occ_out, sem_out, stream);
}

int ivln_mapper_step(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* T, const float* pose,
                     const float* rot, const uint8_t* not_done, int B, uint8_t* occ_out, uint8_t* sem_out,
                     void* stream) {
    if (!T || !rot) return IVLN_E_INVALID;
    return mapper_step(m, depth, labels, T, pose, rot, nullptr, nullptr, nullptr, not_done, B, occ_out, sem_out, stream);
}

int ivln_mapper_step_posed(ivln_mapper* m, const float* depth, const uint8_t* labels, const float* pose,
                           const double* orientation, const uint8_t* not_done, int B, uint8_t* occ_out,
                           uint8_t* sem_out, float* T_out, float* rot_out, void* stream) {
    if (!orientation || !T_out || !rot_out) return IVLN_E_INVALID;
    return mapper_step(m, depth, labels, nullptr, pose, nullptr, orientation, T_out, rot_out, not_done, B, occ_out, sem_out,
                       stream);
}

int ivln_mapper_set_launch_width(ivln_mapper* m, int local_blocks, int world_blocks) {
    if (!m || local_blocks < 0 || world_blocks < 0) return IVLN_E_INVALID;
    m->local_blocks = local_blocks;
    m->world_blocks = world_blocks;
    return IVLN_OK;
}

int ivln_mapper_known_begin(ivln_mapper* m, const uint8_t* not_done, int B, void* stream) {
    if (!m || B <= 0 || B > m->B_max) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_known_clear, dim3(256), dim3(kThreads), 0, s, m->wbuf[0], m->wbuf[1], m->rbuf[0], m->rbuf[1], B,
                       not_done, m->sc, (unsigned)m->capacity);
    hipLaunchKernelGGL(k_swap_counts, dim3(1), dim3(1), 0, s, m->sc, (unsigned)m->capacity);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_load_known(ivln_mapper* m, int b, const float* xyz, const uint8_t* sem, int64_t n, void* stream) {
    if (!m || b < 0 || b >= m->B_max || n < 0) return IVLN_E_INVALID;
    if (n == 0) return IVLN_OK;
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_known_load, dim3(blocks), dim3(kThreads), 0, s, xyz, sem, n, b, m->known_rank, m->sc,
                       m->wbuf[0], m->wbuf[1], m->rbuf[0], m->rbuf[1], (unsigned)m->capacity);
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
    hipLaunchKernelGGL(k_known_raster, dim3(256), dim3(kThreads), 0, s, m->wbuf[0], m->wbuf[1], m->rbuf[0], m->rbuf[1],
                       m->sc, pose, rot, B, m->rows, m->cols, m->res, m->half_h, m->half_w, occ_out, m->cell);
    hipLaunchKernelGGL(k_finalize, dim3((map_cells + kThreads - 1) / kThreads), dim3(kThreads), 0, s, m->cell,
                       sem_out, map_cells, m->sc, 0, (unsigned)m->capacity, 0, 0);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_mapper_status(ivln_mapper* m, int64_t* world_n, void* stream) {
    if (!m) return IVLN_E_INVALID;
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, m->sc, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (world_n) *world_n = (int64_t)h.cnt[h.cur & 1];
    return h.err;
}

int ivln_mapper_world_export(ivln_mapper* m, float* xyz, uint32_t* meta, int64_t* rank, int64_t max_n,
                             int64_t* n_out, void* stream) {
    if (!m || !n_out) return IVLN_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, m->sc, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int cur = h.cur & 1;
    int64_t n = (int64_t)h.cnt[cur];
    *n_out = n;
    if (n > max_n) n = max_n;
    if (n > 0) {
        // strided device->device copies out of the 16-byte point records
        HIPCHK(hipMemcpy2DAsync(xyz, 12, m->wbuf[cur], sizeof(Pt), 12, (size_t)n, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync(meta, 4, (const char*)m->wbuf[cur] + 12, sizeof(Pt), 4, (size_t)n,
                                hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(rank, m->rbuf[cur], sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return IVLN_OK;
}

}  // extern "C"
