// Persistent sequence GRU: the masked GRU of habitat-lab's RNNStateEncoder over a whole time-major (T*N rows)
// trajectory batch - forward with saved gates and its BPTT - as ONE launch each (call sites
// ivlnce_baselines/models/map_cma_policy.py:314-318, 346-353 under base_il_trainer.py:173-219).  Replaces the T
// dependent k_gru_step / k_gru_bwd_step launches that ivln_cma_seq_fwd/bwd used to enqueue (254 launches, 1.76 ms of
// a 512-row update, profiles/r02_update_T64N8_kernel_stats.csv).
//
// Design (H = 512):
//   * 64 workgroups x 256 threads (IVLN_SEQ_UPB=16: 32 x 512); workgroup b owns hidden units [8 b, 8 b + 8).  A 32-lane
//     group owns one unit and keeps its three W_hh rows (forward) / its W_hh^T row (backward) IN REGISTERS for the
//     whole sequence, split over the lanes exactly like k_gru_step<32> / k_gru_bwd_step split K (lane l owns float4
//     l, l + 32, ...): 48 floats per lane either way; the 3 MB matrix is read once per launch, not once per timestep.
//   * Per timestep every workgroup needs the WHOLE vector the others produced in the step before (h_{t-1}: N x 512
//     floats forward, dgh_t: N x 1536 backward).  It is exchanged through the kernel's own OUTPUT tensors - `out` rows
//     of step t-1, `dgh` rows of step t - which every step writes to a fresh location: producers store write-through at
//     agent scope (`global_store ... sc1`), drain (`s_waitcnt vmcnt(0)`), arrive on one monotonic counter; consumers
//     poll the counter relaxed from one lane, then read the rows with 16-byte `sc1` (L1-bypassing) buffer loads, all in
//     flight at once, into LDS.  No fences: every exchanged word is write-through stored and sc1 loaded
//     (MI355X_MICROARCH.md, "valid forms").  Stores nobody waits for (saved gates, dgi, hp) and the next step's
//     prefetches are issued between the arrival and the poll.
//   * The counter and a give-up flag live in a caller-provided 256-byte `sync_ws` whose first 192 bytes a memset node
//     zeroes in front of the launch (stream-ordered, so replay-safe); word 48 is the STICKY error word the host reads -
//     several launches share one workspace (four per update), and a flag the next launch's memset erased would hide a
//     timed-out earlier one.  Every spin is bounded: on a timeout the error word is set, all
//     workgroups leave, and `ivln_seq_sync_status` reports it - a lost workgroup can never hang the GPU.
//   * Same lane -> K mapping, fma chains and element formulas as the per-step kernels; the cross-lane sums run on the
//     DPP path in a different association.  The two paths agree to ~2e-7 (tests/test_gpu_kernels.py, bar 1e-6), and the
//     persistent path is bit-reproducible run to run, idle or beside a bandwidth-heavy stream.
//   Envelope: H == 512, N <= 64 forward / N <= 16 backward (LDS), else the callers fall back to per-step launches.
//
// Measured (MI355X, T = 64, N = 8; tools/gru_seq_bench.py, tools/gru_seq_phases.py; profiles/r03_gru_seq.txt): 4.7 us
// per forward step and 4.9 us per backward step against 5.8 / 6.5 us for the dependent launches (N = 5: 4.2 / 4.4
// against 5.7 / 6.3).  Anatomy of a forward step at N = 8: staging the 16 KB of h_{t-1} 0.8 us (one sc1 round trip),
// matvec + DPP reduction 1.6 us (latency-bound: 0.7 us of it is the 24 x 5 dependent DPP adds, the same with one or
// two waves per SIMD), element part 0.4 us, drain + arrival 0.5 us, counter wait 1.3 us - three serialised memory
// round trips per step, which is the floor of this form (tools/barrier_bench prices the bare exchange at 1.9-2.2 us).
// First version, for the record: 8-byte atomic loads issued one per loop iteration serialised N/2 round trips per
// step (7.1 us per step, slower than the launches); __shfl_xor reductions (LDS permutes) cost another 0.5 us.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>
#include "../../include/ivln_hip.h"
#include "gru_seq.h"
#include "residency.h"

namespace {

constexpr int HH = 512;          // hidden size this kernel is built for
constexpr int LPU = 32;          // lanes per unit
// UPB = hidden units per workgroup (template parameter): 16 -> 32 workgroups x 512 threads, 8 -> 64 x 256
constexpr unsigned SPIN_MAX = 1u << 21;

typedef unsigned long long u64;

#ifdef GRU_SEQ_TIMING  // tools/gru_seq_phases.py: per-workgroup phase stamps of the forward kernel (100 MHz wall clock)
__device__ u64 g_seq_stamp[64 * 256 * 8];
#define SEQ_STAMP(t, k)                                                                             \
    do {                                                                                            \
        if (threadIdx.x == 0 && (t) < 256) g_seq_stamp[(blockIdx.x * 256 + (t)) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define SEQ_STAMP(t, k)
#endif

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void st_pub(float* p, float v) {   // write-through (sc1) 4-byte store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte L1-bypassing (sc1) load through a buffer descriptor: a builtin, so the compiler tracks it in vmcnt and every
// load of a staging pass is in flight before the first wait (an 8-byte atomic load per iteration serialised N/2 memory
// round trips per step: 7.1 us per step, slower than the launches it replaced).
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_pub16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const v4i x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, /*aux: sc1*/ 16);
    return make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
}

// sum over the 32 lanes of a unit's group on the DPP cross-lane path (VALU; __shfl_xor is an LDS permute per step):
// after the four row steps every lane of a 16-lane row holds the row's sum, row_bcast:15 then adds the lower row's
// sum into the upper row - the group total lives in lanes 16..31 of the group.
__device__ __forceinline__ float group_sum_hi(float v) {
#define IVLN_DPP_ADD(ctrl, row_mask)                                                                                   \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    IVLN_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
    IVLN_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
    IVLN_DPP_ADD(0x141, 0xf);  // row_half_mirror
    IVLN_DPP_ADD(0x140, 0xf);  // row_mirror
    IVLN_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1, 3 (the upper half of each 32-lane group)
#undef IVLN_DPP_ADD
    return v;
}

// The exchange in two halves, so that stores nobody waits for (the saved gates) and the next step's prefetches can be
// issued between them.  grid_arrive: every wave drains its write-through stores, one lane bumps the counter.
// (Gathering a workgroup's slice through LDS into a few 16-byte sc1 stores from one wave was measured and is not
// faster: 4.65 vs 4.71 us per forward step, 5.06 vs 4.94 backward - the extra barrier costs what the wide stores save.)
__device__ __forceinline__ void grid_arrive(unsigned* sync_ws) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync_ws[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// grid_wait: one lane polls (relaxed, bounded) until `target` arrivals; false on timeout (error word set).
__device__ __forceinline__ bool grid_wait(unsigned* sync_ws, unsigned target, int* s_fail) {
    if (threadIdx.x == 0) {
        bool ok = false;
        for (unsigned spins = 0; spins < SPIN_MAX; ++spins) {
            if (__hip_atomic_load(&sync_ws[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
            if (__hip_atomic_load(&sync_ws[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // someone gave up
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) {
            __hip_atomic_store(&sync_ws[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // this launch: everybody out
            __hip_atomic_store(&sync_ws[48], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sticky: what the host reads
        }
        *s_fail = ok ? 0 : 1;
    }
    __syncthreads();
    return *s_fail == 0;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int UPB>
__global__ void __launch_bounds__(UPB * LPU)
k_gru_seq_fwd(const float* __restrict__ gi, const float* __restrict__ h0, int64_t ld_h0,
              const uint8_t* __restrict__ masks, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
              float* out, int64_t ldo, float* __restrict__ state_out, int64_t ld_so, int T, int N,
              float* __restrict__ save_r, float* __restrict__ save_z, float* __restrict__ save_n,
              float* __restrict__ save_ghn, unsigned* sync_ws) {
    extern __shared__ __attribute__((aligned(16))) float s_h[];   // N x 512: h_{t-1} * mask_t, then one flag word
    int* s_fail_p = reinterpret_cast<int*>(s_h + N * HH);         // (no static LDS: the dynamic base stays 16-B aligned)
    if (sync_ws[49] != 0 && blockIdx.x == 1) return;  // test hook (word 49): this workgroup "is not resident" - the others' waits time out
    constexpr int NT = UPB * LPU, NWG = HH / UPB;
    const int tid = threadIdx.x, u = tid / LPU, l = tid % LPU;
    const int j = blockIdx.x * UPB + u;
    // the unit's three W_hh rows, lane slice (k = 4 l + 128 i), resident for the whole sequence
    float4 w[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[g][i] = *reinterpret_cast<const float4*>(w_hh + ((int64_t)g * HH + j) * HH + l * 4 + 128 * i);
    float bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = b_hh[g * HH + j];
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        out, 0, (int)((int64_t)T * N * ldo * sizeof(float)), 0x00020000);

    for (int t = 0; t < T; ++t) {
        const int64_t r0 = (int64_t)t * N;
        SEQ_STAMP(t, 0);
        // ---- stage h_{t-1} * mask_t into LDS (t == 0: the caller's h0, written before this launch) ----
        for (int e0 = 0; e0 < N * (HH / 4); e0 += NT * 4) {   // up to four 16-byte loads per thread in flight
            float4 v[4];
            const int last = N * (HH / 4) - 1;
            if (t == 0) {   // (uniform branch; the loads themselves are unconditional on clamped indices, so that the
                            // compiler issues all four before the first wait)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = min(e0 + q * NT + tid, last);
                    v[q] = *reinterpret_cast<const float4*>(h0 + (int64_t)(e / (HH / 4)) * ld_h0 + (e % (HH / 4)) * 4);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = min(e0 + q * NT + tid, last);
                    v[q] = ld_pub16(rs_out, (unsigned)(((r0 - N + e / (HH / 4)) * ldo + (e % (HH / 4)) * 4) * 4));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT + tid;
                if (e <= last) {
                    const float mk = masks[r0 + e / (HH / 4)] ? 1.f : 0.f;
                    v[q].x *= mk, v[q].y *= mk, v[q].z *= mk, v[q].w *= mk;
                    *reinterpret_cast<float4*>(s_h + e * 4) = v[q];
                }
            }
        }
        __syncthreads();
        SEQ_STAMP(t, 1);
        float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_g = 0.f, d_h = 0.f;
        for (int nb = 0; nb < N; nb += 8) {
            // lane 16 + q of the group finishes row nb + q of the unit (the DPP reduction leaves the totals in the upper
            // half of the group): fetch its gate inputs under the matvec
            const int row = nb + l - 16;
            const bool mine = l >= 16 && l < 24 && row < N;
            float gin[3] = {0.f, 0.f, 0.f};
            if (mine) {
#pragma unroll
                for (int g = 0; g < 3; ++g) gin[g] = gi[(r0 + row) * 3 * HH + g * HH + j];
            }
            float acc[8][3];
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                acc[n][0] = acc[n][1] = acc[n][2] = 0.f;
                if (nb + n < N) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 hv = *reinterpret_cast<const float4*>(s_h + (nb + n) * HH + l * 4 + 128 * i);
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            acc[n][g] = fmaf(w[g][i].x, hv.x, acc[n][g]);
                            acc[n][g] = fmaf(w[g][i].y, hv.y, acc[n][g]);
                            acc[n][g] = fmaf(w[g][i].z, hv.z, acc[n][g]);
                            acc[n][g] = fmaf(w[g][i].w, hv.w, acc[n][g]);
                        }
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[n][g] = group_sum_hi(acc[n][g]);
            SEQ_STAMP(t, 2);
            // lane 16 + q takes row nb + q (a select chain, not a dynamic register index)
            float ah[3] = {acc[0][0], acc[0][1], acc[0][2]};
#pragma unroll
            for (int n = 1; n < 8; ++n)
                if (l == 16 + n) { ah[0] = acc[n][0]; ah[1] = acc[n][1]; ah[2] = acc[n][2]; }
            if (mine) {
                const float gh0 = ah[0] + bh[0], gh1 = ah[1] + bh[1], gh2 = ah[2] + bh[2];
                const float hp = s_h[row * HH + j];
                const float rg = sigmoidf_(gin[0] + gh0);
                const float zg = sigmoidf_(gin[1] + gh1);
                const float ng = tanhf(gin[2] + rg * gh2);
                const float hn = (1.f - zg) * ng + zg * hp;
                st_pub(out + (r0 + row) * ldo + j, hn);   // the only store the other workgroups wait for
                if (N <= 8) {   // one row block: everything else is stored after the arrival (nobody waits for it)
                    d_r = rg, d_z = zg, d_n = ng, d_g = gh2, d_h = hn;
                } else {
                    if (state_out && t == T - 1) state_out[(int64_t)row * ld_so + j] = hn;
                    if (save_r) {
                        const int64_t o = (r0 + row) * HH + j;
                        save_r[o] = rg, save_z[o] = zg, save_n[o] = ng, save_ghn[o] = gh2;
                    }
                }
            }
        }
        SEQ_STAMP(t, 3);
        if (t + 1 < T) grid_arrive(sync_ws);
        SEQ_STAMP(t, 4);
        if (N <= 8) {
            const int row = l - 16;
            if (l >= 16 && l < 24 && row < N) {
                if (state_out && t == T - 1) state_out[(int64_t)row * ld_so + j] = d_h;
                if (save_r) {
                    const int64_t o = (r0 + row) * HH + j;
                    save_r[o] = d_r, save_z[o] = d_z, save_n[o] = d_n, save_ghn[o] = d_g;
                }
            }
        }
        if (t + 1 < T && !grid_wait(sync_ws, (unsigned)NWG * (unsigned)(t + 1), s_fail_p)) return;
        SEQ_STAMP(t, 5);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT): element part of step T-1, then for t = T-1 .. 1 the carry of step t + the element part of t-1
// ---------------------------------------------------------------------------------------------------------------
struct ElemIn {
    float dout, h, r, z, n, ghn;
    bool mask;
};

struct ElemOut {   // what only later kernels read: stored after the arrival
    float dr, dz, dn, hp;
};

__device__ __forceinline__ ElemOut bwd_elem(const ElemIn& e, float carry, int64_t row, int j, float* dgh, float& dhz) {
    const float dh = e.dout + carry;
    const float hp = e.mask ? e.h : 0.f;
    const float dn = dh * (1.f - e.z);
    const float dz = dh * (hp - e.n);
    const float dn_pre = dn * (1.f - e.n * e.n);
    const float dz_pre = dz * e.z * (1.f - e.z);
    const float dr_pre = dn_pre * e.ghn * e.r * (1.f - e.r);
    const int64_t o = row * 3 * HH + j;
    st_pub(dgh + o, dr_pre);               // dgh rows are what the next step's matvec reads on every workgroup
    st_pub(dgh + o + HH, dz_pre);
    st_pub(dgh + o + 2 * HH, dn_pre * e.r);
    dhz = dh * e.z;
    return {dr_pre, dz_pre, dn_pre, hp};
}
__device__ __forceinline__ void bwd_store(const ElemOut& v, int64_t row, int j, float* __restrict__ dgi,
                                          float* __restrict__ hp_out) {
    const int64_t o = row * 3 * HH + j;
    dgi[o] = v.dr;
    dgi[o + HH] = v.dz;
    dgi[o + 2 * HH] = v.dn;
    hp_out[row * HH + j] = v.hp;
}

template <int UPB>
__global__ void __launch_bounds__(UPB * LPU)
k_gru_seq_bwd(const float* __restrict__ d_out, int64_t ld_dout, const float* __restrict__ r, const float* __restrict__ z,
              const float* __restrict__ n, const float* __restrict__ ghn, const float* __restrict__ out, int64_t ld_out,
              const float* __restrict__ h0, int64_t ld_h0, const uint8_t* __restrict__ masks,
              const float* __restrict__ whh_t, int T, int N, float* __restrict__ dgi, float* dgh,
              float* __restrict__ hp, unsigned* sync_ws) {
    extern __shared__ __attribute__((aligned(16))) float s_g[];   // N x 1536: dgh_t, then one flag word
    int* s_fail_p = reinterpret_cast<int*>(s_g + N * 3 * HH);
    if (sync_ws[49] != 0 && blockIdx.x == 1) return;  // test hook (word 49): this workgroup "is not resident" - the others' waits time out
    constexpr int K = 3 * HH, NT = UPB * LPU, NWG = HH / UPB;
    const int tid = threadIdx.x, u = tid / LPU, l = tid % LPU;
    const int j = blockIdx.x * UPB + u;
    float4 w[12];   // W_hh^T row j, lane slice
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = *reinterpret_cast<const float4*>(whh_t + (int64_t)j * K + l * 4 + 128 * i);
    const int lr = l - 16;               // lane 16 + q of the group finishes row q of the unit (N <= 16)
    const bool mine = lr >= 0 && lr < N;
    const __amdgpu_buffer_rsrc_t rs_dgh = __builtin_amdgcn_make_buffer_rsrc(
        dgh, 0, (int)((int64_t)T * N * K * sizeof(float)), 0x00020000);
    auto load_elem = [&](int t) {   // inputs of the element part of step t for (row lr, unit j)
        ElemIn e;
        const int64_t row = (int64_t)t * N + lr, idx = row * HH + j;
        e.dout = d_out[row * ld_dout + j];
        e.h = t == 0 ? h0[(int64_t)lr * ld_h0 + j] : out[(row - N) * ld_out + j];
        e.r = r[idx], e.z = z[idx], e.n = n[idx], e.ghn = ghn[idx];
        e.mask = masks[row] != 0;
        return e;
    };
    float dhz = 0.f;
    ElemOut pend = {0.f, 0.f, 0.f, 0.f};
    if (mine) pend = bwd_elem(load_elem(T - 1), 0.f, (int64_t)(T - 1) * N + lr, j, dgh, dhz);
    unsigned epoch = 0;
    for (int t = T - 1; t > 0; --t) {
        grid_arrive(sync_ws);
        ElemIn e;
        bool mask_t = false;
        if (mine) {   // nothing here depends on the exchange: store / fetch it while the others arrive
            bwd_store(pend, (int64_t)t * N + lr, j, dgi, hp);
            e = load_elem(t - 1);
            mask_t = masks[(int64_t)t * N + lr] != 0;
        }
        if (!grid_wait(sync_ws, (unsigned)NWG * ++epoch, s_fail_p)) return;
        // ---- stage dgh_t (N x 1536) into LDS: 16-byte sc1 loads, six per thread in flight at N = 8 ----
        const unsigned src_off = (unsigned)((int64_t)t * N * K * 4);
        for (int e0 = 0; e0 < N * (K / 4); e0 += NT * 6) {
            float4 v[6];
            const int last = N * (K / 4) - 1;
#pragma unroll
            for (int q = 0; q < 6; ++q)   // unconditional loads on clamped indices: all six in flight before the first wait
                v[q] = ld_pub16(rs_dgh, src_off + (unsigned)min(e0 + q * NT + tid, last) * 16u);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int e = e0 + q * NT + tid;
                if (e <= last) *reinterpret_cast<float4*>(s_g + e * 4) = v[q];
            }
        }
        __syncthreads();
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[q] = 0.f;
            if (q < N) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const float4 xv = *reinterpret_cast<const float4*>(s_g + q * K + l * 4 + 128 * i);
                    a0 = fmaf(w[i].x, xv.x, a0);
                    a1 = fmaf(w[i].y, xv.y, a1);
                    a0 = fmaf(w[i].z, xv.z, a0);
                    a1 = fmaf(w[i].w, xv.w, a1);
                }
                acc[q] = a0 + a1;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = group_sum_hi(acc[q]);
        float v = acc[0];
#pragma unroll
        for (int q = 1; q < 16; ++q)
            if (lr == q) v = acc[q];
        if (mine) {
            v = mask_t ? v + dhz : 0.f;   // dh carried into step t-1
            pend = bwd_elem(e, v, (int64_t)(t - 1) * N + lr, j, dgh, dhz);
        }
        __syncthreads();   // s_g is rewritten next step
    }
    if (mine) bwd_store(pend, (int64_t)lr, j, dgi, hp);   // step 0
}

}  // namespace

// hidden units per workgroup: 8 (64 workgroups x 256 threads, one wave per SIMD; 16 per workgroup measured slower, round 3)
static int units_per_wg() { return 8; }


// The persistent kernels spin on counters that every workgroup of the grid feeds: the whole grid has to be resident at
// once.  On a partitioned device (CPX: 32 CUs), a CU-masked stream or a large N (up to 128 KB of LDS per workgroup)
// it may not be - then the caller runs the per-step launches (IVLN_E_UNSUPPORTED), instead of every grid_wait spinning
// to its bound.  Answer cached per (kernel, LDS bytes).
static bool grid_fits(const void* fn, int threads, size_t lds, int grid) {
    return ivln_resident_blocks(fn, threads, lds) >= grid;  // (csrc/residency.h: cached per device, kernel and LDS bytes)
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

/* 1 when ivln_cma_seq_fwd_f32 / _bwd_f32 take the single-launch path for this shape (given a sync_ws). */
int ivln_cma_seq_persistent_ok(int N, int H, int backward) {
    return H == HH && N >= 1 && N <= (backward ? 16 : 64);
}

/* Zeroes a sync workspace (256 bytes), including the sticky error word that the launches never clear. */
int ivln_seq_sync_init(void* sync_ws, void* stream) {
    if (!sync_ws) return IVLN_E_INVALID;
    return hipMemsetAsync(sync_ws, 0, 256, (hipStream_t)stream) == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_gru_seq_fwd_persistent(const float* gi, const float* h0, int64_t ld_h0, const uint8_t* masks, const float* w_hh,
                                const float* b_hh, float* out, int64_t ldo, float* state_out, int64_t ld_so, int T, int N,
                                float* save_r, float* save_z, float* save_n, float* save_ghn, void* sync_ws,
                                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((int64_t)T * N * ldo * (int64_t)sizeof(float) >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;  // 32-bit buffer offsets
    // the exchange reads `out` / h0 with 16-byte buffer loads
    if (!aligned16(out) || !aligned16(h0) || (ldo & 3) || (ld_h0 & 3)) return IVLN_E_UNSUPPORTED;
    const size_t lds = (size_t)N * HH * sizeof(float) + 16;
    static bool attr_set = false;
    if (!attr_set) {
        const int max_lds = 64 * HH * (int)sizeof(float) + 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_seq_fwd<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_seq_fwd<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return IVLN_E_HIP;
        attr_set = true;
    }
    if (units_per_wg() == 8 ? !grid_fits(reinterpret_cast<const void*>(k_gru_seq_fwd<8>), 8 * LPU, lds, HH / 8)
                            : !grid_fits(reinterpret_cast<const void*>(k_gru_seq_fwd<16>), 16 * LPU, lds, HH / 16))
        return IVLN_E_UNSUPPORTED;
    if (hipMemsetAsync(sync_ws, 0, 192, s) != hipSuccess) return IVLN_E_HIP;
    if (units_per_wg() == 8)
        hipLaunchKernelGGL(k_gru_seq_fwd<8>, dim3(HH / 8), dim3(8 * LPU), lds, s, gi, h0, ld_h0, masks, w_hh, b_hh, out,
                           ldo, state_out, ld_so, T, N, save_r, save_z, save_n, save_ghn, (unsigned*)sync_ws);
    else
        hipLaunchKernelGGL(k_gru_seq_fwd<16>, dim3(HH / 16), dim3(16 * LPU), lds, s, gi, h0, ld_h0, masks, w_hh, b_hh,
                           out, ldo, state_out, ld_so, T, N, save_r, save_z, save_n, save_ghn, (unsigned*)sync_ws);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

int ivln_gru_seq_bwd_persistent(const float* d_out, int64_t ld_dout, const float* r, const float* z, const float* n,
                                const float* ghn, const float* out, int64_t ld_out, const float* h0, int64_t ld_h0,
                                const uint8_t* masks, const float* whh_t, int T, int N, float* dgi, float* dgh, float* hp,
                                void* sync_ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if ((int64_t)T * N * 3 * HH * (int64_t)sizeof(float) >= (int64_t)1 << 31) return IVLN_E_UNSUPPORTED;
    if (!aligned16(dgh)) return IVLN_E_UNSUPPORTED;  // (16-byte buffer loads of the exchanged dgh rows)
    const size_t lds = (size_t)N * 3 * HH * sizeof(float) + 16;
    static bool attr_set = false;
    if (!attr_set) {
        const int max_lds = 16 * 3 * HH * (int)sizeof(float) + 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_seq_bwd<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_seq_bwd<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return IVLN_E_HIP;
        attr_set = true;
    }
    if (units_per_wg() == 8 ? !grid_fits(reinterpret_cast<const void*>(k_gru_seq_bwd<8>), 8 * LPU, lds, HH / 8)
                            : !grid_fits(reinterpret_cast<const void*>(k_gru_seq_bwd<16>), 16 * LPU, lds, HH / 16))
        return IVLN_E_UNSUPPORTED;
    if (hipMemsetAsync(sync_ws, 0, 192, s) != hipSuccess) return IVLN_E_HIP;
    if (units_per_wg() == 8)
        hipLaunchKernelGGL(k_gru_seq_bwd<8>, dim3(HH / 8), dim3(8 * LPU), lds, s, d_out, ld_dout, r, z, n, ghn, out, ld_out,
                           h0, ld_h0, masks, whh_t, T, N, dgi, dgh, hp, (unsigned*)sync_ws);
    else
        hipLaunchKernelGGL(k_gru_seq_bwd<16>, dim3(HH / 16), dim3(16 * LPU), lds, s, d_out, ld_dout, r, z, n, ghn, out,
                           ld_out, h0, ld_h0, masks, whh_t, T, N, dgi, dgh, hp, (unsigned*)sync_ws);
    return hipGetLastError() == hipSuccess ? IVLN_OK : IVLN_E_HIP;
}

#ifdef GRU_SEQ_TIMING
int ivln_gru_seq_stamps(void* host, int bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_seq_stamp), bytes) == hipSuccess ? 0 : -1;
}
#endif

/* Synchronises `stream` and reads the error word of a sync workspace: IVLN_OK, or IVLN_E_HIP when a spin of the
 * last persistent launch timed out (its outputs are then undefined). */
/* (word 49 of a sync workspace: test hook, see the kernels; cleared by ivln_seq_sync_init like everything else) */
int ivln_seq_sync_status(const void* sync_ws, void* stream) {
    unsigned err = 0;
    if (hipMemcpyAsync(&err, (const unsigned*)sync_ws + 48, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream) !=
        hipSuccess)
        return IVLN_E_HIP;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return IVLN_E_HIP;
    return err ? IVLN_E_HIP : IVLN_OK;
}

}  // extern "C"
