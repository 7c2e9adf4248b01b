// Inter-workgroup synchronisation microbenchmark for MI355X (gfx950): what one step of a persistent kernel pays
// to exchange data between workgroups, by form.  Prices the designs of the persistent sequence GRU
// (ivln_cma_seq_fwd/bwd: <= 64 workgroups exchange an (N, 512) hidden state every timestep) and of a persistent
// depth-ResNet tail (256-512 workgroups, a grid barrier between convolutions) against the figures of
// /opt/skills/guides/MI355X_MICROARCH.md ("barrier-counter", "barrier-xcd", "allgather", "handoff-1to1").
//
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/barrier_bench tools/barrier_bench.hip
//   run:    tools/barrier_bench            (prints one line per form and grid size: microseconds per step)
//
// Forms (every spin is bounded; a timed-out spin sets an error word and the run reports it):
//   fence    flat counter barrier: lane 0 release fence -> asm vmcnt(0) -> relaxed agent atomic add -> relaxed sc1 poll
//            with s_sleep -> acquire fence.  Payload may be plain stores / plain loads.
//   nofence  the same counter without fences: legal when every exchanged word is stored AND loaded write-through
//            (relaxed agent-scope atomics = `global_store/load ... sc1`), which is how the GRU publishes h_t.
//   xcd      XCD-hierarchical: per-XCC arrival counter; the last arriver of an XCC release-fences and arrives on the top
//            counter, polls it, acquire-fences and bumps the XCC's generation word; the others poll that word and
//            acquire.  (Membership per XCC comes from a census behind a flat barrier at kernel start.)
//   gather   no barrier at all: the data is the flag.  Every workgroup publishes its slice of a 4096-value vector as
//            8-byte {epoch, value} granules (one sc1 store each) and sweeps all 4096 granules until every tag matches
//            (two granule sets by step parity: a fast producer may be one step ahead of a slow reader, never two).
//   nofence+read  `nofence` followed by an sc1 read of the whole 4096-value vector (the flag-and-payload form of the
//            same exchange).
// Each step's published values depend on what the step before read, so steps cannot overlap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int NT = 256;
constexpr int VEC = 4096;            // values exchanged per step (N = 8 rows x H = 512)
constexpr unsigned SPIN_MAX = 1u << 22;

struct State {
    unsigned counter;        // flat barrier arrivals (monotonic)
    unsigned pad0[31];
    unsigned top;            // xcd form: top-level arrivals
    unsigned pad1[31];
    unsigned xcc_arrive[8 * 32];   // one 128-byte line per XCC
    unsigned xcc_gen[8 * 32];
    unsigned xcc_members[8 * 32];
    unsigned err;
};

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* err) {
    for (unsigned spins = 0; spins < SPIN_MAX; ++spins) {
        if (ld_rlx(p) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    st_rlx(err, 1u);
    return false;
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// flat counter barrier; FENCE: release before the arrival, acquire after the match
template <bool FENCE>
__device__ __forceinline__ void barrier_flat(State* st, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        add_rlx(&st->counter, 1u);
        wait_ge(&st->counter, target, &st->err);
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(State* st, unsigned xcc, unsigned members, unsigned n_xcc, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = add_rlx(&st->xcc_arrive[xcc * 32], 1u) + 1u;
        if (ticket == members * epoch) {          // last arriver of this XCC in this epoch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            add_rlx(&st->top, 1u);
            wait_ge(&st->top, n_xcc * epoch, &st->err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            st_rlx(&st->xcc_gen[xcc * 32], epoch);
        } else {
            wait_ge(&st->xcc_gen[xcc * 32], epoch, &st->err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

// form: 0 fence, 1 nofence, 2 xcd, 3 gather, 4 nofence + read of the vector
template <int FORM>
__global__ void __launch_bounds__(NT) k_steps(State* st, u64* gran, unsigned* vec, int steps, unsigned* out) {
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    unsigned xcc = 0, members = 0, n_xcc = 0, base = 0;
    if (FORM == 2) {   // census behind one flat barrier: who shares my XCC
        xcc = xcc_id();
        if (t == 0) add_rlx(&st->xcc_members[xcc * 32], 1u);
        barrier_flat<true>(st, (unsigned)G);
        base = 1;
        members = ld_rlx(&st->xcc_members[xcc * 32]);
        for (int x = 0; x < 8; ++x) n_xcc += ld_rlx(&st->xcc_members[x * 32]) ? 1u : 0u;
    }
    const int per = VEC / G;               // values this workgroup publishes per step
    unsigned carry = (unsigned)b;
    for (int s = 1; s <= steps; ++s) {
        const unsigned epoch = (unsigned)s;
        if (FORM == 3) {
            for (int i = t; i < per; i += NT)
                __hip_atomic_store(&gran[(s & 1) * VEC + b * per + i], ((u64)epoch << 32) | (u64)(carry + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // sweep: every thread owns VEC / NT granules, re-read until every tag is this epoch
            unsigned acc = 0;
            bool ok = false;
            for (unsigned spins = 0; spins < SPIN_MAX && !ok; ++spins) {
                ok = true;
                acc = 0;
#pragma unroll
                for (int k = 0; k < VEC / NT; ++k) {
                    const u64 x = __hip_atomic_load(&gran[(s & 1) * VEC + k * NT + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok &= (unsigned)(x >> 32) == epoch;
                    acc += (unsigned)x;
                }
                ok = __syncthreads_and(ok);
            }
            if (!ok && t == 0) st_rlx(&st->err, 2u);
            if (!ok) break;
            carry = acc;
        } else {
            if (FORM == 4)
                for (int i = t; i < per; i += NT) st_rlx(&vec[(s & 1) * VEC + b * per + i], carry + i);
            else if (t == 0)
                st_rlx(&vec[(s & 1) * VEC + b], carry);
            if (FORM == 0) barrier_flat<true>(st, (unsigned)G * (base + epoch));
            if (FORM == 1 || FORM == 4) barrier_flat<false>(st, (unsigned)G * (base + epoch));
            if (FORM == 2) barrier_xcd(st, xcc, members, n_xcc, epoch);
            if (FORM == 4) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < VEC / NT; ++k) acc += ld_rlx(&vec[(s & 1) * VEC + k * NT + t]);
                carry = acc;
            } else {
                carry += ld_rlx(&vec[(s & 1) * VEC + (b + 1) % G]);
            }
            if (ld_rlx(&st->err)) break;
        }
    }
    if (t == 0) out[b] = carry;
}

template <int FORM>
static float run(int G, int steps, State* st, u64* gran, unsigned* vec, unsigned* out, unsigned* err) {
    CHECK(hipMemset(st, 0, sizeof(State)));
    CHECK(hipMemset(gran, 0, 2 * VEC * sizeof(u64)));
    CHECK(hipMemset(vec, 0, 2 * VEC * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_steps<FORM>, dim3(G), dim3(NT), 0, 0, st, gran, vec, steps, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    State h;
    CHECK(hipMemcpy(&h, st, sizeof(State), hipMemcpyDeviceToHost));
    *err |= h.err;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1000.f;
}

template <int FORM>
static void sweep(const char* name, const std::vector<int>& grids, State* st, u64* gran, unsigned* vec, unsigned* out) {
    for (int G : grids) {
        if (VEC % G) continue;
        unsigned err = 0;
        const int R = 400;
        run<FORM>(G, 8, st, gran, vec, out, &err);           // warm
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            const float t0 = run<FORM>(G, 0, st, gran, vec, out, &err);
            const float t1 = run<FORM>(G, R, st, gran, vec, out, &err);
            const float per = (t1 - t0) / R;
            if (per < best) best = per;
        }
        printf("%-14s G=%4d  %7.2f us/step%s\n", name, G, best, err ? "   SPIN TIMEOUT" : "");
        fflush(stdout);
    }
}

int main() {
    State* st;
    u64* gran;
    unsigned *vec, *out;
    CHECK(hipMalloc(&st, sizeof(State)));
    CHECK(hipMalloc(&gran, 2 * VEC * sizeof(u64)));
    CHECK(hipMalloc(&vec, 2 * VEC * sizeof(unsigned)));
    CHECK(hipMalloc(&out, 4096 * sizeof(unsigned)));
    const std::vector<int> small = {8, 16, 32, 64}, all = {8, 16, 32, 64, 128, 256, 512};
    sweep<0>("fence", all, st, gran, vec, out);
    sweep<1>("nofence", all, st, gran, vec, out);
    sweep<2>("xcd", all, st, gran, vec, out);
    sweep<3>("gather", small, st, gran, vec, out);
    sweep<4>("nofence+read", small, st, gran, vec, out);
    return 0;
}
