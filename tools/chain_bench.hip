// What would a PERSISTENT depth-ResNet tail cost per layer?  ("layers 3-4 as one persistent launch".)
// Emulates the data flow of the k_gn_conv chain without its arithmetic details: G workgroups; per layer every workgroup
// reads RB bytes that 16 OTHER workgroups wrote in the layer before (the 16 partial slabs of its channels), spends a
// fixed amount of FMA work, and writes WB bytes (its slice of the next slabs).  Three ways to run L such layers:
//
//   graph       L launches of one kernel captured in a hipGraph (plain loads / stores; the kernel boundary orders them)
//               = what the product does today
//   persistent  ONE launch; slabs stored and loaded write-through (sc1: they were written by other XCDs, whose L2 this
//               XCD never sees), a counter barrier without fences between layers (tools/barrier_bench.hip, "nofence")
//   fenced      ONE launch; plain stores / loads, a counter barrier with release / acquire fences on both sides
//
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/chain_bench tools/chain_bench.hip
//   run:    tools/chain_bench         (microseconds per layer for G = 64 / 256 / 512 and two payload sizes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 256;
constexpr unsigned SPIN_MAX = 1u << 22;
typedef int v4i __attribute__((ext_vector_type(4)));

struct Sync {
    unsigned counter;
    unsigned pad[31];
    unsigned err;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// one layer of one workgroup.  COHERENT: sc1 loads / stores (aux bit 4 = sc1 on gfx942 / gfx950)
template <bool COHERENT>
__device__ __forceinline__ void layer(const float* src, float* dst, int wg, int G, int rb_f4, int wb_f4, int work, unsigned total_bytes) {
    const int t = threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int per_prod = rb_f4 / 16;  // float4 read from each of 16 producers
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, total_bytes), rd = make_rsrc(dst, total_bytes);
    for (int j = 0; j < 16; ++j) {
        const int prod = (wg + j * (G / 16) + 1) % G;
        for (int i = t; i < per_prod; i += NT) {
            const unsigned off = ((unsigned)prod * wb_f4 + (unsigned)(i % wb_f4)) * 16u;
            float4 v;
            if constexpr (COHERENT) {
                const v4i x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
                v = make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
            } else {
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src) + off);
            }
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
    }
    for (int k = 0; k < work; ++k) {  // stand-in for the statistics + MFMA phase
        acc.x = fmaf(acc.x, 1.0001f, acc.y), acc.y = fmaf(acc.y, 0.9999f, acc.z);
        acc.z = fmaf(acc.z, 1.0001f, acc.w), acc.w = fmaf(acc.w, 0.9999f, acc.x);
    }
    for (int i = t; i < wb_f4; i += NT) {
        const unsigned off = ((unsigned)wg * wb_f4 + (unsigned)i) * 16u;
        const float4 v = make_float4(acc.x * 1e-3f, acc.y * 1e-3f, acc.z * 1e-3f, acc.w * 1e-3f);
        if constexpr (COHERENT) {
            v4i x = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(x, rd, (int)off, 0, 16);
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(dst) + off) = v;
        }
    }
}

__global__ __launch_bounds__(NT) void k_layer(const float* src, float* dst, int G, int rb_f4, int wb_f4, int work, unsigned total_bytes) {
    layer<false>(src, dst, blockIdx.x, G, rb_f4, wb_f4, work, total_bytes);
}

template <bool COHERENT>
__global__ __launch_bounds__(NT) void k_chain(float* a, float* b, int G, int L, int rb_f4, int wb_f4, int work, unsigned total_bytes, Sync* sy,
                                              unsigned base) {
    for (int l = 0; l < L; ++l) {
        layer<COHERENT>((l & 1) ? b : a, (l & 1) ? a : b, blockIdx.x, G, rb_f4, wb_f4, work, total_bytes);
        // ---- grid barrier ----
        if constexpr (!COHERENT) __threadfence();
        __syncthreads();  // (every thread's stores issued; thread 0 waits for the whole block's below)
        if (threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the write-through stores have reached memory
            __hip_atomic_fetch_add(&sy->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = base + (unsigned)(l + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(&sy->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) {
                    __hip_atomic_store(&sy->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if constexpr (!COHERENT) __threadfence();
    }
}

int main() {
    const int L = 27, REPS = 30;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    Sync* sy;
    CHECK(hipMalloc(&sy, sizeof(Sync)));
    CHECK(hipMemset(sy, 0, sizeof(Sync)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("# tools/chain_bench: %d dependent layers, microseconds per layer (MI355X)\n", L);
    printf("# %5s %9s %9s %6s | %9s %11s %9s\n", "G", "read/WG", "write/WG", "work", "graph", "persistent", "fenced");
    const int Gs[] = {64, 256, 512};
    const int payload[][2] = {{32 << 10, 8 << 10}, {128 << 10, 32 << 10}};  // {read, write} bytes per workgroup and layer
    const int works[] = {0, 100};
    for (int G : Gs)
        for (auto& pl : payload)
            for (int work : works) {
                const int rb_f4 = pl[0] / 16, wb_f4 = pl[1] / 16;
                const size_t bytes = (size_t)G * pl[1];
                float *a, *b;
                CHECK(hipMalloc(&a, bytes));
                CHECK(hipMalloc(&b, bytes));
                CHECK(hipMemset(a, 0, bytes));
                CHECK(hipMemset(b, 0, bytes));
                // ---- graph of L launches ----
                hipGraph_t g;
                hipGraphExec_t ge;
                CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int l = 0; l < L; ++l)
                    hipLaunchKernelGGL(k_layer, dim3(G), dim3(NT), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, G, rb_f4, wb_f4, work, (unsigned)bytes);
                CHECK(hipStreamEndCapture(s, &g));
                CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                float ms_graph = 0, ms_pers = 0, ms_fence = 0;
                for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(ge, s));
                CHECK(hipEventRecord(e0, s));
                for (int r = 0; r < REPS; ++r) CHECK(hipGraphLaunch(ge, s));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
                // ---- persistent forms (G <= resident workgroups: 256 CUs x >= 2) ----
                unsigned base = 0;
                auto run_chain = [&](bool coherent, float* ms) {
                    for (int r = 0; r < REPS + 3; ++r) {
                        if (r == 3) CHECK(hipEventRecord(e0, s));
                        if (coherent)
                            hipLaunchKernelGGL(k_chain<true>, dim3(G), dim3(NT), 0, s, a, b, G, L, rb_f4, wb_f4, work, (unsigned)bytes, sy, base);
                        else
                            hipLaunchKernelGGL(k_chain<false>, dim3(G), dim3(NT), 0, s, a, b, G, L, rb_f4, wb_f4, work, (unsigned)bytes, sy, base);
                        base += (unsigned)L * (unsigned)G;
                    }
                    CHECK(hipEventRecord(e1, s));
                    CHECK(hipStreamSynchronize(s));
                    CHECK(hipEventElapsedTime(ms, e0, e1));
                };
                CHECK(hipMemset(sy, 0, sizeof(Sync)));
                run_chain(true, &ms_pers);
                run_chain(false, &ms_fence);
                Sync h;
                CHECK(hipMemcpy(&h, sy, sizeof(h), hipMemcpyDeviceToHost));
                printf("  %5d %8dK %8dK %6d | %9.2f %11.2f %9.2f%s\n", G, pl[0] >> 10, pl[1] >> 10, work, 1e3 * ms_graph / REPS / L,
                       1e3 * ms_pers / REPS / L, 1e3 * ms_fence / REPS / L, h.err ? "  (a spin timed out)" : "");
                CHECK(hipGraphExecDestroy(ge));
                CHECK(hipGraphDestroy(g));
                CHECK(hipFree(a));
                CHECK(hipFree(b));
            }
    return 0;
}
