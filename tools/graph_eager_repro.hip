// Does replaying a hipGraph on a stream make LATER eager launches on that stream slower?  
// Round 3 saw the DAgger update (about 240 eager launches, 12.4 ms) take 17.5 ms on a stream that had replayed the
// collection graphs, and worked around it with a stream that never launches graphs (ops.eager_work_stream).  This is
// the smallest program that asks the same question without torch:
//
//   1. time a fixed eager sequence (N dependent launches of a ~5 us kernel) on a fresh stream            -> "fresh"
//   2. capture a graph of 50 such kernels on that stream, replay it R times
//   3. time the same eager sequence on the SAME stream                                                   -> "same stream after replays"
//   4. ... on a stream created BEFORE the replays that never launched a graph                            -> "other stream"
//   5. ... on a stream created AFTER the replays                                                         -> "new stream"
//   6. ... on the same stream again after hipGraphExecDestroy + hipDeviceSynchronize                     -> "same stream, graph destroyed"
//
// Both the wall time of the whole sequence (host enqueue + GPU) and the GPU span between two events are printed per launch.
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/graph_eager_repro tools/graph_eager_repro.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_work(float* p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0001f, 0.5f);
    if (i < n) p[i] = v;
}

struct Timing { double wall_us, gpu_us; };

static Timing eager_sequence(hipStream_t s, float* buf, int n, int launches, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Timing best = {1e30, 1e30};
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(hipEventRecord(e0, s));
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, iters);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        const auto t1 = std::chrono::steady_clock::now();
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wall = std::chrono::duration<double, std::micro>(t1 - t0).count() / launches;
        if (wall < best.wall_us) best.wall_us = wall;
        if (1e3 * ms / launches < best.gpu_us) best.gpu_us = 1e3 * ms / launches;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv) {
    const int n = 256 * 256, launches = 400, iters = argc > 1 ? atoi(argv[1]) : 600, replays = 200;
    float* buf;
    CHECK(hipMalloc(&buf, n * sizeof(float)));
    CHECK(hipMemset(buf, 0, n * sizeof(float)));
    hipStream_t s1, s_other, s_new;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s_other, hipStreamNonBlocking));
    printf("# tools/graph_eager_repro: %d dependent eager launches of one kernel (%d blocks, %d fma iterations); best of 5; us per launch\n", launches,
           (n + 255) / 256, iters);
    printf("# %-44s %10s %10s\n", "where", "wall", "gpu span");
    auto show = [](const char* name, Timing t) { printf("  %-44s %10.2f %10.2f\n", name, t.wall_us, t.gpu_us); };
    show("fresh stream", eager_sequence(s1, buf, n, launches, iters));
    show("other stream (before any graph)", eager_sequence(s_other, buf, n, launches, iters));
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < 50; ++l) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, s1, buf, n, iters);
    CHECK(hipStreamEndCapture(s1, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipGraphLaunch(ge, s1));
        CHECK(hipEventRecord(e0, s1));
        for (int r = 0; r < replays; ++r) CHECK(hipGraphLaunch(ge, s1));
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipStreamSynchronize(s1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-44s %10s %10.2f\n", "(the same kernels as graph nodes, replayed)", "-", 1e3 * ms / replays / 50);
    }
    show("same stream after replays", eager_sequence(s1, buf, n, launches, iters));
    show("other stream (never launched a graph)", eager_sequence(s_other, buf, n, launches, iters));
    CHECK(hipStreamCreateWithFlags(&s_new, hipStreamNonBlocking));
    show("new stream (created after the replays)", eager_sequence(s_new, buf, n, launches, iters));
    show("null stream", eager_sequence(nullptr, buf, n, launches, iters));
    // alternate: one replay, then the eager sequence, five times (what a DAgger iteration does)
    {
        Timing t = {0, 0};
        for (int it = 0; it < 5; ++it) {
            CHECK(hipGraphLaunch(ge, s1));
            Timing u = eager_sequence(s1, buf, n, launches, iters);
            t.wall_us += u.wall_us / 5;
            t.gpu_us += u.gpu_us / 5;
        }
        show("same stream, alternating replay / eager", t);
    }
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    CHECK(hipDeviceSynchronize());
    show("same stream, graph destroyed", eager_sequence(s1, buf, n, launches, iters));
    return 0;
}
