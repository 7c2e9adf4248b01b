// Issue rate of v_mfma_f32_32x32x16_bf16 streams as the split-bf16 kernels issue them: NACC accumulators visited round-robin,
// groups of NACC instructions sharing their A operand, one or two waves per SIMD.  Prints shader cycles per MFMA (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate tools/mfma_rate.hip && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC, int NA>
__global__ __launch_bounds__(256, 1) void k(const int* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i a[NA], b[NACC];
    for (int i = 0; i < NA; ++i) a[i] = *reinterpret_cast<const v4i*>(src + (threadIdx.x + i * 256) * 4);
    for (int i = 0; i < NACC; ++i) b[i] = *reinterpret_cast<const v4i*>(src + (threadIdx.x + (i + NA) * 256) * 4);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24 / NACC; ++g)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[g % NA]), __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int NA>
void run(const char* name, int blocks, const int* src, float* out, unsigned long long* cyc) {
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NACC, NA>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, NA>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / h.size() / (iters * 24.0);
    printf("%-34s blocks %4d: %6.1f s_memtime ticks per MFMA per wave, kernel %7.1f us -> %5.1f ns per MFMA per wave\n", name, blocks, per, ms * 1e3,
           ms * 1e6 / (iters * 24.0));
}

int main() {
    int* src;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&src, 256 * 4 * 16 * 4);
    hipMemset(src, 0x3c, 256 * 4 * 16 * 4);
    hipMalloc(&out, 2048 * 256 * 4);
    hipMalloc(&cyc, 2048 * 4 * 8);
    for (int blocks : {128, 256, 512}) {
        run<4, 3>("4 accumulators, A shared by 4", blocks, src, out, cyc);
        run<2, 3>("2 accumulators, A shared by 2", blocks, src, out, cyc);
        run<8, 3>("8 accumulators, A shared by 8", blocks, src, out, cyc);
        run<1, 3>("1 accumulator (dependent chain)", blocks, src, out, cyc);
    }
    return 0;
}
