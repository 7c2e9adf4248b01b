// Minimal reproduction of the hazard behind BF3_STORE_GUARD (csrc/conv_bf3.hip): a VALU write to the DATA registers of a
// 16-byte buffer store right behind the store.  The ISA's rule (a VMEM store of more than 64 bits reads its data a few
// cycles after issue: one wait state before a VALU may overwrite it) is padded for by the compiler ONLY when the store's
// soffset field is not a register (LLVM's GCNHazardRecognizer, createsVALUHazard: "this hazard only exists if the
// instruction is not using a register in the soffset field" - true of the parts that rule was written for).  Round 5 saw
// stale lanes behind exactly such stores (scalar channel offset in an SGPR) in the fused bottleneck tail; this tool asks
// the hardware directly, with the whole sequence in inline assembly so that no compiler pass pads or reorders it:
//
//     v[20:23] <- A;  buffer_store_dwordx4 v[20:23] -> slot 2i    (soffset = SGPR | literal 0)
//     <PAD: nothing | s_nop 0 | s_nop 1 | s_nop 3>
//     v[20:23] <- B;  buffer_store_dwordx4 v[20:23] -> slot 2i+1
//
// and a second kernel counts the slots 2i that do not hold A (per lane of the wave).
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_hazard tools/store_hazard.hip && tools/store_hazard [launches]
// Output: one line per variant - workgroups run, stale 16-byte stores, stale dwords by lane group (0-15 | 16-31 | 32-47 | 48-63).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int ITER = 16;     // store pairs per lane and launch (the fused tail stores 16 registers per tile)
constexpr int THREADS = 256;
constexpr unsigned PAD = 4096;  // bytes in front of slot 0

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(2);                                                             \
        }                                                                        \
    } while (0)

// MODE: 0 = SGPR soffset, no pad | 1 = SGPR soffset, s_nop 0 | 2 = SGPR soffset, s_nop 1 | 3 = SGPR soffset, s_nop 3 (the guard)
//       4 = soffset literal 0 ("off"), no pad (the case the compiler DOES pad in compiled code)
template <int MODE>
__global__ __launch_bounds__(THREADS) void k_store_pairs(unsigned* __restrict__ buf, unsigned bytes, int soff_value, unsigned salt) {
    const unsigned lane_id = blockIdx.x * THREADS + threadIdx.x;
    // raw buffer descriptor: base, stride 0, num_records = bytes, dword-addressed raw buffer (the kernels' bf3_rsrc flags)
    const unsigned long long base = (unsigned long long)buf;
    v4i rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffu));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    rsrc[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane(soff_value);
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        const unsigned slot = (lane_id * ITER + it) * 2u;          // 16-byte slots 2i (first store) and 2i + 1 (second)
        const unsigned a = 0xA0000000u ^ (slot * 4u) ^ salt, b = 0xB0000000u ^ (slot * 4u) ^ salt;
        // (the scalar offset is added back by the store; the buffer's range check sees the vector offset alone, which therefore
        //  must not go negative: the slots start PAD bytes into the buffer)
        const unsigned off1 = PAD + slot * 16u - (unsigned)soff, off2 = off1 + 16u;
#define PAIR(PADTXT, SOFF)                                                                                                 \
    asm volatile(                                                                                                          \
        "v_mov_b32 v20, %[a]\n v_add_u32 v21, 1, %[a]\n v_add_u32 v22, 2, %[a]\n v_add_u32 v23, 3, %[a]\n"                  \
        "s_nop 4\n"                                                                                                        \
        "buffer_store_dwordx4 v[20:23], %[o1], %[rs], " SOFF " offen\n" PADTXT                                             \
        "v_mov_b32 v20, %[b]\n v_add_u32 v21, 1, %[b]\n v_add_u32 v22, 2, %[b]\n v_add_u32 v23, 3, %[b]\n"                  \
        "s_nop 4\n"                                                                                                        \
        "buffer_store_dwordx4 v[20:23], %[o2], %[rs], " SOFF " offen\n"                                                    \
        "s_nop 4\n"                                                                                                        \
        :                                                                                                                  \
        : [a] "v"(a), [b] "v"(b), [o1] "v"(MODE == 4 ? off1 + (unsigned)soff : off1), [o2] "v"(MODE == 4 ? off2 + (unsigned)soff : off2), \
          [rs] "s"(rsrc), [so] "s"(soff)                                                                                   \
        : "v20", "v21", "v22", "v23", "memory")
        if constexpr (MODE == 0) PAIR("", "%[so]");
        else if constexpr (MODE == 1) PAIR("s_nop 0\n", "%[so]");
        else if constexpr (MODE == 2) PAIR("s_nop 1\n", "%[so]");
        else if constexpr (MODE == 3) PAIR("s_nop 3\n", "%[so]");
        else PAIR("", "0");
#undef PAIR
    }
}

// slot 2i must hold A (+0..3), slot 2i + 1 must hold B: count the dwords that do not, by lane of the wave
__global__ __launch_bounds__(THREADS) void k_check(const unsigned* __restrict__ buf, unsigned salt, unsigned long long* __restrict__ bad /* [64 + 2] */) {
    const unsigned lane_id = blockIdx.x * THREADS + threadIdx.x;
    unsigned stale = 0, other = 0, stores = 0;
    for (int it = 0; it < ITER; ++it) {
        const unsigned slot = (lane_id * ITER + it) * 2u;
        const unsigned a = 0xA0000000u ^ (slot * 4u) ^ salt, b = 0xB0000000u ^ (slot * 4u) ^ salt;
        unsigned s_here = 0;
        for (int e = 0; e < 4; ++e) {
            const unsigned v1 = buf[PAD / 4 + slot * 4u + e], v2 = buf[PAD / 4 + slot * 4u + 4u + e];
            if (v1 != a + e) {
                if (v1 == b + e) ++stale, ++s_here;  // the first store wrote what the registers held AFTER the overwrite
                else ++other;
            }
            if (v2 != b + e) ++other;
        }
        stores += s_here ? 1 : 0;
    }
    if (stale) atomicAdd(&bad[threadIdx.x & 63], (unsigned long long)stale);
    if (stores) atomicAdd(&bad[64], (unsigned long long)stores);
    if (other) atomicAdd(&bad[65], (unsigned long long)other);
}

template <int MODE>
void run(const char* name, int blocks, int launches, unsigned* buf, size_t bytes, unsigned long long* bad) {
    CHECK(hipMemset(bad, 0, 66 * sizeof(unsigned long long)));
    for (int l = 0; l < launches; ++l) {
        const unsigned salt = 0x01010101u * (unsigned)(l & 15);
        hipLaunchKernelGGL((k_store_pairs<MODE>), dim3(blocks), dim3(THREADS), 0, 0, buf, (unsigned)bytes, 64 * (1 + (l & 7)), salt);
        hipLaunchKernelGGL(k_check, dim3(blocks), dim3(THREADS), 0, 0, buf, salt, bad);
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h[66];
    CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long g[4] = {0, 0, 0, 0}, tot = 0;
    for (int i = 0; i < 64; ++i) g[i / 16] += h[i], tot += h[i];
    printf("%-46s workgroups %9lld  16-byte stores %12lld  stale stores %8llu  stale dwords %8llu  by lanes 0-15 | 16-31 | 32-47 | 48-63: %llu | %llu | %llu | %llu  other mismatches %llu\n",
           name, (long long)blocks * launches, (long long)blocks * launches * THREADS * ITER, h[64], tot, g[0], g[1], g[2], g[3], h[65]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 256;
    const int blocks = 4096;  // x 256 launches = 1.05 M workgroups per variant
    const size_t bytes = (size_t)blocks * THREADS * ITER * 2 * 16 + PAD;  // 512 MB
    unsigned* buf;
    unsigned long long* bad;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&bad, 66 * sizeof(unsigned long long)));
    CHECK(hipMemset(buf, 0, bytes));
    run<0>("soffset in an SGPR, VALU write right behind", blocks, launches, buf, bytes, bad);
    run<1>("soffset in an SGPR, s_nop 0 (1 wait state)", blocks, launches, buf, bytes, bad);
    run<2>("soffset in an SGPR, s_nop 1 (2 wait states)", blocks, launches, buf, bytes, bad);
    run<3>("soffset in an SGPR, s_nop 3 (BF3_STORE_GUARD)", blocks, launches, buf, bytes, bad);
    run<4>("soffset literal 0, VALU write right behind", blocks, launches, buf, bytes, bad);
    return 0;
}
