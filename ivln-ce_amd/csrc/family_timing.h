// Launch macro of the MFMA-family kernels (k_gemm / k_gemm_vec / k_conv_direct / k_wgrad_direct / k_gn_conv / k_nconv /
// k_conv_bf3 / k_depth_net) with an optional per-dispatch duration sink.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

// Kernel-duration sink of the MFMA family (ivln_family_timing_begin / _end, include/ivln_hip.h; bench.py's live roofline).
// While armed, the family's launches go out through hipExtLaunchKernelGGL with a start / stop event each - the
// dispatch's own begin and end timestamps, the figure rocprofv3 reports per kernel - instead of being bracketed by
// events recorded around the launch, which add every launch's dispatch latency (~5 us per launch on this box).
bool ivln_family_timing_next(hipEvent_t* start, hipEvent_t* stop, const char* kernel);  // false: not armed, or out of events
// `name` = the kernel's template name as written at the launch site ("(k_conv_direct<3, ...>)" is cut down to
// "k_conv_direct" by the sink); sites that launch through a function pointer name their kernel explicitly.
#define IVLN_LAUNCH_FAMILY_NAMED(name, kernel, grid, block, shmem, stream, ...)                           \
    do {                                                                                                  \
        hipEvent_t ivln_e0_, ivln_e1_;                                                                    \
        if (ivln_family_timing_next(&ivln_e0_, &ivln_e1_, name))                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ivln_e0_, ivln_e1_, 0, __VA_ARGS__); \
        else                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                          \
    } while (0)
#define IVLN_LAUNCH_FAMILY(kernel, grid, block, shmem, stream, ...) \
    IVLN_LAUNCH_FAMILY_NAMED(#kernel, kernel, grid, block, shmem, stream, __VA_ARGS__)
