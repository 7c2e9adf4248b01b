// Device-aware answers the persistent kernels and the tile heuristics need: how many compute units the CURRENT device
// has, and how many workgroups of a kernel can be resident on it at once for a given (threads, dynamic LDS).  Cached per
// (device, kernel, threads, LDS bytes): a second device or partition mode in the same process gets its own answer
// (the first versions cached one number for whichever device was current on the first call).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <tuple>

static inline int ivln_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it == cache.end()) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        it = cache.emplace(dev, cus).first;
    }
    return it->second;
}

// workgroups of `fn` (threads, lds bytes of dynamic LDS) the current device keeps resident together; 0 on error
static inline int ivln_resident_blocks(const void* fn, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, int, size_t>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, fn, threads, lds);
    auto it = cache.find(key);
    if (it == cache.end()) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            per_cu = cus = 0;
        it = cache.emplace(key, per_cu * cus).first;
    }
    return it->second;
}
