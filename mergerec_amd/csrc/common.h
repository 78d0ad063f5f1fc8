// Shared host/device helpers for libmergerec_hip.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/mergerec_hip.h"

#define MR_WAVE 64

namespace mr {

void set_last_hip_error(const char* msg);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error(hipGetErrorString(e));
        return MR_ELAUNCH;
    }
    return MR_OK;
}

// Dynamic-LDS ceiling of one kernel (hipFuncAttributeMaxDynamicSharedMemorySize), raised once per device and per kernel: declare one
// `static mr::DynLdsCeiling` beside the launch of each instantiation.  The return code is checked (a failed call would otherwise surface
// as an opaque launch error, or only on the second device of a process); racing first calls both set the same value.
struct DynLdsCeiling {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> set[kMaxDevices] = {};
    int ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MR_ELAUNCH;
        if (set[dev].load(std::memory_order_acquire) >= bytes) return MR_OK;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            set_last_hip_error(hipGetErrorString(hipGetLastError()));
            return MR_EUNSUPPORTED;
        }
        set[dev].store(bytes, std::memory_order_release);
        return MR_OK;
    }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups that the dispatcher deals to the
// same XCD (id % 8) get a contiguous range of tile ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int pid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace mr
