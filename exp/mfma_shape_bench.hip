// EXPERIMENT: sustained bf16 MFMA rate of the two shapes on random operands (power-limited clock), operands in registers,
// 2 waves per SIMD, same accumulator footprint (128 fp32 registers per wave).  build: hipcc --offload-arch=gfx950 -O3 -o exp/mfma_shape_bench exp/mfma_shape_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void k32(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + 4 + i) & 4095]);
    }
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {  // 8 tiles x 3 products of 32x32x16
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t + 1) & 3], b[t & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 3], b[(t + 1) & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 3], b[t & 3], acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void k16(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + 4 + i) & 4095]);
    }
    f32x4 acc[32];
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // same flops per iteration: 16 tiles x 3 products of 16x16x32 = 48 x 16384 flops... x2 below
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(t + 1) & 3], b[t & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t + 1) & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[t & 3], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 16; t < 32; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(t + 1) & 3], b[t & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t + 1) & 3], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[t & 3], acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(3);
    for (auto& x : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    uint4* din; float* dout;
    CK(hipMalloc(&din, h.size() * 2)); CK(hipMalloc(&dout, 512 * 256 * 4));
    CK(hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 20000, blocks = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // flops per wave-iteration: k32: 24 MFMA x 32768; k16: 96 MFMA x 16384
            const double fl = (double)blocks * 4 * iters * (which == 0 ? 24.0 * 32768 : 96.0 * 16384);
            printf("%s: %.1f ms  %.0f TFLOP/s bf16 MFMA\n", which == 0 ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
        }
    return 0;
}
