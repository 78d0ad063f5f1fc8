// EXPERIMENT: what one matrix-core instruction costs the power budget, per operand type.  Bare register-operand MFMA loops on random data,
// 2 waves per SIMD, 8 independent accumulators per wave (the GEMM's wave tile), with the in-kernel clock.  Under a power-managed clock the
// sustained instruction rate of a loop is ~ inversely proportional to the energy per instruction, so the table answers: would replacing two
// of the split-bf16 GEMM's three bf16 products by int8 / fp8 / MX-fp8 products (the cross terms only need ~2^-6 relative accuracy) buy time?
// build: hipcc --offload-arch=gfx950 -O3 -o exp/mfma_energy_bench exp/mfma_energy_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_clk[4096 * 4];
#define STAMP0 const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define STAMP1 if (threadIdx.x == 0) { g_clk[blockIdx.x * 4] = c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }

template <int KIND>
__global__ __launch_bounds__(256, 2) void kern(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    uint4 ra[8], rb[8];
    for (int i = 0; i < 8; ++i) {
        ra[i] = in[(threadIdx.x * 16 + i) & 4095];
        rb[i] = in[(threadIdx.x * 16 + 8 + i) & 4095];
    }
    STAMP0
    float s = 0.f;
    if (KIND == 0 || KIND == 1) {  // bf16 / f16 32x32x16
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(t + p) & 3]), __builtin_bit_cast(bf16x8, rb[t & 3]), acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(t + p) & 3]), __builtin_bit_cast(f16x8, rb[t & 3]), acc[t], 0, 0, 0);
                }
        }
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else if (KIND == 2) {  // i8 32x32x32
        i32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, ra[(t + p) & 3]), __builtin_bit_cast(i32x4, rb[t & 3]), acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += (float)acc[t][r];
    } else if (KIND == 3) {  // fp8 (e4m3) 32x32x16, not scaled
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const long a = ((long)ra[(t + p) & 3].y << 32) | ra[(t + p) & 3].x, b = ((long)rb[t & 3].y << 32) | rb[t & 3].x;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, acc[t], 0, 0, 0);
                }
        }
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {  // MX-scaled 32x32x64: KIND 4 = fp8 e4m3 (cbsz = blgp = 0), KIND 5 = fp6 e2m3 (2), KIND 6 = fp4 (4); unit scales (E8M0 127)
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        constexpr int FMT = KIND == 4 ? 0 : (KIND == 5 ? 2 : 4);
        const int sc = 0x7f7f7f7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    i32x8 a, b;
                    const uint4 a0 = ra[(t + p) & 3], a1 = ra[4 + ((t + p) & 3)], b0 = rb[t & 3], b1 = rb[4 + (t & 3)];
                    a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
                    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], FMT, FMT, 0, sc, 0, sc);
                }
        }
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    }
    STAMP1
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    // random bit patterns; bytes masked so no fp8 / fp16 / bf16 NaN or Inf encodings appear (exponent never all ones)
    std::vector<unsigned> h(4096 * 4);
    srand(3);
    for (auto& x : h) x = ((unsigned)rand() ^ ((unsigned)rand() << 15)) & 0xb7b7b7b7u & 0xbbffbbffu;
    uint4* din; float* dout;
    CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, 512 * 256 * 4));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 12000, blocks = 512;
    const char* names[7] = {"bf16 32x32x16", "f16 32x32x16", "i8 32x32x32", "fp8 32x32x16", "MX-fp8 32x32x64", "MX-fp6 32x32x64", "MX-fp4 32x32x64"};
    const double macs[7] = {32. * 32 * 16, 32. * 32 * 16, 32. * 32 * 32, 32. * 32 * 16, 32. * 32 * 64, 32. * 32 * 64, 32. * 32 * 64};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 7; ++which) {
            CK(hipEventRecord(e0, 0));
            switch (which) {
                case 0: hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 1: hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 2: hipLaunchKernelGGL(kern<2>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 3: hipLaunchKernelGGL(kern<3>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 4: hipLaunchKernelGGL(kern<4>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                case 5: hipLaunchKernelGGL(kern<5>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
                default: hipLaunchKernelGGL(kern<6>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); break;
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            static unsigned long long hc[4096 * 4];
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(unsigned long long) * blocks * 4));
            double cs = 0, rs = 0;
            for (int w = 0; w < blocks; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
            const double ninstr = (double)blocks * 4 * iters * 24;          // wave-level MFMA instructions
            const double per_simd_ns = ms * 1e6 / ((double)iters * 24 * 2);  // two waves share a SIMD (512 blocks x 4 waves over 1024 SIMDs)
            printf("%-16s %7.1f ms  clock %.2f GHz  %.1f ns/instr/SIMD = %.1f cycles  %6.0f T-MAC/s x2 = %6.0f TFLOP/s\n", names[which], ms, cs / rs * 0.1,
                   per_simd_ns, per_simd_ns * cs / rs * 0.1, ninstr * macs[which] / ms / 1e9, 2 * ninstr * macs[which] / ms / 1e9);
        }
    return 0;
}
