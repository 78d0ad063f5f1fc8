// EXPERIMENT: what lowers the sustained MFMA rate of the split-bf16 GEMM?  A 32x32x16 bf16 MFMA loop shaped like the library kernel's
// k-tile (24 MFMAs per wave, 2 workgroups of 4 waves per CU, random operands), with the kernel's other activities added one by one:
//   V0 MFMAs only (operands stay in registers)          V1 + the 12 ds_read_b128 fragment reads per k-tile
//   V2 + LDS stores of a staged tile (8 per thread)      V3 + the global loads of a k-tile (6 x 16 B per thread, streaming)
//   V4 + the VALU split work (24 cvt / sub / shift ops)
// build: hipcc --offload-arch=gfx950 -O3 -o exp/power_decomp_bench exp/power_decomp_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pack2(float a, float b) { uint32_t r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ unsigned long long g_clk[512 * 4];
template <int V>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ in, const uint4* __restrict__ stream, size_t stream_n, float* __restrict__ out, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 72 KB like the library kernel
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 72 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = in[i & 4095];
    __syncthreads();
    bf16x8 a[2][2], b[4][2];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) a[i][p] = __builtin_bit_cast(bf16x8, in[(tid + 64 * (i * 2 + p)) & 4095]);
    for (int j = 0; j < 4; ++j) for (int p = 0; p < 2; ++p) b[j][p] = __builtin_bit_cast(bf16x8, in[(tid + 64 * (4 + j * 2 + p)) & 4095]);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int roff = (lane & 31) * 32 + (lane >> 5) * 16;
    size_t gpos = ((size_t)blockIdx.x * 256 + tid) % (stream_n - 8);
    uint4 st[6], sp[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) sp[q] = make_uint4(1, 2, 3, 4);
    float4 fa = make_float4(tid * 0.001f, 0.5f, -0.25f, 1.5f), fb = fa;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* buf = lds + (it & 1) * 36864;
        if (V >= 3) {
#pragma unroll
            for (int q = 0; q < 6; ++q) st[q] = stream[(gpos + (size_t)q * 4099) % stream_n];
            gpos = (gpos + 24593) % (stream_n - 8);
        }
        if (V >= 1) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * 4096 + roff + i * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(buf + 12288 + p * 8192 + roff + j * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        uint2 h0, m0;
        if (V >= 4) {
            const float4 x = (V >= 3) ? make_float4(__uint_as_float(sp[0].x & 0x3fffffff), __uint_as_float(sp[1].y & 0x3fffffff), fa.z, fa.w) : fa;
            h0.x = pack2(x.x, x.y); h0.y = pack2(x.z, x.w);
            m0.x = pack2(x.x - __uint_as_float(h0.x << 16), x.y - __uint_as_float(h0.x & 0xffff0000u));
            m0.y = pack2(x.z - __uint_as_float(h0.y << 16), x.w - __uint_as_float(h0.y & 0xffff0000u));
            uint2 h1, m1;
            h1.x = pack2(fb.x, fb.y); h1.y = pack2(fb.z, fb.w);
            m1.x = pack2(fb.x - __uint_as_float(h1.x << 16), fb.y - __uint_as_float(h1.x & 0xffff0000u));
            m1.y = pack2(fb.z - __uint_as_float(h1.y << 16), fb.w - __uint_as_float(h1.y & 0xffff0000u));
            fa.x += __uint_as_float(m1.x << 16); fb.y += __uint_as_float(m0.y << 16);
        } else {
            h0 = make_uint2(it, tid); m0 = h0;
        }
        if (V >= 2) {
            unsigned char* wb = lds + ((it + 1) & 1) * 36864;
            *reinterpret_cast<uint2*>(wb + tid * 8) = h0;
            *reinterpret_cast<uint2*>(wb + 4096 + tid * 8) = m0;
            *reinterpret_cast<uint2*>(wb + 2048 + tid * 8) = h0;
            *reinterpret_cast<uint2*>(wb + 6144 + tid * 8) = m0;
            const uint4 w = (V >= 3) ? sp[2] : make_uint4(it, tid, 1, 2);
            *reinterpret_cast<uint4*>(wb + 12288 + tid * 16) = w;
            *reinterpret_cast<uint4*>(wb + 16384 + tid * 16) = (V >= 3) ? sp[3] : w;
            *reinterpret_cast<uint4*>(wb + 20480 + tid * 16) = (V >= 3) ? sp[4] : w;
            *reinterpret_cast<uint4*>(wb + 24576 + tid * 16) = (V >= 3) ? sp[5] : w;
            __syncthreads();
        }
        if (V >= 3) {
#pragma unroll
            for (int q = 0; q < 6; ++q) sp[q] = st[q];  // prefetch distance 1: this iteration's loads are consumed in the next one
        }
    }
    float s = fa.x + fb.y + __uint_as_float(sp[0].x & 0x3fffffff);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { g_clk[blockIdx.x * 4] = c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
}

template <int V>
void run(const uint4* din, const uint4* dstream, size_t n, float* dout) {
    const int iters = 12000, blocks = 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, last = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 73728, 0, din, dstream, n, dout, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&last, e0, e1));
        best = last < best ? last : best;
    }
    const double fl = (double)blocks * 4 * iters * 24.0 * 32768;
    unsigned long long hc[512 * 4];
    CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
    double cs = 0, rs = 0;
    for (int w = 0; w < blocks; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
    const double ghz = cs / rs * 0.1;
    printf("V%d: %.1f ms  %.0f TFLOP/s bf16 MFMA   in-kernel clock %.2f GHz   MFMA pipe busy %.0f %% of that clock\n", V, last, fl / last / 1e9, ghz,
           100.0 * (fl / last / 1e9) / (2500.0 * ghz / 2.4));
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(3);
    for (auto& x : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    uint4 *din, *dstream; float* dout;
    const size_t n = (size_t)1 << 26;  // 1 GiB of uint4: streaming source far larger than the caches
    CK(hipMalloc(&din, h.size() * 2)); CK(hipMalloc(&dout, 512 * 256 * 4)); CK(hipMalloc(&dstream, n * 16));
    CK(hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dstream, 0x3c, n * 16));
    run<0>(din, dstream, n, dout); run<1>(din, dstream, n, dout); run<2>(din, dstream, n, dout);
    printf("-- global loads from a 16 MiB window (L2 / Infinity-Cache resident):\n");
    run<3>(din, dstream, (size_t)1 << 20, dout); run<4>(din, dstream, (size_t)1 << 20, dout);
    printf("-- global loads from a 2 MiB window (L2 resident):\n");
    run<3>(din, dstream, (size_t)1 << 17, dout); run<4>(din, dstream, (size_t)1 << 17, dout);
    printf("-- global loads streaming 1 GiB (HBM):\n");
    run<3>(din, dstream, n, dout); run<4>(din, dstream, n, dout);
    return 0;
}
