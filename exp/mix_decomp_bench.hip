// EXPERIMENT (r02): does replacing the three bf16 MFMAs of a split-precision product by ONE f16 MFMA + block-scaled MX MFMAs for the cross
// terms make a GEMM-shaped LOOP faster, with the loop's data movement unchanged?  Upper bound for the f16 + MX-fp8 / MX-fp6 GEMM before
// anyone writes it.  One iteration = one K=64 step of a wave's 64x128 output tile (2 x 4 accumulator tiles), shaped like the library kernel:
// 2 workgroups of 4 waves per CU, 72 KB of LDS each, random operands.
//   MIX 0  bf16x3 : 96 x v_mfma_f32_32x32x16_bf16                       (hi.hi + hi.lo + lo.hi)
//   MIX 1  f16+fp8: 32 x v_mfma_f32_32x32x16_f16 + 16 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)   (hi.hi ; [xh8|xl8].[wl8|wh8] as two K=64 MX products)
//   MIX 2  f16+fp6: the same with e2m3 operands (6 VGPRs per fragment, 24 B per lane)
//   V 0 MFMAs only   V 1 + the fragment reads (LDS)   V 3 + LDS stores of the next stage, 2 barriers, 12 x 16 B global loads per thread
//   V 4 + the VALU split of the staged A operand
// Output: "algorithmic" TFLOP/s = 2 x 64 x 128 x 64 FLOP per wave-iteration / time -- the number to compare across mixes.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/mix_decomp_bench exp/mix_decomp_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pack2(float a, float b) { uint32_t r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ unsigned long long g_clk[512 * 4];

template <int MIX, int V>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ in, const uint4* __restrict__ stream, size_t stream_n, float* __restrict__ out, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 72 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = in[i & 4095];
    __syncthreads();
    // register-resident operands for V0
    uint4 fa[2][2], fb[4][2];  // [tile][piece] 16-bit fragments of one k16 step
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) fa[i][p] = in[(tid + 64 * (i * 2 + p)) & 4095];
    for (int j = 0; j < 4; ++j) for (int p = 0; p < 2; ++p) fb[j][p] = in[(tid + 64 * (4 + j * 2 + p)) & 4095];
    i32x8 ma[2], mb[4];  // MX fragments (K = 64): 32 B (fp8) / 24 B (fp6) per lane
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 8; ++r) ma[i][r] = (int)in[(tid * 3 + i * 17 + r) & 4095].x;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 8; ++r) mb[j][r] = (int)in[(tid * 5 + j * 29 + r) & 4095].y;
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int roff = (lane & 31) * 32 + (lane >> 5) * 16;
    size_t gpos = ((size_t)blockIdx.x * 256 + tid) % (stream_n - 8);
    uint4 st[6], sp[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) sp[q] = make_uint4(1, 2, 3, 4);
    float4 xa = make_float4(tid * 0.001f, 0.5f, -0.25f, 1.5f), xb = xa;
    const int sc = 0x7f7f7f7f;  // E8M0 scale 1.0 in every byte
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned char* buf = lds + ((it * 2 + half) & 1) * 36864;
            if (V >= 3) {
#pragma unroll
                for (int q = 0; q < 6; ++q) st[q] = stream[(gpos + (size_t)q * 4099) % stream_n];
                gpos = (gpos + 24593) % (stream_n - 8);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // two k16 steps per K=32 half
                if (V >= 1) {
                    if (MIX == 0) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) fa[i][p] = *reinterpret_cast<const uint4*>(buf + ks * 18432 + p * 4096 + roff + i * 1024);
#pragma unroll
                            for (int j = 0; j < 4; ++j) fb[j][p] = *reinterpret_cast<const uint4*>(buf + ks * 18432 + 8192 + p * 4096 + roff + j * 1024);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i) fa[i][0] = *reinterpret_cast<const uint4*>(buf + ks * 18432 + roff + i * 1024);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[j][0] = *reinterpret_cast<const uint4*>(buf + ks * 18432 + 8192 + roff + j * 1024);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x16 c = acc[i][j];
                        if (MIX == 0) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][1]), __builtin_bit_cast(bf16x8, fb[j][0]), c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][0]), __builtin_bit_cast(bf16x8, fb[j][1]), c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][0]), __builtin_bit_cast(bf16x8, fb[j][0]), c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[j][0]), c, 0, 0, 0);
                        }
                        acc[i][j] = c;
                    }
                __builtin_amdgcn_sched_barrier(0);  // keep the next step's fragment reads from being hoisted above this step (register pressure)
            }
            if (MIX != 0) {
                // cross terms: this half issues ONE of the two K=64 MX products per accumulator tile (operand set `half`)
                constexpr int FMT = MIX == 1 ? 0 : 2;
                constexpr int NB = MIX == 1 ? 32 : 24;  // bytes per lane and fragment
                if (V >= 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned char* p = buf + 24576 + (lane + 64 * i) * NB;
                        const uint4 q0 = *reinterpret_cast<const uint4*>(p);
                        ma[i][0] = q0.x; ma[i][1] = q0.y; ma[i][2] = q0.z; ma[i][3] = q0.w;
                        if (MIX == 1) {
                            const uint4 q1 = *reinterpret_cast<const uint4*>(p + 16);
                            ma[i][4] = q1.x; ma[i][5] = q1.y; ma[i][6] = q1.z; ma[i][7] = q1.w;
                        } else {
                            const uint2 q1 = *reinterpret_cast<const uint2*>(p + 16);
                            ma[i][4] = q1.x; ma[i][5] = q1.y;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned char* p = buf + 24576 + 4096 + (lane + 64 * j) * NB;
                        const uint4 q0 = *reinterpret_cast<const uint4*>(p);
                        mb[j][0] = q0.x; mb[j][1] = q0.y; mb[j][2] = q0.z; mb[j][3] = q0.w;
                        if (MIX == 1) {
                            const uint4 q1 = *reinterpret_cast<const uint4*>(p + 16);
                            mb[j][4] = q1.x; mb[j][5] = q1.y; mb[j][6] = q1.z; mb[j][7] = q1.w;
                        } else {
                            const uint2 q1 = *reinterpret_cast<const uint2*>(p + 16);
                            mb[j][4] = q1.x; mb[j][5] = q1.y;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[i], mb[j], acc[i][j], FMT, FMT, 0, sc, 0, sc);
            }
            uint2 h0, m0;
            if (V >= 4) {
                const float4 x = make_float4(__uint_as_float(sp[0].x & 0x3fffffff), __uint_as_float(sp[1].y & 0x3fffffff), xa.z, xa.w);
                h0.x = pack2(x.x, x.y); h0.y = pack2(x.z, x.w);
                m0.x = pack2(x.x - __uint_as_float(h0.x << 16), x.y - __uint_as_float(h0.x & 0xffff0000u));
                m0.y = pack2(x.z - __uint_as_float(h0.y << 16), x.w - __uint_as_float(h0.y & 0xffff0000u));
                uint2 h1, m1;
                h1.x = pack2(xb.x, xb.y); h1.y = pack2(xb.z, xb.w);
                m1.x = pack2(xb.x - __uint_as_float(h1.x << 16), xb.y - __uint_as_float(h1.x & 0xffff0000u));
                m1.y = pack2(xb.z - __uint_as_float(h1.y << 16), xb.w - __uint_as_float(h1.y & 0xffff0000u));
                xa.x += __uint_as_float(m1.x << 16); xb.y += __uint_as_float(m0.y << 16);
            } else {
                h0 = make_uint2(it, tid); m0 = h0;
            }
            if (V >= 3) {
                unsigned char* wb = lds + ((it * 2 + half + 1) & 1) * 36864;
                *reinterpret_cast<uint2*>(wb + tid * 8) = h0;
                *reinterpret_cast<uint2*>(wb + 4096 + tid * 8) = m0;
                *reinterpret_cast<uint2*>(wb + 2048 + tid * 8) = h0;
                *reinterpret_cast<uint2*>(wb + 6144 + tid * 8) = m0;
                *reinterpret_cast<uint4*>(wb + 12288 + tid * 16) = sp[2];
                *reinterpret_cast<uint4*>(wb + 16384 + tid * 16) = sp[3];
                *reinterpret_cast<uint4*>(wb + 20480 + tid * 16) = sp[4];
                *reinterpret_cast<uint4*>(wb + 24576 + tid * 16) = sp[5];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 6; ++q) sp[q] = st[q];
            }
        }
    }
    float s = xa.x + xb.y + __uint_as_float(sp[0].x & 0x3fffffff);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { g_clk[blockIdx.x * 4] = c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
}

template <int MIX, int V>
double run(const uint4* din, const uint4* dstream, size_t n, float* dout) {
    const int iters = 6000, blocks = 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MIX, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float last = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MIX, V>), dim3(blocks), dim3(256), 73728, 0, din, dstream, n, dout, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&last, e0, e1));
    }
    const double alg = (double)blocks * 4 * iters * 2.0 * 64 * 128 * 64;
    unsigned long long hc[512 * 4];
    CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
    double cs = 0, rs = 0;
    for (int w = 0; w < blocks; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
    const double ghz = cs / rs * 0.1;
    const char* names[3] = {"bf16x3 ", "f16+fp8", "f16+fp6"};
    const double mfma_cycles = MIX == 0 ? 96 * 32.0 : (MIX == 1 ? 32 * 32.0 + 16 * 64.0 : 32 * 32.0 + 16 * 32.0);
    const double busy = (double)blocks * 4 * iters * mfma_cycles / (256.0 * 4) / (ghz * 1e9) / (last * 1e-3);
    printf("%s V%d: %7.1f ms  %6.0f TFLOP/s algorithmic   clock %.2f GHz   MFMA pipe busy %3.0f %%\n", names[MIX], V, last, alg / last / 1e9, ghz, 100.0 * busy);
    return alg / last / 1e9;
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(3);
    for (auto& x : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); x = u >> 16; }
    uint4 *din, *dstream; float* dout;
    const size_t n = (size_t)1 << 20;  // 16 MiB window: L2 / Infinity-Cache resident, like a GEMM's operand panels
    CK(hipMalloc(&din, h.size() * 2)); CK(hipMalloc(&dout, 512 * 256 * 4)); CK(hipMalloc(&dstream, n * 16));
    CK(hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dstream, 0x3c, n * 16));
    double r[3][4];
    r[0][0] = run<0, 0>(din, dstream, n, dout); r[1][0] = run<1, 0>(din, dstream, n, dout); r[2][0] = run<2, 0>(din, dstream, n, dout);
    r[0][1] = run<0, 1>(din, dstream, n, dout); r[1][1] = run<1, 1>(din, dstream, n, dout); r[2][1] = run<2, 1>(din, dstream, n, dout);
    r[0][2] = run<0, 3>(din, dstream, n, dout); r[1][2] = run<1, 3>(din, dstream, n, dout); r[2][2] = run<2, 3>(din, dstream, n, dout);
    r[0][3] = run<0, 4>(din, dstream, n, dout); r[1][3] = run<1, 4>(din, dstream, n, dout); r[2][3] = run<2, 4>(din, dstream, n, dout);
    const char* v[4] = {"V0", "V1", "V3", "V4"};
    for (int c = 0; c < 4; ++c) printf("%s: f16+fp8 / bf16x3 = %.2fx   f16+fp6 / bf16x3 = %.2fx\n", v[c], r[1][c] / r[0][c], r[2][c] / r[0][c]);
    return 0;
}
