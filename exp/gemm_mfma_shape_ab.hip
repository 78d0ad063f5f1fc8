// EXPERIMENT (not part of the library): does v_mfma_f32_16x16x32_bf16 beat v_mfma_f32_32x32x16_bf16 INSIDE a split-bf16 (bf16x3) GEMM?
// One kernel, templated on the MFMA shape, everything else identical: 128 x 128 block tile, 4 waves (64 x 64 wave tiles), 32-k LDS stages
// (64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3), A = fp32 split to (hi, mid) at staging, B = pre-split k-blocked pieces,
// register-prefetch double buffering, two workgroups per CU (64 KB LDS each).  A bare register loop gives 2.04 vs 1.81 PFLOP/s
// (exp/mfma_shape_bench.hip); the question is what survives next to the data movement of a real k-loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o exp/gemm_mfma_shape_ab exp/gemm_mfma_shape_ab.hip -Lmergerec_amd/lib -lmergerec_hip -Wl,-rpath,'$ORIGIN/../mergerec_amd/lib'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../include/mergerec_hip.h"  // the library kernel in the same harness (link with -Lmergerec_amd/lib -lmergerec_hip)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BK = 32, kThreads = 256;
constexpr int ROWB = 64;                    // bytes per LDS row (32 bf16)
constexpr int PIECE = 128 * ROWB;           // 8 KB per 128-row piece
// stage = A hi, A mid (1 piece each) + B hi, B mid (BN / 128 pieces each)

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// B pieces: element (n, k) at ((k / 16) * N + n) * 16 + k % 16   (the library's k-blocked layout)
template <int SHAPE, int BN, int PIPE = 0>
__global__ __launch_bounds__(kThreads, (BN == 128 ? 2 : 1)) void gemm_ab_kernel(const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ wh,
                                                            const uint16_t* __restrict__ wm, int M, int N, int K, float* __restrict__ C,
                                                            int64_t ldc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int BQ = BN / 128, BPIECE = BQ * PIECE, STAGE = 2 * PIECE + 2 * BPIECE, WN = BN / 2;  // WN = wave tile columns
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm_ = wave >> 1, wn_ = wave & 1;

    // staging maps: A rows sr + 32 q (q < 4), float4 kq (8 per row); B rows br (128), 16-byte chunk bc (4 per row), both pieces
    const int sr = tid >> 3, kq = tid & 7;
    const float* ga[4];
    int wa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = sr + 32 * q;
        int gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        ga[q] = A + (int64_t)gr * lda + kq * 4;
        wa[q] = row * ROWB + swz(row, kq >> 1) * 16 + (kq & 1) * 8;
    }
    const int br = tid >> 1, bhalf = tid & 1;
    int64_t gb[2 * BQ];
    int wb[2 * BQ];
#pragma unroll
    for (int q = 0; q < BQ; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {  // rows br + 128 q; k-block b of the stage -> chunks 2 b + bhalf
            const int row = br + 128 * q;
            gb[q * 2 + b] = ((int64_t)b * N + n0 + row) * 16 + bhalf * 8;
            wb[q * 2 + b] = row * ROWB + swz(row, 2 * b + bhalf) * 16;
        }
    struct Stage { float4 a[4]; uint4 bh[2 * BQ], bm[2 * BQ]; };
    auto gload = [&](Stage& st, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st.a[q] = *reinterpret_cast<const float4*>(ga[q] + k0);
#pragma unroll
        for (int b = 0; b < 2 * BQ; ++b) {
            st.bh[b] = *reinterpret_cast<const uint4*>(wh + gb[b] + (int64_t)(k0 / 16) * N * 16);
            st.bm[b] = *reinterpret_cast<const uint4*>(wm + gb[b] + (int64_t)(k0 / 16) * N * 16);
        }
    };
    auto lstore = [&](const Stage& st, unsigned char* buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint2 h, m;
            h.x = pack2(st.a[q].x, st.a[q].y);
            h.y = pack2(st.a[q].z, st.a[q].w);
            m.x = pack2(st.a[q].x - lo_f(h.x), st.a[q].y - hi_f(h.x));
            m.y = pack2(st.a[q].z - lo_f(h.y), st.a[q].w - hi_f(h.y));
            *reinterpret_cast<uint2*>(buf + 0 * PIECE + wa[q]) = h;
            *reinterpret_cast<uint2*>(buf + 1 * PIECE + wa[q]) = m;
        }
#pragma unroll
        for (int b = 0; b < 2 * BQ; ++b) {
            *reinterpret_cast<uint4*>(buf + 2 * PIECE + wb[b]) = st.bh[b];
            *reinterpret_cast<uint4*>(buf + 2 * PIECE + BPIECE + wb[b]) = st.bm[b];
        }
    };

    constexpr int NJ16 = WN / 16, NJ32 = WN / 32;  // MFMA tiles along N per wave
    f32x4 acc16[SHAPE == 16 ? 4 * NJ16 : 1];
    f32x16 acc32[SHAPE == 32 ? 2 * NJ32 : 1];
    if constexpr (SHAPE == 16) {
#pragma unroll
        for (int i = 0; i < 4 * NJ16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 2 * NJ32; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    }

    auto compute = [&](const unsigned char* buf) {
        if constexpr (SHAPE == 16) {
            // lane (r = lane % 16, g = lane / 16): 16 B = k 8 g .. 8 g + 7 of row r
            const int r = lane & 15, g = lane >> 4;
            bf16x8 a[4][2], b[NJ16][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm_ * 64 + i * 16 + r;
#pragma unroll
                for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * PIECE + ra * ROWB + swz(ra, g) * 16);
            }
#pragma unroll
            for (int j = 0; j < NJ16; ++j) {
                const int rb = wn_ * WN + j * 16 + r;
#pragma unroll
                for (int p = 0; p < 2; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(buf + 2 * PIECE + p * BPIECE + rb * ROWB + swz(rb, g) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ16; ++j) {
                    f32x4 c = acc16[i * NJ16 + j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][0], c, 0, 0, 0);  // mid * hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][1], c, 0, 0, 0);  // hi  * mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], c, 0, 0, 0);  // hi  * hi
                    acc16[i * NJ16 + j] = c;
                }
        } else {
            // lane (lr = lane % 32, lh = lane / 32): 16 B = k 16 s + 8 lh .. + 7 of row lr, for the two 16-k sub-steps s
            const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 a[2][2], b[NJ32][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ra = wm_ * 64 + i * 32 + lr;
#pragma unroll
                    for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * PIECE + ra * ROWB + swz(ra, 2 * s + lh) * 16);
                }
#pragma unroll
                for (int j = 0; j < NJ32; ++j) {
                    const int rb = wn_ * WN + j * 32 + lr;
#pragma unroll
                    for (int p = 0; p < 2; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(buf + 2 * PIECE + p * BPIECE + rb * ROWB + swz(rb, 2 * s + lh) * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ32; ++j) {
                        f32x16 c = acc32[i * NJ32 + j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                        acc32[i * NJ32 + j] = c;
                    }
            }
        }
    };

    const int nk = K / BK;
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + STAGE;
    if constexpr (PIPE == 0) {
        Stage st;
        gload(st, 0);
        lstore(st, buf0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            gload(st, (kt + 1 < nk ? kt + 1 : 0) * BK);
            compute((kt & 1) ? buf1 : buf0);
            lstore(st, (kt & 1) ? buf0 : buf1);
            __syncthreads();
        }
    } else {
        // prefetch distance 2 (nk even) + one scheduling region per stage: this stage's MFMAs interleaved with the split / LDS store of the
        // next stage and the global prefetch two stages ahead (the structure of the library kernel's hot path)
        auto ktile = [&](int kt) { return (kt < nk ? kt : 0) * BK; };
        Stage s0, s1;
        gload(s0, 0);
        lstore(s0, buf0);
        gload(s1, ktile(1));
        gload(s0, ktile(2));
        __syncthreads();
#define SGB(m, n) __builtin_amdgcn_sched_group_barrier(m, n, 0)
#define SLOT_V SGB(0x008, 1); SGB(0x002, 2);
#define SLOT_VD SGB(0x008, 1); SGB(0x002, 1); SGB(0x200, 1);
#define SLOT_VM SGB(0x008, 1); SGB(0x002, 1); SGB(0x020, 1);
#define PIPE48                                                                                                               \
    SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V              \
    SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V SLOT_V                                                                      \
    SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD SLOT_VD                              \
    SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM SLOT_VM
        for (int kt = 0; kt < nk; kt += 2) {
            compute(buf0);
            lstore(s1, buf1);
            gload(s1, ktile(kt + 3));
            if (PIPE == 2 && SHAPE == 16 && BN == 128) { PIPE48 }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            compute(buf1);
            lstore(s0, buf0);
            gload(s0, ktile(kt + 4));
            if (PIPE == 2 && SHAPE == 16 && BN == 128) { PIPE48 }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue (plain stores; M, N multiples of the tile in this experiment except the last row tile)
    if constexpr (SHAPE == 16) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ16; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + wm_ * 64 + i * 16 + 4 * g + e, col = n0 + wn_ * WN + j * 16 + r;
                    if (row < M) C[(int64_t)row * ldc + col] = acc16[i * NJ16 + j][e];
                }
    } else {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ32; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm_ * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, col = n0 + wn_ * WN + j * 32 + lr;
                    if (row < M) C[(int64_t)row * ldc + col] = acc32[i * NJ32 + j][e];
                }
    }
}

static uint16_t f2bf(float f) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int SHAPE, int BN, int PIPE = 0>
static double run(const float* dA, const uint16_t* dh, const uint16_t* dm, int M, int N, int K, float* dC, int reps) {
    constexpr int STAGE = 2 * PIECE + 2 * (BN / 128) * PIECE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ab_kernel<SHAPE, BN, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    const int nwg = ((M + BM - 1) / BM) * (N / BN);
    auto launch = [&] { hipLaunchKernelGGL((gemm_ab_kernel<SHAPE, BN, PIPE>), dim3(nwg), dim3(kThreads), 2 * STAGE, 0, dA, (int64_t)K, dh, dm, M, N, K, dC, (int64_t)N); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536;
    struct Shape { const char* name; int N, K; } shapes[] = {{"out", 768, 768}, {"ffn1", 3072, 768}, {"ffn2", 768, 3072}};
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        std::vector<uint16_t> hh((size_t)N * K), hm((size_t)N * K);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const float w = hW[(size_t)n * K + k];
                const uint16_t h = f2bf(w), m = f2bf(w - bf2f(h));
                const size_t dst = ((size_t)(k / 16) * N + n) * 16 + k % 16;
                hh[dst] = h;
                hm[dst] = m;
            }
        float *dA, *dC;
        uint16_t *dh, *dm;
        CK(hipMalloc(&dA, hA.size() * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMalloc(&dh, hh.size() * 2));
        CK(hipMalloc(&dm, hm.size() * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dm, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
        const double flop = 2.0 * M * N * K;
        std::vector<float> c16(64 * (size_t)N), c32(64 * (size_t)N);
        const double t32 = run<32, 128>(dA, dh, dm, M, N, K, dC, 10);
        CK(hipMemcpy(c32.data(), dC, c32.size() * 4, hipMemcpyDeviceToHost));
        const double t16 = run<16, 128>(dA, dh, dm, M, N, K, dC, 10);
        const double t32w = run<32, 256>(dA, dh, dm, M, N, K, dC, 10);
        const double t16w = run<16, 256>(dA, dh, dm, M, N, K, dC, 10);
        const double t16p1 = run<16, 128, 1>(dA, dh, dm, M, N, K, dC, 10);
        const double t32wp1 = run<32, 256, 1>(dA, dh, dm, M, N, K, dC, 10);
        const double t16p2 = run<16, 128, 2>(dA, dh, dm, M, N, K, dC, 10);
        double tlib = 0;
        {
            auto launch = [&] { if (mr_gemm_nt_bf16x6_f32(dA, K, dh, dm, dh, 0, 0, 0, nullptr, nullptr, nullptr, 1, M, N, K, 0, nullptr, 0, dC, N, 3, 0)) { printf("lib gemm failed\n"); exit(1); } };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            tlib = ms / 10;
        }
        printf("      LIBRARY kernel, same harness: %.3f ms = %.0f TFLOP/s\n", tlib, flop / tlib / 1e9);
        CK(hipMemcpy(c16.data(), dC, c16.size() * 4, hipMemcpyDeviceToHost));
        printf("      prefetch distance 2: 16x16x32 narrow %.3f ms = %.0f TFLOP/s | 32x32x16 wide %.3f ms = %.0f | 16x16x32 narrow + interleave %.3f ms = %.0f TFLOP/s\n",
               t16p1, flop / t16p1 / 1e9, t32wp1, flop / t32wp1 / 1e9, t16p2, flop / t16p2 / 1e9);
        printf("      wide 128x256 (1 workgroup/CU): 32x32x16 %.3f ms = %.0f TFLOP/s | 16x16x32 %.3f ms = %.0f TFLOP/s\n", t32w, flop / t32w / 1e9, t16w, flop / t16w / 1e9);
        double err16 = 0, err32 = 0, ref_max = 0;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < N; c += 37) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)hA[(size_t)r * K + k] * (double)hW[(size_t)c * K + k];
                err16 = fmax(err16, fabs(s - c16[(size_t)r * N + c]));
                err32 = fmax(err32, fabs(s - c32[(size_t)r * N + c]));
                ref_max = fmax(ref_max, fabs(s));
            }
        printf("%-5s M=%d N=%d K=%d: 32x32x16 %.3f ms = %.0f TFLOP/s (err %.1e) | 16x16x32 %.3f ms = %.0f TFLOP/s (err %.1e) | ref max %.2f | 16/32 speed %.3f\n",
               sh.name, M, N, K, t32, flop / t32 / 1e9, err32, t16, flop / t16 / 1e9, err16, ref_max, t32 / t16);
        CK(hipFree(dA)); CK(hipFree(dC)); CK(hipFree(dh)); CK(hipFree(dm));
    }
    return 0;
}
