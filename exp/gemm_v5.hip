// EXPERIMENT (standalone bench, not part of the library): 256x256x32 split-bf16 GEMM, 8 waves, 1 workgroup per CU.
//   A (fp32) : full 128-B lines per row and k-step, split to bf16 pieces in registers, ds_write_b64 to LDS
//   B (bf16 pieces, k-blocked [K/16][N][16]) : LDS-DMA (global_load_lds_dwordx4), swizzle applied on the global side
// build: hipcc --offload-arch=gfx950 -O3 -o exp/gemm_v5 exp/gemm_v5.hip ; run: exp/gemm_v5 [M]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split4x2(const float4 x, uint2& h, uint2& m) {
    h.x = pack2(x.x, x.y);
    h.y = pack2(x.z, x.w);
    m.x = pack2(x.x - lo_f(h.x), x.y - hi_f(h.x));
    m.y = pack2(x.z - lo_f(h.y), x.w - hi_f(h.y));
}

__global__ void split_w_kernel(const float* __restrict__ w, int N, int K, uint16_t* __restrict__ hi, uint16_t* __restrict__ mid) {
    const int64_t total = (int64_t)N * K / 4;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = u * 4, n = e / K, k = e - n * K;
        uint2 h, m;
        split4x2(*reinterpret_cast<const float4*>(w + e), h, m);
        const int64_t dst = ((k >> 4) * N + n) * 16 + (k & 15);
        *reinterpret_cast<uint2*>(hi + dst) = h;
        *reinterpret_cast<uint2*>(mid + dst) = m;
    }
}

__device__ unsigned long long g_clk[8192 * 4];
constexpr int BM = 256, BN = 256, BK = 32, NTHR = 512;
constexpr int ROWB = 32;
constexpr int SUB = 256 * ROWB;          // one piece x one 16-k sub-block x 256 rows = 8 KB
constexpr int ASUB = SUB + 64;           // A sub-block stride (64-B skew: conflict-free ds_write_b64 across sub-blocks)
constexpr int ASTAGE = 4 * ASUB;         // [piece][sub]
constexpr int BSTAGE = 4 * SUB;
constexpr int STAGE = ASTAGE + BSTAGE;
constexpr int LDS_BYTES = 2 * STAGE;

#ifndef V5_RASTER
#define V5_RASTER 0
#endif

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__global__ __launch_bounds__(NTHR, 1) void gemm_v5_kernel(const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ wh,
                                                          const uint16_t* __restrict__ wm_, const float* __restrict__ bias, int M,
                                                          int N, int K, float* __restrict__ C, int64_t ldc, int tiles_n, int nwg) {
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int pid = xcd_remap(blockIdx.x, nwg);
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- A staging: thread -> rows ar + 64 i (i < 4), float4 kq of the 128-B line
    const int ar = tid >> 3, kq = tid & 7;
    const float* ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = m0 + ar + 64 * i;
        r = r < M ? r : M - 1;
        ga[i] = A + (int64_t)r * lda + kq * 4;
    }
    // rows ar + 64 i share the swizzle bit
    const int wa = (kq >> 2) * ASUB + ar * ROWB + ((((kq & 3) >> 1) ^ ((ar >> 3) & 1)) * 16) + (kq & 1) * 8;
    // ---- B DMA: wave -> piece, sub-block, 4 row groups of 32 rows
    const int bp = wave >> 2, bs = (wave >> 1) & 1;
    const uint16_t* __restrict__ bsrc = bp ? wm_ : wh;
    int64_t gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = ((wave & 1) * 4 + q) * 32 + (lane >> 1);
        int n = n0 + row;
        n = n < N ? n : N - 1;
        gb[q] = ((int64_t)bs * N + n) * 16 + (((lane & 1) ^ ((row >> 3) & 1)) * 8);
    }
    const int bdst = ASTAGE + (bp * 2 + bs) * SUB + (wave & 1) * 4 * 1024;  // + q * 1024, wave-uniform
    const int64_t bkstep = (int64_t)2 * N * 16;                               // elements per k-step (two 16-k blocks)
    // ---- fragment read offsets
    const int ra = (wm * 64 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);
    const int rb = ASTAGE + (wn * 128 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct AStage { float4 v[4]; };
    auto gloadA = [&](AStage& st, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st.v[i] = *reinterpret_cast<const float4*>(ga[i] + (int64_t)kt * BK);
    };
    auto dmaB = [&](unsigned char* buf, int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + gb[q] + kt * bkstep),
                                             (__attribute__((address_space(3))) void*)(buf + bdst + q * 1024), 16, 0, 0);
    };
    auto lstoreA = [&](const AStage& st, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, m;
            split4x2(st.v[i], h, m);
            *reinterpret_cast<uint2*>(buf + 0 * 2 * ASUB + wa + i * 64 * ROWB) = h;
            *reinterpret_cast<uint2*>(buf + 1 * 2 * ASUB + wa + i * 64 * ROWB) = m;
        }
    };
    auto compute = [&](const unsigned char* buf) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[2][2], b[4][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(buf + (p * 2 + s) * ASUB + ra + i * 32 * ROWB);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(buf + (p * 2 + s) * SUB + rb + j * 32 * ROWB);
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
        }
    };

    const int nk = K / BK;  // even (host checks)
    auto ktile = [&](int kt) { return kt < nk ? kt : 0; };
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + STAGE;
    AStage s0, s1;
    dmaB(buf0, 0);
    gloadA(s0, 0);
    gloadA(s1, ktile(1));
    lstoreA(s0, buf0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; kt += 2) {
        // step kt (buf0): B tile kt+1 -> buf1 by DMA, A tile kt+2 -> s0, compute, A tile kt+1 (s1) -> buf1
        dmaB(buf1, ktile(kt + 1));
        __builtin_amdgcn_sched_barrier(0);
        gloadA(s0, ktile(kt + 2));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        compute(buf0);
        __builtin_amdgcn_sched_barrier(0);
        lstoreA(s1, buf1);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // DMA landed (A loads of s0 may stay in flight)
        dmaB(buf0, ktile(kt + 2));
        __builtin_amdgcn_sched_barrier(0);
        gloadA(s1, ktile(kt + 3));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        compute(buf1);
        __builtin_amdgcn_sched_barrier(0);
        lstoreA(s0, buf0);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_clk[blockIdx.x * 4] = clk_c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = clk_r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N);
    if (interior) {
        float bz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bz[j] = bias[n0 + wn * 128 + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t col = n0 + wn * 128 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) C[(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[i][j][r] + bz[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 128 + j * 32 + lr;
        const bool col_ok = col < N;
        const float bz = col_ok ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
                if (col_ok && row < M) C[row * ldc + col] = acc[i][j][r] + bz;
            }
        }
    }
}

__global__ void ref_sample_kernel(const float* A, int64_t lda, const float* W, const float* bias, int K, const int* rows,
                                  const int* cols, int ns, double* out, double* mag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    double acc = 0, mg = 0;
    for (int k = 0; k < K; ++k) {
        const double p = (double)A[(int64_t)rows[s] * lda + k] * (double)W[(int64_t)cols[s] * K + k];
        acc += p;
        mg += fabs(p);
    }
    out[s] = acc + bias[cols[s]];
    mag[s] = mg;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out", 768, 768}, {"ffn1", 3072, 768}, {"ffn2", 768, 3072}};
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    srand(1);
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
        for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        for (auto& x : hb) x = (float)rand() / RAND_MAX;
        float *dA, *dW, *db, *dC;
        uint16_t *dh, *dm;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dh, hW.size() * 2)); CK(hipMalloc(&dm, hW.size() * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        hipLaunchKernelGGL(split_w_kernel, dim3(2048), dim3(256), 0, 0, dW, N, K, dh, dm);
        const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
        if ((K / BK) % 2) { printf("K/32 must be even\n"); return 1; }
        auto launch = [&] {
            hipLaunchKernelGGL(gemm_v5_kernel, dim3(nwg), dim3(NTHR), LDS_BYTES, 0, dA, (int64_t)K, dh, dm, db, M, N, K, dC, (int64_t)N,
                               tiles_n, nwg);
        };
        launch();
        CK(hipDeviceSynchronize());
        // ---- check on samples
        const int ns = 4096;
        std::vector<int> hr(ns), hc(ns);
        for (int i = 0; i < ns; ++i) { hr[i] = i < 64 ? (M - 1 - i) : rand() % M; hc[i] = i < 64 ? (N - 1 - i) : rand() % N; }
        int *dr, *dc; double *dref, *dmag;
        CK(hipMalloc(&dr, ns * 4)); CK(hipMalloc(&dc, ns * 4)); CK(hipMalloc(&dref, ns * 8)); CK(hipMalloc(&dmag, ns * 8));
        CK(hipMemcpy(dr, hr.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), ns * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_sample_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, dA, (int64_t)K, dW, db, K, dr, dc, ns, dref, dmag);
        std::vector<double> href(ns), hmag(ns);
        CK(hipMemcpy(href.data(), dref, ns * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hmag.data(), dmag, ns * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int i = 0; i < ns; ++i) {
            float got;
            CK(hipMemcpy(&got, dC + (size_t)hr[i] * N + hc[i], 4, hipMemcpyDeviceToHost));
            const double rel = fabs((double)got - href[i]) / (hmag[i] + 1e-30);
            if (!(rel <= worst)) worst = rel;  // NaN-propagating max
        }
        // ---- time
        std::vector<float> ts;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double fl = 2.0 * M * N * K;
        {
            static unsigned long long hc[8192 * 4];
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
            double cs = 0, rs = 0;
            const int nw = nwg < 8192 ? nwg : 8192;
            for (int w = 0; w < nw; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
            const double ghz = cs / rs * 0.1, mf = 3.0 * fl / ts[ts.size() / 2] / 1e9;
            printf("      main-loop clock %.2f GHz; MFMA %.0f TFLOP/s = %.0f %% of the pipe at that clock\n", ghz, mf, 100.0 * mf / (2500.0 * ghz / 2.4));
        }
        printf("v5 %-5s M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s alg (best %.1f)  max rel-to-magnitude err %.3g %s\n", sh.name, M, N, K,
               ts[ts.size() / 2], fl / ts[ts.size() / 2] / 1e9, fl / ts[0] / 1e9, worst, worst < 3e-5 ? "OK" : "FAIL");
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dh); hipFree(dm); hipFree(dr); hipFree(dc); hipFree(dref); hipFree(dmag);
    }
    return 0;
}
