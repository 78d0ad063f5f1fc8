// EXPERIMENT (standalone bench): 256x256x32 split-bf16 GEMM, 4 waves (128x128 wave tiles), 1 workgroup per CU.
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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
constexpr int BM = 256, BN = 256, BK = 32, NTHR = 256;
constexpr int ROWB = 32;
constexpr int SUB = 256 * ROWB;          // one piece x one 16-k sub-block x 256 rows = 8 KB
constexpr int ASUB = SUB + 64;           // A sub-block stride (64-B skew: conflict-free ds_write_b64 across the two sub-blocks)
constexpr int ASTAGE = 4 * ASUB;         // [piece][sub]
constexpr int BSTAGE = 4 * SUB;
constexpr int STAGE = ASTAGE + BSTAGE;
constexpr int LDS_BYTES = 2 * STAGE;

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// one 1-KiB LDS-DMA piece: lane l's 16 bytes at gbase + voff land at LDS ldst + 16 l  (M0 = LDS base; the compiler does not
// know this writes LDS, so every ordering against ds_read / ds_write is by the explicit waits + barriers below)
__device__ __forceinline__ void dma16(uint32_t ldst, uint32_t voff, const void* gbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldst), "v"(voff), "s"(gbase) : "memory", "m0");
}

struct Frag { bf16x8 a[4][2], b[4][2]; };

// 256 x 256 x 32 tiles, 4 waves (2 x 2), wave tile 128 x 128 = 4 x 4 MFMA tiles (256 accumulator registers -> AGPRs), ONE
// workgroup (one wave per SIMD) per CU: everything a wave needs next is fetched while its MFMAs run.
//   step t, first half : MFMAs of sub-block 0 (fragments F0) | fragment reads F1 <- (t, sub 1) | split + LDS store of A(t+1)
//                        | wait for the B(t+1) DMA | global prefetch of A(t+2)            -- then the only barrier of the step
//   step t, second half: MFMAs of sub-block 1 (F1) | fragment reads F0 <- (t+1, sub 0) | issue the B(t+2) DMA
// tile t lives in LDS stage t & 1; the single barrier per step orders every write of tile t+1 before its first read and
// every read of tile t before the first write of tile t+2.
__global__ __launch_bounds__(NTHR, 1) void gemm_v6_kernel(const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ wh,
                                                          const uint16_t* __restrict__ wm_, const float* __restrict__ bias, int M,
                                                          int N, int K, float* __restrict__ C, int64_t ldc, int tiles_n, int nwg) {
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int pid = xcd_remap(blockIdx.x, nwg);
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- A staging: thread -> rows ar + 32 i (i < 8), float4 kq of the row's 128-byte line
    const int ar = tid >> 3, kq = tid & 7;
    // tile-local buffer resource: rows past M read as zeros (they only feed accumulator rows that are never stored)
    const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (int64_t)m0 * lda), 0,
                                                                         (int)((int64_t)rows_valid * lda * 4), 0x00020000);
    const uint32_t avoff = (uint32_t)ar * (uint32_t)lda * 4u + kq * 16;
    const uint32_t arstep = 32u * (uint32_t)lda * 4u;
    const int wa = (kq >> 2) * ASUB + ar * ROWB + ((((kq & 3) >> 1) ^ ((ar >> 3) & 1)) * 16) + (kq & 1) * 8;
    // ---- B DMA: wave -> piece (wave >> 1), sub-block (wave & 1), eight row groups of 32 rows; one per-lane offset for all
    const uint16_t* bsrc = (wave >> 1) ? wm_ : wh;
    const uint32_t bvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16);
    const uint32_t bdst = lds0 + ASTAGE + ((wave >> 1) * 2 + (wave & 1)) * SUB;
    // ---- fragment read offsets
    const int ra = (wm * 128 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);
    const int rb = ASTAGE + (wn * 128 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 av[8];
    auto gloadA = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (whole-vector bit_cast: subscripting the builtin's result directly replicates element 0 with this compiler)
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, i * arstep + kt * (BK * 4), 0));
            av[i] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    auto dmaB = [&](int stage, int kt) {  // k-blocked weights: block (kt * 2 + sub), rows n0 + 32 g ..
        const uint16_t* base = bsrc + ((int64_t)(kt * 2 + (wave & 1)) * N + n0) * 16;
#pragma unroll
        for (int g = 0; g < 8; ++g) dma16(bdst + stage * STAGE + g * 1024, bvoff, base + g * 32 * 16);
    };
    auto lstoreA = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint2 h, m;
            split4x2(av[i], h, m);
            *reinterpret_cast<uint2*>(buf + 0 * 2 * ASUB + wa + i * 32 * ROWB) = h;
            *reinterpret_cast<uint2*>(buf + 1 * 2 * ASUB + wa + i * 32 * ROWB) = m;
        }
    };
    auto fread = [&](Frag& f, const unsigned char* buf, int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < 4; ++i) f.a[i][p] = *reinterpret_cast<const bf16x8*>(buf + (p * 2 + s) * ASUB + ra + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < 4; ++j) f.b[j][p] = *reinterpret_cast<const bf16x8*>(buf + (p * 2 + s) * SUB + rb + j * 32 * ROWB);
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };

    const int nk = K / BK;
    auto ktile = [&](int kt) { return kt < nk ? kt : 0; };  // past-the-end prefetches re-read tile 0 (never consumed)
    Frag f0, f1;
    // ---- prologue: tile 0 complete in stage 0, B(1) in flight, A(1) in registers, F0 <- (0, sub 0)
    dmaB(0, 0);
    gloadA(0);
    lstoreA(lds);
    gloadA(ktile(1));
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all but the 8 youngest (A(1)) done: DMA B(0) landed
    dmaB(1, ktile(1));
    fread(f0, lds, 0);
    for (int t = 0; t < nk; ++t) {
        unsigned char* cur = lds + (t & 1) * STAGE;
        unsigned char* nxt = lds + ((t + 1) & 1) * STAGE;
        // ---- first half
        fread(f1, cur, 1);
        mma(f0);
        lstoreA(nxt);                                        // A(t+1) -> stage (t+1) & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // B(t+1) landed
        gloadA(ktile(t + 2));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- second half
        fread(f0, nxt, 0);
        mma(f1);
        dmaB(t & 1, ktile(t + 2));                           // B(t+2) -> stage t & 1 (all its reads precede the barrier above)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue
    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_clk[blockIdx.x * 4] = clk_c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = clk_r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N);
    int lr_e = lr, lh_e = lh;
    asm volatile("" : "+v"(lr_e), "+v"(lh_e));
    if (interior) {
        float bz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bz[j] = bias[n0 + wn * 128 + j * 32 + lr_e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t rbase = m0 + wm * 128 + i * 32 + 4 * lh_e;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t col = n0 + wn * 128 + j * 32 + lr_e;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) C[(rbase + r + 8 * q) * ldc + col] = acc[i][j][4 * q + r] + bz[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 128 + j * 32 + lr_e;
        const bool col_ok = col < N;
        const float bz = col_ok ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = m0 + wm * 128 + i * 32 + 4 * lh_e;
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
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
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
        if (K % BK || N % BN) { printf("K %% 32, N %% 256 required\n"); return 1; }
        auto launch = [&] {
            hipLaunchKernelGGL(gemm_v6_kernel, dim3(nwg), dim3(NTHR), LDS_BYTES, 0, dA, (int64_t)K, dh, dm, db, M, N, K, dC, (int64_t)N,
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
        printf("v6 %-5s M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s alg (best %.1f)  max rel-to-magnitude err %.3g %s\n", sh.name, M, N, K,
               ts[ts.size() / 2], fl / ts[ts.size() / 2] / 1e9, fl / ts[0] / 1e9, worst, worst < 3e-5 ? "OK" : "FAIL");
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dh); hipFree(dm); hipFree(dr); hipFree(dc); hipFree(dref); hipFree(dmag);
    }
    return 0;
}
