// EXPERIMENT (standalone bench): split-bf16 GEMM whose BOTH operands arrive pre-split in the k-blocked, swizzle-baked piece layout
//   element (row r, k) of a piece lives at ((k >> 4) * R_pad + r) * 16 + ((((k >> 3) & 1) ^ ((r >> 3) & 1)) * 8) + (k & 7)
// so a (256 rows x 16 k) tile of a piece is ONE contiguous 8 KB chunk that goes global -> LDS by LDS-DMA (global_load_lds_dwordx4),
// lane-linear, with no VGPR staging, no conversion and no ds_write in the main loop.  Roles are swapped w.r.t. the library kernel:
// the MFMA A operand is the WEIGHT tile (features), the B operand the ACTIVATION tile (tokens), so a lane of the accumulator owns
// one token and 4-feature runs -- fp32 rows are stored 16 B per lane, piece rows (GELU + split epilogue) 16 B per lane after a
// v_permlane32_swap.
// 256 (features) x 256 (tokens) x 16 tile, 8 waves (2 x 4), wave tile 128 x 64, NSTAGE LDS stages of NP * 16 KB.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/gemm_pk exp/gemm_pk.hip -L mergerec_amd/lib -lmergerec_hip -Wl,-rpath,$PWD/mergerec_amd/lib
// run:   exp/gemm_pk [M] [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../include/mergerec_hip.h"

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

// fp32 (R, K) row-major -> NP pieces in the k-blocked swizzled layout with R_pad rows per k-block (rows >= R are zeros)
__global__ void split_pk_kernel(const float* __restrict__ x, int R, int K, int64_t R_pad, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const int64_t total = R_pad * (K / 4);
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = u / (K / 4);
        const int k = (int)(u - r * (K / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) v = *reinterpret_cast<const float4*>(x + r * K + k);
        uint2 h, l;
        h.x = pack2(v.x, v.y); h.y = pack2(v.z, v.w);
        l.x = pack2(v.x - lo_f(h.x), v.y - hi_f(h.x)); l.y = pack2(v.z - lo_f(h.y), v.w - hi_f(h.y));
#ifdef PK_LOMASK  // diagnostic: data-dependent MFMA power -- drop the low mantissa bits of the lo piece (changes numerics)
        l.x &= PK_LOMASK; l.y &= PK_LOMASK;
#endif
        const int64_t dst = ((int64_t)(k >> 4) * R_pad + r) * 16 + ((((k >> 3) & 1) ^ (int)((r >> 3) & 1)) * 8) + (k & 7);
        *reinterpret_cast<uint2*>(hi + dst) = h;
        *reinterpret_cast<uint2*>(lo + dst) = l;
    }
}

__device__ unsigned long long g_clk[8192 * 4];
constexpr int TF = 256, TT = 256, BK = 16, NTHR = 512;
constexpr int ROWB = 32;
constexpr int SUB = 256 * ROWB;  // one piece x one k-block x 256 rows = 8 KB

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// OUT: 0 = fp32 C (T, N) row-major (+ bias, + residual R); 1 = GELU(acc + bias) written as hi / lo pieces [N/16][T_pad][16]
template <int NP, int NSTAGE, int OUT, bool HAS_R, int NWT = 4>
__global__ __launch_bounds__(NWT * 128, (NWT == 4 ? 1 : 2)) void gemm_pk_kernel(const uint16_t* __restrict__ wh, const uint16_t* __restrict__ wl, int N,
                                                          const uint16_t* __restrict__ xh, const uint16_t* __restrict__ xl, int64_t T_pad,
                                                          const float* __restrict__ bias, int T, int K, const float* __restrict__ R, int64_t ldr,
                                                          float* __restrict__ C, int64_t ldc, uint16_t* __restrict__ oh, uint16_t* __restrict__ ol,
                                                          int tiles_f, int nwg) {
#ifdef PK_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TTK = NWT * 64;                  // tokens per tile
    constexpr int XSUB = TTK * ROWB;               // one X piece sub-block
    constexpr int STAGE = NP * SUB + NP * XSUB;    // [W pieces][X pieces]
    const int pid = xcd_remap(blockIdx.x, nwg);
    const int tt = pid / tiles_f, tf = pid - tt * tiles_f;  // feature tile fastest: the tiles_f workgroups of one token panel are neighbours
    const int f0 = tf * TF, t0 = tt * TTK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave / NWT, wt = wave % NWT;  // 2 (features) x NWT (tokens)
    const int lr = lane & 31, lh = lane >> 5;

    // ---- DMA: 1 KB (32 rows) per wave instruction, lane-linear.  W tile = 8 chunks, X tile = 2 NWT chunks, dealt over the 2 NWT waves
    constexpr int NWAVE = 2 * NWT;
    constexpr int WCH = 8 / NWAVE;   // W chunks per wave and piece (1 or 2)
    const int dhalf = lane & 1, dr = lane >> 1;
    const int64_t wstep = (int64_t)N * 16, xstep = T_pad * 16;
    const uint16_t* pw[NP][WCH];
    const uint16_t* px[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int c = 0; c < WCH; ++c) pw[p][c] = (p == 0 ? wh : wl) + ((int64_t)(f0 + (wave + c * NWAVE) * 32 + dr)) * 16 + dhalf * 8;
        px[p] = (p == 0 ? xh : xl) + ((int64_t)(t0 + wave * 32 + dr)) * 16 + dhalf * 8;
    }
    auto dma = [&](int stage, bool advance) {
        unsigned char* buf = lds + stage * STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int c = 0; c < WCH; ++c) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pw[p][c],
                                                 (__attribute__((address_space(3))) void*)(buf + p * SUB + (wave + c * NWAVE) * 1024), 16, 0, 0);
                if (advance) pw[p][c] += wstep;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)px[p],
                                             (__attribute__((address_space(3))) void*)(buf + NP * SUB + p * XSUB + wave * 1024), 16, 0, 0);
            if (advance) px[p] += xstep;
        }
    };
    // ---- fragment read offsets (rows + 32 keep the swizzle bit)
    const int ra = (wf * 128 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);
    const int rb = NP * SUB + (wt * 64 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);  // + p * XSUB

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Frags { bf16x8 a[4][NP], b[2][NP]; };
    auto lread = [&](Frags& f, int stage) {
        const unsigned char* buf = lds + stage * STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int j = 0; j < 2; ++j) f.b[j][p] = *reinterpret_cast<const bf16x8*>(buf + p * XSUB + rb + j * 32 * ROWB);
#pragma unroll
            for (int i = 0; i < 4; ++i) f.a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * SUB + ra + i * 32 * ROWB);
        }
    };
    // a = weight piece, b = activation piece; same product order as the library kernel (activation-lo * weight-hi first)
    auto mma = [&](const Frags& f, int i0) {  // accumulator rows i0, i0 + 1 (12 MFMAs)
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);  // w.hi * x.lo
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);  // w.lo * x.hi
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);  // w.hi * x.hi
                acc[i][j] = c;
            }
    };

    const int nk = K / BK;             // host: nk >= NSTAGE, nk even
    constexpr int PER_STAGE = (WCH + 1) * NP;  // DMA instructions per wave and stage
    static_assert(NSTAGE == 4 || NSTAGE == 3, "stage count");
#define PK_WAIT_STAGE() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (NSTAGE - 2)) : "memory")
#define PK_SGB(m, n) __builtin_amdgcn_sched_group_barrier(m, n, 0)
    // prologue: all NSTAGE buffers filling (stages 0 .. NSTAGE-1; host: nk > NSTAGE); fragments of stage 0 in registers
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) dma(s, true);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (NSTAGE - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    Frags f0_, f1_;
    lread(f0_, 0);
    int cur = 0;  // buffer of stage kt
    // One iteration (stage kt in `fc`): first half of its MFMAs; then -- stage kt + 1 landed (stages kt + 2, kt + 3 may still be in
    // flight), barrier (every wave holds stage kt in registers, so its buffer is free) -- refill that buffer with stage kt + NSTAGE and
    // read stage kt + 1's fragments into the other register set while the second half of the MFMAs runs.
    auto iter = [&](Frags& fc, Frags& fn, int kt) {
        __builtin_amdgcn_sched_barrier(0);
        mma(fc, 0);
        __builtin_amdgcn_sched_barrier(0);
        PK_WAIT_STAGE();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nxt = cur + 1 == NSTAGE ? 0 : cur + 1;
        const bool adv = kt + NSTAGE + 1 < nk;   // this issue moved block min(kt + NSTAGE, nk - 1); advance while a next block exists
        dma(cur, false);
#pragma unroll
        for (int p = 0; p < NP; ++p) {  // branch-free: keeps the region whole
#pragma unroll
            for (int c = 0; c < WCH; ++c) pw[p][c] += wstep & -(int64_t)adv;
            px[p] += xstep & -(int64_t)adv;
        }
        lread(fn, nxt);
        mma(fc, 2);
        // second half as ONE scheduling region: the next stage's 12 fragment reads and the DMA issues ride in the gaps of the 12 MFMAs
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            PK_SGB(0x008, 1);
            if (g < PER_STAGE) PK_SGB(0x010, 1);
            PK_SGB(0x100, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        iter(f0_, f1_, kt);
        iter(f1_, f0_, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PK_CLOCK
    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_clk[blockIdx.x * 4] = clk_c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = clk_r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- epilogue.  acc[i][j][r]: feature f0 + wf*128 + i*32 + (r & 3) + 8 * (r >> 2) + 4 * lh, token t0 + wt*64 + j*32 + lr
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = t0 + wt * 64 + j * 32 + lr;  // wt < NWT
        const bool t_ok = t < T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fb = f0 + wf * 128 + i * 32 + 4 * lh;  // + 8 q + (0..3)
            if (OUT == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bz = *reinterpret_cast<const float4*>(bias + fb + 8 * q);
                    float4 v = make_float4(acc[i][j][4 * q] + bz.x, acc[i][j][4 * q + 1] + bz.y, acc[i][j][4 * q + 2] + bz.z, acc[i][j][4 * q + 3] + bz.w);
                    if (t_ok) {
                        if (HAS_R) {
                            const float4 rr = *reinterpret_cast<const float4*>(R + (int64_t)t * ldr + fb + 8 * q);
                            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        }
                        *reinterpret_cast<float4*>(C + (int64_t)t * ldc + fb + 8 * q) = v;
                    }
                }
            } else {
                // two 16-feature k-blocks per 32-feature MFMA tile: q = 0, 1 -> k-block 2 * (tile) + 0, q = 2, 3 -> + 1
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    uint32_t h[2][2], l[2][2];  // [q within the k-block][dword]
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = hb * 2 + qq;
                        const float4 bz = *reinterpret_cast<const float4*>(bias + fb + 8 * q);
                        const float v0 = gelu_erf(acc[i][j][4 * q] + bz.x), v1 = gelu_erf(acc[i][j][4 * q + 1] + bz.y);
                        const float v2 = gelu_erf(acc[i][j][4 * q + 2] + bz.z), v3 = gelu_erf(acc[i][j][4 * q + 3] + bz.w);
                        h[qq][0] = pack2(v0, v1); h[qq][1] = pack2(v2, v3);
                        l[qq][0] = pack2(v0 - lo_f(h[qq][0]), v1 - hi_f(h[qq][0]));
                        l[qq][1] = pack2(v2 - lo_f(h[qq][1]), v3 - hi_f(h[qq][1]));
                    }
                    // lanes < 32 hold features 0-3 (qq 0) and 8-11 (qq 1) of the k-block, lanes >= 32 hold 4-7 and 12-15: after the swap the
                    // low half-wave owns features 0-7, the high half-wave 8-15 -- one 16-byte store per lane and piece
#pragma unroll
                    for (int dw = 0; dw < 2; ++dw) {
                        auto sw = __builtin_amdgcn_permlane32_swap(h[0][dw], h[1][dw], false, false);
                        h[0][dw] = sw[0]; h[1][dw] = sw[1];
                        auto sl = __builtin_amdgcn_permlane32_swap(l[0][dw], l[1][dw], false, false);
                        l[0][dw] = sl[0]; l[1][dw] = sl[1];
                    }
                    const int kb = (f0 + wf * 128 + i * 32) / 16 + hb;
                    const int64_t dst = ((int64_t)kb * T_pad + t) * 16 + ((lh ^ ((t >> 3) & 1)) * 8);
                    if (t_ok) {
                        *reinterpret_cast<uint4*>(oh + dst) = make_uint4(h[0][0], h[0][1], h[1][0], h[1][1]);
                        *reinterpret_cast<uint4*>(ol + dst) = make_uint4(l[0][0], l[0][1], l[1][0], l[1][1]);
                    }
                }
            }
        }
    }
}

// unpack a piece pair back to fp32 (T, N) for checking
__global__ void unsplit_pk_kernel(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, int T, int N, int64_t T_pad, float* __restrict__ out) {
    const int64_t total = (int64_t)T * N;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = u / N;
        const int k = (int)(u - t * N);
        const int64_t src = ((int64_t)(k >> 4) * T_pad + t) * 16 + ((((k >> 3) & 1) ^ (int)((t >> 3) & 1)) * 8) + (k & 7);
        out[u] = __uint_as_float((uint32_t)hi[src] << 16) + __uint_as_float((uint32_t)lo[src] << 16);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 69632;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    struct Shape { const char* name; int N, K, out; bool res; } shapes[] = {{"qkv", 2304, 768, 0, false}, {"out", 768, 768, 0, true}, {"ffn1", 3072, 768, 1, false}, {"ffn2", 768, 3072, 0, true}};
#ifndef PK_NWT
#define PK_NWT 4
#endif
    constexpr int NWT = PK_NWT;
    constexpr int NST = NWT == 4 ? 4 : 3;
    const size_t LDS_BYTES = (size_t)NST * (2 * SUB + 2 * NWT * 64 * ROWB);
    const int NTHR_ = NWT * 128, TTOK = NWT * 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<2, NST, 0, false, NWT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<2, NST, 0, true, NWT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<2, NST, 1, false, NWT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    srand(1);
    const int64_t T_pad = (M + 255) / 256 * 256;
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
        for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        for (auto& x : hb) x = (float)rand() / RAND_MAX;
        float *dA, *dW, *db, *dC, *dC2, *dR;
        uint16_t *wh, *wl, *xh, *xl, *oh, *ol, *lwh, *lwm, *lwl;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4));
        CK(hipMalloc(&wh, hW.size() * 2)); CK(hipMalloc(&wl, hW.size() * 2));
        CK(hipMalloc(&lwh, hW.size() * 2)); CK(hipMalloc(&lwm, hW.size() * 2)); CK(hipMalloc(&lwl, hW.size() * 2));
        CK(hipMalloc(&xh, (size_t)T_pad * K * 2)); CK(hipMalloc(&xl, (size_t)T_pad * K * 2));
        CK(hipMalloc(&oh, (size_t)T_pad * N * 2)); CK(hipMalloc(&ol, (size_t)T_pad * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        CK(hipMemset(dR, 0, (size_t)M * N * 4));
        hipLaunchKernelGGL(split_pk_kernel, dim3(4096), dim3(256), 0, 0, dW, N, K, (int64_t)N, wh, wl);
        hipLaunchKernelGGL(split_pk_kernel, dim3(4096), dim3(256), 0, 0, dA, M, K, T_pad, xh, xl);
        // the library kernel's weight pieces (k-blocked, unswizzled) for the A/B timing and the bitwise comparison
        {
            int64_t tab[3] = {0, N, K}, pref[2] = {0, (int64_t)N * K / 4};
            int64_t *dt, *dp;
            CK(hipMalloc(&dt, 24)); CK(hipMalloc(&dp, 16));
            CK(hipMemcpy(dt, tab, 24, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, pref, 16, hipMemcpyHostToDevice));
            int rc = mr_split_weights_kblock_f32(dW, dt, dp, 1, pref[1], lwh, lwm, lwl, 0);
            if (rc) { printf("split_weights rc %d\n", rc); return 1; }
            CK(hipDeviceSynchronize());
            hipFree(dt); hipFree(dp);
        }
        const int tiles_t = (M + TTOK - 1) / TTOK, tiles_f = N / TF, nwg = tiles_t * tiles_f;
        auto launch = [&] {
            if (sh.out == 1)
                hipLaunchKernelGGL((gemm_pk_kernel<2, NST, 1, false, NWT>), dim3(nwg), dim3(NTHR_), LDS_BYTES, 0, wh, wl, N, xh, xl, T_pad, db, M, K, nullptr, 0, nullptr, 0, oh, ol, tiles_f, nwg);
            else if (sh.res)
                hipLaunchKernelGGL((gemm_pk_kernel<2, NST, 0, true, NWT>), dim3(nwg), dim3(NTHR_), LDS_BYTES, 0, wh, wl, N, xh, xl, T_pad, db, M, K, dR, (int64_t)N, dC, (int64_t)N, nullptr, nullptr, tiles_f, nwg);
            else
                hipLaunchKernelGGL((gemm_pk_kernel<2, NST, 0, false, NWT>), dim3(nwg), dim3(NTHR_), LDS_BYTES, 0, wh, wl, N, xh, xl, T_pad, db, M, K, nullptr, 0, dC, (int64_t)N, nullptr, nullptr, tiles_f, nwg);
        };
        auto launch_lib = [&] {
            int rc = mr_gemm_nt_bf16x6_f32(dA, K, lwh, lwm, lwl, 0, 0, 0, db, nullptr, nullptr, 1, M, N, K, sh.out == 1 ? 1 : 0, sh.res ? dR : nullptr, N, dC2, N, 3, 0);
            if (rc) { printf("lib gemm rc %d\n", rc); exit(1); }
        };
        launch();
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        launch_lib();
        CK(hipDeviceSynchronize());
        if (sh.out == 1) hipLaunchKernelGGL(unsplit_pk_kernel, dim3(4096), dim3(256), 0, 0, oh, ol, M, N, T_pad, dC);
        CK(hipDeviceSynchronize());
        // ---- compare with the library kernel on the whole output
        std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0; size_t nbad = 0, ndiff = 0;
        for (size_t i = 0; i < c1.size(); ++i) {
            const double d = fabs((double)c1[i] - (double)c2[i]);
            if (c1[i] != c2[i]) ++ndiff;
            const double tol = sh.out == 1 ? 1e-4 * (1.0 + fabs((double)c2[i])) : 0.0;  // piece output drops bits below 2^-16 relative
            if (!(d <= tol)) ++nbad;
            if (!(d <= worst)) worst = d;
        }
        // ---- time: interleaved rounds in one process
        std::vector<float> ts, tl;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            CK(hipEventRecord(e0, 0)); launch_lib(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); tl.push_back(ms);
        }
        std::sort(ts.begin(), ts.end()); std::sort(tl.begin(), tl.end());
        const double fl = 2.0 * M * N * K;
        printf("pk %-5s M=%d N=%d K=%d: %.3f ms %.1f TFLOP/s alg (best %.1f) | library %.3f ms %.1f TFLOP/s | max abs diff %.3g, %zu differing, %zu beyond tol %s\n",
               sh.name, M, N, K, ts[ts.size() / 2], fl / ts[ts.size() / 2] / 1e9, fl / ts[0] / 1e9, tl[tl.size() / 2], fl / tl[tl.size() / 2] / 1e9,
               worst, ndiff, nbad, nbad == 0 ? "OK" : "FAIL");
#ifdef PK_CLOCK
        {
            static unsigned long long hc[8192 * 4];
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
            double cs = 0, rs = 0;
            const int nw = nwg < 8192 ? nwg : 8192;
            for (int w = 0; w < nw; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
            const double ghz = cs / rs * 0.1, mf = 3.0 * fl / ts[ts.size() / 2] / 1e9;
            printf("      main-loop clock %.2f GHz; MFMA %.0f TFLOP/s = %.0f %% of the pipe at that clock\n", ghz, mf, 100.0 * mf / (2500.0 * ghz / 2.4));
        }
#endif
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dC2); hipFree(dR); hipFree(wh); hipFree(wl); hipFree(xh); hipFree(xl);
        hipFree(oh); hipFree(ol); hipFree(lwh); hipFree(lwm); hipFree(lwl);
    }
    return 0;
}
