// EXPERIMENT (standalone bench): fp32-grade GEMM as ONE fp16 MFMA + ONE int8 MFMA per 32x32x16 tile-step instead of three bf16 MFMAs.
//   x = xh + xl,  xh = fp16(x) (11 significant bits),  |xl| <= 2^-11 |x|;  same for w.   x w = xh wh  +  (xh wl + xl wh)  +  O(2^-22)
// The cross terms are ~2^-12 of the product, so their operands only need ~2^-6 relative accuracy: both go to int8 with one scale per
// (row, 256-wide K chunk):  xq = rint(x / sx), xlq = rint(xl / (sx 2^-11)),  wq = rint(w / sw), wlq = rint(wl / (sw 2^-11)), and
//   xh wl + xl wh  ~=  sx sw 2^-11 (xq . wlq + xlq . wq)   -- ONE i8 MFMA over the concatenation [xq | xlq] . [wlq | wq] (K = 32 per 16 k).
// Per 16 k: v_mfma_f32_32x32x16_f16 + v_mfma_i32_32x32x32_i8 (about 20 + 19 ns per SIMD at the power-managed clock, exp/mfma_energy_bench)
// against 3 x 19.4 ns for the bf16x3 arithmetic.  The i32 accumulator is folded into the f32 one at every chunk boundary.
// Operands arrive pre-split in k-blocked, swizzle-baked images ([K/16][rows][32 B] for both the fp16 and the int8 image) and stream
// global -> LDS by LDS-DMA as in exp/gemm_pk.hip; 256 (features) x 128 (tokens) x 32 tiles, 8 waves (4 x 2), wave tile 64 x 64.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/gemm_f16i8 exp/gemm_f16i8.hip -L mergerec_amd/lib -lmergerec_hip -Wl,-rpath,$PWD/mergerec_amd/lib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../include/mergerec_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK = 128;          // K elements per scale (= the feature tile of the quantizing epilogue)
constexpr float LO_SCALE = 1.0f / 2048.0f;  // 2^-11

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}
__device__ __forceinline__ float f16_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float f16_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
// round-to-nearest-even int8 of v (|v| <= 127.5) as the low byte of (v + 1.5 * 2^23)
__device__ __forceinline__ uint32_t q8(float v) { return __float_as_uint(fminf(fmaxf(v, -127.0f), 127.0f) + 12582912.0f) & 0xffu; }
__device__ __forceinline__ uint32_t q8x4(float a, float b, float c, float d) { return q8(a) | (q8(b) << 8) | (q8(c) << 16) | (q8(d) << 24); }

// fp32 (R, K) row-major -> fp16 image, int8 image ([first | second] halves), scales [K/256][R_pad].  One wave per row and chunk.
// W_ORDER: false -> int8 halves are (q | lq) (activations); true -> (lq | q) (weights), so half h of X meets half h of W in the MFMA.
template <bool W_ORDER>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float* __restrict__ x, int R, int K, int64_t R_pad, uint16_t* __restrict__ img_h,
                                                           uint8_t* __restrict__ img_i, float* __restrict__ scales) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (row, 256-wide k span = two 128-wide scale chunks, one per half-wave)
    const int nspan = K / 256;
    const int64_t r = unit / nspan;
    const int sp = (int)(unit - r * nspan);
    if (r >= R_pad) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k = sp * 256 + lane * 4;
    const int c = k / CHUNK;
    if (r < R) v = *reinterpret_cast<const float4*>(x + r * K + k);
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float s = m > 0.f ? m * (1.0f / 127.0f) : 1.0f;
    if ((lane & 31) == 0) scales[(int64_t)c * R_pad + r] = s;
    const float inv = 1.0f / s, inv_lo = inv * 2048.0f;
    const uint32_t h0 = pack_f16(v.x, v.y), h1 = pack_f16(v.z, v.w);
    const float l0 = v.x - f16_lo(h0), l1 = v.y - f16_hi(h0), l2 = v.z - f16_lo(h1), l3 = v.w - f16_hi(h1);
    const uint32_t q = q8x4(v.x * inv, v.y * inv, v.z * inv, v.w * inv), lq = q8x4(l0 * inv_lo, l1 * inv_lo, l2 * inv_lo, l3 * inv_lo);
    const int64_t rowbase = ((int64_t)(k >> 4) * R_pad + r) * 32;  // bytes, both images
    const int swz = (int)((r >> 3) & 1);
    // fp16 image: logical half (k >> 3) & 1, 8 bytes at (k & 7) * 2
    *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(img_h) + rowbase + ((((k >> 3) & 1) ^ swz) * 16) + (k & 7) * 2) = make_uint2(h0, h1);
    // int8 image: halves hold the 16 k of the block; byte k & 15
    const int hq = W_ORDER ? 1 : 0, hl = W_ORDER ? 0 : 1;
    *reinterpret_cast<uint32_t*>(img_i + rowbase + ((hq ^ swz) * 16) + (k & 15)) = q;
    *reinterpret_cast<uint32_t*>(img_i + rowbase + ((hl ^ swz) * 16) + (k & 15)) = lq;
}

__device__ unsigned long long g_clk[8192 * 4];
__device__ unsigned long long g_ph[1024 * 8 * 6];
constexpr int TF = 128, TT = 128, NTHR = 256;
constexpr int ROWB = 32;
constexpr int WSUB = TF * ROWB, XSUB = TT * ROWB;        // one image x one k-block = 4 KB
constexpr int SUBBLK = 2 * WSUB + 2 * XSUB;              // [WH][WI][XH][XI] of one k-block = 16 KB
constexpr int STAGE = 2 * SUBBLK;                        // BK = 32
constexpr int OFF_WH = 0, OFF_WI = WSUB, OFF_XH = 2 * WSUB, OFF_XI = 2 * WSUB + XSUB;

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// 128 (features) x 128 (tokens) x 32 tiles, 4 waves (2 x 2), wave tile 64 x 64, two workgroups per CU.  Operands are pre-split images, so
// staging is a plain copy: global -> registers (one stage ahead, under the current stage's math) -> LDS, double-buffered, one barrier per stage.
// OUT 0: fp32 C (T, N) (+ bias, + residual);  OUT 1: GELU(acc + bias) quantized into the NEXT GEMM's images (K' = N; scale chunk = this tile)
template <int OUT, bool HAS_R>
__global__ __launch_bounds__(NTHR, 2) void gemm_f16i8_kernel(const uint16_t* __restrict__ wh, const uint16_t* __restrict__ wi, const float* __restrict__ ws, int N,
                                                            const uint16_t* __restrict__ xh, const uint16_t* __restrict__ xi, const float* __restrict__ xs, int64_t T_pad,
                                                            const float* __restrict__ bias, int T, int K, int chunk_stages, const float* __restrict__ R, int64_t ldr,
                                                            float* __restrict__ C, int64_t ldc, uint16_t* __restrict__ oh, uint8_t* __restrict__ oi,
                                                            float* __restrict__ os, int tiles_f, int nwg) {
#ifdef PK_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int pid = xcd_remap(blockIdx.x, nwg);
    const int tt = pid / tiles_f, tf = pid - tt * tiles_f;
    const int f0 = tf * TF, t0 = tt * TT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave >> 1, wt = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging: per stage 8 units of 16 bytes per thread: (sub-block s, image o) x (row tid >> 1, half tid & 1); lane-linear on both sides
    const int srow = tid >> 1, shalf = tid & 1;
    const uint16_t* gsrc[4];
    gsrc[0] = wh + ((int64_t)(f0 + srow)) * 16 + shalf * 8;
    gsrc[1] = wi + ((int64_t)(f0 + srow)) * 16 + shalf * 8;
    gsrc[2] = xh + ((int64_t)(t0 + srow)) * 16 + shalf * 8;
    gsrc[3] = xi + ((int64_t)(t0 + srow)) * 16 + shalf * 8;
    const int64_t wblk = (int64_t)N * 16, xblk = T_pad * 16;  // uint16 elements per k-block
    const int sdst = tid * 16;
    // eight named staging registers (an array captured by the lambdas stays in scratch memory)
    uint4 st00, st01, st02, st03, st10, st11, st12, st13;
#define F8_GLOAD(kt_)                                                                   \
    do {                                                                                \
        const int64_t kb0_ = 2 * (int64_t)(kt_), kb1_ = kb0_ + 1;                       \
        st00 = *reinterpret_cast<const uint4*>(gsrc[0] + kb0_ * wblk);                  \
        st01 = *reinterpret_cast<const uint4*>(gsrc[1] + kb0_ * wblk);                  \
        st02 = *reinterpret_cast<const uint4*>(gsrc[2] + kb0_ * xblk);                  \
        st03 = *reinterpret_cast<const uint4*>(gsrc[3] + kb0_ * xblk);                  \
        st10 = *reinterpret_cast<const uint4*>(gsrc[0] + kb1_ * wblk);                  \
        st11 = *reinterpret_cast<const uint4*>(gsrc[1] + kb1_ * wblk);                  \
        st12 = *reinterpret_cast<const uint4*>(gsrc[2] + kb1_ * xblk);                  \
        st13 = *reinterpret_cast<const uint4*>(gsrc[3] + kb1_ * xblk);                  \
    } while (0)
#define F8_LSTORE(buf_)                                                                 \
    do {                                                                                \
        *reinterpret_cast<uint4*>((buf_) + OFF_WH + sdst) = st00;                       \
        *reinterpret_cast<uint4*>((buf_) + OFF_WI + sdst) = st01;                       \
        *reinterpret_cast<uint4*>((buf_) + OFF_XH + sdst) = st02;                       \
        *reinterpret_cast<uint4*>((buf_) + OFF_XI + sdst) = st03;                       \
        *reinterpret_cast<uint4*>((buf_) + SUBBLK + OFF_WH + sdst) = st10;              \
        *reinterpret_cast<uint4*>((buf_) + SUBBLK + OFF_WI + sdst) = st11;              \
        *reinterpret_cast<uint4*>((buf_) + SUBBLK + OFF_XH + sdst) = st12;              \
        *reinterpret_cast<uint4*>((buf_) + SUBBLK + OFF_XI + sdst) = st13;              \
    } while (0)

    // ---- fragment read offsets inside a sub-block (rows + 32 keep the swizzle bit)
    const int fsw = (lh ^ ((lr >> 3) & 1)) * 16;
    const int ra = (wf * 64 + lr) * ROWB + fsw;
    const int rb = (wt * 64 + lr) * ROWB + fsw;

    f32x16 accf[2][2];
    i32x16 acci[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accf[i][j][r] = 0.f; acci[i][j][r] = 0; }

    auto compute = [&](const unsigned char* buf) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* b = buf + s * SUBBLK;
            uint4 fwh[2], fwi[2], fxh[2], fxi[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fwh[i] = *reinterpret_cast<const uint4*>(b + OFF_WH + ra + i * 32 * ROWB);
                fxh[i] = *reinterpret_cast<const uint4*>(b + OFF_XH + rb + i * 32 * ROWB);
                fwi[i] = *reinterpret_cast<const uint4*>(b + OFF_WI + ra + i * 32 * ROWB);
                fxi[i] = *reinterpret_cast<const uint4*>(b + OFF_XI + rb + i * 32 * ROWB);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accf[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fwh[i]), __builtin_bit_cast(f16x8, fxh[j]), accf[i][j], 0, 0, 0);
                    acci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, fwi[i]), __builtin_bit_cast(i32x4, fxi[j]), acci[i][j], 0, 0, 0);
                }
        }
    };
    // fold the chunk's integer cross-term sums into the f32 accumulators: accf += acci * (sw[f, c] * sx[t, c] * 2^-11).  The scales are
    // fetched one stage early (with the last stage's prefetch), so the fold waits for nothing the staging loads have not waited for already.
    float sxl0 = 0.f, sxl1 = 0.f;
    float4 sw00, sw01, sw02, sw03, sw10, sw11, sw12, sw13;
    const float* xs_p = xs + t0 + wt * 64 + lr;
    const float* ws_p = ws + f0 + wf * 64 + 4 * lh;
#define F8_SCALES(c_)                                                                              \
    do {                                                                                           \
        sxl0 = xs_p[(int64_t)(c_) * T_pad];                                                        \
        sxl1 = xs_p[(int64_t)(c_) * T_pad + 32];                                                   \
        const float* w_ = ws_p + (int64_t)(c_) * N;                                                \
        sw00 = *reinterpret_cast<const float4*>(w_);      sw01 = *reinterpret_cast<const float4*>(w_ + 8);   \
        sw02 = *reinterpret_cast<const float4*>(w_ + 16); sw03 = *reinterpret_cast<const float4*>(w_ + 24);  \
        sw10 = *reinterpret_cast<const float4*>(w_ + 32); sw11 = *reinterpret_cast<const float4*>(w_ + 40);  \
        sw12 = *reinterpret_cast<const float4*>(w_ + 48); sw13 = *reinterpret_cast<const float4*>(w_ + 56);  \
    } while (0)
    auto fold1 = [&](int i, int q, const float4 sw) {
        const float swv[4] = {sw.x, sw.y, sw.z, sw.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accf[i][0][4 * q + r] = fmaf((float)acci[i][0][4 * q + r] * swv[r], sxl0 * LO_SCALE, accf[i][0][4 * q + r]);
            accf[i][1][4 * q + r] = fmaf((float)acci[i][1][4 * q + r] * swv[r], sxl1 * LO_SCALE, accf[i][1][4 * q + r]);
            acci[i][0][4 * q + r] = 0;
            acci[i][1][4 * q + r] = 0;
        }
    };
    auto fold = [&]() {
        fold1(0, 0, sw00); fold1(0, 1, sw01); fold1(0, 2, sw02); fold1(0, 3, sw03);
        fold1(1, 0, sw10); fold1(1, 1, sw11); fold1(1, 2, sw12); fold1(1, 3, sw13);
    };

#ifdef PK_PHASES
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tph = __builtin_amdgcn_s_memtime();
#define PH(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph[i] += n_ - tph; tph = n_; }
#else
#define PH(i)
#endif
    const int nst = K / 32;
    F8_GLOAD(0);
    F8_LSTORE(lds);
    __syncthreads();
    int in_chunk = 0, chunk = 0;
    PH(0)
    for (int kt = 0; kt < nst; ++kt) {
        unsigned char* cur = lds + (kt & 1) * STAGE;
        unsigned char* nxt = lds + ((kt + 1) & 1) * STAGE;
        F8_GLOAD(kt + 1 < nst ? kt + 1 : kt);  // unconditional prefetch (the last one re-reads the final stage, never stored to a buffer in use)
        if (in_chunk + 1 == chunk_stages) F8_SCALES(chunk);
        __builtin_amdgcn_sched_barrier(0);
        PH(1)
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        PH(2)
        F8_LSTORE(nxt);
        PH(3)
        if (++in_chunk == chunk_stages) { fold(); ++chunk; in_chunk = 0; }
        PH(4)
        __syncthreads();
        PH(5)
    }
#ifdef PK_PHASES
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g_ph[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 6 + i] = ph[i];
    }
#endif
#ifdef PK_CLOCK
    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_clk[blockIdx.x * 4] = clk_c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_clk[blockIdx.x * 4 + 2] = clk_r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- epilogue.  acc[i][j][r]: feature f0 + wf*64 + i*32 + (r & 3) + 8 (r >> 2) + 4 lh, token t0 + wt*64 + j*32 + lr
    if (OUT == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + wt * 64 + j * 32 + lr;
            const bool t_ok = t < T;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int fb_ = f0 + wf * 64 + i * 32 + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bz = *reinterpret_cast<const float4*>(bias + fb_ + 8 * q);
                    float4 v = make_float4(accf[i][j][4 * q] + bz.x, accf[i][j][4 * q + 1] + bz.y, accf[i][j][4 * q + 2] + bz.z, accf[i][j][4 * q + 3] + bz.w);
                    if (t_ok) {
                        if (HAS_R) {
                            const float4 rr = *reinterpret_cast<const float4*>(R + (int64_t)t * ldr + fb_ + 8 * q);
                            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        }
                        *reinterpret_cast<float4*>(C + (int64_t)t * ldc + fb_ + 8 * q) = v;
                    }
                }
            }
        }
    } else {
        // GELU, then quantize: the scale chunk of the next GEMM is this tile's 128 features: per-token max over the 2 feature waves through LDS
        float* smax = reinterpret_cast<float*>(lds);  // [2][128]; the stage buffers are dead after the loop's last barrier
        float mx[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bz = *reinterpret_cast<const float4*>(bias + f0 + wf * 64 + i * 32 + 8 * q + 4 * lh);
                    const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = gelu_erf(accf[i][j][4 * q + r] + bzv[r]);
                        accf[i][j][4 * q + r] = v;
                        mx[j] = fmaxf(mx[j], fabsf(v));
                    }
                }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], 32, 64));
            if (lh == 0) smax[wf * 128 + wt * 64 + j * 32 + lr] = mx[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tl = wt * 64 + j * 32 + lr, t = t0 + tl;
            const float m = fmaxf(smax[tl], smax[128 + tl]);
            const float s = m > 0.f ? m * (1.0f / 127.0f) : 1.0f;
            const bool t_ok = t < T;
            if (wf == 0 && lh == 0 && t < T_pad) os[(int64_t)tf * T_pad + t] = s;
            const float inv = 1.0f / s, inv_lo = inv * 2048.0f;
            const int swz = (t >> 3) & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {  // the 32-feature MFMA tile = two k-blocks of the next GEMM
                    uint32_t h[2][2], qv[2], lv[2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = hb * 2 + qq;
                        const float v0 = accf[i][j][4 * q], v1 = accf[i][j][4 * q + 1], v2 = accf[i][j][4 * q + 2], v3 = accf[i][j][4 * q + 3];
                        h[qq][0] = pack_f16(v0, v1); h[qq][1] = pack_f16(v2, v3);
                        qv[qq] = q8x4(v0 * inv, v1 * inv, v2 * inv, v3 * inv);
                        lv[qq] = q8x4((v0 - f16_lo(h[qq][0])) * inv_lo, (v1 - f16_hi(h[qq][0])) * inv_lo, (v2 - f16_lo(h[qq][1])) * inv_lo, (v3 - f16_hi(h[qq][1])) * inv_lo);
                    }
#pragma unroll
                    for (int dw = 0; dw < 2; ++dw) {
                        auto sw_ = __builtin_amdgcn_permlane32_swap(h[0][dw], h[1][dw], false, false);
                        h[0][dw] = sw_[0]; h[1][dw] = sw_[1];
                    }
                    auto sq = __builtin_amdgcn_permlane32_swap(qv[0], qv[1], false, false);
                    auto sl = __builtin_amdgcn_permlane32_swap(lv[0], lv[1], false, false);
                    const int kb = (f0 + wf * 64 + i * 32) / 16 + hb;
                    const int64_t rowb = ((int64_t)kb * T_pad + t) * 32;
                    if (t_ok) {
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(oh) + rowb + ((lh ^ swz) * 16)) = make_uint4(h[0][0], h[0][1], h[1][0], h[1][1]);
                        *reinterpret_cast<uint2*>(oi + rowb + ((0 ^ swz) * 16) + 8 * lh) = make_uint2(sq[0], sq[1]);
                        *reinterpret_cast<uint2*>(oi + rowb + ((1 ^ swz) * 16) + 8 * lh) = make_uint2(sl[0], sl[1]);
                    }
                }
        }
    }
}

// images -> fp32 (hi + lq * s * 2^-11) for checking the quantizing epilogue
__global__ void dequant_kernel(const uint16_t* __restrict__ ih, const uint8_t* __restrict__ ii, const float* __restrict__ sc, int T, int N, int64_t T_pad,
                               float* __restrict__ out, float* __restrict__ outq) {
    const int64_t total = (int64_t)T * N;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = u / N;
        const int k = (int)(u - t * N);
        const int swz = (int)((t >> 3) & 1);
        const int64_t rowb = ((int64_t)(k >> 4) * T_pad + t) * 32;
        const uint16_t hb = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(ih) + rowb + ((((k >> 3) & 1) ^ swz) * 16) + (k & 7) * 2);
        const float s = sc[(int64_t)(k / CHUNK) * T_pad + t];
        const int8_t q = (int8_t)ii[rowb + ((0 ^ swz) * 16) + (k & 15)], lq = (int8_t)ii[rowb + ((1 ^ swz) * 16) + (k & 15)];
        out[u] = (float)__builtin_bit_cast(_Float16, hb) + (float)lq * s * LO_SCALE;
        outq[u] = (float)q * s;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 69632;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    struct Shape { const char* name; int N, K, out; bool res; } shapes[] = {{"qkv", 2304, 768, 0, false}, {"out", 768, 768, 0, true}, {"ffn1", 3072, 768, 1, false}, {"ffn2", 768, 3072, 0, true}};
    const size_t LDS_BYTES = (size_t)2 * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16i8_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16i8_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16i8_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    srand(1);
    const int64_t T_pad = (M + TT - 1) / TT * TT;
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
        for (auto& x : hA) { float u = (float)rand() / RAND_MAX * 2.f - 1.f; x = u * u * u * 3.f; }   // heavier tails than uniform
        for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        for (auto& x : hb) x = (float)rand() / RAND_MAX;
        float *dA, *dW, *db, *dC, *dC2, *dR, *xs, *ws, *os, *dQ;
        uint16_t *wh, *wi, *xh, *xi, *oh, *lwh, *lwm, *lwl;
        uint8_t* oi;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4)); CK(hipMalloc(&dQ, (size_t)M * N * 4));
        CK(hipMalloc(&wh, hW.size() * 2)); CK(hipMalloc(&wi, hW.size() * 2)); CK(hipMalloc(&ws, (size_t)(K / CHUNK) * N * 4));
        CK(hipMalloc(&lwh, hW.size() * 2)); CK(hipMalloc(&lwm, hW.size() * 2)); CK(hipMalloc(&lwl, hW.size() * 2));
        CK(hipMalloc(&xh, (size_t)T_pad * K * 2)); CK(hipMalloc(&xi, (size_t)T_pad * K * 2)); CK(hipMalloc(&xs, (size_t)(K / CHUNK) * T_pad * 4));
        CK(hipMalloc(&oh, (size_t)T_pad * N * 2)); CK(hipMalloc(&oi, (size_t)T_pad * N * 2)); CK(hipMalloc(&os, (size_t)(N / CHUNK) * T_pad * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        CK(hipMemset(dR, 0, (size_t)M * N * 4));
        hipLaunchKernelGGL(quantize_rows_kernel<true>, dim3((unsigned)(((int64_t)N * (K / 256) + 3) / 4)), dim3(256), 0, 0, dW, N, K, (int64_t)N, wh, reinterpret_cast<uint8_t*>(wi), ws);
        hipEvent_t q0, q1; CK(hipEventCreate(&q0)); CK(hipEventCreate(&q1));
        CK(hipEventRecord(q0, 0));
        hipLaunchKernelGGL(quantize_rows_kernel<false>, dim3((unsigned)((T_pad * (K / 256) + 3) / 4)), dim3(256), 0, 0, dA, M, K, T_pad, xh, reinterpret_cast<uint8_t*>(xi), xs);
        CK(hipEventRecord(q1, 0)); CK(hipEventSynchronize(q1));
        float qms; CK(hipEventElapsedTime(&qms, q0, q1));
        {
            int64_t tab[3] = {0, N, K}, pref[2] = {0, (int64_t)N * K / 4};
            int64_t *dt, *dp;
            CK(hipMalloc(&dt, 24)); CK(hipMalloc(&dp, 16));
            CK(hipMemcpy(dt, tab, 24, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, pref, 16, hipMemcpyHostToDevice));
            int rc = mr_split_weights_kblock_f32(dW, dt, dp, 1, pref[1], lwh, lwm, lwl, 0);
            if (rc) { printf("split_weights rc %d\n", rc); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipFree(dt)); CK(hipFree(dp));
        }
        const int tiles_t = (M + TT - 1) / TT, tiles_f = N / TF, nwg = tiles_t * tiles_f;
        auto launch = [&] {
            if (sh.out == 1)
                hipLaunchKernelGGL((gemm_f16i8_kernel<1, false>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, wh, wi, ws, N, xh, xi, xs, T_pad, db, M, K, CHUNK / 32, nullptr, 0, nullptr, 0, oh, oi, os, tiles_f, nwg);
            else if (sh.res)
                hipLaunchKernelGGL((gemm_f16i8_kernel<0, true>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, wh, wi, ws, N, xh, xi, xs, T_pad, db, M, K, CHUNK / 32, dR, (int64_t)N, dC, (int64_t)N, nullptr, nullptr, nullptr, tiles_f, nwg);
            else
                hipLaunchKernelGGL((gemm_f16i8_kernel<0, false>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, wh, wi, ws, N, xh, xi, xs, T_pad, db, M, K, CHUNK / 32, nullptr, 0, dC, (int64_t)N, nullptr, nullptr, nullptr, tiles_f, nwg);
        };
        auto launch_lib = [&] {
            int rc = mr_gemm_nt_bf16x6_f32(dA, K, lwh, lwm, lwl, 0, 0, 0, db, nullptr, nullptr, 1, M, N, K, sh.out == 1 ? 1 : 0, sh.res ? dR : nullptr, N, dC2, N, 6, 0);
            if (rc) { printf("lib gemm rc %d\n", rc); exit(1); }
        };
        launch();
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        launch_lib();   // the six-product (fp32-grade) library kernel is the accuracy reference; the three-product one the speed reference
        CK(hipDeviceSynchronize());
        if (sh.out == 1) hipLaunchKernelGGL(dequant_kernel, dim3(4096), dim3(256), 0, 0, oh, oi, os, M, N, T_pad, dC, dQ);
        CK(hipDeviceSynchronize());
        std::vector<float> c1((size_t)M * N), c2((size_t)M * N), cq;
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, sq = 0, sref = 0, worstq = 0;
        for (size_t i = 0; i < c1.size(); ++i) {
            const double d = fabs((double)c1[i] - (double)c2[i]);
            if (!(d <= worst)) worst = d;
            sq += d * d; sref += (double)c2[i] * c2[i];
        }
        if (sh.out == 1) {
            cq.resize((size_t)M * N);
            CK(hipMemcpy(cq.data(), dQ, cq.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < cq.size(); ++i) { const double d = fabs((double)cq[i] - (double)c2[i]); if (!(d <= worstq)) worstq = d; }
        }
        auto launch_lib3 = [&] { mr_gemm_nt_bf16x6_f32(dA, K, lwh, lwm, lwl, 0, 0, 0, db, nullptr, nullptr, 1, M, N, K, sh.out == 1 ? 1 : 0, sh.res ? dR : nullptr, N, dC2, N, 3, 0); };
        std::vector<float> ts, tl;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            CK(hipEventRecord(e0, 0)); launch_lib3(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); tl.push_back(ms);
        }
        std::sort(ts.begin(), ts.end()); std::sort(tl.begin(), tl.end());
        const double fl = 2.0 * M * N * K;
        printf("f16i8 %-5s M=%d N=%d K=%d: %.3f ms %.1f TFLOP/s alg (best %.1f) | library bf16x3 %.3f ms %.1f TFLOP/s | speedup %.2fx | vs bf16x6: max abs diff %.3g, rel rms %.3g%s | quantize A %.3f ms\n",
               sh.name, M, N, K, ts[ts.size() / 2], fl / ts[ts.size() / 2] / 1e9, fl / ts[0] / 1e9, tl[tl.size() / 2], fl / tl[tl.size() / 2] / 1e9,
               tl[tl.size() / 2] / ts[ts.size() / 2], worst, sqrt(sq / sref), sh.out == 1 ? " (coarse q image max diff printed next)" : "", qms);
        if (sh.out == 1) printf("      coarse q image vs value: max abs diff %.3g\n", worstq);
#ifdef PK_PHASES
        {
            static unsigned long long hp[1024 * 8 * 6];
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_ph), sizeof(hp)));
            double acc[6] = {0, 0, 0, 0, 0, 0};
            const int nb = nwg < 1024 ? nwg : 1024;
            for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) for (int i = 0; i < 6; ++i) acc[i] += (double)hp[(b * 8 + w) * 6 + i];
            const double n = (double)nb * 4, st_ = K / 32;
            printf("      cycles per wave and stage: issue loads %.0f | compute (16 reads + 16 MFMA) %.0f | wait + LDS store %.0f | fold %.0f | barrier %.0f   (prologue %.0f; stages %d)\n",
                   acc[1] / n / st_, acc[2] / n / st_, acc[3] / n / st_, acc[4] / n / st_, acc[5] / n / st_, acc[0] / n, K / 32);
        }
#endif
#ifdef PK_CLOCK
        {
            static unsigned long long hc[8192 * 4];
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
            double cs = 0, rs = 0;
            const int nw = nwg < 8192 ? nwg : 8192;
            for (int w = 0; w < nw; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
            printf("      main-loop clock %.2f GHz\n", cs / rs * 0.1);
        }
#endif
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dC2); hipFree(dR); hipFree(dQ); hipFree(wh); hipFree(wi); hipFree(ws); hipFree(xh); hipFree(xi); hipFree(xs);
        hipFree(oh); hipFree(oi); hipFree(os); hipFree(lwh); hipFree(lwm); hipFree(lwl);
    }
    return 0;
}
