// EXPERIMENT (standalone bench): fp32-grade GEMM as TWO f16 MFMAs + ONE block-scaled MX-fp8 MFMA per 32x32x32 tile-step instead of six bf16 MFMAs.
//   x = xh + xl,  xh = fp16(x) (11 significant bits, round to nearest even),  |xl| <= 2^-11 |x|;  same for w.
//   x w = xh wh  +  (x wl + xl w)  +  O(2^-22):  the main product on v_mfma_f32_32x32x16_f16, BOTH cross terms in ONE
//   v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands): its 64-deep "virtual" K is the concatenation [x8 | xl8] . [wl8 | w8] over 32 real k,
//   block 0 (virtual k < 32) carrying x8 . wl8, block 1 carrying xl8 . w8, each with its own hardware-applied E8M0 scale per (row, block):
//   s = 2^(floor(log2 amax_32) - 7) for the 8-bit image of x (or w), s 2^-11 for the image of the fp16 residual -- one f32 accumulator serves
//   all three products.  Matrix-pipe cycles per 32 k and 32x32 tile: 2 x 32 + 64 = 128 against 6 x 32 = 192 (exp/mix_decomp_bench: the
//   same LOOP is 1.31-1.35x faster with this mix at equal data movement).
// Operand layout of the MX instruction (exp/mx_probe*.hip): lane (row l & 31, half h = l >> 5) holds 32 bytes; bytes 0-15 = virtual k 16 h + b
// (block 0), bytes 16-31 = virtual k 32 + 16 h + (b - 16) (block 1); the scale byte of lane (row, h) acts on block h of that row.
// Weights are pre-split once into tile images ([K/32][N/256] tiles of [4 planes][256 rows][16 B]: fp16 image, MX image, 2 scale bytes per row)
// and go global -> LDS by LDS-DMA; activations are split while they are staged (fp32 global -> registers -> split -> LDS), as in the library
// kernel.  256 x 256 x 32 tiles, 8 waves (4 x 2, wave tile 64 x 128 = 2 x 4 MFMA tiles), one workgroup per CU, two 66 KB LDS stages, one raw
// barrier per stage with counted waits (the activation reloads stay in flight across it).  -DMX_PINGPONG: the two waves of a SIMD take
// different roles half a stage apart (one stages while the other multiplies).  -DMX_ABL_*: timing ablations (wrong results).
// RESULT (profiles/r02_gemm_f16mx.txt): numerically as predicted (relative rms 9e-6 against the six-product kernel; bf16x3: 3.8e-6), 0.82-0.97x
// the library's bf16x3 kernel: with one 65 KB stage in flight per CU the loop waits on data (about 27 B/clk/CU delivered), not on the matrix pipe.
// build: hipcc --offload-arch=gfx950 -O3 -o exp/gemm_f16mx exp/gemm_f16mx.hip -L mergerec_amd/lib -lmergerec_hip -Wl,-rpath,$PWD/mergerec_amd/lib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../include/mergerec_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#ifndef SCALE_DIVIDES
#define SCALE_DIVIDES 1   // v_cvt_scalef32_pk_fp8_f32 computes fp8(src / scale) (checked by probe_cvt at start-up)
#endif

constexpr int TM = 256, TN = 256, BK = 32, NTHR = 512;
// LDS stage: four images (activations fp16 / MX, weights fp16 / MX) of 256 rows x 64 B each, stored as FOUR PLANES of [256 rows][16 B]: a
// wave's ds_read_b128 of one plane touches consecutive 16-byte slots (conflict-free without a swizzle, one lane base register per operand,
// everything else immediate offsets).  fp16 image: plane p holds k = 8 u .. 8 u + 7 with u = 2 (p & 1) + (p >> 1), so lane half lh reads plane
// 2 lh + s in k16 step s; MX image: plane 2 h + q = bytes 16 q .. 16 q + 15 of MX lane half h (q = 0: block 0, q = 1: block 1).  Planes are
// 64 B apart from a multiple of 256 B so the two halves of a staging wave's ds_write_b128 land on different banks.  Scales: [half][row] bytes.
constexpr int PS = 4096 + 64;             // plane stride
constexpr int IMG = 4 * PS;               // 16,640 B
constexpr int OFF_AH = 0, OFF_AM = IMG, OFF_BH = 2 * IMG, OFF_BM = 3 * IMG, OFF_AS = 4 * IMG, OFF_BS = 4 * IMG + 512;
constexpr int STAGE = 4 * IMG + 1024;     // 67,584 B; two of them per workgroup
constexpr int WTILE = 16384;              // bytes of one weight image tile (256 rows x 32 k) in global memory: [plane][row][16 B], unpadded

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// E8M0 scale bytes of a 32-element block with absolute maximum `amax`: the 8-bit image is scaled to [128, 256), the residual image by 2^-11 more
__device__ __forceinline__ void block_scales(float amax, uint32_t& bs, uint32_t& bl) {
    const int e = (int)(__float_as_uint(amax) >> 23);
    bs = (uint32_t)(e - 7 > 12 ? e - 7 : 12);
    bl = bs - 11;
}
// 16 consecutive k of one row -> fp16 image (2 x 16 B), 8-bit image of x (16 B), 8-bit image of x - fp16(x) (16 B)
__device__ __forceinline__ void split16(const float4 v0, const float4 v1, const float4 v2, const float4 v3, uint32_t bs, uint32_t bl, uint4& h0, uint4& h1,
                                        uint4& q, uint4& ql) {
    const float ss = __uint_as_float(bs << 23), sl = __uint_as_float(bl << 23);
    const float x[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    uint32_t hw[8], qw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x2 xv = {x[2 * p], x[2 * p + 1]};
        const f16x2 hv = __builtin_convertvector(xv, f16x2);
        hw[p] = __builtin_bit_cast(uint32_t, hv);
        const f32x2 bk = __builtin_convertvector(hv, f32x2);
        const float l0 = xv.x - bk.x, l1 = xv.y - bk.y;
        s16x2 oq = __builtin_bit_cast(s16x2, (p & 1) ? qw[p >> 1] : 0u), ol = __builtin_bit_cast(s16x2, (p & 1) ? lw[p >> 1] : 0u);
#if SCALE_DIVIDES
        oq = (p & 1) ? __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(oq, xv.x, xv.y, ss, true) : __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(oq, xv.x, xv.y, ss, false);
        ol = (p & 1) ? __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, l0, l1, sl, true) : __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, l0, l1, sl, false);
#else
        const float is = __uint_as_float((254u - bs) << 23), il = __uint_as_float((254u - bl) << 23);
        oq = (p & 1) ? __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(oq, xv.x, xv.y, is, true) : __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(oq, xv.x, xv.y, is, false);
        ol = (p & 1) ? __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, l0, l1, il, true) : __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, l0, l1, il, false);
#endif
        qw[p >> 1] = __builtin_bit_cast(uint32_t, oq);
        lw[p >> 1] = __builtin_bit_cast(uint32_t, ol);
    }
    h0 = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    h1 = make_uint4(hw[4], hw[5], hw[6], hw[7]);
    q = make_uint4(qw[0], qw[1], qw[2], qw[3]);
    ql = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}
__device__ __forceinline__ float amax16(const float4 v0, const float4 v1, const float4 v2, const float4 v3) {
    float m = fmaxf(fabsf(v0.x), fabsf(v0.y));
    m = fmaxf(fmaxf(fabsf(v0.z), fabsf(v0.w)), m);
    m = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), m);
    m = fmaxf(fmaxf(fabsf(v1.z), fabsf(v1.w)), m);
    m = fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), m);
    m = fmaxf(fmaxf(fabsf(v2.z), fabsf(v2.w)), m);
    m = fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), m);
    m = fmaxf(fmaxf(fabsf(v3.z), fabsf(v3.w)), m);
    return m;
}

// W (N, K) fp32 row-major -> per (k-block of 32, tile of 256 rows): wh [4 planes][256][16 B] fp16, wm [4 planes][256][16 B] (plane 2 h + 0: wl8 of
// k 16 h .. 16 h + 15 = block 0, which meets the activations' x8; plane 2 h + 1: w8 of the same k = block 1, which meets their xl8),
// wsc [2][256] bytes (half h: scale of block h).  One thread per (k-block, row, half): 16 elements.  N % 256 == 0.
// ACTS: the same images for an ACTIVATION matrix (rows = tokens): block 0 = x8 (scale s), block 1 = xl8 (scale s 2^-11).
template <bool ACTS>
__global__ __launch_bounds__(256) void split_weights_mx_kernel(const float* __restrict__ W, int N, int K, uint8_t* __restrict__ wh, uint8_t* __restrict__ wm,
                                                              uint8_t* __restrict__ wsc) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(K / 32) * N * 2;
    if (t >= total) return;  // total is a multiple of 2: both halves of a row are in or out together
    const int h = (int)(t & 1);
    const int64_t rk = t >> 1, kb = rk / N, n = rk - kb * N;
    const float* src = W + n * K + kb * 32 + h * 16;
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4), v2 = *reinterpret_cast<const float4*>(src + 8),
                 v3 = *reinterpret_cast<const float4*>(src + 12);
    float m = amax16(v0, v1, v2, v3);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    uint32_t bs, bl;
    block_scales(m, bs, bl);
    uint4 h0, h1, q, ql;
    split16(v0, v1, v2, v3, bs, bl, h0, h1, q, ql);
    const int64_t tile = kb * (N / 256) + n / 256;
    const int r = (int)(n & 255);
    uint8_t* th = wh + tile * WTILE + r * 16;
    *reinterpret_cast<uint4*>(th + (h) * 4096) = h0;       // unit 2 h     -> plane h
    *reinterpret_cast<uint4*>(th + (2 + h) * 4096) = h1;   // unit 2 h + 1 -> plane 2 + h
    uint8_t* tm = wm + tile * WTILE + r * 16;
    *reinterpret_cast<uint4*>(tm + (2 * h) * 4096) = ACTS ? q : ql;      // block 0: weights: residual image; activations: 8-bit image
    *reinterpret_cast<uint4*>(tm + (2 * h + 1) * 4096) = ACTS ? ql : q;  // block 1: weights: 8-bit image; activations: residual image
    wsc[tile * 512 + h * 256 + r] = (uint8_t)((h == 0) == ACTS ? bs : bl);
}

__device__ unsigned long long g_clk[4096 * 4];
__device__ unsigned long long g_ph[1024 * 8 * 6];

template <int ACT, bool HAS_R>
__global__ __launch_bounds__(NTHR, 1) void gemm_f16mx_kernel(const float* __restrict__ A, int64_t lda, const uint8_t* __restrict__ wh, const uint8_t* __restrict__ wm,
                                                            const uint8_t* __restrict__ wsc, const float* __restrict__ bias, int M, int N, int K,
                                                            const float* __restrict__ R, int64_t ldr, float* __restrict__ C, int64_t ldc, int tiles_n, int nwg,
                                                            const uint8_t* __restrict__ ah, const uint8_t* __restrict__ am, const uint8_t* __restrict__ asc, int tiles_m, int group_n) {
#ifdef MX_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int pid = xcd_remap(blockIdx.x, nwg);
    // column-group-major tile order: all row tiles of the first group_n column tiles, then the next group -- the group's weight images
    // (group_n x 0.75 MB at K = 768) stay in the XCD's L2 while the activation panels stream past once per group
    const int per_group = tiles_m * group_n;
    const int ng = pid / per_group, rem_ = pid - ng * per_group;
    const int gw = (tiles_n - ng * group_n) < group_n ? (tiles_n - ng * group_n) : group_n;
    const int tm = rem_ / gw, tn = ng * group_n + (rem_ - tm * gw);
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_ = wave >> 1, wn = wave & 1;  // 4 x 2 waves: 64 rows x 128 columns each
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging maps: thread -> (row srow of the tile, 16-k half sh)
    const int srow = tid >> 1, sh = tid & 1;
    int arow = m0 + srow;
    arow = arow < M ? arow : M - 1;
    const int wb_h = srow * 16 + sh * PS;        // fp16 image: unit 2 sh + q -> plane 2 q + sh
    const int wb_m = srow * 16 + sh * 2 * PS;    // MX image: plane 2 sh + q
    const int wb_s = sh * 256 + srow;
    // ---- fragment read bases: row slot + the lane half's plane pair; planes / tiles / images are immediate offsets
    const int ra = (wm_ * 64 + lr) * 16 + lh * 2 * PS, rb = (wn * 128 + lr) * 16 + lh * 2 * PS;
    const int rsa = lh * 256 + wm_ * 64 + lr, rsb = lh * 256 + wn * 128 + lr;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging.  Weights (pre-split tile images): LDS-DMA, 1 KB (64 rows of one plane) per wave instruction, two per image and wave.
    // Activations: fp32 rows -> registers (one named set, reloaded as soon as the split has consumed it) -> split -> LDS.
    float4 xa0, xa1, xa2, xa3;
    const float* abase = A + (int64_t)m0 * lda;                       // wave-uniform bases; lane parts are 32-bit offsets
    const uint32_t aoff = (uint32_t)(arow - m0) * (uint32_t)lda + sh * 16;
    const int64_t wblk = (int64_t)(N / 256) * WTILE, sblk = (int64_t)(N / 256) * 512;  // bytes per k-block
    const uint8_t* hbase = wh + (int64_t)tn * WTILE;
    const uint8_t* mbase = wm + (int64_t)tn * WTILE;
    const uint8_t* cbase = wsc + (int64_t)tn * 512;
    const int ch0 = wave, ch1 = wave + 8;                                                        // chunk = plane * 4 + quarter
    const int dl0 = (ch0 >> 2) * PS + (ch0 & 3) * 1024, dl1 = (ch1 >> 2) * PS + (ch1 & 3) * 1024; // LDS offsets inside an image (wave-uniform)
    const uint32_t dg0 = (uint32_t)ch0 * 1024 + lane * 16, dg1 = (uint32_t)ch1 * 1024 + lane * 16;
    const uint32_t dgs = (uint32_t)(wave & 1) * 256 + lane * 4;
    const int dls = (wave & 1) * 256;
#define MX_GLDS(gp_, lp_, sz_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp_), (__attribute__((address_space(3))) void*)(lp_), sz_, 0, 0)
#ifdef MX_ABL_NODMA
#define MX_DMA_B(buf_, kt_) do { } while (0)
#else
#define MX_DMA_B(buf_, kt_)                                                                  \
    do {                                                                                     \
        unsigned char* b_ = (buf_);                                                          \
        const uint8_t* ph_ = hbase + (int64_t)(kt_) * wblk;                                   \
        const uint8_t* pm_ = mbase + (int64_t)(kt_) * wblk;                                   \
        MX_GLDS(ph_ + dg0, b_ + OFF_BH + dl0, 16);                                           \
        MX_GLDS(ph_ + dg1, b_ + OFF_BH + dl1, 16);                                           \
        MX_GLDS(pm_ + dg0, b_ + OFF_BM + dl0, 16);                                           \
        MX_GLDS(pm_ + dg1, b_ + OFF_BM + dl1, 16);                                           \
        MX_GLDS(cbase + (int64_t)(kt_) * sblk + dgs, b_ + OFF_BS + dls, 4);                   \
    } while (0)
#endif
    // -DMX_APRESPLIT: the activations arrive as images too (their producers would write them) and take the same DMA path; no staging registers,
    // no split, no LDS stores
    const int64_t ablk = (int64_t)tiles_m * WTILE, asblk = (int64_t)tiles_m * 512;
    const uint8_t* ahbase = ah + (int64_t)tm * WTILE;
    const uint8_t* ambase = am + (int64_t)tm * WTILE;
    const uint8_t* acbase = asc + (int64_t)tm * 512;
#ifdef MX_ABL_NODMA
#define MX_DMA_A(buf_, kt_) do { } while (0)
#else
#define MX_DMA_A(buf_, kt_)                                                                  \
    do {                                                                                     \
        unsigned char* b_ = (buf_);                                                          \
        const uint8_t* ph_ = ahbase + (int64_t)(kt_) * ablk;                                  \
        const uint8_t* pm_ = ambase + (int64_t)(kt_) * ablk;                                  \
        MX_GLDS(ph_ + dg0, b_ + OFF_AH + dl0, 16);                                           \
        MX_GLDS(ph_ + dg1, b_ + OFF_AH + dl1, 16);                                           \
        MX_GLDS(pm_ + dg0, b_ + OFF_AM + dl0, 16);                                           \
        MX_GLDS(pm_ + dg1, b_ + OFF_AM + dl1, 16);                                           \
        MX_GLDS(acbase + (int64_t)(kt_) * asblk + dgs, b_ + OFF_AS + dls, 4);                 \
    } while (0)
#endif
#ifdef MX_ABL_NOALOAD
#define MX_GLOAD_A(kt_) do { if ((kt_) < 0) { xa0.x += 1.f; } } while (0)
#else
#define MX_GLOAD_A(kt_)                                                                      \
    do {                                                                                     \
        const float* pa_ = abase + (int64_t)(kt_) * 32 + aoff;                                \
        xa0 = *reinterpret_cast<const float4*>(pa_);                                         \
        xa1 = *reinterpret_cast<const float4*>(pa_ + 4);                                     \
        xa2 = *reinterpret_cast<const float4*>(pa_ + 8);                                     \
        xa3 = *reinterpret_cast<const float4*>(pa_ + 12);                                    \
    } while (0)
#endif
#ifdef MX_ABL_NOSPLIT
#define MX_SPLIT_A() uint32_t bs_ = 127, bl_ = 127; uint4 h0_ = __builtin_bit_cast(uint4, xa0), h1_ = __builtin_bit_cast(uint4, xa1), q_ = __builtin_bit_cast(uint4, xa2), ql_ = __builtin_bit_cast(uint4, xa3);
#else
#define MX_SPLIT_A()                                                                         \
    float amx_ = amax16(xa0, xa1, xa2, xa3);                                                  \
    amx_ = fmaxf(amx_, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, amx_), 0xB1, 0xf, 0xf, true))); /* quad_perm [1,0,3,2] */ \
    uint32_t bs_, bl_;                                                                       \
    block_scales(amx_, bs_, bl_);                                                            \
    uint4 h0_, h1_, q_, ql_;                                                                 \
    split16(xa0, xa1, xa2, xa3, bs_, bl_, h0_, h1_, q_, ql_);
#endif
#ifdef MX_ABL_NOLDSW
#define MX_LSTORE_A(buf_) asm volatile("" ::"v"(h0_.x), "v"(h0_.w), "v"(h1_.x), "v"(h1_.w), "v"(q_.x), "v"(q_.w), "v"(ql_.x), "v"(ql_.w), "v"(bs_), "v"(bl_))
#else
#define MX_LSTORE_A(buf_)                                                                    \
    do {                                                                                     \
        unsigned char* b_ = (buf_);                                                          \
        *reinterpret_cast<uint4*>(b_ + OFF_AH + wb_h) = h0_;                                 \
        *reinterpret_cast<uint4*>(b_ + OFF_AH + wb_h + 2 * PS) = h1_;                        \
        *reinterpret_cast<uint4*>(b_ + OFF_AM + wb_m) = q_;       /* block 0: x8 */          \
        *reinterpret_cast<uint4*>(b_ + OFF_AM + wb_m + PS) = ql_; /* block 1: xl8 */         \
        b_[OFF_AS + wb_s] = (unsigned char)(sh == 0 ? bs_ : bl_);                            \
    } while (0)
#endif
#define MX_PACK8(d_, q0_, q1_) d_[0] = q0_.x; d_[1] = q0_.y; d_[2] = q0_.z; d_[3] = q0_.w; d_[4] = q1_.x; d_[5] = q1_.y; d_[6] = q1_.z; d_[7] = q1_.w;

#ifdef MX_ABL_NOMX
#define MX_MFMA_MX(i, j) asm volatile("" ::"v"(am_[i]), "v"(bm_[j]), "v"(sca_[i]), "v"(scb_[j]));
#else
#define MX_MFMA_MX(i, j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am_[i], bm_[j], acc[i][j], 0, 0, 0, sca_[i], 0, scb_[j]);
#endif
#ifdef MX_ABL_NOH
#define MX_MFMA_H(a_, b_, i, j) asm volatile("" ::"v"(a_[i]), "v"(b_[j]));
#else
#define MX_MFMA_H(a_, b_, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[i], b_[j], acc[i][j], 0, 0, 0);
#endif
#define MX_COMPUTE(bufc_)                                                                                                  \
    do {                                                                                                                   \
        const unsigned char* c_ = (bufc_);                                                                                 \
        i32x8 am_[2], bm_[4];                                                                                              \
        int sca_[2], scb_[4];                                                                                              \
        f16x8 a0_[2], b0_[4], a1_[2], b1_[4];                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                    \
            const uint4 q0 = *reinterpret_cast<const uint4*>(c_ + OFF_AM + ra + i * 512);                                  \
            const uint4 q1 = *reinterpret_cast<const uint4*>(c_ + OFF_AM + ra + i * 512 + PS);                             \
            MX_PACK8(am_[i], q0, q1)                                                                                       \
            sca_[i] = c_[OFF_AS + rsa + i * 32];                                                                           \
        }                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            const uint4 q0 = *reinterpret_cast<const uint4*>(c_ + OFF_BM + rb + j * 512);                                  \
            const uint4 q1 = *reinterpret_cast<const uint4*>(c_ + OFF_BM + rb + j * 512 + PS);                             \
            MX_PACK8(bm_[j], q0, q1)                                                                                       \
            scb_[j] = c_[OFF_BS + rsb + j * 32];                                                                           \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a0_[i] = *reinterpret_cast<const f16x8*>(c_ + OFF_AH + ra + i * 512); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) b0_[j] = *reinterpret_cast<const f16x8*>(c_ + OFF_BH + rb + j * 512); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
                MX_MFMA_MX(i, j)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a1_[i] = *reinterpret_cast<const f16x8*>(c_ + OFF_AH + ra + i * 512 + PS); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) b1_[j] = *reinterpret_cast<const f16x8*>(c_ + OFF_BH + rb + j * 512 + PS); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) MX_MFMA_H(a0_, b0_, i, j)                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) MX_MFMA_H(a1_, b1_, i, j)                                        \
    } while (0)

#define MX_COMPUTE_SPLIT(bufc_, bufn_)                                                                                                  \
    do {                                                                                                                   \
        const unsigned char* c_ = (bufc_);                                                                                 \
        i32x8 am_[2], bm_[4];                                                                                              \
        int sca_[2], scb_[4];                                                                                              \
        f16x8 a0_[2], b0_[4], a1_[2], b1_[4];                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                    \
            const uint4 q0 = *reinterpret_cast<const uint4*>(c_ + OFF_AM + ra + i * 512);                                  \
            const uint4 q1 = *reinterpret_cast<const uint4*>(c_ + OFF_AM + ra + i * 512 + PS);                             \
            MX_PACK8(am_[i], q0, q1)                                                                                       \
            sca_[i] = c_[OFF_AS + rsa + i * 32];                                                                           \
        }                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            const uint4 q0 = *reinterpret_cast<const uint4*>(c_ + OFF_BM + rb + j * 512);                                  \
            const uint4 q1 = *reinterpret_cast<const uint4*>(c_ + OFF_BM + rb + j * 512 + PS);                             \
            MX_PACK8(bm_[j], q0, q1)                                                                                       \
            scb_[j] = c_[OFF_BS + rsb + j * 32];                                                                           \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a0_[i] = *reinterpret_cast<const f16x8*>(c_ + OFF_AH + ra + i * 512); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) b0_[j] = *reinterpret_cast<const f16x8*>(c_ + OFF_BH + rb + j * 512); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
                MX_MFMA_MX(i, j)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a1_[i] = *reinterpret_cast<const f16x8*>(c_ + OFF_AH + ra + i * 512 + PS); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) b1_[j] = *reinterpret_cast<const f16x8*>(c_ + OFF_BH + rb + j * 512 + PS); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) MX_MFMA_H(a0_, b0_, i, j)                                        \
        MX_SPLIT_A()                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) MX_MFMA_H(a1_, b1_, i, j)                                        \
        MX_LSTORE_A(bufn_);                                                                                                \
    } while (0)

    // Ping-pong roles.  The two waves of a SIMD (wave w and w + 4) run half a stage out of phase so that one's staging work (split of the next
    // stage's activations, LDS stores, load issue) runs under the other's MFMAs, with every load in flight for about a whole stage:
    //   role X (waves 0-3): DMA + activation loads of stage kt + 1 | fragment reads + 24 MFMAs of stage kt | split + LDS stores | barrier
    //   role Y (waves 4-7): split + LDS stores of stage kt + 1 (loaded during stage kt - 1) | DMA, activation loads of stage kt + 2 |
    //                       fragment reads + 24 MFMAs of stage kt | barrier (the activation loads stay in flight across it)
#ifdef MX_PHASES
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime();
#define MX_PH(i_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_[i_] += n_ - pt_; pt_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MX_PH(i_) do { } while (0)
#endif
#define MX_STAGE_X(bufc_, bufn_, ktn_)                                                       \
    do {                                                                                     \
        MX_DMA_B(bufn_, ktn_);                                                               \
        MX_GLOAD_A(ktn_);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        MX_PH(0);                                                                            \
        MX_COMPUTE(bufc_);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        MX_PH(1);                                                                            \
        {                                                                                    \
            MX_SPLIT_A()                                                                     \
            MX_LSTORE_A(bufn_);                                                              \
            MX_PH(2);                                                                        \
        }                                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        MX_PH(3);                                                                            \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        MX_PH(4);                                                                            \
    } while (0)
#define MX_STAGE_Y(bufc_, bufn_, ktn_, kta_)                                                 \
    do {                                                                                     \
        {                                                                                    \
            MX_SPLIT_A()                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                               \
            MX_DMA_B(bufn_, ktn_);                                                           \
            MX_LSTORE_A(bufn_);                                                              \
        }                                                                                    \
        MX_GLOAD_A(kta_);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        MX_PH(0);                                                                            \
        MX_COMPUTE(bufc_);                                                                   \
        MX_PH(1);                                                                            \
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                          \
        MX_PH(3);                                                                            \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        MX_PH(4);                                                                            \
    } while (0)

    // Default (one role): this stage's MFMAs with the split of the next stage's activations between the two fp16 steps (the compiler spreads it
    // over the fp16 MFMAs), their LDS stores and the reload of the staging registers behind it; the reloads stay in flight across the barrier.
#define MX_STAGE(bufc_, bufn_, ktn_, kta_)                                                   \
    do {                                                                                     \
        MX_DMA_B(bufn_, ktn_);                                                               \
        MX_COMPUTE_SPLIT(bufc_, bufn_);                                                      \
        MX_GLOAD_A(kta_);                                                                    \
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)

    const int nk = K / BK;  // even (host-checked)
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + STAGE;
    const int role = wave >> 2;
    {
        MX_GLOAD_A(0);
        MX_DMA_B(buf0, 0);
        MX_SPLIT_A()
        MX_LSTORE_A(buf0);
        MX_GLOAD_A(1);   // role Y starts with stage 1 in its registers; role X reloads it (an L2 hit) at the top of its first stage
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef MX_APRESPLIT
    (void)role;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // (the prologue above filled buf0's activation images through the split path: overwrite them consistently)
    MX_DMA_A(buf0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; kt += 2) {
        const int k2 = kt + 2 < nk ? kt + 2 : 0;
        MX_DMA_A(buf1, kt + 1);
        MX_DMA_B(buf1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        MX_COMPUTE(buf0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        MX_DMA_A(buf0, k2);
        MX_DMA_B(buf0, k2);
        __builtin_amdgcn_sched_barrier(0);
        MX_COMPUTE(buf1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#elif !defined(MX_PINGPONG)
    (void)role;
    for (int kt = 0; kt < nk; kt += 2) {   // past-the-end prefetches re-read stage 0 and are never consumed
        const int k2 = kt + 2 < nk ? kt + 2 : 0, k3 = kt + 3 < nk ? kt + 3 : 0;
        MX_STAGE(buf0, buf1, kt + 1, k2);
        MX_STAGE(buf1, buf0, k2, k3);
    }
#else
    if (role == 0) {
        for (int kt = 0; kt < nk; kt += 2) {
            const int k2 = kt + 2 < nk ? kt + 2 : 0;
            MX_STAGE_X(buf0, buf1, kt + 1);
            MX_STAGE_X(buf1, buf0, k2);
        }
    } else {
        for (int kt = 0; kt < nk; kt += 2) {
            const int k2 = kt + 2 < nk ? kt + 2 : 0, k3 = kt + 3 < nk ? kt + 3 : 0;
            MX_STAGE_Y(buf0, buf1, kt + 1, k2);
            MX_STAGE_Y(buf1, buf0, k2, k3);
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MX_PHASES
    if (lane == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 6; ++i) g_ph[(blockIdx.x * 8 + wave) * 6 + i] = ph_[i];
    }
#endif

    // ---- epilogue: as the library kernel (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
    // tile-local buffer resources drop rows past M
    const int rows_valid = (M - m0) < TM ? (M - m0) : TM;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C + (int64_t)m0 * ldc + n0, 0, (int)(((int64_t)(rows_valid - 1) * ldc + TN) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_R ? R + (int64_t)m0 * ldr + n0 : C), 0,
                                                                         HAS_R ? (int)(((int64_t)(rows_valid - 1) * ldr + TN) * 4) : 0, 0x00020000);
    int lr_e = lr, lh_e = lh;
    asm volatile("" : "+v"(lr_e), "+v"(lh_e));
    const uint32_t ldc4 = (uint32_t)ldc * 4u, ldr4 = (uint32_t)ldr * 4u;
    float bz[4];
    uint32_t coff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int colt = wn * 128 + j * 32 + lr_e;
        bz[j] = bias ? bias[n0 + colt] : 0.f;
        coff[j] = (uint32_t)colt * 4u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t rowt = wm_ * 64 + i * 32 + 4 * lh_e;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4], rr[4];
                if (HAS_R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (rowt + r + 8 * q) * ldr4 + coff[j], 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][4 * q + r] + bz[j];
                    if (ACT == 1) v[r] = gelu_erf(v[r]);
                    if (HAS_R) v[r] += rr[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[r]), crs, (rowt + r + 8 * q) * ldc4 + coff[j], 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef MX_CLOCK
    if (tid == 0 && blockIdx.x < 4096) {
        g_clk[blockIdx.x * 4] = clk_c0; g_clk[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        g_clk[blockIdx.x * 4 + 2] = clk_r0; g_clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// start-up probe: does v_cvt_scalef32_pk_fp8_f32 divide by the scale?  fp8(3.0 / 2.0) = 1.5 = 0x3c; fp8(3.0 * 2.0) = 6.0 = 0x4c
__global__ void probe_cvt(uint32_t* out) {
    s16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, 3.0f, 3.0f, 2.0f, false);
    out[0] = __builtin_bit_cast(uint32_t, o);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 69632;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    {
        uint32_t* d; uint32_t h = 0;
        CK(hipMalloc(&d, 4));
        hipLaunchKernelGGL(probe_cvt, dim3(1), dim3(1), 0, 0, d);
        CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
        const bool divides = (h & 0xff) == 0x3c;
        printf("v_cvt_scalef32_pk_fp8_f32(3.0, scale 2.0) -> 0x%02x: %s by the scale (built for SCALE_DIVIDES=%d)\n", h & 0xff, divides ? "divides" : "multiplies", SCALE_DIVIDES);
        if (divides != (SCALE_DIVIDES != 0)) { printf("rebuild with -DSCALE_DIVIDES=%d\n", divides ? 1 : 0); return 2; }
        CK(hipFree(d));
    }
    struct Shape { const char* name; int N, K, act; bool res; } shapes[] = {{"qkv", 2304, 768, 0, false}, {"out", 768, 768, 0, true}, {"ffn1", 3072, 768, 1, false}, {"ffn2", 768, 3072, 0, true}};
    const size_t LDS_BYTES = (size_t)2 * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16mx_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16mx_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16mx_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    srand(1);
    const char* only = argc > 3 ? argv[3] : nullptr;
    for (auto& sh : shapes) {
        if (only && !strstr(only, sh.name)) continue;
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hR((size_t)M * N);
        for (auto& x : hA) { float u = (float)rand() / (float)RAND_MAX * 2.f - 1.f; x = u * u * u * 3.f; }   // heavier tails than uniform
        for (auto& x : hW) x = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f;
        for (auto& x : hb) x = (float)rand() / (float)RAND_MAX;
        for (auto& x : hR) x = (float)rand() / (float)RAND_MAX - 0.5f;
        float *dA, *dW, *db, *dC, *dC2, *dC3, *dR;
        uint8_t *wh, *wm, *wsc;
        uint16_t *lwh, *lwm, *lwl;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4)); CK(hipMalloc(&dC3, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4));
        CK(hipMalloc(&wh, hW.size() * 2)); CK(hipMalloc(&wm, hW.size() * 2)); CK(hipMalloc(&wsc, (size_t)(K / 32) * N * 2));
        CK(hipMalloc(&lwh, hW.size() * 2)); CK(hipMalloc(&lwm, hW.size() * 2)); CK(hipMalloc(&lwl, hW.size() * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        hipLaunchKernelGGL(split_weights_mx_kernel<false>, dim3((unsigned)(((int64_t)(K / 32) * N * 2 + 255) / 256)), dim3(256), 0, 0, dW, N, K, wh, wm, wsc);
        uint8_t *ah = nullptr, *am = nullptr, *asc = nullptr;
#ifdef MX_APRESPLIT
        if (M % 256) { printf("MX_APRESPLIT needs M %% 256 == 0\n"); return 1; }
        CK(hipMalloc(&ah, hA.size() * 2)); CK(hipMalloc(&am, hA.size() * 2)); CK(hipMalloc(&asc, (size_t)(K / 32) * M * 2));
        hipEvent_t q0, q1; CK(hipEventCreate(&q0)); CK(hipEventCreate(&q1));
        CK(hipEventRecord(q0, 0));
        hipLaunchKernelGGL(split_weights_mx_kernel<true>, dim3((unsigned)(((int64_t)(K / 32) * M * 2 + 255) / 256)), dim3(256), 0, 0, dA, M, K, ah, am, asc);
        CK(hipEventRecord(q1, 0)); CK(hipEventSynchronize(q1));
        float qms; CK(hipEventElapsedTime(&qms, q0, q1));
        printf("      (activation images by a separate pass: %.3f ms)\n", qms);
#endif
        CK(hipGetLastError());
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
        if ((K / BK) % 2 || N % TN) { printf("shape not supported\n"); return 1; }
        const int tiles_m = (M + TM - 1) / TM, tiles_n = N / TN, nwg = tiles_m * tiles_n;
        const int group_arg = argc > 4 ? atoi(argv[4]) : 0;
        const int group_n = group_arg > 0 && group_arg < tiles_n ? group_arg : tiles_n;
        auto launch = [&] {
            if (sh.act == 1)
                hipLaunchKernelGGL((gemm_f16mx_kernel<1, false>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, dA, (int64_t)K, wh, wm, wsc, db, M, N, K, nullptr, 0, dC, (int64_t)N, tiles_n, nwg, ah, am, asc, tiles_m, group_n);
            else if (sh.res)
                hipLaunchKernelGGL((gemm_f16mx_kernel<0, true>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, dA, (int64_t)K, wh, wm, wsc, db, M, N, K, dR, (int64_t)N, dC, (int64_t)N, tiles_n, nwg, ah, am, asc, tiles_m, group_n);
            else
                hipLaunchKernelGGL((gemm_f16mx_kernel<0, false>), dim3(nwg), dim3(NTHR), LDS_BYTES, 0, dA, (int64_t)K, wh, wm, wsc, db, M, N, K, nullptr, 0, dC, (int64_t)N, tiles_n, nwg, ah, am, asc, tiles_m, group_n);
        };
        auto launch_lib = [&](int products, float* out) {
            int rc = mr_gemm_nt_bf16x6_f32(dA, K, lwh, lwm, lwl, 0, 0, 0, db, nullptr, nullptr, 1, M, N, K, sh.act, sh.res ? dR : nullptr, N, out, N, products, 0);
            if (rc) { printf("lib gemm rc %d\n", rc); exit(1); }
        };
        launch();
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        launch_lib(6, dC2);   // the six-product (fp32-grade) library kernel is the accuracy reference; the three-product one the speed reference
        launch_lib(3, dC3);
        CK(hipDeviceSynchronize());
        std::vector<float> c1((size_t)M * N), c2((size_t)M * N), c3((size_t)M * N);
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c3.data(), dC3, c3.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, sq = 0, sref = 0, worst3 = 0, sq3 = 0;
        size_t nbad = 0;
        for (size_t i = 0; i < c1.size(); ++i) {
            const double d = fabs((double)c1[i] - (double)c2[i]), d3 = fabs((double)c3[i] - (double)c2[i]);
            if (!(d <= worst)) worst = d;
            if (!(d3 <= worst3)) worst3 = d3;
            if (!(d <= 1e-2)) ++nbad;
            sq += d * d; sq3 += d3 * d3; sref += (double)c2[i] * c2[i];
        }
        // fp64 reference on a few rows
        double w64 = 0, w64_3 = 0;
        for (int s = 0; s < 8; ++s) {
            const int m = (int)(((int64_t)s * 7919 + 13) % M);
            for (int n = 0; n < N; n += 7) {
                double a = 0;
                for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                a += hb[n];
                if (sh.act == 1) a = 0.5 * a * (1.0 + erf(a * 0.70710678118654752440));
                if (sh.res) a += hR[(size_t)m * N + n];
                w64 = std::max(w64, fabs(a - (double)c1[(size_t)m * N + n]));
                w64_3 = std::max(w64_3, fabs(a - (double)c3[(size_t)m * N + n]));
            }
        }
        std::vector<float> ts, tl;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            CK(hipEventRecord(e0, 0)); launch_lib(3, dC3); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); tl.push_back(ms);
        }
        std::sort(ts.begin(), ts.end()); std::sort(tl.begin(), tl.end());
        const double fl = 2.0 * M * N * K;
        printf("f16mx %-5s (group_n %d) M=%d N=%d K=%d: %.3f ms %.1f TFLOP/s alg (best %.1f) | library bf16x3 %.3f ms %.1f TFLOP/s | speedup %.2fx\n", sh.name, group_n, M, N, K,
               ts[ts.size() / 2], fl / ts[ts.size() / 2] / 1e9, fl / ts[0] / 1e9, tl[tl.size() / 2], fl / tl[tl.size() / 2] / 1e9, tl[tl.size() / 2] / ts[ts.size() / 2]);
        printf("      vs library bf16x6: max abs diff %.3g, rel rms %.3g (bf16x3: %.3g, %.3g); elements off by > 1e-2: %zu; vs fp64 on sampled rows: max abs %.3g (bf16x3 %.3g)\n",
               worst, sqrt(sq / sref), worst3, sqrt(sq3 / sref), nbad, w64, w64_3);
#ifdef MX_PHASES
        {
            static unsigned long long hp[1024 * 8 * 6];
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_ph), sizeof(hp)));
            const int nb = nwg < 1024 ? nwg : 1024;
            for (int role = 0; role < 2; ++role) {
                double a[6] = {0, 0, 0, 0, 0, 0};
                for (int b = 0; b < nb; ++b) for (int w = role * 4; w < role * 4 + 4; ++w) for (int i = 0; i < 6; ++i) a[i] += (double)hp[(b * 8 + w) * 6 + i];
                const double n = (double)nb * 4 * (K / 32);
                printf("      role %c, s_memtime ticks (100 MHz) per wave and stage: staging/issue %.1f | reads + MFMA issue %.1f | split + stores %.1f | wait %.1f | barrier %.1f   (sum %.1f = %.0f ns)\n",
                       role ? 'Y' : 'X', a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, (a[0] + a[1] + a[2] + a[3] + a[4]) / n, (a[0] + a[1] + a[2] + a[3] + a[4]) / n * 10);
            }
        }
#endif
#ifdef MX_CLOCK
        {
            static unsigned long long hc[4096 * 4];
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
            double cs = 0, rs = 0;
            const int nw = nwg < 4096 ? nwg : 4096;
            for (int w = 0; w < nw; ++w) { cs += (double)(hc[w * 4 + 1] - hc[w * 4]); rs += (double)(hc[w * 4 + 3] - hc[w * 4 + 2]); }
            const double ghz = cs / rs * 0.1;
            const double mfma_cyc = (double)nwg * 8 * (K / 32) * (8 * 64.0 + 16 * 32.0) / (256.0 * 4);
            printf("      in-kernel clock %.2f GHz; matrix pipe busy %.0f %% of the kernel time at that clock\n", ghz, 100.0 * mfma_cyc / (ghz * 1e9) / (ts[ts.size() / 2] * 1e-3));
        }
#endif
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dC2); hipFree(dC3); hipFree(dR); hipFree(wh); hipFree(wm); hipFree(wsc); hipFree(lwh); hipFree(lwm); hipFree(lwl);
    }
    return 0;
}
