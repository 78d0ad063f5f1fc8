// K3': fp32-accurate GEMM on the bf16 matrix cores ("bf16x6" split precision).
//
// Every fp32 operand x is written as x = hi + mid + lo with three bf16 pieces (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); the two subtractions are exact in fp32), and a*b is evaluated as the six products
// hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid (all terms >= 2^-24 relative), each an exact bf16*bf16 product
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Per 32x32x16 tile-step that is 6 MFMAs x 32 clk = 192 clk
// against 8 x 64 = 512 clk for v_mfma_f32_32x32x2_f32: 2.67x fewer matrix-pipe cycles at fp32-grade accuracy
// (measured: 12-layer BLaIR-base embeddings within 2e-7 of the fp32 path; tolerance of the path is 1e-4).
// Weights are pre-split once per merge into three bf16 arenas in a K-BLOCKED layout (mr_split_weights_kblock_f32:
// matrix (N, K) at arena offset off is stored as [K/16][N][16] starting at the same offset), so the B tile a workgroup
// needs per k-step (128 rows x 16 k) is ONE contiguous 4 KB chunk per piece -- full cache lines instead of 32-byte row
// fragments (the row-major form ran at half the speed: global-load throughput bound).  Activations are split while
// they are staged global -> registers -> LDS.  Structure otherwise as gemm.hip: 128x128x16 block tile, 4 waves
// (2x2), wave tile 64x64 = 2x2 MFMA tiles, double-buffered LDS, one barrier per k-tile, prefetch pinned ahead of the
// MFMA block.  LDS rows are 16 bf16 (32 B, unpadded) with an XOR swizzle of the 16-B halves: conflict-free b128 reads,
// 48 KB of LDS per workgroup (double-buffered) -> 3 workgroups per CU.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

// phase-timing hooks: empty in the library; exp/gemm_phases.hip defines them to accumulate s_memtime deltas per phase
#ifndef MR_PH_DECL
#define MR_PH_DECL
#define MR_PH(i)
#define MR_PH_FLUSH(pid)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- "f16x3" (r04): the same three products hi*hi + hi*lo + lo*hi with FP16 pieces.  fp16 carries 11 significand bits, so two pieces hold
// 22 (bf16: 16) and the dropped lo*lo term is 2^-22 (2^-16): per-product error ~2^-21 at the matrix-pipe cost of bf16x3.  Range instead of
// precision is the price: |x| must stay below 65504 (activations of a post-LayerNorm encoder are far inside; weights are stored scaled by
// kF16WScale = 2^8 -- exact -- so that the low piece of a typical 0.01..0.1 weight is a NORMAL fp16 number, and the GEMM epilogue multiplies
// the accumulator by 2^-8).  Low pieces below 2^-14 are fp16 subnormals, which v_cvt_pk_f16_f32 produces and the matrix pipe honours
// exactly (exp/f16_subnormal_probe.hip): a piece pair then represents x to an ABSOLUTE 2^-25, i.e. better than fp32's own ulp for |x| > 0.25.
constexpr float kF16WScale = 256.0f, kF16WScaleInv = 1.0f / 256.0f;

constexpr int BM = 128, BK = 16;
constexpr int ROWB = 32;                 // bytes per LDS row: 16 bf16, unpadded; the two 16-B halves of rows 8..15 (mod 16)
                                         // are swapped (chunk ^= (row >> 3) & 1) so a ds_read_b128 of 16 rows is conflict-free
constexpr int PIECE = 128 * ROWB;        // bytes per 128-row piece tile
constexpr int kThreads = 256;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// two fp32 -> one dword of two bf16 (round-to-nearest-even), low half = first argument
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// two fp32 -> one dword of two fp16 (round-to-nearest-even), low half = first argument
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_h(uint32_t u) {
    float r;
    asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(u));  // reads the low 16 bits
    return r;
}
__device__ __forceinline__ float hi_h(uint32_t u) {
    float r;
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(u));  // the high 16 bits, no shift
    return r;
}
// split 4 fp32 into two pieces of 4 fp16 (hi, and the exact remainder rounded to fp16)
__device__ __forceinline__ void split4h(const float4 x, uint2& h, uint2& m) {
    h.x = pack2h(x.x, x.y);
    h.y = pack2h(x.z, x.w);
    m.x = pack2h(x.x - lo_h(h.x), x.y - hi_h(h.x));
    m.y = pack2h(x.z - lo_h(h.y), x.w - hi_h(h.y));
}

// split 4 fp32 into three pieces of 4 bf16 (2 dwords each)
__device__ __forceinline__ void split4(const float4 x, uint2& h, uint2& m, uint2& l) {
    h.x = pack2(x.x, x.y);
    h.y = pack2(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = pack2(r0, r1);
    m.y = pack2(r2, r3);
    l.x = pack2(r0 - lo_f(m.x), r1 - hi_f(m.x));
    l.y = pack2(r2 - lo_f(m.y), r3 - hi_f(m.y));
}

__global__ __launch_bounds__(kThreads) void split_bf16x3_kernel(const float* __restrict__ x, int64_t n4,
                                                               uint2* __restrict__ hi, uint2* __restrict__ mid,
                                                               uint2* __restrict__ lo) {
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < n4; v += (int64_t)gridDim.x * kThreads) {
        uint2 h, m, l;
        split4(reinterpret_cast<const float4*>(x)[v], h, m, l);
        hi[v] = h;
        mid[v] = m;
        lo[v] = l;
    }
}

// split 8 fp32 (two float4) into pieces of 8 bf16 (one 16-byte chunk each)
__device__ __forceinline__ void split8(const float4 x0, const float4 x1, uint4& h, uint4& m, uint4& l) {
    uint2 h0, m0, l0, h1, m1, l1;
    split4(x0, h0, m0, l0);
    split4(x1, h1, m1, l1);
    h = make_uint4(h0.x, h0.y, h1.x, h1.y);
    m = make_uint4(m0.x, m0.y, m1.x, m1.y);
    l = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// table[3*i .. 3*i+2] = (arena offset, N, K) of weight matrix i; unit_prefix[i] = number of 4-element units before it.
// One thread = one 16-byte OUTPUT chunk (8 consecutive k of one row): chunk c of a matrix is (k-block, row, half) with the half
// fastest, so a wave writes 1 KB of contiguous k-blocked output per piece and reads 32 rows x 64 B (the other half of each 128-B
// line is read by the wave handling the next k-block: an L2 hit).  `lo` may be NULL (bf16x3 keeps two pieces).
// F16: two fp16 pieces of kF16WScale * w; a weight whose scaled magnitude leaves fp16's range raises *overflow (checked by the host at its
// next synchronisation point: such a model needs the bf16 pieces).
template <bool F16>
__global__ __launch_bounds__(kThreads) void split_weights_kblock_kernel(const float* __restrict__ arena,
                                                                       const int64_t* __restrict__ table,
                                                                       const int64_t* __restrict__ unit_prefix, int n_mat,
                                                                       uint16_t* __restrict__ hi, uint16_t* __restrict__ mid,
                                                                       uint16_t* __restrict__ lo, int32_t* __restrict__ overflow) {
    const int64_t total = unit_prefix[n_mat] >> 1;  // 8-element chunks
    for (int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x; c < total; c += (int64_t)gridDim.x * kThreads) {
        int a = 0, b = n_mat - 1;  // largest i with unit_prefix[i] / 2 <= c
        while (a < b) {
            const int mdl = (a + b + 1) >> 1;
            if ((unit_prefix[mdl] >> 1) <= c) a = mdl; else b = mdl - 1;
        }
        const int64_t off = table[3 * a], N = table[3 * a + 1], K = table[3 * a + 2];
        const int64_t cm = c - (unit_prefix[a] >> 1);   // chunk inside the matrix: (kb * N + n) * 2 + half
        const int half = (int)(cm & 1);
        const int64_t rowk = cm >> 1, kb = rowk / N, n = rowk - kb * N;
        const float* src = arena + off + n * K + kb * 16 + half * 8;
        uint4 h, m, l;
        const int64_t dst = off + cm * 8;
        if (F16) {
            float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
            x0.x *= kF16WScale; x0.y *= kF16WScale; x0.z *= kF16WScale; x0.w *= kF16WScale;
            x1.x *= kF16WScale; x1.y *= kF16WScale; x1.z *= kF16WScale; x1.w *= kF16WScale;
            const float big = fmaxf(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))),
                                    fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
            const float nan_probe = (x0.x + x0.y) + (x0.z + x0.w) + (x1.x + x1.y) + (x1.z + x1.w);  // fmaxf drops NaNs; a sum keeps them
            if ((!(big <= 65504.0f) || nan_probe != nan_probe) && overflow) *overflow = 1;
            uint2 h0, m0, h1, m1;
            split4h(x0, h0, m0);
            split4h(x1, h1, m1);
            h = make_uint4(h0.x, h0.y, h1.x, h1.y);
            m = make_uint4(m0.x, m0.y, m1.x, m1.y);
        } else {
            split8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), h, m, l);
        }
        *reinterpret_cast<uint4*>(hi + dst) = h;
        *reinterpret_cast<uint4*>(mid + dst) = m;
        if (!F16 && lo) *reinterpret_cast<uint4*>(lo + dst) = l;
    }
}

// Token-major activation x (T, C) -> the hi / mid pieces of x^T (C, T_pad) in the k-blocked layout, in ONE pass: the operand of a
// weight gradient dW = dY^T x (k = tokens) without materialising the fp32 transpose.  One thread = one column c of one block of 16
// tokens: 16 reads, each coalesced across the threads of a wave, and 32 contiguous bytes written per piece.  Tokens >= T are zeros.
__global__ __launch_bounds__(kThreads) void split_tokens_kblock_kernel(const float* __restrict__ x, int64_t ldx, int T, int C,
                                                                      uint16_t* __restrict__ hi, uint16_t* __restrict__ mid) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    const int t0 = blockIdx.y * 16;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (t0 + j < T) ? x[(int64_t)(t0 + j) * ldx + c] : 0.f;
    uint32_t h[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = pack2(v[2 * j], v[2 * j + 1]);
        m[j] = pack2(v[2 * j] - lo_f(h[j]), v[2 * j + 1] - hi_f(h[j]));
    }
    const int64_t dst = ((int64_t)blockIdx.y * C + c) * 16;
    uint4* ph = reinterpret_cast<uint4*>(hi + dst);
    uint4* pm = reinterpret_cast<uint4*>(mid + dst);
    ph[0] = make_uint4(h[0], h[1], h[2], h[3]);
    ph[1] = make_uint4(h[4], h[5], h[6], h[7]);
    pm[0] = make_uint4(m[0], m[1], m[2], m[3]);
    pm[1] = make_uint4(m[4], m[5], m[6], m[7]);
}

// NP = number of bf16 pieces per operand: 3 -> six products (fp32-grade, ~2^-24), 2 -> three products hi*hi + hi*lo + lo*hi (~2^-16)
// NT = MFMA tiles along N per wave: 2 -> 128x128 block tile (3 workgroups/CU), 4 -> 128x256 block tile (wave tile 64x128:
// twice the MFMAs per barrier / LDS read / A byte; 128 accumulator VGPRs, 2 workgroups/CU)
// SK (split-K, training weight gradients: small outputs, token-deep K): blockIdx.y owns k in [y * kchunk, min(K, (y + 1) * kchunk))
// and writes its partial product to C + y * split_stride; the caller adds the partials (splitk_sum_kernel).
// F16 (NP == 2 only): fp16 pieces -- "f16x3"; the weight pieces hold kF16WScale * w and the epilogue undoes the scale.
template <int ACT, bool HAS_R, bool PF2, int NP, int NT, bool SK = false, bool F16 = false>
__global__ __launch_bounds__(kThreads, (NT == 4 ? 2 : 3)) void gemm_nt_bf16x6_kernel(
    const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ wh, const uint16_t* __restrict__ wm_,
    const uint16_t* __restrict__ wl, int64_t off0, int64_t off1, int64_t off2, const float* __restrict__ b0,
    const float* __restrict__ b1, const float* __restrict__ b2, int M, int seg_n, int K,
    const float* __restrict__ R, int64_t ldr, float* __restrict__ C, int64_t ldc, int tiles_n_seg, int tiles_n,
    int nwg, int group_n, int tiles_m, int kchunk = 0, int64_t split_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 * BUF bytes
    constexpr int BN = 64 * NT;                 // 128 or 256 columns per workgroup
    constexpr int BPIECE = (BN / 128) * PIECE;  // bytes of one B piece tile
    constexpr int BOFF = 3 * PIECE;             // B pieces start behind the (up to) three A pieces
    constexpr int BUF = BOFF + 3 * BPIECE;

    // Column-group-major tile order: all row tiles of the first group_n column tiles, then the next group.  The weight
    // panels of one group (group_n x BN x K x 4 B) stay resident in each XCD's 4 MiB L2 while the activations stream past
    // once per group; inside a group the column tile runs fastest, so the workgroups sharing an A panel are co-resident.
    const int pid = mr::xcd_remap(blockIdx.x, nwg);
    const int per_group = tiles_m * group_n;
    const int ng = pid / per_group;
    const int rem = pid - ng * per_group;
    const int gw = (tiles_n - ng * group_n) < group_n ? (tiles_n - ng * group_n) : group_n;  // width of this (maybe last) group
    const int tm = rem / gw, tn = ng * group_n + (rem - tm * gw);
    const int seg = tn / tiles_n_seg;
    const int n0 = (tn - seg * tiles_n_seg) * BN;
    const int m0 = tm * BM;
    int64_t woff = seg == 0 ? off0 : (seg == 1 ? off1 : off2);
    const float* __restrict__ bias = seg == 0 ? b0 : (seg == 1 ? b1 : b2);
    if (SK) {  // this workgroup's k range: shift the operand origins, shorten K
        const int kbeg = blockIdx.y * kchunk;
        A += kbeg;
        woff += (int64_t)kbeg * seg_n;  // k-blocked: advancing k by one block of 16 skips N * 16 elements
        C += (int64_t)blockIdx.y * split_stride;
        K = (K - kbeg) < kchunk ? (K - kbeg) : kchunk;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging maps
    // A (fp32): thread -> rows sr, sr + 64; k-quad kq (4 consecutive k)
    const int sr = tid >> 2, kq = tid & 3;
    int ar0 = m0 + sr, ar1 = m0 + sr + 64;
    ar0 = ar0 < M ? ar0 : M - 1;
    ar1 = ar1 < M ? ar1 : M - 1;
    const float* ga0 = A + (int64_t)ar0 * lda + kq * 4;
    const float* ga1 = A + (int64_t)ar1 * lda + kq * 4;
    const int swz_a = (sr >> 3) & 1;  // same for row sr + 64
    const int wa0 = sr * ROWB + (((kq >> 1) ^ swz_a) * 16) + (kq & 1) * 8;  // byte offsets inside an A piece
    const int wa1 = wa0 + 64 * ROWB;
    // B (pre-split bf16): thread -> row br, 16-byte half bh (8 consecutive k) of each piece
    const int brow = tid >> 1, bh = tid & 1;
    constexpr int BQ = BN / 128;  // 128-row passes over the B tile
    int64_t gboff[BQ];            // k-blocked: element (n, k) lives at off + ((k / 16) * N + n) * 16 + k % 16
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        int br = n0 + brow + 128 * q;
        br = br < seg_n ? br : seg_n - 1;
        gboff[q] = woff + (int64_t)br * 16 + bh * 8;
    }
    const int64_t kstep = (int64_t)seg_n;                     // elements per k-tile advance = N * 16 / 16 per unit k -> k0 * N
    const int wb = brow * ROWB + ((bh ^ ((brow >> 3) & 1)) * 16);
    // fragment read offsets (bytes)
    const int ra = (wm * 64 + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);  // rows +32 keep the same swizzle bit
    const int rb = (wn * 32 * NT + lr) * ROWB + ((lh ^ ((lr >> 3) & 1)) * 16);

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Stage {  // one k-tile of this thread's global data: 2 float4 of A, NP x BQ x 16 B of pre-split B
        float4 a0, a1;
        uint4 bh[BQ], bm[BQ], bl[BQ];
    };
    auto gload = [&](Stage& st, int k0) {
        st.a0 = *reinterpret_cast<const float4*>(ga0 + k0);
        st.a1 = *reinterpret_cast<const float4*>(ga1 + k0);
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            st.bh[q] = *reinterpret_cast<const uint4*>(wh + gboff[q] + k0 * kstep);
            st.bm[q] = *reinterpret_cast<const uint4*>(wm_ + gboff[q] + k0 * kstep);
            if (NP == 3) st.bl[q] = *reinterpret_cast<const uint4*>(wl + gboff[q] + k0 * kstep);
        }
    };
    auto lstore = [&](const Stage& st, unsigned char* buf) {
        uint2 h, m, l;
        if (F16) split4h(st.a0, h, m); else split4(st.a0, h, m, l);
        *reinterpret_cast<uint2*>(buf + 0 * PIECE + wa0) = h;
        *reinterpret_cast<uint2*>(buf + 1 * PIECE + wa0) = m;
        if (NP == 3) *reinterpret_cast<uint2*>(buf + 2 * PIECE + wa0) = l;
        if (F16) split4h(st.a1, h, m); else split4(st.a1, h, m, l);
        *reinterpret_cast<uint2*>(buf + 0 * PIECE + wa1) = h;
        *reinterpret_cast<uint2*>(buf + 1 * PIECE + wa1) = m;
        if (NP == 3) *reinterpret_cast<uint2*>(buf + 2 * PIECE + wa1) = l;
#pragma unroll
        for (int q = 0; q < BQ; ++q) {  // rows brow + 128 q: same swizzle bit
            *reinterpret_cast<uint4*>(buf + BOFF + 0 * BPIECE + q * PIECE + wb) = st.bh[q];
            *reinterpret_cast<uint4*>(buf + BOFF + 1 * BPIECE + q * PIECE + wb) = st.bm[q];
            if (NP == 3) *reinterpret_cast<uint4*>(buf + BOFF + 2 * BPIECE + q * PIECE + wb) = st.bl[q];
        }
    };
    auto compute = [&](const unsigned char* buf) {
        bf16x8 a[2][3], b[NT][3];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * PIECE + ra + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(buf + BOFF + p * BPIECE + rb + j * 32 * ROWB);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x16 c = acc[i][j];
                // smallest terms first, hi*hi last
                if (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);  // lo  * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);  // hi  * lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);  // mid * mid
                }
                if (F16) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);  // lo * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][1]), c, 0, 0, 0);  // hi * lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);  // hi * hi
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);  // mid * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);  // hi  * mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);  // hi  * hi
                }
                acc[i][j] = c;
            }
    };

    MR_PH_DECL
    const int nk = K / BK;
    auto ktile = [&](int kt) { return (kt < nk ? kt : 0) * BK; };  // past-the-end prefetches re-read tile 0 (never consumed)
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + BUF;
    if (PF2) {
        // Prefetch distance 2 (nk even): a k-tile's global loads are issued two steps before they are split and stored
        // to LDS, so the store waits on loads that have had two MFMA phases to land (counted vmcnt: the newer tile's
        // loads stay in flight).  Two named staging sets keep every register index static.
        Stage s0, s1;
        gload(s0, 0);
        lstore(s0, buf0);
        gload(s1, ktile(1));
        gload(s0, ktile(2));
        __syncthreads();
        MR_PH(0)
        // NP == 2, NT == 4 (the bf16x3 hot path): one scheduling region per k-tile -- this tile's 24 MFMAs interleaved with
        // the split + LDS store of the next tile and the global prefetch two tiles further on, so the wave's own VALU / LDS /
        // VMEM work runs in the gaps of its MFMA stream instead of after it (measured +6..9 %; a wave does not start the
        // staging phase until its queued MFMAs have drained, and the co-resident workgroup covers little of that).
#define MR_SGB(m, n) __builtin_amdgcn_sched_group_barrier(m, n, 0)
#define MR_SLOT_V MR_SGB(0x008, 1); MR_SGB(0x002, 2);
#define MR_SLOT_VD MR_SGB(0x008, 1); MR_SGB(0x002, 2); MR_SGB(0x200, 1);
#define MR_SLOT_VM MR_SGB(0x008, 1); MR_SGB(0x002, 1); MR_SGB(0x020, 1);
#define MR_PIPE24                                                                                                        \
    MR_SLOT_V MR_SLOT_V MR_SLOT_V MR_SLOT_V MR_SLOT_V MR_SLOT_V MR_SLOT_V MR_SLOT_V                                      \
    MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD MR_SLOT_VD                              \
    MR_SLOT_VM MR_SLOT_VM MR_SLOT_VM MR_SLOT_VM MR_SLOT_VM MR_SLOT_VM MR_SLOT_V MR_SLOT_V
        // (the fp16 split carries two more VALU instructions per float4 than the bf16 one; pipes with three VALU per slot measured the same:
        // tools/ab_gemm_flag.sh, r04)
        constexpr bool ILV = (NP == 2 && NT == 4);
        for (int kt = 0; kt < nk; kt += 2) {
            compute(buf0);
            MR_PH(1)
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
            lstore(s1, buf1);
            MR_PH(5)
            gload(s1, ktile(kt + 3));
            MR_PH(2)
            if (ILV) { MR_PIPE24 }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            MR_PH(3)
            compute(buf1);
            MR_PH(1)
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
            lstore(s0, buf0);
            MR_PH(5)
            gload(s0, ktile(kt + 4));
            MR_PH(2)
            if (ILV) { MR_PIPE24 }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            MR_PH(3)
        }
#undef MR_PIPE24
#undef MR_SLOT_VM
#undef MR_SLOT_VD
#undef MR_SLOT_V
#undef MR_SGB
    } else {
        Stage s0;
        gload(s0, 0);
        lstore(s0, buf0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            gload(s0, ktile(kt + 1));  // unconditional prefetch, see gemm.hip
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            compute((kt & 1) ? buf1 : buf0);
            __builtin_amdgcn_sched_barrier(0);
            lstore(s0, (kt & 1) ? buf0 : buf1);
            __syncthreads();
        }
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
    // One branch-free path for interior and edge tiles: C and R are addressed through tile-local buffer resources whose
    // num_records ends at the tile's last valid element, so rows past M are dropped by the hardware bounds check and
    // lanes whose column is past seg_n carry an out-of-range offset.  (Per-element guards put every store in its own
    // basic block behind `s_waitcnt vmcnt(0)`, i.e. each store waited for the previous one to complete.)
    const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
    const int cols_valid = (seg_n - n0) < BN ? (seg_n - n0) : BN;
    const int64_t col0 = (int64_t)seg * seg_n + n0;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        C + (int64_t)m0 * ldc + col0, 0, (int)(((int64_t)(rows_valid - 1) * ldc + cols_valid) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(HAS_R ? R + (int64_t)m0 * ldr + col0 : C), 0,
        HAS_R ? (int)(((int64_t)(rows_valid - 1) * ldr + cols_valid) * 4) : 0, 0x00020000);
    // opaque copies of the lane coordinates: keeps this address arithmetic from being hoisted above the main loop
    int lr_e = lr, lh_e = lh;
    asm volatile("" : "+v"(lr_e), "+v"(lh_e));
    const uint32_t ldc4 = (uint32_t)ldc * 4u, ldr4 = (uint32_t)ldr * 4u;
    float bz[NT];
    uint32_t coff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int colt = wn * 32 * NT + j * 32 + lr_e;
        const bool ok = colt < cols_valid;
        bz[j] = bias ? bias[n0 + (ok ? colt : cols_valid - 1)] : 0.f;
        coff[j] = ok ? (uint32_t)colt * 4u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t rowt = wm * 64 + i * 32 + 4 * lh_e;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // four rows at a time (one accumulator quad)
                float v[4], rr[4];
                if (HAS_R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (rowt + r + 8 * q) * ldr4 + coff[j], 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = (F16 ? acc[i][j][4 * q + r] * kF16WScaleInv : acc[i][j][4 * q + r]) + bz[j];  // (the power-of-two scale is exact)
                    if (ACT == MR_ACT_GELU_ERF) v[r] = gelu_erf(v[r]);
                    if (HAS_R) v[r] += rr[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[r]), crs, (rowt + r + 8 * q) * ldc4 + coff[j], 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    MR_PH(4)
    MR_PH_FLUSH(pid)
}

// C[m][n] = sum over splits (ascending) of part[s][m][n] (+ bias[n]) (+ R[m][n]); one float4 per thread, N % 4 == 0
__global__ __launch_bounds__(kThreads) void splitk_sum_kernel(const float* __restrict__ part, int splits, int64_t split_stride, int M, int N,
                                                             const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
                                                             float* __restrict__ C, int64_t ldc) {
    const int nq = N >> 2;
    const int64_t total = (int64_t)M * nq;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int m = (int)(e / nq), n = (int)(e - (int64_t)m * nq) * 4;
        const float* p = part + (int64_t)m * N + n;
        float4 a = *reinterpret_cast<const float4*>(p);
        for (int s = 1; s < splits; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(p + (int64_t)s * split_stride);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) { a.x += bias[n]; a.y += bias[n + 1]; a.z += bias[n + 2]; a.w += bias[n + 3]; }
        float* c = C + (int64_t)m * ldc + n;
        if (R) {
            const float* r = R + (int64_t)m * ldr + n;
            a.x += r[0]; a.y += r[1]; a.z += r[2]; a.w += r[3];
        }
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
    }
}

}  // namespace

extern "C" int mr_split_bf16x3_f32(const float* x, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                                   mr_stream_t stream) {
    if (!x || !hi || !mid || !lo || n < 0) return MR_EINVAL;
    if ((n & 3) || !mr::aligned16(x) || (reinterpret_cast<uintptr_t>(hi) & 7) || (reinterpret_cast<uintptr_t>(mid) & 7) ||
        (reinterpret_cast<uintptr_t>(lo) & 7))
        return MR_EALIGN;
    if (n == 0) return MR_OK;
    int64_t blocks = (n / 4 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, x, n / 4,
                       reinterpret_cast<uint2*>(hi), reinterpret_cast<uint2*>(mid), reinterpret_cast<uint2*>(lo));
    return mr::check_launch();
}

extern "C" int mr_split_weights_kblock_f32(const float* arena, const int64_t* table, const int64_t* unit_prefix, int n_mat,
                                          int64_t total_units, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                                          mr_stream_t stream) {
    if (!arena || !table || !unit_prefix || !hi || !mid || n_mat < 0 || total_units < 0) return MR_EINVAL;
    if (!mr::aligned16(arena) || (reinterpret_cast<uintptr_t>(hi) & 15) || (reinterpret_cast<uintptr_t>(mid) & 15) ||
        (lo && (reinterpret_cast<uintptr_t>(lo) & 15)))
        return MR_EALIGN;
    if (n_mat == 0 || total_units == 0) return MR_OK;
    int64_t blocks = (total_units / 2 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split_weights_kblock_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, arena, table,
                       unit_prefix, n_mat, hi, mid, lo, nullptr);
    return mr::check_launch();
}

// fp16 pieces ("f16x3"): hi / lo of 2^8 * w in the same k-blocked layout.  *overflow (device int32, optional, never cleared here) is set when
// a scaled weight leaves fp16's range (|w| >= 255.9) or is NaN: that model cannot use the fp16 pieces.
extern "C" int mr_split_weights_kblock_f16_f32(const float* arena, const int64_t* table, const int64_t* unit_prefix, int n_mat,
                                              int64_t total_units, uint16_t* hi, uint16_t* lo, int32_t* overflow, mr_stream_t stream) {
    if (!arena || !table || !unit_prefix || !hi || !lo || n_mat < 0 || total_units < 0) return MR_EINVAL;
    if (!mr::aligned16(arena) || (reinterpret_cast<uintptr_t>(hi) & 15) || (reinterpret_cast<uintptr_t>(lo) & 15)) return MR_EALIGN;
    if (n_mat == 0 || total_units == 0) return MR_OK;
    int64_t blocks = (total_units / 2 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split_weights_kblock_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, arena, table,
                       unit_prefix, n_mat, hi, lo, nullptr, overflow);
    return mr::check_launch();
}

extern "C" int mr_split_tokens_kblock_f32(const float* x, int64_t ldx, int T, int C, int T_pad, uint16_t* hi, uint16_t* mid,
                                          mr_stream_t stream) {
    if (!x || !hi || !mid || T < 0 || C < 1 || ldx < C || T_pad < T) return MR_EINVAL;
    if (T_pad % 16) return MR_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(hi) & 15) || (reinterpret_cast<uintptr_t>(mid) & 15)) return MR_EALIGN;
    if (T_pad == 0) return MR_OK;
    if (T_pad / 16 > 65535) return MR_EUNSUPPORTED;
    hipLaunchKernelGGL(split_tokens_kblock_kernel, dim3((C + kThreads - 1) / kThreads, T_pad / 16), dim3(kThreads), 0, (hipStream_t)stream, x,
                       ldx, T, C, hi, mid);
    return mr::check_launch();
}

extern "C" int mr_gemm_nt_bf16x6_f32(const float* A, int64_t lda, const uint16_t* w_hi, const uint16_t* w_mid,
                                     const uint16_t* w_lo, int64_t off0, int64_t off1, int64_t off2, const float* b0,
                                     const float* b1, const float* b2, int nseg, int M, int seg_n, int K, int act,
                                     const float* R, int64_t ldr, float* C, int64_t ldc, int products, mr_stream_t stream) {
    if (products != 6 && products != 3 && products != MR_PRODUCTS_F16X3) return MR_EUNSUPPORTED;
    const bool f16 = products == MR_PRODUCTS_F16X3;  // fp16 pieces (w_hi / w_mid from mr_split_weights_kblock_f16_f32), three products
    if (f16) products = 3;
    if (!A || !w_hi || !w_mid || (products == 6 && !w_lo) || !C || nseg < 1 || nseg > 3 || M < 0 || seg_n < 1 || K < 1) return MR_EINVAL;
    if (K % BK) return MR_EUNSUPPORTED;
    if (nseg > 1 && (seg_n % 128)) return MR_EUNSUPPORTED;
    if (act != MR_ACT_NONE && act != MR_ACT_GELU_ERF) return MR_EUNSUPPORTED;
    if ((lda & 3) || !mr::aligned16(A) || !mr::aligned16(w_hi) || !mr::aligned16(w_mid) || (w_lo && !mr::aligned16(w_lo)) ||
        (off0 & 7) || (nseg > 1 && (off1 & 7)) || (nseg > 2 && (off2 & 7)))
        return MR_EALIGN;
    if (ldc < 1 || ldc > (1 << 21) || (R && (ldr < 1 || ldr > (1 << 21)))) return MR_EUNSUPPORTED;  // 32-bit tile-local offsets in the epilogue
    if (M == 0) return MR_OK;
    // wide (128 x 256) tiles when a segment is a multiple of 256 columns and the grid still fills the chip
    static const int force_nt = [] { const char* e = getenv("MR_GEMM_NT"); return e ? atoi(e) : 0; }();
    const int tiles_m = (M + BM - 1) / BM;
    bool wide = (seg_n % 256 == 0) && ((int64_t)tiles_m * (seg_n / 256) * nseg >= 512);
    if (force_nt == 2) wide = false;
    if (force_nt == 4 && (seg_n % 256 == 0 || nseg == 1)) wide = true;
    const int BN = wide ? 256 : 128;
    const int tiles_n_seg = (seg_n + BN - 1) / BN;
    const int tiles_n = tiles_n_seg * nseg;
    const int64_t nwg64 = (int64_t)tiles_m * tiles_n;
    if (nwg64 > 0x7fffffff) return MR_EUNSUPPORTED;
    const int nwg = (int)nwg64;
    static const int force_gn = [] { const char* e = getenv("MR_GEMM_GROUPN"); return e ? atoi(e) : 0; }();
    // column tiles per group of the tile order (kernel comment): 768 columns.  Against the whole-row-panel order (group = all column tiles)
    // this fetches 19 % fewer bytes into the L2s over the bench's launches (FETCH_SIZE 475 vs 593 MB raw per launch, tools/ab_fetch.sh) at
    // the same kernel time; groups of 1 re-read the activations per column tile (897 MB, +5 % time).
    int group_n = 768 / BN < tiles_n ? 768 / BN : tiles_n;
    if (force_gn > 0) group_n = force_gn < tiles_n ? force_gn : tiles_n;
    if (force_gn < 0) group_n = tiles_n;  // A/B: the r01-r03 order
    hipStream_t st = (hipStream_t)stream;
    static const int shm_pad = [] { const char* e = getenv("MR_GEMM_SHM_PAD"); return e ? atoi(e) : 0; }();  // diagnostic: lowers occupancy
    const size_t shm = 2 * (size_t)(3 * PIECE + 3 * (BN / 128) * PIECE) + shm_pad;  // 48 KB (narrow) / 72 KB (wide)
    // (three pieces + 128 accumulators + two staging sets do not fit 256 VGPRs: the wide x6 kernel prefetches one tile ahead)
    const bool pf2 = ((K / BK) % 2 == 0) && !(wide && products == 6);
#define MR_GEMM_LAUNCH6(ACT_, HASR_, PF2_, NP_, NT_, F16_)                                                                                \
    do {                                                                                                                                  \
        static mr::DynLdsCeiling lds_ceiling;                                                                                             \
        if (const int e_ = lds_ceiling.ensure(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<ACT_, HASR_, PF2_, NP_, NT_, false, F16_>), \
                                              2 * (3 * PIECE + 3 * (NT_ / 2) * PIECE) + shm_pad))                                         \
            return e_;                                                                                                                    \
        hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<ACT_, HASR_, PF2_, NP_, NT_, false, F16_>), dim3(nwg), dim3(kThreads), shm, st, A, lda,  \
                           w_hi, w_mid, w_lo, off0, off1, off2, b0, b1, b2, M, seg_n, K, R, ldr, C, ldc, tiles_n_seg, tiles_n, nwg,        \
                           group_n, tiles_m);                                                                                             \
    } while (0)
#define MR_GEMM_LAUNCH5(ACT_, HASR_, PF2_, NP_, NT_)                           \
    do {                                                                       \
        if (NP_ == 2 && f16) MR_GEMM_LAUNCH6(ACT_, HASR_, PF2_, 2, NT_, true); \
        else MR_GEMM_LAUNCH6(ACT_, HASR_, PF2_, NP_, NT_, false);              \
    } while (0)
#define MR_GEMM_LAUNCH4(ACT_, HASR_, PF2_, NP_)                           \
    do {                                                                  \
        if (wide) MR_GEMM_LAUNCH5(ACT_, HASR_, PF2_, NP_, 4);             \
        else MR_GEMM_LAUNCH5(ACT_, HASR_, PF2_, NP_, 2);                  \
    } while (0)
#define MR_GEMM_LAUNCH3(ACT_, HASR_, PF2_)                          \
    do {                                                            \
        if (products == 6) MR_GEMM_LAUNCH4(ACT_, HASR_, PF2_, 3);   \
        else MR_GEMM_LAUNCH4(ACT_, HASR_, PF2_, 2);                 \
    } while (0)
#define MR_GEMM_LAUNCH(ACT_, HASR_)                         \
    do {                                                    \
        if (pf2) MR_GEMM_LAUNCH3(ACT_, HASR_, true);        \
        else MR_GEMM_LAUNCH3(ACT_, HASR_, false);           \
    } while (0)
    if (act == MR_ACT_GELU_ERF) {
        if (R) MR_GEMM_LAUNCH(MR_ACT_GELU_ERF, true); else MR_GEMM_LAUNCH(MR_ACT_GELU_ERF, false);
    } else {
        if (R) MR_GEMM_LAUNCH(MR_ACT_NONE, true); else MR_GEMM_LAUNCH(MR_ACT_NONE, false);
    }
#undef MR_GEMM_LAUNCH
#undef MR_GEMM_LAUNCH3
#undef MR_GEMM_LAUNCH4
#undef MR_GEMM_LAUNCH5
#undef MR_GEMM_LAUNCH6
    return mr::check_launch();
}

// ---- split-K variant of the bf16x3 GEMM (one weight segment, no activation): C = A W^T (+ bias) (+ R) with K cut into `splits`
// chunks whose partial products meet in `ws` and are added in chunk order.  For the fine-tuning weight gradients dW = dY^T X:
// outputs of a few dozen tiles, K = the number of tokens.
extern "C" size_t mr_gemm_nt_bf16x3_splitk_ws_bytes(int M, int N, int splits) {
    if (M < 0 || N < 1 || splits < 1) return 0;
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

extern "C" int mr_gemm_nt_bf16x3_splitk_f32(const float* A, int64_t lda, const uint16_t* w_hi, const uint16_t* w_mid, int64_t off,
                                            const float* bias, int M, int N, int K, const float* R, int64_t ldr, float* C, int64_t ldc,
                                            int splits, void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (!A || !w_hi || !w_mid || !C || M < 0 || N < 1 || K < 1 || splits < 1) return MR_EINVAL;
    if (splits == 1)
        return mr_gemm_nt_bf16x6_f32(A, lda, w_hi, w_mid, w_hi, off, 0, 0, bias, nullptr, nullptr, 1, M, N, K, MR_ACT_NONE, R, ldr, C, ldc, 3, stream);
    if ((K % BK) || (N & 3)) return MR_EUNSUPPORTED;
    if ((lda & 3) || !mr::aligned16(A) || !mr::aligned16(w_hi) || !mr::aligned16(w_mid) || (off & 7)) return MR_EALIGN;
    if (!ws || ws_bytes < mr_gemm_nt_bf16x3_splitk_ws_bytes(M, N, splits) || !mr::aligned16(ws)) return MR_EWS;
    if (M == 0) return MR_OK;
    // chunk = a multiple of two k-tiles (the kernel's prefetch-distance-2 pipeline) covering K in `splits` pieces
    int kchunk = ((K + splits - 1) / splits + 2 * BK - 1) / (2 * BK) * (2 * BK);
    const int nsplit = (K + kchunk - 1) / kchunk;
    const bool pf2 = ((K - (nsplit - 1) * kchunk) / BK) % 2 == 0;  // the last chunk may hold an odd number of k-tiles
    const int tiles_m = (M + BM - 1) / BM;
    const bool wide = (N % 256 == 0) && ((int64_t)tiles_m * (N / 256) * nsplit >= 512);
    const int BN = wide ? 256 : 128;
    const int tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    float* part = reinterpret_cast<float*>(ws);
    const int64_t stride = (int64_t)M * N;
    if (N > (1 << 21)) return MR_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define MR_SK_LAUNCH(PF2_, NT_)                                                                                                          \
    do {                                                                                                                                 \
        static mr::DynLdsCeiling lds_ceiling;                                                                                            \
        if (const int e_ = lds_ceiling.ensure(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<MR_ACT_NONE, false, PF2_, 2, NT_, true>), \
                                              2 * (3 * PIECE + 3 * (NT_ / 2) * PIECE)))                                                  \
            return e_;                                                                                                                   \
        hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<MR_ACT_NONE, false, PF2_, 2, NT_, true>), dim3(nwg, nsplit), dim3(kThreads),            \
                           2 * (size_t)(3 * PIECE + 3 * (BN / 128) * PIECE), st, A, lda, w_hi, w_mid, w_hi, off, 0, 0, nullptr, nullptr,  \
                           nullptr, M, N, K, nullptr, 0, part, (int64_t)N, tiles_n, tiles_n, nwg, tiles_n, tiles_m, kchunk, stride);      \
    } while (0)
    if (pf2) { if (wide) MR_SK_LAUNCH(true, 4); else MR_SK_LAUNCH(true, 2); }
    else { if (wide) MR_SK_LAUNCH(false, 4); else MR_SK_LAUNCH(false, 2); }
#undef MR_SK_LAUNCH
    int64_t blocks = ((int64_t)M * (N / 4) + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, part, nsplit, stride, M, N, bias, R, ldr, C, ldc);
    return mr::check_launch();
}
