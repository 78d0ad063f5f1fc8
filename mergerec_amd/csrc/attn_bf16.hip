// K4': the attention kernel of attn.hip with both products (S^T = K Q^T and O^T = V^T P^T) evaluated in split precision
// on the bf16 matrix cores (see gemm_bf16.hip): every fp32 operand is NP bf16 pieces (NP = 2: three products, NP = 3: six),
// products accumulate in fp32.  Per 32-query x 32-key tile that is 24 (48) v_mfma_f32_32x32x16_bf16 = 768 (1536) matrix-pipe
// cycles instead of 64 x 64 = 4096 for the fp32 MFMA kernel.
//
// Same decomposition as attn.hip: workgroup = 4 waves = 128 queries of one (sequence, head); K/V tiles of 32 keys go through
// double-buffered LDS with the next tile's loads in flight under the current tile's math; scores are computed transposed so P
// stays in registers as the B operand of the second product.  Differences:
//   * K is split into its pieces while it is staged; LDS image per piece = [32 keys][64 bf16] (128-B rows), the eight 16-B
//     chunks of a row XOR-swizzled by (row >> 1) & 7, so the ds_read_b128 of 16 keys is conflict-free.
//   * Q is split once per wave into registers; P is split in registers after the softmax (register r = 8 s + j of the score
//     accumulator is exactly element j of k-step s of the B operand).
//   * V is split while it is staged, too: LDS image per piece = [32 keys][64 bf16] row-major (coalesced 8-byte stores), and
//     the A operand of O^T = V^T P^T (k = key, so column-wise through that image) comes from gfx950's transposed LDS read
//     ds_read_b64_tr_b16: a 16-lane group fetches a 4-key x 16-d block and lane i receives column i.  The 16-B chunks of a
//     row are XOR-swizzled by 4 * ((row >> 1) & 1), which makes the 4 rows x 64 B of a 32-lane half hit all 64 banks once.
//     (Splitting V in registers per wave, as before, cost more VALU time than both MFMA products together.)
#include "common.h"
#include "dropout.h"
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

// phase-timing hooks: empty in the library; exp/attn_phases.hip defines them (s_memtime deltas per phase)
#ifndef MR_PH_DECL
#define MR_PH_DECL
#define MR_PH(i)
#define MR_PH_FLUSH(pid)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kDh = 64;
constexpr int KROWB = 128;           // bytes per K piece row (64 bf16)
constexpr int KPIECE = 32 * KROWB;   // 4 KB per piece per tile

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// FP16 pieces ("f16x3", see gemm_bf16.hip): q, k, v and the probabilities are activations -- unscaled, |x| < 65504; a probability's low piece
// may be an fp16 subnormal, which the matrix pipe honours (absolute error 2^-25 on a value in [0, 1])
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_h(uint32_t u) {
    float r;
    asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(u));
    return r;
}
__device__ __forceinline__ float hi_h(uint32_t u) {
    float r;
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(u));
    return r;
}
// the fp16 low piece of (a, b) given their packed high piece w: f16(a - float(w.lo)), f16(b - float(w.hi)).  One v_fma_mix per value
// (fp16 operand read in place, difference exact in fp32, one rounding to fp16) instead of convert + subtract + a shared pack: the same
// bits as pack2h(a - lo_h(w), b - hi_h(w))
__device__ __forceinline__ uint32_t resid2h(float a, float b, uint32_t w) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r)
        : "v"(w), "v"(a), "v"(b));
    return r;
}
// x of lane i and of lane i ^ 32, the same pair in both halves (gfx950 v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ void half_pair(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
template <bool F16> __device__ __forceinline__ uint32_t pk(float a, float b) { return F16 ? pack2h(a, b) : pack2(a, b); }
template <bool F16> __device__ __forceinline__ float lo_v(uint32_t u) { return F16 ? lo_h(u) : lo_f(u); }
template <bool F16> __device__ __forceinline__ float hi_v(uint32_t u) { return F16 ? hi_h(u) : hi_f(u); }

// 8 fp32 -> NP pieces of 8 bf16 / fp16 (4 dwords each); piece 0 = hi
template <int NP, bool F16 = false>
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&out)[NP]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = pk<F16>(r[2 * j], r[2 * j + 1]);
            out[p][j] = w;
            if (p + 1 < NP) {
                r[2 * j] -= lo_v<F16>(w);
                r[2 * j + 1] -= hi_v<F16>(w);
            }
        }
    }
}

// split8 of the work-list kernels: fp16 low pieces through resid2h (attn_split_kernel keeps split8, so the bit-identity test of the two
// kernels also compares the two formulations)
template <int NP, bool F16>
__device__ __forceinline__ void split8w(const float (&x)[8], u32x4 (&out)[NP]) {
    if (F16) {  // NP == 2
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = pack2h(x[2 * j], x[2 * j + 1]);
            out[0][j] = w;
            out[1][j] = resid2h(x[2 * j], x[2 * j + 1], w);
        }
        return;
    }
    split8<NP, false>(x, out);
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// acc += sum over the product set of piece pairs: NP = 2 -> (lo,hi) (hi,lo) (hi,hi); NP = 3 -> + (lo2,hi) (hi,lo2) (mid,mid)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NP, bool F16 = false>
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[NP], const u32x4 (&b)[NP], f32x16 c) {
    if (F16) {  // NP == 2
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
        return c;
    }
    if (NP == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[2]), as_bf16x8(b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[0]), as_bf16x8(b[2]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[1]), as_bf16x8(b[1]), c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[1]), as_bf16x8(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[0]), as_bf16x8(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[0]), as_bf16x8(b[0]), c, 0, 0, 0);
    return c;
}

template <bool WINDOWED, int NP, bool F16 = false>
__global__ __launch_bounds__(kThreads, (NP == 3 ? 2 : 3)) void attn_split_kernel(const float* __restrict__ qkv,
                                                                const int32_t* __restrict__ cu,
                                                                const int32_t* __restrict__ seq_order, int H,
                                                                float scale_log2e, int window, float* __restrict__ ctx) {
    constexpr int BUFB = 2 * NP * KPIECE;  // K pieces, then V pieces
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 * BUFB
    const int b = seq_order ? seq_order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    const int q_base = blockIdx.x * 128;
    if (q_base >= len) return;  // the whole workgroup leaves together, before any barrier
    const int64_t ld = (int64_t)3 * H * kDh;
    const float* __restrict__ Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Kb = Qb + H * kDh;
    const float* __restrict__ Vb = Qb + 2 * H * kDh;

    const int q0 = q_base + wave * 32;
    const bool wave_active = q0 < len;  // wave-uniform
    const int qi = q0 + lr;

    // Q pieces as the B operand of S^T = K Q^T: k-step s covers d = 16 s + 8 lh + j; pre-scaled by scale * log2(e)
    u32x4 qp[4][NP];
    {
        const int qrow = qi < len ? qi : len - 1;
        const float* qr = Qb + (int64_t)qrow * ld + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 x0 = *reinterpret_cast<const float4*>(qr + 16 * s);
            const float4 x1 = *reinterpret_cast<const float4*>(qr + 16 * s + 4);
            const float x[8] = {x0.x * scale_log2e, x0.y * scale_log2e, x0.z * scale_log2e, x0.w * scale_log2e,
                                x1.x * scale_log2e, x1.y * scale_log2e, x1.z * scale_log2e, x1.w * scale_log2e};
            split8<NP, F16>(x, qp[s]);
        }
    }

    // key-tile schedule of the workgroup (as attn.hip)
    int k_lo = 0, k_hi = len;
    if (WINDOWED) {
        k_lo = q_base - window;
        k_lo = k_lo < 0 ? 0 : (k_lo & ~31);
        k_hi = q_base + 127 + window + 1;
        k_hi = k_hi > len ? len : k_hi;
    }
    const bool extra0 = WINDOWED && k_lo > 0;
    const int ntiles = (k_hi - k_lo + 31) / 32 + (extra0 ? 1 : 0);
    auto tile_base = [&](int it) { return extra0 ? (it == 0 ? 0 : k_lo + (it - 1) * 32) : k_lo + it * 32; };

    // staging map: thread -> (row sr / sr + 16, 4 consecutive d at sc)
    const int sr = tid >> 4, sc4 = tid & 15, sc = sc4 * 4;
    float4 kreg0, kreg1, vreg0, vreg1;
    auto gload = [&](int kb) {
        int r0 = kb + sr, r1 = kb + sr + 16;
        r0 = r0 < len ? r0 : len - 1;
        r1 = r1 < len ? r1 : len - 1;
        kreg0 = *reinterpret_cast<const float4*>(Kb + (int64_t)r0 * ld + sc);
        kreg1 = *reinterpret_cast<const float4*>(Kb + (int64_t)r1 * ld + sc);
        vreg0 = *reinterpret_cast<const float4*>(Vb + (int64_t)r0 * ld + sc);
        vreg1 = *reinterpret_cast<const float4*>(Vb + (int64_t)r1 * ld + sc);
    };
    // K piece position of (row, 4-d group sc4): 16-B chunk (sc4 >> 1) ^ ((row >> 1) & 7), 8-B half sc4 & 1
    const int kw0 = sr * KROWB + ((((sc4 >> 1) ^ ((sr >> 1) & 7)) << 4) | ((sc4 & 1) << 3));
    const int kw1 = (sr + 16) * KROWB + ((((sc4 >> 1) ^ (((sr + 16) >> 1) & 7)) << 4) | ((sc4 & 1) << 3));
    auto store_k = [&](const float4 x, unsigned char* dst) {
        float r0 = x.x, r1 = x.y, r2 = x.z, r3 = x.w;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t w0 = pk<F16>(r0, r1), w1 = pk<F16>(r2, r3);
            *reinterpret_cast<uint2*>(dst + p * KPIECE) = make_uint2(w0, w1);
            if (p + 1 < NP) { r0 -= lo_v<F16>(w0); r1 -= hi_v<F16>(w0); r2 -= lo_v<F16>(w1); r3 -= hi_v<F16>(w1); }
        }
    };
    // V piece position of (row, 4-d group sc4): 16-B chunk (sc4 >> 1) ^ (4 * ((row >> 1) & 1)); rows sr and sr + 16 share the bit
    const int vw0 = sr * KROWB + ((((sc4 >> 1) ^ (((sr >> 1) & 1) << 2)) << 4) | ((sc4 & 1) << 3));
    const int vw1 = vw0 + 16 * KROWB;
    auto lstore = [&](unsigned char* buf) {
        store_k(kreg0, buf + kw0);
        store_k(kreg1, buf + kw1);
        store_k(vreg0, buf + NP * KPIECE + vw0);
        store_k(vreg1, buf + NP * KPIECE + vw1);
    };
    // transposed V read: lane 4 q + p4 of a 16-lane group addresses row (4 lh + q) + 16 st + 8 jj, d columns 4 p4 .. + 3 of the
    // 16-d block g1 of d tile dt; it receives column (lane & 15) of the block's four keys
    int vtr[2];
    {
        const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, g1 = (lane >> 4) & 1, bsw = (q >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            vtr[dt] = (4 * lh + q) * KROWB + ((4 * (dt ^ bsw) + 2 * g1 + (p4 >> 1)) << 4) + ((p4 & 1) << 3);
    }
    // K fragment read: lane (key lr, half lh), k-step s -> logical 16-B chunk 2 s + lh of row lr
    const int kswz = (lr >> 1) & 7;

    float m = -INFINITY, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    MR_PH_DECL
    gload(tile_base(0));
    lstore(lds);
    __syncthreads();
    MR_PH(0)

    for (int it = 0; it < ntiles; ++it) {
        const unsigned char* buf = lds + (it & 1) * BUFB;
        const int kb = tile_base(it);
        gload(tile_base(it + 1 < ntiles ? it + 1 : it));  // unconditional: keeps the loads in flight under the math
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        bool relevant = wave_active;
        if (WINDOWED) relevant = relevant && (kb == 0 || (kb + 31 >= q0 - window && kb <= q0 + 31 + window));
        if (relevant) {
            // ---- S^T tile = K Q^T
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                u32x4 ka[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ka[p] = *reinterpret_cast<const u32x4*>(buf + p * KPIECE + lr * KROWB + (((2 * st + lh) ^ kswz) << 4));
                s = mfma_split<NP, F16>(ka, qp[st], s);
            }
            MR_PH(1)
            // ---- mask + online softmax (base 2); s[r] is key kb + (r&3) + 8*(r>>2) + 4*lh for query q0 + lr
            float mx = -INFINITY;
            if (!WINDOWED && kb + 32 <= len) {  // interior tile of full attention: every key valid, no masking work
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    bool ok = key < len;
                    if (WINDOWED) {
                        const int dlt = qi - key;
                        ok = ok && (key == 0 || (dlt <= window && dlt >= -window));
                    }
                    s[r] = ok ? s[r] : -INFINITY;
                    mx = fmaxf(mx, s[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float corr = (m == -INFINITY) ? ((m_new == -INFINITY) ? 1.f : 0.f) : __builtin_amdgcn_exp2f(m - m_use);
            float ps = 0.f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = __builtin_amdgcn_exp2f(s[r] - (F16 ? m_use - 10.f : m_use));  // raw v_exp_f32; F16: p' = 2^10 p, see the work-list kernel
                ps += pv[r];
            }
            ps += __shfl_xor(ps, 32, 64);
            l = l * corr + ps;
            m = m_new;
            if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0) {  // wave-uniform: the running maxima usually stop moving early
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
            }
            // P pieces: k-step st of the second product takes registers 8 st .. 8 st + 7
            u32x4 pp[2][NP];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float x[8] = {pv[8 * st], pv[8 * st + 1], pv[8 * st + 2], pv[8 * st + 3],
                                    pv[8 * st + 4], pv[8 * st + 5], pv[8 * st + 6], pv[8 * st + 7]};
                split8<NP, F16>(x, pp[st]);
            }
            MR_PH(2)
            // ---- O^T += V^T P^T: A operand element j of k-step st is V[kb + (j & 3) + 8 (2 st + (j >> 2)) + 4 lh][d]
            const unsigned char* vbase = buf + NP * KPIECE;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    u32x4 va[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const unsigned char* a0 = vbase + p * KPIECE + vtr[dt] + (16 * st) * KROWB;
                        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                       (__attribute__((address_space(3))) s16x4*)a0));
                        const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                       (__attribute__((address_space(3))) s16x4*)(a0 + 8 * KROWB)));
                        va[p][0] = lo.x; va[p][1] = lo.y; va[p][2] = hi.x; va[p][3] = hi.y;
                    }
                    if (dt == 0) o0 = mfma_split<NP, F16>(va, pp[st], o0);
                    else o1 = mfma_split<NP, F16>(va, pp[st], o1);
                }
            }
        }
        MR_PH(3)
        __builtin_amdgcn_sched_barrier(0);
        lstore(lds + ((it + 1) & 1) * BUFB);
        MR_PH(4)
        __syncthreads();
        MR_PH(5)
    }
    MR_PH_FLUSH((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)

    if (wave_active && qi < len && !(WINDOWED && qi == 0)) {
        const float inv = 1.0f / l;
        float* op = ctx + (int64_t)(t0 + qi) * ((int64_t)H * kDh) + h * kDh + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(op + 8 * g) =
                make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + 8 * g) =
                make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Work-list form (r03).  One workgroup = QT * 128 query rows of one (sequence, head): every wave owns up to QT 32-row query tiles
// (tile w and, for QT = 2, tile w + 4 of the block), so a staged 32-key K / V tile and every K / V fragment a wave reads from LDS
// serve QT times the matrix work of attn_split_kernel; K and V of a sequence are split and written to LDS once per 256 queries
// instead of once per 128.  The grid is a host-built list of the (sequence, query block) pairs that exist -- no empty workgroups --
// dealt over the 8 XCD dispatch queues (workgroup id % 8) so that the blocks which read the same K / V sit in one XCD's L2,
// heaviest sequences first.  K / V rows come through buffer loads whose range ends at the sequence's last row (rows past it read
// as zeros and are masked as before): two vector adds per key tile instead of the clamped 64-bit row arithmetic.
// Per (query row, key tile) the operations and their order are those of attn_split_kernel: results are bit-identical.
// r04: interior key tiles of two-tile full attention (NP = 2) run in software-pipelined order -- the softmax and the P split of one
// query tile issue under the other tile's MFMAs, the next K / V tile's split + LDS stores under the last product -- with the fp16 low
// pieces from one v_fma_mix per value and the cross-half maximum / sum through v_permlane32_swap: 340 -> 224 vector instructions per
// 48 MFMAs, still bit-identical to attn_split_kernel (which keeps the plain formulation as the test twin).
template <int J>
struct QIdx { static constexpr int value = J; };

// DROP (training graph only): probabilities masked by mr::dropout_keep(key, query token * H + head, key position), scaled by 1 / (1 - p),
// after the row sum has taken the un-dropped values (softmax, then dropout).
template <bool WINDOWED, int NP, int QT, bool DROP = false, bool F16 = false>
__global__ __launch_bounds__(kThreads, (QT == 2 ? 2 : (NP == 3 ? 2 : 3))) void attn_split_work_kernel(
    const float* __restrict__ qkv, const int32_t* __restrict__ cu, const int32_t* __restrict__ work, int H, float scale_log2e,
    int window, float* __restrict__ ctx, int nseq, uint32_t drop_thresh = 0u, float drop_inv = 1.f, uint32_t drop_key = 0u) {
    constexpr int BUFB = 2 * NP * KPIECE;  // K pieces, then V pieces
    constexpr int QROWS = 128 * QT;
    constexpr bool PIPED = QT == 2 && !WINDOWED && !DROP && NP == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 * BUFB
    const int id = blockIdx.x, slot = id >> 3;
    const int e = slot / H, h = slot - e * H;
    const int ent = __builtin_amdgcn_readfirstlane(work[e * 8 + (id & 7)]);
    if (ent < 0) return;  // padding entry of a shorter XCD queue: the whole workgroup leaves together, before any barrier
    const int b = ent & 0xffffff, q_base = (ent >> 24) * QROWS;
    if (b >= nseq) return;  // an entry of a list planned for another batch: never index cu[] past its B + 1 entries
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int t0 = __builtin_amdgcn_readfirstlane(cu[b]), len = __builtin_amdgcn_readfirstlane(cu[b + 1]) - t0;
    if (q_base >= len) return;
    const int64_t ld = (int64_t)3 * H * kDh;
    const float* __restrict__ Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Kb = Qb + H * kDh;
    const float* __restrict__ Vb = Qb + 2 * H * kDh;
    const int rows = (len - q_base) < QROWS ? (len - q_base) : QROWS;
    const int ntq = (rows + 31) >> 5;  // 32-row query tiles of this block: tile t belongs to wave t & 3

    int q0[QT];
    bool act[QT];  // wave-uniform
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        act[j] = wave + 4 * j < ntq;
        q0[j] = q_base + 32 * (wave + 4 * j);
    }

    // Q pieces as the B operand of S^T = K Q^T: k-step s covers d = 16 s + 8 lh + j; pre-scaled by scale * log2(e)
    u32x4 qp[QT][4][NP];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        if (act[j]) {
            const int qi = q0[j] + lr;
            const int qrow = qi < len ? qi : len - 1;
            const float* qr = Qb + (int64_t)qrow * ld + 8 * lh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 x0 = *reinterpret_cast<const float4*>(qr + 16 * s);
                const float4 x1 = *reinterpret_cast<const float4*>(qr + 16 * s + 4);
                const float x[8] = {x0.x * scale_log2e, x0.y * scale_log2e, x0.z * scale_log2e, x0.w * scale_log2e,
                                    x1.x * scale_log2e, x1.y * scale_log2e, x1.z * scale_log2e, x1.w * scale_log2e};
                split8<NP, F16>(x, qp[j][s]);  // once per workgroup: the plain form (the compiler contracts q * scale - hi into one fma there; keep its bits)
            }
        }
    }

    // key-tile schedule of the workgroup
    int k_lo = 0, k_hi = len;
    if (WINDOWED) {
        k_lo = q_base - window;
        k_lo = k_lo < 0 ? 0 : (k_lo & ~31);
        k_hi = q_base + rows - 1 + window + 1;
        k_hi = k_hi > len ? len : k_hi;
    }
    const bool extra0 = WINDOWED && k_lo > 0;
    const int ntiles = (k_hi - k_lo + 31) / 32 + (extra0 ? 1 : 0);
    auto tile_base = [&](int it) { return extra0 ? (it == 0 ? 0 : k_lo + (it - 1) * 32) : k_lo + it * 32; };

    // staging: thread -> (row sr / sr + 16, 4 consecutive d at sc) through range-checked buffer loads
    const int sr = tid >> 4, sc4 = tid & 15;
    const uint32_t ld4 = (uint32_t)ld * 4u;
    const uint32_t span = (uint32_t)(len - 1) * ld4 + kDh * 4u;  // bytes from (row 0, d 0) of this head to the end of its last row
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Kb), 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vb), 0, (int)span, 0x00020000);
    const uint32_t voff_t = (uint32_t)sr * ld4 + (uint32_t)sc4 * 16u;
    u32x4 kreg0, kreg1, vreg0, vreg1;
    auto gload = [&](int kb) {
        const uint32_t o0 = (uint32_t)kb * ld4 + voff_t, o1 = o0 + 16u * ld4;
        kreg0 = __builtin_amdgcn_raw_buffer_load_b128(krs, o0, 0, 0);
        kreg1 = __builtin_amdgcn_raw_buffer_load_b128(krs, o1, 0, 0);
        vreg0 = __builtin_amdgcn_raw_buffer_load_b128(vrs, o0, 0, 0);
        vreg1 = __builtin_amdgcn_raw_buffer_load_b128(vrs, o1, 0, 0);
    };
    const int kw0 = sr * KROWB + ((((sc4 >> 1) ^ ((sr >> 1) & 7)) << 4) | ((sc4 & 1) << 3));
    const int kw1 = (sr + 16) * KROWB + ((((sc4 >> 1) ^ (((sr + 16) >> 1) & 7)) << 4) | ((sc4 & 1) << 3));
    auto store_k = [&](const u32x4 xb, unsigned char* dst) {
        float r0 = __uint_as_float(xb[0]), r1 = __uint_as_float(xb[1]), r2 = __uint_as_float(xb[2]), r3 = __uint_as_float(xb[3]);
        if (F16) {  // NP == 2
            const uint32_t w0 = pack2h(r0, r1), w1 = pack2h(r2, r3);
            *reinterpret_cast<uint2*>(dst) = make_uint2(w0, w1);
            *reinterpret_cast<uint2*>(dst + KPIECE) = make_uint2(resid2h(r0, r1, w0), resid2h(r2, r3, w1));
            return;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t w0 = pk<F16>(r0, r1), w1 = pk<F16>(r2, r3);
            *reinterpret_cast<uint2*>(dst + p * KPIECE) = make_uint2(w0, w1);
            if (p + 1 < NP) { r0 -= lo_v<F16>(w0); r1 -= hi_v<F16>(w0); r2 -= lo_v<F16>(w1); r3 -= hi_v<F16>(w1); }
        }
    };
    const int vw0 = sr * KROWB + ((((sc4 >> 1) ^ (((sr >> 1) & 1) << 2)) << 4) | ((sc4 & 1) << 3));
    const int vw1 = vw0 + 16 * KROWB;
    auto lstore = [&](unsigned char* buf) {
        store_k(kreg0, buf + kw0);
        store_k(kreg1, buf + kw1);
        store_k(vreg0, buf + NP * KPIECE + vw0);
        store_k(vreg1, buf + NP * KPIECE + vw1);
    };
    int vtr[2];
    {
        const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, g1 = (lane >> 4) & 1, bsw = (q >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            vtr[dt] = (4 * lh + q) * KROWB + ((4 * (dt ^ bsw) + 2 * g1 + (p4 >> 1)) << 4) + ((p4 & 1) << 3);
    }
    const int kswz = (lr >> 1) & 7;

    float m[QT], l[QT];
    f32x16 o[QT][2];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[j][0][r] = 0.f; o[j][1][r] = 0.f; }
    }

    MR_PH_DECL
    gload(tile_base(0));
    lstore(lds);
    __syncthreads();
    MR_PH(0)

    for (int it = 0; it < ntiles; ++it) {
        const unsigned char* buf = lds + (it & 1) * BUFB;
        const int kb = tile_base(it);
        gload(tile_base(it + 1 < ntiles ? it + 1 : it));  // unconditional: keeps the loads in flight under the math
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        bool rel[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            rel[j] = act[j];
            if (WINDOWED) rel[j] = rel[j] && (kb == 0 || (kb + 31 >= q0[j] - window && kb <= q0[j] + 31 + window));
        }
        const bool both = QT == 2 && rel[0] && rel[QT - 1];
        bool staged = false;
        if (PIPED && both && kb + 32 <= len) {
            // ---- interior key tile, both query tiles live (the common case of full attention): the four products and the two softmaxes
            // in software-pipelined order, so that the vector work of one query tile issues under the matrix work of the other --
            //   A  S0 = K Q0^T            B  S1 = K Q1^T   || softmax(S0) -> P0      C  O0 += V^T P0^T || softmax(S1) -> P1
            //   D  O1 += V^T P1^T || split + store of the next K / V tile
            // Per (query row, key tile) the operations and their order are those of the general path below: bit-identical results.
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
            auto kfrag = [&](int st, u32x4 (&ka)[NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ka[p] = *reinterpret_cast<const u32x4*>(buf + p * KPIECE + lr * KROWB + (((2 * st + lh) ^ kswz) << 4));
            };
            const unsigned char* vbase = buf + NP * KPIECE;
            auto vfrag = [&](int dt, int st, u32x4 (&va)[NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned char* a0 = vbase + p * KPIECE + vtr[dt] + (16 * st) * KROWB;
                    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (__attribute__((address_space(3))) s16x4*)a0));
                    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (__attribute__((address_space(3))) s16x4*)(a0 + 8 * KROWB)));
                    va[p][0] = lo.x; va[p][1] = lo.y; va[p][2] = hi.x; va[p][3] = hi.y;
                }
            };
            // branch-free softmax of an unmasked tile: returns the rescale factor of the running output
            auto softmax_in = [&](auto J, const f32x16& sc, u32x4 (&pq)[2][NP]) -> float {
                constexpr int j = decltype(J)::value;
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
                float ma, mb;
                half_pair(mx, ma, mb);
                mx = fmaxf(ma, mb);
                const float m_new = fmaxf(m[j], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float corr =
                    (m[j] == -INFINITY) ? ((m_new == -INFINITY) ? 1.f : 0.f) : __builtin_amdgcn_exp2f(m[j] - m_use);
                const float sub = F16 ? m_use - 10.f : m_use;
                float ps = 0.f;
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = __builtin_amdgcn_exp2f(sc[r] - sub);
                    ps += pv[r];
                }
                float pa, pb;
                half_pair(ps, pa, pb);
                ps = pa + pb;
                l[j] = l[j] * corr + ps;
                m[j] = m_new;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a = pv[8 * st + 2 * q], c = pv[8 * st + 2 * q + 1];
                        const uint32_t w = pk<F16>(a, c);
                        pq[st][0][q] = w;
                        pq[st][1][q] = F16 ? resid2h(a, c, w) : pk<F16>(a - lo_v<F16>(w), c - hi_v<F16>(w));
                    }
                return corr;
            };
            u32x4 pp0[2][NP], pp1[2][NP];
            // A: all eight K fragment loads in flight together
            u32x4 ka[4][NP];
#pragma unroll
            for (int st = 0; st < 4; ++st) kfrag(st, ka[st]);
#pragma unroll
            for (int st = 0; st < 4; ++st) s0 = mfma_split<NP, F16>(ka[st], qp[0][st], s0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            __builtin_amdgcn_sched_barrier(0);
            // B
#pragma unroll
            for (int st = 0; st < 4; ++st) kfrag(st, ka[st]);
#pragma unroll
            for (int st = 0; st < 4; ++st) s1 = mfma_split<NP, F16>(ka[st], qp[QT - 1][st], s1);
            const float corr0 = softmax_in(QIdx<0>{}, s0, pp0);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (__builtin_amdgcn_ballot_w64(corr0 != 1.f) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][0][r] *= corr0; o[0][1][r] *= corr0; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // C
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    u32x4 va[NP];
                    vfrag(dt, st, va);
                    o[0][dt] = mfma_split<NP, F16>(va, pp0[st], o[0][dt]);
                }
            const float corr1 = softmax_in(QIdx<QT - 1>{}, s1, pp1);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (__builtin_amdgcn_ballot_w64(corr1 != 1.f) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[QT - 1][0][r] *= corr1; o[QT - 1][1][r] *= corr1; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // D
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    u32x4 va[NP];
                    vfrag(dt, st, va);
                    o[QT - 1][dt] = mfma_split<NP, F16>(va, pp1[st], o[QT - 1][dt]);
                }
            lstore(lds + ((it + 1) & 1) * BUFB);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            staged = true;
        } else if (rel[0] || rel[QT - 1]) {
            // ---- S^T tiles = K Q^T: one read of the K fragments serves both query tiles
            f32x16 s[QT];
#pragma unroll
            for (int j = 0; j < QT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
            auto kfrag = [&](int st, u32x4 (&ka)[NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ka[p] = *reinterpret_cast<const u32x4*>(buf + p * KPIECE + lr * KROWB + (((2 * st + lh) ^ kswz) << 4));
            };
            if (both) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    u32x4 ka[NP];
                    kfrag(st, ka);
                    s[0] = mfma_split<NP, F16>(ka, qp[0][st], s[0]);
                    s[QT - 1] = mfma_split<NP, F16>(ka, qp[QT - 1][st], s[QT - 1]);
                }
            } else if (rel[0]) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    u32x4 ka[NP];
                    kfrag(st, ka);
                    s[0] = mfma_split<NP, F16>(ka, qp[0][st], s[0]);
                }
            } else {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    u32x4 ka[NP];
                    kfrag(st, ka);
                    s[QT - 1] = mfma_split<NP, F16>(ka, qp[QT - 1][st], s[QT - 1]);
                }
            }
            MR_PH(1)
            // ---- mask + online softmax (base 2) per query tile; s[r] is key kb + (r&3) + 8*(r>>2) + 4*lh for query q0 + lr
            u32x4 pp[QT][2][NP];
            auto softmax = [&](auto J) {
                constexpr int j = decltype(J)::value;
                const int qi = q0[j] + lr;
                float mx = -INFINITY;
                if (!WINDOWED && kb + 32 <= len) {  // interior tile of full attention: every key valid, no masking work
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        bool ok = key < len;
                        if (WINDOWED) {
                            const int dlt = qi - key;
                            ok = ok && (key == 0 || (dlt <= window && dlt >= -window));
                        }
                        s[j][r] = ok ? s[j][r] : -INFINITY;
                        mx = fmaxf(mx, s[j][r]);
                    }
                }
                {
                    float ma, mb;
                    half_pair(mx, ma, mb);
                    mx = fmaxf(ma, mb);
                }
                const float m_new = fmaxf(m[j], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float corr =
                    (m[j] == -INFINITY) ? ((m_new == -INFINITY) ? 1.f : 0.f) : __builtin_amdgcn_exp2f(m[j] - m_use);
                float ps = 0.f;
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // raw v_exp_f32 (arguments <= 0; tiny results flush to 0).  F16: probabilities carry a factor 2^10 (p' = 2^10 p <= 1024;
                    // the row sum l and the output accumulate the same factor, which cancels in o / l) so that the fp16 low piece of an
                    // ordinary 1e-3 probability is a NORMAL number: without it p is only held to an absolute 2^-25
                    pv[r] = __builtin_amdgcn_exp2f(s[j][r] - (F16 ? m_use - 10.f : m_use));
                    ps += pv[r];
                }
                {
                    float pa, pb;
                    half_pair(ps, pa, pb);
                    ps = pa + pb;
                }
                l[j] = l[j] * corr + ps;
                m[j] = m_new;
                if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0) {  // wave-uniform: the running maxima usually stop moving early
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[j][0][r] *= corr; o[j][1][r] *= corr; }
                }
                if (DROP) {
                    const uint32_t drow = (uint32_t)(t0 + qi) * (uint32_t)H + (uint32_t)h;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t key = (uint32_t)(kb + (r & 3) + 8 * (r >> 2) + 4 * lh);
                        pv[r] = mr::dropout_keep(drop_key, drow, key, drop_thresh) ? pv[r] * drop_inv : 0.f;
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const float x[8] = {pv[8 * st], pv[8 * st + 1], pv[8 * st + 2], pv[8 * st + 3],
                                        pv[8 * st + 4], pv[8 * st + 5], pv[8 * st + 6], pv[8 * st + 7]};
                    split8w<NP, F16>(x, pp[j][st]);
                }
            };
            if (rel[0]) softmax(QIdx<0>{});
            if (QT == 2 && rel[QT - 1]) softmax(QIdx<QT - 1>{});
            MR_PH(2)
            // ---- O^T += V^T P^T: A operand element j of k-step st is V[kb + (j & 3) + 8 (2 st + (j >> 2)) + 4 lh][d]
            const unsigned char* vbase = buf + NP * KPIECE;
            auto vfrag = [&](int dt, int st, u32x4 (&va)[NP]) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned char* a0 = vbase + p * KPIECE + vtr[dt] + (16 * st) * KROWB;
                    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (__attribute__((address_space(3))) s16x4*)a0));
                    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (__attribute__((address_space(3))) s16x4*)(a0 + 8 * KROWB)));
                    va[p][0] = lo.x; va[p][1] = lo.y; va[p][2] = hi.x; va[p][3] = hi.y;
                }
            };
            if (both) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        u32x4 va[NP];
                        vfrag(dt, st, va);
                        o[0][dt] = mfma_split<NP, F16>(va, pp[0][st], o[0][dt]);
                        o[QT - 1][dt] = mfma_split<NP, F16>(va, pp[QT - 1][st], o[QT - 1][dt]);
                    }
            } else if (rel[0]) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        u32x4 va[NP];
                        vfrag(dt, st, va);
                        o[0][dt] = mfma_split<NP, F16>(va, pp[0][st], o[0][dt]);
                    }
            } else {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        u32x4 va[NP];
                        vfrag(dt, st, va);
                        o[QT - 1][dt] = mfma_split<NP, F16>(va, pp[QT - 1][st], o[QT - 1][dt]);
                    }
            }
        }
        MR_PH(3)
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) lstore(lds + ((it + 1) & 1) * BUFB);
        MR_PH(4)
        __syncthreads();
        MR_PH(5)
    }
    MR_PH_FLUSH(blockIdx.x)

#pragma unroll
    for (int j = 0; j < QT; ++j) {
        const int qi = q0[j] + lr;
        if (act[j] && qi < len && !(WINDOWED && qi == 0)) {
            const float inv = 1.0f / l[j];
            float* op = ctx + (int64_t)(t0 + qi) * ((int64_t)H * kDh) + h * kDh + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o[j][0][4 * g] * inv, o[j][0][4 * g + 1] * inv,
                                                                     o[j][0][4 * g + 2] * inv, o[j][0][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + 32 + 8 * g) = make_float4(o[j][1][4 * g] * inv, o[j][1][4 * g + 1] * inv,
                                                                          o[j][1][4 * g + 2] * inv, o[j][1][4 * g + 3] * inv);
            }
        }
    }
}

}  // namespace

extern "C" int mr_attn_split_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh,
                                 int max_len, float scale, int window, int products, float* ctx, mr_stream_t stream) {
    if (!qkv || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh || (products != 3 && products != 6 && products != MR_PRODUCTS_F16X3)) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const dim3 grid((max_len + 127) / 128, H, B);
    const float scale_log2e = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define MR_ATTN_LAUNCH(W_, NP_, F_)                                                                                          \
    hipLaunchKernelGGL((attn_split_kernel<W_, NP_, F_>), grid, dim3(kThreads), (size_t)2 * (2 * NP_ * KPIECE), st, qkv,       \
                       cu_seqlens, seq_order, H, scale_log2e, window, ctx)
    if (window >= 0) {
        if (products == MR_PRODUCTS_F16X3) MR_ATTN_LAUNCH(true, 2, true); else if (products == 3) MR_ATTN_LAUNCH(true, 2, false); else MR_ATTN_LAUNCH(true, 3, false);
    } else {
        if (products == MR_PRODUCTS_F16X3) MR_ATTN_LAUNCH(false, 2, true); else if (products == 3) MR_ATTN_LAUNCH(false, 2, false); else MR_ATTN_LAUNCH(false, 3, false);
    }
#undef MR_ATTN_LAUNCH
    return mr::check_launch();
}

// ---- work-list form ------------------------------------------------------------------------------------------------------------
extern "C" int mr_attn_split_q_rows(int window, int products) {
    if (products != 3 && products != 6 && products != MR_PRODUCTS_F16X3) return MR_EUNSUPPORTED;
    return (products != 6 && window < 0) ? 256 : 128;
}

extern "C" int64_t mr_attn_work_plan(const int64_t* lens, int B, int q_rows, int32_t* work, int64_t capacity) {
    // host side: lens[b] (host memory) -> work[n_slots][8] (host memory); returns n_slots, also when work == NULL or too small
    // (then nothing is written: call again with n_slots * 8 entries); < 0 on bad arguments
    if (B < 0 || (B > 0 && !lens) || (q_rows != 128 && q_rows != 256) || B > 0xffffff) return MR_EINVAL;
    std::vector<int32_t> order((size_t)B);
    for (int i = 0; i < B; ++i) {
        order[(size_t)i] = i;
        if (lens[i] < 0 || (lens[i] + q_rows - 1) / q_rows > 127) return MR_EINVAL;
    }
    // sequences by decreasing length (stable), dealt over the 8 queues in snake order: near-equal key-tile totals per queue
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) { return lens[a] > lens[c]; });
    auto queue_of = [](int i) { return ((i >> 3) & 1) ? 7 - (i & 7) : (i & 7); };
    int64_t qn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < B; ++i) qn[queue_of(i)] += (lens[order[(size_t)i]] + q_rows - 1) / q_rows;
    const int64_t n_slots = *std::max_element(qn, qn + 8);
    if (!work || capacity < n_slots * 8) return n_slots;
    std::fill(work, work + n_slots * 8, -1);
    int64_t fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < B; ++i) {
        const int b = order[(size_t)i], x = queue_of(i);
        const int nb = (int)((lens[b] + q_rows - 1) / q_rows);
        for (int qb = 0; qb < nb; ++qb) work[(fill[x]++) * 8 + x] = b | (qb << 24);
    }
    return n_slots;
}

static int attn_split_work_launch(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh,
                                  float scale, int window, int products, float* ctx, uint32_t thresh, float inv, uint32_t key, mr_stream_t stream);

extern "C" int mr_attn_split_work_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H,
                                      int dh, float scale, int window, int products, float* ctx, mr_stream_t stream) {
    return attn_split_work_launch(qkv, cu_seqlens, work, n_slots, B, q_rows, H, dh, scale, window, products, ctx, 0u, 1.f, 0u, stream);
}

// training-graph form: dropout on the attention probabilities (drop_p in [0, 1); 0 = mr_attn_split_work_f32, bit for bit)
extern "C" int mr_attn_split_work_train_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows,
                                            int H, int dh, float scale, int window, int products, float drop_p, uint32_t drop_key, float* ctx,
                                            mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    return attn_split_work_launch(qkv, cu_seqlens, work, n_slots, B, q_rows, H, dh, scale, window, products, ctx, thresh, inv, drop_key, stream);
}

static int attn_split_work_launch(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh,
                                  float scale, int window, int products, float* ctx, uint32_t thresh, float inv, uint32_t key, mr_stream_t stream) {
    if (!qkv || !cu_seqlens || !ctx || n_slots < 0 || B < 0 || H < 1 || (n_slots > 0 && !work)) return MR_EINVAL;
    if (dh != kDh || (products != 3 && products != 6 && products != MR_PRODUCTS_F16X3)) return MR_EUNSUPPORTED;
    // the list must have been planned for THIS kernel's block height (an entry holds a block index, not a row) and for this batch
    if (q_rows != mr_attn_split_q_rows(window, products)) return MR_EINVAL;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (n_slots == 0) return MR_OK;
    if (n_slots * 8 * (int64_t)H > 0x7fffffff) return MR_EINVAL;
    const dim3 grid((unsigned)(n_slots * 8 * H));
    const float scale_log2e = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define MR_ATTN_LAUNCH(W_, NP_, QT_, D_, F_)                                                                                         \
    hipLaunchKernelGGL((attn_split_work_kernel<W_, NP_, QT_, D_, F_>), grid, dim3(kThreads), (size_t)2 * (2 * NP_ * KPIECE), st, qkv, \
                       cu_seqlens, work, H, scale_log2e, window, ctx, B, thresh, inv, key)
    const bool f16 = products == MR_PRODUCTS_F16X3;
    if (thresh == 0u) {
        if (window >= 0) {
            if (f16) MR_ATTN_LAUNCH(true, 2, 1, false, true); else if (products == 3) MR_ATTN_LAUNCH(true, 2, 1, false, false); else MR_ATTN_LAUNCH(true, 3, 1, false, false);
        } else {
            if (f16) MR_ATTN_LAUNCH(false, 2, 2, false, true); else if (products == 3) MR_ATTN_LAUNCH(false, 2, 2, false, false); else MR_ATTN_LAUNCH(false, 3, 1, false, false);
        }
    } else {  // training graph with dropout: the bf16x3 arithmetic only (the other forms are inference modes)
        if (products != 3) return MR_EUNSUPPORTED;
        if (window >= 0) MR_ATTN_LAUNCH(true, 2, 1, true, false); else MR_ATTN_LAUNCH(false, 2, 2, true, false);
    }
#undef MR_ATTN_LAUNCH
    return mr::check_launch();
}
