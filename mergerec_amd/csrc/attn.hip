// K4 / K4b: fused multi-head self-attention over packed tokens, exact fp32 on the matrix cores.
//
// One wavefront owns a 32-query tile of one (sequence, head) and walks the key tiles with an online
// softmax.  Scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x2_f32, 32 steps over
// dh = 64), so the accumulator layout (column = query on the lane, rows = keys in the 16 registers)
// is already the B operand of the second product O^T = V^T P^T: P never leaves registers and no LDS
// or cross-lane traffic is needed except one lane^32 exchange for the row max / row sum.  K, Q and V
// are read straight from global memory in operand layout (Q/K: 128 contiguous bytes per lane;
// V: 128-byte segments per half-wave), which L2 serves: fp32 MFMA is slow enough (64 clk per
// instruction) that operand traffic is ~1 B/clk/CU.  Attention is ~6 % of the encoder's FLOPs.
#include "common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int kDh = 64;

template <bool WINDOWED>
__global__ __launch_bounds__(kThreads) void attn_kernel(const float* __restrict__ qkv,
                                                       const int32_t* __restrict__ cu, int H, float scale,
                                                       int window, float* __restrict__ ctx) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    const int q0 = (blockIdx.x * (kThreads / MR_WAVE) + wave) * 32;
    if (q0 >= len) return;  // whole wave exits together; no barriers in this kernel
    const int64_t ld = (int64_t)3 * H * kDh;
    const float* __restrict__ Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Kb = Qb + H * kDh;
    const float* __restrict__ Vb = Qb + 2 * H * kDh;

    // Q as the B operand of S^T = K Q^T: lane (query lr, half lh) holds Q[q][32*lh + s], s = 0..31
    const int qi = q0 + lr;
    const int qrow = qi < len ? qi : len - 1;
    float qv[32];
    {
        const float4* qp = reinterpret_cast<const float4*>(Qb + (int64_t)qrow * ld + 32 * lh);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float4 x = qp[v];
            qv[4 * v + 0] = x.x * scale;
            qv[4 * v + 1] = x.y * scale;
            qv[4 * v + 2] = x.z * scale;
            qv[4 * v + 3] = x.w * scale;
        }
    }

    int k_lo = 0, k_hi = len;
    if (WINDOWED) {
        k_lo = q0 - window;
        k_lo = k_lo < 0 ? 0 : (k_lo & ~31);
        k_hi = q0 + 31 + window + 1;
        k_hi = k_hi > len ? len : k_hi;
    }
    float m = -INFINITY, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    // key tiles: tile 0 first when the band does not reach it (the global key column), then the band
    const bool extra0 = WINDOWED && k_lo > 0;
    const int ntiles = (k_hi - k_lo + 31) / 32 + (extra0 ? 1 : 0);
    for (int it = 0; it < ntiles; ++it) {
        const int kb = extra0 ? (it == 0 ? 0 : k_lo + (it - 1) * 32) : k_lo + it * 32;
        // ---- S^T tile: A operand = K[kb + lr][32*lh + s]
        int krow = kb + lr;
        krow = krow < len ? krow : len - 1;
        const float4* kp = reinterpret_cast<const float4*>(Kb + (int64_t)krow * ld + 32 * lh);
        float4 kx[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) kx[v] = kp[v];
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx[v].x, qv[4 * v + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx[v].y, qv[4 * v + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx[v].z, qv[4 * v + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx[v].w, qv[4 * v + 3], s, 0, 0, 0);
        }
        // ---- mask + online softmax; s[r] is key kb + (r&3) + 8*(r>>2) + 4*lh for query q0 + lr
        float mx = -INFINITY;
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
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // nothing visible yet: p = 0, corr = 1
        const float corr = (m == -INFINITY) ? ((m_new == -INFINITY) ? 1.f : 0.f) : expf(m - m_use);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_use);  // exp(-inf) = 0 for masked keys
            ps += s[r];
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * corr + ps;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
        // ---- O^T += V^T P^T: A operand = V[kb + kappa(r, lh)][dt*32 + lr], B operand = s[r]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int vrow = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            vrow = vrow < len ? vrow : len - 1;  // p == 0 there
            const float* vp = Vb + (int64_t)vrow * ld + lr;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], s[r], o1, 0, 0, 0);
        }
    }

    if (qi < len && !(WINDOWED && qi == 0)) {
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

// Longformer global row: one wave per (sequence, head); scores staged in LDS.
__global__ __launch_bounds__(MR_WAVE) void attn_global_row_kernel(const float* __restrict__ qg,
                                                                 const float* __restrict__ kvg,
                                                                 const int32_t* __restrict__ cu, int H, float scale,
                                                                 float* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    if (len <= 0) return;
    const int64_t ld = (int64_t)2 * H * kDh;
    const float* __restrict__ q = qg + (int64_t)b * H * kDh + h * kDh;
    const float* __restrict__ Kg = kvg + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Vg = Kg + H * kDh;
    float qreg[kDh];
#pragma unroll
    for (int d = 0; d < kDh; ++d) qreg[d] = q[d] * scale;
    float mx = -INFINITY;
    for (int j = lane; j < len; j += MR_WAVE) {
        const float4* kp = reinterpret_cast<const float4*>(Kg + (int64_t)j * ld);
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < kDh / 4; ++v) {
            const float4 x = kp[v];
            acc = fmaf(qreg[4 * v], x.x, acc);
            acc = fmaf(qreg[4 * v + 1], x.y, acc);
            acc = fmaf(qreg[4 * v + 2], x.z, acc);
            acc = fmaf(qreg[4 * v + 3], x.w, acc);
        }
        sc[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = mr::wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < len; j += MR_WAVE) {
        const float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = mr::wave_sum(sum);
    __syncthreads();
    float out = 0.f;
    for (int j = 0; j < len; ++j) out = fmaf(sc[j], Vg[(int64_t)j * ld + lane], out);
    ctx[(int64_t)t0 * ((int64_t)H * kDh) + h * kDh + lane] = out / sum;
}

}  // namespace

extern "C" int mr_attn_f32(const float* qkv, const int32_t* cu_seqlens, int B, int H, int dh, int max_len, float scale,
                           int window, float* ctx, mr_stream_t stream) {
    if (!qkv || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const int qtiles = (max_len + 31) / 32;
    const dim3 grid((qtiles + 3) / 4, H, B);
    if (window >= 0)
        hipLaunchKernelGGL((attn_kernel<true>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, H, scale, window, ctx);
    else
        hipLaunchKernelGGL((attn_kernel<false>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, H, scale, window, ctx);
    return mr::check_launch();
}

extern "C" int mr_attn_global_row_f32(const float* qg, const float* kvg, const int32_t* cu_seqlens, int B, int H, int dh,
                                      int max_len, float scale, float* ctx, mr_stream_t stream) {
    if (!qg || !kvg || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (max_len > 16384) return MR_EUNSUPPORTED;  // scores of one row live in LDS
    if (!mr::aligned16(qg) || !mr::aligned16(kvg)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const size_t shm = ((size_t)max_len * sizeof(float) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(attn_global_row_kernel, dim3(H, B), dim3(MR_WAVE), shm, (hipStream_t)stream, qg, kvg, cu_seqlens, H, scale, ctx);
    return mr::check_launch();
}
