// K4 / K4b: fused multi-head self-attention over packed tokens, exact fp32 on the matrix cores.
//
// One wavefront owns a 32-query tile of one (sequence, head); the 4 waves of a workgroup cover 128
// consecutive queries and share the K/V tiles through LDS.  Each wave walks the key tiles with an online
// softmax.  Scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x2_f32, 32 steps over
// dh = 64), so the accumulator layout (column = query on the lane, rows = keys in the 16 registers)
// is already the B operand of the second product O^T = V^T P^T: P never leaves registers and no LDS
// round trip or cross-lane traffic is needed for it except one lane^32 exchange for the row max / row sum.
// Attention is ~6 % of the encoder's FLOPs.
#include "common.h"
#include "dropout.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int kDh = 64;

constexpr int kKS = 68;  // K tile row stride in floats: conflict-free ds_read_b128 of 16 distinct rows
constexpr int kVS = 64;  // V tile row stride (read with ds_read_b32, lanes along d)
constexpr int kTileFloats = 32 * kKS + 32 * kVS;

// One workgroup = 4 waves = 128 consecutive queries of one (sequence, head); wave w owns queries
// [q_base + 32w, +32).  All 256 threads stage each 32-key K and V tile through double-buffered LDS
// (coalesced 256-B rows; the next tile's global loads are issued before the current tile's 64 MFMAs and
// stored to the idle buffer after them, one barrier per tile), every wave reads its MFMA operands from LDS.
// DROP (training graph only): the normalised probabilities are masked by mr::dropout_keep(key, query token * H + head, key position)
// and scaled by 1 / (1 - p) before they multiply V -- the row sum l keeps the un-dropped probabilities, as softmax-then-dropout does.
template <bool WINDOWED, bool DROP = false>
__global__ __launch_bounds__(kThreads, 4) void attn_kernel(const float* __restrict__ qkv,
                                                       const int32_t* __restrict__ cu,
                                                       const int32_t* __restrict__ seq_order, int H,
                                                       float scale_log2e, int window, float* __restrict__ ctx,
                                                       uint32_t drop_thresh = 0u, float drop_inv = 1.f, uint32_t drop_key = 0u,
                                                       const int32_t* __restrict__ work = nullptr, int nseq = 0x7fffffff) {
    __shared__ __attribute__((aligned(16))) float lds[2][kTileFloats];
    // work != NULL: the 1-D work-list grid of mr_attn_split_work_f32 (csrc/attn_bf16.hip; 128-row blocks: only the (sequence, block) pairs
    // that exist, dealt over the XCD queues).  Else the (blocks, H, B) box; seq_order (optional) = sequence ids by decreasing length
    int b, h, q_base;
    if (work) {
        const int slot = blockIdx.x >> 3, e = slot / H, ent = work[e * 8 + (blockIdx.x & 7)];
        if (ent < 0) return;
        h = slot - e * H; b = ent & 0xffffff; q_base = (ent >> 24) * 128;
        if (b >= nseq) return;  // a list planned for another batch: never index cu[] past its B + 1 entries
    } else {
        b = seq_order ? seq_order[blockIdx.z] : (int)blockIdx.z; h = blockIdx.y; q_base = blockIdx.x * 128;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    if (q_base >= len) return;  // the whole workgroup leaves together, before any barrier
    const int64_t ld = (int64_t)3 * H * kDh;
    const float* __restrict__ Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Kb = Qb + H * kDh;
    const float* __restrict__ Vb = Qb + 2 * H * kDh;

    const int q0 = q_base + wave * 32;
    const bool wave_active = q0 < len;  // wave-uniform
    const int qi = q0 + lr;
    // Q as the B operand of S^T = K Q^T: lane (query lr, half lh) holds Q[q][32*lh + s], pre-scaled by
    // scale * log2(e) so that the softmax runs on exp2
    float qv[32];
    {
        const int qrow = qi < len ? qi : len - 1;
        const float4* qp = reinterpret_cast<const float4*>(Qb + (int64_t)qrow * ld + 32 * lh);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float4 x = qp[v];
            qv[4 * v + 0] = x.x * scale_log2e;
            qv[4 * v + 1] = x.y * scale_log2e;
            qv[4 * v + 2] = x.z * scale_log2e;
            qv[4 * v + 3] = x.w * scale_log2e;
        }
    }

    // key-tile schedule of the workgroup
    int k_lo = 0, k_hi = len;
    if (WINDOWED) {
        k_lo = q_base - window;
        k_lo = k_lo < 0 ? 0 : (k_lo & ~31);
        k_hi = q_base + 127 + window + 1;
        k_hi = k_hi > len ? len : k_hi;
    }
    const bool extra0 = WINDOWED && k_lo > 0;  // tile 0 carries the global key when the band does not reach it
    const int ntiles = (k_hi - k_lo + 31) / 32 + (extra0 ? 1 : 0);
    auto tile_base = [&](int it) { return extra0 ? (it == 0 ? 0 : k_lo + (it - 1) * 32) : k_lo + it * 32; };

    // staging map: thread -> (row sr / sr + 16, 16-byte column sc)
    const int sr = tid >> 4, sc = (tid & 15) * 4;
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
    auto lstore = [&](float* buf) {
        *reinterpret_cast<float4*>(buf + sr * kKS + sc) = kreg0;
        *reinterpret_cast<float4*>(buf + (sr + 16) * kKS + sc) = kreg1;
        *reinterpret_cast<float4*>(buf + 32 * kKS + sr * kVS + sc) = vreg0;
        *reinterpret_cast<float4*>(buf + 32 * kKS + (sr + 16) * kVS + sc) = vreg1;
    };

    float m = -INFINITY, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    gload(tile_base(0));
    lstore(lds[0]);
    __syncthreads();

    for (int it = 0; it < ntiles; ++it) {
        const float* buf = lds[it & 1];
        const int kb = tile_base(it);
        gload(tile_base(it + 1 < ntiles ? it + 1 : it));  // unconditional: keeps the loads in flight under the MFMAs
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        bool relevant = wave_active;
        if (WINDOWED) relevant = relevant && (kb == 0 || (kb + 31 >= q0 - window && kb <= q0 + 31 + window));
        if (relevant) {
            // ---- S^T tile = K Q^T: A operand = K[kb + lr][32*lh + s] from LDS
            const float4* kp = reinterpret_cast<const float4*>(buf + lr * kKS + 32 * lh);
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const float4 kx = kp[v];
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx.x, qv[4 * v + 0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx.y, qv[4 * v + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx.z, qv[4 * v + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kx.w, qv[4 * v + 3], s, 0, 0, 0);
            }
            // ---- mask + online softmax (base 2); s[r] is key kb + (r&3) + 8*(r>>2) + 4*lh for query q0 + lr
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
            const float corr = (m == -INFINITY) ? ((m_new == -INFINITY) ? 1.f : 0.f) : exp2f(m - m_use);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] - m_use);  // exp2(-inf) = 0 for masked keys
                ps += s[r];
            }
            ps += __shfl_xor(ps, 32, 64);
            l = l * corr + ps;
            m = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
            if (DROP) {
                const uint32_t drow = (uint32_t)(t0 + qi) * (uint32_t)H + (uint32_t)h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t key = (uint32_t)(kb + (r & 3) + 8 * (r >> 2) + 4 * lh);
                    s[r] = mr::dropout_keep(drop_key, drow, key, drop_thresh) ? s[r] * drop_inv : 0.f;
                }
            }
            // ---- O^T += V^T P^T: A operand = V[kb + kappa(r, lh)][dt*32 + lr] from LDS, B operand = s[r]
            const float* vp = buf + 32 * kKS + (4 * lh) * kVS + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vr = vp + ((r & 3) + 8 * (r >> 2)) * kVS;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[r], o1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        lstore(lds[(it + 1) & 1]);
        __syncthreads();
    }

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

// Longformer global row: one wave per (sequence, head); scores staged in LDS.
// kbase / vbase: column 0 of head 0 of the keys / values, row stride ld (kvg layout: vbase = kbase + H * 64, ld = 2 H 64; packed qkv:
// kbase = qkv + H * 64, vbase = qkv + 2 H * 64, ld = 3 H 64).  compact: write row b of a (B, H 64) matrix instead of row cu[b] of ctx.
// drop_thresh != 0 (training graph): probabilities masked by mr::dropout_keep(key, sequence * H + head, key position) after normalisation.
__global__ __launch_bounds__(MR_WAVE) void attn_global_row_kernel(const float* __restrict__ qg, const float* __restrict__ kbase,
                                                                 const float* __restrict__ vbase, int64_t ld,
                                                                 const int32_t* __restrict__ cu, int H, float scale,
                                                                 float* __restrict__ ctx, int compact, uint32_t drop_thresh = 0u,
                                                                 float drop_inv = 1.f, uint32_t drop_key = 0u) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    if (len <= 0) return;
    const float* __restrict__ q = qg + (int64_t)b * H * kDh + h * kDh;
    const float* __restrict__ Kg = kbase + (int64_t)t0 * ld + h * kDh;
    const float* __restrict__ Vg = vbase + (int64_t)t0 * ld + h * kDh;
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
        sum += p;
        sc[j] = (drop_thresh == 0u || mr::dropout_keep(drop_key, (uint32_t)b * (uint32_t)H + (uint32_t)h, (uint32_t)j, drop_thresh)) ? p * drop_inv : 0.f;
    }
    sum = mr::wave_sum(sum);
    __syncthreads();
    float out = 0.f;
    for (int j = 0; j < len; ++j) out = fmaf(sc[j], Vg[(int64_t)j * ld + lane], out);
    ctx[(int64_t)(compact ? b : t0) * ((int64_t)H * kDh) + h * kDh + lane] = out / sum;
}

}  // namespace

extern "C" int mr_attn_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh,
                           int max_len, float scale, int window, float* ctx, mr_stream_t stream) {
    if (!qkv || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const dim3 grid((max_len + 127) / 128, H, B);
    const float scale_log2e = scale * 1.4426950408889634f;
    if (window >= 0)
        hipLaunchKernelGGL((attn_kernel<true>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, seq_order, H, scale_log2e, window, ctx);
    else
        hipLaunchKernelGGL((attn_kernel<false>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, seq_order, H, scale_log2e, window, ctx);
    return mr::check_launch();
}

// exact-fp32 attention on the work-list grid (work / n_slots from mr_attn_work_plan(lens, B, 128, ...)); drop_p as mr_attn_train_f32.
// Same results as mr_attn_f32 / mr_attn_train_f32 bit for bit.
extern "C" int mr_attn_work_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh,
                                float scale, int window, float drop_p, uint32_t drop_key, float* ctx, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    if (!qkv || !cu_seqlens || !ctx || n_slots < 0 || B < 0 || H < 1 || (n_slots > 0 && !work)) return MR_EINVAL;
    if (q_rows != 128) return MR_EINVAL;  // this kernel's block height: a list planned with another one would skip or repeat rows
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (n_slots == 0) return MR_OK;
    if (n_slots * 8 * (int64_t)H > 0x7fffffff) return MR_EINVAL;
    const dim3 grid((unsigned)(n_slots * 8 * H));
    const float scale_log2e = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define MR_ATTN_W(W_, D_) hipLaunchKernelGGL((attn_kernel<W_, D_>), grid, dim3(kThreads), 0, st, qkv, cu_seqlens, nullptr, H, scale_log2e, window, ctx, \
                                             thresh, thresh ? inv : 1.f, drop_key, work, B)
    if (window >= 0) { if (thresh) MR_ATTN_W(true, true); else MR_ATTN_W(true, false); }
    else { if (thresh) MR_ATTN_W(false, true); else MR_ATTN_W(false, false); }
#undef MR_ATTN_W
    return mr::check_launch();
}

// training-graph forms: dropout on the attention probabilities (drop_p in [0, 1); 0 = the kernels above, bit for bit)
extern "C" int mr_attn_train_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh, int max_len,
                                 float scale, int window, float drop_p, uint32_t drop_key, float* ctx, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    if (thresh == 0u) return mr_attn_f32(qkv, cu_seqlens, seq_order, B, H, dh, max_len, scale, window, ctx, stream);
    if (!qkv || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const dim3 grid((max_len + 127) / 128, H, B);
    const float scale_log2e = scale * 1.4426950408889634f;
    if (window >= 0)
        hipLaunchKernelGGL((attn_kernel<true, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, seq_order, H, scale_log2e, window,
                           ctx, thresh, inv, drop_key);
    else
        hipLaunchKernelGGL((attn_kernel<false, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, qkv, cu_seqlens, seq_order, H, scale_log2e, window,
                           ctx, thresh, inv, drop_key);
    return mr::check_launch();
}

extern "C" int mr_attn_global_row_train_f32(const float* qg, const float* kvg, const int32_t* cu_seqlens, int B, int H, int dh, int max_len,
                                            float scale, float drop_p, uint32_t drop_key, float* ctx, int compact, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    if (!qg || !kvg || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (max_len > 16384) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qg) || !mr::aligned16(kvg)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const size_t shm = ((size_t)max_len * sizeof(float) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(attn_global_row_kernel, dim3(H, B), dim3(MR_WAVE), shm, (hipStream_t)stream, qg, kvg, kvg + H * kDh, (int64_t)2 * H * kDh,
                       cu_seqlens, H, scale, ctx, compact, thresh, thresh ? inv : 1.f, drop_key);
    return mr::check_launch();
}

extern "C" int mr_attn_global_row_f32(const float* qg, const float* kvg, const int32_t* cu_seqlens, int B, int H, int dh,
                                      int max_len, float scale, float* ctx, int compact, mr_stream_t stream) {
    if (!qg || !kvg || !cu_seqlens || !ctx || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (max_len > 16384) return MR_EUNSUPPORTED;  // scores of one row live in LDS
    if (!mr::aligned16(qg) || !mr::aligned16(kvg)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    const size_t shm = ((size_t)max_len * sizeof(float) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(attn_global_row_kernel, dim3(H, B), dim3(MR_WAVE), shm, (hipStream_t)stream, qg, kvg, kvg + H * kDh, (int64_t)2 * H * kDh,
                       cu_seqlens, H, scale, ctx, compact);
    return mr::check_launch();
}
