// K2 / K2b and the row-wise helpers of the encoder: token packing, embedding gather + LayerNorm,
// LayerNorm, CLS pooling + L2 normalise, row gather.  All HBM-bound: one wavefront per row,
// 16-byte loads, wave shuffles for the reductions, one write per output element.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / MR_WAVE;

// Optional input checks (err != nullptr): what the host used to verify with min / max reductions and a D2H sync per batch is
// tallied here as bits of one device word, read back at the next natural sync point (epoch end).  The gathers below clamp
// their indices, so a bad id can never fault; it is reported instead.
__global__ __launch_bounds__(MR_WAVE) void pack_tokens_kernel(const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ mask,
                                                             const int64_t* __restrict__ tt,
                                                             const int64_t* __restrict__ ip, int L, int pad_id,
                                                             const int32_t* __restrict__ cu,
                                                             int32_t* __restrict__ tok_word,
                                                             int32_t* __restrict__ tok_pos,
                                                             int32_t* __restrict__ tok_tt,
                                                             int32_t* __restrict__ tok_ip,
                                                             const int64_t* __restrict__ gmask, int vocab, int n_type,
                                                             int n_ip, int32_t* __restrict__ err) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t row = (int64_t)b * L;
    const int t0 = cu[b], t1 = cu[b + 1];
    int run_pos = 0, run_tok = 0;
    int bad = 0;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int l0 = 0; l0 < L; l0 += MR_WAVE) {
        const int l = l0 + lane;
        const bool valid = l < L;
        const int64_t id = valid ? ids[row + l] : (int64_t)pad_id;
        const bool m = valid && mask[row + l] != 0;
        const bool nonpad = valid && id != (int64_t)pad_id;
        const unsigned long long bnp = __ballot(nonpad), bm = __ballot(m);
        if (err && valid) {
            if (id < 0 || id >= (int64_t)vocab) bad |= MR_IN_BAD_ID;
            if (l == 0 && !m) bad |= MR_IN_NO_CLS;
            if (gmask && (gmask[row + l] != (l == 0 ? 1 : 0))) bad |= MR_IN_GLOBAL_PATTERN;
        }
        if (m) {
            const int t = t0 + run_tok + __popcll(bm & lt);
            const int64_t ttv = tt ? tt[row + l] : 0, ipv = ip ? ip[row + l] : 0;
            if (err) {
                if (tt && (ttv < 0 || ttv >= (int64_t)n_type)) bad |= MR_IN_BAD_TOKEN_TYPE;
                if (ip && (ipv < 0 || ipv >= (int64_t)n_ip)) bad |= MR_IN_BAD_ITEM_POS;
            }
            if (t < t1) {  // cu_seqlens is caller data: never write past this row's slot
                const int incl = run_pos + __popcll(bnp & lt) + 1;
                // what is stored is always a row of its table (the flag above reports the original value): the inference gathers clamp
                // again, the training graph's row gathers / scatter-adds and the row-sparse merge take these as they are
                tok_word[t] = (int32_t)(vocab > 0 ? (id < 0 ? 0 : (id >= (int64_t)vocab ? (int64_t)vocab - 1 : id)) : id);
                tok_pos[t] = nonpad ? incl + pad_id : pad_id;
                if (tok_tt) tok_tt[t] = (int32_t)(n_type > 0 ? (ttv < 0 ? 0 : (ttv >= (int64_t)n_type ? (int64_t)n_type - 1 : ttv)) : ttv);
                if (tok_ip) tok_ip[t] = (int32_t)(n_ip > 0 ? (ipv < 0 ? 0 : (ipv >= (int64_t)n_ip ? (int64_t)n_ip - 1 : ipv)) : ipv);
            }
        }
        run_pos += __popcll(bnp);
        run_tok += __popcll(bm);
    }
    if (err) {
        if (lane == 0 && run_tok != t1 - t0) bad |= MR_IN_LEN_MISMATCH;  // cu_seqlens disagrees with the mask
        if (bad) atomicOr(err, bad);
    }
}

__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// LayerNorm of a row held as NV float4 per lane (column = (j*64 + lane)*4); two-pass mean/variance.
template <int NV>
__device__ __forceinline__ void ln_row_store(float4 (&x)[NV], int d, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
    }
    const float mean = mr::wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) {
            const float a = x[j].x - mean, b = x[j].y - mean, cc = x[j].z - mean, dd = x[j].w - mean;
            v += (a * a + b * b) + (cc * cc + dd * dd);
        }
    }
    const float rstd = 1.0f / sqrtf(mr::wave_sum(v) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) {
            const float4 g = ld4(gamma + c), b = ld4(beta + c);
            float4 o;
            o.x = (x[j].x - mean) * rstd * g.x + b.x;
            o.y = (x[j].y - mean) * rstd * g.y + b.y;
            o.z = (x[j].z - mean) * rstd * g.z + b.z;
            o.w = (x[j].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4*>(out + c) = o;
        }
    }
}

// kTokPerWave consecutive tokens per wave: LayerNorm's gamma / beta and (RoBERTa mode) the single token-type row are loaded once per wave
// and kept in registers -- per token that leaves the word row (HBM) and the position row (L2) instead of five rows through L2, which is
// what bounded the one-token-per-wave form (1.3 GB of L2 reads per 0.2 GB of HBM reads at 69 k tokens).
constexpr int kTokPerWave = 2;

template <int NV>
__global__ __launch_bounds__(kThreads) void embed_gather_ln_kernel(
    const int32_t* __restrict__ tok_word, const int32_t* __restrict__ tok_pos, const int32_t* __restrict__ tok_tt,
    const int32_t* __restrict__ tok_ip, const float* __restrict__ word, const float* __restrict__ pos,
    const float* __restrict__ type, const float* __restrict__ itempos, int n_word, int n_pos, int n_type, int n_ip,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int T, int d, int mode,
    float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t0 = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * kTokPerWave;
    if (t0 >= T) return;
    float4 g[NV], b[NV], ty0[NV];
    const bool fixed_type = (tok_tt == nullptr);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        const bool in = c < d;
        g[j] = in ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[j] = in ? ld4(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ty0[j] = (in && fixed_type) ? ld4(type + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int u = 0; u < kTokPerWave; ++u) {
        const int t = t0 + u;
        if (t >= T) break;
        const float* wr = word + (int64_t)clampi(tok_word[t], n_word) * d;
        const float* pr = pos + (int64_t)clampi(tok_pos[t], n_pos) * d;
        const float* tr = fixed_type ? nullptr : type + (int64_t)clampi(tok_tt[t], n_type) * d;
        const float* ir = (mode == MR_EMBED_RECFORMER) ? itempos + (int64_t)clampi(tok_ip[t], n_ip) * d : nullptr;
        float4 x[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (j * MR_WAVE + lane) * 4;
            if (c < d) {
                const float4 w = ld4(wr + c), p = ld4(pr + c), ty = fixed_type ? ty0[j] : ld4(tr + c);
                if (mode == MR_EMBED_RECFORMER) x[j] = add4(add4(add4(w, p), ty), ld4(ir + c));  // recformer/models.py:131
                else x[j] = add4(add4(w, ty), p);  // RobertaEmbeddings: (inputs + token_type) + position
                s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            } else {
                x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // two-pass mean / variance exactly as ln_row_store
        const float mean = mr::wave_sum(s) / (float)d;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (j * MR_WAVE + lane) * 4;
            if (c < d) {
                const float a0 = x[j].x - mean, a1 = x[j].y - mean, a2 = x[j].z - mean, a3 = x[j].w - mean;
                v += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(mr::wave_sum(v) / (float)d + eps);
        float* o = out + (int64_t)t * d;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (j * MR_WAVE + lane) * 4;
            if (c < d) {
                float4 r;
                r.x = (x[j].x - mean) * rstd * g[j].x + b[j].x;
                r.y = (x[j].y - mean) * rstd * g[j].y + b[j].y;
                r.z = (x[j].z - mean) * rstd * g[j].z + b[j].z;
                r.w = (x[j].w - mean) * rstd * g[j].w + b[j].w;
                *reinterpret_cast<float4*>(o + c) = r;
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const float* __restrict__ xin, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int T, int d,
                                                            float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= T) return;
    float4 x[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        x[j] = (c < d) ? ld4(xin + (int64_t)t * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ln_row_store<NV>(x, d, gamma, beta, eps, out + (int64_t)t * ldo);
}

template <int NV>
__global__ __launch_bounds__(kThreads) void cls_pool_kernel(const float* __restrict__ xin, int64_t ldx,
                                                           const int32_t* __restrict__ cu, int B, int d,
                                                           int normalize, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* r = xin + (int64_t)(cu ? cu[b] : b) * ldx;  // cu == NULL: row b (the rows are already one per sequence)
    float4 x[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        x[j] = (c < d) ? ld4(r + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
    }
    float inv = 1.0f;
    if (normalize) inv = 1.0f / fmaxf(sqrtf(mr::wave_sum(s)), 1e-12f);  // F.normalize eps
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) {
            float4 o = x[j];
            if (normalize) { o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv; }
            *reinterpret_cast<float4*>(out + (int64_t)b * d + c) = o;
        }
    }
}

// pooling_method = "mean" (encoder/_base.py:42-43: ``last_hidden_state.mean(dim=1)`` over the PADDED batch length, pad positions included):
// out[b] = (sum of sequence b's token rows + (pad_len[b] - len_b) * xpad[b]) / pad_len[b], optionally L2-normalised.  xpad[b] is the hidden
// state every pad position of sequence b has (one query row per sequence carried through the layers, engine.forward_packed); one wave per
// sequence, tokens summed in ascending order.
template <int NV>
__global__ __launch_bounds__(kThreads) void mean_pool_kernel(const float* __restrict__ xin, int64_t ldx, const int32_t* __restrict__ cu,
                                                            const float* __restrict__ xpad, const int32_t* __restrict__ pad_len, int B, int d,
                                                            int normalize, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;
    const int t0 = cu[b], t1 = cu[b + 1];
    float4 x[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t0; t < t1; ++t) {
        const float* r = xin + (int64_t)t * ldx;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (j * MR_WAVE + lane) * 4;
            if (c < d) x[j] = add4(x[j], ld4(r + c));
        }
    }
    const int width = pad_len[b] > t1 - t0 ? pad_len[b] : t1 - t0;  // never narrower than the sequence itself
    const float npad = (float)(width - (t1 - t0)), inv_w = 1.0f / (float)(width > 0 ? width : 1);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) {
            if (npad > 0.f) {
                const float4 p = ld4(xpad + (int64_t)b * d + c);
                x[j].x += npad * p.x; x[j].y += npad * p.y; x[j].z += npad * p.z; x[j].w += npad * p.w;
            }
            x[j].x *= inv_w; x[j].y *= inv_w; x[j].z *= inv_w; x[j].w *= inv_w;
            s += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
        }
    }
    float inv = 1.0f;
    if (normalize) inv = 1.0f / fmaxf(sqrtf(mr::wave_sum(s)), 1e-12f);  // F.normalize eps
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * MR_WAVE + lane) * 4;
        if (c < d) {
            float4 o = x[j];
            if (normalize) { o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv; }
            *reinterpret_cast<float4*>(out + (int64_t)b * d + c) = o;
        }
    }
}

__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ xin, int64_t ldx,
                                                              const int32_t* __restrict__ idx, int n, int d,
                                                              float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* r = xin + (int64_t)idx[i] * ldx;
    for (int c = lane * 4; c < d; c += MR_WAVE * 4) *reinterpret_cast<float4*>(out + (int64_t)i * ldo + c) = ld4(r + c);
}

// any width / alignment (teacher-score rows have the catalog's length): one dword per lane, still coalesced.  A wave copies one
// kScalarChunk-column piece of a row (blockIdx.y = piece): the alpha-learning step gathers a handful of catalog-length rows, which one
// wave per row turns into a few hundred dependent loads
constexpr int kScalarChunk = 2048;
__global__ __launch_bounds__(kThreads) void gather_rows_scalar_kernel(const float* __restrict__ xin, int64_t ldx,
                                                                     const int32_t* __restrict__ idx, int n, int d,
                                                                     float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* r = xin + (int64_t)idx[i] * ldx;
    const int c0 = blockIdx.y * kScalarChunk, c1 = (c0 + kScalarChunk < d) ? c0 + kScalarChunk : d;
    for (int c = c0 + lane; c < c1; c += MR_WAVE) out[(int64_t)i * ldo + c] = r[c];
}

#define MR_DISPATCH_NV(d, CALL)                                  \
    do {                                                         \
        const int nv_ = ((d) / 4 + MR_WAVE - 1) / MR_WAVE;       \
        switch (nv_) {                                           \
            case 1: { constexpr int NV = 1; CALL; } break;       \
            case 2: { constexpr int NV = 2; CALL; } break;       \
            case 3: { constexpr int NV = 3; CALL; } break;       \
            case 4: { constexpr int NV = 4; CALL; } break;       \
            case 5: case 6: case 7: case 8: { constexpr int NV = 8; CALL; } break; \
            default: return MR_EUNSUPPORTED;                     \
        }                                                        \
    } while (0)

}  // namespace

extern "C" int mr_pack_tokens_checked(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                                      const int64_t* item_position_ids, const int64_t* global_attention_mask, int B, int L,
                                      int pad_id, int vocab, int n_type, int n_ip, const int32_t* cu_seqlens, int32_t* tok_word,
                                      int32_t* tok_pos, int32_t* tok_tt, int32_t* tok_ip, int32_t* err_bits,
                                      mr_stream_t stream) {
    if (!input_ids || !attention_mask || !cu_seqlens || !tok_word || !tok_pos || B < 0 || L < 0) return MR_EINVAL;
    if (err_bits && (vocab < 1 || (token_type_ids && n_type < 1) || (item_position_ids && n_ip < 1))) return MR_EINVAL;
    if (B == 0 || L == 0) return MR_OK;
    hipLaunchKernelGGL(pack_tokens_kernel, dim3(B), dim3(MR_WAVE), 0, (hipStream_t)stream, input_ids, attention_mask,
                       token_type_ids, item_position_ids, L, pad_id, cu_seqlens, tok_word, tok_pos, tok_tt, tok_ip,
                       global_attention_mask, vocab, n_type, n_ip, err_bits);
    return mr::check_launch();
}

extern "C" int mr_pack_tokens(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                              const int64_t* item_position_ids, int B, int L, int pad_id, const int32_t* cu_seqlens,
                              int32_t* tok_word, int32_t* tok_pos, int32_t* tok_tt, int32_t* tok_ip,
                              mr_stream_t stream) {
    return mr_pack_tokens_checked(input_ids, attention_mask, token_type_ids, item_position_ids, nullptr, B, L, pad_id, 0, 0, 0,
                                  cu_seqlens, tok_word, tok_pos, tok_tt, tok_ip, nullptr, stream);
}

extern "C" int mr_embed_gather_ln_f32(const int32_t* tok_word, const int32_t* tok_pos, const int32_t* tok_tt,
                                      const int32_t* tok_ip, const float* word, const float* pos, const float* type,
                                      const float* itempos, int n_word, int n_pos, int n_type, int n_ip,
                                      const float* gamma, const float* beta, float eps, int T, int d, int mode,
                                      float* out, mr_stream_t stream) {
    if (!tok_word || !tok_pos || !word || !pos || !type || !gamma || !beta || !out || T < 0 || d <= 0) return MR_EINVAL;
    if (mode == MR_EMBED_RECFORMER && (!itempos || !tok_ip || !tok_tt)) return MR_EINVAL;
    if (mode != MR_EMBED_ROBERTA && mode != MR_EMBED_RECFORMER) return MR_EUNSUPPORTED;
    if (n_word < 1 || n_pos < 1 || n_type < 1 || (mode == MR_EMBED_RECFORMER && n_ip < 1)) return MR_EINVAL;
    if ((d & 3) || d > 2048) return MR_EUNSUPPORTED;
    if (!mr::aligned16(word) || !mr::aligned16(pos) || !mr::aligned16(type) || !mr::aligned16(gamma) ||
        !mr::aligned16(beta) || !mr::aligned16(out) || (itempos && !mr::aligned16(itempos)))
        return MR_EALIGN;
    if (T == 0) return MR_OK;
    const unsigned blocks = (unsigned)((T + kWavesPerBlock * kTokPerWave - 1) / (kWavesPerBlock * kTokPerWave));
    MR_DISPATCH_NV(d, hipLaunchKernelGGL((embed_gather_ln_kernel<NV>), dim3(blocks), dim3(kThreads), 0,
                                         (hipStream_t)stream, tok_word, tok_pos, tok_tt, tok_ip, word, pos, type, itempos,
                                         n_word, n_pos, n_type, n_ip, gamma, beta, eps, T, d, mode, out));
    return mr::check_launch();
}

extern "C" int mr_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int T,
                                int d, float* out, int64_t ldo, mr_stream_t stream) {
    if (!x || !gamma || !beta || !out || T < 0 || d <= 0) return MR_EINVAL;
    if ((d & 3) || d > 2048) return MR_EUNSUPPORTED;
    if ((ldx & 3) || (ldo & 3) || !mr::aligned16(x) || !mr::aligned16(out) || !mr::aligned16(gamma) || !mr::aligned16(beta))
        return MR_EALIGN;
    if (T == 0) return MR_OK;
    const unsigned blocks = (unsigned)((T + kWavesPerBlock - 1) / kWavesPerBlock);
    MR_DISPATCH_NV(d, hipLaunchKernelGGL((layernorm_kernel<NV>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x,
                                         ldx, gamma, beta, eps, T, d, out, ldo));
    return mr::check_launch();
}

extern "C" int mr_cls_pool_normalize_f32(const float* x, int64_t ldx, const int32_t* cu_seqlens, int B, int d,
                                         int normalize, float* out, mr_stream_t stream) {
    if (!x || !out || B < 0 || d <= 0) return MR_EINVAL;
    if ((d & 3) || d > 2048) return MR_EUNSUPPORTED;
    if ((ldx & 3) || !mr::aligned16(x) || !mr::aligned16(out)) return MR_EALIGN;
    if (B == 0) return MR_OK;
    const unsigned blocks = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    MR_DISPATCH_NV(d, hipLaunchKernelGGL((cls_pool_kernel<NV>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x,
                                         ldx, cu_seqlens, B, d, normalize, out));
    return mr::check_launch();
}

extern "C" int mr_mean_pool_f32(const float* x, int64_t ldx, const int32_t* cu_seqlens, const float* xpad, const int32_t* pad_len, int B, int d,
                                int normalize, float* out, mr_stream_t stream) {
    if (!x || !cu_seqlens || !xpad || !pad_len || !out || B < 0 || d <= 0) return MR_EINVAL;
    if ((d & 3) || d > 2048) return MR_EUNSUPPORTED;
    if ((ldx & 3) || !mr::aligned16(x) || !mr::aligned16(xpad) || !mr::aligned16(out)) return MR_EALIGN;
    if (B == 0) return MR_OK;
    const unsigned blocks = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    MR_DISPATCH_NV(d, hipLaunchKernelGGL((mean_pool_kernel<NV>), dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, cu_seqlens, xpad,
                                         pad_len, B, d, normalize, out));
    return mr::check_launch();
}

extern "C" int mr_gather_rows_f32(const float* x, int64_t ldx, const int32_t* row_idx, int n, int d, float* out,
                                  int64_t ldo, mr_stream_t stream) {
    if (!x || !row_idx || !out || n < 0 || d <= 0) return MR_EINVAL;
    if (n == 0) return MR_OK;
    const unsigned blocks = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    if ((d & 3) || (ldx & 3) || (ldo & 3) || !mr::aligned16(x) || !mr::aligned16(out)) {
        hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3(blocks, (unsigned)((d + kScalarChunk - 1) / kScalarChunk)), dim3(kThreads), 0,
                           (hipStream_t)stream, x, ldx, row_idx, n, d, out, ldo);
        return mr::check_launch();
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, row_idx, n, d, out, ldo);
    return mr::check_launch();
}
