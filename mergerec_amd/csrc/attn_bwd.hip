// Attention backward on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): fine-tuning batches are token-sized (64 sequences of up
// to 512 tokens per micro-step), where the per-thread FMA formulation spends 60 % of the step.
//
// qkv (T, 3 H 64) = [Q | K | V] as the forward; ctx, dctx (T, H 64); rowstat (T, H, 2) = (logsumexp of the scaled scores, delta);
//     s_ij = scale q_i . k_j,  p_ij = exp(s_ij - lse_i),  delta_i = dO_i . O_i,  dS_ij = p_ij (dO_i . v_j - delta_i) scale
//     dQ_i = sum_j dS_ij k_j      dK_j = sum_i dS_ij q_i      dV_j = sum_i p_ij dO_i
// Longformer mode (window >= 0): query i sees key j iff j == 0 (global key) or |i - j| <= window; query row 0 belongs to the
// global-row kernel and sees nothing here.
//
// Three launches over (128-row tile, head, sequence; sequences in the caller's order -- longest first), four waves per workgroup,
// each wave owning 32 rows:
//   stats   query-owned: S^T = K Q^T tile by tile -> online logsumexp per query (one lane per query, 16 keys per lane and step)
//   dq      query-owned: S^T and dP^T = V dO^T  -> dS^T, which in the MFMA result layout IS the A operand of dQ += dS K
//   dkv     key-owned:   S = Q K^T and dP = dO V^T -> P, dS, which in the result layout ARE the A operands of dV += P^T dO,
//           dK += dS^T Q  (the k index of an MFMA may be enumerated in any order as long as both operands agree)
// so no score tile is ever transposed or written anywhere.  The owned rows live in registers as B operands (k = d enumerated as
// d = s + 32 * (lane / 32)); the other side's 32 x 64 tiles are staged in LDS with a 68-float pitch: rows along lanes are read
// with ds_read_b128 (4 k-steps per read), rows along the k index with ds_read_b32, both conflict-free, and always in batches of
// 8 - 16 reads ahead of the 16 - 32 MFMAs that consume them (one read per MFMA behind an lgkmcnt(0) leaves the matrix pipe 60 % idle).
// Deterministic: every output element has one owner, fixed order.
#include "common.h"
#include "dropout.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kDh = 64;
constexpr int kPitch = 68;  // floats: 16-byte aligned rows; 16 consecutive rows hit 64 distinct banks with ds_read_b128, and a row's
                           // consecutive floats are conflict-free ds_read_b32
constexpr int kTile = 32 * kPitch;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool allowed(int i, int j, int window) {
    if (window < 0) return true;
    if (i == 0) return false;
    const int dlt = i - j;
    return j == 0 || (dlt <= window && dlt >= -window);
}
// rows [r0, r0 + nr) against columns [c0, c0 + 32): can any (row, col) pair be allowed?
__device__ __forceinline__ bool tile_needed(int r0, int nr, int c0, int window) {
    if (window < 0 || c0 == 0) return true;
    return c0 <= r0 + nr - 1 + window && c0 + 31 >= r0 - window;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// result-layout row of accumulator register i for this lane's half
__device__ __forceinline__ int crow(int i, int lh) { return (i & 3) + 8 * (i >> 2) + 4 * lh; }

// staging of rows [row0, row0 + 32) x 64 floats of a (T, ld) matrix (column offset folded into src), rows clamped to len - 1, in two
// halves so that the global loads of the NEXT tile are in flight while the current one is being multiplied:
//   fetch_tile: this thread's 8 floats -> registers;   put_tile: registers -> LDS (65-float pitch)
template <int NW>
struct TileRegs { float4 a[4 / NW], b[4 / NW]; };  // NW = waves per workgroup (4 or 2): 64 * NW threads cover 8 * NW rows per pass
template <int NW>
__device__ __forceinline__ TileRegs<NW> fetch_tile(const float* __restrict__ src, int64_t ld, int row0, int len) {
    const int r = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
    TileRegs<NW> t;
#pragma unroll
    for (int q = 0; q < 4 / NW; ++q) {
        int row = row0 + r + 8 * NW * q;
        row = row < len ? row : len - 1;
        t.a[q] = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c);
        t.b[q] = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c + 4);
    }
    return t;
}
template <int NW>
__device__ __forceinline__ void put_tile(const TileRegs<NW>& t, float* __restrict__ dst) {
    const int r = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int q = 0; q < 4 / NW; ++q) {
        float* d = dst + (r + 8 * NW * q) * kPitch + c;
        *reinterpret_cast<float4*>(d) = t.a[q];
        *reinterpret_cast<float4*>(d + 4) = t.b[q];
    }
}

// the owned 32 rows as an MFMA B operand: lane (lr, lh) holds row lr, d = s + 32 lh for s = 0 .. 31
__device__ __forceinline__ void load_owned(const float* __restrict__ src, int64_t ld, int row, int lh, float (&reg)[32]) {
    const float* p = src + (int64_t)row * ld + 32 * lh;
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + s);
        reg[s] = v.x; reg[s + 1] = v.y; reg[s + 2] = v.z; reg[s + 3] = v.w;
    }
}

// acc = T O^T: T = the staged tile (rows along lanes), O = the owned rows (registers).  The lane's 32 k-values of row lr are 8
// ds_read_b128, issued 4 at a time ahead of the 16 MFMAs that consume them.
__device__ __forceinline__ f32x16 tile_times_owned(const float* __restrict__ tile, int lr, int lh, const float (&own)[32]) {
    f32x16 acc = zero16();
    const float4* a = reinterpret_cast<const float4*>(tile + lr * kPitch + 32 * lh);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a[4 * hf + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 16 * hf + 4 * j;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].x, own[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].y, own[s + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].z, own[s + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].w, own[s + 3], acc, 0, 0, 0);
        }
    }
    return acc;
}

// two products against the same lanes' rows at once (S and dP): independent accumulator chains interleaved, reads batched as above
__device__ __forceinline__ void two_tiles_times_owned(const float* __restrict__ t0, const float (&own0)[32], const float* __restrict__ t1,
                                                      const float (&own1)[32], int lr, int lh, f32x16& acc0, f32x16& acc1) {
    acc0 = zero16();
    acc1 = zero16();
    const float4* a0 = reinterpret_cast<const float4*>(t0 + lr * kPitch + 32 * lh);
    const float4* a1 = reinterpret_cast<const float4*>(t1 + lr * kPitch + 32 * lh);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        float4 v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = a0[4 * hf + j]; v1[j] = a1[4 * hf + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 16 * hf + 4 * j;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j].x, own0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j].x, own1[s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j].y, own0[s + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j].y, own1[s + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j].z, own0[s + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j].z, own1[s + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j].w, own0[s + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j].w, own1[s + 3], acc1, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- query-owned
template <bool STATS, int NW>
__global__ __launch_bounds__(64 * NW, (STATS ? 4 : 3)) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                                const float* __restrict__ dctx, const int32_t* __restrict__ cu,
                                                                const int32_t* __restrict__ order, int H, float scale, int window,
                                                                float* __restrict__ rowstat, float* __restrict__ dqkv,
                                                                uint32_t drop_thresh = 0u, float drop_inv = 1.f, uint32_t drop_key = 0u,
                                                                const int32_t* __restrict__ work = nullptr, int nseq = 0x7fffffff) {
    // drop_thresh != 0: the forward dropped attention probabilities (mask = mr::dropout_keep(key, query token * H + head, key position)):
    // dP = (dO . v) * mask / (1 - p); delta = dO . O already holds the dropped forward
    __shared__ float ks[kTile], vs[STATS ? 1 : kTile];
    constexpr int kRows = 32 * NW;  // rows owned by this workgroup
    // work != NULL: the 1-D work-list grid of mr_attn_split_work_f32 (csrc/attn_bf16.hip: only the (sequence, 128-row block) pairs that
    // exist, dealt over the XCD queues); else the (blocks, H, B) box with the longest sequences first when the caller passes the order
    int b, h, Q0;
    if (work) {
        const int slot = blockIdx.x >> 3, e = slot / H, ent = work[e * 8 + (blockIdx.x & 7)];
        if (ent < 0) return;
        h = slot - e * H; b = ent & 0xffffff; Q0 = (ent >> 24) * kRows;
        if (b >= nseq) return;
    } else {
        b = order ? order[blockIdx.z] : blockIdx.z; h = blockIdx.y; Q0 = blockIdx.x * kRows;
    }
    const int t0 = cu[b], len = cu[b + 1] - t0;
    if (Q0 >= len) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int64_t ld = (int64_t)3 * H * kDh, ldc = (int64_t)H * kDh;
    const float* Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* Kb = Qb + H * kDh;
    const float* Vb = Qb + 2 * H * kDh;
    const float* Gb = dctx + (int64_t)t0 * ldc + h * kDh;
    const int q0 = Q0 + wave * 32, q = q0 + lr;
    const bool q_on = q < len;
    const int qc = q_on ? q : len - 1;
    float qr[32], gr[STATS ? 1 : 32];
    load_owned(Qb, ld, qc, lh, qr);
    float lse = 0.f, dl = 0.f;
    if constexpr (!STATS) {
        load_owned(Gb, ldc, qc, lh, gr);
        lse = rowstat[((int64_t)(t0 + qc) * H + h) * 2];
        dl = rowstat[((int64_t)(t0 + qc) * H + h) * 2 + 1];
    }
    float m = -INFINITY, l = 0.f;
    f32x16 dq0 = zero16(), dq1 = zero16();
    // key tiles this workgroup needs, in order (uniform over the workgroup); the next tile's global loads overlap this tile's MFMAs
    auto next_tile = [&](int j) {
        while (j < len && !tile_needed(Q0, kRows, j, window)) j += 32;
        return j;
    };
    int j0 = next_tile(0);
    TileRegs<NW> kt = fetch_tile<NW>(Kb, ld, j0, len), vt = kt;
    if constexpr (!STATS) vt = fetch_tile<NW>(Vb, ld, j0, len);
    while (j0 < len) {
        __syncthreads();
        put_tile(kt, ks);
        if constexpr (!STATS) put_tile(vt, vs);
        __syncthreads();
        const int jn = next_tile(j0 + 32);
        if (jn < len) {
            kt = fetch_tile<NW>(Kb, ld, jn, len);
            if constexpr (!STATS) vt = fetch_tile<NW>(Vb, ld, jn, len);
        }
        const int jc = j0;
        j0 = jn;
        if (q0 >= len || !tile_needed(q0, 32, jc, window)) continue;  // uniform over the wave (no barrier below)
        f32x16 st, dpt;  // S^T[key crow(i)][query lr], dP^T[key][query]
        if constexpr (STATS) st = tile_times_owned(ks, lr, lh, qr);
        else two_tiles_times_owned(ks, qr, vs, gr, lr, lh, st, dpt);
        if constexpr (STATS) {
            float sv[16], tm = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = jc + crow(i, lh);
                const bool ok = key < len && allowed(q, key, window);
                sv[i] = ok ? st[i] * scale : -INFINITY;
                tm = fmaxf(tm, sv[i]);
            }
            if (tm > -INFINITY) {
                const float mn = fmaxf(m, tm);
                float add = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) add += __expf(sv[i] - mn);  // exp(-inf) = 0 for the masked ones
                l = l * __expf(m - mn) + add;
                m = mn;
            }
        } else {
            float ds[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = jc + crow(i, lh);
                const bool ok = key < len && allowed(q, key, window);
                const float p = ok ? __expf(st[i] * scale - lse) : 0.f;
                float dpv = dpt[i];
                if (drop_thresh) dpv = mr::dropout_keep(drop_key, (uint32_t)(t0 + q) * (uint32_t)H + (uint32_t)h, (uint32_t)key, drop_thresh) ? dpv * drop_inv : 0.f;
                ds[i] = p * (dpv - dl) * scale;
            }
            // dQ[query lr][d] += sum_key dS[query][key] K[key][d]: A = dS^T in its result layout, k = key crow(i, lh).  Registers
            // 4 b .. 4 b + 3 are four consecutive keys: their 8 K values are read together ahead of the 8 MFMAs
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                float k0v[4], k1v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* kr = ks + (8 * bq + 4 * lh + j) * kPitch + lr;
                    k0v[j] = kr[0];
                    k1v[j] = kr[32];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[4 * bq + j], k0v[j], dq0, 0, 0, 0);
                    dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[4 * bq + j], k1v[j], dq1, 0, 0, 0);
                }
            }
        }
    }
    if constexpr (STATS) {
        // the two halves of a query's lane pair saw disjoint keys: merge, then delta = dO . O
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float M = fmaxf(m, m2);
        float L = 0.f;
        if (M > -INFINITY) L = l * __expf(m - M) + l2 * __expf(m2 - M);
        const float* o = ctx + (int64_t)(t0 + qc) * ldc + h * kDh + 32 * lh;
        const float* g = Gb + (int64_t)qc * ldc + 32 * lh;
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < 32; s += 4) {
            const float4 a = *reinterpret_cast<const float4*>(o + s), c = *reinterpret_cast<const float4*>(g + s);
            dsum += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        }
        dsum += __shfl_xor(dsum, 32, 64);
        if (q_on && lh == 0) {
            rowstat[((int64_t)(t0 + q) * H + h) * 2] = (L > 0.f) ? M + __logf(L) : 0.f;  // (no allowed key: the windowed mode's row 0)
            rowstat[((int64_t)(t0 + q) * H + h) * 2 + 1] = dsum;
        }
    } else {
        // result layout: row = query q0 + crow(i, lh), col = d lr (+ 32)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qq = q0 + crow(i, lh);
            if (qq < len) {
                float* o = dqkv + (int64_t)(t0 + qq) * ld + h * kDh + lr;
                o[0] = dq0[i];
                o[32] = dq1[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ key-owned
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                                 const float* __restrict__ rowstat, const int32_t* __restrict__ cu,
                                                                 const int32_t* __restrict__ order, int H, float scale, int window,
                                                                 float* __restrict__ dqkv, uint32_t drop_thresh = 0u, float drop_inv = 1.f,
                                                                 uint32_t drop_key = 0u, const int32_t* __restrict__ work = nullptr, int nseq = 0x7fffffff) {
    __shared__ float qs[kTile], gs[kTile], stat[32][2];
    constexpr int kRows = 32 * NW;
    int b, h, K0;
    if (work) {  // as in attn_bwd_q_kernel: a key block of 128 rows per list entry
        const int slot = blockIdx.x >> 3, e = slot / H, ent = work[e * 8 + (blockIdx.x & 7)];
        if (ent < 0) return;
        h = slot - e * H; b = ent & 0xffffff; K0 = (ent >> 24) * kRows;
        if (b >= nseq) return;
    } else {
        b = order ? order[blockIdx.z] : blockIdx.z; h = blockIdx.y; K0 = blockIdx.x * kRows;
    }
    const int t0 = cu[b], len = cu[b + 1] - t0;
    if (K0 >= len) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int64_t ld = (int64_t)3 * H * kDh, ldc = (int64_t)H * kDh;
    const float* Qb = qkv + (int64_t)t0 * ld + h * kDh;
    const float* Kb = Qb + H * kDh;
    const float* Vb = Qb + 2 * H * kDh;
    const float* Gb = dctx + (int64_t)t0 * ldc + h * kDh;
    const int k0 = K0 + wave * 32, key = k0 + lr;
    const int kc = key < len ? key : len - 1;
    float kr[32], vr[32];
    load_owned(Kb, ld, kc, lh, kr);
    load_owned(Vb, ld, kc, lh, vr);
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    // query tiles this workgroup needs ("needed" is symmetric in the band part; the global key 0 column makes every query tile needed
    // for the first key tile); the next tile's global loads overlap this tile's MFMAs
    auto next_tile = [&](int i) {
        while (i < len && !(window < 0 || K0 == 0 || (i <= K0 + kRows - 1 + window && i + 31 >= K0 - window))) i += 32;
        return i;
    };
    auto fetch_stat = [&](int i0) {
        const int ir = (threadIdx.x & 63) >> 1, i = i0 + ir < len ? i0 + ir : len - 1;
        return rowstat[((int64_t)(t0 + i) * H + h) * 2 + (threadIdx.x & 1)];
    };
    int i0 = next_tile(0);
    TileRegs<NW> qt = fetch_tile<NW>(Qb, ld, i0, len), gt = fetch_tile<NW>(Gb, ldc, i0, len);
    float stv = fetch_stat(i0);
    while (i0 < len) {
        __syncthreads();
        put_tile(qt, qs);
        put_tile(gt, gs);
        if (threadIdx.x < 64) stat[threadIdx.x >> 1][threadIdx.x & 1] = stv;
        __syncthreads();
        const int in = next_tile(i0 + 32);
        if (in < len) {
            qt = fetch_tile<NW>(Qb, ld, in, len);
            gt = fetch_tile<NW>(Gb, ldc, in, len);
            stv = fetch_stat(in);
        }
        const int ic = i0;
        i0 = in;
        const bool w_need = window < 0 || k0 == 0 || (ic <= k0 + 31 + window && ic + 31 >= k0 - window);
        if (k0 >= len || !w_need) continue;
        f32x16 s, dp;  // S[query crow(i)][key lr], dP[query][key]
        two_tiles_times_owned(qs, kr, gs, vr, lr, lh, s, dp);
        float p[16], ds[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = crow(i, lh), qi = ic + r;
            const bool ok = qi < len && allowed(qi, key, window);
            p[i] = ok ? __expf(s[i] * scale - stat[r][0]) : 0.f;
            float dpv = dp[i];
            if (drop_thresh) {  // dV takes the dropped probabilities, dS the masked dP
                const bool keep = mr::dropout_keep(drop_key, (uint32_t)(t0 + qi) * (uint32_t)H + (uint32_t)h, (uint32_t)key, drop_thresh);
                dpv = keep ? dpv * drop_inv : 0.f;
                ds[i] = p[i] * (dpv - stat[r][1]) * scale;
                p[i] = keep ? p[i] * drop_inv : 0.f;
            } else {
                ds[i] = p[i] * (dpv - stat[r][1]) * scale;
            }
        }
        // dV[key lr][d] += sum_query P[query][key] dO[query][d];  dK[key][d] += sum_query dS[query][key] Q[query][d]
        // (registers 4 b .. 4 b + 3 are four consecutive queries: their 16 operand values are read together ahead of the 16 MFMAs)
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            float g0[4], g1[4], q0v[4], q1v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 8 * bq + 4 * lh + j;
                const float* gr = gs + r * kPitch + lr;
                const float* qr = qs + r * kPitch + lr;
                g0[j] = gr[0]; g1[j] = gr[32];
                q0v[j] = qr[0]; q1v[j] = qr[32];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p[4 * bq + j], g0[j], dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p[4 * bq + j], g1[j], dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[4 * bq + j], q0v[j], dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[4 * bq + j], q1v[j], dk1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kk = k0 + crow(i, lh);
        if (kk < len) {
            float* ok = dqkv + (int64_t)(t0 + kk) * ld + (H + h) * kDh + lr;
            float* ov = dqkv + (int64_t)(t0 + kk) * ld + (2 * H + h) * kDh + lr;
            ok[0] = dk0[i];
            ok[32] = dk1[i];
            ov[0] = dv0[i];
            ov[32] = dv1[i];
        }
    }
}

}  // namespace

static int attn_bwd_launch(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H,
                           int dh, int max_len, float scale, int window, float* rowstat, float* dqkv, uint32_t thresh, float inv, uint32_t key,
                           mr_stream_t stream);

// work-list launch of the three backward kernels (work / n_slots from mr_attn_work_plan(lens, B, 128, ...)): no empty workgroups on ragged
// batches; same results as mr_attn_bwd_train_f32 bit for bit (every output element has one owner and a fixed order in both)
extern "C" int mr_attn_bwd_work_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* work,
                                    int64_t n_slots, int B, int q_rows, int H, int dh, float scale, int window, float drop_p, uint32_t drop_key,
                                    float* rowstat, float* dqkv, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    if (!qkv || !ctx || !dctx || !cu_seqlens || !rowstat || !dqkv || n_slots < 0 || B < 0 || H < 1 || (n_slots > 0 && !work)) return MR_EINVAL;
    if (q_rows != 128) return MR_EINVAL;  // the three kernels own 128 rows per list entry
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx) || !mr::aligned16(dctx)) return MR_EALIGN;
    if (n_slots == 0) return MR_OK;
    if (n_slots * 8 * (int64_t)H > 0x7fffffff) return MR_EINVAL;
    const dim3 grid((unsigned)(n_slots * 8 * H));
    hipStream_t st = (hipStream_t)stream;
    if (!thresh) inv = 1.f;
    hipLaunchKernelGGL((attn_bwd_q_kernel<true, 4>), grid, dim3(256), 0, st, qkv, ctx, dctx, cu_seqlens, nullptr, H, scale, window, rowstat, dqkv, thresh,
                       inv, drop_key, work, B);
    hipLaunchKernelGGL((attn_bwd_q_kernel<false, 4>), grid, dim3(256), 0, st, qkv, ctx, dctx, cu_seqlens, nullptr, H, scale, window, rowstat, dqkv, thresh,
                       inv, drop_key, work, B);
    hipLaunchKernelGGL((attn_bwd_kv_kernel<4>), grid, dim3(256), 0, st, qkv, dctx, rowstat, cu_seqlens, nullptr, H, scale, window, dqkv, thresh, inv, drop_key,
                       work, B);
    return mr::check_launch();
}

extern "C" int mr_attn_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order, int B,
                               int H, int dh, int max_len, float scale, int window, float* rowstat, float* dqkv, mr_stream_t stream) {
    return attn_bwd_launch(qkv, ctx, dctx, cu_seqlens, seq_order, B, H, dh, max_len, scale, window, rowstat, dqkv, 0u, 1.f, 0u, stream);
}

// backward of mr_attn_train_f32 / mr_attn_split_work_train_f32: the same drop_p / drop_key as the forward (the mask is recomputed)
extern "C" int mr_attn_bwd_train_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order,
                                     int B, int H, int dh, int max_len, float scale, int window, float drop_p, uint32_t drop_key, float* rowstat,
                                     float* dqkv, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    return attn_bwd_launch(qkv, ctx, dctx, cu_seqlens, seq_order, B, H, dh, max_len, scale, window, rowstat, dqkv, thresh, thresh ? inv : 1.f, drop_key,
                           stream);
}

static int attn_bwd_launch(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H,
                           int dh, int max_len, float scale, int window, float* rowstat, float* dqkv, uint32_t thresh, float inv, uint32_t key,
                           mr_stream_t stream) {
    if (!qkv || !ctx || !dctx || !cu_seqlens || !rowstat || !dqkv || B < 0 || H < 1 || max_len < 0) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (!mr::aligned16(qkv) || !mr::aligned16(ctx) || !mr::aligned16(dctx)) return MR_EALIGN;
    if (B == 0 || max_len == 0) return MR_OK;
    if (H > 65535 || B > 65535) return MR_EUNSUPPORTED;
    // 128-row workgroups (four waves) by default; MR_ATTNBWD_ROWS=64 selects two-wave workgroups (finer granularity at sequence ends, twice
    // the staging per MFMA: measured equal or slightly slower on uniform and on Amazon-shaped ragged batches)
    static const int rows = [] { const char* e = getenv("MR_ATTNBWD_ROWS"); return e ? atoi(e) : 128; }();
    hipStream_t st = (hipStream_t)stream;
    if (rows == 128) {
        const dim3 grid((max_len + 127) / 128, H, B);
        hipLaunchKernelGGL((attn_bwd_q_kernel<true, 4>), grid, dim3(256), 0, st, qkv, ctx, dctx, cu_seqlens, seq_order, H, scale, window, rowstat, dqkv, thresh, inv, key);
        hipLaunchKernelGGL((attn_bwd_q_kernel<false, 4>), grid, dim3(256), 0, st, qkv, ctx, dctx, cu_seqlens, seq_order, H, scale, window, rowstat, dqkv, thresh, inv, key);
        hipLaunchKernelGGL((attn_bwd_kv_kernel<4>), grid, dim3(256), 0, st, qkv, dctx, rowstat, cu_seqlens, seq_order, H, scale, window, dqkv, thresh, inv, key);
    } else {
        const dim3 grid((max_len + 63) / 64, H, B);
        hipLaunchKernelGGL((attn_bwd_q_kernel<true, 2>), grid, dim3(128), 0, st, qkv, ctx, dctx, cu_seqlens, seq_order, H, scale, window, rowstat, dqkv, thresh, inv, key);
        hipLaunchKernelGGL((attn_bwd_q_kernel<false, 2>), grid, dim3(128), 0, st, qkv, ctx, dctx, cu_seqlens, seq_order, H, scale, window, rowstat, dqkv, thresh, inv, key);
        hipLaunchKernelGGL((attn_bwd_kv_kernel<2>), grid, dim3(128), 0, st, qkv, dctx, rowstat, cu_seqlens, seq_order, H, scale, window, dqkv, thresh, inv, key);
    }
    return mr::check_launch();
}
