// K5: full-catalog scoring epilogue -- exact row-wise top-k in canonical order, plus the per-row
// log-sum-exp / label logit / label rank the evaluator and the CE loss need.
//
// One 256-thread workgroup per score row.  Exact radix select on order-preserving 32-bit keys
// (4 passes x 8 bits, histograms in LDS) finds the k-th largest key; a collection pass takes
// everything above it plus the lowest-index ties, and one wavefront bitonic-sorts the <= 64 survivors
// by (score desc, index asc).  Rows of up to kMaxCachedCols scores are copied into LDS once (one 16-byte-per-lane sweep of the
// score block: the kernel's only global read) and every later pass -- four histogram passes, the collection, the log-sum-exp --
// runs out of LDS; longer rows are re-read from L2 on every pass.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <atomic>

// phase-timing hooks: empty in the library; exp/topk_phases.hip defines them (s_memtime deltas of thread 0)
#ifndef MR_TK_DECL
#define MR_TK_DECL
#define MR_TK(i)
#define MR_TK_FLUSH(row)
#endif

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned ord_key(float f) {
    unsigned u = __float_as_uint(f);
    if (f != f) return 0xffffffffu;       // NaN ranks above everything (torch.topk convention)
    if (u == 0x80000000u) u = 0u;         // -0.0 == +0.0: same key, tie broken by index
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int kMaxCachedCols = 38 * 1024;  // 152 KB of the CU's 160 KB LDS (the rest: histogram, candidates)

template <bool CACHED>
__global__ __launch_bounds__(kThreads) void topk_rows_kernel(const float* __restrict__ scores, int64_t ld, int ncols,
                                                            int k, float* __restrict__ top_val,
                                                            int64_t* __restrict__ top_idx,
                                                            const int64_t* __restrict__ labels, float inv_temp,
                                                            float* __restrict__ row_lse, float* __restrict__ row_lab,
                                                            int32_t* __restrict__ label_rank) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining, s_ngt;
    __shared__ unsigned wave_cnt[kThreads / MR_WAVE];
    __shared__ unsigned long long cand[64];
    __shared__ float red[kThreads / MR_WAVE];

    extern __shared__ __attribute__((aligned(16))) float row_cache[];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ g = scores + (int64_t)row * ld;
    if (CACHED) {  // ld % 4 == 0 and a 16-byte aligned block are checked by the host for this variant
        const int nv = ncols >> 2;
        for (int v = tid; v < nv; v += kThreads) reinterpret_cast<float4*>(row_cache)[v] = reinterpret_cast<const float4*>(g)[v];
        for (int i = (nv << 2) + tid; i < ncols; i += kThreads) row_cache[i] = g[i];
        __syncthreads();
    }
    // `s[i]` below: LDS when cached (the address-space-3 array), global otherwise
#define s (CACHED ? row_cache : g)

    if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)k; s_ngt = 0u; }
    if (tid < 64) cand[tid] = 0ull;
    unsigned mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        // run-length pre-aggregation per thread: cosine scores share sign, exponent and the top mantissa bits, so in the first passes
        // (nearly) every element falls into ONE bucket -- one LDS atomic per run instead of one per element keeps that from serialising
        unsigned run_b = 0xffffffffu, run_n = 0u;
        for (int i = tid; i < ncols; i += kThreads) {
            const unsigned key = ord_key(s[i]);
            if ((key & mask) == prefix) {
                const unsigned bk = (key >> shift) & 0xffu;
                if (bk == run_b) {
                    ++run_n;
                } else {
                    if (run_n) atomicAdd(&hist[run_b], run_n);
                    run_b = bk;
                    run_n = 1u;
                }
            }
        }
        if (run_n) atomicAdd(&hist[run_b], run_n);
        __syncthreads();
        if (tid == 0) {
            unsigned rem = s_remaining, c = 0u;
            int bkt = 255;
            for (; bkt > 0; --bkt) {
                if (c + hist[bkt] >= rem) break;
                c += hist[bkt];
            }
            s_remaining = rem - c;
            s_prefix = prefix | ((unsigned)bkt << shift);
        }
        mask |= 0xffu << shift;
        __syncthreads();
    }
    const unsigned thr = s_prefix;          // key of the k-th largest element
    const unsigned need_eq = s_remaining;   // how many elements equal to it belong to the top-k
    const unsigned n_gt = (unsigned)k - need_eq;

    // collection: keys > thr in any order, keys == thr by ascending index (first need_eq of them)
    unsigned eq_seen = 0u;  // block-uniform running count of == thr elements in earlier chunks
    for (int base = 0; base < ncols; base += kThreads) {
        const int i = base + tid;
        unsigned key = 0u;
        bool gt = false, eq = false;
        if (i < ncols) {
            key = ord_key(s[i]);
            gt = key > thr;
            eq = key == thr;
        }
        if (gt) {
            const unsigned slot = atomicAdd(&s_ngt, 1u);
            if (slot < 64u) cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
        }
        const unsigned long long beq = __ballot(eq);
        if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(beq);
        __syncthreads();
        unsigned before = eq_seen, total = 0u;
#pragma unroll
        for (int w = 0; w < kThreads / MR_WAVE; ++w) {
            if (w < wave) before += wave_cnt[w];
            total += wave_cnt[w];
        }
        if (eq) {
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const unsigned rank = before + (unsigned)__popcll(beq & lt);
            if (rank < need_eq)
                cand[n_gt + rank] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
        }
        eq_seen += total;
        __syncthreads();
        if (eq_seen >= need_eq && base + kThreads < ncols) {
            // remaining chunks can only contribute keys > thr
            for (int j = base + kThreads + tid; j < ncols; j += kThreads) {
                const unsigned kj = ord_key(s[j]);
                if (kj > thr) {
                    const unsigned slot = atomicAdd(&s_ngt, 1u);
                    if (slot < 64u) cand[slot] = ((unsigned long long)kj << 32) | (unsigned long long)(0xffffffffu - (unsigned)j);
                }
            }
            break;
        }
    }
    __syncthreads();

    // bitonic sort (descending) of the 64 candidate slots in wave 0; unused slots are 0 and sink
    if (wave == 0) {
        unsigned long long v = cand[lane];
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const unsigned long long o = __shfl_xor(v, stride, 64);
                const bool up = ((lane & size) == 0);          // descending block
                const bool lower = ((lane & stride) == 0);
                const bool take_max = (up == lower);
                v = take_max ? (v > o ? v : o) : (v < o ? v : o);
            }
        }
        if (lane < k) {
            unsigned idx = 0xffffffffu - (unsigned)(v & 0xffffffffull);
            if (idx >= (unsigned)ncols) idx = 0u;  // cannot happen for ncols >= k; never read out of bounds
            top_idx[(int64_t)row * k + lane] = (int64_t)idx;
            top_val[(int64_t)row * k + lane] = s[idx];
        }
        if (labels) {
            const int64_t lab = labels[row];
            const unsigned idx = 0xffffffffu - (unsigned)(v & 0xffffffffull);
            const unsigned long long hit = __ballot(lane < k && (int64_t)idx == lab);
            if (lane == 0 && label_rank) label_rank[row] = hit ? (int32_t)__builtin_ctzll(hit) : -1;
        }
        cand[lane] = v;
    }
    __syncthreads();

    if (labels && row_lse) {
        // row max = best candidate's score (NaN rows propagate NaN like torch.cross_entropy)
        unsigned best = 0xffffffffu - (unsigned)(cand[0] & 0xffffffffull);
        if (best >= (unsigned)ncols) best = 0u;
        const float mx = s[best] * inv_temp;
        float acc = 0.f;
        for (int i = tid; i < ncols; i += kThreads) acc += expf(s[i] * inv_temp - mx);
        acc = mr::wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            const float tot = (red[0] + red[1]) + (red[2] + red[3]);
            row_lse[row] = mx + logf(tot);
            if (row_lab) {
                const int64_t lab = labels[row];
                row_lab[row] = (lab >= 0 && lab < ncols) ? s[lab] * inv_temp : NAN;
            }
        }
    }
#undef s
}


// ------------------------------------------------------------------------------------------------------------------------------
// Register-resident form (r03): one 1024-thread workgroup per row, the row's order-preserving keys live in registers (NV float4 loads
// per thread, all in flight at once: ONE memory latency instead of a dependent load per 256 columns), so the four radix passes, the
// collection and the log-sum-exp never touch memory again; the bucket scan after each pass is one wavefront's suffix sum instead of
// a serial walk over 256 LDS words.  Same canonical result as topk_rows_kernel (score desc, index asc; NaN first; -0 == +0), also
// for k up to 1024 (the candidates are then sorted by the whole workgroup).  Covers ncols <= 4096 * NV (NV <= 12: 49,152 columns; longer rows take topk_rows_kernel).
constexpr int kRegThreads = 1024;
constexpr int kFastBins = 2048, kFastCand = 2048;
constexpr int kRegWaves = kRegThreads / MR_WAVE;

__device__ __forceinline__ float key_value(unsigned key) {  // inverse of ord_key up to the sign of zero and the NaN payload
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

template <int NV, int KCAP>
__global__ __launch_bounds__(kRegThreads) void topk_rows_reg_kernel(const float* __restrict__ scores, int64_t ld, int ncols, int k,
                                                                   float* __restrict__ top_val, int64_t* __restrict__ top_idx,
                                                                   const int64_t* __restrict__ labels, float inv_temp,
                                                                   float* __restrict__ row_lse, float* __restrict__ row_lab,
                                                                   int32_t* __restrict__ label_rank) {
    __shared__ unsigned hist4[4][256];  // one histogram per radix pass, all zeroed once (no zero-and-barrier inside the pass loop)
    __shared__ unsigned s_prefix, s_remaining, s_ngt, s_neq, s_eqtotal, s_rank;
    __shared__ unsigned wcnt[kRegWaves];
    __shared__ float red[kRegWaves];
    __shared__ unsigned long long cand[KCAP];
    // fast path (below): one 2048-bin histogram over the row's own key range, candidate lists
    __shared__ unsigned h2[kFastBins];
    __shared__ unsigned long long c2[kFastCand], c3[kFastCand];
    __shared__ unsigned wlo[kRegWaves], whi[kRegWaves], wsum[kRegWaves];
    __shared__ unsigned s_nc, s_bstar, s_nctot;

    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ g = scores + (int64_t)row * ld;
    // element (j, c) of this thread is column (j * 1024 + tid) * 4 + c; columns past ncols carry key 0 (below every real key)
    MR_TK_DECL
    unsigned key[NV * 4];
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15u) == 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) {  // the loads first (bit patterns straight into the key registers), all in flight together
        const int c0 = (j * kRegThreads + tid) * 4;
        if (vec && c0 + 3 < ncols) {
            const uint4 v = *reinterpret_cast<const uint4*>(g + c0);
            key[4 * j] = v.x; key[4 * j + 1] = v.y; key[4 * j + 2] = v.z; key[4 * j + 3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) key[4 * j + c] = (c0 + c < ncols) ? __float_as_uint(g[c0 + c]) : 0u;
        }
    }
#pragma unroll
    for (int e = 0; e < NV * 4; ++e) {
        const int i = ((e >> 2) * kRegThreads + tid) * 4 + (e & 3);
        key[e] = i < ncols ? ord_key(__uint_as_float(key[e])) : 0u;
    }

    MR_TK(0)
    if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)k; s_ngt = 0u; s_neq = 0u; s_eqtotal = 0u; s_rank = 0xffffffffu; s_nc = 0u; s_nctot = 0xffffffffu; }
    for (int i = tid; i < KCAP; i += kRegThreads) cand[i] = 0ull;
    (&hist4[0][0])[tid] = 0u;  // 4 x 256 words, one per thread
    h2[tid] = 0u;
    h2[tid + kRegThreads] = 0u;
    // ---- fast path: ONE histogram pass.  The keys of a row of cosine scores share their sign, exponent and leading mantissa bits, so a
    // fixed-digit radix pass spends its first two sweeps on bits that do not separate anything.  Instead the 2048 bins are laid over the
    // row's own key range [kmin, kmax] (bin = (key - kmin) >> sh): the bin that holds the k-th largest key and every bin above it then
    // contain k plus a few keys -- those go to a candidate list that is rank-sorted directly ((key, index) pairs are unique; ties at the
    // threshold share a bin, so the canonical order falls out).  Rows whose range is blown up by an outlier (NaN, inf) or that hold
    // thousands of equal keys put more than kFastCand keys into the list: they take the exact four-pass radix select below.
    constexpr bool kTryFast = NV >= 4;  // short rows (<= 8,192 columns): the fixed cost of this path exceeds what the radix passes cost there
    bool fast = false;
    if (kTryFast) {
        unsigned klo = 0xffffffffu, khi = 0u;
#pragma unroll
        for (int e = 0; e < NV * 4; ++e)
            if (key[e]) { klo = key[e] < klo ? key[e] : klo; khi = key[e] > khi ? key[e] : khi; }  // key 0 = padding (no real key is 0)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned a = __shfl_xor(klo, o, 64), c = __shfl_xor(khi, o, 64);
            klo = a < klo ? a : klo;
            khi = c > khi ? c : khi;
        }
        if (lane == 0) { wlo[wave] = klo; whi[wave] = khi; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kRegWaves; ++w) { klo = wlo[w] < klo ? wlo[w] : klo; khi = whi[w] > khi ? whi[w] : khi; }
        const unsigned range = khi - klo;
        const int sh = range ? ((32 - __clz(range) - 11) > 0 ? (32 - __clz(range) - 11) : 0) : 0;  // (range >> sh) < 2048
#pragma unroll
        for (int e = 0; e < NV * 4; ++e)
            if (key[e]) atomicAdd(&h2[(key[e] - klo) >> sh], 1u);
        __syncthreads();
        {
            // thread t owns bins 2 t, 2 t + 1; `above` = keys in higher bins (suffix sum over the lanes above, then the waves above)
            const unsigned a0 = h2[2 * tid], a1 = h2[2 * tid + 1], loc = a0 + a1;
            unsigned incl = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += t;
            }
            if (lane == 0) wsum[wave] = incl;
            __syncthreads();
            unsigned above = incl - loc;
#pragma unroll
            for (int w = 0; w < kRegWaves; ++w)
                if (w > wave) above += wsum[w];
            if (above < (unsigned)k && (unsigned)k <= above + loc) {  // exactly one thread: its bins hold the k-th largest key
                if (above + a1 >= (unsigned)k) { s_bstar = 2u * tid + 1u; s_nctot = above + a1; }
                else { s_bstar = 2u * tid; s_nctot = above + loc; }
            }
        }
        __syncthreads();
        fast = s_nctot <= (unsigned)kFastCand;  // block-uniform
        if (fast) {
            const unsigned bstar = s_bstar, nc = s_nctot;
#pragma unroll
            for (int e = 0; e < NV * 4; ++e) {
                if (key[e] && ((key[e] - klo) >> sh) >= bstar) {
                    const unsigned slot = atomicAdd(&s_nc, 1u);
                    const unsigned i = (unsigned)(((e >> 2) * kRegThreads + tid) * 4 + (e & 3));
                    c2[slot] = ((unsigned long long)key[e] << 32) | (unsigned long long)(0xffffffffu - i);
                }
            }
            __syncthreads();
            for (unsigned t = tid; t < nc; t += kRegThreads) {  // rank sort, descending by (key, -index): the pairs are unique
                const unsigned long long v = c2[t];
                unsigned r = 0u;
                for (unsigned j = 0; j < nc; ++j) r += c2[j] > v ? 1u : 0u;
                c3[r] = v;
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
    }
    MR_TK(1)
    const unsigned long long* __restrict__ sorted = fast ? c3 : cand;
    const int64_t lab = labels ? labels[row] : -1;
    if (!fast) {
    unsigned mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        unsigned* hist = hist4[pass];
        const unsigned prefix = s_prefix, rem = s_remaining;
        // run-length pre-aggregation per thread (cosine scores share sign, exponent and the top mantissa bits: one atomic per run)
        unsigned run_b = 0xffffffffu, run_n = 0u;
#pragma unroll
        for (int e = 0; e < NV * 4; ++e) {
            if ((key[e] & mask) == prefix) {
                const unsigned bk = (key[e] >> shift) & 0xffu;
                if (bk == run_b) {
                    ++run_n;
                } else {
                    if (run_n) atomicAdd(&hist[run_b], run_n);
                    run_b = bk;
                    run_n = 1u;
                }
            }
        }
        if (run_n) atomicAdd(&hist[run_b], run_n);
        __syncthreads();
        if (wave == 0) {  // lane l owns buckets 4 l .. 4 l + 3; `above` = elements in the buckets of higher lanes
            const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const unsigned c4 = (h0 + h1) + (h2 + h3);
            unsigned incl = c4;  // inclusive suffix sum over lanes >= lane
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += t;
            }
            const unsigned above = incl - c4;
            // the bucket of the rem-th largest: the one lane with above < rem <= above + c4, then its highest bucket that reaches rem
            const bool mine = above < rem && rem <= incl;
            if (mine) {
                unsigned c = above;
                int bkt;
                if (c + h3 >= rem) { bkt = 3; }
                else { c += h3; if (c + h2 >= rem) { bkt = 2; } else { c += h2; if (c + h1 >= rem) { bkt = 1; } else { c += h1; bkt = 0; } } }
                const unsigned hsel = bkt == 3 ? h3 : bkt == 2 ? h2 : bkt == 1 ? h1 : h0;
                s_remaining = rem - c;
                s_prefix = prefix | ((unsigned)(4 * lane + bkt) << shift);
                s_eqtotal = hsel;  // after the last pass: how many keys equal the k-th largest
            }
        }
        mask |= 0xffu << shift;
        __syncthreads();
    }
    const unsigned thr = s_prefix;          // key of the k-th largest element
    const unsigned need_eq = s_remaining;   // how many elements equal to it belong to the top-k
    const unsigned n_gt = (unsigned)k - need_eq;
    const bool ties_cut = s_eqtotal > need_eq;  // block-uniform: more keys equal the threshold than fit -> lowest indices win

    // collection: keys > thr in any order (the sort below orders them)
#pragma unroll
    for (int e = 0; e < NV * 4; ++e) {
        if (key[e] > thr) {
            const unsigned slot = atomicAdd(&s_ngt, 1u);
            const unsigned i = (unsigned)(((e >> 2) * kRegThreads + tid) * 4 + (e & 3));
            if (slot < (unsigned)KCAP) cand[slot] = ((unsigned long long)key[e] << 32) | (unsigned long long)(0xffffffffu - i);
        }
    }
    if (!ties_cut) {  // every key == thr belongs to the top-k
#pragma unroll
        for (int e = 0; e < NV * 4; ++e) {
            if (key[e] == thr) {
                const unsigned slot = n_gt + atomicAdd(&s_neq, 1u);
                const unsigned i = (unsigned)(((e >> 2) * kRegThreads + tid) * 4 + (e & 3));
                if (slot < (unsigned)KCAP) cand[slot] = ((unsigned long long)thr << 32) | (unsigned long long)(0xffffffffu - i);
            }
        }
    } else {  // keys == thr by ascending column: columns ascend with (j, tid, c)
        unsigned eq_seen = 0u;
#pragma unroll 1
        for (int j = 0; j < NV; ++j) {
            unsigned cnt = 0u;
#pragma unroll
            for (int e = 0; e < NV * 4; ++e)
                if ((e >> 2) == j && key[e] == thr) ++cnt;
            unsigned incl = cnt;  // inclusive prefix over the wave's lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wcnt[wave] = incl;
            __syncthreads();
            unsigned before = eq_seen + (incl - cnt), total = 0u;
#pragma unroll
            for (int w = 0; w < kRegWaves; ++w) {
                if (w < wave) before += wcnt[w];
                total += wcnt[w];
            }
            unsigned r = before;
#pragma unroll
            for (int e = 0; e < NV * 4; ++e) {
                if ((e >> 2) == j && key[e] == thr) {
                    const unsigned i = (unsigned)((j * kRegThreads + tid) * 4 + (e & 3));
                    if (r < need_eq) cand[n_gt + r] = ((unsigned long long)thr << 32) | (unsigned long long)(0xffffffffu - i);
                    ++r;
                }
            }
            eq_seen += total;
            __syncthreads();
            if (eq_seen >= need_eq) break;  // block-uniform
        }
    }
    __syncthreads();

    MR_TK(2)
    // sort the candidates descending by (key, -index); unused slots are 0 and sink
    if (KCAP == 64) {
        if (wave == 0) {
            unsigned long long v = cand[lane];
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const unsigned long long o = __shfl_xor(v, stride, 64);
                    const bool up = ((lane & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    const bool take_max = (up == lower);
                    v = take_max ? (v > o ? v : o) : (v < o ? v : o);
                }
            }
            cand[lane] = v;
        }
    } else {
        for (int size = 2; size <= KCAP; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < KCAP / 2; t += kRegThreads) {
                    const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo | stride;
                    const unsigned long long a = cand[lo], c = cand[hi];
                    const bool desc = ((lo & size) == 0);
                    if (desc ? (a < c) : (a > c)) { cand[lo] = c; cand[hi] = a; }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    }  // !fast
    for (int t = tid; t < k; t += kRegThreads) {
        unsigned idx = 0xffffffffu - (unsigned)(sorted[t] & 0xffffffffull);
        if (idx >= (unsigned)ncols) idx = 0u;  // cannot happen for ncols >= k; never read out of bounds
        top_idx[(int64_t)row * k + t] = (int64_t)idx;
        top_val[(int64_t)row * k + t] = g[idx];
        if (labels && (int64_t)idx == lab) atomicMin(&s_rank, (unsigned)t);
    }
    if (labels) {
        __syncthreads();
        if (tid == 0 && label_rank) label_rank[row] = s_rank == 0xffffffffu ? -1 : (int32_t)s_rank;
    }

    MR_TK(3)
    if (labels && row_lse) {
        // row max = best candidate's score (NaN rows propagate NaN like torch.cross_entropy)
        unsigned best = 0xffffffffu - (unsigned)(sorted[0] & 0xffffffffull);
        if (best >= (unsigned)ncols) best = 0u;
        const float mx = g[best] * inv_temp;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < NV * 4; ++e)
            if (key[e] != 0u) acc += __expf(key_value(key[e]) * inv_temp - mx);  // arguments <= 0: v_exp_f32 on x * log2(e) is within 2 ulp
        acc = mr::wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < kRegWaves; ++w) tot += red[w];
            row_lse[row] = mx + logf(tot);
            if (row_lab) row_lab[row] = (lab >= 0 && lab < ncols) ? g[lab] * inv_temp : NAN;
        }
    }
    MR_TK(4)
    MR_TK_FLUSH(row)
}

}  // namespace

constexpr int kMaxTopK = 1024;

extern "C" int mr_topk_max_k(void) { return kMaxTopK; }

template <int NV>
static void launch_topk_reg(int k, const float* scores, int64_t ld, int nrows, int ncols, float* top_val, int64_t* top_idx, const int64_t* labels,
                            float inv_temp, float* row_lse, float* row_lab, int32_t* label_rank, hipStream_t st) {
    if (k <= 64)
        hipLaunchKernelGGL((topk_rows_reg_kernel<NV, 64>), dim3(nrows), dim3(kRegThreads), 0, st, scores, ld, ncols, k, top_val, top_idx, labels,
                           inv_temp, row_lse, row_lab, label_rank);
    else if (k <= 256)
        hipLaunchKernelGGL((topk_rows_reg_kernel<NV, 256>), dim3(nrows), dim3(kRegThreads), 0, st, scores, ld, ncols, k, top_val, top_idx, labels,
                           inv_temp, row_lse, row_lab, label_rank);
    else
        hipLaunchKernelGGL((topk_rows_reg_kernel<NV, kMaxTopK>), dim3(nrows), dim3(kRegThreads), 0, st, scores, ld, ncols, k, top_val, top_idx,
                           labels, inv_temp, row_lse, row_lab, label_rank);
}

extern "C" int mr_topk_rows_f32(const float* scores, int64_t ld, int nrows, int ncols, int k, float* top_val,
                                int64_t* top_idx, const int64_t* labels, float inv_temp, float* row_lse, float* row_lab,
                                int32_t* label_rank, mr_stream_t stream) {
    if (!scores || !top_val || !top_idx || nrows < 0 || ncols < 1 || k < 1) return MR_EINVAL;
    if (k > ncols) return MR_EUNSUPPORTED;
    if (ld < ncols) return MR_EINVAL;
    if (nrows == 0) return MR_OK;
    hipStream_t st = (hipStream_t)stream;
    static const bool force_lds = [] { const char* e = getenv("MR_TOPK_LDS"); return e && e[0] == '1'; }();  // A/B: the r02 kernel
    if (ncols <= 4096 * 12 && k <= kMaxTopK && !force_lds) {
        // register-resident rows: NV float4 per thread of a 1024-thread workgroup
#define MR_TOPK_REG(NV_) launch_topk_reg<NV_>(k, scores, ld, nrows, ncols, top_val, top_idx, labels, inv_temp, row_lse, row_lab, label_rank, st)
        if (ncols <= 4096 * 2) MR_TOPK_REG(2);
        else if (ncols <= 4096 * 4) MR_TOPK_REG(4);
        else if (ncols <= 4096 * 6) MR_TOPK_REG(6);
        else if (ncols <= 4096 * 8) MR_TOPK_REG(8);
        else MR_TOPK_REG(12);
#undef MR_TOPK_REG
        return mr::check_launch();
    }
    if (k > 64) return MR_EUNSUPPORTED;  // rows beyond 49,152 columns: the LDS / L2 kernel below (k <= 64)
    const bool cached = ncols <= kMaxCachedCols && (ld & 3) == 0 && mr::aligned16(scores);
    if (cached) {
        // per-device, checked: a failed attribute call would otherwise surface as an opaque launch error on that device only
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MR_ELAUNCH;
        static std::atomic<int> attr_state[64];  // 0 = not tried, 1 = ok, 2 = failed
        int stt = attr_state[dev].load(std::memory_order_acquire);
        if (stt == 0) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_rows_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxCachedCols * (int)sizeof(float));
            stt = e == hipSuccess ? 1 : 2;
            if (e != hipSuccess) (void)hipGetLastError();
            attr_state[dev].store(stt, std::memory_order_release);
        }
        if (stt == 1) {
            hipLaunchKernelGGL(topk_rows_kernel<true>, dim3(nrows), dim3(kThreads), (size_t)((ncols + 3) & ~3) * sizeof(float), st,
                               scores, ld, ncols, k, top_val, top_idx, labels, inv_temp, row_lse, row_lab, label_rank);
            return mr::check_launch();
        }
    }
    hipLaunchKernelGGL(topk_rows_kernel<false>, dim3(nrows), dim3(kThreads), 0, st, scores, ld, ncols, k, top_val,
                       top_idx, labels, inv_temp, row_lse, row_lab, label_rank);
    return mr::check_launch();
}

namespace mr {  // score_fused.hip
bool score_fused_supported(int64_t nU, int64_t M, int d, int k);
size_t score_fused_ws_bytes(int64_t nU, int64_t M, int k);
int score_fused_launch(const float* U, const float* E, int64_t nU, int64_t M, int d, int k, float* top_val, int64_t* top_idx, const int64_t* labels,
                       float inv_temp, float* row_lse, float* row_lab, int32_t* label_rank, void* ws, hipStream_t st);
}  // namespace mr

// Route of mr_score_topk_f32(scores_out = NULL).  2 = auto (default): the fused kernels when the (users x M) block would exceed
// kFusedAutoBytes -- it could not stay in the 256 MB Infinity Cache between the scoring GEMM's writes and the select's reads, i.e. it would
// cost HBM traffic; below that the two-kernel route is faster (its block never leaves the caches: 0.23 vs 0.31 ms at 256 x 22,855).
// 0 = never, 1 = always.  Initial value from MR_SCORE_FUSED; mr_score_fused_mode() changes it at run time (tests, A/B runs).
constexpr int64_t kFusedAutoBytes = (int64_t)128 << 20;
static int g_fused_mode = [] { const char* e = getenv("MR_SCORE_FUSED"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }();

extern "C" int mr_score_fused_mode(int mode) {
    const int old = g_fused_mode;
    if (mode >= 0 && mode <= 2) g_fused_mode = mode;
    return old;
}

static bool score_use_fused(int64_t nU, int64_t M, int d, int k) {
    if (g_fused_mode == 0 || nU <= 0 || M <= 0 || !mr::score_fused_supported(nU, M, d, k)) return false;
    return g_fused_mode == 1 || nU * M * (int64_t)sizeof(float) > kFusedAutoBytes;
}

static size_t score_unfused_ws(int64_t nU, int64_t M) {
    const int64_t ldm = (M + 3) & ~(int64_t)3;
    return (size_t)nU * (size_t)ldm * sizeof(float) + 256;
}

// workspace of mr_score_topk_f32(scores_out = NULL) for exactly this call shape: the candidate lists of the fused path when it applies
// (mr_score_fused_mode; k <= 64, d % 32 == 0, at most 512 parts of 768 items), else the (nU, M) block of the two-kernel path
extern "C" size_t mr_score_topk_ws_bytes_ex(int64_t nU, int64_t M, int d, int k) {
    if (nU < 0 || M < 0) return 0;
    if (!score_use_fused(nU, M, d, k)) return score_unfused_ws(nU, M);
    return mr::score_fused_ws_bytes(nU, M, k);
}

// shape-agnostic upper bound (either path, any d, k <= 64)
extern "C" size_t mr_score_topk_ws_bytes(int64_t nU, int64_t M) {
    if (nU < 0 || M < 0) return 0;
    const size_t a = score_unfused_ws(nU, M);
    const size_t b = (nU > 0 && M > 0 && mr::score_fused_supported(nU, M, 32, 1)) ? mr::score_fused_ws_bytes(nU, M, 64) : 0;
    return a > b ? a : b;
}

extern "C" int mr_score_topk_f32(const float* U, const float* E, int64_t nU, int64_t M, int d, int k, float* top_val,
                                 int64_t* top_idx, float* scores_out, const int64_t* labels, float inv_temp,
                                 float* row_lse, float* row_lab, int32_t* label_rank, void* ws, size_t ws_bytes,
                                 mr_stream_t stream) {
    if (!U || !E || nU < 0 || M < 1 || d < 1) return MR_EINVAL;
    if (nU > 0x7fffffff || M > 0x7fffffff) return MR_EUNSUPPORTED;
    if (!scores_out && score_use_fused(nU, M, d, k)) {
        // nobody asked for the (users x M) block: selection inside the scoring kernel (score_fused.hip), candidates only in the workspace
        if (!top_val || !top_idx) return MR_EINVAL;
        if (!ws) return MR_EINVAL;
        if (ws_bytes < mr::score_fused_ws_bytes(nU, M, k)) return MR_EWS;
        if ((d & 3) || !mr::aligned16(U) || !mr::aligned16(E)) return MR_EALIGN;
        return mr::score_fused_launch(U, E, nU, M, d, k, top_val, top_idx, labels, inv_temp, row_lse, row_lab, label_rank, ws, (hipStream_t)stream);
    }
    float* sc = scores_out;
    int64_t ldm = M;
    if (!sc) {
        if (!ws) return MR_EINVAL;
        if (ws_bytes < score_unfused_ws(nU, M)) return MR_EWS;
        sc = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
        ldm = (M + 3) & ~(int64_t)3;
    }
    if (nU == 0) return MR_OK;
    int rc = mr_gemm_nt_bias_act_f32(U, d, E, nullptr, nullptr, nullptr, nullptr, nullptr, 1, (int)nU, (int)M, d,
                                     MR_ACT_NONE, nullptr, 0, sc, ldm, stream);
    if (rc != MR_OK) return rc;
    return mr_topk_rows_f32(sc, ldm, (int)nU, (int)M, k, top_val, top_idx, labels, inv_temp, row_lse, row_lab, label_rank,
                            stream);
}
