// K5 fused: full-catalog scoring with the top-k selection inside the scoring kernel -- the (users x M) score block is never written.
//
//   score_part_topk_kernel   one workgroup = 32 users x one PART of the catalog (768 consecutive items).  The part's scores are produced
//                            256 columns at a time (8 waves x 32) by the exact-fp32 MFMA product of gemm.hip (v_mfma_f32_32x32x2_f32: every score is the
//                            ascending-k fp32 FMA chain, bit for bit what mr_gemm_nt_bias_act_f32 writes) into a 32 x 768 strip in LDS;
//                            then each wave radix-selects its rows' top-k out of LDS (as topk_rows_kernel: 4 passes x 8 bits on
//                            order-preserving keys, ties at the threshold by ascending index, one bitonic sort of the <= 64 survivors)
//                            and emits k candidates (key, item), their raw scores, the part's (max, sum exp) and the label's logit.
//   score_merge_kernel       one wave per user: the same wave select over the parts' candidates (array order = item order among equal
//                            scores), the log-sum-exp of the parts' partial sums, the label's position in the list.
// Results equal mr_gemm_nt_bias_act_f32 + mr_topk_rows_f32 exactly (indices, values, label rank, label logit; the log-sum-exp to rounding:
// its terms are added in a different order).  Traffic: E is read once per 32-user strip out of L2 (8 x 70 MB at 256 users x 22,855 items
// against 23 MB written + 23 MB re-read before); candidates: users x parts x k x 12 B.
#include "common.h"
#include <math.h>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 512;            // part kernel: 8 waves = two per SIMD (one workgroup per CU: 153 KB of LDS), each 32 columns of a chunk
constexpr int kWaves = kThreads / 64;
constexpr int SBM = 32, SBN = 256, SBK = 16;   // d % (2 SBK) == 0: the k loop is unrolled by two
constexpr int SSTR = 20;                 // floats per staged row: 8 even-k | 8 odd-k | 4 pad (gemm.hip)
constexpr int PART = 768;                // catalog columns per workgroup (strip row length)
constexpr int STAGE_F = (SBM + SBN) * SSTR;
constexpr int kMaxParts = 512;           // merge kernel: the candidates of one user (parts x k x 4 B) must fit LDS; larger catalogs take the two-kernel path

__device__ __forceinline__ unsigned ord_key(float f) {
    unsigned u = __float_as_uint(f);
    if (f != f) return 0xffffffffu;       // NaN ranks above everything (torch.topk convention)
    if (u == 0x80000000u) u = 0u;         // -0.0 == +0.0: same key, tie broken by index
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xffffffffull), m, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// One wave: the k best of s[0 .. n) in canonical order (score descending, index ascending) as composites (key << 32 | ~global column):
// lane i returns the i-th best, 0 beyond min(k, n).  hist: 256 words and slots: 64 composites of this wave's LDS scratch.
__device__ __forceinline__ unsigned long long wave_topk(const float* __restrict__ s, int n, int k, int col0, volatile unsigned* hist,
                                                        volatile unsigned long long* slots, int lane) {
    const int keff = k < n ? k : n;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned prefix = 0u, mask = 0u, remaining = (unsigned)keff;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int j = 0; j < 4; ++j) hist[4 * lane + j] = 0u;
        __builtin_amdgcn_wave_barrier();
        // wave-aggregated adds: cosine scores share sign, exponent and the top mantissa bits, so in the first passes (nearly) every lane
        // hits ONE bucket -- the first lane's bucket gets one add for all its lanes instead of a 64-way serialised one
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            unsigned bk = 0xffffffffu;
            if (i < n) {
                const unsigned key = ord_key(s[i]);
                if ((key & mask) == prefix) bk = (key >> shift) & 0xffu;
            }
            const unsigned long long todo = __ballot(bk != 0xffffffffu);
            if (todo) {  // wave-uniform
                const int leader = __builtin_ctzll(todo);
                const unsigned b0 = (unsigned)__shfl((int)bk, leader, 64);
                const unsigned long long same = __ballot(bk == b0);
                if (lane == leader) atomicAdd(const_cast<unsigned*>(&hist[b0]), (unsigned)__popcll(same));
                else if (bk != 0xffffffffu && bk != b0) atomicAdd(const_cast<unsigned*>(&hist[bk]), 1u);  // the other buckets: one add per lane
            }
        }
        __builtin_amdgcn_wave_barrier();
        // lane l owns buckets 255 - 4 l ... 252 - 4 l (descending): the bucket where the running count from the top reaches `remaining`
        unsigned c[4], local = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = hist[255 - (4 * lane + j)]; local += c[j]; }
        unsigned incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += t;
        }
        unsigned run = incl - local, bkt = 0u, newrem = 0u;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!found && run < remaining && run + c[j] >= remaining) { found = true; bkt = 255u - (unsigned)(4 * lane + j); newrem = remaining - run; }
            run += c[j];
        }
        const unsigned long long who = __ballot(found);  // exactly one lane (remaining >= 1 and the counts sum to >= remaining)
        const int src = who ? __builtin_ctzll(who) : 0;
        bkt = (unsigned)__shfl((int)bkt, src, 64);
        newrem = (unsigned)__shfl((int)newrem, src, 64);
        prefix |= bkt << shift;
        remaining = newrem;
        mask |= 0xffu << shift;
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned thr = prefix, need_eq = remaining, n_gt = (unsigned)keff - need_eq;
    slots[lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    unsigned cnt_gt = 0u, eq_seen = 0u;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        unsigned key = 0u;
        bool gt = false, eq = false;
        if (i < n) {
            key = ord_key(s[i]);
            gt = key > thr;
            eq = key == thr;
        }
        const unsigned long long comp = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned)(col0 + i));
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        if (gt) slots[cnt_gt + (unsigned)__popcll(bg & lt)] = comp;   // cnt_gt never exceeds n_gt <= 64 - need_eq
        const unsigned r = eq_seen + (unsigned)__popcll(be & lt);
        if (eq && r < need_eq) slots[n_gt + r] = comp;
        cnt_gt += (unsigned)__popcll(bg);
        eq_seen += (unsigned)__popcll(be);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long v = keff > 0 ? slots[lane] : 0ull;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, stride);
            const bool up = ((lane & size) == 0), lower = ((lane & stride) == 0);
            const bool take_max = (up == lower);
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}

__global__ __launch_bounds__(kThreads, 1) void score_part_topk_kernel(const float* __restrict__ U, const float* __restrict__ E, int nU, int M, int d, int k,
                                                                     int P, int part_cols, unsigned long long* __restrict__ cand, float* __restrict__ cand_val,
                                                                     float* __restrict__ pmax, float* __restrict__ psum,
                                                                     const int64_t* __restrict__ labels, float inv_temp, float* __restrict__ row_lab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage = lds;                                   // 2 x STAGE_F
    float* strip = lds + 2 * STAGE_F;                     // SBM x PART
    unsigned* hist_all = reinterpret_cast<unsigned*>(strip + SBM * PART);                            // kWaves x 256
    unsigned long long* slots_all = reinterpret_cast<unsigned long long*>(hist_all + kWaves * 256);  // kWaves x 64

    const int part = blockIdx.x, m0 = blockIdx.y * SBM;
    const int c_begin = part * part_cols;  // part_cols: 256 / 512 / 768 (<= PART, the strip's row length), chosen per launch to fill the chip
    const int ncols = (M - c_begin) < part_cols ? (M - c_begin) : part_cols;   // >= 1 by construction of P
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;

    // staging maps (gemm.hip): A rows by threads 0..127, B rows tid >> 2 + 128 q
    const int sr = tid >> 2, kq = tid & 3;
    int ar = m0 + (sr & 31);
    ar = ar < nU ? ar : nU - 1;
    const float* ga = U + (int64_t)ar * d + kq * 4;
    const int wa = (sr & 31) * SSTR + 2 * kq;
    const bool stage_a = tid < 128;  // wave-uniform
    const int ra = lr * SSTR + lh * 8;
    const int rb = (SBM + wave * 32 + lr) * SSTR + lh * 8;
    const int nk = d / SBK;

    for (int c0 = 0; c0 < ncols; c0 += SBN) {
        const float* gb[2];
        int wb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int br = c_begin + c0 + sr + 128 * q;
            br = br < M ? br : M - 1;
            gb[q] = E + (int64_t)br * d + kq * 4;
            wb[q] = (SBM + sr + 128 * q) * SSTR + 2 * kq;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // two named staging sets: a k-tile's global loads are issued two tiles before they are stored to LDS (one workgroup per CU and one
        // wave per SIMD: nothing else covers the load latency), as the prefetch-distance-2 loop of gemm_bf16.hip
        float4 xa, xb0, xb1, ya, yb0, yb1;
#define SF_GLOAD(S, k0_)                                                      \
    do {                                                                      \
        if (stage_a) S##a = *reinterpret_cast<const float4*>(ga + (k0_));     \
        S##b0 = *reinterpret_cast<const float4*>(gb[0] + (k0_));              \
        S##b1 = *reinterpret_cast<const float4*>(gb[1] + (k0_));              \
    } while (0)
#define SF_ST(buf_, off_, v_)                                                            \
    do {                                                                                 \
        *reinterpret_cast<float2*>((buf_) + (off_)) = make_float2((v_).x, (v_).z);       \
        *reinterpret_cast<float2*>((buf_) + (off_) + 8) = make_float2((v_).y, (v_).w);   \
    } while (0)
#define SF_LSTORE(S, buf_)                      \
    do {                                        \
        if (stage_a) SF_ST(buf_, wa, S##a);     \
        SF_ST(buf_, wb[0], S##b0);              \
        SF_ST(buf_, wb[1], S##b1);              \
    } while (0)
#define SF_COMPUTE(buf_)                                                                                                   \
    do {                                                                                                                   \
        const float* b_ = (buf_);                                                                                          \
        const float4 a0 = *reinterpret_cast<const float4*>(b_ + ra), a1 = *reinterpret_cast<const float4*>(b_ + ra + 4);   \
        const float4 b0 = *reinterpret_cast<const float4*>(b_ + rb), b1 = *reinterpret_cast<const float4*>(b_ + rb + 4);   \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                                    \
            const float4 fa = (s >> 2) ? a1 : a0, fb = (s >> 2) ? b1 : b0;                                                 \
            const float av = (s & 3) == 0 ? fa.x : ((s & 3) == 1 ? fa.y : ((s & 3) == 2 ? fa.z : fa.w));                   \
            const float bv = (s & 3) == 0 ? fb.x : ((s & 3) == 1 ? fb.y : ((s & 3) == 2 ? fb.z : fb.w));                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);                                              \
        }                                                                                                                  \
    } while (0)
        float* buf0 = stage;
        float* buf1 = stage + STAGE_F;
        const int last = nk - 1;
        SF_GLOAD(x, 0);
        SF_LSTORE(x, buf0);
        SF_GLOAD(y, (1 <= last ? 1 : 0) * SBK);
        SF_GLOAD(x, (2 <= last ? 2 : 0) * SBK);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {   // nk even (host-checked); past-the-end prefetches re-read tile 0 and are never consumed
            SF_COMPUTE(buf0);
            __builtin_amdgcn_sched_barrier(0);
            SF_LSTORE(y, buf1);
            SF_GLOAD(y, (kt + 3 <= last ? kt + 3 : 0) * SBK);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            SF_COMPUTE(buf1);
            __builtin_amdgcn_sched_barrier(0);
            SF_LSTORE(x, buf0);
            SF_GLOAD(x, (kt + 4 <= last ? kt + 4 : 0) * SBK);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
#undef SF_COMPUTE
#undef SF_GLOAD
#undef SF_ST
#undef SF_LSTORE
        // accumulators -> strip (C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
        {
            const int col = c0 + wave * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * PART + col] = acc[r];
        }
        // (the next chunk's first staging store follows a barrier that every wave reaches only after these stores were issued; the
        // selection below waits at its own barrier)
    }
    __syncthreads();

    // ---- per-row selection: wave w takes rows w, w + 8, ...
    volatile unsigned* hist = hist_all + wave * 256;
    volatile unsigned long long* slots = slots_all + wave * 64;
    for (int r = wave; r < SBM; r += kWaves) {
        const int row = m0 + r;
        if (row >= nU) break;  // wave-uniform
        const float* s = strip + r * PART;
        const unsigned long long v = wave_topk(s, ncols, k, c_begin, hist, slots, lane);
        const int64_t base = ((int64_t)row * P + part) * k;
        if (lane < k) {
            cand[base + lane] = v;
            unsigned idx = 0xffffffffu - (unsigned)(v & 0xffffffffull);
            cand_val[base + lane] = v ? s[idx - (unsigned)c_begin] : 0.f;
        }
        if (labels) {
            const unsigned best_lo = (unsigned)__shfl((int)(unsigned)(v & 0xffffffffull), 0, 64);  // lane 0 holds the part's best candidate
            const unsigned best = 0xffffffffu - best_lo - (unsigned)c_begin;
            const float mx = s[best < (unsigned)ncols ? best : 0u] * inv_temp;  // NaN rows propagate NaN like torch.cross_entropy
            float acc_e = 0.f;
            for (int i = lane; i < ncols; i += 64) acc_e += expf(s[i] * inv_temp - mx);
            acc_e = mr::wave_sum(acc_e);
            if (lane == 0) {
                pmax[(int64_t)row * P + part] = mx;
                psum[(int64_t)row * P + part] = acc_e;
                const int64_t lab = labels[row];
                if (row_lab) {
                    if (lab >= c_begin && lab < c_begin + ncols) row_lab[row] = s[lab - c_begin] * inv_temp;
                    else if (part == 0 && (lab < 0 || lab >= M)) row_lab[row] = NAN;
                }
            }
        }
    }
}

// One wave per user: the top-k of the parts' candidates.  Among equal scores the candidate array's order (part-major, canonical inside a
// part) IS ascending item order, so the same wave select, run on the candidates' raw scores with ties by array position, yields the
// canonical list; positions are mapped back to items through the composites.  n_valid: the candidates form a prefix of the (P, k) block
// (only the last part can hold fewer than k items).
__global__ __launch_bounds__(64) void score_merge_kernel(const unsigned long long* __restrict__ cand, const float* __restrict__ cand_val,
                                                        const float* __restrict__ pmax, const float* __restrict__ psum, int P, int k, int n_valid,
                                                        float* __restrict__ top_val, int64_t* __restrict__ top_idx,
                                                        const int64_t* __restrict__ labels, float* __restrict__ row_lse,
                                                        int32_t* __restrict__ label_rank) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, row = blockIdx.x;
    const int n_pad = (n_valid + 3) & ~3;
    volatile unsigned* hist = reinterpret_cast<unsigned*>(sm + n_pad);
    volatile unsigned long long* slots = reinterpret_cast<unsigned long long*>(sm + n_pad + 256);
    const unsigned long long* __restrict__ c = cand + (int64_t)row * P * k;
    const float* __restrict__ cv = cand_val + (int64_t)row * P * k;
    for (int i = lane; i < n_valid; i += 64) sm[i] = cv[i];
    __builtin_amdgcn_wave_barrier();
    const unsigned long long v = wave_topk(sm, n_valid, k, 0, hist, slots, lane);
    const unsigned pos = 0xffffffffu - (unsigned)(v & 0xffffffffull);
    int64_t col = -1;
    if (lane < k && v != 0ull) {
        col = (int64_t)(0xffffffffu - (unsigned)(c[pos] & 0xffffffffull));
        top_idx[(int64_t)row * k + lane] = col;
        top_val[(int64_t)row * k + lane] = sm[pos];
    }
    if (labels) {
        const int64_t lab = labels[row];
        const unsigned long long hit = __ballot(col >= 0 && col == lab);
        if (lane == 0 && label_rank) label_rank[row] = hit ? (int32_t)__builtin_ctzll(hit) : -1;
        if (row_lse) {
            float mx = -INFINITY;
            bool bad = false;
            for (int p = lane; p < P; p += 64) {
                const float m = pmax[(int64_t)row * P + p];
                bad = bad || (m != m);
                mx = fmaxf(mx, m);
            }
            mx = mr::wave_max(mx);
            float tot = 0.f;
            for (int p = lane; p < P; p += 64) tot += psum[(int64_t)row * P + p] * expf(pmax[(int64_t)row * P + p] - mx);
            tot = mr::wave_sum(tot);
            const unsigned long long anybad = __ballot(bad);
            if (lane == 0) row_lse[row] = anybad ? NAN : mx + logf(tot);
        }
    }
}

}  // namespace

namespace mr {

// columns per workgroup: the widest part (fewest candidates, best reuse of the staged user rows) that still gives the chip ~one workgroup
// per CU; small catalogs / few users take narrower parts
int score_part_cols(int64_t nU, int64_t M) {
    const int64_t strips = (nU + SBM - 1) / SBM;
    for (int pc = PART; pc > SBN; pc -= SBN)
        if (((M + pc - 1) / pc) * strips >= 224) return pc;
    return SBN;
}
int score_parts(int64_t nU, int64_t M) { const int pc = score_part_cols(nU, M); return (int)((M + pc - 1) / pc); }

bool score_fused_supported(int64_t nU, int64_t M, int d, int k) {
    return k >= 1 && k <= 64 && k <= M && d % (2 * SBK) == 0 && score_parts(nU, M) <= kMaxParts && nU > 0 && (nU + SBM - 1) / SBM <= 65535;
}

size_t score_fused_ws_bytes(int64_t nU, int64_t M, int k) {
    const size_t P = (size_t)score_parts(nU, M);
    return (size_t)nU * P * (size_t)k * 12 + (size_t)nU * P * 8 + 512;
}

constexpr int kMaxDevices = 64;

int score_fused_launch(const float* U, const float* E, int64_t nU, int64_t M, int d, int k, float* top_val, int64_t* top_idx, const int64_t* labels,
                       float inv_temp, float* row_lse, float* row_lab, int32_t* label_rank, void* ws, hipStream_t st) {
    const int pc = score_part_cols(nU, M), P = score_parts(nU, M);
    unsigned char* w = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(w);
    float* cand_val = reinterpret_cast<float*>(w + (size_t)nU * P * k * 8);
    float* pmax = cand_val + (size_t)nU * P * k;
    float* psum = pmax + (size_t)nU * P;
    const size_t shm = (size_t)(2 * STAGE_F + SBM * PART) * sizeof(float) + kWaves * 256 * 4 + kWaves * 64 * 8;
    // dynamic-LDS ceilings: per device, return code checked (a failed call would otherwise surface as an opaque launch error, or only on
    // the second device of a process)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MR_ELAUNCH;
    static std::mutex attr_mu;
    static size_t attr_part[kMaxDevices], attr_merge[kMaxDevices];
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (shm > attr_part[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_part_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
                (void)hipGetLastError();
                return MR_EUNSUPPORTED;
            }
            attr_part[dev] = shm;
        }
    }
    const int last = (int)(M - (int64_t)(P - 1) * pc);
    const int n_valid = (P - 1) * k + (k < last ? k : last);
    const size_t shm_m = (size_t)((n_valid + 3) & ~3) * sizeof(float) + 256 * 4 + 64 * 8;
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (shm_m > attr_merge[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&score_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_m) != hipSuccess) {
                (void)hipGetLastError();
                return MR_EUNSUPPORTED;
            }
            attr_merge[dev] = shm_m;
        }
    }
    hipLaunchKernelGGL(score_part_topk_kernel, dim3(P, (unsigned)((nU + SBM - 1) / SBM)), dim3(kThreads), shm, st, U, E, (int)nU, (int)M, d, k, P, pc, cand,
                       cand_val, pmax, psum, labels, inv_temp, row_lab);
    hipLaunchKernelGGL(score_merge_kernel, dim3((unsigned)nU), dim3(64), shm_m, st, cand, cand_val, pmax, psum, P, k, n_valid, top_val, top_idx, labels,
                       row_lse, label_rank);
    return check_launch();
}

}  // namespace mr
