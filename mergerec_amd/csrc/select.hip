// Task-vector pre-processing (run once at init): exact global top-k by magnitude over a flat
// parameter vector (radix select, HBM-streaming), and the elementwise combines of TIES and Localize-and-Stitch
// written to mirror torch's arithmetic bit for bit.
//
//   mr_abs_kth_largest_f32 : 4 x (256-bin histogram pass over |x| keys + one-thread bucket pick), no host sync
//   mr_abs_topk_mask_f32   : y = x where x is among the k largest |x| (ties at the threshold -> lowest indices), else 0
//   mr_ties_combine_f32    : algorithms/ties.py:31-72 (sign election, disjoint mean) in place on the (N, P) masked updates
//   mr_lns_combine_f32     : algorithms/localize_and_stitch.py:43-49
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 2048;  // elements per workgroup in the ordered (tie-ranking) passes

struct SelectState {  // lives in the caller's workspace
    unsigned prefix, mask;
    long long remaining;  // how many more elements (from the current prefix class) belong to the top-k
    unsigned hist[256];
    long long total_eq;
};

__device__ __forceinline__ unsigned abs_key(float f) { return __float_as_uint(f) & 0x7fffffffu; }
// order-preserving key of a signed float (-0.0 == +0.0); larger value <=> larger key
__device__ __forceinline__ unsigned ord_key(float f) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
template <int SIGNED>
__device__ __forceinline__ unsigned sel_key(float f) { return SIGNED ? ord_key(f) : abs_key(f); }

__global__ void select_init_kernel(SelectState* st, long long k) {
    if (threadIdx.x == 0) { st->prefix = 0u; st->mask = 0u; st->remaining = k; st->total_eq = 0; }
    st->hist[threadIdx.x] = 0u;
}

template <int SIGNED>
__global__ __launch_bounds__(kThreads) void abs_hist_kernel(const float* __restrict__ x, int64_t n, int shift,
                                                           SelectState* __restrict__ st) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned prefix = st->prefix, mask = st->mask;
    const int64_t n4 = n >> 2;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < n4; v += (int64_t)gridDim.x * kThreads) {
        const float4 f = reinterpret_cast<const float4*>(x)[v];
        const unsigned k0 = sel_key<SIGNED>(f.x), k1 = sel_key<SIGNED>(f.y), k2 = sel_key<SIGNED>(f.z), k3 = sel_key<SIGNED>(f.w);
        if ((k0 & mask) == prefix) atomicAdd(&h[(k0 >> shift) & 0xffu], 1u);
        if ((k1 & mask) == prefix) atomicAdd(&h[(k1 >> shift) & 0xffu], 1u);
        if ((k2 & mask) == prefix) atomicAdd(&h[(k2 >> shift) & 0xffu], 1u);
        if ((k3 & mask) == prefix) atomicAdd(&h[(k3 >> shift) & 0xffu], 1u);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const unsigned kk = sel_key<SIGNED>(x[i]);
        if ((kk & mask) == prefix) atomicAdd(&h[(kk >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void select_pick_kernel(SelectState* st, int shift) {
    if (threadIdx.x == 0) {
        long long rem = st->remaining, c = 0;
        int b = 255;
        for (; b > 0; --b) {
            if (c + (long long)st->hist[b] >= rem) break;
            c += st->hist[b];
        }
        st->remaining = rem - c;
        st->prefix |= (unsigned)b << shift;
        st->mask |= 0xffu << shift;
        if (shift == 0) st->total_eq = st->hist[b];
    }
    __syncthreads();
    st->hist[threadIdx.x] = 0u;
}

__global__ void select_publish_kernel(const SelectState* st, unsigned* thr_bits, long long* need_eq) {
    if (threadIdx.x == 0) { *thr_bits = st->prefix; *need_eq = st->remaining; }
}

// ordered pass 1: how many elements equal to the threshold key does each chunk hold
__global__ __launch_bounds__(kThreads) void eq_count_kernel(const float* __restrict__ x, int64_t n,
                                                           const unsigned* __restrict__ thr_bits,
                                                           long long* __restrict__ chunk_cnt) {
    __shared__ unsigned cnt;
    if (threadIdx.x == 0) cnt = 0u;
    __syncthreads();
    const unsigned thr = *thr_bits;
    const int64_t b = (int64_t)blockIdx.x * kChunk;
    unsigned c = 0;
    for (int i = threadIdx.x; i < kChunk; i += kThreads)
        if (b + i < n && abs_key(x[b + i]) == thr) ++c;
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = cnt;
}

// ordered pass 2: exclusive scan of the chunk counts (one workgroup, sequential over tiles of 256)
__global__ __launch_bounds__(kThreads) void eq_scan_kernel(long long* __restrict__ chunk_cnt, int64_t nchunk) {
    __shared__ long long tile[kThreads];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < nchunk; t0 += kThreads) {
        const int64_t i = t0 + threadIdx.x;
        const long long v = i < nchunk ? chunk_cnt[i] : 0;
        tile[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < kThreads; o <<= 1) {  // Hillis-Steele inclusive scan
            const long long a = threadIdx.x >= o ? tile[threadIdx.x - o] : 0;
            __syncthreads();
            tile[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nchunk) chunk_cnt[i] = carry + tile[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) carry += tile[kThreads - 1];
        __syncthreads();
    }
}

// ordered pass 3: write the masked vector (and the 0/1 mask)
__global__ __launch_bounds__(kThreads) void topk_mask_kernel(const float* __restrict__ x, int64_t n,
                                                            const unsigned* __restrict__ thr_bits,
                                                            const long long* __restrict__ need_eq_p,
                                                            const long long* __restrict__ chunk_off,
                                                            float* __restrict__ y, uint8_t* __restrict__ m) {
    __shared__ unsigned wave_cnt[kThreads / MR_WAVE];
    const unsigned thr = *thr_bits;
    const long long need_eq = *need_eq_p;
    const int64_t b = (int64_t)blockIdx.x * kChunk;
    long long run = chunk_off[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int s = 0; s < kChunk; s += kThreads) {
        const int64_t i = b + s + threadIdx.x;
        const bool in = i < n;
        const float v = in ? x[i] : 0.f;
        const unsigned key = abs_key(v);
        const bool eq = in && key == thr;
        const unsigned long long beq = __ballot(eq);
        if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(beq);
        __syncthreads();
        long long before = run;
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / MR_WAVE; ++w) {
            if (w < wave) before += wave_cnt[w];
            total += wave_cnt[w];
        }
        bool keep = in && key > thr;
        if (eq) keep = (before + (long long)__popcll(beq & lt)) < need_eq;
        if (in) {
            y[i] = keep ? v : 0.f;
            if (m) m[i] = keep ? 1 : 0;
        }
        run += total;
        __syncthreads();
    }
}

// ties.py:31-72 on the (N, P) sparse updates, in place.  All sums run sequentially over the task index like torch's dim-0 sum.
__global__ __launch_bounds__(kThreads) void ties_combine_kernel(float* __restrict__ sp, int64_t stride, int N, int64_t P) {
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
        float pos = 0.f, neg = 0.f;
        for (int i = 0; i < N; ++i) {
            const float v = sp[(int64_t)i * stride + p];
            pos = __fadd_rn(pos, v > 0.f ? v : 0.f);
            neg = __fadd_rn(neg, v < 0.f ? v : 0.f);
        }
        float sign;
        if (pos != 0.f && neg != 0.f) {
            sign = fabsf(pos) >= fabsf(neg) ? 1.f : -1.f;
        } else {
            const float t = __fadd_rn(pos, neg);
            sign = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);  // torch.sign (NaN -> NaN is not reproduced: inputs are finite)
        }
        if (sign == 0.f) sign = 1.f;
        int cnt = 0;
        for (int i = 0; i < N; ++i) {
            const float v = sp[(int64_t)i * stride + p];
            const float sel = sign > 0.f ? (v > 0.f ? v : 0.f) : (v < 0.f ? v : 0.f);
            cnt += (sel != 0.f);
        }
        for (int i = 0; i < N; ++i) {
            const float v = sp[(int64_t)i * stride + p];
            const float sel = sign > 0.f ? (v > 0.f ? v : 0.f) : (v < 0.f ? v : 0.f);
            sp[(int64_t)i * stride + p] = cnt ? __fdiv_rn(sel, (float)cnt) : 0.f;  // 0/0 -> nan_to_num(0)
        }
    }
}

// localize_and_stitch.py:43-49: out_i = (mask_i / max(sum_j mask_j, 1)) * tau_i
__global__ __launch_bounds__(kThreads) void lns_combine_kernel(const float* __restrict__ tv, const uint8_t* __restrict__ m,
                                                              int64_t stride, int N, int64_t P, float* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
        float cnt = 0.f;
        for (int i = 0; i < N; ++i) cnt += (float)m[(int64_t)i * stride + p];
        const float denom = cnt < 1.f ? 1.f : cnt;
        for (int i = 0; i < N; ++i) {
            const float pm = __fdiv_rn((float)m[(int64_t)i * stride + p], denom);
            out[(int64_t)i * stride + p] = __fmul_rn(pm, tv[(int64_t)i * stride + p]);
        }
    }
}

// converts the selected key back to the float value it encodes
__global__ void select_publish_value_kernel(const SelectState* st, int is_signed, float* out) {
    if (threadIdx.x == 0) {
        unsigned k = st->prefix;
        if (is_signed) k = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        *out = __uint_as_float(k);
    }
}

// pcb.py:44-52 for one task row: a = clamp(|tau|, lo, hi); clamped = sign(tau) * a; self = ((a - lo) / (hi - lo))^2;
// task_pcb = exp(n * self) * tanh(tau * sum_j tau_j).  q holds [lo, hi] of this row (device).
__global__ __launch_bounds__(kThreads) void pcb_stage1_kernel(const float* __restrict__ tv_all, int64_t stride, int N, int row,
                                                             int64_t P, const float* __restrict__ q,
                                                             float* __restrict__ clamped, float* __restrict__ task_pcb) {
    const float lo = q[0], hi = q[1];
    const float* tv = tv_all + (int64_t)row * stride;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
        const float t = tv[p];
        float a = fabsf(t);
        a = fminf(fmaxf(a, lo), hi);
        const float sgn = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
        clamped[p] = sgn * a;
        float nrm = (a - lo) / (hi - lo);
        nrm = nrm * nrm;
        float sum = 0.f;
        for (int j = 0; j < N; ++j) sum += tv_all[(int64_t)j * stride + p];
        task_pcb[p] = expf((float)N * nrm) * tanhf(t * sum);
    }
}

// pcb.py:54-58: scale_i = (clamp(task_pcb_i, q_i, max_i) - q_i) / (max_i - q_i); out_i = clamped_i * scale_i / max(sum_j scale_j, 1e-12) / n
__global__ __launch_bounds__(kThreads) void pcb_stage2_kernel(const float* __restrict__ clamped, const float* __restrict__ task_pcb,
                                                             int64_t stride, int N, int64_t P, const float* __restrict__ q2,
                                                             float* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < P; p += (int64_t)gridDim.x * kThreads) {
        float ssum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float lo = q2[2 * j], hi = q2[2 * j + 1];
            const float x = fminf(fmaxf(task_pcb[(int64_t)j * stride + p], lo), hi);
            ssum += (x - lo) / (hi - lo);
        }
        const float den = fmaxf(ssum, 1e-12f);
        for (int j = 0; j < N; ++j) {
            const float lo = q2[2 * j], hi = q2[2 * j + 1];
            const float x = fminf(fmaxf(task_pcb[(int64_t)j * stride + p], lo), hi);
            const float sc = (x - lo) / (hi - lo);
            out[(int64_t)j * stride + p] = clamped[(int64_t)j * stride + p] * sc / den / (float)N;
        }
    }
}

inline int64_t nchunks(int64_t n) { return (n + kChunk - 1) / kChunk; }
inline unsigned stream_blocks(int64_t n) {
    int64_t b = (n / 4 + kThreads - 1) / kThreads;
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" size_t mr_select_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    return 4096 + (size_t)(nchunks(n) + 1) * sizeof(long long);
}

extern "C" int mr_abs_kth_largest_f32(const float* x, int64_t n, int64_t k, uint32_t* thr_bits, int64_t* need_eq, void* ws,
                                      size_t ws_bytes, mr_stream_t stream) {
    if (!x || !thr_bits || !need_eq || !ws || n < 1 || k < 1 || k > n) return MR_EINVAL;
    if (!mr::aligned16(x) || !mr::aligned16(ws)) return MR_EALIGN;
    if (ws_bytes < mr_select_ws_bytes(n)) return MR_EWS;
    hipStream_t st = (hipStream_t)stream;
    SelectState* s = reinterpret_cast<SelectState*>(ws);
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, st, s, (long long)k);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hipLaunchKernelGGL(abs_hist_kernel<0>, dim3(stream_blocks(n)), dim3(kThreads), 0, st, x, n, shift, s);
        hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, st, s, shift);
    }
    hipLaunchKernelGGL(select_publish_kernel, dim3(1), dim3(64), 0, st, s, thr_bits, reinterpret_cast<long long*>(need_eq));
    return mr::check_launch();
}

extern "C" int mr_abs_topk_mask_f32(const float* x, int64_t n, const uint32_t* thr_bits, const int64_t* need_eq, float* y,
                                    uint8_t* mask_or_null, void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (!x || !thr_bits || !need_eq || !y || !ws || n < 1) return MR_EINVAL;
    if (ws_bytes < mr_select_ws_bytes(n)) return MR_EWS;
    hipStream_t st = (hipStream_t)stream;
    long long* chunk = reinterpret_cast<long long*>(reinterpret_cast<char*>(ws) + 4096);
    const int64_t nc = nchunks(n);
    hipLaunchKernelGGL(eq_count_kernel, dim3((unsigned)nc), dim3(kThreads), 0, st, x, n, thr_bits, chunk);
    hipLaunchKernelGGL(eq_scan_kernel, dim3(1), dim3(kThreads), 0, st, chunk, nc);
    hipLaunchKernelGGL(topk_mask_kernel, dim3((unsigned)nc), dim3(kThreads), 0, st, x, n, thr_bits,
                       reinterpret_cast<const long long*>(need_eq), chunk, y, mask_or_null);
    return mr::check_launch();
}

extern "C" int mr_ties_combine_f32(float* sparse, int64_t stride, int N, int64_t P, mr_stream_t stream) {
    if (!sparse || N < 1 || P < 0 || stride < P) return MR_EINVAL;
    if (P == 0) return MR_OK;
    hipLaunchKernelGGL(ties_combine_kernel, dim3(stream_blocks(P * 4)), dim3(kThreads), 0, (hipStream_t)stream, sparse, stride, N, P);
    return mr::check_launch();
}

extern "C" int mr_lns_combine_f32(const float* tv, const uint8_t* mask, int64_t stride, int N, int64_t P, float* out,
                                  mr_stream_t stream) {
    if (!tv || !mask || !out || N < 1 || P < 0 || stride < P) return MR_EINVAL;
    if (P == 0) return MR_OK;
    hipLaunchKernelGGL(lns_combine_kernel, dim3(stream_blocks(P * 4)), dim3(kThreads), 0, (hipStream_t)stream, tv, mask, stride, N, P, out);
    return mr::check_launch();
}

// value of the k-th largest element of x (is_signed: by value; else by magnitude), written to *out (device float)
extern "C" int mr_kth_largest_value_f32(const float* x, int64_t n, int64_t k, int is_signed, float* out, void* ws, size_t ws_bytes,
                                        mr_stream_t stream) {
    if (!x || !out || !ws || n < 1 || k < 1 || k > n) return MR_EINVAL;
    if (!mr::aligned16(x) || !mr::aligned16(ws)) return MR_EALIGN;
    if (ws_bytes < mr_select_ws_bytes(n)) return MR_EWS;
    hipStream_t st = (hipStream_t)stream;
    SelectState* s = reinterpret_cast<SelectState*>(ws);
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, st, s, (long long)k);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (is_signed)
            hipLaunchKernelGGL(abs_hist_kernel<1>, dim3(stream_blocks(n)), dim3(kThreads), 0, st, x, n, shift, s);
        else
            hipLaunchKernelGGL(abs_hist_kernel<0>, dim3(stream_blocks(n)), dim3(kThreads), 0, st, x, n, shift, s);
        hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, st, s, shift);
    }
    hipLaunchKernelGGL(select_publish_value_kernel, dim3(1), dim3(64), 0, st, s, is_signed, out);
    return mr::check_launch();
}

extern "C" int mr_pcb_stage1_f32(const float* tv, int64_t stride, int N, int row, int64_t P, const float* q_lo_hi, float* clamped,
                                 float* task_pcb, mr_stream_t stream) {
    if (!tv || !q_lo_hi || !clamped || !task_pcb || N < 1 || row < 0 || row >= N || P < 0 || stride < P) return MR_EINVAL;
    if (P == 0) return MR_OK;
    hipLaunchKernelGGL(pcb_stage1_kernel, dim3(stream_blocks(P * 4)), dim3(kThreads), 0, (hipStream_t)stream, tv, stride, N, row, P,
                       q_lo_hi, clamped, task_pcb);
    return mr::check_launch();
}

extern "C" int mr_pcb_stage2_f32(const float* clamped, const float* task_pcb, int64_t stride, int N, int64_t P, const float* q2,
                                 float* out, mr_stream_t stream) {
    if (!clamped || !task_pcb || !q2 || !out || N < 1 || P < 0 || stride < P) return MR_EINVAL;
    if (P == 0) return MR_OK;
    hipLaunchKernelGGL(pcb_stage2_kernel, dim3(stream_blocks(P * 4)), dim3(kThreads), 0, (hipStream_t)stream, clamped, task_pcb, stride,
                       N, P, q2, out);
    return mr::check_launch();
}
