// K1 / K1b / K6: the N-way alpha-weighted parameter interpolation and its alpha-gradient.
// HBM-streaming kernels: 16-byte coalesced loads, no LDS staging needed (no reuse), alpha in SGPRs.
//
// Arithmetic contract (bit-exact with torch CPU, see include/mergerec_hip.h): products are rounded
// separately (no FMA contraction), summed sequentially from 0 in task order, then added to base.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kVPT = 4;  // float4 per thread per chunk and stream: (N + 1) x 4 16-byte loads in flight per thread (r04: 2 -> 4, +5 %)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    float4 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2);
    v.w = __builtin_nontemporal_load(p + 3);
    return v;
}

__device__ __forceinline__ void st4_nt(float* p, const float4 v) {
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
    __builtin_nontemporal_store(v.z, p + 2);
    __builtin_nontemporal_store(v.w, p + 3);
}

template <int N, bool SEG>
__global__ __launch_bounds__(kThreads) void merge_nway_kernel(const float* __restrict__ base,
                                                             const float* __restrict__ tv, int64_t tv_stride,
                                                             const float* __restrict__ alpha,
                                                             const int64_t* __restrict__ seg_off, int S,
                                                             int64_t p_begin, int64_t p_count,
                                                             float* __restrict__ out) {
    const int64_t nvec = p_count >> 2;
    constexpr int64_t kChunk = (int64_t)kThreads * kVPT;
    const int64_t nchunk = (nvec + kChunk - 1) / kChunk;
    for (int64_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
        const int64_t v0 = chunk * kChunk;
        int s_lo = 0;
        bool uniform = true;
        if (SEG) {
            const int64_t pf = p_begin + v0 * 4;
            int64_t pl = p_begin + (v0 + kChunk) * 4;
            if (pl > p_begin + p_count) pl = p_begin + p_count;
            int lo = 0, hi = S - 1;  // largest s with seg_off[s] <= pf
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (seg_off[mid] <= pf) lo = mid; else hi = mid - 1;
            }
            s_lo = lo;
            uniform = (pl <= seg_off[s_lo + 1]);
        }
        float a[N];
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = alpha[(int64_t)s_lo * N + i];

        float4 t[kVPT][N];
        float4 b[kVPT];
        int64_t p[kVPT];
#pragma unroll
        for (int u = 0; u < kVPT; ++u) {
            const int64_t v = v0 + (int64_t)u * kThreads + threadIdx.x;
            p[u] = (v < nvec) ? p_begin + v * 4 : -1;
            if (p[u] >= 0) {
                b[u] = ld4_nt(base + p[u]);  // every operand is read once and the result written once: non-temporal both ways (r04: +9 %)
#pragma unroll
                for (int i = 0; i < N; ++i) t[u][i] = ld4_nt(tv + (int64_t)i * tv_stride + p[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kVPT; ++u) {
            if (p[u] < 0) continue;
            float al[N];
#pragma unroll
            for (int i = 0; i < N; ++i) al[i] = a[i];
            if (SEG && !uniform) {
                int s = s_lo;
                while (s + 1 < S && p[u] >= seg_off[s + 1]) ++s;
#pragma unroll
                for (int i = 0; i < N; ++i) al[i] = alpha[(int64_t)s * N + i];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                acc.x = __fadd_rn(acc.x, __fmul_rn(al[i], t[u][i].x));
                acc.y = __fadd_rn(acc.y, __fmul_rn(al[i], t[u][i].y));
                acc.z = __fadd_rn(acc.z, __fmul_rn(al[i], t[u][i].z));
                acc.w = __fadd_rn(acc.w, __fmul_rn(al[i], t[u][i].w));
            }
            float4 o;
            o.x = __fadd_rn(b[u].x, acc.x);
            o.y = __fadd_rn(b[u].y, acc.y);
            o.z = __fadd_rn(b[u].z, acc.z);
            o.w = __fadd_rn(b[u].w, acc.w);
            st4_nt(out + p[u], o);
        }
    }
}

// Rows of ONE table of the arena (the alpha-learning step reads only its batch's word-embedding rows): out[off + r d + c] for the rows r
// listed in idx (duplicates write the same bits), the per-element operations of merge_nway_kernel in its order.  One wave per listed row.
__global__ __launch_bounds__(kThreads) void merge_rows_kernel(const float* __restrict__ base, const float* __restrict__ tv, int64_t tv_stride,
                                                             const float* __restrict__ alpha, int N, const int32_t* __restrict__ idx, int T,
                                                             int rows, int d, int64_t off, float* __restrict__ out) {
    const int t = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= T) return;
    const int r = idx[t];
    if (r < 0 || r >= rows) return;  // (ids are validated when the batch is packed; never index outside the table)
    const int64_t p0 = off + (int64_t)r * d;
    for (int c = (threadIdx.x & 63) * 4; c < d; c += 256) {
        const int64_t p = p0 + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < N; ++i) {
            const float al = alpha[i];
            const float4 tt = ld4(tv + (int64_t)i * tv_stride + p);
            acc.x = __fadd_rn(acc.x, __fmul_rn(al, tt.x));
            acc.y = __fadd_rn(acc.y, __fmul_rn(al, tt.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(al, tt.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(al, tt.w));
        }
        const float4 b = ld4(base + p);
        float4 o;
        o.x = __fadd_rn(b.x, acc.x);
        o.y = __fadd_rn(b.y, acc.y);
        o.z = __fadd_rn(b.z, acc.z);
        o.w = __fadd_rn(b.w, acc.w);
        *reinterpret_cast<float4*>(out + p) = o;
    }
}

// generic N (> 8): runtime loop, one float4 per thread per step
template <bool SEG>
__global__ __launch_bounds__(kThreads) void merge_nway_generic_kernel(const float* __restrict__ base,
                                                                     const float* __restrict__ tv,
                                                                     int64_t tv_stride,
                                                                     const float* __restrict__ alpha,
                                                                     const int64_t* __restrict__ seg_off, int N,
                                                                     int S, int64_t p_begin, int64_t p_count,
                                                                     float* __restrict__ out) {
    const int64_t nvec = p_count >> 2;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kThreads) {
        const int64_t p = p_begin + v * 4;
        int s = 0;
        if (SEG) {
            int lo = 0, hi = S - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (seg_off[mid] <= p) lo = mid; else hi = mid - 1;
            }
            s = lo;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < N; ++i) {
            const float al = alpha[(int64_t)s * N + i];
            const float4 t = ld4(tv + (int64_t)i * tv_stride + p);
            acc.x = __fadd_rn(acc.x, __fmul_rn(al, t.x));
            acc.y = __fadd_rn(acc.y, __fmul_rn(al, t.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(al, t.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(al, t.w));
        }
        const float4 b = ld4(base + p);
        float4 o;
        o.x = __fadd_rn(b.x, acc.x);
        o.y = __fadd_rn(b.y, acc.y);
        o.z = __fadd_rn(b.z, acc.z);
        o.w = __fadd_rn(b.w, acc.w);
        *reinterpret_cast<float4*>(out + p) = o;
    }
}

__global__ __launch_bounds__(kThreads) void task_vector_kernel(const float* __restrict__ theta,
                                                              const float* __restrict__ base, int64_t n,
                                                              float* __restrict__ tv) {
    const int64_t nvec = n >> 2;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kThreads) {
        const float4 a = ld4(theta + v * 4), b = ld4(base + v * 4);
        float4 o;
        o.x = __fsub_rn(a.x, b.x);
        o.y = __fsub_rn(a.y, b.y);
        o.z = __fsub_rn(a.z, b.z);
        o.w = __fsub_rn(a.w, b.w);
        *reinterpret_cast<float4*>(tv + v * 4) = o;
    }
    const int64_t tail = nvec * 4 + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (tail < n) tv[tail] = __fsub_rn(theta[tail], base[tail]);
}

// ---- K6: dalpha[s, i] = <tv_i[seg s], g[seg s]> ------------------------------------------------
// Stage 1: fixed chunks of kBwdChunk elements that never cross a segment boundary relative to the
// segment start; block (chunk c of segment s) writes N partial sums.  Stage 2: one wave per (s, i)
// adds that segment's partials in chunk order.  Both orders are fixed => bitwise reproducible.
constexpr int kBwdChunk = 16384;

__global__ __launch_bounds__(kThreads) void merge_bwd_stage1(const float* __restrict__ tv, int64_t tv_stride,
                                                            const float* __restrict__ g,
                                                            const int64_t* __restrict__ seg_off, int N, int S,
                                                            int64_t P, const int64_t* __restrict__ chunk_first,
                                                            float* __restrict__ partial) {
    // chunk_first[s] = index of the first chunk of segment s (S+1 entries), computed by the prologue kernel
    const int64_t c = blockIdx.x;
    int lo = 0, hi = S - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunk_first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int s = lo;
    const int64_t seg_b = seg_off ? seg_off[s] : 0, seg_e = seg_off ? seg_off[s + 1] : P;
    const int64_t b = seg_b + (c - chunk_first[s]) * kBwdChunk;
    int64_t e = b + kBwdChunk;
    if (e > seg_e) e = seg_e;
    __shared__ float red[kThreads / MR_WAVE];
    for (int i = 0; i < N; ++i) {
        const float* t = tv + (int64_t)i * tv_stride;
        float acc = 0.f;
        for (int64_t p = b + (int64_t)threadIdx.x * 4; p < e; p += (int64_t)kThreads * 4) {
            const float4 x = ld4(t + p), y = ld4(g + p);
            acc = fmaf(x.x, y.x, acc);
            acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc);
            acc = fmaf(x.w, y.w, acc);
        }
        acc = mr::wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[c * N + i] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// The same sums for N <= 8 with every stream of a chunk in flight at once: one pass over the chunk, g read once, N running sums per thread
// (each the SAME fmaf chain over the same elements in the same order as the loop above, and the same wave / workgroup combine -- results are
// bit-identical), one barrier instead of 2 N.  The per-vector loop above leaves only four 16-byte loads per thread in flight between barriers:
// 4.1 TB/s at N = 8 against the forward merge's 5.3 TB/s over the same streams.
template <int NN>
__global__ __launch_bounds__(kThreads) void merge_bwd_stage1_n(const float* __restrict__ tv, int64_t tv_stride, const float* __restrict__ g,
                                                              const int64_t* __restrict__ seg_off, int S, int64_t P,
                                                              const int64_t* __restrict__ chunk_first, float* __restrict__ partial) {
    const int64_t c = blockIdx.x;
    int lo = 0, hi = S - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunk_first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    const int s = lo;
    const int64_t seg_b = seg_off ? seg_off[s] : 0, seg_e = seg_off ? seg_off[s + 1] : P;
    const int64_t b = seg_b + (c - chunk_first[s]) * kBwdChunk;
    int64_t e = b + kBwdChunk;
    if (e > seg_e) e = seg_e;
    __shared__ float red[NN][kThreads / MR_WAVE];
    float acc[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) acc[i] = 0.f;
#pragma unroll
    for (int it = 0; it < kBwdChunk / (kThreads * 4); ++it) {
        const int64_t p = b + (int64_t)threadIdx.x * 4 + (int64_t)it * kThreads * 4;
        if (p < e) {
            const float4 y = ld4(g + p);
            float4 x[NN];
#pragma unroll
            for (int i = 0; i < NN; ++i) x[i] = ld4(tv + (int64_t)i * tv_stride + p);
#pragma unroll
            for (int i = 0; i < NN; ++i) {
                acc[i] = fmaf(x[i].x, y.x, acc[i]);
                acc[i] = fmaf(x[i].y, y.y, acc[i]);
                acc[i] = fmaf(x[i].z, y.z, acc[i]);
                acc[i] = fmaf(x[i].w, y.w, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        const float w = mr::wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < NN) partial[c * NN + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void merge_bwd_prologue(const int64_t* __restrict__ seg_off, int S, int64_t P,
                                   int64_t* __restrict__ chunk_first) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t c = 0;
        for (int s = 0; s < S; ++s) {
            chunk_first[s] = c;
            const int64_t len = seg_off ? seg_off[s + 1] - seg_off[s] : P;
            c += (len + kBwdChunk - 1) / kBwdChunk;
        }
        chunk_first[S] = c;
    }
}

__global__ __launch_bounds__(MR_WAVE) void merge_bwd_stage2(const float* __restrict__ partial,
                                                           const int64_t* __restrict__ chunk_first, int N,
                                                           float* __restrict__ dalpha) {
    const int s = blockIdx.x / N, i = blockIdx.x % N;
    const int64_t c0 = chunk_first[s], c1 = chunk_first[s + 1];
    // fixed order: lane l sums chunks c0+l, c0+l+64, ... then a fixed xor-tree across lanes
    double acc = 0.0;
    for (int64_t c = c0 + threadIdx.x; c < c1; c += MR_WAVE) acc += (double)partial[c * N + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) dalpha[s * N + i] = (float)acc;
}

// ModelMerger.merge("task_vector" | "linear") (merger.py:46-93): a running sum in model order, every operation rounded on its own:
//   task_vector: acc = base;  acc = acc + w_i * (theta_i - base)      (algorithms/task_vector.py:30-32)
//   linear     : acc = 0;     acc = acc + w_i * theta_i               (algorithms/linear.py:23-25)
// (the learnable-alpha module sums the products first and adds base last: 1-ulp differences)
template <bool TASK_VECTOR>
__global__ __launch_bounds__(kThreads) void merge_running_kernel(const float* __restrict__ base, const float* __restrict__ models,
                                                                int64_t stride, const float* __restrict__ w, int N, int64_t nvec,
                                                                float* __restrict__ out) {
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kThreads) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TASK_VECTOR) b = ld4(base + v * 4);
        float4 acc = b;
        for (int i = 0; i < N; ++i) {
            const float4 m = ld4_nt(models + (int64_t)i * stride + v * 4);
            const float wi = w[i];
            if (TASK_VECTOR) {
                acc.x = __fadd_rn(acc.x, __fmul_rn(wi, __fsub_rn(m.x, b.x)));
                acc.y = __fadd_rn(acc.y, __fmul_rn(wi, __fsub_rn(m.y, b.y)));
                acc.z = __fadd_rn(acc.z, __fmul_rn(wi, __fsub_rn(m.z, b.z)));
                acc.w = __fadd_rn(acc.w, __fmul_rn(wi, __fsub_rn(m.w, b.w)));
            } else {
                acc.x = __fadd_rn(acc.x, __fmul_rn(wi, m.x));
                acc.y = __fadd_rn(acc.y, __fmul_rn(wi, m.y));
                acc.z = __fadd_rn(acc.z, __fmul_rn(wi, m.z));
                acc.w = __fadd_rn(acc.w, __fmul_rn(wi, m.w));
            }
        }
        *reinterpret_cast<float4*>(out + v * 4) = acc;
    }
}

int64_t host_chunk_upper_bound(int S, int64_t P) { return (P + kBwdChunk - 1) / kBwdChunk + S; }

}  // namespace

extern "C" int mr_task_vector_f32(const float* theta, const float* base, int64_t n, float* tv, mr_stream_t stream) {
    if (!theta || !base || !tv || n < 0) return MR_EINVAL;
    if (!mr::aligned16(theta) || !mr::aligned16(base) || !mr::aligned16(tv)) return MR_EALIGN;
    if (n == 0) return MR_OK;
    int64_t blocks = (n / 4 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(task_vector_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, theta, base, n, tv);
    return mr::check_launch();
}

extern "C" int mr_merge_running_f32(const float* base, const float* models, int64_t stride, const float* weights, int N, int64_t P,
                                    float* out, mr_stream_t stream) {
    if (!models || !weights || !out || N < 1 || P < 0) return MR_EINVAL;
    if ((P & 3) || (stride & 3)) return MR_EALIGN;
    if (!mr::aligned16(models) || !mr::aligned16(out) || (base && !mr::aligned16(base))) return MR_EALIGN;
    if (P == 0) return MR_OK;
    int64_t blocks = (P / 4 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (base)
        hipLaunchKernelGGL((merge_running_kernel<true>), dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, base, models, stride, weights, N, P / 4, out);
    else
        hipLaunchKernelGGL((merge_running_kernel<false>), dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, base, models, stride, weights, N, P / 4, out);
    return mr::check_launch();
}

template <int N>
static void launch_merge(bool seg, unsigned blocks, hipStream_t st, const float* base, const float* tv, int64_t stride,
                         const float* alpha, const int64_t* seg_off, int S, int64_t pb, int64_t pc, float* out) {
    if (seg)
        hipLaunchKernelGGL((merge_nway_kernel<N, true>), dim3(blocks), dim3(kThreads), 0, st, base, tv, stride, alpha, seg_off, S, pb, pc, out);
    else
        hipLaunchKernelGGL((merge_nway_kernel<N, false>), dim3(blocks), dim3(kThreads), 0, st, base, tv, stride, alpha, seg_off, S, pb, pc, out);
}

extern "C" int mr_merge_nway_f32(const float* base, const float* tv, int64_t tv_stride, const float* alpha,
                                 const int64_t* seg_off, int N, int S, int64_t p_begin, int64_t p_count, float* out,
                                 mr_stream_t stream) {
    if (!base || !tv || !alpha || !out || N < 1 || S < 1 || p_begin < 0 || p_count < 0) return MR_EINVAL;
    if (S > 1 && !seg_off) return MR_EINVAL;
    if ((p_begin & 3) || (p_count & 3) || (tv_stride & 3)) return MR_EALIGN;
    if (!mr::aligned16(base) || !mr::aligned16(tv) || !mr::aligned16(out)) return MR_EALIGN;
    if (p_count == 0) return MR_OK;
    const bool seg = (S > 1);
    const int64_t nvec = p_count / 4;
    int64_t blocks = (nvec + kThreads * kVPT - 1) / (kThreads * kVPT);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = (unsigned)blocks;
    switch (N) {
        case 1: launch_merge<1>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 2: launch_merge<2>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 3: launch_merge<3>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 4: launch_merge<4>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 5: launch_merge<5>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 6: launch_merge<6>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 7: launch_merge<7>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        case 8: launch_merge<8>(seg, nb, st, base, tv, tv_stride, alpha, seg_off, S, p_begin, p_count, out); break;
        default:
            if (seg)
                hipLaunchKernelGGL((merge_nway_generic_kernel<true>), dim3(nb), dim3(kThreads), 0, st, base, tv, tv_stride, alpha, seg_off, N, S, p_begin, p_count, out);
            else
                hipLaunchKernelGGL((merge_nway_generic_kernel<false>), dim3(nb), dim3(kThreads), 0, st, base, tv, tv_stride, alpha, seg_off, N, S, p_begin, p_count, out);
    }
    return mr::check_launch();
}

extern "C" int mr_merge_rows_f32(const float* base, const float* tv, int64_t tv_stride, const float* alpha, int N, const int32_t* idx, int T,
                                 int rows, int d, int64_t table_off, float* out, mr_stream_t stream) {
    if (!base || !tv || !alpha || !out || N < 1 || T < 0 || rows < 1 || d < 4 || table_off < 0 || (T > 0 && !idx)) return MR_EINVAL;
    if ((d & 3) || (table_off & 3) || (tv_stride & 3) || !mr::aligned16(base) || !mr::aligned16(tv) || !mr::aligned16(out)) return MR_EALIGN;
    if (T == 0) return MR_OK;
    hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)((T + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, (hipStream_t)stream, base, tv,
                       tv_stride, alpha, N, idx, T, rows, d, table_off, out);
    return mr::check_launch();
}

extern "C" size_t mr_merge_bwd_alpha_ws_bytes(int N, int S, int64_t P) {
    if (N < 1 || S < 1 || P < 0) return 0;
    const int64_t nchunk = host_chunk_upper_bound(S, P);
    return (size_t)(S + 1) * sizeof(int64_t) + 64 + (size_t)nchunk * N * sizeof(float);
}

// A/B and the bit-identity test: 1 = the per-vector loop for every N.  Initial value from MR_MERGE_BWD_GENERIC, read ONCE at load;
// mr_merge_bwd_generic() changes it at run time (tests, tools/merge_bwd_bench.py) and returns the previous value.
static int g_bwd_generic = [] { const char* e = getenv("MR_MERGE_BWD_GENERIC"); return (e && e[0] == '1') ? 1 : 0; }();
extern "C" int mr_merge_bwd_generic(int on) {
    const int old = g_bwd_generic;
    if (on == 0 || on == 1) g_bwd_generic = on;
    return old;
}

extern "C" int mr_merge_bwd_alpha_f32(const float* tv, int64_t tv_stride, const float* g, const int64_t* seg_off, int N,
                                      int S, int64_t P, float* dalpha, void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (!tv || !g || !dalpha || !ws || N < 1 || S < 1 || P < 0) return MR_EINVAL;
    if (S > 1 && !seg_off) return MR_EINVAL;
    if ((P & 3) || (tv_stride & 3) || !mr::aligned16(tv) || !mr::aligned16(g) || !mr::aligned16(ws)) return MR_EALIGN;
    if (ws_bytes < mr_merge_bwd_alpha_ws_bytes(N, S, P)) return MR_EWS;
    hipStream_t st = (hipStream_t)stream;
    int64_t* chunk_first = reinterpret_cast<int64_t*>(ws);
    size_t off = ((size_t)(S + 1) * sizeof(int64_t) + 63) & ~(size_t)63;
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + off);
    const int64_t nchunk = host_chunk_upper_bound(S, P);  // grid upper bound; surplus blocks map to the last segment's empty tail
    hipLaunchKernelGGL(merge_bwd_prologue, dim3(1), dim3(64), 0, st, seg_off, S, P, chunk_first);
    // exact chunk count is only known on device; launch the upper bound and let surplus blocks write zeros
#define MR_BWD1(NN_) hipLaunchKernelGGL(merge_bwd_stage1_n<NN_>, dim3((unsigned)nchunk), dim3(kThreads), 0, st, tv, tv_stride, g, seg_off, S, P, chunk_first, partial)
    switch (g_bwd_generic ? 0 : N) {
        case 1: MR_BWD1(1); break;
        case 2: MR_BWD1(2); break;
        case 3: MR_BWD1(3); break;
        case 4: MR_BWD1(4); break;
        case 5: MR_BWD1(5); break;
        case 6: MR_BWD1(6); break;
        case 7: MR_BWD1(7); break;
        case 8: MR_BWD1(8); break;
        default:
            hipLaunchKernelGGL(merge_bwd_stage1, dim3((unsigned)nchunk), dim3(kThreads), 0, st, tv, tv_stride, g, seg_off, N, S, P, chunk_first, partial);
    }
#undef MR_BWD1
    hipLaunchKernelGGL(merge_bwd_stage2, dim3((unsigned)(S * N)), dim3(MR_WAVE), 0, st, partial, chunk_first, N, dalpha);
    return mr::check_launch();
}
