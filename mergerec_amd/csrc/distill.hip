// K7: fused distillation row losses (next-row 2 of the scope table).
//
// The reference (module/recommender/loss_fn.py) evaluates one of eleven small torch expressions per sample over the
// (num_items,) logit rows "merged model" z and "single model" t, and averages over the batch.  All of them are a weighted
// sum of six per-row terms, which this kernel computes in one launch, one workgroup per row, together with d loss / d z:
//     CE(z, label)            label = argmax t (teacher pseudo-label) or argmax z (merged pseudo-label)   loss_fn.py:40-44,95-104,135-142
//     T^2 KL(softmax(t/T) || softmax(z/T))                                                                loss_fn.py:52-60
//     H(softmax(z)) with log(p + 1e-8)                                                                    loss_fn.py:64-69
//     mean_j (z_j - t_j)^2                                                                                loss_fn.py:171-175
//     relu(margin - (z[pos] - z[neg])), pos / neg = best / second best of t                               loss_fn.py:183-199
//     -sum_j softmax(t/T)_j log_softmax(z/T)_j                                                            loss_fn.py:208-215
// Rows are at most a few hundred KB (M = catalog size), so the passes after the first are L2 hits; the kernel is bound by
// launch latency at the reference's batch of 16 rows and by HBM (2 rows read once, 1 gradient row written) beyond that.
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 1024;  // one workgroup per row: at the reference's 16 rows per step the row length sets the latency

struct ArgMax {
    float v;
    int i;
};
// larger value wins; ties go to the lower index (torch.argmax on CPU returns the first maximum)
__device__ __forceinline__ ArgMax amax(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

__device__ __forceinline__ ArgMax block_argmax(ArgMax x, ArgMax* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = amax(x, y);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = x;
    __syncthreads();
    ArgMax r = sh[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) r = amax(r, sh[w]);
    return r;
}

template <int K>
__device__ __forceinline__ void block_sum(float (&x)[K], float* sh) {
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = mr::wave_sum(x[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[wave * K + k] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = sh[k];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) s += sh[w * K + k];
        x[k] = s;
    }
}

__global__ __launch_bounds__(kThreads) void distill_rows_kernel(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ Tt,
                                                               int64_t ldt, int M, int label_src, float w_ce, float w_kd, float temp,
                                                               float w_ent, float w_mse, float w_pair, float margin, float w_ln,
                                                               float* __restrict__ loss_row, float* __restrict__ dZ, int64_t lddz,
                                                               float grad_scale, const int32_t* __restrict__ row_M) {
    __shared__ ArgMax sh_am[kThreads / 64];
    __shared__ float sh_f[(kThreads / 64) * 8];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row_M) M = row_M[row];  // rows of different catalogs in one launch (the step's 16 samples over 8 domains): uniform per workgroup
    const float* __restrict__ z = Z + (int64_t)row * ldz;
    const float* __restrict__ t = Tt + (int64_t)row * ldt;
    const bool need_t = (label_src == 1) || w_kd != 0.f || w_mse != 0.f || w_pair != 0.f || w_ln != 0.f;

    // ---- pass 1: maxima and arg-maxima
    ArgMax az = {-INFINITY, 0x7fffffff}, at = {-INFINITY, 0x7fffffff};
    for (int j = tid; j < M; j += kThreads) {
        az = amax(az, ArgMax{z[j], j});
        if (need_t) at = amax(at, ArgMax{t[j], j});
    }
    az = block_argmax(az, sh_am);
    if (need_t) at = block_argmax(at, sh_am);
    const int label = label_src == 1 ? at.i : az.i;
    const int pos = at.i;
    const float mz = az.v, mt = at.v;
    const float mzT = mz / temp, mtT = mt / temp;  // max of (x / T) = max(x) / T: the division is monotone

    // ---- pass 2: partition sums, squared error, second best of t
    float s[4] = {0.f, 0.f, 0.f, 0.f};  // sum exp(z - mz), sum exp(z/T - mz/T), sum exp(t/T - mt/T), sum (z - t)^2
    ArgMax an = {-INFINITY, 0x7fffffff};
    const bool need_T = w_kd != 0.f || w_ln != 0.f;
    for (int j = tid; j < M; j += kThreads) {
        const float zj = z[j];
        s[0] += __expf(zj - mz);
        if (need_T) s[1] += __expf(zj / temp - mzT);
        if (need_t) {
            const float tj = t[j];
            if (need_T) s[2] += __expf(tj / temp - mtT);
            const float d = zj - tj;
            s[3] += d * d;
            if (j != pos) an = amax(an, ArgMax{tj, j});
        }
    }
    block_sum<4>(s, sh_f);
    int neg = 0;
    if (w_pair != 0.f) {
        neg = block_argmax(an, sh_am).i;
        if (neg == 0x7fffffff) neg = 0;  // M == 1: torch.argmax of an all -inf row is 0
    }
    const float S1 = s[0], SzT = s[1], StT = s[2];
    const float lse1 = __logf(S1), lseZ = __logf(SzT), lseT = __logf(StT);

    // ---- pass 3: KL, ListNet cross entropy, entropy (and the entropy gradient's mean term)
    float a[4] = {0.f, 0.f, 0.f, 0.f};  // sum p (log p - log q), sum p log q, sum p1 log(p1 + eps), sum p1 g
    const bool need_ent = w_ent != 0.f;
    if (need_T || need_ent) {
        for (int j = tid; j < M; j += kThreads) {
            const float zj = z[j];
            if (need_T) {
                const float lq = (zj / temp - mzT) - lseZ;
                const float p = __expf(t[j] / temp - mtT) / StT;
                // torch: target * (log(target) - input), with 0 where target == 0 (xlogy)
                if (p > 0.f) a[0] += p * (__logf(p) - lq);
                a[1] += p * lq;
            }
            if (need_ent) {
                const float p1 = __expf(zj - mz) / S1;
                const float lg = __logf(p1 + 1e-8f);
                a[2] += p1 * lg;
                a[3] += p1 * (-(lg + p1 / (p1 + 1e-8f)));
            }
        }
        block_sum<4>(a, sh_f);
    }

    const float zl = z[label];
    const float ce = -((zl - mz) - lse1);
    const float kd = a[0] * (temp * temp);
    const float ent = -a[2];
    const float mse = s[3] / (float)M;
    const float ln = -a[1];
    float pair = 0.f;
    bool pair_on = false;
    if (w_pair != 0.f) {
        const float h = margin - (z[pos] - z[neg]);
        pair_on = h > 0.f;
        pair = pair_on ? h : 0.f;
    }
    if (tid == 0) {
        float L = 0.f;
        if (w_ce != 0.f) L += w_ce * ce;
        if (w_kd != 0.f) L += w_kd * kd;
        if (w_ent != 0.f) L += w_ent * ent;
        if (w_mse != 0.f) L += w_mse * mse;
        if (w_pair != 0.f) L += w_pair * pair;
        if (w_ln != 0.f) L += w_ln * ln;
        loss_row[row] = L;
    }

    // ---- pass 4: d (row loss) / d z, scaled by grad_scale (= upstream gradient / rows)
    if (dZ) {
        float* __restrict__ dz = dZ + (int64_t)row * lddz;
        const float G = a[3];
        for (int j = tid; j < M; j += kThreads) {
            const float zj = z[j];
            float g = 0.f;
            float p1 = 0.f;
            if (w_ce != 0.f || need_ent) p1 = __expf(zj - mz) / S1;
            if (w_ce != 0.f) g += w_ce * (p1 - (j == label ? 1.f : 0.f));
            if (need_T) {
                const float q = __expf(zj / temp - mzT) / SzT;
                const float p = __expf(t[j] / temp - mtT) / StT;
                g += (w_kd * temp + w_ln / temp) * (q - p);  // T^2 * (1/T) (q - p)  and  (1/T) (q - p)
            }
            if (need_ent) {
                const float gj = -(__logf(p1 + 1e-8f) + p1 / (p1 + 1e-8f));
                g += w_ent * p1 * (gj - G);
            }
            if (w_mse != 0.f) g += w_mse * 2.f * (zj - t[j]) / (float)M;
            if (pair_on) g += w_pair * ((j == neg ? 1.f : 0.f) - (j == pos ? 1.f : 0.f));
            dz[j] = g * grad_scale;
        }
    }
}

}  // namespace

static int distill_rows_launch(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M, const int32_t* row_M, int label_src,
                               float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin, float w_listnet,
                               float* loss_row, float* dz, int64_t lddz, float grad_scale, mr_stream_t stream);

extern "C" int mr_distill_loss_rows_f32(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M, int label_src,
                                        float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin,
                                        float w_listnet, float* loss_row, float* dz, int64_t lddz, float grad_scale,
                                        mr_stream_t stream) {
    return distill_rows_launch(z, ldz, t, ldt, rows, M, nullptr, label_src, w_ce, w_kd, temperature, w_ent, w_mse, w_pair, margin, w_listnet, loss_row,
                               dz, lddz, grad_scale, stream);
}

// the same with a per-row length: row r holds row_M[r] <= M_max logits (device int32; every entry in [1, M_max]) -- one launch for the rows of
// several catalogs (a step's samples spread over the domains: 8 launches of 2 rows were 8 x 51 us of pure latency)
extern "C" int mr_distill_loss_rows_var_f32(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M_max,
                                            const int32_t* row_M, int label_src, float w_ce, float w_kd, float temperature, float w_ent, float w_mse,
                                            float w_pair, float margin, float w_listnet, float* loss_row, float* dz, int64_t lddz,
                                            float grad_scale, mr_stream_t stream) {
    if (!row_M) return MR_EINVAL;
    return distill_rows_launch(z, ldz, t, ldt, rows, M_max, row_M, label_src, w_ce, w_kd, temperature, w_ent, w_mse, w_pair, margin, w_listnet,
                               loss_row, dz, lddz, grad_scale, stream);
}

static int distill_rows_launch(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M, const int32_t* row_M, int label_src,
                               float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin, float w_listnet,
                               float* loss_row, float* dz, int64_t lddz, float grad_scale, mr_stream_t stream) {
    if (!z || !loss_row || rows < 0 || M < 1 || ldz < M || (dz && lddz < M)) return MR_EINVAL;
    if (label_src < 0 || label_src > 2) return MR_EINVAL;
    const bool need_t = (label_src == 1) || w_kd != 0.f || w_mse != 0.f || w_pair != 0.f || w_listnet != 0.f;
    if (need_t && (!t || ldt < M)) return MR_EINVAL;
    if ((w_kd != 0.f || w_listnet != 0.f) && !(temperature > 0.f)) return MR_EINVAL;
    if (w_ce != 0.f && label_src == 0) return MR_EINVAL;
    if (rows > 0x7fffffff || M > 0x7fffffff) return MR_EUNSUPPORTED;
    if (rows == 0) return MR_OK;
    hipLaunchKernelGGL(distill_rows_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, z, ldz, t ? t : z, ldt, (int)M,
                       label_src, w_ce, w_kd, temperature > 0.f ? temperature : 1.f, w_ent, w_mse, w_pair, margin, w_listnet, loss_row, dz,
                       lddz, grad_scale, row_M);
    return mr::check_launch();
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Skinny scoring of the distillation step (module/distiller/sequence/module.py:62-72: ``rep_i @ E_ds.T`` per sample): a HANDFUL of
// representation rows (n <= 8 per launch) against a whole catalog (M ~ 20 k rows of d floats).  The 128 x 128 MFMA tile kernel spent
// 104 us per 2-row group on it (one row tile, 179 column tiles, a full k loop each); the job is one stream over E: every wave takes catalog
// rows, holds the n representation rows in registers and emits n dot products per catalog row -- HBM-bound (70 MB per Arts-sized catalog).
// Backward: d rep_i = sum_m dz[i][m] E[m][:], the same stream with the roles swapped; per-workgroup partial sums, then a fixed-order sum.
namespace {

constexpr int kSkThreads = 256;
constexpr int kSkRowsPerWg = 128;  // catalog rows per workgroup (32 per wave)

template <int NR, int NP>   // NR = representation rows held (1, 2, 4, 8); NP = float4 per lane and row: d <= 256 * NP
__global__ __launch_bounds__(kSkThreads) void skinny_scores_kernel(const float* __restrict__ reps, int64_t ldr, int n, const float* __restrict__ E,
                                                                  int64_t lde, int M, int d, float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 r[NR][NP];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = 4 * (p * 64 + lane);
            r[i][p] = (i < n && c < d) ? *reinterpret_cast<const float4*>(reps + (int64_t)i * ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int m_end = min(M, (int)(blockIdx.x + 1) * kSkRowsPerWg);
    for (int m = blockIdx.x * kSkRowsPerWg + wave; m < m_end; m += kSkThreads / 64) {
        float4 e[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = 4 * (p * 64 + lane);
            e[p] = c < d ? *reinterpret_cast<const float4*>(E + (int64_t)m * lde + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float acc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float a = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p) a += r[i][p].x * e[p].x + r[i][p].y * e[p].y + r[i][p].z * e[p].z + r[i][p].w * e[p].w;
            acc[i] = mr::wave_sum(a);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (i < n) out[(int64_t)i * ldo + m] = acc[i];
        }
    }
}

template <int NR, int NP>
__global__ __launch_bounds__(kSkThreads) void skinny_bwd_kernel(const float* __restrict__ dz, int64_t lddz, int n, const float* __restrict__ E,
                                                               int64_t lde, int M, int d, float* __restrict__ part) {
    __shared__ float4 red[kSkThreads / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc[NR][NP];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[i][p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int m_end = min(M, (int)(blockIdx.x + 1) * kSkRowsPerWg);
    for (int m = blockIdx.x * kSkRowsPerWg + wave; m < m_end; m += kSkThreads / 64) {
        float g[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) g[i] = i < n ? dz[(int64_t)i * lddz + m] : 0.f;   // wave-uniform addresses
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = 4 * (p * 64 + lane);
            if (c < d) {
                const float4 e = *reinterpret_cast<const float4*>(E + (int64_t)m * lde + c);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    acc[i][p].x += g[i] * e.x; acc[i][p].y += g[i] * e.y; acc[i][p].z += g[i] * e.z; acc[i][p].w += g[i] * e.w;
                }
            }
        }
    }
    // the four waves' sums, wave 0 .. 3 in order, into part[blockIdx][i][:]
    float* __restrict__ dst = part + (int64_t)blockIdx.x * n * d;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            __syncthreads();
            red[wave][lane] = acc[i][p];
            __syncthreads();
            const int c = 4 * (p * 64 + lane);
            if (wave == 0 && i < n && c < d) {
                float4 s = red[0][lane];
#pragma unroll
                for (int w = 1; w < kSkThreads / 64; ++w) { s.x += red[w][lane].x; s.y += red[w][lane].y; s.z += red[w][lane].z; s.w += red[w][lane].w; }
                *reinterpret_cast<float4*>(dst + (int64_t)i * d + c) = s;
            }
        }
}

// out[e] = scale * sum_w part[w][e]: 64 elements per workgroup, the partials dealt over 4 threads per element (w = q mod 4, ascending) whose
// four sums are added q = 0..3 -- a fixed order, and a quarter of the dependent loads of one thread walking every partial
__global__ __launch_bounds__(256) void skinny_bwd_sum_kernel(const float* __restrict__ part, int nwg, int nd, float scale, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;
    float s = 0.f;
    if (e < nd)
        for (int w = q; w < nwg; w += 4) s += part[(int64_t)w * nd + e];
    sh[q][l] = s;
    __syncthreads();
    if (q == 0 && e < nd) out[e] = (((sh[0][l] + sh[1][l]) + sh[2][l]) + sh[3][l]) * scale;
}

}  // namespace

extern "C" int mr_skinny_scores_f32(const float* reps, int64_t ldr, int n, const float* E, int64_t lde, int64_t M, int d, float* out, int64_t ldo,
                                    mr_stream_t stream) {
    if (!reps || !E || !out || n < 0 || M < 0 || d < 1 || ldr < d || lde < d || ldo < M) return MR_EINVAL;
    if (n > 8 || d > 1024 || M > 0x7fffffff) return MR_EUNSUPPORTED;
    if ((d & 3) || (ldr & 3) || (lde & 3) || !mr::aligned16(reps) || !mr::aligned16(E)) return MR_EALIGN;
    if (n == 0 || M == 0) return MR_OK;
    const dim3 grid((unsigned)((M + kSkRowsPerWg - 1) / kSkRowsPerWg));
    hipStream_t st = (hipStream_t)stream;
    const int np = (d + 255) / 256;
#define MR_SK(NR_, NP_) hipLaunchKernelGGL((skinny_scores_kernel<NR_, NP_>), grid, dim3(kSkThreads), 0, st, reps, ldr, n, E, lde, (int)M, d, out, ldo)
#define MR_SK2(NR_) do { if (np == 1) MR_SK(NR_, 1); else if (np == 2) MR_SK(NR_, 2); else if (np == 3) MR_SK(NR_, 3); else MR_SK(NR_, 4); } while (0)
    if (n <= 2) MR_SK2(2); else if (n <= 4) MR_SK2(4); else MR_SK2(8);
#undef MR_SK2
#undef MR_SK
    return mr::check_launch();
}

extern "C" size_t mr_skinny_bwd_ws_bytes(int n, int64_t M, int d) {
    if (n < 1 || M < 1 || d < 1) return 0;
    return (size_t)((M + kSkRowsPerWg - 1) / kSkRowsPerWg) * n * d * sizeof(float);
}

// d_reps[i][:] = scale * sum_m dz[i][m] E[m][:]
extern "C" int mr_skinny_bwd_f32(const float* dz, int64_t lddz, int n, const float* E, int64_t lde, int64_t M, int d, float scale, float* d_reps,
                                 void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (!dz || !E || !d_reps || n < 0 || M < 0 || d < 1 || lddz < M || lde < d) return MR_EINVAL;
    if (n > 8 || d > 1024 || M > 0x7fffffff) return MR_EUNSUPPORTED;
    if ((d & 3) || (lde & 3) || !mr::aligned16(E) || !mr::aligned16(ws)) return MR_EALIGN;
    if (n == 0) return MR_OK;
    if (M == 0) return hipMemsetAsync(d_reps, 0, (size_t)n * d * 4, (hipStream_t)stream) == hipSuccess ? MR_OK : MR_ELAUNCH;
    if (!ws || ws_bytes < mr_skinny_bwd_ws_bytes(n, M, d)) return MR_EWS;
    const int nwg = (int)((M + kSkRowsPerWg - 1) / kSkRowsPerWg);
    hipStream_t st = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(ws);
    const int np = (d + 255) / 256;
#define MR_SB(NR_, NP_) hipLaunchKernelGGL((skinny_bwd_kernel<NR_, NP_>), dim3(nwg), dim3(kSkThreads), 0, st, dz, lddz, n, E, lde, (int)M, d, part)
#define MR_SB2(NR_) do { if (np == 1) MR_SB(NR_, 1); else if (np == 2) MR_SB(NR_, 2); else if (np == 3) MR_SB(NR_, 3); else MR_SB(NR_, 4); } while (0)
    if (n <= 2) MR_SB2(2); else if (n <= 4) MR_SB2(4); else MR_SB2(8);
#undef MR_SB2
#undef MR_SB
    hipLaunchKernelGGL(skinny_bwd_sum_kernel, dim3((n * d + 63) / 64), dim3(256), 0, st, part, nwg, n * d, scale, d_reps);
    return mr::check_launch();
}
