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
                                                               float grad_scale) {
    __shared__ ArgMax sh_am[kThreads / 64];
    __shared__ float sh_f[(kThreads / 64) * 8];
    const int row = blockIdx.x, tid = threadIdx.x;
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

extern "C" int mr_distill_loss_rows_f32(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M, int label_src,
                                        float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin,
                                        float w_listnet, float* loss_row, float* dz, int64_t lddz, float grad_scale,
                                        mr_stream_t stream) {
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
                       lddz, grad_scale);
    return mr::check_launch();
}
