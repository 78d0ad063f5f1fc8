// K8: fused AdamW over the flat parameter arena (fine-tuning, finetune_train.py).
//
// The reference builds torch.optim.AdamW over two parameter groups -- weight decay on matrices, none on names containing
// "bias" / "LayerNorm.weight" (module/recommender/module.py:44-72) -- and Lightning clips the global gradient norm before
// the step (finetune_train.py:106).  Per tensor that is ~8 elementwise launches x 199 tensors; here ONE launch streams the
// arena once: p, g, m, v read, p, m, v written (28 B per parameter), the decay looked up per arena segment, the clip
// coefficient derived in-kernel from a device-resident sum of squares (no host sync between backward and the step).
//
// Arithmetic (torch/optim/adamw.py, single-tensor path; fp32 throughout, scalars prepared in double on the host):
//     g   <- g * min(1, max_norm / (||g|| + 1e-6))                 (torch.nn.utils.clip_grad_norm_)
//     p   <- p * (1 - lr * wd)
//     m   <- m + (1 - beta1) * (g - m)                             (Tensor.lerp_)
//     v   <- beta2 * v + (1 - beta2) * g * g
//     p   <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// HBM-bound: 28 B / parameter.
#include "common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kVPT = 2;  // float4 per thread per chunk

struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps;
    double lr;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float decay, float coef, const AdamScalars& s) {
    g = g * coef;
    p = p * decay;
    m = m + s.one_minus_b1 * (g - m);
    v = __fadd_rn(__fmul_rn(v, s.b2), __fmul_rn(__fmul_rn(g, g), s.one_minus_b2));
    const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
    p = p - s.step_size * (m / denom);
}

__global__ __launch_bounds__(kThreads) void adamw_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M,
                                                        float* __restrict__ V, int64_t n, const int64_t* __restrict__ seg_off,
                                                        const float* __restrict__ seg_wd, int S, float wd_default, AdamScalars sc,
                                                        const float* __restrict__ grad_sumsq, float max_norm) {
    float coef = 1.f;
    if (grad_sumsq) {
        const float c = max_norm / (sqrtf(*grad_sumsq) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const int64_t nvec = n >> 2;
    constexpr int64_t kChunk = (int64_t)kThreads * kVPT;
    const int64_t nchunk = (nvec + kChunk - 1) / kChunk;
    for (int64_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
        const int64_t v0 = chunk * kChunk;
        int s_lo = 0;
        bool uniform = true;
        if (seg_off) {
            const int64_t pf = v0 * 4;
            int64_t pl = (v0 + kChunk) * 4;
            if (pl > n) pl = n;
            int lo = 0, hi = S - 1;  // largest s with seg_off[s] <= pf
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (seg_off[mid] <= pf) lo = mid; else hi = mid - 1;
            }
            s_lo = lo;
            uniform = (pl <= seg_off[s_lo + 1]);
        }
        const float wd_lo = seg_off ? seg_wd[s_lo] : wd_default;
#pragma unroll
        for (int u = 0; u < kVPT; ++u) {
            const int64_t vi = v0 + (int64_t)u * kThreads + threadIdx.x;
            if (vi >= nvec) continue;
            const int64_t e = vi * 4;
            float wd = wd_lo;
            if (!uniform) {  // segment starts are multiples of 4: one float4 never straddles two segments
                int s = s_lo;
                while (s + 1 < S && e >= seg_off[s + 1]) ++s;
                wd = seg_wd[s];
            }
            const float decay = (float)(1.0 - sc.lr * (double)wd);  // python-float arithmetic of `1 - lr * weight_decay`
            float4 p = *reinterpret_cast<const float4*>(P + e);
            const float4 g = *reinterpret_cast<const float4*>(G + e);
            float4 m = *reinterpret_cast<const float4*>(M + e);
            float4 v = *reinterpret_cast<const float4*>(V + e);
            adam1(p.x, g.x, m.x, v.x, decay, coef, sc);
            adam1(p.y, g.y, m.y, v.y, decay, coef, sc);
            adam1(p.z, g.z, m.z, v.z, decay, coef, sc);
            adam1(p.w, g.w, m.w, v.w, decay, coef, sc);
            *reinterpret_cast<float4*>(P + e) = p;
            *reinterpret_cast<float4*>(M + e) = m;
            *reinterpret_cast<float4*>(V + e) = v;
        }
    }
}

}  // namespace

extern "C" int mr_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_off,
                                 const float* seg_wd, int S, double lr, double beta1, double beta2, double eps, double weight_decay,
                                 int64_t step, const float* grad_sumsq, float max_grad_norm, mr_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return MR_EINVAL;
    if ((seg_off == nullptr) != (seg_wd == nullptr) || (seg_off && S < 1)) return MR_EINVAL;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(lr >= 0.0)) return MR_EINVAL;
    if (grad_sumsq && !(max_grad_norm > 0.f)) return MR_EINVAL;
    if ((n & 3) || !mr::aligned16(param) || !mr::aligned16(grad) || !mr::aligned16(exp_avg) || !mr::aligned16(exp_avg_sq)) return MR_EALIGN;
    if (n == 0) return MR_OK;
    // scalars as torch prepares them: python floats (double), cast to the tensors' dtype at the call
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars sc;
    sc.one_minus_b1 = (float)(1.0 - beta1);
    sc.b2 = (float)beta2;
    sc.one_minus_b2 = (float)(1.0 - beta2);
    sc.step_size = (float)(lr / bc1);
    sc.bc2_sqrt = (float)sqrt(bc2);
    sc.eps = (float)eps;
    sc.lr = lr;
    const int64_t nvec = n / 4;
    int64_t blocks = (nvec + kThreads * kVPT - 1) / (kThreads * kVPT);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       seg_off, seg_wd, S, (float)weight_decay, sc, grad_sumsq, max_grad_norm);
    return mr::check_launch();
}
