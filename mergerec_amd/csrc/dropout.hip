// Hidden-state dropout of the training graph: out = dropout(x) (+ residual), mask = mr::dropout_keep (dropout.h).  One launch serves the
// forward sites (embedding LayerNorm output; the attention-output and FFN-output dense results, fused with their residual adds) and
// the backward of the same sites (dX = dropout-mask(dY): the same call with the same key and no residual).
#include "common.h"
#include "dropout.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void dropout_rows_kernel(const float* __restrict__ x, int64_t ldx, int T, int d, uint32_t thresh,
                                                               float inv_keep, uint32_t key, const float* __restrict__ res, int64_t ldr,
                                                               float* __restrict__ out, int64_t ldo) {
    const int64_t n4 = (int64_t)T * (d >> 2);
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < n4; e += (int64_t)gridDim.x * kThreads) {
        const int t = (int)(e / (d >> 2)), c = (int)(e - (int64_t)t * (d >> 2)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)t * ldx + c);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4*>(res + (int64_t)t * ldr + c);
        float4 o;
        o.x = (mr::dropout_keep(key, (uint32_t)t, (uint32_t)c, thresh) ? v.x * inv_keep : 0.f) + r.x;
        o.y = (mr::dropout_keep(key, (uint32_t)t, (uint32_t)c + 1u, thresh) ? v.y * inv_keep : 0.f) + r.y;
        o.z = (mr::dropout_keep(key, (uint32_t)t, (uint32_t)c + 2u, thresh) ? v.z * inv_keep : 0.f) + r.z;
        o.w = (mr::dropout_keep(key, (uint32_t)t, (uint32_t)c + 3u, thresh) ? v.w * inv_keep : 0.f) + r.w;
        *reinterpret_cast<float4*>(out + (int64_t)t * ldo + c) = o;
    }
}

}  // namespace

extern "C" int mr_dropout_rows_f32(const float* x, int64_t ldx, int T, int d, float p, uint32_t key, const float* residual, int64_t ldr,
                                   float* out, int64_t ldo, mr_stream_t stream) {
    if (!x || !out || T < 0 || d < 1 || ldx < d || ldo < d || (residual && ldr < d)) return MR_EINVAL;
    uint32_t thresh;
    float inv_keep;
    if (!mr::dropout_params(p, &thresh, &inv_keep)) return MR_EINVAL;
    if ((d & 3) || (ldx & 3) || (ldo & 3) || (residual && (ldr & 3)) || !mr::aligned16(x) || !mr::aligned16(out) || (residual && !mr::aligned16(residual)))
        return MR_EALIGN;
    if (T == 0) return MR_OK;
    int64_t blocks = ((int64_t)T * (d >> 2) + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, T, d, thresh, inv_keep, key,
                       residual, ldr, out, ldo);
    return mr::check_launch();
}

// the mask itself (tests, and callers that want to inspect a site): keep[t * d + c] = 1 / 0
extern "C" int mr_dropout_site_key(uint32_t seed, uint32_t step, uint32_t layer, uint32_t site, uint32_t* key_out) {
    if (!key_out) return MR_EINVAL;
    *key_out = mr::lowbias32(mr::lowbias32(mr::lowbias32(seed) + step) + layer * 8u + site);
    return MR_OK;
}
