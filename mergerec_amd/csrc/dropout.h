// Counter-based dropout mask of the training graph (r03).  The reference trains under Lightning's train() mode, i.e. with torch's
// dropout at HF's sites (embeddings after the LayerNorm, attention probabilities, the two output-dense results before their residual
// adds; recformer/models.py:93,135 and transformers' RobertaSelfOutput / RobertaOutput / *SelfAttention).  torch's Philox stream is
// not reproducible outside torch, so the build DEFINES its mask as a pure function of (site key, row, column) -- restated in
// oracle/ref_cpu.py (`dropout_keep`), which is what makes the step testable exactly -- and recomputes it in the backward kernels
// instead of storing it:
//     keep(key, row, col) = lowbias32(row * 0x9E3779B1 + col * 0x85EBCA77 + key) >= thresh,   thresh = floor(p * 2^32)
//     y = keep ? x * (1 / (1 - p)) : 0
// key = the host-side hash of (seed, step, layer, site); row / col per site:
//     hidden sites (embedding LayerNorm output, attention-output dense, FFN-output dense): row = packed token index, col = feature
//     attention probabilities: row = packed query token * H + head, col = key position inside the sequence
//     Longformer global row:   row = sequence * H + head,            col = key position inside the sequence
#pragma once
#include <stdint.h>

namespace mr {

__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ bool dropout_keep(uint32_t key, uint32_t row, uint32_t col, uint32_t thresh) {
    return lowbias32(row * 0x9E3779B1u + col * 0x85EBCA77u + key) >= thresh;
}

// p in [0, 1) -> (thresh, 1 / (1 - p)); p == 0 -> thresh 0 (everything kept, scale 1)
inline bool dropout_params(float p, uint32_t* thresh, float* inv_keep) {
    if (!(p >= 0.f) || p >= 1.f) return false;
    const double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    *inv_keep = 1.0f / (1.0f - p);
    return true;
}

}  // namespace mr
