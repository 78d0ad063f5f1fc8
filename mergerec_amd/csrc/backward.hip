// K8: backward kernels of the BLaIR (RoBERTa) encoder for the collaborative-merging optimisation loop (merge_train.py,
// BASELINE config 5 / scripts/3_mergerec/blair_base_taskvector_taskwise.sh): d loss / d merged parameters, which
// mr_merge_bwd_alpha_f32 reduces to d loss / d alpha.
//
// A training step there is 16 short pseudo-user sequences (item texts, ~40 tokens): a few hundred tokens against 125 M
// parameters, so the step is bound by the parameter-sized streams (merge forward / backward, weight gradients), not by the
// token-sized math.  These kernels are therefore plain fp32 (exact, deterministic except for the embedding scatter):
//   transpose          operand re-layout so every backward product runs on the forward's NT GEMM kernel
//                      (dX = dY W: W^T; dW = dY^T X: dY^T and X^T, the token dimension zero-padded to the GEMM's k-tile)
//   colsum             bias gradients
//   gelu_bwd           d pre-activation of the FFN
//   layernorm_bwd      dx per row (+ saved mean / rstd), dgamma / dbeta by a column pass
//   attn_bwd           softmax attention backward per (sequence, head): row statistics, dQ (query-owned), dK / dV (key-owned)
//   scatter_add_rows   embedding-table gradients (atomic adds), CLS-row scatter
#include "common.h"
#include "dropout.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kDh = 64;

// ------------------------------------------------------------------------------------------------ transpose
// out[c][r] = in[r][c] for r < R, c < C; out rows have ldo >= R_pad columns and columns R .. R_pad - 1 are zero-filled
__global__ __launch_bounds__(kThreads) void transpose_kernel(const float* __restrict__ in, int64_t ldi, int R, int C,
                                                            float* __restrict__ out, int64_t ldo, int R_pad) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < R_pad) out[(int64_t)c * ldo + r] = tile[tx][ty + 8 * k];
    }
}

// the same on 64 x 64 tiles with 16-byte global accesses on both sides (C, ldi, ldo, R_pad multiples of 4, 16-byte aligned bases): a
// thread loads four float4 of the tile's rows, scatters them into a 65-float-pitch LDS tile, gathers four elements of an LDS column
// (conflict-free both ways) and stores them as one float4 of an output row.  The 32 x 32 form above moves 4 bytes per lane and access.
__global__ __launch_bounds__(kThreads) void transpose_vec_kernel(const float* __restrict__ in, int64_t ldi, int R, int C,
                                                                float* __restrict__ out, int64_t ldo, int R_pad) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 float4 columns x 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) v = *reinterpret_cast<const float4*>(in + (int64_t)r * ldi + c);
        float* t = &tile[ty + 16 * k][4 * tx];
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;
        if (c < C && r < R_pad)
            *reinterpret_cast<float4*>(out + (int64_t)c * ldo + r) =
                make_float4(tile[4 * tx][ty + 16 * k], tile[4 * tx + 1][ty + 16 * k], tile[4 * tx + 2][ty + 16 * k], tile[4 * tx + 3][ty + 16 * k]);
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] = sum_r x[r][c].  Workgroup = 32 columns x 8 row lanes over one chunk of kRowsPerChunk rows (rows r = lane, lane + 8, ...),
// partials combined in a fixed order through LDS.  Tall matrices (token-sized: fine-tuning batches) are cut into row chunks whose
// partial sums go to a workspace and are added in chunk order by a second launch: deterministic, and R / 256 x more workgroups.
constexpr int kColW = 32, kRowL = kThreads / kColW, kRowsPerChunk = 256;
static inline int row_chunks(int R) { return R > 0 ? (R + kRowsPerChunk - 1) / kRowsPerChunk : 1; }

__global__ __launch_bounds__(kThreads) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int R, int C, float* __restrict__ out) {
    __shared__ float part[kRowL][kColW];
    const int cl = threadIdx.x % kColW, rl = threadIdx.x / kColW;
    const int c = blockIdx.x * kColW + cl;
    const int r0 = blockIdx.y * kRowsPerChunk, r1 = (r0 + kRowsPerChunk < R) ? r0 + kRowsPerChunk : R;
    float s = 0.f;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += kRowL) s += x[(int64_t)r * ldx + c];
    part[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = part[0][cl];
#pragma unroll
        for (int k = 1; k < kRowL; ++k) t += part[k][cl];
        out[(int64_t)blockIdx.y * C + c] = t;
    }
}

// out[j][c] = sum over chunks of part[j][chunk][c], j < nout.  Workgroup = 32 columns x 8 chunk lanes: lane l adds chunks l, l + 8, ...
// in ascending order, the eight lane sums are combined in lane order through LDS -- a fixed association (deterministic), with 8 x the
// loads in flight of one thread per column walking every chunk (which took 33 us for 71 chunks x 768 columns: three workgroups).
__global__ __launch_bounds__(kThreads) void chunk_reduce_kernel(const float* __restrict__ part, int nchunk, int C, int nout, float* __restrict__ out0,
                                                               float* __restrict__ out1) {
    __shared__ float lds[kRowL][kColW];
    const int cl = threadIdx.x % kColW, rl = threadIdx.x / kColW;
    const int c = blockIdx.x * kColW + cl;
    for (int j = 0; j < nout; ++j) {
        float t = 0.f;
        if (c < C) {
            const float* p = part + (int64_t)j * nchunk * C + c;
            for (int k = rl; k < nchunk; k += kRowL) t += p[(int64_t)k * C];
        }
        lds[rl][cl] = t;
        __syncthreads();
        if (rl == 0 && c < C) {
            float r = lds[0][cl];
#pragma unroll
            for (int k = 1; k < kRowL; ++k) r += lds[k][cl];
            (j == 0 ? out0 : out1)[c] = r;
        }
        __syncthreads();
    }
}

// out[r] = sum_c x[r][c]: one workgroup per row, threads stride the row (coalesced), fixed-order wave + LDS combine.  The bias gradient of
// a Linear is the row sum of dY^T, which the weight-gradient GEMM needs in that layout anyway.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void rowsum_kernel(const float* __restrict__ x, int64_t ldx, int C, float* __restrict__ out) {
    __shared__ float part[kThreads / 64];
    const float* row = x + (int64_t)blockIdx.x * ldx;
    float s = 0.f;
    if (VEC) {  // 16-byte loads, four running sums per thread
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int c = threadIdx.x; c < (C >> 2); c += kThreads) {
            const float4 v = r4[c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        s = (a.x + a.y) + (a.z + a.w);
    } else {
        for (int c = threadIdx.x; c < C; c += kThreads) s += row[c];
    }
    s = mr::wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = part[0];
#pragma unroll
        for (int k = 1; k < kThreads / 64; ++k) t += part[k];
        out[blockIdx.x] = t;
    }
}

// h = gelu_erf(u) (the forward of the training graph keeps the pre-activation, so the GEMM epilogue's fused GELU is not used)
__global__ __launch_bounds__(kThreads) void gelu_fwd_kernel(const float* __restrict__ u, int64_t n, float* __restrict__ h) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float x = u[i];
        h[i] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    }
}

// ------------------------------------------------------------------------------------------------ GELU(erf) backward
__global__ __launch_bounds__(kThreads) void gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dh, int64_t n,
                                                           float* __restrict__ du) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float x = u[i];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
        du[i] = dh[i] * (cdf + x * pdf);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// one wave per row: stats[row] = (mean, rstd) of x; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma
__global__ __launch_bounds__(kThreads) void layernorm_bwd_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                     int64_t ldy, const float* __restrict__ gamma, float eps, int T, int d,
                                                                     float* __restrict__ dx, int64_t lddx, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= T) return;
    const float* xr = x + (int64_t)row * ldx;
    const float* gr = dy + (int64_t)row * ldy;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = mr::wave_sum(s) / (float)d;
    float v = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float t = xr[c] - mean;
        v += t * t;
    }
    const float rstd = rsqrtf(mr::wave_sum(v) / (float)d + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float g = gr[c] * gamma[c];
        a += g;
        b += g * (xr[c] - mean) * rstd;
    }
    a = mr::wave_sum(a) / (float)d;
    b = mr::wave_sum(b) / (float)d;
    float* o = dx + (int64_t)row * lddx;
    for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        o[c] = rstd * (gr[c] * gamma[c] - a - xh * b);
    }
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// the same with the row held in registers: d = NV * 256 (768, 1024, ...), one 16-byte load per lane and 256 columns of x and of dy --
// each element is read once (the loop form above walks the row four times with 4-byte loads: 2.6 TB/s at 18,000 x 768)
template <int NV>
__global__ __launch_bounds__(kThreads) void layernorm_bwd_rows_vec_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                         int64_t ldy, const float* __restrict__ gamma, float eps, int T,
                                                                         float* __restrict__ dx, int64_t lddx, float* __restrict__ stats) {
    constexpr int d = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= T) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
    const float4* gr = reinterpret_cast<const float4*>(dy + (int64_t)row * ldy);
    const float4* gm = reinterpret_cast<const float4*>(gamma);
    float4 xv[NV], gv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        xv[k] = xr[lane + 64 * k];
        gv[k] = gr[lane + 64 * k];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
    const float mean = mr::wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        xv[k].x -= mean; xv[k].y -= mean; xv[k].z -= mean; xv[k].w -= mean;
        v += (xv[k].x * xv[k].x + xv[k].y * xv[k].y) + (xv[k].z * xv[k].z + xv[k].w * xv[k].w);
    }
    const float rstd = rsqrtf(mr::wave_sum(v) / (float)d + eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float4 w = gm[lane + 64 * k];
        gv[k].x *= w.x; gv[k].y *= w.y; gv[k].z *= w.z; gv[k].w *= w.w;           // g = dy * gamma
        xv[k].x *= rstd; xv[k].y *= rstd; xv[k].z *= rstd; xv[k].w *= rstd;       // xhat
        a += (gv[k].x + gv[k].y) + (gv[k].z + gv[k].w);
        b += (gv[k].x * xv[k].x + gv[k].y * xv[k].y) + (gv[k].z * xv[k].z + gv[k].w * xv[k].w);
    }
    a = mr::wave_sum(a) / (float)d;
    b = mr::wave_sum(b) / (float)d;
    float4* o = reinterpret_cast<float4*>(dx + (int64_t)row * lddx);
#pragma unroll
    for (int k = 0; k < NV; ++k)
        o[lane + 64 * k] = make_float4(rstd * (gv[k].x - a - xv[k].x * b), rstd * (gv[k].y - a - xv[k].y * b),
                                       rstd * (gv[k].z - a - xv[k].z * b), rstd * (gv[k].w - a - xv[k].w * b));
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// dgamma[c] = sum_t dy[t][c] * xhat[t][c], dbeta[c] = sum_t dy[t][c]; 32 columns x 8 row lanes per workgroup and row chunk,
// fixed-order LDS combine; outputs indexed [chunk][c] (chunk_reduce_kernel adds the chunks when there is more than one)
__global__ __launch_bounds__(kThreads) void layernorm_bwd_params_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                       int64_t ldy, const float* __restrict__ stats, int T, int d,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float pg[kRowL][kColW], pb[kRowL][kColW];
    const int cl = threadIdx.x % kColW, rl = threadIdx.x / kColW;
    const int c = blockIdx.x * kColW + cl;
    const int t0 = blockIdx.y * kRowsPerChunk, t1 = (t0 + kRowsPerChunk < T) ? t0 + kRowsPerChunk : T;
    float g = 0.f, b = 0.f;
    if (c < d)
        for (int t = t0 + rl; t < t1; t += kRowL) {
            const float dyv = dy[(int64_t)t * ldy + c];
            g += dyv * (x[(int64_t)t * ldx + c] - stats[2 * t]) * stats[2 * t + 1];
            b += dyv;
        }
    pg[rl][cl] = g;
    pb[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < d) {
        float tg = pg[0][cl], tb = pb[0][cl];
#pragma unroll
        for (int k = 1; k < kRowL; ++k) { tg += pg[k][cl]; tb += pb[k][cl]; }
        dgamma[(int64_t)blockIdx.y * d + c] = tg;
        dbeta[(int64_t)blockIdx.y * d + c] = tb;
    }
}

// Longformer global row backward: per (sequence, head) ONE query (qg, from query_global(x_cls)) against all keys / values of the
// sequence (kvg = [key_global(x) | value_global(x)]).  dctx_cls (B, H 64) = d loss / d ctx[cls rows], ctx_cls the forward output.
// Outputs dqg (B, H 64) and dkvg (T, 2 H 64).  One workgroup per (sequence, head), threads over keys.
__global__ __launch_bounds__(kThreads) void attn_global_row_bwd_kernel(const float* __restrict__ qg, const float* __restrict__ kvg,
                                                                      const float* __restrict__ ctx_cls, const float* __restrict__ dctx_cls,
                                                                      const int32_t* __restrict__ cu, int H, float scale,
                                                                      float* __restrict__ dqg, float* __restrict__ dkvg,
                                                                      uint32_t drop_thresh = 0u, float drop_inv = 1.f, uint32_t drop_key = 0u) {
    // drop_thresh != 0: the forward dropped the row's probabilities (mask = mr::dropout_keep(key, sequence * H + head, key position))
    __shared__ float red[kThreads / 64][kDh + 1];
    __shared__ float bc[2];
    const int b = blockIdx.x, h = blockIdx.y;
    const int t0 = cu[b], len = cu[b + 1] - t0;
    const int64_t ld = (int64_t)2 * H * kDh, ldq = (int64_t)H * kDh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float q[kDh], g[kDh];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < kDh; ++d) {
        q[d] = qg[(int64_t)b * ldq + h * kDh + d];
        g[d] = dctx_cls[(int64_t)b * ldq + h * kDh + d];
        dl = fmaf(g[d], ctx_cls[(int64_t)b * ldq + h * kDh + d], dl);
    }
    // pass 1: max and sum of exp over the keys
    float m = -INFINITY;
    for (int j = threadIdx.x; j < len; j += kThreads) {
        const float* k = kvg + (int64_t)(t0 + j) * ld + h * kDh;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < kDh; ++d) s = fmaf(q[d], k[d], s);
        m = fmaxf(m, s * scale);
    }
    m = mr::wave_max(m);
    if (lane == 0) red[wave][0] = m;
    __syncthreads();
    if (threadIdx.x == 0) bc[0] = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    __syncthreads();
    m = bc[0];
    float l = 0.f;
    for (int j = threadIdx.x; j < len; j += kThreads) {
        const float* k = kvg + (int64_t)(t0 + j) * ld + h * kDh;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < kDh; ++d) s = fmaf(q[d], k[d], s);
        l += expf(s * scale - m);
    }
    l = mr::wave_sum(l);
    __syncthreads();
    if (lane == 0) red[wave][0] = l;
    __syncthreads();
    if (threadIdx.x == 0) bc[1] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
    __syncthreads();
    const float lse = m + logf(bc[1]);
    // pass 2: per key dK, dV; dq accumulated per thread, then combined in a fixed order
    float dq[kDh];
#pragma unroll
    for (int d = 0; d < kDh; ++d) dq[d] = 0.f;
    for (int j = threadIdx.x; j < len; j += kThreads) {
        const float* k = kvg + (int64_t)(t0 + j) * ld + h * kDh;
        const float* v = k + H * kDh;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < kDh; ++d) {
            s = fmaf(q[d], k[d], s);
            dp = fmaf(g[d], v[d], dp);
        }
        const float p = expf(s * scale - lse);
        float pd = p;
        if (drop_thresh) {
            const bool keep = mr::dropout_keep(drop_key, (uint32_t)b * (uint32_t)H + (uint32_t)h, (uint32_t)j, drop_thresh);
            dp = keep ? dp * drop_inv : 0.f;
            pd = keep ? p * drop_inv : 0.f;
        }
        const float ds = p * (dp - dl) * scale;
        float* ok = dkvg + (int64_t)(t0 + j) * ld + h * kDh;
        float* ov = ok + H * kDh;
#pragma unroll
        for (int d = 0; d < kDh; ++d) {
            ok[d] = ds * q[d];
            ov[d] = pd * g[d];
            dq[d] = fmaf(ds, k[d], dq[d]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < kDh; ++d) {
        const float t = mr::wave_sum(dq[d]);
        if (lane == 0) red[wave][d] = t;
    }
    __syncthreads();
    if (threadIdx.x < kDh) {
        const int d = threadIdx.x;
        dqg[(int64_t)b * ldq + h * kDh + d] = ((red[0][d] + red[1][d]) + red[2][d]) + red[3][d];
    }
}

// ------------------------------------------------------------------------------------------------ scatter-add of rows
// table[idx[t]][:] += src[t][:]  (atomic: several tokens may share a row)
__global__ __launch_bounds__(kThreads) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ idx,
                                                                   int T, int d, float* __restrict__ table, int64_t ldt) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (t >= T) return;
    float* row = table + (int64_t)idx[t] * ldt;
    const float* s = src + (int64_t)t * lds_;
    for (int c = lane; c < d; c += 64) atomicAdd(row + c, s[c]);
}

}  // namespace

extern "C" int mr_transpose_f32(const float* in, int64_t ldi, int R, int C, float* out, int64_t ldo, int R_pad, mr_stream_t stream) {
    if (!in || !out || R < 0 || C < 0 || R_pad < R || ldi < C || ldo < R_pad) return MR_EINVAL;
    if (R_pad == 0 || C == 0) return MR_OK;
    if (!((C | R_pad) & 3) && !((ldi | ldo) & 3) && mr::aligned16(in) && mr::aligned16(out)) {
        const dim3 grid((C + 63) / 64, (R_pad + 63) / 64);
        hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, in, ldi, R, C, out, ldo, R_pad);
        return mr::check_launch();
    }
    const dim3 grid((C + 31) / 32, (R_pad + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, in, ldi, R, C, out, ldo, R_pad);
    return mr::check_launch();
}

extern "C" size_t mr_colsum_ws_bytes(int R, int C) {
    const int n = row_chunks(R);
    return (n > 1 && C > 0) ? (size_t)n * C * sizeof(float) : 0;
}

extern "C" int mr_colsum_f32(const float* x, int64_t ldx, int R, int C, float* out, void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (!x || !out || R < 0 || C < 0 || ldx < C) return MR_EINVAL;
    if (C == 0) return MR_OK;
    const int n = row_chunks(R);
    if (n > 1 && (!ws || ws_bytes < mr_colsum_ws_bytes(R, C))) return MR_EWS;
    float* part = n > 1 ? reinterpret_cast<float*>(ws) : out;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + kColW - 1) / kColW, n), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, R, C, part);
    if (n > 1)
        hipLaunchKernelGGL(chunk_reduce_kernel, dim3((C + kColW - 1) / kColW), dim3(kThreads), 0, (hipStream_t)stream, part, n, C, 1, out, out);
    return mr::check_launch();
}

extern "C" int mr_rowsum_f32(const float* x, int64_t ldx, int R, int C, float* out, mr_stream_t stream) {
    if (!x || !out || R < 0 || C < 0 || ldx < C) return MR_EINVAL;
    if (R == 0) return MR_OK;
    const bool vec = !(C & 3) && !(ldx & 3) && mr::aligned16(x);
    if (vec) hipLaunchKernelGGL(rowsum_kernel<true>, dim3(R), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, C, out);
    else hipLaunchKernelGGL(rowsum_kernel<false>, dim3(R), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, C, out);
    return mr::check_launch();
}

extern "C" int mr_gelu_fwd_f32(const float* u, int64_t n, float* h, mr_stream_t stream) {
    if (!u || !h || n < 0) return MR_EINVAL;
    if (n == 0) return MR_OK;
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, u, n, h);
    return mr::check_launch();
}

extern "C" int mr_gelu_bwd_f32(const float* u, const float* dh, int64_t n, float* du, mr_stream_t stream) {
    if (!u || !dh || !du || n < 0) return MR_EINVAL;
    if (n == 0) return MR_OK;
    int64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, u, dh, n, du);
    return mr::check_launch();
}

extern "C" size_t mr_layernorm_bwd_ws_bytes(int T, int d) {
    const int n = row_chunks(T);
    return (n > 1 && d > 0) ? (size_t)2 * n * d * sizeof(float) : 0;
}

extern "C" int mr_layernorm_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t ldy, const float* gamma, float eps, int T, int d,
                                    float* dx, int64_t lddx, float* stats, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                    mr_stream_t stream) {
    if (!x || !dy || !gamma || !dx || !stats || T < 0 || d < 1 || ldx < d || ldy < d || lddx < d) return MR_EINVAL;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return MR_EINVAL;
    if (T == 0) {
        if (dgamma) {
            hipMemsetAsync(dgamma, 0, (size_t)d * 4, (hipStream_t)stream);
            hipMemsetAsync(dbeta, 0, (size_t)d * 4, (hipStream_t)stream);
        }
        return mr::check_launch();
    }
    const int n = row_chunks(T);
    if (dgamma && n > 1 && (!ws || ws_bytes < mr_layernorm_bwd_ws_bytes(T, d))) return MR_EWS;
    const dim3 rgrid((T + kThreads / 64 - 1) / (kThreads / 64));
    const bool vec = (d % 256 == 0) && d <= 1024 && !((ldx | ldy | lddx) & 3) && mr::aligned16(x) && mr::aligned16(dy) && mr::aligned16(gamma) &&
                     mr::aligned16(dx);
#define MR_LNB(NV_) hipLaunchKernelGGL(layernorm_bwd_rows_vec_kernel<NV_>, rgrid, dim3(kThreads), 0, (hipStream_t)stream, x, ldx, dy, ldy, gamma, eps, T, dx, lddx, stats)
    if (vec && d == 256) MR_LNB(1);
    else if (vec && d == 512) MR_LNB(2);
    else if (vec && d == 768) MR_LNB(3);
    else if (vec && d == 1024) MR_LNB(4);
    else
        hipLaunchKernelGGL(layernorm_bwd_rows_kernel, rgrid, dim3(kThreads), 0, (hipStream_t)stream, x, ldx, dy, ldy, gamma, eps, T, d, dx, lddx, stats);
#undef MR_LNB
    if (dgamma) {
        float* pg = n > 1 ? reinterpret_cast<float*>(ws) : dgamma;
        float* pb = n > 1 ? pg + (size_t)n * d : dbeta;
        hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3((d + kColW - 1) / kColW, n), dim3(kThreads), 0, (hipStream_t)stream, x, ldx, dy,
                           ldy, stats, T, d, pg, pb);
        if (n > 1)
            hipLaunchKernelGGL(chunk_reduce_kernel, dim3((d + kColW - 1) / kColW), dim3(kThreads), 0, (hipStream_t)stream, pg, n, d, 2, dgamma,
                               dbeta);
    }
    return mr::check_launch();
}

extern "C" int mr_attn_global_row_bwd_f32(const float* qg, const float* kvg, const float* ctx_cls, const float* dctx_cls,
                                          const int32_t* cu_seqlens, int B, int H, int dh, float scale, float* dqg, float* dkvg,
                                          mr_stream_t stream) {
    if (!qg || !kvg || !ctx_cls || !dctx_cls || !cu_seqlens || !dqg || !dkvg || B < 0 || H < 1) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (B == 0) return MR_OK;
    hipLaunchKernelGGL(attn_global_row_bwd_kernel, dim3(B, H), dim3(kThreads), 0, (hipStream_t)stream, qg, kvg, ctx_cls, dctx_cls, cu_seqlens,
                       H, scale, dqg, dkvg);
    return mr::check_launch();
}

// backward of mr_attn_global_row_train_f32: the same drop_p / drop_key as the forward
extern "C" int mr_attn_global_row_bwd_train_f32(const float* qg, const float* kvg, const float* ctx_cls, const float* dctx_cls,
                                                const int32_t* cu_seqlens, int B, int H, int dh, float scale, float drop_p, uint32_t drop_key,
                                                float* dqg, float* dkvg, mr_stream_t stream) {
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    if (!qg || !kvg || !ctx_cls || !dctx_cls || !cu_seqlens || !dqg || !dkvg || B < 0 || H < 1) return MR_EINVAL;
    if (dh != kDh) return MR_EUNSUPPORTED;
    if (B == 0) return MR_OK;
    hipLaunchKernelGGL(attn_global_row_bwd_kernel, dim3(B, H), dim3(kThreads), 0, (hipStream_t)stream, qg, kvg, ctx_cls, dctx_cls, cu_seqlens,
                       H, scale, dqg, dkvg, thresh, thresh ? inv : 1.f, drop_key);
    return mr::check_launch();
}

extern "C" int mr_scatter_add_rows_f32(const float* src, int64_t lds_, const int32_t* idx, int T, int d, float* table, int64_t ldt,
                                       mr_stream_t stream) {
    if (!src || !idx || !table || T < 0 || d < 1 || lds_ < d || ldt < d) return MR_EINVAL;
    if (T == 0) return MR_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((T + kThreads / 64 - 1) / (kThreads / 64)), dim3(kThreads), 0, (hipStream_t)stream, src,
                       lds_, idx, T, d, table, ldt);
    return mr::check_launch();
}
