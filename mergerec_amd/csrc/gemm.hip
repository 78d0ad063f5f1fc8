// K3 / K5-GEMM: C = act(A W^T + bias) (+ R) in exact fp32 on the matrix cores.
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TFLOP/s chip peak) and is bit-for-bit an
// fp32 FMA chain over k, so the product path keeps fp32-reference numerics (logits within 1e-4,
// ranked indices exact) while staying on MFMA.  Block tile 128x128x16, 4 waves (2x2), each wave a
// 64x64 patch = 2x2 MFMA tiles (64 accumulator VGPRs).  Both operands are K-contiguous ("NT"), staged
// global -> registers -> LDS (double-buffered LDS, one barrier per k-tile; next tile's global loads are
// in flight under the current tile's 32 MFMAs per wave).  LDS rows are split into even-k | odd-k halves
// so that lane (row, h) fetches its 8 operands for the 8 MFMA steps of a k-tile with two ds_read_b128
// and every output element accumulates k = 0, 1, 2, ... in ascending order.  Row stride 20 floats makes
// the b128 reads conflict-free.  fp32 MFMA needs only ~8 B/clk/CU of operand traffic, far below L2/LDS
// limits: the kernel is MFMA-issue bound by construction.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_STRIDE = 20;  // floats per LDS row: 8 even-k | 8 odd-k | 4 pad
constexpr int kThreads = 256;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// BMT = rows of the block tile: 128 (waves 2 x 2, wave tile 64 x 64) or 64 (wave tile 32 x 64: half the work per workgroup, for shapes whose
// 128-row tiling leaves the CUs unevenly loaded -- 256 users x 22,855 items is 358 tiles on 256 CUs, i.e. 2 rounds of time for 1.4 of
// work; 716 half tiles are 3 half-rounds).  Every output element's k chain is the same in both: results are bit-identical.
template <int ACT, bool HAS_R, int BMT = BM>
__global__ __launch_bounds__(kThreads, 4) void gemm_nt_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ w0, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ b0, const float* __restrict__ b1,
    const float* __restrict__ b2, int M, int seg_n, int K, const float* __restrict__ R, int64_t ldr,
    float* __restrict__ C, int64_t ldc, int tiles_n_seg, int tiles_n, int nwg, int kchunk, int64_t split_stride) {
    constexpr int MI = BMT / 64;  // 32-row MFMA tiles per wave along M
    __shared__ __attribute__((aligned(16))) float lds[2][(BMT + BN) * LDS_STRIDE];
    // split-K (mr_gemm_nt_splitk_f32 only; every other caller passes kchunk = K, gridDim.y = 1): slice blockIdx.y of the k range,
    // raw partial sums to C + blockIdx.y * split_stride
    const int k_begin = blockIdx.y * kchunk;
    const int k_len = (K - k_begin) < kchunk ? (K - k_begin) : kchunk;
    C += (int64_t)blockIdx.y * split_stride;

    const int pid = mr::xcd_remap(blockIdx.x, nwg);
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int seg = tn / tiles_n_seg;
    const int n0 = (tn - seg * tiles_n_seg) * BN;  // column inside the segment
    const int m0 = tm * BMT;
    const float* __restrict__ W = seg == 0 ? w0 : (seg == 1 ? w1 : w2);
    const float* __restrict__ bias = seg == 0 ? b0 : (seg == 1 ? b1 : b2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging map: thread -> (row r / r+64, k-quad kq)
    const int sr = tid >> 2, kq = tid & 3;
    int ar0 = m0 + sr, ar1 = m0 + sr + 64;
    ar0 = ar0 < M ? ar0 : M - 1;
    ar1 = ar1 < M ? ar1 : M - 1;
    int br0 = n0 + sr, br1 = n0 + sr + 64;
    br0 = br0 < seg_n ? br0 : seg_n - 1;
    br1 = br1 < seg_n ? br1 : seg_n - 1;
    const float* ga0 = A + (int64_t)ar0 * lda + kq * 4 + k_begin;
    const float* ga1 = A + (int64_t)ar1 * lda + kq * 4 + k_begin;
    const float* gb0 = W + (int64_t)br0 * K + kq * 4 + k_begin;
    const float* gb1 = W + (int64_t)br1 * K + kq * 4 + k_begin;
    // LDS write offsets (floats): even half at 2*kq, odd half at 8 + 2*kq
    const int wa0 = sr * LDS_STRIDE + 2 * kq, wa1 = (sr + 64) * LDS_STRIDE + 2 * kq;
    const int wb0 = (BMT + sr) * LDS_STRIDE + 2 * kq, wb1 = (BMT + sr + 64) * LDS_STRIDE + 2 * kq;
    // LDS read offsets (floats)
    const int ra = (wm * 32 * MI + lr) * LDS_STRIDE + lh * 8;
    const int rb = (BMT + wn * 64 + lr) * LDS_STRIDE + lh * 8;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 sa0, sa1, sb0, sb1;
    auto gload = [&](int k0) {
        sa0 = *reinterpret_cast<const float4*>(ga0 + k0);
        if (MI == 2) sa1 = *reinterpret_cast<const float4*>(ga1 + k0);
        sb0 = *reinterpret_cast<const float4*>(gb0 + k0);
        sb1 = *reinterpret_cast<const float4*>(gb1 + k0);
    };
    auto lstore = [&](float* buf) {
        *reinterpret_cast<float2*>(buf + wa0) = make_float2(sa0.x, sa0.z);
        *reinterpret_cast<float2*>(buf + wa0 + 8) = make_float2(sa0.y, sa0.w);
        if (MI == 2) {
            *reinterpret_cast<float2*>(buf + wa1) = make_float2(sa1.x, sa1.z);
            *reinterpret_cast<float2*>(buf + wa1 + 8) = make_float2(sa1.y, sa1.w);
        }
        *reinterpret_cast<float2*>(buf + wb0) = make_float2(sb0.x, sb0.z);
        *reinterpret_cast<float2*>(buf + wb0 + 8) = make_float2(sb0.y, sb0.w);
        *reinterpret_cast<float2*>(buf + wb1) = make_float2(sb1.x, sb1.z);
        *reinterpret_cast<float2*>(buf + wb1 + 8) = make_float2(sb1.y, sb1.w);
    };

    const int nk = k_len / BK;
    gload(0);
    lstore(lds[0]);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const float* buf = lds[kt & 1];
        // Unconditional prefetch (the last iteration harmlessly re-reads tile 0 into the idle buffer): a
        // conditional load makes hipcc copy the staged registers and wait vmcnt(0) right behind the loads.
        gload((kt + 1 < nk) ? (kt + 1) * BK : 0);
        // Pin the prefetch ahead of the MFMA block: left alone, hipcc sinks these loads behind the MFMAs to
        // reuse the fragment registers and then waits for them at once (memory latency exposed per k-tile).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float4 a[MI][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < MI) {
                a[i][0] = *reinterpret_cast<const float4*>(buf + ra + i * 32 * LDS_STRIDE);
                a[i][1] = *reinterpret_cast<const float4*>(buf + ra + i * 32 * LDS_STRIDE + 4);
            }
            b[i][0] = *reinterpret_cast<const float4*>(buf + rb + i * 32 * LDS_STRIDE);
            b[i][1] = *reinterpret_cast<const float4*>(buf + rb + i * 32 * LDS_STRIDE + 4);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float av[MI], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 fb = b[i][s >> 2];
                bv[i] = (s & 3) == 0 ? fb.x : ((s & 3) == 1 ? fb.y : ((s & 3) == 2 ? fb.z : fb.w));
                if (i < MI) {
                    const float4 fa = a[i][s >> 2];
                    av[i] = (s & 3) == 0 ? fa.x : ((s & 3) == 1 ? fa.y : ((s & 3) == 2 ? fa.z : fa.w));
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the vmcnt wait + LDS stores behind all 32 MFMAs
        lstore(lds[(kt + 1) & 1]);
        __syncthreads();
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
    // One branch-free path for interior and edge tiles: C and R are addressed through tile-local buffer resources whose
    // num_records ends at the tile's last valid element, so rows past M are dropped by the hardware bounds check and
    // lanes whose column is past seg_n carry an out-of-range offset.  (Per-element guards put every store in its own
    // basic block behind `s_waitcnt vmcnt(0)`, i.e. each store waited for the previous one to complete.)
    const int rows_valid = (M - m0) < BMT ? (M - m0) : BMT;
    const int cols_valid = (seg_n - n0) < BN ? (seg_n - n0) : BN;
    const int64_t col0 = (int64_t)seg * seg_n + n0;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        C + (int64_t)m0 * ldc + col0, 0, (int)(((int64_t)(rows_valid - 1) * ldc + cols_valid) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(HAS_R ? R + (int64_t)m0 * ldr + col0 : C), 0,
        HAS_R ? (int)(((int64_t)(rows_valid - 1) * ldr + cols_valid) * 4) : 0, 0x00020000);
    // opaque copies of the lane coordinates: keeps this address arithmetic from being hoisted above the main loop
    int lr_e = lr, lh_e = lh;
    asm volatile("" : "+v"(lr_e), "+v"(lh_e));
    const uint32_t ldc4 = (uint32_t)ldc * 4u, ldr4 = (uint32_t)ldr * 4u;
    float bz[2];
    uint32_t coff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int colt = wn * 64 + j * 32 + lr_e;
        const bool ok = colt < cols_valid;
        bz[j] = bias ? bias[n0 + (ok ? colt : cols_valid - 1)] : 0.f;
        coff[j] = ok ? (uint32_t)colt * 4u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const uint32_t rowt = wm * 32 * MI + i * 32 + 4 * lh_e;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // four rows at a time (one accumulator quad)
                float v[4], rr[4];
                if (HAS_R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (rowt + r + 8 * q) * ldr4 + coff[j], 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][4 * q + r] + bz[j];
                    if (ACT == MR_ACT_GELU_ERF) v[r] = gelu_erf(v[r]);
                    if (ACT == MR_ACT_TANH) v[r] = tanhf(v[r]);
                    if (HAS_R) v[r] += rr[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[r]), crs, (rowt + r + 8 * q) * ldc4 + coff[j], 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// C[m][n] = sum_z ws[z][m][n] (ascending z) + bias[n] + R[m][n]
__global__ __launch_bounds__(kThreads) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t split_stride, int M, int N,
                                                                const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
                                                                float* __restrict__ C, int64_t ldc) {
    const int64_t total = (int64_t)M * N;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
        float s = ws[e];
        for (int z = 1; z < splits; ++z) s += ws[(int64_t)z * split_stride + e];
        if (bias) s += bias[n];
        if (R) s += R[(int64_t)m * ldr + n];
        C[(int64_t)m * ldc + n] = s;
    }
}

}  // namespace

extern "C" size_t mr_gemm_nt_splitk_ws_bytes(int M, int N, int splits) { return (size_t)M * N * 4 * (splits > 1 ? splits : 0); }

// C = A W^T (+ bias) (+ R) with the k range cut into `splits` slices computed by separate workgroups (a product with few output
// tiles and a long k loop -- the training graph's token-sized GEMMs -- otherwise leaves most of the chip idle).  Partial sums are
// combined in ascending slice order: deterministic, but NOT the single ascending-k FMA chain of mr_gemm_nt_bias_act_f32.
extern "C" int mr_gemm_nt_splitk_f32(const float* A, int64_t lda, const float* W, const float* bias, int M, int N, int K, const float* R,
                                     int64_t ldr, float* C, int64_t ldc, int splits, void* ws, size_t ws_bytes, mr_stream_t stream) {
    if (splits <= 1) return mr_gemm_nt_bias_act_f32(A, lda, W, nullptr, nullptr, bias, nullptr, nullptr, 1, M, N, K, MR_ACT_NONE, R, ldr, C, ldc, stream);
    if (!A || !W || !C || !ws || M < 0 || N < 1 || K < 1) return MR_EINVAL;
    if (K % BK) return MR_EUNSUPPORTED;
    if ((lda & 3) || !mr::aligned16(A) || !mr::aligned16(W) || !mr::aligned16(ws)) return MR_EALIGN;
    if (ws_bytes < mr_gemm_nt_splitk_ws_bytes(M, N, splits)) return MR_EWS;
    if (M == 0) return MR_OK;
    const int nk = K / BK;
    if (splits > nk) splits = nk;
    const int kchunk = ((nk + splits - 1) / splits) * BK;
    splits = (K + kchunk - 1) / kchunk;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nwg64 = (int64_t)tiles_m * tiles_n;
    if (nwg64 > 0x7fffffff || (int64_t)M * N > 0x7fffffff) return MR_EUNSUPPORTED;
    const int nwg = (int)nwg64;
    hipStream_t st = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(ws);
    const int64_t stride = (int64_t)M * N;
    hipLaunchKernelGGL((gemm_nt_kernel<MR_ACT_NONE, false>), dim3(nwg, splits), dim3(kThreads), 0, st, A, lda, W, nullptr, nullptr, nullptr,
                       nullptr, nullptr, M, N, K, nullptr, (int64_t)0, part, (int64_t)N, tiles_n, tiles_n, nwg, kchunk, stride);
    int64_t blocks = (stride + kThreads - 1) / kThreads;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, part, splits, stride, M, N, bias, R, ldr, C, ldc);
    return mr::check_launch();
}

extern "C" int mr_gemm_nt_bias_act_f32(const float* A, int64_t lda, const float* w0, const float* w1, const float* w2,
                                       const float* b0, const float* b1, const float* b2, int nseg, int M, int seg_n,
                                       int K, int act, const float* R, int64_t ldr, float* C, int64_t ldc,
                                       mr_stream_t stream) {
    if (!A || !w0 || !C || nseg < 1 || nseg > 3 || M < 0 || seg_n < 1 || K < 1) return MR_EINVAL;
    if ((nseg > 1 && !w1) || (nseg > 2 && !w2)) return MR_EINVAL;
    if (K % BK) return MR_EUNSUPPORTED;
    if (nseg > 1 && (seg_n % BN)) return MR_EUNSUPPORTED;
    if (act != MR_ACT_NONE && act != MR_ACT_GELU_ERF && act != MR_ACT_TANH) return MR_EUNSUPPORTED;
    if (act == MR_ACT_TANH && R) return MR_EUNSUPPORTED;  // the pooler head has no residual
    if (lda & 3) return MR_EALIGN;  // A and W rows are read as float4; C and R are accessed per element
    if (!mr::aligned16(A) || !mr::aligned16(w0) || (w1 && !mr::aligned16(w1)) || (w2 && !mr::aligned16(w2)))
        return MR_EALIGN;
    if (ldc < 1 || ldc > (1 << 21) || (R && (ldr < 1 || ldr > (1 << 21)))) return MR_EUNSUPPORTED;  // 32-bit tile-local offsets in the epilogue
    if (M == 0) return MR_OK;
    const int tiles_n_seg = (seg_n + BN - 1) / BN;
    const int tiles_n = tiles_n_seg * nseg;
    hipStream_t st = (hipStream_t)stream;
    if (act == MR_ACT_NONE && !R) {
        // plain product (the scoring GEMM): 64-row tiles when they load the 256 CUs more evenly than 128-row tiles.  The kernel is
        // matrix-pipe bound, so a launch takes ceil(workgroups / CUs) rounds of one workgroup's time (5 % charged to the half tile
        // for its extra LDS traffic per MFMA).
        const int64_t n128 = (int64_t)((M + 127) / 128) * tiles_n, n64 = (int64_t)((M + 63) / 64) * tiles_n;
        static const int force = [] { const char* e = getenv("MR_GEMM_F32_BM"); return e ? atoi(e) : 0; }();  // A/B: 64 / 128
        const bool half = force ? force == 64 : (double)((n64 + 255) / 256) * 0.5 * 1.05 < (double)((n128 + 255) / 256);
        if (half) {
            if (n64 > 0x7fffffff) return MR_EUNSUPPORTED;
            hipLaunchKernelGGL((gemm_nt_kernel<MR_ACT_NONE, false, 64>), dim3((unsigned)n64), dim3(kThreads), 0, st, A, lda, w0, w1, w2, b0, b1, b2,
                               M, seg_n, K, R, ldr, C, ldc, tiles_n_seg, tiles_n, (int)n64, K, (int64_t)0);
            return mr::check_launch();
        }
    }
    const int tiles_m = (M + BM - 1) / BM;
    const int64_t nwg64 = (int64_t)tiles_m * tiles_n;
    if (nwg64 > 0x7fffffff) return MR_EUNSUPPORTED;
    const int nwg = (int)nwg64;
#define MR_GEMM_LAUNCH(ACT_, HASR_)                                                                                   \
    hipLaunchKernelGGL((gemm_nt_kernel<ACT_, HASR_>), dim3(nwg), dim3(kThreads), 0, st, A, lda, w0, w1, w2, b0, b1, b2, \
                       M, seg_n, K, R, ldr, C, ldc, tiles_n_seg, tiles_n, nwg, K, (int64_t)0)
    if (act == MR_ACT_TANH) {
        MR_GEMM_LAUNCH(MR_ACT_TANH, false);
    } else if (act == MR_ACT_GELU_ERF) {
        if (R) MR_GEMM_LAUNCH(MR_ACT_GELU_ERF, true); else MR_GEMM_LAUNCH(MR_ACT_GELU_ERF, false);
    } else {
        if (R) MR_GEMM_LAUNCH(MR_ACT_NONE, true); else MR_GEMM_LAUNCH(MR_ACT_NONE, false);
    }
#undef MR_GEMM_LAUNCH
    return mr::check_launch();
}
