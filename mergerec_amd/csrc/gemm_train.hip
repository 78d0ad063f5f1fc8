// Token-sized exact-fp32 products of the collaborative-merging step (merge_train.py: 16 pseudo-user sequences ~ 600 tokens against freshly
// merged weights): forward  Y = X W^T,  input gradient  dX = dY W,  weight gradient  dW = dY^T X  -- all three through ONE kernel that
// reads either operand in either orientation, so no operand is ever transposed in memory (r03: 168 / 480 transpose launches per step) and no
// product is cut along k (r03: 176 / 512 split-K reduce launches): the tile is small enough that a 600 x 768 output alone fills the chip.
//
//   C[m][n] = sum_k Aop[m][k] * Bop[n][k]      TA = 0: A is (M, K) row-major (k contiguous)      TA = 1: A is (K, M) row-major (m contiguous)
//                                              TB = 0: B is (N, K) row-major (a Linear weight)   TB = 1: B is (K, N) row-major
//   forward  : TA = 0, TB = 0 (B = W)          dX : TA = 0, TB = 1 (A = dY, B = W, k = out features)
//   dW       : TA = 1, TB = 1 (A = dY, B = X, k = tokens; K need not be a multiple of 16: rows past K read as zero)
//
// Block tile 64 x BN x 64 (BN = 64 or 32), 12 waves: 8 CONSUMER waves own the accumulators and do nothing but LDS operand reads and
// v_mfma_f32_16x16x4_f32 (wave tile 32 x 16 / 16 x 16; two consumer waves per SIMD, so one wave's LDS-read and MFMA-dependency latency is
// the other's issue slot), 4 PRODUCER waves stage the operand tiles global -> registers -> LDS (a 3-deep register ring).  The MFMA is an
// fp32 FMA chain over k and a workgroup walks its whole k range, so every output element is the single ascending-k chain of the C oracle
// (oracle/oracle_c.c gemm_nt_ref) -- bit for bit, for every orientation and tile width.
// LDS holds both operand tiles k-major ([k][m], [k][n]): a lane's MFMA operand for a k-step is one ds_read_b32, conflict-free by the row
// pitch; m-contiguous sources are stored with one ds_write_b128 per thread, k-contiguous sources with four ds_write_b32.
// Fused epilogue: bias, dropout (the counter mask of dropout.h), residual, GELU forward (pre-activation AND activation written), GELU
// backward (x gelu'(u)); weight-gradient launches also emit the bias gradient (column sums of dY, fixed order).
// Measured (profiles/r04_gemm_tile_bench.txt, r04_sq_gemm_tile.txt): 0.31-0.46 of the fp32-MFMA peak on the 600-token products of
// BLaIR-base -- per layer 417 us against 472 us for the r03 route (NT kernel + split-K + reduce + transposes) with a quarter of its
// launches; five structures were tried on the way (one 256-thread workgroup with a register ring 2-8 tiles deep; BK 16 / 32 / 64;
// a scheduling region per k-tile; 4 + 4 and 8 + 4 specialised waves): the wave specialisation is what moved it (483 -> 407 us).
#include "common.h"
#include "dropout.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, kThreads = 768;   // 8 consumer waves (MFMA: two per SIMD) + 4 producer waves (global -> LDS staging)

struct Args {
    const float* A; int64_t lda;
    const float* B[3]; int64_t ldb; int seg_b;      // TB = 0: segments along N (seg_b rows of N each); TB = 1: along K (seg_b rows of K each)
    const float* bias[3];                           // TB = 0: per N segment; TB = 1: bias[0] over all N (or NULL)
    int M, N, K;
    const float* R; int64_t ldr;
    float* C[3]; int64_t ldc; int seg_c;            // segments of C along M (seg_c rows each): the stacked q / k / v weight gradients
    float* colsum[3];                               // per C segment: sum_k Aop[m][k] (bias gradient); NULL = not wanted
    int epi; const float* E; int64_t lde; float* C2; int64_t ldc2;
    uint32_t drop_thresh; float drop_inv; uint32_t drop_key;
    int tiles_m, tiles_n, nwg;
};

// ---- bf16x6 arithmetic (ARITH = 1): x = hi + mid + lo with three bf16 pieces, six products, fp32 accumulation (as csrc/gemm_bf16.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPitchB = 144;   // bytes per row of a piece image: 64 bf16 (one k-tile) + 16 -> the 16 rows of a ds_read_b128 hit 64 distinct banks
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split4(const float4 x, uint2& h, uint2& m, uint2& l) {
    h.x = pack2(x.x, x.y);
    h.y = pack2(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = pack2(r0, r1);
    m.y = pack2(r2, r3);
    l.x = pack2(r0 - lo_f(m.x), r1 - hi_f(m.x));
    l.y = pack2(r2 - lo_f(m.y), r3 - hi_f(m.y));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {  // as csrc/backward.hip gelu_bwd_kernel
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// BK = k depth of one LDS tile (one barrier).  Global loads are shaped for the texture path first: a k-contiguous source is read with
// BK / 4 lanes per row (a full 128- or 256-byte piece of each row per instruction -- one lane per row, 16 bytes each, costs 4 x the address
// processing and ran the whole kernel at 0.2 of the fp32 peak), an m-contiguous source with ROWS / 4 lanes per k row.
template <bool TRANS, int ROWS, int BK, bool WIDE>
struct Tile {
    static constexpr int NV = ROWS * BK / 1024;                       // float4 per thread and k-tile
    // LDS pitch (floats per k row).  Reads: lanes of different k must hit different banks (32x32x2: two k per instruction, 16x16x4: four).
    // Writes of a k-contiguous source are four ds_write_b32 per float4 with BK / 4 lanes along k: the pitch makes those lanes hit
    // different banks too (pitch = 1 mod 16 for 16 lanes per row, 2 mod 16 for 8).
    static constexpr int BASE = ROWS == 64 ? 80 : 48;   // 16x16x4 reads: the four k rows of an instruction land 16 banks apart
    static constexpr int S = TRANS ? BASE : BASE + (BK == 64 ? 1 : 2);
    static constexpr int LPR = TRANS ? ROWS / 4 : BK / 4;             // lanes along the contiguous direction
    static constexpr int PER_PASS = 256 / LPR;                        // rows (k rows if TRANS) covered by one pass of the workgroup
};

// ZF: K is not a multiple of BK -- the last k-tile is zero-filled past K (token-deep weight gradients); otherwise no select is compiled in
// ARITH: 0 = exact fp32 (v_mfma_f32_16x16x4_f32; every output element the single ascending-k FMA chain); 1 = bf16x6 split precision
// (three bf16 pieces per operand, six v_mfma_f32_16x16x32_bf16 products per 32 k: fp32-grade ~2^-24 per product, full fp32 range -- gradients
// of 1e-6 included, which fp16 pieces would not hold -- at 2.7 x fewer matrix-pipe cycles; the producers split while they stage).
template <bool TA, bool TB, int BN, int BK, int KD, bool ZF, int ARITH = 0>
__global__ __launch_bounds__(kThreads, 1) void gemm_tile_kernel(const Args g) {
    constexpr bool WIDE = BN == 64;
    static_assert(ARITH == 0 || BK == 64, "the piece images hold one 64-k tile per row");
    using TAi = Tile<TA, BM, BK, WIDE>;
    using TBi = Tile<TB, BN, BK, WIDE>;
    constexpr int SA = TAi::S, SB = TBi::S;
    constexpr int kDepth = KD;                // k-tiles in flight (even: the LDS buffer of a ring slot is then a compile-time constant)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float* const As0 = lds_dyn;               // [2][BK * SA]                       (ARITH 0)
    float* const Bs0 = lds_dyn + 2 * BK * SA; // [2][BK * SB]
    // ARITH 1: per buffer three piece images of A (64 rows x kPitchB bytes each) then three of B (BN rows): [row][k] with k contiguous
    unsigned char* const Hs0 = reinterpret_cast<unsigned char*>(lds_dyn);
    constexpr int kImgA = BM * kPitchB, kImgB = BN * kPitchB, kBufH = 3 * (kImgA + kImgB);

    // tile order: row tile fastest -- the (few) row tiles that share a weight panel run together on one XCD
    const int pid = mr::xcd_remap(blockIdx.x, g.nwg);
    const int tn = pid / g.tiles_m, tm = pid - tn * g.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = g.M, N = g.N, K = g.K;
    // Wave specialisation: waves 0-3 own the accumulators and do nothing but LDS operand reads and MFMAs; waves 4-7 stage the operand tiles
    // (global -> registers -> LDS).  A SIMD then holds one wave of each kind, and the staging work (address updates, zero fill, LDS writes,
    // waiting for global memory) overlaps the other wave's MFMA chain instead of queueing behind it in one instruction stream -- with one
    // 256-thread workgroup per CU the matrix pipe sat idle 70 % of the time (SQ counters, profiles/r04_sq_gemm_tile.txt).
    const bool producer = threadIdx.x >= 512;   // wave-uniform
    const int tid = producer ? (int)threadIdx.x - 512 : (int)threadIdx.x;   // producers: 0..255 (staging maps); consumers: 0..511
    const int lane = tid & 63, wave = tid >> 6;
    // consumer wave tile: BN = 64 -> 32 x 16 (waves 2 x 4, two accumulator chains), BN = 32 -> 16 x 16 (waves 4 x 2, one chain): two consumer
    // waves per SIMD, so one wave's LDS-read and MFMA-dependency latency is the other's issue slot
    constexpr int NI = WIDE ? 2 : 1;                       // 16-row MFMA tiles per wave along M
    const int wm = WIDE ? (wave >> 2) : (wave >> 1), wn = WIDE ? (wave & 3) : (wave & 1);

    // ---- global -> registers -> LDS ([k][m] / [k][n] images)
    const int ca = tid % TAi::LPR, ra_ = tid / TAi::LPR;   // position along the contiguous direction / across it
    const int cb = tid % TBi::LPR, rb_ = tid / TBi::LPR;
    struct Stage { float4 a[TAi::NV], b[TBi::NV]; };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // per-thread source OFFSETS of k-tile 0 (32-bit, in floats, against a workgroup-uniform base: one v_add per load instead of 64-bit
    // pointer arithmetic), one per pass; rows / columns clamped once (re-reads of valid data, discarded by the epilogue).  gload only adds the
    // tile's k offset: no data-dependent work behind a load (a select on the loaded value would make the wave wait for its own prefetch
    // at once) -- rows past K are zeroed when the stage is written to LDS, and only in launches whose K is not a multiple of BK (ZF).
    uint32_t oa[TAi::NV], ob[TBi::NV];
    const uint32_t lda = (uint32_t)g.lda, ldb = (uint32_t)g.ldb;
#pragma unroll
    for (int p = 0; p < TAi::NV; ++p) {
        if (TA) {   // (K, M) source: k row ra_ + p * PER_PASS (+ k0), columns m0 + 4 ca .. + 3
            int col = m0 + 4 * ca;
            col = col <= M - 4 ? col : M - 4;
            oa[p] = (uint32_t)col + (uint32_t)(ra_ + p * TAi::PER_PASS) * lda;
        } else {    // (M, K) source: row, k = 4 ca .. + 3 (+ k0)
            int row = m0 + ra_ + p * TAi::PER_PASS;
            row = row < M ? row : M - 1;
            oa[p] = (uint32_t)row * lda + 4u * ca;
        }
    }
    const int nseg = TB ? 0 : (n0 >= g.seg_b) + (n0 >= 2 * g.seg_b);   // !TB: this tile lies inside one N segment (seg_b % 64 == 0)
    const float* __restrict__ Bn = TB ? nullptr : g.B[nseg];
#pragma unroll
    for (int p = 0; p < TBi::NV; ++p) {
        if (TB) {   // (K, N) sources stacked along K: k row rb_ + p * PER_PASS (+ k0 inside its segment)
            int col = n0 + 4 * cb;
            col = col <= N - 4 ? col : N - 4;
            ob[p] = (uint32_t)col + (uint32_t)(rb_ + p * TBi::PER_PASS) * ldb;
        } else {
            int row = n0 + rb_ + p * TBi::PER_PASS;
            row = row < N ? row : N - 1;
            ob[p] = (uint32_t)(row - nseg * g.seg_b) * ldb + 4u * cb;
        }
    }
    const float* __restrict__ Ag = g.A;
    auto gload = [&](int kt, Stage& st) {
        const int k0 = kt * BK;   // uniform
        // past-K rows / columns (ZF launches only) re-read the tile's first row / column: any valid address, zeroed in lstore
#pragma unroll
        for (int p = 0; p < TAi::NV; ++p) {
            if (TA) {
                const bool in = !ZF || (k0 + ra_ + p * TAi::PER_PASS < K);
                st.a[p] = *reinterpret_cast<const float4*>(Ag + (in ? oa[p] : oa[p] - (uint32_t)(ra_ + p * TAi::PER_PASS) * lda) + (uint32_t)k0 * lda);
            } else {
                const bool in = !ZF || (k0 + 4 * ca < K);
                st.a[p] = *reinterpret_cast<const float4*>(Ag + (in ? oa[p] : oa[p] - 4u * ca) + (uint32_t)k0);
            }
        }
        const int kseg = TB ? ((k0 >= g.seg_b) + (k0 >= 2 * g.seg_b)) : 0;   // TB: a k-tile lies inside one K segment (seg_b % 64 == 0)
        const float* __restrict__ Bk = TB ? g.B[kseg] : Bn;
        const uint32_t kin = (uint32_t)(k0 - kseg * g.seg_b);
#pragma unroll
        for (int p = 0; p < TBi::NV; ++p) {
            if (TB) {
                const bool in = !ZF || (k0 + rb_ + p * TBi::PER_PASS < K);
                st.b[p] = *reinterpret_cast<const float4*>(Bk + (in ? ob[p] : ob[p] - (uint32_t)(rb_ + p * TBi::PER_PASS) * ldb) + kin * ldb);
            } else {
                const bool in = !ZF || (k0 + 4 * cb < K);
                st.b[p] = *reinterpret_cast<const float4*>(Bk + (in ? ob[p] : ob[p] - 4u * cb) + (uint32_t)k0);
            }
        }
    };
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);   // ARITH 1, TA: this producer thread's share of sum_k Aop[m][k] (the bias gradient)
    auto lstore = [&](int buf, const Stage& st, int kt, bool real) {   // kt = the k-tile the stage holds (zero fill past K); real: not the re-read behind the last tile
        float* as = As0 + buf * (BK * SA);
        float* bs = Bs0 + buf * (BK * SB);
        const int k0 = kt * BK;
        if (ARITH == 1) {
            unsigned char* ha = Hs0 + buf * kBufH;
            unsigned char* hb = ha + 3 * kImgA;
#pragma unroll
            for (int p = 0; p < TAi::NV; ++p) {
                const bool ok = !ZF || (TA ? (k0 + ra_ + p * TAi::PER_PASS < K) : (k0 + 4 * ca < K));
                const float4 v = ok ? st.a[p] : zero4;
                uint2 h, m, l;
                split4(v, h, m, l);
                if (TA) {   // four rows m .. m + 3 at ONE k: a 16-bit store per row and piece
                    const int kr = ra_ + p * TAi::PER_PASS;
                    if (real) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
                    unsigned char* d = ha + (4 * ca) * kPitchB + 2 * kr;
                    const uint2 pc[3] = {h, m, l};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        unsigned char* dq = d + q * kImgA;
                        *reinterpret_cast<uint16_t*>(dq) = (uint16_t)(pc[q].x & 0xffffu);
                        *reinterpret_cast<uint16_t*>(dq + kPitchB) = (uint16_t)(pc[q].x >> 16);
                        *reinterpret_cast<uint16_t*>(dq + 2 * kPitchB) = (uint16_t)(pc[q].y & 0xffffu);
                        *reinterpret_cast<uint16_t*>(dq + 3 * kPitchB) = (uint16_t)(pc[q].y >> 16);
                    }
                } else {    // four consecutive k of ONE row: an 8-byte store per piece
                    unsigned char* d = ha + (ra_ + p * TAi::PER_PASS) * kPitchB + 8 * ca;
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + kImgA) = m;
                    *reinterpret_cast<uint2*>(d + 2 * kImgA) = l;
                }
            }
#pragma unroll
            for (int p = 0; p < TBi::NV; ++p) {
                const bool ok = !ZF || (TB ? (k0 + rb_ + p * TBi::PER_PASS < K) : (k0 + 4 * cb < K));
                const float4 v = ok ? st.b[p] : zero4;
                uint2 h, m, l;
                split4(v, h, m, l);
                if (TB) {
                    const int kr = rb_ + p * TBi::PER_PASS;
                    unsigned char* d = hb + (4 * cb) * kPitchB + 2 * kr;
                    const uint2 pc[3] = {h, m, l};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        unsigned char* dq = d + q * kImgB;
                        *reinterpret_cast<uint16_t*>(dq) = (uint16_t)(pc[q].x & 0xffffu);
                        *reinterpret_cast<uint16_t*>(dq + kPitchB) = (uint16_t)(pc[q].x >> 16);
                        *reinterpret_cast<uint16_t*>(dq + 2 * kPitchB) = (uint16_t)(pc[q].y & 0xffffu);
                        *reinterpret_cast<uint16_t*>(dq + 3 * kPitchB) = (uint16_t)(pc[q].y >> 16);
                    }
                } else {
                    unsigned char* d = hb + (rb_ + p * TBi::PER_PASS) * kPitchB + 8 * cb;
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + kImgB) = m;
                    *reinterpret_cast<uint2*>(d + 2 * kImgB) = l;
                }
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < TAi::NV; ++p) {
            const bool ok = !ZF || (TA ? (k0 + ra_ + p * TAi::PER_PASS < K) : (k0 + 4 * ca < K));
            const float4 v = ok ? st.a[p] : zero4;
            if (TA) {
                *reinterpret_cast<float4*>(as + (ra_ + p * TAi::PER_PASS) * SA + 4 * ca) = v;
            } else {
                float* d = as + (4 * ca) * SA + ra_ + p * TAi::PER_PASS;
                d[0] = v.x; d[SA] = v.y; d[2 * SA] = v.z; d[3 * SA] = v.w;
            }
        }
#pragma unroll
        for (int p = 0; p < TBi::NV; ++p) {
            const bool ok = !ZF || (TB ? (k0 + rb_ + p * TBi::PER_PASS < K) : (k0 + 4 * cb < K));
            const float4 v = ok ? st.b[p] : zero4;
            if (TB) {
                *reinterpret_cast<float4*>(bs + (rb_ + p * TBi::PER_PASS) * SB + 4 * cb) = v;
            } else {
                float* d = bs + (4 * cb) * SB + rb_ + p * TBi::PER_PASS;
                d[0] = v.x; d[SB] = v.y; d[2 * SB] = v.z; d[3 * SB] = v.w;
            }
        }
    };

    // ---- MFMA operand reads and accumulators (v_mfma_f32_16x16x4_f32: lane = (row / col li, k residue lk))
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const bool want_colsum = ARITH == 0 && (g.colsum[0] != nullptr) && tn == 0 && wn == 0;  // wave-uniform (ARITH 1: the producers do it)
    float cs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) cs[i] = 0.f;
    auto compute = [&](int buf) {
        if (ARITH == 1) {   // lane (row / col li, k group lk): 8 consecutive k = one 16-byte read per piece and fragment
            const unsigned char* ha = Hs0 + buf * kBufH + (wm * (16 * NI) + li) * kPitchB + 16 * lk;
            const unsigned char* hb = Hs0 + buf * kBufH + 3 * kImgA + (wn * 16 + li) * kPitchB + 16 * lk;
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // two 32-k steps per k-tile
                bf16x8 b[3], a[NI][3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    b[q] = *reinterpret_cast<const bf16x8*>(hb + q * kImgB + 64 * s);
#pragma unroll
                    for (int i = 0; i < NI; ++i) a[i][q] = *reinterpret_cast<const bf16x8*>(ha + q * kImgA + i * 16 * kPitchB + 64 * s);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {   // smallest terms first, hi * hi last (csrc/gemm_bf16.hip)
                    f32x4 c = acc[i];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[0], c, 0, 0, 0);
                    acc[i] = c;
                }
            }
            return;
        }
        const float* ap = As0 + buf * (BK * SA) + lk * SA + wm * (16 * NI) + li;
        const float* bp = Bs0 + buf * (BK * SB) + lk * SB + wn * 16 + li;
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            const float bv = bp[4 * s * SB];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float av = ap[4 * s * SA + 16 * i];
                if (want_colsum) cs[i] += av;
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    };

    // ---- main loop: double-buffered LDS, ONE barrier per k-tile, executed by both kinds of wave (the same count on either path)
    const int nk = (K + BK - 1) / BK;
    auto ktile = [&](int kt) { return kt < nk ? kt : 0; };  // past-the-end prefetches re-read tile 0 (never consumed)
    if (producer) {
        Stage ring[kDepth];   // register ring: tile kt + kDepth is requested while tile kt + 1 is written to LDS
#pragma unroll
        for (int j = 0; j < kDepth; ++j) gload(ktile(j), ring[j]);
        lstore(0, ring[0], 0, true);
        __syncthreads();                                   // tile 0 is in LDS
        for (int kt0 = 0; kt0 < nk; kt0 += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; ++j) {
                const int kt = kt0 + j;
                if (kt < nk) {  // uniform
                    gload(ktile(kt + kDepth), ring[j]);    // slot j's tile (kt) is already in LDS: refill it
                    lstore((kt + 1) & 1, ring[(j + 1) % kDepth], ktile(kt + 1), kt + 1 < nk);  // buffer (kt + 1) & 1 was last read for tile kt - 1
                    __syncthreads();                       // consumers finished tile kt; tile kt + 1 is in LDS
                }
            }
        }
        if (ARITH == 1 && TA && g.colsum[0] != nullptr && tn == 0) {
            // bias gradient: sum_k Aop[m][k] = the fp32 values this thread staged (its k rows, columns 4 ca .. + 3), then the 16 threads of a
            // column group in k-row order -- through LDS buffer 0, which nobody reads any more (one barrier, matched by the consumers below)
            float4* red = reinterpret_cast<float4*>(lds_dyn);   // [16 k-row groups][16 column groups]
            red[ra_ * 16 + ca] = csum;
            __syncthreads();
            if (tid < 16) {
                float4 t = red[tid];
                for (int r = 1; r < 16; ++r) { const float4 u = red[r * 16 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                const int cseg_ = (m0 >= g.seg_c) + (m0 >= 2 * g.seg_c);
                float* out = g.colsum[cseg_] + (m0 - cseg_ * g.seg_c) + 4 * tid;
                const int m = m0 + 4 * tid;
                if (m < M) out[0] = t.x;
                if (m + 1 < M) out[1] = t.y;
                if (m + 2 < M) out[2] = t.z;
                if (m + 3 < M) out[3] = t.w;
            }
        }
        return;   // the epilogue belongs to the consumer waves
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        compute(kt & 1);
        __syncthreads();
    }
    if (ARITH == 1 && TA && g.colsum[0] != nullptr && tn == 0) __syncthreads();   // the producers' bias-gradient reduction (above)

    // ---- epilogue
    const int cseg = (m0 >= g.seg_c) + (m0 >= 2 * g.seg_c);  // a row tile lies inside one C segment (seg_c % 64 == 0, or one segment)
    float* __restrict__ Cb = g.C[cseg];
    const int mloc0 = m0 - cseg * g.seg_c;         // row of the tile inside its segment
    if (want_colsum) {
        float* out = g.colsum[cseg];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float c = cs[i];
            c += __shfl_xor(c, 16, 64);
            c += __shfl_xor(c, 32, 64);
            const int ml = wm * (16 * NI) + 16 * i + li;
            if (lk == 0 && m0 + ml < M) out[mloc0 + ml] = c;
        }
    }
    const int bseg = TB ? 0 : (n0 >= g.seg_b) + (n0 >= 2 * g.seg_b);
    const float* __restrict__ bias = g.bias[bseg];
    const int nb0 = TB ? 0 : bseg * g.seg_b;       // first column of the bias segment
    auto finish = [&](float v, int m, int n) {     // m, n global; m < M and n < N
        if (bias) v += bias[n - nb0];
        if (g.drop_thresh) v = mr::dropout_keep(g.drop_key, (uint32_t)m, (uint32_t)n, g.drop_thresh) ? v * g.drop_inv : 0.f;
        if (g.R) v += g.R[(int64_t)m * g.ldr + n];
        const int64_t ci = (int64_t)(m - cseg * g.seg_c) * g.ldc + n;
        if (g.epi == MR_EPI_GELU_BWD) v *= gelu_grad(g.E[(int64_t)m * g.lde + n]);
        Cb[ci] = v;
        if (g.epi == MR_EPI_GELU_FWD) g.C2[(int64_t)m * g.ldc2 + n] = gelu_erf(v);
    };
    const int n = n0 + wn * 16 + li;
    if (n < N) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (16 * NI) + i * 16 + 4 * lk + r;
                if (m < M) finish(acc[i][r], m, n);
            }
    }
}

}  // namespace

// C = Aop Bop^T with the fused epilogue (header).  Segments: up to three B matrices (trans_b = 0: stacked along N, seg_b columns each -- the
// fused q / k / v projection; trans_b = 1: stacked along K, seg_b rows each -- the input gradient through the stacked projections) and up
// to three C matrices stacked along M (seg_c rows each; with their bias gradients colsum) for trans_a = 1 -- the q / k / v weight gradients
// from one (T, 3 d) dY.  bn: 0 = choose, 32 / 64 = force the tile width (A/B runs).
extern "C" int mr_gemm_tile_f32(const float* A, int64_t lda, int trans_a, const float* b0, const float* b1, const float* b2, int64_t ldb, int trans_b,
                                int nseg_b, int seg_b, const float* bias0, const float* bias1, const float* bias2, int M, int N, int K,
                                const float* R, int64_t ldr, float* c0, float* c1, float* c2, int64_t ldc, int nseg_c, int seg_c, float* colsum0,
                                float* colsum1, float* colsum2, int epi, const float* E, int64_t lde, float* C2, int64_t ldc2, float drop_p,
                                uint32_t drop_key, int products, int bn, mr_stream_t stream) {
    if (!A || !b0 || !c0 || M < 0 || N < 1 || K < 1 || nseg_b < 1 || nseg_b > 3 || nseg_c < 1 || nseg_c > 3) return MR_EINVAL;
    if ((nseg_b > 1 && !b1) || (nseg_b > 2 && !b2) || (nseg_c > 1 && !c1) || (nseg_c > 2 && !c2)) return MR_EINVAL;
    if (epi != MR_EPI_NONE && epi != MR_EPI_GELU_FWD && epi != MR_EPI_GELU_BWD) return MR_EUNSUPPORTED;
    if (products != 0 && products != 6) return MR_EUNSUPPORTED;   // 0 = exact fp32 (the FMA chain), 6 = bf16x6 split precision
    if ((epi == MR_EPI_GELU_FWD && !C2) || (epi == MR_EPI_GELU_BWD && !E)) return MR_EINVAL;
    uint32_t thresh;
    float inv;
    if (!mr::dropout_params(drop_p, &thresh, &inv)) return MR_EINVAL;
    // orientation-specific shape rules
    if ((!trans_a || !trans_b) && (K & 3)) return MR_EUNSUPPORTED;  // k-contiguous rows are read as float4 (zeros past K)
    if (trans_a && (M < 4 || (M & 3))) return MR_EUNSUPPORTED;
    if (trans_b && (N < 4 || (N & 3))) return MR_EUNSUPPORTED;
    if (nseg_b > 1) {
        if (trans_b ? (seg_b % 64 || (int64_t)seg_b * nseg_b != K) : (seg_b % 64 || (int64_t)seg_b * nseg_b != N)) return MR_EUNSUPPORTED;
    } else {
        seg_b = trans_b ? K : N;
    }
    if (nseg_c > 1) {
        if (seg_c % BM || (int64_t)seg_c * nseg_c != M) return MR_EUNSUPPORTED;
    } else {
        seg_c = M > 0 ? M : 1;
    }
    if ((lda & 3) || (ldb & 3) || !mr::aligned16(A) || !mr::aligned16(b0) || (b1 && !mr::aligned16(b1)) || (b2 && !mr::aligned16(b2))) return MR_EALIGN;
    if (bn != 0 && bn != 32 && bn != 64) return MR_EINVAL;
    if (M == 0) return MR_OK;
    const int tiles_m = (M + BM - 1) / BM;
    if (bn == 0) {
        // a launch takes about ceil(workgroups / 256 CUs) x (tile width) of time: the narrow tile when the wide one leaves CUs idle
        const int64_t w64 = (int64_t)tiles_m * ((N + 63) / 64), w32 = (int64_t)tiles_m * ((N + 31) / 32);
        const double t64 = (double)((w64 + 255) / 256) * 1.0, t32 = (double)((w32 + 255) / 256) * 0.55;
        bn = t32 < t64 ? 32 : 64;
        static const int force = [] { const char* e = getenv("MR_GEMM_TILE_BN"); return e ? atoi(e) : 0; }();
        if (force == 32 || force == 64) bn = force;
    }
    Args g;
    g.A = A; g.lda = lda;
    g.B[0] = b0; g.B[1] = b1; g.B[2] = b2; g.ldb = ldb; g.seg_b = seg_b;
    g.bias[0] = bias0; g.bias[1] = bias1; g.bias[2] = bias2;
    g.M = M; g.N = N; g.K = K;
    g.R = R; g.ldr = ldr;
    g.C[0] = c0; g.C[1] = c1; g.C[2] = c2; g.ldc = ldc; g.seg_c = seg_c;
    g.colsum[0] = colsum0; g.colsum[1] = colsum1; g.colsum[2] = colsum2;
    g.epi = epi; g.E = E; g.lde = lde; g.C2 = C2; g.ldc2 = ldc2;
    g.drop_thresh = thresh; g.drop_inv = thresh ? inv : 1.f; g.drop_key = drop_key;
    g.tiles_m = tiles_m; g.tiles_n = (N + bn - 1) / bn;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
    if (nwg > 0x7fffffff) return MR_EUNSUPPORTED;
    g.nwg = (int)nwg;
    hipStream_t st = (hipStream_t)stream;
    // 32-bit element offsets inside the kernel
    if ((int64_t)(trans_a ? K : M) * lda >= (1ll << 31) || (int64_t)(trans_b ? seg_b : N) * ldb >= (1ll << 31)) return MR_EUNSUPPORTED;
    const bool zf = (K % 64) != 0;
    static const int kd_env = [] { const char* e = getenv("MR_GEMM_TILE_KD"); return e ? atoi(e) : 0; }();  // A/B: 2 / 3 / 4 tiles in flight
    const int kd = (kd_env >= 2 && kd_env <= 4) ? kd_env : 3;
#define MR_GT(TA_, TB_, BN_, ZF_, KD_, AR_)                                                                                                      \
    do {                                                                                                                                         \
        constexpr size_t shm_ = AR_ ? (size_t)2 * 3 * kPitchB * (BM + BN_)                                                                        \
                                    : (size_t)2 * 64 * (Tile<TA_, BM, 64, BN_ == 64>::S + Tile<TB_, BN_, 64, BN_ == 64>::S) * sizeof(float);       \
        static mr::DynLdsCeiling lds_ceiling;                                                                                                    \
        if (const int e_ = lds_ceiling.ensure(reinterpret_cast<const void*>(&gemm_tile_kernel<TA_, TB_, BN_, 64, KD_, ZF_, AR_>), shm_)) return e_; \
        hipLaunchKernelGGL((gemm_tile_kernel<TA_, TB_, BN_, 64, KD_, ZF_, AR_>), dim3(g.nwg), dim3(kThreads), shm_, st, g);                        \
    } while (0)
#define MR_GT4(TA_, TB_, BN_, ZF_)                                                                                           \
    do {                                                                                                                     \
        if (products == 6) MR_GT(TA_, TB_, BN_, ZF_, 3, 1);                                                                  \
        else if (kd == 2) MR_GT(TA_, TB_, BN_, ZF_, 2, 0); else if (kd == 3) MR_GT(TA_, TB_, BN_, ZF_, 3, 0); else MR_GT(TA_, TB_, BN_, ZF_, 4, 0); \
    } while (0)
#define MR_GT3(TA_, TB_, BN_) do { if (zf) MR_GT4(TA_, TB_, BN_, true); else MR_GT4(TA_, TB_, BN_, false); } while (0)
    if (bn == 64) {
        if (trans_a) { if (trans_b) MR_GT3(true, true, 64); else MR_GT3(true, false, 64); }
        else { if (trans_b) MR_GT3(false, true, 64); else MR_GT3(false, false, 64); }
    } else {
        if (trans_a) { if (trans_b) MR_GT3(true, true, 32); else MR_GT3(true, false, 32); }
        else { if (trans_b) MR_GT3(false, true, 32); else MR_GT3(false, false, 32); }
    }
#undef MR_GT3
#undef MR_GT4
#undef MR_GT
    return mr::check_launch();
}
