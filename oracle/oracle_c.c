/* oracle_c.c -- plain C restatement of the bit-exact pieces of the path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/ref_cpu.py header): linked/loaded only by tests/, smoke() and bench.py's cpu_baseline.
 *
 * What is bit-exact and why:
 *   - gemm_nt_ref: every output element is ONE fp32 fmaf chain over ascending k from 0, which is what
 *     v_mfma_f32_32x32x2_f32 computes when k-steps are issued in ascending order (the product GEMM
 *     does).  Mathematically this is the reference's `user @ item.T` (module/recommender/module.py:137)
 *     and F.linear inside the encoder; torch's own CPU GEMM blocks k differently, so torch agrees to
 *     ~1e-6, this function agrees to the bit.
 *   - merge_nway_ref: torch CPU order of `base + (alpha[:,None] * T).sum(0)`
 *     (merger/weight_learning/module/task_wise.py:43-47): rounded products, sequential sum from 0.
 *   - topk_rows_ref: evaluator/evaluator.py:43 with the canonical tie order (score desc, index asc,
 *     NaN first).
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void gemm_nt_ref(const float* A, int64_t lda, const float* W, const float* bias, int M, int N, int K, float* C,
                 int64_t ldc) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* a = A + (int64_t)m * lda;
        for (int n = 0; n < N; ++n) {
            const float* w = W + (int64_t)n * K;
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(a[k], w[k], acc);
            C[(int64_t)m * ldc + n] = bias ? acc + bias[n] : acc;
        }
    }
}

void merge_nway_ref(const float* base, const float* tv, int64_t tv_stride, const float* alpha, const int64_t* seg_off,
                    int N, int S, int64_t P, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        int s = 0;
        if (seg_off) {
            while (s + 1 < S && seg_off[s + 1] <= p) ++s;
        }
        float acc = 0.0f;
        for (int i = 0; i < N; ++i) {
            const float prod = alpha[(int64_t)s * N + i] * tv[(int64_t)i * tv_stride + p];
            acc = acc + prod;
        }
        out[p] = base[p] + acc;
    }
}

static int nan_first_desc(float a, int64_t ia, float b, int64_t ib) {
    /* returns 1 if (a, ia) ranks before (b, ib) */
    const int na = isnan(a), nb = isnan(b);
    if (na != nb) return na;
    if (!na && a != b) return a > b;
    return ia < ib;
}

void topk_rows_ref(const float* scores, int64_t ld, int nrows, int ncols, int k, float* top_val, int64_t* top_idx) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < nrows; ++r) {
        const float* s = scores + (int64_t)r * ld;
        float* tv = top_val + (int64_t)r * k;
        int64_t* ti = top_idx + (int64_t)r * k;
        int have = 0;
        for (int c = 0; c < ncols; ++c) { /* insertion into a sorted list of <= k */
            int pos = have;
            while (pos > 0 && nan_first_desc(s[c], c, tv[pos - 1], ti[pos - 1])) --pos;
            if (pos >= k) continue;
            const int last = have < k ? have : k - 1;
            for (int j = last; j > pos; --j) { tv[j] = tv[j - 1]; ti[j] = ti[j - 1]; }
            tv[pos] = s[c];
            ti[pos] = c;
            if (have < k) ++have;
        }
    }
}
