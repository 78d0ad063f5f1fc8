// DIAGNOSTIC (r03): the library's register-resident row select with s_memtime stamps of thread 0 per workgroup:
// 0 load + key conversion, 1 four radix passes, 2 collection, 3 sort + top-k write + label rank, 4 log-sum-exp.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imergerec_amd/csrc -o exp/topk_phases exp/topk_phases.hip -L mergerec_amd/lib -lmergerec_hip -Wl,-rpath,$PWD/mergerec_amd/lib
#include <hip/hip_runtime.h>
__device__ unsigned long long g_tk[4096 * 8];
#define MR_TK_DECL unsigned long long tk_t = __builtin_amdgcn_s_memtime(), tk_acc[5] = {0, 0, 0, 0, 0};
#define MR_TK(i) { __syncthreads(); unsigned long long n_ = __builtin_amdgcn_s_memtime(); tk_acc[i] += n_ - tk_t; tk_t = n_; }
#define MR_TK_FLUSH(row) if (threadIdx.x == 0 && (row) < 4096) { for (int z = 0; z < 5; ++z) g_tk[(row) * 8 + z] = tk_acc[z]; }
#include "../mergerec_amd/csrc/score.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 256, C = argc > 2 ? atoi(argv[2]) : 22855, k = 50, ld = (C + 3) / 4 * 4;
    std::vector<float> h((size_t)R * ld);
    for (auto& x : h) x = 0.5f + 0.3f * ((float)rand() / (float)RAND_MAX - 0.5f);
    std::vector<int64_t> lab(R, 7);
    float *ds, *dv, *dl, *dlab; int64_t *di, *dlb; int32_t* dr;
    CK(hipMalloc(&ds, h.size() * 4)); CK(hipMalloc(&dv, R * k * 4)); CK(hipMalloc(&di, R * k * 8)); CK(hipMalloc(&dl, R * 4)); CK(hipMalloc(&dlab, R * 4));
    CK(hipMalloc(&dlb, R * 8)); CK(hipMalloc(&dr, R * 4));
    CK(hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlb, lab.data(), R * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) if (mr_topk_rows_f32(ds, ld, R, C, k, dv, di, dlb, 20.f, dl, dlab, dr, 0)) { printf("rc\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0)); mr_topk_rows_f32(ds, ld, R, C, k, dv, di, dlb, 20.f, dl, dlab, dr, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> ph((size_t)4096 * 8);
    CK(hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_tk), ph.size() * 8));
    double s[5] = {0, 0, 0, 0, 0};
    const int n = R < 4096 ? R : 4096;
    for (int w = 0; w < n; ++w) for (int z = 0; z < 5; ++z) s[z] += (double)ph[w * 8 + z];
    printf("rows %d x cols %d: %.1f us (instrumented); cycles per workgroup: load+keys %.0f  radix passes %.0f  collect %.0f  sort+write %.0f  log-sum-exp %.0f\n", R, C,
           ms * 1e3, s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n);
    return 0;
}
