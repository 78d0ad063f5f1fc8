// DIAGNOSTIC (not part of the library): the library's split-bf16 GEMM compiled with phase-timing hooks.
// Each workgroup's wave 0 accumulates s_memtime deltas: 0 prologue, 1 compute (LDS fragment reads + MFMA issue),
// 2 split + LDS store + global prefetch issue, 3 barrier wait, 4 epilogue issue.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imergerec_amd/csrc -o exp/gemm_phases exp/gemm_phases.hip
//        ... -DPH_CLOCK_ONLY -o exp/gemm_clock ...   (in-kernel clock of the unperturbed kernel: profiles/r03_inkernel_clock.txt)
#include <hip/hip_runtime.h>
__device__ unsigned long long g_ph[16384 * 8];
__device__ unsigned long long g_rt[16384 * 2];
#define MR_PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long ph_t0 = ph_t; const unsigned long long ph_r0 = __builtin_amdgcn_s_memrealtime();
#ifdef PH_CLOCK_ONLY  /* r03: no phase stamps -- only the workgroup's first / last s_memtime + s_memrealtime: the UNPERTURBED library kernel's clock */
#define MR_PH(i)
#define MR_PH_WAITLOADS(i)
#define MR_PH_LAST ph_t = __builtin_amdgcn_s_memtime();
#else
#define MR_PH_LAST
#define MR_PH(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += n_ - ph_t; ph_t = n_; }
#define MR_PH_WAITLOADS(i) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); MR_PH(i) }  /* NT=4, NP=2: the newer stage's 6 loads stay in flight */
#endif
#define MR_PH_FLUSH(pid) MR_PH_LAST if (threadIdx.x == 0 && (pid) < 16384) { for (int z = 0; z < 6; ++z) g_ph[(pid) * 8 + z] = ph_acc[z]; g_ph[(pid) * 8 + 6] = ph_t0; g_ph[(pid) * 8 + 7] = ph_t; g_rt[(pid) * 2] = ph_r0; g_rt[(pid) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
#include "../mergerec_amd/csrc/gemm_bf16.hip"
#include "../mergerec_amd/csrc/capi.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536;
    const int products = argc > 2 ? atoi(argv[2]) : 3;
    {
        int nb = -1;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<0, false, true, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_nt_bf16x6_kernel<0, false, true, 2, 4>, 256, 73728));
        printf("occupancy: NP=2 NT=4 PF2: %d workgroups/CU at 72 KB LDS\n", nb);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<0, false, true, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_nt_bf16x6_kernel<0, false, true, 2, 2>, 256, 49152));
        printf("occupancy: NP=2 NT=2 PF2: %d workgroups/CU at 48 KB LDS\n", nb);
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<0, false, true, 2, 4>)));
        printf("NT=4: numRegs %d, sharedSizeBytes %zu, localSizeBytes %zu, maxThreadsPerBlock %d\n", fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock);
    }
    struct Shape { const char* name; int N, K; } shapes[] = {{"out", 768, 768}, {"ffn1", 3072, 768}, {"ffn2", 768, 3072}};
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& x : hW) x = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
        float *dA, *dW, *dC, *db;
        uint16_t *dh, *dm, *dl;
        int64_t *dtab, *dpre;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dh, hW.size() * 2)); CK(hipMalloc(&dm, hW.size() * 2)); CK(hipMalloc(&dl, hW.size() * 2));
        CK(hipMalloc(&dtab, 24)); CK(hipMalloc(&dpre, 16));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(db, 0, N * 4));
        int64_t tab[3] = {0, N, K}, pre[2] = {0, (int64_t)N * K / 4};
        CK(hipMemcpy(dtab, tab, 24, hipMemcpyHostToDevice)); CK(hipMemcpy(dpre, pre, 16, hipMemcpyHostToDevice));
        if (mr_split_weights_kblock_f32(dW, dtab, dpre, 1, pre[1], dh, dm, dl, 0)) { printf("split failed\n"); return 1; }
        auto launch = [&] {
            int rc = mr_gemm_nt_bf16x6_f32(dA, K, dh, dm, dl, 0, 0, 0, db, nullptr, nullptr, 1, M, N, K, 0, nullptr, 0, dC, N, products, 0);
            if (rc) { printf("gemm rc %d\n", rc); exit(1); }
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const int nwg = ((M + 127) / 128) * (N / 256);
        std::vector<unsigned long long> ph((size_t)16384 * 8);
        CK(hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_ph), ph.size() * 8));
        const int n = std::min(nwg, 16384);
        double sum[6] = {0, 0, 0, 0, 0, 0}, tot = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < n; ++w) {
            for (int z = 0; z < 6; ++z) sum[z] += (double)ph[w * 8 + z];
            tot += (double)(ph[w * 8 + 7] - ph[w * 8 + 6]);
            tmin = std::min(tmin, ph[w * 8 + 6]); tmax = std::max(tmax, ph[w * 8 + 7]);
        }
        std::vector<unsigned long long> rt((size_t)16384 * 2);
        CK(hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(g_rt), rt.size() * 8));
        double cs = 0, rs = 0;  // per workgroup (the counters of different XCDs are not aligned): sum of both deltas
        for (int w = 0; w < n; ++w) { cs += (double)(ph[w * 8 + 7] - ph[w * 8 + 6]); rs += (double)(rt[w * 2 + 1] - rt[w * 2]); }
        printf("      in-kernel clock: s_memtime / s_memrealtime (100 MHz) over all workgroups = %.3f GHz; mean workgroup life %.1f us\n", cs / rs * 0.1, rs / n / 100.0);
        const int nk = K / 16;
        printf("%-5s M=%d N=%d K=%d products=%d: %.3f ms (%.1f TF alg), %d WGs; kernel span %.0f cyc; per WG avg %.0f cyc = prologue %.0f + loop[compute %.0f + stage %.0f + barrier %.0f] + epilogue %.0f;  per k-tile: compute %.0f split+ldswrite %.0f gload-issue %.0f barrier %.0f\n",
               sh.name, M, N, K, products, ms, 2.0 * M * N * K / ms / 1e9, nwg, (double)(tmax - tmin), tot / n, sum[0] / n, sum[1] / n, sum[2] / n,
               sum[3] / n, sum[4] / n, sum[1] / n / nk, sum[5] / n / nk, sum[2] / n / nk, sum[3] / n / nk);
        hipFree(dA); hipFree(dW); hipFree(dC); hipFree(db); hipFree(dh); hipFree(dm); hipFree(dl); hipFree(dtab); hipFree(dpre);
    }
    return 0;
}
