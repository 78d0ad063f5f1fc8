// DIAGNOSTIC (r03): the library's work-list split attention kernel with s_memtime phase stamps (thread 0 of each workgroup):
// 0 prologue, 1 scores (K fragment reads + MFMAs issued, both query tiles), 2 softmax + P split (both tiles), 3 P V, 4 split + LDS store of
// the next tile + issue of the loads two tiles ahead, 5 barrier.  Uniform-length batch, bf16x3 (256-row blocks, two query tiles per wave).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imergerec_amd/csrc -o exp/attn_work_phases exp/attn_work_phases.hip
#include <hip/hip_runtime.h>
__device__ unsigned long long g_ph[65536 * 8];
__device__ unsigned long long g_rt[65536 * 4];
#define MR_PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long ph_t0 = ph_t, ph_r0 = __builtin_amdgcn_s_memrealtime();
#define MR_PH(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += n_ - ph_t; ph_t = n_; }
#define MR_PH_FLUSH(pid) if (threadIdx.x == 0 && (pid) < 65536) { for (int z = 0; z < 6; ++z) g_ph[(pid) * 8 + z] = ph_acc[z]; g_rt[(pid) * 4] = ph_t0; g_rt[(pid) * 4 + 1] = ph_t; g_rt[(pid) * 4 + 2] = ph_r0; g_rt[(pid) * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#include "../mergerec_amd/csrc/attn_bf16.hip"
#include "../mergerec_amd/csrc/capi.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 128, H = 12;
    const int T = B * L;
    std::vector<float> h((size_t)T * 3 * H * 64);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    std::vector<int> cu(B + 1);
    std::vector<int64_t> lens(B, L);
    for (int b = 0; b <= B; ++b) cu[b] = b * L;
    const int64_t ns = mr_attn_work_plan(lens.data(), B, 256, nullptr, 0);
    std::vector<int32_t> work((size_t)ns * 8);
    mr_attn_work_plan(lens.data(), B, 256, work.data(), (int64_t)work.size());
    float *dq, *dc; int *dcu, *dw;
    CK(hipMalloc(&dq, h.size() * 4)); CK(hipMalloc(&dc, (size_t)T * H * 64 * 4)); CK(hipMalloc(&dcu, (B + 1) * 4)); CK(hipMalloc(&dw, work.size() * 4));
    CK(hipMemcpy(dq, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, work.data(), work.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) if (mr_attn_split_work_f32(dq, dcu, dw, ns, B, 256, H, 64, 0.125f, -1, 3, dc, 0)) { printf("rc\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0)); mr_attn_split_work_f32(dq, dcu, dw, ns, B, 256, H, 64, 0.125f, -1, 3, dc, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nwg = (int)(ns * 8 * H), n = nwg < 65536 ? nwg : 65536;
    std::vector<unsigned long long> ph((size_t)65536 * 8), rt((size_t)65536 * 4);
    CK(hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_ph), ph.size() * 8));
    CK(hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(g_rt), rt.size() * 8));
    double s[6] = {0, 0, 0, 0, 0, 0}, cs = 0, rs = 0;
    int live = 0;
    for (int w = 0; w < n; ++w) {
        if (rt[w * 4 + 1] == 0) continue;  // padding entries leave no record
        ++live;
        for (int z = 0; z < 6; ++z) s[z] += (double)ph[w * 8 + z];
        cs += (double)(rt[w * 4 + 1] - rt[w * 4]); rs += (double)(rt[w * 4 + 3] - rt[w * 4 + 2]);
    }
    printf("in-kernel clock (s_memtime / s_memrealtime @100 MHz): %.3f GHz; mean workgroup life %.1f us; %d workgroups\n", cs / rs * 0.1, rs / live / 100.0, live);
    const double nt = (double)((L + 31) / 32);
    printf("L=%d B=%d: %.3f ms (instrumented), %.1f TFLOP/s algorithmic; per workgroup: prologue %.0f; per key tile (48 MFMAs = 1536 pipe cycles per wave): scores %.0f  softmax+Psplit %.0f  PV %.0f  stage-next %.0f  barrier %.0f  = %.0f  (s_memtime ticks; each stamp costs ~200)\n",
           L, B, ms, 4.0 * 768 * (double)B * L * L / ms / 1e9, s[0] / live, s[1] / live / nt, s[2] / live / nt, s[3] / live / nt, s[4] / live / nt, s[5] / live / nt,
           (s[1] + s[2] + s[3] + s[4] + s[5]) / live / nt);
    return 0;
}
