// DIAGNOSTIC: the library's split attention kernel with s_memtime phase stamps (wave 0 of each workgroup):
// 0 prologue, 1 scores (K fragment reads + 12 MFMAs issued), 2 softmax + P split, 3 P V (tr reads + 12 MFMAs issued), 4 split + LDS store of
// the next tile (includes the wait for its global loads), 5 barrier.  Uniform-length batch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imergerec_amd/csrc -o exp/attn_phases exp/attn_phases.hip
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
    const int L = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 256, H = 12;
    const int T = B * L;
    std::vector<float> h((size_t)T * 3 * H * 64);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    std::vector<int> cu(B + 1);
    for (int b = 0; b <= B; ++b) cu[b] = b * L;
    float *dq, *dc; int* dcu;
    CK(hipMalloc(&dq, h.size() * 4)); CK(hipMalloc(&dc, (size_t)T * H * 64 * 4)); CK(hipMalloc(&dcu, (B + 1) * 4));
    CK(hipMemcpy(dq, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) if (mr_attn_split_f32(dq, dcu, nullptr, B, H, 64, L, 0.125f, -1, 3, dc, 0)) { printf("rc\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0)); mr_attn_split_f32(dq, dcu, nullptr, B, H, 64, L, 0.125f, -1, 3, dc, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nwg = ((L + 127) / 128) * H * B, n = nwg < 65536 ? nwg : 65536;
    std::vector<unsigned long long> ph((size_t)65536 * 8);
    CK(hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_ph), ph.size() * 8));
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < n; ++w) for (int z = 0; z < 6; ++z) s[z] += (double)ph[w * 8 + z];
    std::vector<unsigned long long> rt((size_t)65536 * 4);
    CK(hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(g_rt), rt.size() * 8));
    double cs = 0, rs = 0;
    for (int w = 0; w < n; ++w) { cs += (double)(rt[w * 4 + 1] - rt[w * 4]); rs += (double)(rt[w * 4 + 3] - rt[w * 4 + 2]); }
    printf("in-kernel clock (s_memtime / s_memrealtime @100 MHz): %.3f GHz; mean workgroup life %.1f us\n", cs / rs * 0.1, rs / n / 100.0);
    const double nt = (double)((L + 31) / 32);
    printf("L=%d B=%d: %.3f ms (instrumented); per workgroup: prologue %.0f; per key tile: scores %.0f  softmax+Psplit %.0f  PV %.0f  stage-next %.0f  barrier %.0f  (s_memtime ticks; each stamp costs ~200)\n",
           L, B, ms, s[0] / n, s[1] / n / nt, s[2] / n / nt, s[3] / n / nt, s[4] / n / nt, s[5] / n / nt);
    return 0;
}
