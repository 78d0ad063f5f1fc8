// EXPERIMENT: which (row, k-block) does lane l's scale byte act on in v_mfma_scale_f32_32x32x64_f8f6f4?  A = B = all ones (fp8 1.0), unit scales
// everywhere except ONE (lane, byte) of the A-scale register set to 2^1: D[i][j] = 64 + (number of k whose A element got the doubled scale).
// build: hipcc --offload-arch=gfx950 -O3 -o exp/mx_probe2 exp/mx_probe2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OPSEL>
__global__ void probe(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb, float* __restrict__ d) {
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = 0x38383838; vb[i] = 0x38383838; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, OPSEL, sa[l], OPSEL, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

int main() {
    uint32_t *dsa, *dsb; float* dd;
    CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 16 * 4));
    for (int opsel = 0; opsel < 4; ++opsel)
        for (int byte = 0; byte < 4; ++byte)
            for (int which = 0; which < 2; ++which) {  // 0: perturb an A scale, 1: perturb a B scale
                printf("opsel %d, perturbed byte %d of the %c scale register:\n", opsel, byte, which ? 'B' : 'A');
                for (int la = 0; la < 64; ++la) {
                    uint32_t hsa[64], hsb[64];
                    for (int l = 0; l < 64; ++l) { hsa[l] = 0x7f7f7f7fu; hsb[l] = 0x7f7f7f7fu; }
                    uint32_t* tgt = which ? hsb : hsa;
                    tgt[la] = (tgt[la] & ~(0xffu << (8 * byte))) | (128u << (8 * byte));
                    CK(hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice));
                    switch (opsel) {
                        case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dsa, dsb, dd); break;
                        case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dsa, dsb, dd); break;
                        case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dsa, dsb, dd); break;
                        default: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dsa, dsb, dd); break;
                    }
                    CK(hipDeviceSynchronize());
                    float hd[64][16];
                    CK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
                    // summarise: which rows / cols deviate from 64 and by how much
                    int nrow = 0, ncol = 0, first = -1; float val = 0;
                    bool rowhit[32] = {false}, colhit[32] = {false};
                    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                        const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                        if (hd[l][r] != 64.f) { rowhit[i] = true; colhit[j] = true; val = hd[l][r]; }
                    }
                    for (int i = 0; i < 32; ++i) { if (rowhit[i]) { ++nrow; if (first < 0) first = i; } if (colhit[i]) ++ncol; }
                    if (la < 4 || (la >= 30 && la < 36) || la >= 62)
                        printf("   lane %2d: %2d rows x %2d cols changed (first %s %d), value %.0f\n", la, nrow, ncol, which ? "col-major idx" : "row", which ? -1 : first, val);
                }
            }
    return 0;
}
