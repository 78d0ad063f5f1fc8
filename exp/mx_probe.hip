// EXPERIMENT: operand and scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands) -- not documented in the guides at hand.
// Exact small-integer data under a layout hypothesis, one wave, compared with a host reference.
//   HYP 0: lane l, byte b of the 32-byte operand  <->  A[row l & 31][k = 32 (l >> 5) + b]  (and B[k][col l & 31]); the lane's scale byte
//          (opsel picks it from the 32-bit scale register) applies to exactly those 32 elements.
//   HYP 1: k = 16 (l >> 5) + (b & 15) + 32 (b >> 4)   (two K = 32 halves interleaved), the lane's scale acting on its own 32 elements
//   HYP 2: data as HYP 1, but the scale of lane (row, h) acts on the contiguous block k in [32 h, 32 h + 32) = bytes 16 h .. 16 h + 15 of BOTH lanes of the row
// build: hipcc --offload-arch=gfx950 -O3 -o exp/mx_probe exp/mx_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OPSEL>
__global__ void probe(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb,
                      float* __restrict__ d) {
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[l * 8 + i]; vb[i] = b[l * 8 + i]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, OPSEL, sa[l], OPSEL, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

static uint8_t fp8_of_int(int v) {  // e4m3 encoding of small non-negative integers 0..8 (exact)
    static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
    return t[v];
}

int main() {
    srand(7);
    static int A[32][64], B[64][32], SA[32][2], SB[2][32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = rand() % 4;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = rand() % 4;
    for (int i = 0; i < 32; ++i) for (int h = 0; h < 2; ++h) { SA[i][h] = 125 + rand() % 5; SB[h][i] = 125 + rand() % 5; }
    uint32_t *da, *db, *dsa, *dsb; float* dd;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 16 * 4));
    for (int hyp = 0; hyp < 3; ++hyp)
        for (int scaled = 0; scaled < 2; ++scaled)
            for (int opsel = 0; opsel < 4; opsel += 3) {
                uint8_t ha[64][32], hb[64][32]; uint32_t hsa[64], hsb[64];
                for (int l = 0; l < 64; ++l) {
                    const int rc = l & 31, h = l >> 5;
                    for (int bt = 0; bt < 32; ++bt) {
                        const int k = hyp == 0 ? 32 * h + bt : 16 * h + (bt & 15) + 32 * (bt >> 4);  // HYP 2: data as HYP 1, scale block = k >> 5 from lane (row, k >> 5)
                        ha[l][bt] = fp8_of_int(A[rc][k]);
                        hb[l][bt] = fp8_of_int(B[k][rc]);
                    }
                    // the scale byte of this lane's block in byte `opsel` of the register, garbage (0x55) in the other bytes
                    const int ea = scaled ? SA[rc][h] : 127, eb = scaled ? SB[h][rc] : 127;
                    hsa[l] = 0x55555555u; hsb[l] = 0x55555555u;
                    hsa[l] = (hsa[l] & ~(0xffu << (8 * opsel))) | ((uint32_t)ea << (8 * opsel));
                    hsb[l] = (hsb[l] & ~(0xffu << (8 * opsel))) | ((uint32_t)eb << (8 * opsel));
                }
                CK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
                CK(hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice));
                if (opsel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
                else hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
                CK(hipDeviceSynchronize());
                float hd[64][16];
                CK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
                // C/D layout (dtype-independent): lane l, reg r -> col l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)
                int bad = 0; double worst = 0;
                for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                    const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) {
                        const int h = k >> 5;  // block index under HYP 0; under HYP 1 a lane's block is k in {16h..16h+15} u {32+16h..}
                        const int hb_ = hyp == 1 ? ((k >> 4) & 1) : h;
                        const double sa_ = scaled ? ldexp(1.0, SA[i][hb_] - 127) : 1.0, sb_ = scaled ? ldexp(1.0, SB[hb_][j] - 127) : 1.0;
                        ref += A[i][k] * sa_ * B[k][j] * sb_;
                    }
                    const double dlt = fabs(ref - hd[l][r]);
                    if (dlt > 1e-6 * (1 + fabs(ref))) ++bad;
                    if (dlt > worst) worst = dlt;
                }
                printf("HYP %d  %s  opsel %d: %d of 1024 outputs differ (max abs diff %.4g)\n", hyp, scaled ? "per-block scales" : "unit scales", opsel, bad, worst);
            }
    return 0;
}
