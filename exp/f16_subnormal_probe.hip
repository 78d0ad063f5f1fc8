// Does the f16 matrix pipe of gfx950 honour fp16 SUBNORMAL inputs, and does v_cvt_pk_f16_f32 produce them?  (Decides whether a two-piece
// fp16 split -- x = hi + lo, lo ~ 2^-11 |x| -- keeps its low piece for |x| < 2^-3 without any scaling.)
//   hipcc --offload-arch=gfx950 -O3 -o exp/f16_subnormal_probe exp/f16_subnormal_probe.hip && exp/f16_subnormal_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void split(const float* x, float* hi, float* lo, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i];
    _Float16 h = (_Float16)a;
    float r = a - (float)h;
    _Float16 l = (_Float16)r;
    hi[i] = (float)h; lo[i] = (float)l;
}
__global__ void mm(float aval, float bval, float* c) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)bval; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) c[0] = acc[0];
}
int main() {
    float *c; hipMalloc(&c, 4);
    const float vals[] = {ldexpf(1.f, -20), ldexpf(1.f, -24), ldexpf(3.f, -16), 1.0f};
    for (float v : vals) {
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, v, 1.0f, c);
        float h; hipMemcpy(&h, c, 4, hipMemcpyDeviceToHost);
        printf("mfma f16: 16 x (%.6e * 1) = %.6e   expected %.6e  %s\n", v, h, 16 * v, h == 16 * v ? "EXACT (subnormal inputs honoured)" : "DIFFERENT");
    }
    const int n = 8; float hx[n] = {0.1f, 0.01f, 1e-3f, 3.1415926f, 150.123f, 2e-5f, 6e-8f, 0.0625f + 1e-6f};
    float *x, *hi, *lo; hipMalloc(&x, 4 * n); hipMalloc(&hi, 4 * n); hipMalloc(&lo, 4 * n);
    hipMemcpy(x, hx, 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split, dim3(1), dim3(64), 0, 0, x, hi, lo, n);
    float hh[n], hl[n]; hipMemcpy(hh, hi, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hl, lo, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("x %.9e = hi %.9e + lo %.9e ; residual %.3e (relative %.2e = 2^%.1f)\n", hx[i], hh[i], hl[i], hx[i] - hh[i] - hl[i],
                                       fabs((hx[i] - hh[i] - hl[i]) / hx[i]), log2(fabs((hx[i] - hh[i] - hl[i]) / hx[i]) + 1e-300));
    return 0;
}
