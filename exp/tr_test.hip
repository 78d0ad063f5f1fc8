// semantics check of ds_read_b64_tr_b16 (gfx950): LDS holds element (row r, col c) = 64 r + c as 16-bit, 128-byte rows
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint2* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<unsigned short*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* a = lds + q * 128 + (lane >> 4) * 32 + p * 8;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    out[lane] = __builtin_bit_cast(uint2, v);
}
int main() {
    uint2* d; hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d);
    uint2 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int col = 16 * (l >> 4) + (l & 15);
        const unsigned short e[4] = {(unsigned short)(h[l].x & 0xffff), (unsigned short)(h[l].x >> 16), (unsigned short)(h[l].y & 0xffff), (unsigned short)(h[l].y >> 16)};
        for (int r = 0; r < 4; ++r) if (e[r] != 64 * r + col) ++bad;
        if (l < 20 || bad) printf("lane %2d: %4d %4d %4d %4d (expect col %d of rows 0..3)\n", l, e[0], e[1], e[2], e[3], col);
        if (bad > 8) break;
    }
    printf(bad ? "MISMATCH\n" : "tr16_b64 semantics OK\n");
    return bad != 0;
}
