#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* x, const float* y, float* out) {
    // A[32 x 16] row i = lane&31, k = 8*(lane>>5) + j ; all rows equal x[k]; B same with y
    h8 a, b;
    const int kb = 8 * (threadIdx.x >> 5);
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)x[kb + j]; b[j] = (_Float16)y[kb + j]; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float hx[16], hy[16];
    for (int i = 0; i < 16; ++i) { hx[i] = 0; hy[i] = 0; }
    hx[0] = 3e-6f;  hy[0] = 1024.0f;      // fp16 denormal times normal
    float *x, *y, *o; hipMalloc(&x, 64); hipMalloc(&y, 64); hipMalloc(&o, 4);
    hipMemcpy(x, hx, 64, hipMemcpyHostToDevice); hipMemcpy(y, hy, 64, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, y, o);
    float r; hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    printf("denormal a * 1024 = %g (expect ~%g if denormals kept, 0 if flushed)\n", r, (double)(float)(_Float16)3e-6f * 1024.0);
    return 0;
}
