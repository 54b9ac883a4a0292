// Developer microbenchmark: cycles per element of the de-normalise chain and of its parts (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int denorm_u8(float x, double s, double m) {
    const float t = (float)((double)x * s);
    const float v = (float)((double)t + m);
    return (int)v & 0xff;
}
template <int MODE>
__global__ void k(const float* in, int* out, long long* cyc, int iters, double s, double m) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[threadIdx.x * 16 + i];
    int acc = 0; double dacc = 0; float facc = 0;
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) acc += denorm_u8(x[i] + (float)it, s, m);
            if (MODE == 1) dacc += (double)(x[i] + (float)it);                       // cvt f32->f64 + add f64
            if (MODE == 2) dacc = dacc * s + (double)it;                              // fma f64 (dependent)
            if (MODE == 3) facc += (float)((double)(x[i] + (float)it) * s);          // cvt, mul, cvt
            if (MODE == 4) { double v = (double)(x[i] + (float)it) * s; dacc += cbrt(v + 1.0); }
        }
    }
    long long c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc + (int)dacc + (int)facc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
int main() {
    float* in; int* out; long long* cyc; hipMalloc(&in, 64 * 16 * 4 * 1024); hipMalloc(&out, 4 * 64 * 4096); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 64 * 16 * 4 * 1024);
    const int iters = 200; long long c;
    const char* names[] = {"denorm_u8 (full chain)", "cvt f32->f64 + add f64", "fma f64 dependent", "cvt,mul f64,cvt", "cvt,mul,cbrt f64,add"};
    for (int waves : {1, 4}) {
        printf("-- %d wave(s) per SIMD (blocks of 64 threads x %d per CU)\n", waves, waves * 4);
#define RUN(M) hipLaunchKernelGGL(k<M>, 256 * 4 * waves, 64, 0, 0, in, out, cyc, iters, 58.395, 123.675); hipDeviceSynchronize(); \
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %.1f cycles per element per wave\n", names[M], (double)c / (iters * 16));
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    }
    return 0;
}
