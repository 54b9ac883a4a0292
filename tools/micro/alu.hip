// Developer microbenchmark: cycles per element of the de-normalise chain and of its parts (one or four waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int denorm_old(float x, double s, double m) {
    const float t = (float)((double)x * s);
    const float v = (float)((double)t + m);
    return (int)v & 0xff;
}
__device__ __forceinline__ int denorm_u8(float x, double s, float m) {
    const float t = (float)((double)x * s);
    const float v = __fadd_rn(t, m);
    return (int)v & 0xff;
}
// RN32(x * s) for a double s = sh + sl (+ residual), in f32 arithmetic with a doubt flag for near-ties
__device__ __forceinline__ float mul_f32_by_f64(float x, float sh, float sl, bool& doubt) {
    const float p = x * sh;
    const float e = __fmaf_rn(x, sh, -p);
    const float r = __fmaf_rn(x, sl, e);
    const float t = __fadd_rn(p, r);
    const float rem = __fsub_rn(r, __fsub_rn(t, p));                 // rounding error of the add
    const float ulp = __builtin_amdgcn_ldexpf(1.0f, __builtin_amdgcn_frexp_expf(t) - 24);
    doubt = fabsf(rem) * 2.00001f >= ulp;
    return t;
}
template <int MODE>
__global__ void k(const float* in, int* out, long long* cyc, int iters, double s, double m) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[threadIdx.x * 16 + i];
    int acc = 0; double dacc = 0; float facc = 0;
    const float mf = (float)m, sh = (float)s, sl = (float)(s - (double)sh);
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        bool any_doubt = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) acc += denorm_old(x[i] + (float)it, s, m);
            if (MODE == 1) dacc += (double)(x[i] + (float)it);                       // cvt f32->f64 + add f64
            if (MODE == 2) dacc = dacc * s + (double)it;                              // fma f64 (dependent)
            if (MODE == 3) facc += (float)((double)(x[i] + (float)it) * s);          // cvt, mul, cvt
            if (MODE == 4) { double v = (double)(x[i] + (float)it) * s; dacc += cbrt(v + 1.0); }
            if (MODE == 5) acc += denorm_u8(x[i] + (float)it, s, mf);
            if (MODE == 6) { bool d; const float t = mul_f32_by_f64(x[i] + (float)it, sh, sl, d); any_doubt |= d;
                             acc += (int)__fadd_rn(t, mf) & 0xff; }
            if (MODE == 7) { double v = (double)(x[i] + (float)it) * s; dacc += v / 0.95047; }
            if (MODE == 8) { facc += (x[i] + (float)it) * sh; }
        }
        if (MODE == 6 && __any(any_doubt)) acc += 1000;
    }
    long long c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc + (int)dacc + (int)facc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
int main() {
    float* in; int* out; long long* cyc; hipMalloc(&in, 64 * 16 * 4 * 1024); hipMalloc(&out, 4 * 64 * 4096); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 64 * 16 * 4 * 1024);
    const int iters = 200; long long c;
    const char* names[] = {"old denorm (f64 add)", "cvt f32->f64 + add f64", "fma f64 dependent", "cvt,mul f64,cvt", "cvt,mul,cbrt f64,add",
                           "denorm_u8 (f32 add)", "denorm, f32-emulated mul", "cvt,mul,div f64,add", "add,mul f32,add (floor)"};
    for (int waves : {1, 4}) {
        printf("-- %d wave(s) per SIMD (blocks of 64 threads x %d per CU)\n", waves, waves * 4);
#define RUN(M) hipLaunchKernelGGL(k<M>, 256 * 4 * waves, 64, 0, 0, in, out, cyc, iters, 58.395, 123.675); hipDeviceSynchronize(); \
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %.1f cycles per element per wave\n", names[M], (double)c / (iters * 16));
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    }
    return 0;
}
