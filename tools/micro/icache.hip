// Developer microbenchmark: what does straight-line (fully unrolled) code cost on its first execution after a launch?
// The same arithmetic as a rolled loop (I$ hits after the first trip) and unrolled N times (every line cold).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N, bool UNROLL>
__global__ __launch_bounds__(64) void k_code(float* out, long long* dur, float seed) {
    float a = seed + threadIdx.x, b = seed * 2.f, c = 1.f, d = 0.5f;
    const long long t0 = wall_clock64();
    if (UNROLL) {
#pragma unroll
        for (int i = 0; i < N; ++i) {      // 8 VALU per trip, constants differ per trip so nothing folds
            a = a * 1.0001f + (float)i; b = b * a + 0.25f; c = c * 0.999f + b; d = d + c * (float)(i + 1);
            a = a - d * 1e-9f; b = b - a * 1e-9f; c = c - b * 1e-9f; d = d - c * 1e-9f;
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
            a = a * 1.0001f + (float)i; b = b * a + 0.25f; c = c * 0.999f + b; d = d + c * (float)(i + 1);
            a = a - d * 1e-9f; b = b - a * 1e-9f; c = c - b * 1e-9f; d = d - c * 1e-9f;
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) dur[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 123.456f) out[0] = a;
}

template <int N, bool U>
int run(const char* label, int grid, float* o, long long* d_t, hipStream_t s) {
    std::vector<long long> h(grid);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k_code<N, U>), grid, 64, 0, s, o, d_t, 1.f + rep); CK(hipStreamSynchronize(s)); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL((k_code<N, U>), grid, 64, 0, s, o, d_t, 1.f + rep);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    CK(hipMemcpy(h.data(), d_t, 8 * grid, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-34s grid %5d: kernel %.2f us | per-wave body: min %.2f med %.2f max %.2f us\n", label, grid, ms * 1e3 / 50, h[0] * 0.01, h[grid / 2] * 0.01,
           h[grid - 1] * 0.01);
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* o; CK(hipMalloc(&o, 256)); long long* d_t; CK(hipMalloc(&d_t, 8 * 8192));
    for (int grid : {256, 1024, 2048}) {
        run<128, false>("128 trips rolled   (~ 64 B code)", grid, o, d_t, s);
        run<128, true>("128 trips unrolled (~ 4 KB code)", grid, o, d_t, s);
        run<512, false>("512 trips rolled", grid, o, d_t, s);
        run<512, true>("512 trips unrolled (~16 KB code)", grid, o, d_t, s);
        run<1024, false>("1024 trips rolled", grid, o, d_t, s);
        run<1024, true>("1024 trips unrolled (~32 KB code)", grid, o, d_t, s);
    }
    return 0;
}
