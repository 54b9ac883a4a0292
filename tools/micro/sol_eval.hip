// Developer microbenchmark (not part of the product): the speed of light of the evaluation's BYTES.
// Two launches that move exactly what prep_kernel / pair_kernel have to move at 2 x 800 x 1024 images, 32 instances
// (39.3 MB per evaluation) and do no arithmetic beyond keeping the loads alive:
//   sol_prep : read the images (19.66 MB) and the mask logits (6.55 MB), write the zero-filled gradient (6.55 MB) and Lab (1.23 MB)
//   sol_pair : read the box tiles' logits + 3 Lab planes with their halo (611 tiles x 12 rows x 64 columns x 4 planes = 7.5 MB),
//              write the gradient tiles (611 x 8 x 60 x 4 B = 1.2 MB)
// 8 input sets are rotated (312 MB > the 256 MB Infinity Cache), as bench.py does.  Printed: us per launch pair, each launch
// alone, and two empty launches -- the floor the real kernels (9.3 + 9.9 us at the end of round 2) are to be read against.
// Build: hipcc --offload-arch=gfx950 -O3 -o sol_eval sol_eval.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kB = 2, kH = 800, kW = 1024, kN = 32, kh = 200, kw = 256;
constexpr long kImg4 = (long)kB * 3 * kH * kW / 4;      // float4 counts
constexpr long kLog4 = (long)kN * kh * kw / 4;
constexpr long kLab4 = (long)kB * 3 * kh * kw / 4;
constexpr int kTiles = 611;

// one pass, every load of a thread in flight before the first use (what the real kernel's roles do)
__global__ __launch_bounds__(256) void sol_prep(const float4* __restrict__ img, const float4* __restrict__ logit, float4* __restrict__ g,
                                                float4* __restrict__ lab) {
    const long T = (long)gridDim.x * blockDim.x, t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 a[4], l[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = t + k * T < kImg4 ? img[t + k * T] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) l[k] = t + k * T < kLog4 ? logit[t + k * T] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (t + k * T < kLog4) g[t + k * T] = make_float4(0, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += a[k].x + a[k].y + a[k].z + a[k].w;
#pragma unroll
    for (int k = 0; k < 2; ++k) s += fmaxf(fmaxf(l[k].x, l[k].y), fmaxf(l[k].z, l[k].w));
    if (t < kLab4) lab[t] = make_float4(s, s, s, s);
}

// one wave per tile: 12 rows x 64 columns of 4 planes in, 8 rows x 60 columns out
__global__ __launch_bounds__(256) void sol_pair(const float* __restrict__ logit, const float* __restrict__ lab, float* __restrict__ g) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= kTiles) return;
    const int n = tile % kN, tr = (tile / kN) % 20, tc = (tile / (kN * 20)) % 3;     // spread over instances, rows, columns
    const int r0 = 8 * tr + 2, c0 = 60 * tc + 2;
    const long P = (long)kh * kw;
    float v[4][12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const long o = (long)(r0 - 2 + j) * kw + c0 - 2 + lane;
        v[0][j] = logit[(long)n * P + o];
#pragma unroll
        for (int p = 0; p < 3; ++p) v[1 + p][j] = lab[((long)(n & 1) * 3 + p) * P + o];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) s += v[p][j] + v[p][j + 2] + v[p][j + 4];
        if (lane >= 2 && lane < 62) g[(long)n * P + (long)(r0 + j) * kw + c0 - 2 + lane] = s;
    }
}

__global__ void empty_kernel() {}

int main() {
    const int kSets = 8;
    std::vector<float*> img(kSets), logit(kSets), g(kSets), lab(kSets);
    for (int i = 0; i < kSets; ++i) {
        CK(hipMalloc(&img[i], kImg4 * 16)); CK(hipMalloc(&logit[i], kLog4 * 16)); CK(hipMalloc(&g[i], kLog4 * 16)); CK(hipMalloc(&lab[i], kLab4 * 16));
        CK(hipMemset(img[i], 0, kImg4 * 16)); CK(hipMemset(logit[i], 0, kLog4 * 16)); CK(hipMemset(lab[i], 0, kLab4 * 16));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](int mode, int grid1) -> float {
        for (int rep = 0; rep < 2; ++rep) {      // first repetition warms up
            hipEventRecord(e0, 0);
            for (int it = 0; it < iters; ++it) {
                const int s = it % kSets;
                if (mode == 0 || mode == 1)
                    hipLaunchKernelGGL(sol_prep, dim3(grid1), dim3(256), 0, 0, (const float4*)img[s], (const float4*)logit[s], (float4*)g[s], (float4*)lab[s]);
                if (mode == 0 || mode == 2)
                    hipLaunchKernelGGL(sol_pair, dim3((kTiles + 3) / 4), dim3(256), 0, 0, (const float*)logit[s], (const float*)lab[s], g[s]);
                if (mode == 3) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / iters;
    };
    const double prep_bytes = (kImg4 + 2 * kLog4 + kLab4) * 16.0, pair_bytes = kTiles * (12 * 64 * 4 * 4.0 + 8 * 60 * 4.0);
    printf("bytes: sol_prep %.2f MB, sol_pair %.2f MB\n", prep_bytes / 1e6, pair_bytes / 1e6);
    for (int grid1 : {1280, 2560}) {
        const float both = run(0, grid1), p = run(1, grid1), q = run(2, grid1);
        printf("grid %d: sol_prep + sol_pair %.2f us per evaluation | sol_prep alone %.2f us (%.2f TB/s) | sol_pair alone %.2f us (%.2f TB/s)\n", grid1, both, p,
               prep_bytes / p / 1e6, q, pair_bytes / q / 1e6);
    }
    printf("two empty launches: %.2f us\n", run(3, 0));
    CK(hipDeviceSynchronize());
    return 0;
}
