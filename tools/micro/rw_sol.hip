// Developer microbenchmark (not part of the product): what this part does for the op-level pairwise_nlog kernels' BYTES.
//   write52 : 6.55 MB read + 52.4 MB written (the forward's traffic: logits in, 8 planes out), 16-byte accesses, contiguous per wave
//   read59  : 59 MB read + 6.55 MB written (the backward's traffic)
// with 8 rotating buffer sets (cold), timed back to back like tools/bench_pairwise_op.py.  Grid variants: one pass per thread (everything
// in flight at once) at several workgroup counts.
// Build: hipcc --offload-arch=gfx950 -O3 -o rw_sol rw_sol.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr long kP4 = 32L * 200 * 256 / 4;      // float4 per plane set (one logits tensor)

template <int PER>
__global__ __launch_bounds__(256) void write52(const float4* __restrict__ in, float4* __restrict__ out) {
    const long T = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long t = t0; t < kP4; t += T * PER) {
        float4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = t + k * T < kP4 ? in[t + k * T] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (t + k * T < kP4)
#pragma unroll
                for (int p = 0; p < 8; ++p) out[p * kP4 + t + k * T] = make_float4(v[k].x + p, v[k].y, v[k].z, v[k].w);
    }
}
template <int PER>
__global__ __launch_bounds__(256) void read59(const float4* __restrict__ in, const float4* __restrict__ planes, float4* __restrict__ out) {
    const long T = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long t = t0; t < kP4; t += T * PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const long i = t + k * T;
            if (i < kP4) {
                float4 a = in[i], s = make_float4(0, 0, 0, 0);
                float4 q[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) q[p] = planes[p * kP4 + i];
#pragma unroll
                for (int p = 0; p < 8; ++p) { s.x += q[p].x; s.y += q[p].y; s.z += q[p].z; s.w += q[p].w; }
                out[i] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
            }
        }
    }
}
int main() {
    const int SETS = 8;
    std::vector<float4*> in(SETS), pl(SETS), out(SETS);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&in[s], kP4 * 16)); CK(hipMalloc(&pl[s], 8 * kP4 * 16)); CK(hipMalloc(&out[s], kP4 * 16));
        CK(hipMemset(in[s], 0, kP4 * 16)); CK(hipMemset(pl[s], 0, 8 * kP4 * 16));
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch, double mb) {
        for (int i = 0; i < 20; ++i) launch(i % SETS);
        hipDeviceSynchronize();
        hipEventRecord(a);
        const int n = 200;
        for (int i = 0; i < n; ++i) launch(i % SETS);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %.2f us  (%.2f TB/s on %.1f MB)\n", name, ms / n * 1e3, mb / (ms / n * 1e-3) / 1e6, mb);
        return 0;
    };
    for (int grid : {1280, 1664, 1792, 2560, 3328, 6400}) {
        char nm[96];
        snprintf(nm, 96, "write52 grid %d x1", grid);
        run(nm, [&](int s) { hipLaunchKernelGGL(write52<1>, dim3(grid), dim3(256), 0, 0, in[s], pl[s]); }, 59.0);
        snprintf(nm, 96, "read59  grid %d x1", grid);
        run(nm, [&](int s) { hipLaunchKernelGGL(read59<1>, dim3(grid), dim3(256), 0, 0, in[s], pl[s], out[s]); }, 65.5);
    }
    run("write52 grid 1600 x4 (everything in flight)", [&](int s) { hipLaunchKernelGGL(write52<4>, dim3(1600), dim3(256), 0, 0, in[s], pl[s]); }, 59.0);
    run("read59  grid 1600 x1 (one pass)", [&](int s) { hipLaunchKernelGGL(read59<1>, dim3(1600), dim3(256), 0, 0, in[s], pl[s], out[s]); }, 65.5);
    return 0;
}
