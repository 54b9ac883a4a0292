// Developer microbenchmark (not part of the product): launch floor, dependent-load latency,
// shader clock and streaming bandwidth in the regime of the BoxInst loss kernels (tiny, cold).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_scalar2(const float* a, const float* b, float* o) { if (*a == 2.f && *b == 3.f) o[0] = 1.f; }
__global__ void k_chase(const uint32_t* p, int n, uint32_t start, uint32_t* out, long long* cyc, long long* wall) {
    uint32_t i = start;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int k = 0; k < n; ++k) i = p[i];
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = i; cyc[0] = c1 - c0; wall[0] = w1 - w0;
}
__global__ void k_spin(int n, float* out, long long* cyc, long long* wall) {
    float x = threadIdx.x;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int k = 0; k < n; ++k) x = x * 1.0001f + 0.5f;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
    if (x == 123.456f) out[0] = x;
}
__global__ void k_read4(const float4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
template <int U>
__global__ void k_read4u(const float4* __restrict__ p, size_t n4, float* out) {   // U loads in flight per thread
    size_t i = ((size_t)blockIdx.x * blockDim.x) * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + (size_t)u * blockDim.x < n4) ? p[i + (size_t)u * blockDim.x] : make_float4(0, 0, 0, 0);
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_write4(float4* p, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(0, 0, 0, 0);
}

// pool_rgb access pattern without the arithmetic: lane = one 4x4x3 window of a [B,3,H,W] f32 image
template <int BLOCK>
__global__ void k_poolpat(const float* __restrict__ img, int B, int H, int W, float* out) {
    const int h = H / 4, w = W / 4;
    const long long o = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= (long long)B * h * w) return;
    const int c = o % w, r = (o / w) % h, b = o / ((long long)w * h);
    const long long plane = (long long)H * W;
    const float* base = img + (long long)b * 3 * plane + (long long)(4 * r) * W + 4 * c;
    float4 v[12];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ch * 4 + i] = *reinterpret_cast<const float4*>(base + ch * plane + (long long)i * W);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (acc == 123.456f) out[0] = acc;
}
// streaming-tile pattern: block = 4 waves, wave reads RW rows of 256 floats (1 KiB) of a [N,h,256] map, writes zeros
template <int RW, bool WRITE>
__global__ void k_streampat(const float* __restrict__ L, float* __restrict__ G, int N, int h, float* out) {
    const int T = (h + 4 * RW - 1) / (4 * RW);
    const int n = blockIdx.x / T, t = blockIdx.x % T, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = t * 4 * RW;
    float4 v[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) { const int r = r0 + wv + 4 * i; v[i] = r < h ? *reinterpret_cast<const float4*>(L + ((long long)n * h + r) * 256 + lane * 4) : make_float4(0, 0, 0, 0); }
    if (WRITE) {
#pragma unroll
        for (int i = 0; i < RW; ++i) { const int r = r0 + wv + 4 * i; if (r < h) *reinterpret_cast<float4*>(G + ((long long)n * h + r) * 256 + lane * 4) = make_float4(0, 0, 0, 0); }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < RW; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (acc == 123.456f) out[0] = acc;
}

template <typename F> float time_us(F f, int reps, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f(i);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s CUs %d clock %d kHz memclk %d kHz wallclock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate, 100000);
    float* o; CK(hipMalloc(&o, 256));
    long long *cyc, *wall; CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&wall, 8));
    float h2[2] = {1.f, 1.f}; float* sc; CK(hipMalloc(&sc, 8)); CK(hipMemcpy(sc, h2, 8, hipMemcpyHostToDevice));
    printf("empty kernel 1 block      : %.2f us/launch (back-to-back)\n", time_us([&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); }, 200, s));
    printf("empty kernel 256x256      : %.2f us/launch\n", time_us([&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 0, s); }, 200, s));
    printf("empty kernel 3200x256     : %.2f us/launch\n", time_us([&](int) { hipLaunchKernelGGL(k_empty, 3200, 256, 0, s); }, 200, s));
    printf("empty kernel 3200x256 lds22k: %.2f us/launch\n", time_us([&](int) { hipLaunchKernelGGL(k_empty, 3200, 256, 22000, s); }, 200, s));
    printf("2 scalar loads 128x256    : %.2f us/launch\n", time_us([&](int) { hipLaunchKernelGGL(k_scalar2, 128, 256, 0, s, sc, sc + 1, o); }, 200, s));
    // clock under a busy loop
    {
        hipLaunchKernelGGL(k_spin, 1024, 256, 0, s, 200000, o, cyc, wall); hipStreamSynchronize(s);
        long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
        printf("busy loop: %lld shader cycles in %lld wall ticks (100 MHz) -> %.0f MHz\n", c, w, (double)c / ((double)w / 100.0));
    }
    // pointer chase over 512 MB (cold) and 64 KB (L2-warm)
    for (size_t bytes : {(size_t)512 << 20, (size_t)64 << 10}) {
        size_t n = bytes / 4; std::vector<uint32_t> h(n);
        // stride permutation: jump by a large odd stride modulo n (n power of two)
        uint32_t stride = (uint32_t)((n / 2 + 12345) | 1); for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)((i + stride) & (n - 1));
        uint32_t* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
        uint32_t* out; CK(hipMalloc(&out, 4));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_chase, 1, 1, 0, s, d, 256, (uint32_t)(rep * 977), out, cyc, wall); hipStreamSynchronize(s);
            long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
            printf("chase %zu KB rep %d: %.0f cycles/load, %.0f ns/load, clock %.0f MHz\n", bytes >> 10, rep, c / 256.0, w * 10.0 / 256.0, (double)c / (w / 100.0));
        }
        hipFree(d); hipFree(out);
    }
    // streaming: 20 MB and 640 MB (rotating 32 x 20 MB)
    {
        const size_t one = 20u << 20; const int nbuf = 16; char* big; CK(hipMalloc(&big, one * nbuf)); CK(hipMemset(big, 1, one * nbuf));
        size_t n4 = one / 16;
        for (int blocks : {256, 1024, 2048, 4096}) {
            float t = time_us([&](int i) { hipLaunchKernelGGL(k_read4, blocks, 256, 0, s, (const float4*)(big + one * (i % nbuf)), n4, o); }, 160, s);
            printf("read 20MB cold grid-stride blocks=%d: %.2f us -> %.0f GB/s\n", blocks, t, one / t / 1e3);
        }
        {
            float t = time_us([&](int i) { hipLaunchKernelGGL(k_read4u<4>, (unsigned)(n4 / (256 * 4)), 256, 0, s, (const float4*)(big + one * (i % nbuf)), n4, o); }, 160, s);
            printf("read 20MB cold 4 loads/thread one-shot: %.2f us -> %.0f GB/s\n", t, one / t / 1e3);
            t = time_us([&](int i) { hipLaunchKernelGGL(k_read4u<8>, (unsigned)(n4 / (256 * 8)), 256, 0, s, (const float4*)(big + one * (i % nbuf)), n4, o); }, 160, s);
            printf("read 20MB cold 8 loads/thread one-shot: %.2f us -> %.0f GB/s\n", t, one / t / 1e3);
            t = time_us([&](int i) { hipLaunchKernelGGL(k_read4u<1>, (unsigned)(n4 / 256), 256, 0, s, (const float4*)(big + one * (i % nbuf)), n4, o); }, 160, s);
            printf("read 20MB cold 1 load/thread one-shot: %.2f us -> %.0f GB/s\n", t, one / t / 1e3);
            t = time_us([&](int i) { hipLaunchKernelGGL(k_read4u<4>, (unsigned)(n4 / (256 * 4)), 256, 0, s, (const float4*)(big), n4, o); }, 160, s);
            printf("read 20MB WARM 4 loads/thread one-shot: %.2f us -> %.0f GB/s\n", t, one / t / 1e3);
            t = time_us([&](int i) { hipLaunchKernelGGL(k_write4, (unsigned)(n4 / 256), 256, 0, s, (float4*)(big + one * (i % nbuf)), n4); }, 160, s);
            printf("write 20MB rotating: %.2f us -> %.0f GB/s\n", t, one / t / 1e3);
            size_t n4b = one * nbuf / 16;
            t = time_us([&](int i) { hipLaunchKernelGGL(k_read4, 4096, 256, 0, s, (const float4*)big, n4b, o); }, 20, s);
            printf("read 320MB grid-stride 4096 blocks: %.2f us -> %.0f GB/s\n", t, one * nbuf / t / 1e3);
        }
        hipFree(big);
    }
    // ---- access patterns of stage1 (2x3x800x1024 image, 32x200x256 logits), rotating over 8 cold sets ----
    {
        const int nset = 8; const size_t img_b = (size_t)2 * 3 * 800 * 1024 * 4, log_b = (size_t)32 * 200 * 256 * 4;
        char *imgs, *logs, *grads; CK(hipMalloc(&imgs, img_b * nset)); CK(hipMalloc(&logs, log_b * nset)); CK(hipMalloc(&grads, log_b * nset));
        CK(hipMemset(imgs, 1, img_b * nset)); CK(hipMemset(logs, 1, log_b * nset));
        float t;
        t = time_us([&](int i) { hipLaunchKernelGGL((k_poolpat<256>), 400, 256, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, o); }, 160, s);
        printf("pool pattern 19.7MB, 400x256 threads : %.2f us -> %.0f GB/s\n", t, img_b / t / 1e3);
        t = time_us([&](int i) { hipLaunchKernelGGL((k_poolpat<64>), 1600, 64, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, o); }, 160, s);
        printf("pool pattern 19.7MB, 1600x64 threads : %.2f us -> %.0f GB/s\n", t, img_b / t / 1e3);
        t = time_us([&](int i) { hipLaunchKernelGGL((k_poolpat<128>), 800, 128, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, o); }, 160, s);
        printf("pool pattern 19.7MB, 800x128 threads : %.2f us -> %.0f GB/s\n", t, img_b / t / 1e3);
        t = time_us([&](int i) { hipLaunchKernelGGL((k_streampat<4, false>), 32 * 13, 256, 0, s, (const float*)(logs + log_b * (i % nset)), (float*)(grads + log_b * (i % nset)), 32, 200, o); }, 160, s);
        printf("stream pattern read 6.5MB (16-row tiles): %.2f us -> %.0f GB/s\n", t, log_b / t / 1e3);
        t = time_us([&](int i) { hipLaunchKernelGGL((k_streampat<4, true>), 32 * 13, 256, 0, s, (const float*)(logs + log_b * (i % nset)), (float*)(grads + log_b * (i % nset)), 32, 200, o); }, 160, s);
        printf("stream pattern read+write 13MB (16-row tiles): %.2f us -> %.0f GB/s\n", t, 2 * log_b / t / 1e3);
        t = time_us([&](int i) { hipLaunchKernelGGL((k_streampat<2, true>), 32 * 25, 256, 0, s, (const float*)(logs + log_b * (i % nset)), (float*)(grads + log_b * (i % nset)), 32, 200, o); }, 160, s);
        printf("stream pattern read+write 13MB (8-row tiles): %.2f us -> %.0f GB/s\n", t, 2 * log_b / t / 1e3);
        // both back to back on two streams is what one fused launch does; emulate with sequential launches
        t = time_us([&](int i) { hipLaunchKernelGGL((k_poolpat<256>), 400, 256, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, o);
                                 hipLaunchKernelGGL((k_streampat<4, true>), 32 * 13, 256, 0, s, (const float*)(logs + log_b * (i % nset)), (float*)(grads + log_b * (i % nset)), 32, 200, o); }, 160, s);
        printf("pool + stream as two launches: %.2f us\n", t);
    }
    return 0;
}
