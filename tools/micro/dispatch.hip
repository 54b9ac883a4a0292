// Developer microbenchmark (not part of the product): what the fused BoxInst evaluation's design hinges on.
//   A. workgroup dispatch rate vs workgroup size / LDS / VGPR footprint
//   B. producer -> consumer flag latency inside one launch (release / acquire at agent scope, across XCDs)
//   C. arrival counter: N waves add + wait for all (the "global count" protocol)
//   D. load-then-ALU kernels: all-resident one-shot waves vs 4x lighter waves vs persistent waves with prefetch
// Build: hipcc --offload-arch=gfx950 -O3 -o dispatch dispatch.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void spin_us(int ticks) {   // 100 MHz ticks
    const long long t0 = wall_clock64();
    while ((int)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
}

// ---- A ------------------------------------------------------------------------------------------
template <int BS, bool BIGV>
__global__ __launch_bounds__(BS) void k_disp(long long* start, float* o) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) start[blockIdx.x] = wall_clock64();
    if (BIGV) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    spin_us(600);                                                   // stay resident 6 us
    if (o && threadIdx.x == 9999) o[0] = sm[threadIdx.x];
}

// ---- B ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_flag(unsigned* flag, float* data, long long* t_seen, long long* t_prod, float* o) {
    if (blockIdx.x == 0) {
        spin_us(300);
        if (threadIdx.x < 64) data[threadIdx.x] = 3.f + threadIdx.x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            t_prod[0] = wall_clock64();
            __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        int it = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && it < 100000) { __builtin_amdgcn_s_sleep(1); ++it; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float v = data[threadIdx.x];
        if (threadIdx.x == 0) t_seen[blockIdx.x] = wall_clock64();
        if (v != 3.f + threadIdx.x) o[1] = -1.f;     // stale data would be a protocol bug
    }
}

// ---- C ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_arrive(unsigned* cnt, unsigned long long* acc, long long* t_arr, long long* t_done, int expect) {
    spin_us(200 + (int)(blockIdx.x % 7) * 10);     // arrivals spread over ~0.6 us
    if (threadIdx.x == 0) {
        atomicAdd(acc, (unsigned long long)(blockIdx.x + 1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_arr[blockIdx.x] = wall_clock64();
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)expect && it < 100000) { __builtin_amdgcn_s_sleep(1); ++it; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned long long a = __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t_done[blockIdx.x] = wall_clock64() | ((long long)(a == (unsigned long long)expect * (expect + 1) / 2 ? 0 : 1) << 62);
    }
}

// ---- D ------------------------------------------------------------------------------------------
// pool pattern: item = 64 pooled pixels of a [B,3,H,W] image, a lane loads 3 ch x 4 rows x float4, then `alu` dependent FMAs (fp64)
__device__ __forceinline__ double alu_chain(double x, int n) {
    for (int k = 0; k < n; ++k) x = __builtin_fma(x, 1.0000001, 0.5);
    return x;
}
__global__ __launch_bounds__(64) void k_pool_oneshot(const float* __restrict__ img, int B, int H, int W, int alu, float* out, long long* tl) {
    const int h = H / 4, w = W / 4;
    const long long o = (long long)blockIdx.x * 64 + threadIdx.x;
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
    if (tl && threadIdx.x == 0) tl[blockIdx.x] = wall_clock64();
    const double y = alu_chain((double)acc, alu);
    if (y == 123.456) out[0] = (float)y;
}
// quad layout: item = 16 pooled pixels x 4 input rows; a lane loads 3 ch x 1 float4, alu/4 chain
__global__ __launch_bounds__(64) void k_pool_quad(const float* __restrict__ img, int B, int H, int W, int alu, float* out, long long* tl) {
    const int h = H / 4, w = W / 4;
    const long long o = (long long)blockIdx.x * 16 + (threadIdx.x >> 2);
    const int i = threadIdx.x & 3;
    const int c = o % w, r = (o / w) % h, b = o / ((long long)w * h);
    const long long plane = (long long)H * W;
    const float* base = img + (long long)b * 3 * plane + (long long)(4 * r + i) * W + 4 * c;
    float4 v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = *reinterpret_cast<const float4*>(base + ch * plane);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    if (tl && threadIdx.x == 0) tl[blockIdx.x] = wall_clock64();
    const double y = alu_chain((double)acc, alu / 4);
    if (y == 123.456) out[0] = (float)y;
}
// row layout: item = 64 consecutive float4 of ONE input row (1 KiB per channel), 3 loads per lane; alu/4 chain
__global__ __launch_bounds__(64) void k_pool_row(const float* __restrict__ img, int B, int H, int W, int alu, float* out, long long* tl) {
    const int segs = W / 256;
    const long long o = blockIdx.x;
    const int sg = o % segs; const int y = (o / segs) % H; const int b = o / ((long long)segs * H);
    const long long plane = (long long)H * W;
    const float* base = img + (long long)b * 3 * plane + (long long)y * W + sg * 256 + threadIdx.x * 4;
    float4 v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = *reinterpret_cast<const float4*>(base + ch * plane);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    if (tl && threadIdx.x == 0) tl[blockIdx.x] = wall_clock64();
    const double yv = alu_chain((double)acc, alu / 4);
    if (yv == 123.456) out[0] = (float)yv;
}
// persistent: `waves` one-wave workgroups loop over the quad items with the next item's loads in flight
__global__ __launch_bounds__(64) void k_pool_persist(const float* __restrict__ img, int B, int H, int W, int alu, int n_items, float* out) {
    const int h = H / 4, w = W / 4;
    const long long plane = (long long)H * W;
    const int i = threadIdx.x & 3;
    auto addr = [&](long long item) {
        const long long o = item * 16 + (threadIdx.x >> 2);
        const int c = o % w, r = (o / w) % h, b = o / ((long long)w * h);
        return img + (long long)b * 3 * plane + (long long)(4 * r + i) * W + 4 * c;
    };
    float4 cur[3], nxt[3];
    long long it = blockIdx.x;
    if (it < n_items) { const float* p = addr(it);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) cur[ch] = *reinterpret_cast<const float4*>(p + ch * plane); }
    double tot = 0.0;
    for (; it < n_items; it += gridDim.x) {
        const long long nx = it + gridDim.x;
        if (nx < n_items) { const float* p = addr(nx);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) nxt[ch] = *reinterpret_cast<const float4*>(p + ch * plane); }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += cur[k].x + cur[k].y + cur[k].z + cur[k].w;
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
        tot += alu_chain((double)acc, alu / 4);
#pragma unroll
        for (int k = 0; k < 3; ++k) cur[k] = nxt[k];
    }
    if (tot == 123.456) out[0] = (float)tot;
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

static void quant(const char* label, std::vector<long long> v, long long t0) {
    std::sort(v.begin(), v.end());
    auto q = [&](double f) { return (v[(size_t)(f * (v.size() - 1))] - t0) * 0.01; };
    printf("%s n=%zu  us after t0: min %.2f q25 %.2f q50 %.2f q75 %.2f q95 %.2f max %.2f\n", label, v.size(), q(0), q(.25), q(.5), q(.75), q(.95), q(1));
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* o; CK(hipMalloc(&o, 256));
    long long* d_t; CK(hipMalloc(&d_t, 8 * 16384)); long long* d_t2; CK(hipMalloc(&d_t2, 8 * 16384));
    std::vector<long long> ht(16384), ht2(16384);
    // pre-roll so clocks are up
    { long long *cy; CK(hipMalloc(&cy, 16)); hipLaunchKernelGGL((k_disp<256, false>), 2048, 256, 0, s, d_t, o); hipStreamSynchronize(s); }

    printf("== A. dispatch: start time of the last workgroup relative to the first\n");
    auto runA = [&](const char* label, auto kern, int grid, int bs, int lds) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, grid, bs, lds, s, d_t, o); hipStreamSynchronize(s); }
        hipMemcpy(ht.data(), d_t, 8 * grid, hipMemcpyDeviceToHost);
        std::vector<long long> v(ht.begin(), ht.begin() + grid);
        long long t0 = *std::min_element(v.begin(), v.end());
        char buf[128]; snprintf(buf, sizeof buf, "%-28s grid %5d", label, grid);
        quant(buf, v, t0);
    };
    for (int grid : {512, 2048, 4096}) runA("64 thr, no lds", k_disp<64, false>, grid, 64, 0);
    for (int grid : {512, 2048}) runA("64 thr, 2 KB lds", k_disp<64, false>, grid, 64, 2048);
    for (int grid : {512, 2048}) runA("64 thr, 17 KB lds", k_disp<64, false>, grid, 64, 17408);
    for (int grid : {512, 2048}) runA("64 thr, >=121 vgpr", k_disp<64, true>, grid, 64, 0);
    for (int grid : {184, 512, 1056}) runA("256 thr, no lds", k_disp<256, false>, grid, 256, 0);
    for (int grid : {184, 512, 1056}) runA("256 thr, 17 KB lds", k_disp<256, false>, grid, 256, 17408);
    for (int grid : {184, 512, 1056}) runA("256 thr, >=121 vgpr", k_disp<256, true>, grid, 256, 0);
    for (int grid : {64, 256, 512}) runA("1024 thr, no lds", k_disp<1024, false>, grid, 1024, 0);

    printf("== B. flag latency inside a launch (producer = block 0, 511 consumer waves)\n");
    {
        unsigned* flag; float* data; CK(hipMalloc(&flag, 256)); CK(hipMalloc(&data, 256));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flag, 0, 256, s)); CK(hipMemsetAsync(o, 0, 256, s));
            hipLaunchKernelGGL(k_flag, 512, 64, 0, s, flag, data, d_t, d_t2, o); CK(hipStreamSynchronize(s));
            hipMemcpy(ht.data(), d_t, 8 * 512, hipMemcpyDeviceToHost); hipMemcpy(ht2.data(), d_t2, 8, hipMemcpyDeviceToHost);
            float ho[2]; hipMemcpy(ho, o, 8, hipMemcpyDeviceToHost);
            std::vector<long long> v(ht.begin() + 1, ht.begin() + 512);
            quant(ho[1] < 0 ? "flag seen + data read (STALE DATA!)" : "flag seen + data read", v, ht2[0]);
        }
    }
    printf("== C. arrival counter (N waves add into a u64, arrive, wait for all, read the sum)\n");
    for (int n : {152, 606, 1212}) {
        unsigned* cnt; unsigned long long* acc; CK(hipMalloc(&cnt, 256)); CK(hipMalloc(&acc, 256));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(cnt, 0, 256, s)); CK(hipMemsetAsync(acc, 0, 256, s));
            hipLaunchKernelGGL(k_arrive, n, 64, 0, s, cnt, acc, d_t, d_t2, n); CK(hipStreamSynchronize(s));
        }
        hipMemcpy(ht.data(), d_t, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(ht2.data(), d_t2, 8 * n, hipMemcpyDeviceToHost);
        long long last_arr = *std::max_element(ht.begin(), ht.begin() + n);
        int bad = 0; std::vector<long long> v(n);
        for (int i = 0; i < n; ++i) { bad += (int)((ht2[i] >> 62) & 1); v[i] = ht2[i] & ((1ll << 62) - 1); }
        char buf[96]; snprintf(buf, sizeof buf, "N=%d wrong sums %d; all-seen after last arrival", n, bad);
        quant(buf, v, last_arr);
    }
    printf("== D. 2x3x800x1024 image, load + fp64 ALU chain of `alu` FMAs per pooled pixel (8 cold sets)\n");
    {
        const int nset = 8; const size_t img_b = (size_t)2 * 3 * 800 * 1024 * 4;
        char* imgs; CK(hipMalloc(&imgs, img_b * nset)); CK(hipMemset(imgs, 1, img_b * nset));
        const int n_pix = 2 * 200 * 256;
        for (int alu : {0, 150, 300, 600}) {
            float t1 = time_us([&](int i) { hipLaunchKernelGGL(k_pool_oneshot, n_pix / 64, 64, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, alu, o, (long long*)nullptr); }, 80, s);
            float t2 = time_us([&](int i) { hipLaunchKernelGGL(k_pool_quad, n_pix / 16, 64, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, alu, o, (long long*)nullptr); }, 80, s);
            float t3 = time_us([&](int i) { hipLaunchKernelGGL(k_pool_row, 2 * 800 * 4, 64, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, alu, o, (long long*)nullptr); }, 80, s);
            printf("alu %4d: one-shot(1600 waves) %.2f us | quad(6400 waves) %.2f us | row(6400 waves) %.2f us", alu, t1, t2, t3);
            for (int waves : {1024, 2048, 4096}) {
                float t4 = time_us([&](int i) { hipLaunchKernelGGL(k_pool_persist, waves, 64, 0, s, (const float*)(imgs + img_b * (i % nset)), 2, 800, 1024, alu, n_pix / 16, o); }, 80, s);
                printf(" | persist(%d) %.2f us", waves, t4);
            }
            printf("\n");
        }
        // when does the data of each wave arrive (one-shot)?
        hipLaunchKernelGGL(k_pool_oneshot, n_pix / 64, 64, 0, s, (const float*)(imgs + img_b * 3), 2, 800, 1024, 0, o, d_t); hipStreamSynchronize(s);
        hipMemcpy(ht.data(), d_t, 8 * (n_pix / 64), hipMemcpyDeviceToHost);
        { std::vector<long long> v(ht.begin(), ht.begin() + n_pix / 64); long long t0 = *std::min_element(v.begin(), v.end());
          quant("one-shot: data-arrival time per wave", v, t0);
          double corr = 0; for (int i = 0; i < (int)v.size(); ++i) corr += (double)(v[i] - t0) * (i - 800.0); printf("   (covariance with wave index %.1f; > 0 = later waves later)\n", corr / v.size() / 800.0); }
        hipLaunchKernelGGL(k_pool_quad, n_pix / 16, 64, 0, s, (const float*)(imgs + img_b * 5), 2, 800, 1024, 0, o, d_t); hipStreamSynchronize(s);
        hipMemcpy(ht.data(), d_t, 8 * (n_pix / 16), hipMemcpyDeviceToHost);
        { std::vector<long long> v(ht.begin(), ht.begin() + n_pix / 16); long long t0 = *std::min_element(v.begin(), v.end());
          quant("quad: data-arrival time per wave", v, t0); }
    }
    return 0;
}
