// Developer microbenchmark (not part of the product): the op-level pairwise_nlog backward kernels of csrc/pairwise_op.hip, stand-alone.
//   * "wide" (every pixel evaluates its eight taps) against "pair" (every unordered pair once), 32 x 200 x 256, dilation 2, 8 cold input sets,
//     timed back to back like tools/bench_pairwise_op.py, outputs compared with each other;
//   * one traced launch of each: per-workgroup phase stamps (100 MHz wall clock) -> when the loads are out, the logits in, the passes done.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBXI_PW_TRACE -mllvm -amdgpu-kernarg-preload-count=16 -o pw_bwd pw_bwd.hip   [-D variants]
#include "../../boxinstseg_amd/csrc/pairwise_op.hip"
#include "pw_bwd_wide_ref.inc"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace bxi {
void set_last_hip_error(int) {}
std::atomic<bxi_launch_hook> g_hook{nullptr};
std::atomic<void*> g_hook_user{nullptr};
}
// the backward's bytes moved by a copy: 8 planes + logits read, one plane written (16-byte accesses, one pass)
__global__ __launch_bounds__(256) void copy59(const float4* __restrict__ in, const float4* __restrict__ planes, float4* __restrict__ out, long P4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P4) return;
    const long n = i / (P4 / 32), p = i % (P4 / 32);
    float4 a = in[i], s = make_float4(0, 0, 0, 0), q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = planes[(n * 8 + k) * (P4 / 32) + p];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s.x += q[k].x; s.y += q[k].y; s.z += q[k].z; s.w += q[k].w; }
    out[i] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
}
// the same bytes in the pair kernel's geometry: a 16 x 64 tile of quads per workgroup (+ XR rows of one pixel per thread), XCD-aware tile order
template <int XR>
__global__ __launch_bounds__(256) void copy_tile(const float* __restrict__ in, const float* __restrict__ planes, float* __restrict__ out, int H, int W) {
    constexpr int TRT = 16 + XR;
    const int tiles_x = W / 64, tiles_y = (H + TRT - 1) / TRT;
    const unsigned x = blockIdx.x % 8u, q = gridDim.x / 8u, rr = gridDim.x % 8u;
    int t = (int)(x * q + (x < rr ? x : rr) + blockIdx.x / 8u);
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const long n = t / tiles_y, P = (long)H * W;
    const int r = ty * TRT + threadIdx.x / 16, c = tx * 64 + (threadIdx.x % 16) * 4;
    const int r2 = ty * TRT + 16 + threadIdx.x / 64, c2 = tx * 64 + threadIdx.x % 64;
    float4 q4[8]; float q1[8];
    const bool live = r < H, live2 = XR > 0 && r2 < H;
    float4 a = make_float4(0, 0, 0, 0); float a1 = 0.f;
    if (live) { a = *reinterpret_cast<const float4*>(in + n * P + (long)r * W + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) q4[k] = *reinterpret_cast<const float4*>(planes + (n * 8 + k) * P + (long)r * W + c); }
    if (live2) { a1 = in[n * P + (long)r2 * W + c2];
#pragma unroll
        for (int k = 0; k < 8; ++k) q1[k] = planes[(n * 8 + k) * P + (long)r2 * W + c2]; }
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { a.x += q4[k].x; a.y += q4[k].y; a.z += q4[k].z; a.w += q4[k].w; }
        *reinterpret_cast<float4*>(out + n * P + (long)r * W + c) = a; }
    if (live2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a1 += q1[k];
        out[n * P + (long)r2 * W + c2] = a1; }
}
template <int TR, int TC, int SWZ>
__global__ __launch_bounds__(256) void copy_geo(const float* __restrict__ in, const float* __restrict__ planes, float* __restrict__ out, int H, int W) {
    const int tiles_x = W / TC, tiles_y = (H + TR - 1) / TR;
    int t = (int)blockIdx.x;
    if (SWZ) { const unsigned x = blockIdx.x % 8u, q = gridDim.x / 8u, rr = gridDim.x % 8u; t = (int)(x * q + (x < rr ? x : rr) + blockIdx.x / 8u); }
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const long n = t / tiles_y, P = (long)H * W;
    const int r = ty * TR + threadIdx.x / (TC / 4), c = tx * TC + (threadIdx.x % (TC / 4)) * 4;
    if (r >= H) return;
    float4 q4[8];
    float4 a = *reinterpret_cast<const float4*>(in + n * P + (long)r * W + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) q4[k] = *reinterpret_cast<const float4*>(planes + (n * 8 + k) * P + (long)r * W + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a.x += q4[k].x; a.y += q4[k].y; a.z += q4[k].z; a.w += q4[k].w; }
    *reinterpret_cast<float4*>(out + n * P + (long)r * W + c) = a;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = 32, H = 200, W = 256, D = 2, SETS = 8;
    const size_t P = (size_t)H * W, nl = N * P, ng = 8 * nl;
    std::vector<float> hx(nl), hg(ng);
    srand(1);
    for (auto& v : hx) v = 6.f * ((float)rand() / RAND_MAX - 0.5f);
    for (auto& v : hg) v = 2.f * ((float)rand() / RAND_MAX - 0.5f);
    std::vector<float*> x(SETS), g(SETS), o(SETS);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&x[s], nl * 4)); CK(hipMalloc(&g[s], ng * 4)); CK(hipMalloc(&o[s], nl * 4));
        CK(hipMemcpy(x[s], hx.data(), nl * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g[s], hg.data(), ng * 4, hipMemcpyHostToDevice));
    }
    float* o2; CK(hipMalloc(&o2, nl * 4));
    const int tiles20 = N * (H / 20) * (W / 64);
    auto wide = [&](int s, float* out) {
        const size_t ldw = 2 * sizeof(float) * (size_t)(20 + 2 * D) * bxi::PwGeom<2, 64>::PC;
        hipLaunchKernelGGL((bxi::pairwise3_bwd_wide_kernel<2, 16, 64, 4>), dim3(tiles20), dim3(256), ldw, 0, x[s], g[s], H, W, out, 1);
    };
    int swz = 3;
    auto pair = [&](int s, float* out) {
        hipLaunchKernelGGL((bxi::pairwise3_bwd_pair_kernel<2, 4>), dim3(tiles20), dim3(256), (bxi::PwPairGeom<2, 4>::lds_bytes), 0, x[s], g[s], H, W, out, swz);
    };
    const int tiles16 = N * ((H + 15) / 16) * (W / 64);
    auto pair16 = [&](int s, float* out) {
        hipLaunchKernelGGL((bxi::pairwise3_bwd_pair_kernel<2, 0>), dim3(tiles16), dim3(256), (bxi::PwPairGeom<2, 0>::lds_bytes), 0, x[s], g[s], H, W, out, 1);
    };
    // agreement
    CK(hipMemset(o[0], 0xff, nl * 4)); CK(hipMemset(o2, 0xff, nl * 4));
    wide(0, o[0]); pair(0, o2); CK(hipDeviceSynchronize());
    std::vector<float> a(nl), b(nl);
    CK(hipMemcpy(a.data(), o[0], nl * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, nl * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < nl; ++i) { if (!(std::fabs(a[i] - b[i]) <= 1e-4 * std::max(1.0, (double)std::fabs(a[i])))) ++bad; md = std::max(md, (double)std::fabs(a[i] - b[i])); mx = std::max(mx, (double)std::fabs(a[i])); }
    printf("pair vs wide: max |diff| %.3g (max |wide| %.3g), %zu of %zu beyond 1e-4\n", md, mx, bad, nl);
    CK(hipMemset(o2, 0xff, nl * 4)); pair16(0, o2); CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), o2, nl * 4, hipMemcpyDeviceToHost));
    md = 0; bad = 0;
    for (size_t i = 0; i < nl; ++i) { if (!(std::fabs(a[i] - b[i]) <= 1e-4 * std::max(1.0, (double)std::fabs(a[i])))) ++bad; md = std::max(md, (double)std::fabs(a[i] - b[i])); }
    printf("pair16 vs wide: max |diff| %.3g, %zu beyond 1e-4\n", md, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 20; ++i) launch(i % SETS, o[i % SETS]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 200;
        for (int i = 0; i < n; ++i) launch(i % SETS, o[i % SETS]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-18s %.2f us cold\n", name, ms / n * 1e3);
    };
    auto copy = [&](int s, float* out) {
        hipLaunchKernelGGL(copy59, dim3((unsigned)(nl / 4 / 256)), dim3(256), 0, 0, (const float4*)x[s], (const float4*)g[s], (float4*)out, (long)(nl / 4));
    };
    for (int rep = 0; rep < 2; ++rep) { run("wide", wide); run("pair", pair); run("pair16", pair16); run("copy", copy);
#define GEO(TR, TC, SWZ) run("geo " #TR "x" #TC " swz" #SWZ, [&](int s, float* out) { hipLaunchKernelGGL((copy_geo<TR, TC, SWZ>), dim3(N * ((H + TR - 1) / TR) * (W / TC)), dim3(256), 0, 0, x[s], g[s], out, H, W); })
        for (swz = 0; swz <= 5; ++swz) { char nm[32]; snprintf(nm, 32, "pair swz %d", swz); run(nm, pair); }
        swz = 3;
        if (rep == 0) { GEO(4, 256, 0); GEO(4, 256, 1); GEO(8, 128, 0); GEO(8, 128, 1); GEO(16, 64, 0); GEO(16, 64, 1); GEO(32, 32, 1); }
        run("copy_t16", [&](int s, float* out) { hipLaunchKernelGGL(copy_tile<0>, dim3(tiles16), dim3(256), 0, 0, x[s], g[s], out, H, W); });
        run("copy_t20", [&](int s, float* out) { hipLaunchKernelGGL(copy_tile<4>, dim3(tiles20), dim3(256), 0, 0, x[s], g[s], out, H, W); }); }
#ifdef BXI_PW_TRACE
    long long* tr; const size_t tn = (size_t)4 * 8192 * 8;
    CK(hipMalloc(&tr, tn * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_pw_trace), &tr, sizeof(tr)));
    std::vector<long long> ht(tn);
    for (int which = 0; which < 2; ++which) {
        CK(hipMemset(tr, 0, tn * 8));
        for (int i = 0; i < 8; ++i) { if (which) pair(i % SETS, o[i % SETS]); else wide(i % SETS, o[i % SETS]); }   // warm the code, cold data for the traced one
        CK(hipDeviceSynchronize()); CK(hipMemset(tr, 0, tn * 8)); CK(hipDeviceSynchronize());
        if (which) pair(5, o[5]); else wide(5, o[5]);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ht.data(), tr, tn * 8, hipMemcpyDeviceToHost));
        long long t0 = -1;
        for (int blk = 0; blk < tiles20; ++blk) { const long long v = ht[((size_t)which * 8192 + blk) * 8]; if (v > 0 && (t0 < 0 || v < t0)) t0 = v; }
        printf("%s: us from the first workgroup's start; quantiles [min 10%% 50%% 90%% max] over %d workgroups\n", which ? "pair" : "wide", tiles20);
        for (int ph = 0; ph < 8; ++ph) {
            std::vector<double> v;
            for (int blk = 0; blk < tiles20; ++blk) { const long long s = ht[((size_t)which * 8192 + blk) * 8 + ph]; if (s > 0) v.push_back((s - t0) * 0.01); }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            auto q = [&](double f) { return v[(size_t)(f * (v.size() - 1))]; };
            printf("  phase %d: n %4zu  %6.2f %6.2f %6.2f %6.2f %6.2f\n", ph, v.size(), q(0), q(.1), q(.5), q(.9), q(1));
        }
    }
#endif
    return 0;
}
