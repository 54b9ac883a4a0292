// pairwise_op.hip -- op-level drop-in for the reference's `pairwise_ext` (bind.cpp:15-36):
//   pairwise_nlog_forward  (pairwise.cu:68-104, launcher :154-175)
//   pairwise_nlog_backward (pairwise.cu:106-149, launcher :177-202)
// written for gfx950: wave64, x-fastest coalesced planes, no atomics (gather backward).
//
// Roofline: HBM.  Forward moves 4*(1+K) B per pixel (K = size^2-1 output planes, write-bound);
// backward 4*(1+K+1) B per pixel (g_pairwise read once through L2: every element is used by the
// pixel itself (channel k) and by one neighbour (channel K-1-k)).
#include "common.hpp"

namespace bxi {

// f(x,y) = -log(s(x)s(y) + s(-x)s(-y)) evaluated in log space exactly as pairwise.cu:38-50.
template <typename T>
__device__ __forceinline__ T pair_nlog(T ax, T bx, T ay, T by) {
    T e1 = ax + ay, e0 = bx + by;
    T mx = e1 > e0 ? e1 : e0;
    T df = e1 > e0 ? e1 - e0 : e0 - e1;
    return logsig(df) - mx;
}

template <typename T>
__global__ __launch_bounds__(256) void pairwise_fwd_kernel(const T* __restrict__ logits, int N, int H, int W,
                                                           int size, int dil, T* __restrict__ out) {
    const int K = size * size - 1;
    const int R = size / 2 * dil;
    const int64_t P = (int64_t)H * W;
    const int64_t total = (int64_t)N * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int64_t n = idx / P;
        const T* L = logits + n * P;
        const T here = L[(int64_t)y * W + x];
        const T ax = logsig(here), bx = logsig(-here);
        T* o = out + n * K * P + (int64_t)y * W + x;
        int k = 0;
        for (int dy = -R; dy <= R; dy += dil)
            for (int dx = -R; dx <= R; dx += dil) {
                if (dx == 0 && dy == 0) continue;
                const int x2 = x + dx, y2 = y + dy;
                T v = T(0);  // padded neighbour: log-probs 0 => pair == 0 (pairwise.cu:43-44)
                if (x2 >= 0 && x2 < W && y2 >= 0 && y2 < H) {
                    const T there = L[(int64_t)y2 * W + x2];
                    v = pair_nlog(ax, bx, logsig(there), logsig(-there));
                }
                o[(int64_t)k * P] = v;
                ++k;
            }
    }
}

// d f(a,b) / d a = -(s(b) - s(-b)) * exp(logs(a) + logs(-a) + f(a,b))      (pairwise.cu:56-58)
template <typename T>
__global__ __launch_bounds__(256) void pairwise_bwd_kernel(const T* __restrict__ logits,
                                                           const T* __restrict__ g_pair, int N, int H, int W,
                                                           int size, int dil, T* __restrict__ g_logits) {
    const int K = size * size - 1;
    const int R = size / 2 * dil;
    const int64_t P = (int64_t)H * W;
    const int64_t total = (int64_t)N * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int64_t n = idx / P;
        const T* L = logits + n * P;
        const T* GP = g_pair + n * K * P;
        const int64_t p = (int64_t)y * W + x;
        const T here = L[p];
        const T ax = logsig(here), bx = logsig(-here);
        T acc = T(0);
        int k = 0;
        for (int dy = -R; dy <= R; dy += dil)
            for (int dx = -R; dx <= R; dx += dil) {
                if (dx == 0 && dy == 0) continue;
                const int x2 = x + dx, y2 = y + dy;
                if (x2 >= 0 && x2 < W && y2 >= 0 && y2 < H) {
                    const int64_t q = (int64_t)y2 * W + x2;
                    const T there = L[q];
                    const T ay = logsig(there), by = logsig(-there);
                    const T pair = pair_nlog(ax, bx, ay, by);
                    // channel k at p is the pair (p,q); channel K-1-k at q is the pair (q,p)
                    const T g = GP[(int64_t)k * P + p] + GP[(int64_t)(K - 1 - k) * P + q];
                    acc += -(t_exp(ay) - t_exp(by)) * t_exp(ax + bx + pair) * g;
                }
                ++k;
            }
        g_logits[n * P + p] = acc;
    }
}

template <typename T>
static int launch_fwd(const T* logits, int N, int H, int W, int size, int dil, T* out, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0 || size == 1) return BXI_OK;
    if (!logits || !out) return BXI_ERR_NULL_POINTER;
    const int64_t total = (int64_t)N * H * W;
    if (!fits_i32(total * (size * size - 1))) return BXI_ERR_BAD_SHAPE;
    const int block = 256;
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 64) grid = 256 * 64;  // 256 CUs x 8 waves/SIMD; grid-stride the rest
    BXI_LAUNCH("pairwise_fwd", as_stream(stream), (pairwise_fwd_kernel<T>), dim3((unsigned)grid), dim3(block), 0, as_stream(stream), logits,
                       N, H, W, size, dil, out);
    return check_launch();
}

template <typename T>
static int launch_bwd(const T* logits, const T* g_pair, int N, int H, int W, int size, int dil, T* g_logits,
                      void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0) return BXI_OK;
    if (!logits || !g_logits || (size > 1 && !g_pair)) return BXI_ERR_NULL_POINTER;
    const int64_t total = (int64_t)N * H * W;
    if (!fits_i32(total * (size * size - 1 > 0 ? size * size - 1 : 1))) return BXI_ERR_BAD_SHAPE;
    const int block = 256;
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 64) grid = 256 * 64;
    BXI_LAUNCH("pairwise_bwd", as_stream(stream), (pairwise_bwd_kernel<T>), dim3((unsigned)grid), dim3(block), 0, as_stream(stream), logits,
                       g_pair, N, H, W, size, dil, g_logits);
    return check_launch();
}

}  // namespace bxi

extern "C" {

int bxi_pairwise_nlog_forward_f32(const float* logits, int N, int H, int W, int size, int dilation,
                                  float* pairwise, void* stream) {
    return bxi::launch_fwd<float>(logits, N, H, W, size, dilation, pairwise, stream);
}
int bxi_pairwise_nlog_forward_f64(const double* logits, int N, int H, int W, int size, int dilation,
                                  double* pairwise, void* stream) {
    return bxi::launch_fwd<double>(logits, N, H, W, size, dilation, pairwise, stream);
}
int bxi_pairwise_nlog_backward_f32(const float* logits, const float* pairwise, const float* g_pairwise, int N,
                                   int H, int W, int size, int dilation, float* g_logits, void* stream) {
    (void)pairwise;
    return bxi::launch_bwd<float>(logits, g_pairwise, N, H, W, size, dilation, g_logits, stream);
}
int bxi_pairwise_nlog_backward_f64(const double* logits, const double* pairwise, const double* g_pairwise, int N,
                                   int H, int W, int size, int dilation, double* g_logits, void* stream) {
    (void)pairwise;
    return bxi::launch_bwd<double>(logits, g_pairwise, N, H, W, size, dilation, g_logits, stream);
}

}  // extern "C"
