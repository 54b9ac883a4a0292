// The reference's own pairwise.cu KERNELS (mmdet/ops/pairwise/csrc/pairwise/pairwise.cu, namespace pairwise_kernel:
// the device maths :17-66, pairwise_nlog_forward_kernel :68-104, pairwise_nlog_backward_kernel :106-147), run on the CPU.
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile (target `ref`, build container only) extracts that namespace into a temporary
// file outside the repository (REF_PAIRWISE_KERNELS_INC), compiles it in here through cuda_on_cpu.h and deletes it; only the
// .so lands in oracle/_ref/.  The launchers (:151-202: ATen, AT_DISPATCH, <<< >>>) are not compiled; what they do around the
// kernels is restated below: the grid (:159-162), `empty` pairwise [B, size^2-1, H, W] (:155-157), `zeros_like` g_logits
// (:186).  The kernels have no barrier, so the CUDA threads run one after another: one fixed order of the backward's
// atomicAdds (on a GPU their order varies from run to run).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "cuda_on_cpu.h"

// the one torch type the kernels touch: a 4-d accessor with .size(i) and [b][c][y][x]
namespace torch {
template <class T, int N>
struct PackedTensorAccessor32 {
    T* data; int sizes[4]; int strides[4];
    int size(int i) const { return sizes[i]; }
    struct Sub3 { T* d; const int* st; struct Sub2 { T* d; const int* st; struct Sub1 { T* d; const int* st; T& operator[](int i) const { return d[i * st[0]]; } };
                                                    Sub1 operator[](int i) const { return Sub1{d + i * st[0], st + 1}; } };
                  Sub2 operator[](int i) const { return Sub2{d + i * st[0], st + 1}; } };
    Sub3 operator[](int i) const { return Sub3{data + i * strides[0], strides + 1}; }
};
}  // namespace torch

using std::exp; using std::log;
#include REF_PAIRWISE_KERNELS_INC

template <class T>
static torch::PackedTensorAccessor32<T, 4> acc(T* p, int B, int C, int H, int W) {
    torch::PackedTensorAccessor32<T, 4> a;
    a.data = p; a.sizes[0] = B; a.sizes[1] = C; a.sizes[2] = H; a.sizes[3] = W;
    a.strides[0] = C * H * W; a.strides[1] = H * W; a.strides[2] = W; a.strides[3] = 1;
    return a;
}

static dim3 grid_of(long n) {           // :159-162  min(ceil_div(numel / size(1), threadsPerBlock), maxGridDim)
    const long blocks = (n + pairwise_kernel::threadsPerBlock - 1) / pairwise_kernel::threadsPerBlock;
    return dim3((unsigned)std::min(blocks, pairwise_kernel::maxGridDim));
}

template <class T>
static int forward(int size, int dil, const T* logits, int B, int H, int W, T* pairwise) {
    auto lg = acc(const_cast<T*>(logits), B, 1, H, W);
    auto pw = acc(pairwise, B, size * size - 1, H, W);
    cpu_cuda::launch_serial(grid_of((long)B * H * W), dim3((unsigned)pairwise_kernel::threadsPerBlock),
                            [&] { pairwise_kernel::pairwise_nlog_forward_kernel<T>(size, dil, lg, pw); });
    return 0;
}

template <class T>
static int backward(int size, int dil, const T* logits, const T* pairwise, const T* g_pairwise, int B, int H, int W, T* g_logits) {
    std::memset(g_logits, 0, sizeof(T) * (size_t)B * H * W);
    auto lg = acc(const_cast<T*>(logits), B, 1, H, W);
    auto pw = acc(const_cast<T*>(pairwise), B, size * size - 1, H, W);
    auto gp = acc(const_cast<T*>(g_pairwise), B, size * size - 1, H, W);
    auto gl = acc(g_logits, B, 1, H, W);
    cpu_cuda::launch_serial(grid_of((long)B * H * W), dim3((unsigned)pairwise_kernel::threadsPerBlock),
                            [&] { pairwise_kernel::pairwise_nlog_backward_kernel<T>(size, dil, lg, pw, gl, gp); });
    return 0;
}

extern "C" {
int ref_pairwise_nlog_forward_f32(int size, int dil, const float* logits, int B, int H, int W, float* pairwise) { return forward(size, dil, logits, B, H, W, pairwise); }
int ref_pairwise_nlog_forward_f64(int size, int dil, const double* logits, int B, int H, int W, double* pairwise) { return forward(size, dil, logits, B, H, W, pairwise); }
int ref_pairwise_nlog_backward_f32(int size, int dil, const float* logits, const float* pairwise, const float* g_pairwise, int B, int H, int W,
                                   float* g_logits) { return backward(size, dil, logits, pairwise, g_pairwise, B, H, W, g_logits); }
int ref_pairwise_nlog_backward_f64(int size, int dil, const double* logits, const double* pairwise, const double* g_pairwise, int B, int H, int W,
                                   double* g_logits) { return backward(size, dil, logits, pairwise, g_pairwise, B, H, W, g_logits); }
}
