// Minimal CUDA-kernel-on-CPU shim.  TEST INFRASTRUCTURE ONLY.
// Lets g++ compile the `__global__` kernels of the reference's tree_filter sources unchanged and run them with the CUDA
// execution model they rely on: one OS thread per CUDA thread of a block, blocks one after another, `__syncthreads()` =
// a barrier over the threads of the block that have not returned yet (the kernels call it inside data-dependent loops and
// leave at different times), `__shared__` = one object per block (blocks run sequentially, so a function-local static),
// `atomicAdd` = a sequentially consistent fetch-add.  Nothing here comes from the reference.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 { unsigned x = 1, y = 1, z = 1; dim3() {} dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static thread_local dim3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

namespace cpu_cuda {
// The threads of a block are OS threads, but only ONE of them runs kernel code at any time: a thread holds the block's
// mutex from the moment it starts until it returns, and gives it up only while it waits in __syncthreads().  Every
// barrier phase of a thread is therefore atomic with respect to the others -- one of the interleavings the CUDA model
// allows, and free of data races by construction.  (The kernels do contain races that are benign under warp execution,
// e.g. bfs.cu publishes sorted_index[pos] before parent_index[pos] and a reader tests only the former; with freely running
// OS threads that window is wide enough to corrupt the traversal.)
class BlockBarrier {
  public:
    explicit BlockBarrier(int n) : alive_(n) {}
    void enter() { lock_.lock(); }                  // start of a thread's kernel body
    void wait() {                                   // __syncthreads(); the caller holds the mutex
        const unsigned g = gen_;
        if (++waiting_ == alive_) { waiting_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lock_, [&] { return gen_ != g; });
    }
    void leave() {                                  // the calling thread has returned from the kernel
        --alive_;
        if (alive_ > 0 && waiting_ == alive_) { waiting_ = 0; ++gen_; cv_.notify_all(); }
        lock_.unlock();
    }
  private:
    std::mutex lock_; std::condition_variable_any cv_; int alive_, waiting_ = 0; unsigned gen_ = 0;
};
static BlockBarrier* g_barrier = nullptr;

template <class F>
void run_block(dim3 block, unsigned bx, unsigned by, F&& body) {
    BlockBarrier bar((int)block.x);
    g_barrier = &bar;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < block.x; ++t)
        th.emplace_back([&, t] {
            bar.enter();
            threadIdx = dim3(t); blockIdx = dim3(bx, by);
            body();
            bar.leave();
        });
    for (auto& x : th) x.join();
    g_barrier = nullptr;
}

// kernel<<<grid, block>>>(args...).  `before_each_block` (optional) is run as a block of its own first: a kernel that
// initialises a __shared__ array and reads other threads' entries without a barrier in between relies on the threads of a
// block starting together (true of warps, not of OS threads); running the same kernel on an empty problem leaves the
// array in that initial state, so a late starter's entry already holds what it is about to write.
template <class F, class G>
void launch(dim3 grid, dim3 block, F&& body, G&& before_each_block) {
    gridDim = grid; blockDim = block;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            run_block(block, bx, by, before_each_block);
            run_block(block, bx, by, body);
        }
}
template <class F>
void launch(dim3 grid, dim3 block, F&& body) { launch(grid, block, body, [] {}); }

// For kernels without __syncthreads(): the CUDA threads run one after another on the calling thread (one particular, fixed
// interleaving of their atomics).
template <class F>
void launch_serial(dim3 grid, dim3 block, F&& body) {
    gridDim = grid; blockDim = block;
    for (unsigned bx = 0; bx < grid.x; ++bx)
        for (unsigned t = 0; t < block.x; ++t) {
            threadIdx = dim3(t); blockIdx = dim3(bx);
            body();
        }
}
}  // namespace cpu_cuda

#define __global__ static
#define __device__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static volatile
static inline void __syncthreads() { cpu_cuda::g_barrier->wait(); }
static inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(volatile int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T>
static inline T atomic_add_fp(T* p, T v) {
    std::atomic_ref<T> a(*p);
    T old = a.load();
    while (!a.compare_exchange_weak(old, old + v)) {}
    return old;
}
static inline float atomicAdd(float* p, float v) { return atomic_add_fp(p, v); }
static inline double atomicAdd(double* p, double v) { return atomic_add_fp(p, v); }
