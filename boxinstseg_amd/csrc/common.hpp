// common.hpp -- shared host/device helpers for libboxinst_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/boxinst_hip.h"

namespace bxi {

constexpr int kWave = 64;  // CDNA wavefront

// ---- host side ------------------------------------------------------------------------------
void set_last_hip_error(int e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error((int)e);
        return BXI_ERR_LAUNCH;
    }
    return BXI_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// measurement hook (bxi_set_launch_hook): brackets one kernel launch
typedef void (*bxi_launch_hook)(const char* kernel_name, int phase, void* stream, void* user);      // include/boxinst_hip_dev.h
extern std::atomic<bxi_launch_hook> g_hook;
extern std::atomic<void*> g_hook_user;
struct LaunchScope {
    const char* name; hipStream_t s; bxi_launch_hook hook; void* user;
    LaunchScope(const char* n, hipStream_t st) : name(n), s(st), hook(g_hook.load(std::memory_order_acquire)), user(nullptr) {
        if (hook) { user = g_hook_user.load(std::memory_order_acquire); hook(name, 0, (void*)s, user); }
    }
    ~LaunchScope() { if (hook) hook(name, 1, (void*)s, user); }
};

#define BXI_LAUNCH(label, stream, ...)                 \
    do {                                               \
        ::bxi::LaunchScope _scope(label, stream);      \
        hipLaunchKernelGGL(__VA_ARGS__);               \
    } while (0)

inline bool fits_i32(int64_t v) { return v >= 0 && v <= 0x7fffffffLL; }

// compute units of the current device (256 on an MI355X in SPX mode, 32 per partition in CPX): grids are sized so that a launch is
// resident in one round.  Cached per device ordinal; a wrong value costs time, never correctness.
inline int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// ---- developer tracing (-DBXI_TRACE builds only; never in the shipped library) ---------------
// BXI_T(kernel_id, block, phase) stores the 100 MHz wall clock of lane 0 into a global buffer.
#ifdef BXI_TRACE
constexpr int kTraceBlocks = 8192, kTracePhases = 8;
static __device__ long long* g_trace = nullptr;   // per translation unit
#define BXI_T(kid, blk, ph)                                                                            \
    do {                                                                                               \
        if (threadIdx.x == 0 && g_trace && (blk) < ::bxi::kTraceBlocks)                                \
            g_trace[((size_t)(kid) * ::bxi::kTraceBlocks + (blk)) * ::bxi::kTracePhases + (ph)] = wall_clock64(); \
    } while (0)
#else
#define BXI_T(kid, blk, ph) do {} while (0)
#endif

// ---- device side ----------------------------------------------------------------------------
// log sigmoid, the reference's formula (pairwise.cu:27-37) in the overflow-free arrangement
// min(x,0) - log(1 + exp(-|x|)).
__device__ __forceinline__ float logsig(float x) { return fminf(x, 0.f) - logf(1.f + expf(-fabsf(x))); }
__device__ __forceinline__ double logsig(double x) { return fmin(x, 0.0) - log(1.0 + exp(-fabs(x))); }

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }

// order-preserving map float -> uint32 (larger float <=> larger key); -inf maps above 0.
__device__ __forceinline__ uint32_t float_key(float x) {
    uint32_t u = __float_as_uint(x);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu);
    return __uint_as_float(u);
}
// (value, index) -> one 64-bit key whose max is "largest value, then smallest index".
__device__ __forceinline__ unsigned long long pack_max(float x, uint32_t idx) {
    return ((unsigned long long)float_key(x) << 32) | (unsigned long long)(0xffffffffu - idx);
}
__device__ __forceinline__ float unpack_val(unsigned long long k) { return key_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t unpack_idx(unsigned long long k) { return 0xffffffffu - (uint32_t)k; }

// Wave totals on the VALU's data-parallel path (DPP) instead of six dependent trips through the LDS crossbar
// (__shfl_xor = ds_bpermute, ~130 clk each): quad swaps, the two mirrors within a row of 16 lanes, then the four rows by
// v_readlane.  Every lane gets the total; fixed order, so run-to-run identical.
template <typename F>
__device__ __forceinline__ void wave_total_steps(F&& step) {
    step(0xB1);    // quad_perm:[1,0,3,2]
    step(0x4E);    // quad_perm:[2,3,0,1]
    step(0x141);   // row_half_mirror
    step(0x140);   // row_mirror
}
__device__ __forceinline__ int dpp_i32(int v, int ctrl) {
    switch (ctrl) {
        case 0xB1: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
        case 0x4E: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
        case 0x141: return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    }
}
__device__ __forceinline__ float wave_total_f32(float v) {
    wave_total_steps([&](int c) { v += __int_as_float(dpp_i32(__float_as_int(v), c)); });
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ int wave_total_i32(int v) {
    wave_total_steps([&](int c) { v += dpp_i32(v, c); });
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ double wave_total_f64(double v) {
    wave_total_steps([&](int c) {
        const long long b = __double_as_longlong(v);
        const int lo = dpp_i32((int)b, c), hi = dpp_i32((int)(b >> 32), c);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    });
    auto row = [&](int l) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    return (row(0) + row(16)) + (row(32) + row(48));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    wave_total_steps([&](int c) {
        const unsigned long long o = ((unsigned long long)(unsigned int)dpp_i32((int)(v >> 32), c) << 32) | (unsigned int)dpp_i32((int)v, c);
        v = o > v ? o : v;
    });
    auto row = [&](int l) {
        return ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)v, l);
    };
    const unsigned long long a = row(0), b = row(16), c2 = row(32), d = row(48);
    const unsigned long long ab = a > b ? a : b, cd = c2 > d ? c2 : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    wave_total_steps([&](int c) { v = fmaxf(v, __int_as_float(dpp_i32(__float_as_int(v), c))); });
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
// (the historical names: every kernel's wave reduction goes through the DPP forms above)
__device__ __forceinline__ float wave_sum_f32(float v) { return wave_total_f32(v); }
__device__ __forceinline__ double wave_sum_f64(double v) { return wave_total_f64(v); }
__device__ __forceinline__ int wave_sum_i32(int v) { return wave_total_i32(v); }

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for the
// wave's outstanding global stores (vmcnt), so zero-fill / output stores stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- stores that are written through to memory as they are issued (sc1 = agent scope) --------------------------------
// Used (a) for bulk output that should drain while the launch is still reading instead of in a burst at the kernel
// boundary, and (b) for data another workgroup of the SAME launch reads after an arrival counter (MI355X_MICROARCH.md,
// "inter-workgroup visibility", form R1: sc1 payload -> s_waitcnt vmcnt(0) -> agent-scope atomic).  16-byte forms: a narrower
// sc1 store is one fabric write per lane.  The asm forms are invisible to the compiler's vmcnt bookkeeping: callers drain
// with drain_vmem() before they publish.
__device__ __forceinline__ void store4_through(float* p, float x, float y, float z, float w) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_u64x2_through(unsigned long long* p, unsigned long long a, unsigned long long b) {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v v = {(unsigned int)a, (unsigned int)(a >> 32), (unsigned int)b, (unsigned int)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// python slice [a:b] on an axis of length n -> [lo,hi)   (condinst_head.py:1429-1430)
__device__ __forceinline__ void py_slice(int a, int b, int n, int& lo, int& hi) {
    if (a < 0) a += n;
    a = a < 0 ? 0 : (a > n ? n : a);
    if (b < 0) b += n;
    b = b < 0 ? 0 : (b > n ? n : b);
    lo = a;
    hi = b > a ? b : a;
}

// Box (x1,y1,x2,y2 in canvas pixels) -> half-open rectangle [r0,r1) x [c0,c1) on the sampled
// grid y = start + r*stride: the cells whose sample lies inside the python slice.
struct Rect { int r0, r1, c0, c1; };
__device__ __forceinline__ void sampled_range(int lo, int hi, int start, int stride, int n, int& a, int& b) {
    // smallest r with start + r*stride >= lo ; smallest r with start + r*stride >= hi
    int ra = lo - start <= 0 ? 0 : (lo - start + stride - 1) / stride;
    int rb = hi - start <= 0 ? 0 : (hi - start + stride - 1) / stride;
    a = ra > n ? n : ra;
    b = rb > n ? n : rb;
    if (b < a) b = a;
}
__device__ __forceinline__ Rect box_rect(const float* box, int Hc, int Wc, int stride, int start, int h, int w) {
    int y0, y1, x0, x1;
    py_slice((int)box[1], (int)box[3] + 1, Hc, y0, y1);
    py_slice((int)box[0], (int)box[2] + 1, Wc, x0, x1);
    Rect rc;
    sampled_range(y0, y1, start, stride, h, rc.r0, rc.r1);
    sampled_range(x0, x1, start, stride, w, rc.c0, rc.c1);
    if (rc.r1 <= rc.r0 || rc.c1 <= rc.c0) { rc.r0 = rc.r1 = rc.c0 = rc.c1 = 0; }
    return rc;
}

// per-image metadata carried in kernel arguments (no device copy, no sync)
struct GtTable {
    const float* boxes[BXI_MAX_IMAGES];  // device pointers, [G_i,4]
    int first[BXI_MAX_IMAGES + 1];       // first[b] = sum_{i<b} G_i
    int B;
};
__device__ __forceinline__ const float* gt_box(const GtTable& t, int g, int& img) {
    int b = 0;
    while (b + 1 < t.B && g >= t.first[b + 1]) ++b;
    img = b;
    return t.boxes[b] + 4 * (g - t.first[b]);
}

}  // namespace bxi
