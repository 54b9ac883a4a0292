// sol_eval.hip -- a measuring stick, not a kernel of the product: ONE launch with eval1_kernel's grid (the same numbers of 256-thread
// workgroups in the same order, four per CU) that moves the evaluation's bytes -- the same loads and stores by the same roles -- and
// does nothing else: no arithmetic beyond keeping the loads alive, and NO workgroup waits for another.  bench.py times it on the
// same cold input sets next to the evaluation (`roofline.sol_us`): what is left between the two is what the evaluation's in-grid
// dependencies (Lab -> predicates -> sum W -> gradient -> losses; condinst_head.py:1297-1337 as one launch) cost, not bytes.
// Declared in include/boxinst_hip_dev.h; nothing in boxinstseg_amd/ calls it.
#include "common.hpp"
#include "../../include/boxinst_hip_dev.h"

namespace bxi {

// the product's cross-workgroup read: 16-byte records past the vector L1 (sc1), four in flight, one wait (as pred_item does)
__device__ __forceinline__ void sol_load16_past_x4(const void* p0, const void* p1, const void* p2, const void* p3, float4& a, float4& b, float4& c, float4& d) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v va, vb, vc, vd;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(va), "=&v"(vb), "=&v"(vc), "=&v"(vd) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w);
    c = make_float4(vc.x, vc.y, vc.z, vc.w); d = make_float4(vd.x, vd.y, vd.z, vd.w);
}

struct SolArgs {
    const float* imgs; int B, Hc, Wc;
    const float* logits; int N, h, w;
    float* g_logits;
    float4* lab4; unsigned int* pred; unsigned int* sink;
    int n_stream, n_pool, n_items, n_pb, n_lead, n_tb, tiles;
};

__global__ __launch_bounds__(256, 4) void sol_eval_kernel(SolArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int idx = (int)blockIdx.x;
    float acc = 0.f;
    const int64_t P = (int64_t)a.h * a.w;
    if (idx < a.n_stream) {                                   // stream role: 4 waves x 8 rows of one instance map: read + zero-fill
        const int Sn = (a.h + 31) / 32, n = idx / Sn, s = idx % Sn;
        const int r0 = s * 32 + wv * 8;
        for (int cb = 0; cb < a.w; cb += 256) {
            const int c = cb + lane * 4;
            if (c < a.w)
                for (int i = 0; i < 8; ++i)
                    if (r0 + i < a.h) {
                        store4_through(a.g_logits + n * P + (int64_t)(r0 + i) * a.w + c, 0.f, 0.f, 0.f, 0.f);
                        const float4 v = *reinterpret_cast<const float4*>(a.logits + n * P + (int64_t)(r0 + i) * a.w + c);
                        acc += v.x + v.y + v.z + v.w;
                    }
        }
    } else if ((idx -= a.n_stream) < a.n_pool) {              // pool role: the 4 input rows of 64 pooled pixels per item, 16 B out per pooled pixel
        const int segs = (a.w + 63) >> 6;
        const int64_t plane = (int64_t)a.Hc * a.Wc;
        for (int item = idx; item < a.n_items; item += a.n_pool) {
            const int seg = item % segs, r = (item / segs) % a.h, b = item / (segs * a.h);
            const int c = seg * 64 + lane;
            if (c < a.w) {
                const float* base = a.imgs + (int64_t)b * 3 * plane + (int64_t)(4 * r + wv) * a.Wc + 4 * c;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int ch = 0; ch < 3; ++ch) {
                    const float4 v = *reinterpret_cast<const float4*>(base + ch * plane);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                if (wv == 3) store4_through(reinterpret_cast<float*>(a.lab4 + ((int64_t)b * a.h + r) * a.w + c), s.x, s.y, s.z, s.w);
                else acc += s.x + s.y + s.z + s.w;
            }
        }
    } else if ((idx -= a.n_pool) < a.n_pb) {                  // predicate role: 4 x 16 B in, 4 B out per pooled pixel
        const int segs = (a.w + 63) >> 6;
        for (int item = idx * 4 + wv; item < a.n_items; item += a.n_pb * 4) {
            const int seg = item % segs, r = (item / segs) % a.h, b = item / (segs * a.h);
            const int c = min(seg * 64 + lane, a.w - 1), rD = min(r + 2, a.h - 1), cx = min(c + 2, a.w - 1);
            const float4* L4 = a.lab4 + (int64_t)b * P;
            float4 o0, oD, x0, xD;
            sol_load16_past_x4(L4 + (int64_t)r * a.w + c, L4 + (int64_t)rD * a.w + c, L4 + (int64_t)r * a.w + cx, L4 + (int64_t)rD * a.w + cx, o0, oD, x0, xD);
            __hip_atomic_store(a.pred + (int64_t)b * P + (int64_t)r * a.w + c, __float_as_uint(o0.x + oD.y + x0.z + xD.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if ((idx -= a.n_pb) < 1 + a.n_lead) {              // reducer + leaders: nothing to move that matters (KB)
    } else if ((idx -= 1 + a.n_lead) < a.n_tb) {              // tile role: a wave per tile: 8 logit rows + 6 predicate words in, 4 rows of gradient added
        const int t = idx * 4 + wv;
        if (t < a.tiles) {
            const int n = t % a.N, k = t / a.N;
            const int r0 = (k * 12) % max(a.h - 8, 1), c0 = ((k * 60) % max(a.w - 64, 1)) & ~3;
            const float* Lg = a.logits + n * P;
            for (int j = 0; j < 8; ++j) acc += Lg[(int64_t)(r0 + j) * a.w + c0 + lane];
            for (int j = 0; j < 6; ++j) acc += __uint_as_float(__hip_atomic_load(a.pred + (int64_t)(r0 + j) * a.w + c0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * 1e-30f;
            for (int j = 0; j < 4; ++j)
                (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(a.g_logits + n * P + (int64_t)(r0 + 2 + j) * a.w + c0 + lane), acc * 0.f);
        }
    }
    if (acc == 1.2345e-33f) a.sink[0] = 1u;                   // keeps every load alive
    (void)smem;
}

// ---- the op-level pairwise_nlog kernels' bytes (pairwise.cu:68-202 at size 3: 8 planes), nothing else ------------------------------------
// mode 0: the forward's traffic -- logits [N,1,H,W] read, planes [N,8,H,W] written; mode 1: the backward's -- logits and planes read,
// out [N,1,H,W] written.  16-byte accesses, a wave's accesses contiguous, one pass per thread: what a copy can do on this part.
__global__ __launch_bounds__(256) void sol_pairwise_kernel(const float4* __restrict__ logits, float4* __restrict__ planes, float4* __restrict__ out,
                                                           long n4, int mode) {
    const long T = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += T) {
        const float4 a = logits[i];
        if (mode == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) planes[p * n4 + i] = make_float4(a.x + p, a.y, a.z, a.w);
        } else {
            float4 q[8], s = a;
#pragma unroll
            for (int p = 0; p < 8; ++p) q[p] = planes[p * n4 + i];
#pragma unroll
            for (int p = 0; p < 8; ++p) { s.x += q[p].x; s.y += q[p].y; s.z += q[p].z; s.w += q[p].w; }
            out[i] = s;
        }
    }
}

}  // namespace bxi

extern "C" int bxi_dev_sol_pairwise_f32(const float* logits, float* planes, float* out, int N, int H, int W, int mode, void* stream) {
    if (!logits || !planes || (mode == 1 && !out)) return BXI_ERR_NULL_POINTER;
    const long n = (long)N * H * W;
    if (N <= 0 || H <= 0 || W <= 0 || (n & 3) || mode < 0 || mode > 1) return BXI_ERR_BAD_SHAPE;
    const long n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 1664 ? 1664 : (n4 + 255) / 256);
    hipStream_t s = bxi::as_stream(stream);
    BXI_LAUNCH("sol_pairwise", s, bxi::sol_pairwise_kernel, dim3(grid), dim3(256), 0, s, (const float4*)logits, (float4*)planes, (float4*)out, n4, mode);
    return bxi::check_launch();
}

extern "C" int bxi_dev_sol_eval_f32(const float* imgs, int B, int Hc, int Wc, const float* logits, int N, int h, int w, float* g_logits, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!logits || !g_logits || !workspace) return BXI_ERR_NULL_POINTER;          // imgs == NULL: the loss-given-targets bytes (no image roles)
    if (B <= 0 || N <= 0 || h < 12 || w < 68 || Hc != 4 * h || Wc != 4 * w || (w & 3)) return BXI_ERR_BAD_SHAPE;   // (the tile role's stand-in rows / columns)
    const size_t Pp = (size_t)B * h * w;
    if (workspace_bytes < 16 * Pp + 4 * Pp + 256) return BXI_ERR_WORKSPACE;
    bxi::SolArgs a;
    a.imgs = imgs; a.B = B; a.Hc = Hc; a.Wc = Wc; a.logits = logits; a.N = N; a.h = h; a.w = w; a.g_logits = g_logits;
    a.lab4 = (float4*)workspace; a.pred = (unsigned int*)((char*)workspace + 16 * Pp); a.sink = (unsigned int*)((char*)workspace + 20 * Pp);
    // the grid of eval1_kernel (fused_eval.hip, launch_fused_eval) at this shape: stream blocks, pool blocks filling the rest of 4 slots per
    // CU with several items each, one predicate block per four row segments (at most half the slots), the reducer, N leaders, a tile
    // wave per ~3% of an instance map (the headline batch's tile count: 1153 at 32 instances), the finisher
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int slots = 4 * cus;
    a.n_stream = N * ((h + 31) / 32);
    a.n_items = B * h * ((w + 63) / 64);
    const int front = slots - a.n_stream;
    const int room = front > slots / 4 ? front : slots / 4;
    const int per = (a.n_items + room - 1) / room;
    a.n_pool = (a.n_items + per - 1) / per;
    a.n_pb = (a.n_items + 3) / 4 > slots / 2 ? slots / 2 : (a.n_items + 3) / 4;
    if (!imgs) a.n_pool = a.n_pb = 0;
    a.n_lead = N;
    a.tiles = (int)((int64_t)N * h * w * 36 / 51200 / 32);     // 1152 at 32 x 200 x 256 (the headline batch has 1153)
    a.n_tb = (a.tiles + 3) / 4;
    const unsigned grid = (unsigned)(a.n_stream + a.n_pool + a.n_pb + 1 + a.n_lead + a.n_tb + 1);
    hipStream_t s = bxi::as_stream(stream);
    BXI_LAUNCH("sol_eval", s, bxi::sol_eval_kernel, dim3(grid), dim3(256), 8192, s, a);
    return bxi::check_launch();
}
