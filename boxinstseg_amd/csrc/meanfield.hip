// meanfield.hip -- SURVEY 8(f-3): the DiscoBox pseudo-label path on gfx950
// (mmdet/models/dense_heads/discobox_head.py: MeanField :585-655, dice_loss :542-550, mil_loss :552-562).
//
// MeanField is a 3x3 (ksize x ksize) Gaussian-bilateral stencil iterated `iters` times on a per-instance
// probability map that is re-thresholded to two values {base, 1-base} after every step (:653): the state of
// an instance is therefore ONE BIT per pixel, and outside the target (box) mask it is constantly "low"
// (f[:,1] *= targets, :649).  Here:
//   * the state is kept as 64-pixel words (row segment = wavefront): a wave evaluates 64 consecutive
//     pixels of a row, `__ballot` of the new decisions IS the new state word -- no atomics, no bit fiddling
//     per lane; neighbour bits come from nine scalar word loads and a shift;
//   * one launch per iteration over all (instance, row, segment) items (the reference: ~12 torch ops per
//     iteration, each materialising [n,2,9,HW] unfold tensors); items outside the target exit at once;
//   * the per-pixel arithmetic is the reference's fp32 op sequence, step by step (products and sums are not
//     contracted, 9 neighbours summed in unfold order), so that the thresholded decisions agree.
// The neighbourhood kernel K[k,p] (MeanField.__init__) is built once per image by mf_kernel_build.
//
// dice_loss / mil_loss: row-wise reductions + dense, atomic-free backward (same structure as the BoxInst
// projection term, but on probabilities, arbitrary 0/1 targets and the +0.001 form of the dice).
#include "common.hpp"

namespace bxi {

// ---------------------------------------------------------------------------------------------------
// MeanField.__init__ (:597-611)
template <int KS>
__global__ __launch_bounds__(256) void mf_kernel_build(const float* __restrict__ feat, int B, int C, int H, int W,
                                                       float alpha0, float d0, float d1, float* __restrict__ K) {
    constexpr int HALF = KS / 2;
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * HW) return;
    const int b = (int)(i / HW);
    const int p = (int)(i % HW), r = p / W, c = p % W;
    const float* F = feat + (int64_t)b * C * HW;
    float* out = K + (int64_t)b * KS * KS * HW + p;
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        const int dy = k / KS - HALF, dx = k % KS - HALF;
        const int r2 = r + dy, c2 = c + dx;
        const bool inb = r2 >= 0 && r2 < H && c2 >= 0 && c2 < W;
        float acc = 0.f;
        for (int ch = 0; ch < C; ++ch) {               // .sum(1): channel order
            const float fp = __fadd_rn(F[ch * HW + p], 10.f);
            const float fq = inb ? __fadd_rn(F[ch * HW + (int64_t)r2 * W + c2], 10.f) : 0.f;   // zero padding of (feat + 10)
            const float d = __fsub_rn(fq, fp);
            acc = __fadd_rn(acc, -__fmul_rn(d, d));
        }
        const float e = __fadd_rn(__fdiv_rn(acc, d0), -__fdiv_rn((float)(dy * dy + dx * dx), d1));
        out[(int64_t)k * HW] = __fmul_rn(alpha0, expf(e));
    }
}

// ---------------------------------------------------------------------------------------------------
// MeanField.forward (:617-638) + simple_forward (:640-655)
//
// State = one bit per pixel, stored as 64-pixel words [N][H][segs] in the workspace (target mask, state A, state B).
// One launch per mean-field iteration over ALL (instance, row, segment) items: a wave per item, every CU busy
// (a first version ran one workgroup per instance with the state in LDS and all iterations in one launch: the
// ~130 VALU instructions per pixel and iteration then sit on ONE CU per instance -- 390 us for 16 instances while
// 240 CUs idle; a launch boundary costs 2.5 us).  Items whose target word is empty exit at once: outside the target
// the state is constantly low (f[:, 1:] *= targets, :649).
typedef unsigned long long u64;

struct MfArgs {
    const float* K;           // [B,KS*KS,H,W]
    const float* x;           // [N,H,W]
    const void* t;            // [N,H,W] f32 or u8
    const int64_t* img;       // [N] or null
    const float* inter;       // [N,2,H,W] or null
    float* ret;               // [N,H,W]
    float* valid;             // [N]
    u64* tw;                  // [N,H,segs] target words
    u64* sa;                  // [N,H,segs] state, read
    u64* sb;                  // [N,H,segs] state, written
    int B, H, W, N, t_u8, segs;
    float nl_lo0, nl_lo1, nl_hi0, nl_hi1;   // -log(1-x), -log(x) for x = lo / hi
    float gamma, vlo, vhi;
};

constexpr int kMfWaves = 4;   // waves per workgroup of the per-item kernels
// IPW = consecutive items per wave: 4 when there are many items (4x fewer workgroups to dispatch), 1 when few
// (all the parallelism there is)

// first item of this wave
template <int IPW>
__device__ __forceinline__ int mf_first_item() {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return ((int)blockIdx.x * kMfWaves + wave) * IPW;
}

// initial state (:621-622) and the target mask
template <int IPW>
__global__ __launch_bounds__(256) void mf_init_kernel(MfArgs a) {
    constexpr int kMfIPW = IPW;
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int first = mf_first_item<IPW>(), total = a.H * a.segs;
    const int64_t HW = (int64_t)a.H * a.W;
    float tv[kMfIPW], xv[kMfIPW];
#pragma unroll
    for (int u = 0; u < kMfIPW; ++u) {                          // all loads first
        const int item = first + u, r = item / a.segs, c = (item % a.segs) * 64 + lane;
        tv[u] = 0.f; xv[u] = 0.f;
        if (item < total && c < a.W) {
            const int64_t o = (int64_t)n * HW + (int64_t)r * a.W + c;
            tv[u] = a.t_u8 ? (float)reinterpret_cast<const uint8_t*>(a.t)[o] : reinterpret_cast<const float*>(a.t)[o];
            xv[u] = a.x[o];
        }
    }
#pragma unroll
    for (int u = 0; u < kMfIPW; ++u) {
        const int item = first + u;
        if (item >= total) break;
        const u64 mt = __ballot(tv[u] != 0.f);
        const u64 ms = __ballot(__fmul_rn(xv[u], tv[u]) > 0.5f);
        if (lane == 0) {
            const int64_t w = (int64_t)n * total + item;
            a.tw[w] = mt; a.sa[w] = ms; a.sb[w] = 0ull;      // both buffers: items without target are never written again
        }
    }
}

// one simple_forward (:640-655) of every item
template <int KS>
__device__ __forceinline__ void mf_eval_item(const MfArgs& a, const u64* sa, u64* sb, int n, int r, int sg, u64 mt, int lane) {
    constexpr int HALF = KS / 2, KK = KS * KS;
    const int H = a.H, W = a.W, segs = a.segs, c = sg * 64 + lane;
    const int64_t wbase = (int64_t)n * H * segs;
    const int64_t HW = (int64_t)H * W;
    const int cc = c < W ? c : W - 1;                             // clamped: loads need no branch
    // img_inds is device data: clamp it into [0, B) so that a bad index reads a wrong kernel plane, never foreign memory
    const int64_t bi = a.img ? (a.img[n] < 0 ? 0 : (a.img[n] >= a.B ? a.B - 1 : a.img[n])) : 0;
    const float* Kp = a.K + bi * (int64_t)KK * HW + (int64_t)r * W + cc;
    float kv[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) kv[k] = Kp[(int64_t)k * HW];
    float i0 = 0.f, i1 = 0.f;
    if (a.inter) { const float* ip = a.inter + (int64_t)n * 2 * HW + (int64_t)r * W + cc; i0 = ip[0]; i1 = ip[HW]; }
    // the state words of the neighbourhood rows (uniform addresses); outside the map = 0 and never used
    u64 wl[KS], wm[KS], wr[KS];
#pragma unroll
    for (int dy = -HALF; dy <= HALF; ++dy) {
        const int r2 = r + dy;
        const bool rin = r2 >= 0 && r2 < H;
        const u64* row = sa + wbase + (int64_t)(rin ? r2 : r) * segs;
        wm[dy + HALF] = rin ? row[sg] : 0ull;
        wl[dy + HALF] = rin && sg > 0 ? row[sg - 1] : 0ull;
        wr[dy + HALF] = rin && sg + 1 < segs ? row[sg + 1] : 0ull;
    }
    const bool act = (mt >> lane) & 1ull;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int dy = -HALF; dy <= HALF; ++dy) {
        const int r2 = r + dy;
        const bool rin = r2 >= 0 && r2 < H;
#pragma unroll
        for (int dx = -HALF; dx <= HALF; ++dx) {
            const int k = (dy + HALF) * KS + (dx + HALF);
            const u64 m = wm[dy + HALF];
            const u64 sh = dx < 0 ? (m << (-dx)) | (wl[dy + HALF] >> (64 + dx)) : dx > 0 ? (m >> dx) | (wr[dy + HALF] << (64 - dx)) : m;
            const bool bit = (sh >> lane) & 1ull;
            const int c2 = c + dx;
            if (rin && c2 >= 0 && c2 < W) {                   // outside the map nn.Unfold pads -log(U) with 0: adds nothing
                acc0 = __fadd_rn(acc0, __fmul_rn(bit ? a.nl_hi0 : a.nl_lo0, kv[k]));
                acc1 = __fadd_rn(acc1, __fmul_rn(bit ? a.nl_hi1 : a.nl_lo1, kv[k]));
            }
        }
    }
    float f0 = expf(-acc0), f1 = expf(-acc1);
    if (a.inter) { f0 = __fadd_rn(f0, __fmul_rn(i0, a.gamma)); f1 = __fadd_rn(f1, __fmul_rn(i1, a.gamma)); }
    f1 = act ? f1 : 0.f;                                      // f[:, 1:] *= targets
    f0 = __fadd_rn(f0, 1e-6f); f1 = __fadd_rn(f1, 1e-6f);
    const float r1v = __fdiv_rn(f1, __fadd_rn(f0, f1));
    const u64 nw = __ballot(act && c < W && r1v > 0.5f);
    if (lane == 0) sb[wbase + (int64_t)r * segs + sg] = nw;
}

// One launch per iteration.  (Tried: all iterations in one launch of a persistent grid meeting at an agent-scope
// counter between iterations -- correct, but a meeting of 1024 workgroups across the 8 XCDs costs ~135 us on this
// part, 1.4 ms in total; a kernel boundary costs 2.5 us.)
template <int KS, int IPW>
__global__ __launch_bounds__(256) void mf_step_kernel(MfArgs a) {
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int first = mf_first_item<IPW>(), total = a.H * a.segs;
    if (first >= total) return;
    const u64* twp = a.tw + (int64_t)n * total + first;
    u64 mt[IPW];
#pragma unroll
    for (int u = 0; u < IPW; ++u) mt[u] = first + u < total ? twp[u] : 0ull;
#pragma unroll
    for (int u = 0; u < IPW; ++u)
        if (mt[u]) mf_eval_item<KS>(a, a.sa, a.sb, n, (first + u) / a.segs, (first + u) % a.segs, mt[u], lane);   // wave-uniform
}

// ret = (state > 0.5) (:631-632): a wave per IPW items
template <int IPW>
__global__ __launch_bounds__(256) void mf_ret_kernel(MfArgs a) {
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int first = mf_first_item<IPW>(), total = a.H * a.segs;
    if (first >= total) return;
    const u64* st = a.sa + (int64_t)n * total + first;
    float* out = a.ret + (int64_t)n * a.H * a.W;
    u64 w[IPW];
#pragma unroll
    for (int u = 0; u < IPW; ++u) w[u] = first + u < total ? st[u] : 0ull;
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
        const int item = first + u;
        if (item >= total) break;
        const int r = item / a.segs, c = (item % a.segs) * 64 + lane;
        if (c < a.W) out[(int64_t)r * a.W + c] = ((w[u] >> lane) & 1ull) ? 1.f : 0.f;
    }
}

// valid (:633-636): a wave per instance counts the foreground bits of the final state (no atomics)
__global__ __launch_bounds__(64) void mf_valid_kernel(MfArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x, total = a.H * a.segs;
    const u64* st = a.sa + (int64_t)n * total;
    int cnt = 0;
    for (int i = lane; i < total; i += 64) cnt += __popcll(st[i]);
    cnt = wave_sum_i32(cnt);
    if (lane == 0) {
        const float count = (float)cnt;
        a.valid[n] = (count >= a.vlo && count <= a.vhi) ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------
// dice_loss (:542-550)
__device__ __forceinline__ float load_t(const void* t, int u8, int64_t o) {
    return u8 ? (float)reinterpret_cast<const uint8_t*>(t)[o] : reinterpret_cast<const float*>(t)[o];
}

__device__ __forceinline__ double block_sum_f64(double v, double* red /*[16]*/) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];   // fixed order
    return s;
}

__global__ __launch_bounds__(1024) void dice_fwd_kernel(const float* __restrict__ in, const void* __restrict__ tg, int t_u8,
                                                        int64_t L, float* __restrict__ loss, float* __restrict__ sums) {
    __shared__ double red[16];
    const int n = blockIdx.x;
    double a = 0.0, bc = 0.0;
    for (int64_t j = threadIdx.x; j < L; j += 1024) {
        const float i = in[(int64_t)n * L + j], t = load_t(tg, t_u8, (int64_t)n * L + j);
        a += (double)i * t; bc += (double)i * i + (double)t * t;
    }
    a = block_sum_f64(a, red);
    bc = block_sum_f64(bc, red) + 0.002;
    if (threadIdx.x == 0) {
        loss[n] = (float)(1.0 - 2.0 * a / bc);
        sums[2 * n] = (float)a; sums[2 * n + 1] = (float)bc;
    }
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ in, const void* __restrict__ tg, int t_u8,
                                                       int N, int64_t L, const float* __restrict__ sums,
                                                       const float* __restrict__ g_loss, float* __restrict__ g_in) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * L) return;
    const int n = (int)(i / L);
    const float a = sums[2 * n], bc = sums[2 * n + 1];
    const float x = in[i], t = load_t(tg, t_u8, i);
    g_in[i] = g_loss[n] * (-2.f * t / bc + 4.f * a * x / (bc * bc));
}

// ---------------------------------------------------------------------------------------------------
// mil_loss (:552-562): state = [N][ argc i32[W] | argr i32[H] | gcol f32[W] | grow f32[H] ] followed by scratch of the
// forward.  Two launches so that the whole GPU reads the maps (one workgroup per instance kept 16 CUs busy: 28 us):
//   mil_band_kernel   grid (bands of 16 rows, N): a wave takes 4 rows, lanes over the columns (all 4 x kCT x 2 loads of a step
//                     are independent); row maxima (complete) by wave reductions of packed (value, first column) keys,
//                     column maxima of the band by 64-bit LDS atomic max of packed (value, first row) keys -- max is
//                     order-independent, so the result is deterministic -- written as per-band partials;
//   mil_finish_kernel one workgroup per instance: combines the column partials, the four dice sums in fp64 (fixed order),
//                     loss, unit gradients, arg-max positions.
// eps = what the two dice denominators carry in total (mil_loss/dice_loss: 0.001 + 0.001; BoxProjectionLoss: 1e-5),
// weight = loss_weight (folded into the loss and the unit gradients)
constexpr int kMilBandRows = 16;       // 4 waves x 4 rows

struct MilScratch { u64* colkey; uint32_t* coltk; float* rowv; float* rowt; };   // [N][nb][W] x 2, [N][H] x 2
__host__ __device__ inline size_t mil_state_words(int H, int W) { return 2 * (size_t)(H + W); }
__host__ __device__ inline size_t mil_carve(void* state, int N, int H, int W, MilScratch* ms) {
    const size_t N1 = N > 0 ? N : 1, nb = (size_t)(H + kMilBandRows - 1) / kMilBandRows;
    size_t off = (N1 * mil_state_words(H, W) * 4 + 15) & ~(size_t)15;
    char* p = static_cast<char*>(state);
    MilScratch t;
    t.colkey = reinterpret_cast<u64*>(p + off); off += 8 * N1 * nb * W;
    t.coltk = reinterpret_cast<uint32_t*>(p + off); off += 4 * N1 * nb * W;
    t.rowv = reinterpret_cast<float*>(p + off); off += 4 * N1 * H;
    t.rowt = reinterpret_cast<float*>(p + off); off += 4 * N1 * H;
    if (ms) *ms = t;
    return off;
}

__global__ __launch_bounds__(256) void mil_band_kernel(const float* __restrict__ in, const void* __restrict__ tg, int t_u8, int H, int W,
                                                       int* __restrict__ state, MilScratch ms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mil_raw[];
    u64* colkey = reinterpret_cast<u64*>(mil_raw);                   // [W]
    uint32_t* coltk = reinterpret_cast<uint32_t*>(colkey + W);       // [W] target maxima as ordered keys
    const int band = blockIdx.x, n = blockIdx.y, nb = gridDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t HW = (int64_t)H * W;
    const float* x = in + (int64_t)n * HW;
    int* argr = state + (int64_t)n * mil_state_words(H, W) + W;
    for (int c = tid; c < W; c += 256) { colkey[c] = 0ull; coltk[c] = 0u; }
    __syncthreads();
    constexpr int R = 4;
    const int rb = band * kMilBandRows + wave * R;
    if (rb < H) {
        u64 rkey[R]; float rt[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { rkey[i] = 0ull; rt[i] = -INFINITY; }
        // kCT column trips (4 rows x 2 arrays each) are loaded before any of them is consumed.  Columns / rows past the end
        // are clamped to the last one: a duplicate changes no maximum.
        constexpr int kCT = 5;
        for (int c0 = 0; c0 < W; c0 += 64 * kCT) {
            float v[kCT][R], t[kCT][R];
#pragma unroll
            for (int q = 0; q < kCT; ++q) {
                const int c = min(c0 + q * 64 + lane, W - 1);
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int r = min(rb + i, H - 1);
                    v[q][i] = x[(int64_t)r * W + c];
                    t[q][i] = load_t(tg, t_u8, (int64_t)n * HW + (int64_t)r * W + c);
                }
            }
#pragma unroll
            for (int q = 0; q < kCT; ++q) {
                const int c = min(c0 + q * 64 + lane, W - 1);
                u64 ck = 0ull; float ct = -INFINITY;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int r = min(rb + i, H - 1);
                    const u64 kc_ = pack_max(v[q][i], (uint32_t)r); ck = kc_ > ck ? kc_ : ck;      // first row wins ties
                    ct = fmaxf(ct, t[q][i]);
                    const u64 kr = pack_max(v[q][i], (uint32_t)c); rkey[i] = kr > rkey[i] ? kr : rkey[i];
                    rt[i] = fmaxf(rt[i], t[q][i]);
                }
                atomicMax(&colkey[c], ck);
                atomicMax(&coltk[c], float_key(ct));
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u64 k = wave_max_u64(rkey[i]);
            const float tm = wave_max_f32(rt[i]);
            if (lane == 0 && rb + i < H) {
                ms.rowv[(int64_t)n * H + rb + i] = unpack_val(k); ms.rowt[(int64_t)n * H + rb + i] = tm; argr[rb + i] = (int)unpack_idx(k);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < W; c += 256) {
        ms.colkey[((int64_t)n * nb + band) * W + c] = colkey[c];
        ms.coltk[((int64_t)n * nb + band) * W + c] = coltk[c];
    }
}

__global__ __launch_bounds__(256) void mil_finish_kernel(int H, int W, int nb, double eps, double weight, float* __restrict__ loss,
                                                         int* __restrict__ state, MilScratch ms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mil_raw[];
    u64* colkey = reinterpret_cast<u64*>(mil_raw);                   // [W]
    uint32_t* coltk = reinterpret_cast<uint32_t*>(colkey + W);       // [W]
    __shared__ double red[16];
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int* argc = state + (int64_t)n * mil_state_words(H, W);
    int* argr = argc + W;
    float* gcol = reinterpret_cast<float*>(argr + H);
    float* grow = gcol + W;
    const float* rowv = ms.rowv + (int64_t)n * H;
    const float* rowt = ms.rowt + (int64_t)n * H;
    double ac = 0.0, bcc = 0.0, ar = 0.0, bcr = 0.0;
    for (int c = tid; c < W; c += 256) {
        u64 k = 0ull; uint32_t tk = 0u;
        for (int b0 = 0; b0 < nb; b0 += 8) {    // 16 loads in flight; a larger key wins: larger value, then smaller row
            u64 o[8]; uint32_t ot[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = min(b0 + i, nb - 1);                       // clamped: a duplicate changes no maximum
                o[i] = ms.colkey[((int64_t)n * nb + b) * W + c];
                ot[i] = ms.coltk[((int64_t)n * nb + b) * W + c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { k = o[i] > k ? o[i] : k; tk = ot[i] > tk ? ot[i] : tk; }
        }
        colkey[c] = k; coltk[c] = tk;
        const double v = unpack_val(k), t = key_float(tk);
        ac += v * t; bcc += v * v + t * t;
    }
    for (int r = tid; r < H; r += 256) { ar += (double)rowv[r] * rowt[r]; bcr += (double)rowv[r] * rowv[r] + (double)rowt[r] * rowt[r]; }
    // the four sums together (one after another each costs six dependent cross-lane steps and two barriers)
    ac = wave_sum_f64(ac); bcc = wave_sum_f64(bcc); ar = wave_sum_f64(ar); bcr = wave_sum_f64(bcr);
    if (lane == 0) { red[wave * 4 + 0] = ac; red[wave * 4 + 1] = bcc; red[wave * 4 + 2] = ar; red[wave * 4 + 3] = bcr; }
    __syncthreads();
    ac = bcc = ar = bcr = 0.0;
    for (int wv = 0; wv < 4; ++wv) { ac += red[wv * 4 + 0]; bcc += red[wv * 4 + 1]; ar += red[wv * 4 + 2]; bcr += red[wv * 4 + 3]; }   // fixed order
    bcc += eps; bcr += eps;
    if (tid == 0) loss[n] = (float)(weight * ((1.0 - 2.0 * ar / bcr) + (1.0 - 2.0 * ac / bcc)));   // loss_func(column..) + loss_func(row..)
    for (int c = tid; c < W; c += 256) {        // each thread reads back what it wrote itself
        const double v = unpack_val(colkey[c]), t = key_float(coltk[c]);
        gcol[c] = (float)(weight * (-2.0 * t / bcc + 4.0 * ac * v / (bcc * bcc)));
        argc[c] = (int)unpack_idx(colkey[c]);
    }
    for (int r = tid; r < H; r += 256) grow[r] = (float)(weight * (-2.0 * rowt[r] / bcr + 4.0 * ar * rowv[r] / (bcr * bcr)));
}

__global__ __launch_bounds__(256) void mil_bwd_kernel(int N, int H, int W, const int* __restrict__ state,
                                                      const float* __restrict__ g_loss, float* __restrict__ g_in) {
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * HW) return;
    const int n = (int)(i / HW), p = (int)(i % HW), r = p / W, c = p % W;
    const int* argc = state + (int64_t)n * 2 * (H + W);
    const int* argr = argc + W;
    const float* gcol = reinterpret_cast<const float*>(argr + H);
    const float* grow = gcol + W;
    float g = 0.f;
    if (argc[c] == r) g += gcol[c];
    if (argr[r] == c) g += grow[r];
    g_in[i] = g_loss[n] * g;
}

}  // namespace bxi

extern "C" {

int bxi_meanfield_kernel_f32(const float* feat, int B, int C, int H, int W, int ksize, float alpha0, float theta0,
                             float theta1, float* kernel, void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (ksize != 3 && ksize != 5) return BXI_ERR_UNSUPPORTED;
    if (!(theta0 > 0.f) || !(theta1 > 0.f)) return BXI_ERR_BAD_ARGUMENT;
    if (!feat || !kernel) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)B * H * W * ksize * ksize)) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const float d0 = (float)(2.0 * (double)theta0 * (double)theta0), d1 = (float)(2.0 * (double)theta1 * (double)theta1);
    const unsigned grid = (unsigned)(((int64_t)B * H * W + 255) / 256);
    if (ksize == 3) BXI_LAUNCH("mf_kernel_build", s, bxi::mf_kernel_build<3>, dim3(grid), dim3(256), 0, s, feat, B, C, H, W, alpha0, d0, d1, kernel);
    else BXI_LAUNCH("mf_kernel_build", s, bxi::mf_kernel_build<5>, dim3(grid), dim3(256), 0, s, feat, B, C, H, W, alpha0, d0, d1, kernel);
    return bxi::check_launch();
}

size_t bxi_meanfield_workspace_bytes(int N, int H, int W) {
    if (N < 0 || H <= 0 || W <= 0) return 0;
    return 3 * sizeof(unsigned long long) * (size_t)(N > 0 ? N : 1) * H * ((W + 63) / 64);
}

int bxi_meanfield_forward_f32(const float* kernel, int B, int H, int W, int ksize, const float* x, const void* targets,
                              int targets_u8, const int64_t* img_inds, int N, int iters, float base,
                              const float* inter_img_mask, float gamma, float* ret, float* valid, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || N < 0 || iters < 0) return BXI_ERR_BAD_SHAPE;
    if (ksize != 3 && ksize != 5) return BXI_ERR_UNSUPPORTED;
    if (!(base > 0.f) || !(base < 0.5f)) return BXI_ERR_BAD_ARGUMENT;
    if (N == 0) return BXI_OK;
    if (N > 65535) return BXI_ERR_UNSUPPORTED;
    if (!kernel || !x || !targets || !ret || !valid) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * H * W) || !bxi::fits_i32((int64_t)B * H * W * ksize * ksize)) return BXI_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < bxi_meanfield_workspace_bytes(N, H, W) || (reinterpret_cast<uintptr_t>(workspace) & 7))
        return BXI_ERR_WORKSPACE;
    bxi::MfArgs a;
    a.K = kernel; a.x = x; a.t = targets; a.img = img_inds; a.inter = inter_img_mask; a.ret = ret; a.valid = valid;
    a.B = B; a.H = H; a.W = W; a.N = N; a.t_u8 = targets_u8 ? 1 : 0; a.segs = (W + 63) / 64;
    const size_t plane = (size_t)N * H * a.segs;
    a.tw = reinterpret_cast<bxi::u64*>(workspace); a.sa = a.tw + plane; a.sb = a.sa + plane;
    // the two values a thresholded probability takes, (f > 0.5).float() * (1 - base*2) + base (:622, :653), in fp32
    const float span = (float)(1.0 - (double)base * 2.0);
    const float lo = 0.f * span + base, hi = 1.f * span + base;
    a.nl_lo0 = -logf(1.f - lo); a.nl_lo1 = -logf(lo);
    a.nl_hi0 = -logf(1.f - hi); a.nl_hi1 = -logf(hi);
    a.gamma = gamma;
    a.vlo = (float)((double)(H * W) * 0.05); a.vhi = (float)((double)(H * W) * 0.95);
    hipStream_t s = bxi::as_stream(stream);
    const bool many = (int64_t)N * H * a.segs > 32768;
    const int ipw = many ? 4 : 1, per_wg = bxi::kMfWaves * ipw;
    const dim3 grid((unsigned)((H * a.segs + per_wg - 1) / per_wg), (unsigned)N), block(64 * bxi::kMfWaves);
    if (many) BXI_LAUNCH("mf_init", s, bxi::mf_init_kernel<4>, grid, block, 0, s, a);
    else BXI_LAUNCH("mf_init", s, bxi::mf_init_kernel<1>, grid, block, 0, s, a);
    int rc = bxi::check_launch();
    for (int it = 0; it < iters && rc == BXI_OK; ++it) {
        if (ksize == 3) {
            if (many) BXI_LAUNCH("mf_step", s, (bxi::mf_step_kernel<3, 4>), grid, block, 0, s, a);
            else BXI_LAUNCH("mf_step", s, (bxi::mf_step_kernel<3, 1>), grid, block, 0, s, a);
        } else {
            if (many) BXI_LAUNCH("mf_step", s, (bxi::mf_step_kernel<5, 4>), grid, block, 0, s, a);
            else BXI_LAUNCH("mf_step", s, (bxi::mf_step_kernel<5, 1>), grid, block, 0, s, a);
        }
        rc = bxi::check_launch();
        bxi::u64* tmp = a.sa; a.sa = a.sb; a.sb = tmp;
    }
    if (rc != BXI_OK) return rc;
    if (many) BXI_LAUNCH("mf_ret", s, bxi::mf_ret_kernel<4>, grid, block, 0, s, a);
    else BXI_LAUNCH("mf_ret", s, bxi::mf_ret_kernel<1>, grid, block, 0, s, a);
    rc = bxi::check_launch();
    if (rc != BXI_OK) return rc;
    BXI_LAUNCH("mf_valid", s, bxi::mf_valid_kernel, dim3((unsigned)N), dim3(64), 0, s, a);
    return bxi::check_launch();
}

int bxi_dice_loss_forward_f32(const float* input, const void* target, int target_u8, int N, int64_t L, float* loss,
                              float* sums, void* stream) {
    if (N < 0 || L <= 0) return BXI_ERR_BAD_SHAPE;
    if (N == 0) return BXI_OK;
    if (!input || !target || !loss || !sums) return BXI_ERR_NULL_POINTER;
    hipStream_t s = bxi::as_stream(stream);
    BXI_LAUNCH("dice_fwd", s, bxi::dice_fwd_kernel, dim3(N), dim3(1024), 0, s, input, target, target_u8 ? 1 : 0, L, loss, sums);
    return bxi::check_launch();
}

int bxi_dice_loss_backward_f32(const float* input, const void* target, int target_u8, int N, int64_t L,
                               const float* sums, const float* g_loss, float* g_input, void* stream) {
    if (N < 0 || L <= 0) return BXI_ERR_BAD_SHAPE;
    if (N == 0) return BXI_OK;
    if (!input || !target || !sums || !g_loss || !g_input) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32(((int64_t)N * L + 255) / 256)) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)N * L + 255) / 256);
    BXI_LAUNCH("dice_bwd", s, bxi::dice_bwd_kernel, dim3(grid), dim3(256), 0, s, input, target, target_u8 ? 1 : 0, N, L, sums,
               g_loss, g_input);
    return bxi::check_launch();
}

size_t bxi_mil_loss_state_bytes(int N, int H, int W) {
    if (N < 0 || H <= 0 || W <= 0) return 0;
    return bxi::mil_carve(nullptr, N, H, W, nullptr);        // what the backward reads + the forward's per-band scratch
}

static int launch_mil_fwd(const float* input, const void* target, int target_u8, int N, int H, int W, double eps, double weight,
                          float* loss, void* state, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (N == 0) return BXI_OK;
    if (!input || !target || !loss || !state) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * H * W)) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 15) return BXI_ERR_WORKSPACE;
    const size_t lds = (size_t)W * 12;
    if (lds > 64 * 1024 || N > 65535) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = bxi::as_stream(stream);
    bxi::MilScratch ms;
    bxi::mil_carve(state, N, H, W, &ms);
    const int nb = (H + bxi::kMilBandRows - 1) / bxi::kMilBandRows;
    BXI_LAUNCH("mil_band", s, bxi::mil_band_kernel, dim3(nb, N), dim3(256), lds, s, input, target, target_u8 ? 1 : 0, H, W,
               reinterpret_cast<int*>(state), ms);
    int rc = bxi::check_launch();
    if (rc != BXI_OK) return rc;
    BXI_LAUNCH("mil_finish", s, bxi::mil_finish_kernel, dim3(N), dim3(256), lds, s, H, W, nb, eps, weight, loss,
               reinterpret_cast<int*>(state), ms);
    return bxi::check_launch();
}

int bxi_mil_loss_forward_f32(const float* input, const void* target, int target_u8, int N, int H, int W, float* loss,
                             void* state, void* stream) {
    return launch_mil_fwd(input, target, target_u8, N, H, W, 0.002, 1.0, loss, state, stream);
}

int bxi_projection_loss_forward_f32(const float* mask_scores, const float* box_bitmask, int N, int H, int W, float loss_weight,
                                    float* loss, void* state, void* stream) {
    return launch_mil_fwd(mask_scores, box_bitmask, 0, N, H, W, 1e-5, (double)loss_weight, loss, state, stream);
}

int bxi_mil_loss_backward_f32(int N, int H, int W, const void* state, const float* g_loss, float* g_input, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return BXI_ERR_BAD_SHAPE;
    if (N == 0) return BXI_OK;
    if (!state || !g_loss || !g_input) return BXI_ERR_NULL_POINTER;
    if (!bxi::fits_i32((int64_t)N * H * W)) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = bxi::as_stream(stream);
    const unsigned grid = (unsigned)(((int64_t)N * H * W + 255) / 256);
    BXI_LAUNCH("mil_bwd", s, bxi::mil_bwd_kernel, dim3(grid), dim3(256), 0, s, N, H, W, reinterpret_cast<const int*>(state),
               g_loss, g_input);
    return bxi::check_launch();
}

}  // extern "C"
