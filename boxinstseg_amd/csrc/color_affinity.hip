// color_affinity.hip -- target side of the BoxInst loss on gfx950, fully device-resident.
//
// Replaces (reference, LiWentomng/BoxInstSeg):
//   get_original_image            condinst_head.py:170-186  (mmcv tensor2imgs: GPU->CPU->GPU per image)
//   CondInstMaskHead.get_targets  condinst_head.py:1345-1393
//   get_bitmasks_from_boxes       condinst_head.py:1395-1448 (skimage rgb2lab on CPU, per-box loops)
//   get_image_color_similarity    condinst_head.py:220-246  (two F.unfold materialisations)
//
// Kernel A  pool_rgb      : imgs [B,3,Hc,Wc] f32 -> rgb_small [B,3,h,w] u8
//     de-normalise (double mul, round to f32, double add, round to f32 = OpenCV's arithmetic on
//     an f32 image with f64 scalars), truncate to u8, sum the stride x stride window, >> log2.
//     Pure stream: 12 B read per input pixel (3 x f32), 3/stride^2 B written.  HBM roofline.
//     One lane = one output pixel = `stride` rows x 16 B per channel, so a wave reads 1 KiB
//     contiguous per load instruction; all 3*stride loads are issued before the first use.
// Kernel B  affinity      : rgb_small -> Lab (fp64, LUT companding) in an LDS tile with halo ->
//     8 (K) neighbour distances -> sim [B,K,h,w] f32 and/or K-bit threshold mask per pixel.
//     Reads 3 B / pixel, writes 4K B / pixel (sim) + 1 B (mask): write-bound, tiny.
#include "common.hpp"
#include "srgb_lut.h"

namespace bxi {

struct ImageMeta {
    int img_h[BXI_MAX_IMAGES];
    int img_w[BXI_MAX_IMAGES];
    int first_removed[BXI_MAX_IMAGES];  // rows >= this are zeroed in the validity mask
};

struct Denorm {
    double mean[3], stdv[3];
    int src_ch[3];  // output channel c (RGB) reads tensor channel src_ch[c]
};

__device__ __forceinline__ int denorm_u8(float x, double s, double m) {
    // cv2.multiply(img_f32, std_f64) -> f32 ; cv2.add(img_f32, mean_f64) -> f32 ; astype(uint8)
    float t = (float)((double)x * s);
    float v = (float)((double)t + m);
    return (int)v & 0xff;
}

// ---- Kernel A, stride 4, vector path ---------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pool_rgb_s4_kernel(const float* __restrict__ imgs, int B, int Hc, int Wc,
                                                            ImageMeta meta, Denorm dn,
                                                            uint8_t* __restrict__ out) {
    const int h = Hc >> 2, w = Wc >> 2;
    const int64_t total = (int64_t)B * h * w;
    const int64_t o = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= total) return;
    const int c = (int)(o % w);
    const int r = (int)((o / w) % h);
    const int b = (int)(o / ((int64_t)w * h));
    const int ih = meta.img_h[b], iw = meta.img_w[b];
    const int64_t plane = (int64_t)Hc * Wc;
    const float* base = imgs + (int64_t)b * 3 * plane + (int64_t)(4 * r) * Wc + 4 * c;

    float4 v[3][4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[ch][i] = *reinterpret_cast<const float4*>(base + dn.src_ch[ch] * plane + (int64_t)i * Wc);

    const int x0 = 4 * c, y0 = 4 * r;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double s = dn.stdv[dn.src_ch[ch]], m = dn.mean[dn.src_ch[ch]];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool yin = (y0 + i) < ih;
            sum += (yin && x0 + 0 < iw) ? denorm_u8(v[ch][i].x, s, m) : 0;
            sum += (yin && x0 + 1 < iw) ? denorm_u8(v[ch][i].y, s, m) : 0;
            sum += (yin && x0 + 2 < iw) ? denorm_u8(v[ch][i].z, s, m) : 0;
            sum += (yin && x0 + 3 < iw) ? denorm_u8(v[ch][i].w, s, m) : 0;
        }
        out[((int64_t)(b * 3 + ch) * h + r) * w + c] = (uint8_t)(sum >> 4);
    }
}

// ---- Kernel A, any stride, scalar path (unaligned canvases, stride 1/2/8) ---------------------
__global__ __launch_bounds__(256) void pool_rgb_generic_kernel(const float* __restrict__ imgs, int B, int Hc, int Wc,
                                                               int stride, ImageMeta meta, Denorm dn,
                                                               uint8_t* __restrict__ out) {
    const int h = Hc / stride, w = Wc / stride;
    const int64_t total = (int64_t)B * h * w;
    const int64_t plane = (int64_t)Hc * Wc;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % w);
        const int r = (int)((o / w) % h);
        const int b = (int)(o / ((int64_t)w * h));
        const int ih = meta.img_h[b], iw = meta.img_w[b];
        for (int ch = 0; ch < 3; ++ch) {
            const int sc = dn.src_ch[ch];
            const double s = dn.stdv[sc], m = dn.mean[sc];
            const float* p = imgs + ((int64_t)b * 3 + sc) * plane;
            float sum = 0.f;  // F.avg_pool2d accumulates in f32; exact for u8-valued inputs
            for (int i = 0; i < stride; ++i)
                for (int j = 0; j < stride; ++j) {
                    const int y = r * stride + i, x = c * stride + j;
                    if (y < ih && x < iw) sum += (float)denorm_u8(p[(int64_t)y * Wc + x], s, m);
                }
            const float avg = sum / (float)(stride * stride);
            out[((int64_t)(b * 3 + ch) * h + r) * w + c] = (uint8_t)(int)avg;
        }
    }
}

// ---- Kernel B ---------------------------------------------------------------------------------
__constant__ double kSrgbLut[256] = BXI_SRGB_LUT_INIT;

// skimage.color.rgb2lab (rgb2xyz + xyz2lab, D65 / 2 deg) in fp64, no contraction, result to f32.
__device__ __forceinline__ void rgb2lab_f32(int r8, int g8, int b8, float& L, float& A, float& Bv) {
    const double r = kSrgbLut[r8], g = kSrgbLut[g8], b = kSrgbLut[b8];
    double xyz[3];
    const double M[3][3] = {{0.412453, 0.357580, 0.180423},
                            {0.212671, 0.715160, 0.072169},
                            {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double acc = __dadd_rn(__dadd_rn(__dmul_rn(M[i][0], r), __dmul_rn(M[i][1], g)), __dmul_rn(M[i][2], b));
        double v = acc / white[i];
        xyz[i] = v > 0.008856 ? cbrt(v) : __dadd_rn(__dmul_rn(7.787, v), 16.0 / 116.0);
    }
    L = (float)__dadd_rn(__dmul_rn(116.0, xyz[1]), -16.0);
    A = (float)__dmul_rn(500.0, __dadd_rn(xyz[0], -xyz[1]));
    Bv = (float)__dmul_rn(200.0, __dadd_rn(xyz[1], -xyz[2]));
}

constexpr int kAffRows = 4, kAffCols = 64;  // output tile of the affinity kernel (256 threads)

template <typename BitsT>
__global__ __launch_bounds__(256) void affinity_kernel(const uint8_t* __restrict__ rgb_small, int B, int h, int w,
                                                       int stride, int size, int dil, float thresh, ImageMeta meta,
                                                       const float* __restrict__ image_masks, int Hc, int Wc,
                                                       float* __restrict__ sim, BitsT* __restrict__ bits) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int R = size / 2 * dil;
    const int TW = kAffCols + 2 * R, TH = kAffRows + 2 * R;
    float* labL = lds;
    float* labA = lds + TW * TH;
    float* labB = lds + 2 * TW * TH;
    float* msk = lds + 3 * TW * TH;

    const int tiles_x = (w + kAffCols - 1) / kAffCols;
    const int tiles_y = (h + kAffRows - 1) / kAffRows;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int r0 = ty * kAffRows, c0 = tx * kAffCols;
    const int64_t P = (int64_t)h * w;
    const uint8_t* src = rgb_small + (int64_t)b * 3 * P;
    const int start = stride / 2;
    const int ih = meta.img_h[b], iw = meta.img_w[b], fr = meta.first_removed[b];

    for (int i = threadIdx.x; i < TW * TH; i += blockDim.x) {
        const int rr = r0 - R + i / TW, cc = c0 - R + i % TW;
        float L = 0.f, A = 0.f, Bv = 0.f, m = 0.f;  // zero padding of F.unfold (condinst_head.py:203-207)
        if (rr >= 0 && rr < h && cc >= 0 && cc < w) {
            const int64_t p = (int64_t)rr * w + cc;
            rgb2lab_f32(src[p], src[P + p], src[2 * P + p], L, A, Bv);
            const int y = rr * stride + start, x = cc * stride + start;
            if (image_masks)
                m = image_masks[((int64_t)b * Hc + y) * Wc + x];
            else
                m = (y < ih && x < iw && y < fr) ? 1.f : 0.f;  // condinst_head.py:1354-1369,1405
        }
        labL[i] = L; labA[i] = A; labB[i] = Bv; msk[i] = m;
    }
    __syncthreads();

    const int lr = threadIdx.x / kAffCols, lc = threadIdx.x % kAffCols;
    const int r = r0 + lr, c = c0 + lc;
    if (r >= h || c >= w) return;
    const int ci = (lr + R) * TW + (lc + R);
    const float L0 = labL[ci], A0 = labA[ci], B0 = labB[ci];
    const int K = size * size - 1;
    uint32_t word = 0;
    int k = 0;
    for (int dy = -R; dy <= R; dy += dil)
        for (int dx = -R; dx <= R; dx += dil) {
            if (dx == 0 && dy == 0) continue;
            const int qi = ci + dy * TW + dx;
            const float dL = L0 - labL[qi], dA = A0 - labA[qi], dB = B0 - labB[qi];
            const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(dL, dL), __fmul_rn(dA, dA)), __fmul_rn(dB, dB));
            const float s = __fmul_rn(expf(__fmul_rn(-__fsqrt_rn(n2), 0.5f)), msk[qi]);  // :237,:246
            if (sim) sim[((int64_t)b * K + k) * P + (int64_t)r * w + c] = s;
            word |= (s >= thresh ? 1u : 0u) << k;                                            // :1324
            ++k;
        }
    if (bits) bits[(int64_t)b * P + (int64_t)r * w + c] = (BitsT)word;
}

// ---- per-box bitmasks (condinst_head.py:1426-1432) ---------------------------------------------
__global__ __launch_bounds__(256) void box_bitmask_kernel(GtTable gt, int G, int Hc, int Wc, int stride, int start,
                                                          int h, int w, float* __restrict__ out) {
    const int g = blockIdx.y;
    int img;
    const float* box = gt_box(gt, g, img);
    const Rect rc = box_rect(box, Hc, Wc, stride, start, h, w);
    const int64_t P = (int64_t)h * w;
    float* o = out + (int64_t)g * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / w), c = (int)(i % w);
        o[i] = (r >= rc.r0 && r < rc.r1 && c >= rc.c0 && c < rc.c1) ? 1.f : 0.f;
    }
}

// ---- host side ---------------------------------------------------------------------------------
int fill_image_meta(const bxi_image_batch* bt, ImageMeta& meta, Denorm& dn) {
    if (!bt) return BXI_ERR_NULL_POINTER;
    if (bt->B < 0 || bt->B > BXI_MAX_IMAGES || bt->Hc <= 0 || bt->Wc <= 0) return BXI_ERR_BAD_SHAPE;
    if (bt->B > 0 && (!bt->img_h_host || !bt->img_w_host || !bt->rows_removed_host)) return BXI_ERR_NULL_POINTER;
    for (int b = 0; b < bt->B; ++b) {
        const int ih = bt->img_h_host[b], iw = bt->img_w_host[b], pr = bt->rows_removed_host[b];
        if (ih < 0 || iw < 0 || ih > bt->Hc || iw > bt->Wc) return BXI_ERR_BAD_SHAPE;
        meta.img_h[b] = ih;
        meta.img_w[b] = iw;
        int fr = pr > 0 ? ih - pr : ih;  // original_image_masks[-pr:, :] = 0  (python slice clamps at 0)
        meta.first_removed[b] = fr < 0 ? 0 : fr;
    }
    for (int c = 0; c < 3; ++c) {
        dn.mean[c] = bt->mean[c];
        dn.stdv[c] = bt->std[c];
        dn.src_ch[c] = bt->to_rgb ? c : 2 - c;  // tensor2imgs flips iff to_rgb, the caller flips back (:180-183)
    }
    return BXI_OK;
}

int launch_color_affinity(const bxi_image_batch* bt, int stride, int size, int dil, float thresh, uint8_t* rgb_small,
                          float* sim, void* affinity, void* stream) {
    ImageMeta meta;
    Denorm dn;
    int st = fill_image_meta(bt, meta, dn);
    if (st != BXI_OK) return st;
    if (stride < 1 || size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (bt->Hc % stride || bt->Wc % stride) return BXI_ERR_BAD_SHAPE;  // asserts at condinst_head.py:1400-1401
    const int K = size * size - 1;
    if (affinity && K > 32) return BXI_ERR_UNSUPPORTED;
    if (bt->B == 0) return BXI_OK;
    if (!bt->imgs || !rgb_small) return BXI_ERR_NULL_POINTER;
    const int h = bt->Hc / stride, w = bt->Wc / stride;
    const int64_t total = (int64_t)bt->B * h * w;
    if (!fits_i32(total * (K > 3 ? K : 3))) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = as_stream(stream);

    const bool vec = stride == 4 && (reinterpret_cast<uintptr_t>(bt->imgs) & 15) == 0;
    if (vec) {
        constexpr int BLOCK = 64;  // one wave per workgroup: 1600 workgroups at 2x800x1024 -> even CU fill
        hipLaunchKernelGGL((pool_rgb_s4_kernel<BLOCK>), dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                           s, bt->imgs, bt->B, bt->Hc, bt->Wc, meta, dn, rgb_small);
    } else {
        int64_t grid = (total + 255) / 256;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(pool_rgb_generic_kernel, dim3((unsigned)grid), dim3(256), 0, s, bt->imgs, bt->B, bt->Hc,
                           bt->Wc, stride, meta, dn, rgb_small);
    }
    st = check_launch();
    if (st != BXI_OK) return st;
    if (!sim && !affinity) return BXI_OK;

    const int R = size / 2 * dil;
    const size_t lds = sizeof(float) * 4 * (size_t)(kAffCols + 2 * R) * (size_t)(kAffRows + 2 * R);
    if (lds > 64 * 1024) return BXI_ERR_UNSUPPORTED;
    const int tiles = ((w + kAffCols - 1) / kAffCols) * ((h + kAffRows - 1) / kAffRows) * bt->B;
    if (K <= 8)
        hipLaunchKernelGGL((affinity_kernel<uint8_t>), dim3(tiles), dim3(256), lds, s, rgb_small, bt->B, h, w, stride,
                           size, dil, thresh, meta, bt->image_masks, bt->Hc, bt->Wc, sim, (uint8_t*)affinity);
    else
        hipLaunchKernelGGL((affinity_kernel<uint32_t>), dim3(tiles), dim3(256), lds, s, rgb_small, bt->B, h, w, stride,
                           size, dil, thresh, meta, bt->image_masks, bt->Hc, bt->Wc, sim, (uint32_t*)affinity);
    return check_launch();
}

int fill_gt_table(const float* const* boxes_per_img_host, const int* gt_count_host, int B, GtTable& gt, int& G) {
    if (B < 0 || B > BXI_MAX_IMAGES) return BXI_ERR_BAD_SHAPE;
    if (B > 0 && (!boxes_per_img_host || !gt_count_host)) return BXI_ERR_NULL_POINTER;
    gt.B = B;
    gt.first[0] = 0;
    for (int b = 0; b < B; ++b) {
        if (gt_count_host[b] < 0) return BXI_ERR_BAD_SHAPE;
        if (gt_count_host[b] > 0 && !boxes_per_img_host[b]) return BXI_ERR_NULL_POINTER;
        gt.boxes[b] = boxes_per_img_host[b];
        gt.first[b + 1] = gt.first[b] + gt_count_host[b];
    }
    for (int b = B; b < BXI_MAX_IMAGES; ++b) { gt.boxes[b] = nullptr; gt.first[b + 1] = gt.first[B]; }
    G = gt.first[B];
    return BXI_OK;
}

}  // namespace bxi

extern "C" {

int bxi_color_affinity_f32(const bxi_image_batch* batch_host, int stride, int size, int dilation, float color_thresh,
                           uint8_t* rgb_small, float* sim, void* affinity, void* stream) {
    return bxi::launch_color_affinity(batch_host, stride, size, dilation, color_thresh, rgb_small, sim, affinity,
                                      stream);
}

int bxi_box_bitmasks_f32(const float* const* boxes_per_img_host, const int* gt_count_host, int B, int Hc, int Wc,
                         int stride, int start, float* out, void* stream) {
    bxi::GtTable gt;
    int G = 0;
    int st = bxi::fill_gt_table(boxes_per_img_host, gt_count_host, B, gt, G);
    if (st != BXI_OK) return st;
    if (Hc <= 0 || Wc <= 0) return BXI_ERR_BAD_SHAPE;
    if (stride < 1 || start < 0 || start >= stride) return BXI_ERR_BAD_ARGUMENT;
    if (G == 0) return BXI_OK;
    if (!out) return BXI_ERR_NULL_POINTER;
    // bitmask_full[start::stride] has ceil((n-start)/stride) entries; the reference asserts n % stride == 0
    const int h = (Hc - start + stride - 1) / stride, w = (Wc - start + stride - 1) / stride;
    const int64_t P = (int64_t)h * w;
    int gx = (int)((P + 255) / 256);
    if (gx > 1024) gx = 1024;
    if (G > 65535) return BXI_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(bxi::box_bitmask_kernel, dim3(gx, G), dim3(256), 0, bxi::as_stream(stream), gt, G, Hc, Wc,
                       stride, start, h, w, out);
    return bxi::check_launch();
}

}  // extern "C"
