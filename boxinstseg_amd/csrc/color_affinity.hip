// color_affinity.hip -- target side of the BoxInst loss on gfx950, fully device-resident.
//
// Replaces (reference, LiWentomng/BoxInstSeg):
//   get_original_image            condinst_head.py:170-186  (mmcv tensor2imgs: GPU->CPU->GPU per image)
//   CondInstMaskHead.get_targets  condinst_head.py:1345-1393
//   get_bitmasks_from_boxes       condinst_head.py:1395-1448 (skimage rgb2lab on CPU, per-box loops)
//   get_image_color_similarity    condinst_head.py:220-246  (two F.unfold materialisations)
//
// Kernel A  pool_rgb      : imgs [B,3,Hc,Wc] f32 -> lab [B,3,h,w] f32 (+ rgb_small [B,3,h,w] u8 on request)
//     de-normalise (double mul, round to f32, double add, round to f32 = OpenCV's arithmetic on
//     an f32 image with f64 scalars), truncate to u8, sum the stride x stride window, >> log2.
//     Pure stream: 12 B read per input pixel (3 x f32), 3/stride^2 B written.  HBM roofline.
//     One lane = one output pixel = `stride` rows x 16 B per channel, so a wave reads 1 KiB
//     contiguous per load instruction; all 3*stride loads are issued before the first use.
//     then rgb -> Lab per pooled pixel (skimage algorithm, fp64, LUT companding staged in LDS).
//     In the fused evaluation this body runs inside stage1_kernel (mask_loss.hip), next to the
//     logit-streaming workgroups; the stand-alone kernel here serves get_targets().
// Kernel B  affinity      : Lab tile + halo in LDS -> K neighbour distances -> sim [B,K,h,w] f32
//     and/or K-bit threshold mask per pixel.  Reads 12 B / pixel, writes 4K B / pixel (sim) +
//     1 B (mask): write-bound, tiny.  Only get_targets() needs it: the fused loss derives the
//     bits it needs from Lab inside box_kernel.
#include "image_device.hpp"

namespace bxi {

// ---- Kernel A, stride 4, vector path ---------------------------------------------------------
__global__ __launch_bounds__(256) void pool_rgb_s4_kernel(PoolArgs pa) {
    __shared__ double lut[256];
    const int64_t total = (int64_t)pa.B * (pa.Hc >> 2) * (pa.Wc >> 2);
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    PoolRegs pr;
    if (o < total) pool_load_s4(pa, o, pr);
    lut[threadIdx.x] = kSrgbLut[threadIdx.x];
    __syncthreads();
    if (o < total) pool_finish_s4(pa, o, pr, lut);
}

// ---- Kernel A, any stride, scalar path (unaligned canvases, stride 1/2/8) ---------------------
__global__ __launch_bounds__(256) void pool_rgb_generic_kernel(PoolArgs pa, int stride) {
    const int Hc = pa.Hc, Wc = pa.Wc;
    const int h = Hc / stride, w = Wc / stride;
    const int64_t total = (int64_t)pa.B * h * w;
    const int64_t plane = (int64_t)Hc * Wc;
    const int64_t P = (int64_t)h * w;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % w);
        const int r = (int)((o / w) % h);
        const int b = (int)(o / ((int64_t)w * h));
        const int ih = pa.meta.img_h[b], iw = pa.meta.img_w[b];
        int px[3];
        for (int ch = 0; ch < 3; ++ch) {
            const int sc = pa.dn.src_ch[ch];
            const double s = pa.dn.stdv[sc], m = pa.dn.mean[sc];
            const float* p = pa.imgs + ((int64_t)b * 3 + sc) * plane;
            float sum = 0.f;  // F.avg_pool2d accumulates in f32; exact for u8-valued inputs
            for (int i = 0; i < stride; ++i)
                for (int j = 0; j < stride; ++j) {
                    const int y = r * stride + i, x = c * stride + j;
                    if (y < ih && x < iw) sum += (float)denorm_u8(p[(int64_t)y * Wc + x], s, m);
                }
            px[ch] = (int)(sum / (float)(stride * stride)) & 0xff;
            if (pa.rgb_small) pa.rgb_small[((int64_t)b * 3 + ch) * P + (int64_t)r * w + c] = (uint8_t)px[ch];
        }
        if (pa.lab) {
            float L, A, Bv;
            rgb2lab_f32(kSrgbLut, px[0], px[1], px[2], L, A, Bv);
            float* o3 = pa.lab + (int64_t)b * 3 * P + (int64_t)r * w + c;
            o3[0] = L; o3[P] = A; o3[2 * P] = Bv;
        }
    }
}

// ---- Kernel B: Lab -> similarity / affinity bits ----------------------------------------------
constexpr int kAffRows = 4, kAffCols = 64;  // output tile of the affinity kernel (256 threads)

template <typename BitsT>
__global__ __launch_bounds__(256) void affinity_kernel(const float* __restrict__ lab, int B, int h, int w,
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
    const float* src = lab + (int64_t)b * 3 * P;
    const int start = stride / 2;

    for (int i = threadIdx.x; i < TW * TH; i += blockDim.x) {
        const int rr = r0 - R + i / TW, cc = c0 - R + i % TW;
        float L = 0.f, A = 0.f, Bv = 0.f, m = 0.f;  // zero padding of F.unfold (condinst_head.py:203-207)
        if (rr >= 0 && rr < h && cc >= 0 && cc < w) {
            const int64_t p = (int64_t)rr * w + cc;
            L = src[p]; A = src[P + p]; Bv = src[2 * P + p];
            if (image_masks)
                m = image_masks[((int64_t)b * Hc + (rr * stride + start)) * Wc + (cc * stride + start)];
            else
                m = geom_mask(meta, b, rr, cc, stride);
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
            const float s = color_sim(L0, A0, B0, labL[qi], labA[qi], labB[qi], msk[qi]);
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

int fill_pool_args(const bxi_image_batch* bt, uint8_t* rgb_small, float* lab, PoolArgs& pa) {
    int st = fill_image_meta(bt, pa.meta, pa.dn);
    if (st != BXI_OK) return st;
    pa.imgs = bt->imgs; pa.B = bt->B; pa.Hc = bt->Hc; pa.Wc = bt->Wc;
    pa.rgb_small = rgb_small; pa.lab = lab;
    return BXI_OK;
}

bool pool_vec_ok(const bxi_image_batch* bt, int stride) {
    return stride == 4 && (bt->Wc & 3) == 0 && (reinterpret_cast<uintptr_t>(bt->imgs) & 15) == 0;
}

int launch_pool(const bxi_image_batch* bt, int stride, uint8_t* rgb_small, float* lab, hipStream_t s) {
    PoolArgs pa;
    int st = fill_pool_args(bt, rgb_small, lab, pa);
    if (st != BXI_OK) return st;
    const int h = bt->Hc / stride, w = bt->Wc / stride;
    const int64_t total = (int64_t)bt->B * h * w;
    if (pool_vec_ok(bt, stride)) {
        BXI_LAUNCH("pool_rgb", s, pool_rgb_s4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pa);
    } else {
        int64_t grid = (total + 255) / 256;
        if (grid > 4096) grid = 4096;
        BXI_LAUNCH("pool_rgb", s, pool_rgb_generic_kernel, dim3((unsigned)grid), dim3(256), 0, s, pa, stride);
    }
    return check_launch();
}

int launch_color_affinity(const bxi_image_batch* bt, int stride, int size, int dil, float thresh, float* lab,
                          uint8_t* rgb_small, float* sim, void* affinity, void* stream) {
    if (!bt) return BXI_ERR_NULL_POINTER;
    if (stride < 1 || size < 1 || (size & 1) == 0 || dil < 1) return BXI_ERR_BAD_ARGUMENT;
    if (bt->B < 0 || bt->B > BXI_MAX_IMAGES || bt->Hc <= 0 || bt->Wc <= 0) return BXI_ERR_BAD_SHAPE;
    if (bt->Hc % stride || bt->Wc % stride) return BXI_ERR_BAD_SHAPE;  // asserts at condinst_head.py:1400-1401
    const int K = size * size - 1;
    if (affinity && K > 32) return BXI_ERR_UNSUPPORTED;
    if (bt->B == 0) { ImageMeta m; Denorm d; return fill_image_meta(bt, m, d); }
    if (!bt->imgs || !lab) return BXI_ERR_NULL_POINTER;
    const int h = bt->Hc / stride, w = bt->Wc / stride;
    const int64_t total = (int64_t)bt->B * h * w;
    if (!fits_i32(total * (K > 3 ? K : 3))) return BXI_ERR_BAD_SHAPE;
    hipStream_t s = as_stream(stream);
    int st = launch_pool(bt, stride, rgb_small, lab, s);
    if (st != BXI_OK) return st;
    if (!sim && !affinity) return BXI_OK;

    ImageMeta meta;
    Denorm dn;
    fill_image_meta(bt, meta, dn);
    const int R = size / 2 * dil;
    const size_t lds = sizeof(float) * 4 * (size_t)(kAffCols + 2 * R) * (size_t)(kAffRows + 2 * R);
    if (lds > 64 * 1024) return BXI_ERR_UNSUPPORTED;
    const int tiles = ((w + kAffCols - 1) / kAffCols) * ((h + kAffRows - 1) / kAffRows) * bt->B;
    if (K <= 8)
        BXI_LAUNCH("affinity", s, (affinity_kernel<uint8_t>), dim3(tiles), dim3(256), lds, s, lab, bt->B, h, w, stride,
                   size, dil, thresh, meta, bt->image_masks, bt->Hc, bt->Wc, sim, (uint8_t*)affinity);
    else
        BXI_LAUNCH("affinity", s, (affinity_kernel<uint32_t>), dim3(tiles), dim3(256), lds, s, lab, bt->B, h, w, stride,
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
                           float* lab, uint8_t* rgb_small, float* sim, void* affinity, void* stream) {
    return bxi::launch_color_affinity(batch_host, stride, size, dilation, color_thresh, lab, rgb_small, sim,
                                      affinity, stream);
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
    BXI_LAUNCH("box_bitmask", bxi::as_stream(stream), bxi::box_bitmask_kernel, dim3(gx, G), dim3(256), 0, bxi::as_stream(stream), gt, G, Hc, Wc,
                       stride, start, h, w, out);
    return bxi::check_launch();
}

}  // extern "C"
