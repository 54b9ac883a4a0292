// image_device.hpp -- device code shared by the image-side kernels (pool_rgb / stage1 / affinity / box).
//
// Reference semantics restated here (LiWentomng/BoxInstSeg, condinst_head.py =
// mmdet/models/dense_heads/condinst_head.py):
//   denorm_u8      get_original_image :170-186 (mmcv tensor2imgs -> imdenormalize -> astype(uint8))
//   pool + .byte() get_bitmasks_from_boxes :1403, :1413
//   rgb2lab_f32    skimage.color.rgb2lab as called at :1413, cast to f32 at :1415-1416
//   affinity_word  get_image_color_similarity :220-246 + the threshold of loss() :1324
#pragma once

#include "common.hpp"
#include "srgb_lut.h"

namespace bxi {

struct ImageMeta {
    int img_h[BXI_MAX_IMAGES];
    int img_w[BXI_MAX_IMAGES];
    int first_removed[BXI_MAX_IMAGES];  // rows >= this are zeroed in the validity mask (:1358-1363)
};

struct Denorm {
    double mean[3], stdv[3];
    int src_ch[3];  // output channel c (RGB) reads tensor channel src_ch[c]
};

struct PoolArgs {
    const float* imgs;   // [B,3,Hc,Wc]
    int B, Hc, Wc;
    ImageMeta meta;
    Denorm dn;
    uint8_t* rgb_small;  // [B,3,h,w] u8, nullable
    float* lab;          // [B,3,h,w] f32, nullable
};

static __device__ const double kSrgbLut[256] = BXI_SRGB_LUT_INIT;

__device__ __forceinline__ int denorm_u8(float x, double s, double m) {
    // OpenCV arithm_op: cv2.multiply against the float64 std row works in double (mul/div force the
    // scalar depth to CV_64F) and rounds to the f32 image; cv2.add demotes a float64 scalar to CV_32F
    // when the array is CV_32F, so the add is a plain f32 add of (float)mean.  astype(uint8) truncates.
    const float t = (float)((double)x * s);
    const float v = __fadd_rn(t, (float)m);
    return (int)v & 0xff;
}

// skimage.color.rgb2lab (rgb2xyz + xyz2lab, D65 / 2 deg) in fp64, no FMA contraction, result to f32.
// `lut` = the 256-entry inverse companding table (LDS copy or kSrgbLut).
__device__ __forceinline__ void rgb2lab_f32(const double* lut, int r8, int g8, int b8, float& L, float& A, float& Bv) {
    const double r = lut[r8], g = lut[g8], b = lut[b8];
    double f[3];
    const double M[3][3] = {{0.412453, 0.357580, 0.180423},
                            {0.212671, 0.715160, 0.072169},
                            {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double acc =
            __dadd_rn(__dadd_rn(__dmul_rn(M[i][0], r), __dmul_rn(M[i][1], g)), __dmul_rn(M[i][2], b));
        const double v = acc / white[i];
        f[i] = v > 0.008856 ? cbrt(v) : __dadd_rn(__dmul_rn(7.787, v), 16.0 / 116.0);
    }
    L = (float)__dadd_rn(__dmul_rn(116.0, f[1]), -16.0);
    A = (float)__dmul_rn(500.0, __dadd_rn(f[0], -f[1]));
    Bv = (float)__dmul_rn(200.0, __dadd_rn(f[1], -f[2]));
}

// One pooled pixel of the stride-4 vector path, in two halves so that the caller can put other
// latency (staging the LUT in LDS) between issuing the 12 loads and consuming them:
//   pool_load_s4  : 3 channels x 4 rows x float4, all issued back to back
//   pool_finish_s4: de-normalise, truncate, 4x4 sum >> 4, optional Lab
struct PoolRegs { float4 v[3][4]; };

__device__ __forceinline__ void pool_load_s4(const PoolArgs& pa, int64_t o, PoolRegs& pr) {
    const int h = pa.Hc >> 2, w = pa.Wc >> 2;
    const int c = (int)(o % w);
    const int r = (int)((o / w) % h);
    const int b = (int)(o / ((int64_t)w * h));
    const int64_t plane = (int64_t)pa.Hc * pa.Wc;
    const float* base = pa.imgs + (int64_t)b * 3 * plane + (int64_t)(4 * r) * pa.Wc + 4 * c;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            pr.v[ch][i] = *reinterpret_cast<const float4*>(base + pa.dn.src_ch[ch] * plane + (int64_t)i * pa.Wc);
}

__device__ __forceinline__ void pool_finish_s4(const PoolArgs& pa, int64_t o, const PoolRegs& pr, const double* lut) {
    const int h = pa.Hc >> 2, w = pa.Wc >> 2;
    const int c = (int)(o % w);
    const int r = (int)((o / w) % h);
    const int b = (int)(o / ((int64_t)w * h));
    const int ih = pa.meta.img_h[b], iw = pa.meta.img_w[b];
    const int x0 = 4 * c, y0 = 4 * r;
    int px[3];
    const bool inside = y0 + 3 < ih && x0 + 3 < iw;      // the whole 4x4 window is image, not canvas padding
    if (__all(inside)) {                                  // wave-uniform: no per-element selects (the common case)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
            int sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sum += denorm_u8(pr.v[ch][i].x, s, m) + denorm_u8(pr.v[ch][i].y, s, m) + denorm_u8(pr.v[ch][i].z, s, m) +
                       denorm_u8(pr.v[ch][i].w, s, m);
            px[ch] = sum >> 4;
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
            int sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool yin = (y0 + i) < ih;
                sum += (yin && x0 + 0 < iw) ? denorm_u8(pr.v[ch][i].x, s, m) : 0;
                sum += (yin && x0 + 1 < iw) ? denorm_u8(pr.v[ch][i].y, s, m) : 0;
                sum += (yin && x0 + 2 < iw) ? denorm_u8(pr.v[ch][i].z, s, m) : 0;
                sum += (yin && x0 + 3 < iw) ? denorm_u8(pr.v[ch][i].w, s, m) : 0;
            }
            px[ch] = sum >> 4;
        }
    }
    const int64_t P = (int64_t)h * w;
    const int64_t p = (int64_t)r * w + c;
    if (pa.rgb_small) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) pa.rgb_small[((int64_t)b * 3 + ch) * P + p] = (uint8_t)px[ch];
    }
    if (pa.lab) {
        float L, A, Bv;
        rgb2lab_f32(lut, px[0], px[1], px[2], L, A, Bv);
        float* o3 = pa.lab + (int64_t)b * 3 * P + p;
        o3[0] = L; o3[P] = A; o3[2 * P] = Bv;
    }
}

// validity of pooled pixel (rr,cc) of image b: the padded mask sampled at [start::stride] (:1354-1369, :1405)
__device__ __forceinline__ float geom_mask(const ImageMeta& meta, int b, int rr, int cc, int stride) {
    const int y = rr * stride + stride / 2, x = cc * stride + stride / 2;
    return (y < meta.img_h[b] && x < meta.img_w[b] && y < meta.first_removed[b]) ? 1.f : 0.f;
}

// similarity of one pair, exactly get_image_color_similarity's arithmetic in f32, no contraction (:237,:246)
__device__ __forceinline__ float color_sim(float L0, float A0, float B0, float L1, float A1, float B1, float m1) {
    const float dL = L0 - L1, dA = A0 - A1, dB = B0 - B1;
    const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(dL, dL), __fmul_rn(dA, dA)), __fmul_rn(dB, dB));
    return __fmul_rn(expf(__fmul_rn(-__fsqrt_rn(n2), 0.5f)), m1);
}

}  // namespace bxi
