/*
 * boxinst_oracle.c -- CPU restatement (plain C) of the BoxInst mask-loss path.
 * TEST INFRASTRUCTURE ONLY; see boxinst_oracle.h for the parity status of each stage.
 *
 * Build:  make -C oracle          (gcc -O2 -fopenmp -shared -> oracle/_build/libboxinst_oracle.so)
 */
#include "boxinst_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int bxo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void bxo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* image side                                                                                  */
/* ------------------------------------------------------------------------------------------- */

/* condinst_head.py:170-186.  tensor2imgs(x, mean, std, to_rgb) = imdenormalize(x, mean, std,
 * to_bgr=to_rgb).astype(uint8): multiply by std, add mean (channel c with mean[c]), optional
 * RGB->BGR flip; the caller then reverses the channel axis again (:182-183).  Net effect:
 *   to_rgb=True : out channel c = trunc(x[c]*std[c]+mean[c])
 *   to_rgb=False: out channel c = trunc(x[2-c]*std[2-c]+mean[2-c])
 * mmcv passes mean/std as 1x3 float64 rows, which OpenCV's arithm_op treats as scalars:
 *   - cv2.multiply: mul/div force the scalar depth to CV_64F, so the work type is double and the
 *     product is rounded once to the float32 destination;
 *   - cv2.add: a CV_64F scalar against a CV_32F array is demoted to CV_32F, so the sum is a plain
 *     float32 add of (float)mean.
 * (UNPINNED: restated from OpenCV's arithm.cpp; cv2 is not available here to confirm.) */
void bxo_denormalize_u8(const float* img, int Hc, int Wc, int img_h, int img_w,
                        const double mean[3], const double std[3], int to_rgb, uint8_t* out) {
    const int64_t P = (int64_t)Hc * Wc;
    memset(out, 0, (size_t)(3 * P));
    for (int c = 0; c < 3; ++c) {
        const int sc = to_rgb ? c : 2 - c;
        const float* src = img + sc * P;
        uint8_t* dst = out + c * P;
        for (int y = 0; y < img_h && y < Hc; ++y)
            for (int x = 0; x < img_w && x < Wc; ++x) {
                float t = (float)((double)src[(int64_t)y * Wc + x] * std[sc]);
                const float mf = (float)mean[sc];
                float v = t + mf; /* t is already rounded to f32: no contraction possible */
                dst[(int64_t)y * Wc + x] = (uint8_t)(int32_t)v; /* numpy astype(uint8) on x86 */
            }
    }
}

/* condinst_head.py:1403 + :1413.  avg_pool2d sums the window in f32 (exact for <= 2^24) and
 * divides by stride^2; .byte() truncates. */
void bxo_pool_u8(const uint8_t* rgb, int Hc, int Wc, int stride, uint8_t* out) {
    const int h = Hc / stride, w = Wc / stride;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < h; ++r)
            for (int q = 0; q < w; ++q) {
                float s = 0.f;
                for (int i = 0; i < stride; ++i)
                    for (int j = 0; j < stride; ++j)
                        s += (float)rgb[((int64_t)c * Hc + (r * stride + i)) * Wc + (q * stride + j)];
                float avg = s / (float)(stride * stride);
                out[((int64_t)c * h + r) * w + q] = (uint8_t)(int32_t)avg;
            }
}

/* skimage.color.rgb2lab (rgb2xyz -> xyz2lab, D65 / 2 deg), float64. SURVEY App. B. */
void bxo_rgb2lab_one(uint8_t r8, uint8_t g8, uint8_t b8, double lab[3]) {
    double c[3] = { r8 / 255.0, g8 / 255.0, b8 / 255.0 };
    for (int i = 0; i < 3; ++i)
        c[i] = c[i] > 0.04045 ? pow((c[i] + 0.055) / 1.055, 2.4) : c[i] / 12.92;
    static const double M[3][3] = { { 0.412453, 0.357580, 0.180423 },
                                    { 0.212671, 0.715160, 0.072169 },
                                    { 0.019334, 0.119193, 0.950227 } };
    static const double white[3] = { 0.95047, 1.0, 1.08883 };
    double f[3];
    for (int i = 0; i < 3; ++i) {
        double v = (M[i][0] * c[0] + M[i][1] * c[1] + M[i][2] * c[2]) / white[i];
        f[i] = v > 0.008856 ? cbrt(v) : 7.787 * v + 16.0 / 116.0;
    }
    lab[0] = 116.0 * f[1] - 16.0;
    lab[1] = 500.0 * (f[0] - f[1]);
    lab[2] = 200.0 * (f[1] - f[2]);
}

void bxo_rgb2lab_u8(const uint8_t* rgb, int64_t n, float* lab) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double v[3];
        bxo_rgb2lab_one(rgb[i], rgb[n + i], rgb[2 * n + i], v);
        lab[i] = (float)v[0]; lab[n + i] = (float)v[1]; lab[2 * n + i] = (float)v[2];
    }
}

/* condinst_head.py:1354-1369, :1405 */
void bxo_image_mask(int Hc, int Wc, int img_h, int img_w, int rows_removed, int stride, float* out) {
    const int h = Hc / stride, w = Wc / stride, start = stride / 2;
    /* original_image_masks[-pixels_removed:, :] = 0 : python slice, start clamps at row 0 */
    int first_removed = rows_removed > 0 ? img_h - rows_removed : img_h;
    if (first_removed < 0) first_removed = 0;
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            int y = r * stride + start, x = c * stride + start;
            out[(int64_t)r * w + c] = (y < img_h && x < img_w && y < first_removed) ? 1.f : 0.f;
        }
}

/* condinst_head.py:220-246 with the unfold order of :190-217 (row-major kernel window, centre
 * dropped).  Zero padding: an out-of-canvas neighbour has Lab = 0 and mask = 0. */
void bxo_color_similarity(const float* lab, const float* mask, int h, int w, int size, int dilation,
                          float* sim) {
    const int R = size / 2 * dilation;
    const int64_t P = (int64_t)h * w;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            int k = 0;
            for (int dy = -R; dy <= R; dy += dilation)
                for (int dx = -R; dx <= R; dx += dilation) {
                    if (dx == 0 && dy == 0) continue;
                    int r2 = r + dy, c2 = c + dx;
                    int valid = (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w);
                    float acc = 0.f;
                    for (int ch = 0; ch < 3; ++ch) {
                        float a = lab[ch * P + (int64_t)r * w + c];
                        float b = valid ? lab[ch * P + (int64_t)r2 * w + c2] : 0.f;
                        float d = a - b;
                        acc += d * d;
                    }
                    float s = expf(-sqrtf(acc) * 0.5f);
                    float wv = valid ? mask[(int64_t)r2 * w + c2] : 0.f;
                    sim[(int64_t)k * P + (int64_t)r * w + c] = s * wv;
                    ++k;
                }
        }
}

/* python slice [a:b] on an axis of length n -> [lo,hi) */
static void py_slice(int a, int b, int n, int* lo, int* hi) {
    if (a < 0) a += n;
    if (a < 0) a = 0;
    if (a > n) a = n;
    if (b < 0) b += n;
    if (b < 0) b = 0;
    if (b > n) b = n;
    *lo = a; *hi = b > a ? b : a;
}

/* condinst_head.py:1426-1432 */
void bxo_box_bitmask(const float box[4], int Hc, int Wc, int stride, float* out) {
    const int h = Hc / stride, w = Wc / stride, start = stride / 2;
    int y0, y1, x0, x1;
    py_slice((int)box[1], (int)box[3] + 1, Hc, &y0, &y1);
    py_slice((int)box[0], (int)box[2] + 1, Wc, &x0, &x1);
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            int y = r * stride + start, x = c * stride + start;
            out[(int64_t)r * w + c] = (y >= y0 && y < y1 && x >= x0 && x < x1) ? 1.f : 0.f;
        }
}

/* ------------------------------------------------------------------------------------------- */
/* loss side: two precisions from one source                                                   */
/* ------------------------------------------------------------------------------------------- */
#define REAL float
#define SUF(x) x##_f32
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#define LOGSIG_CUT -20.0
#include "loss_terms.inc"
#undef REAL
#undef SUF
#undef R_EXP
#undef R_LOG
#undef R_SQRT
#undef LOGSIG_CUT

#define REAL double
#define SUF(x) x##_f64
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#define LOGSIG_CUT -40.0
#include "loss_terms.inc"
#undef REAL
#undef SUF
#undef R_EXP
#undef R_LOG
#undef R_SQRT
#undef LOGSIG_CUT

/* ------------------------------------------------------------------------------------------- */
/* whole path                                                                                  */
/* ------------------------------------------------------------------------------------------- */
void bxo_boxinst_path_f32(const float* imgs, int B, int Hc, int Wc, const int* img_hw,
                          const int* rows_removed, const double mean[3], const double std[3], int to_rgb,
                          const float* boxes, const int* gt_count, const int64_t* gt_inds,
                          const float* logits, int N, int stride, int size, int dil,
                          float color_thresh, float warmup, float g_prj, float g_pw,
                          float losses[2], float* g_logits, float* sim_out, float* bitmask_out) {
    const int h = Hc / stride, w = Wc / stride, K = size * size - 1;
    const int64_t P = (int64_t)h * w, Pf = (int64_t)Hc * Wc;
    int G = 0;
    for (int b = 0; b < B; ++b) G += gt_count[b];

    /* get_targets (:1345-1393) -> get_bitmasks_from_boxes (:1395-1448) */
    float* sim = sim_out ? sim_out : (float*)malloc(sizeof(float) * (size_t)B * K * P);
    float* bitm = bitmask_out ? bitmask_out : (float*)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1) * P);
    int* img_of_gt = (int*)malloc(sizeof(int) * (size_t)(G > 0 ? G : 1));
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        uint8_t* full = (uint8_t*)malloc((size_t)(3 * Pf));
        uint8_t* small = (uint8_t*)malloc((size_t)(3 * P));
        float* lab = (float*)malloc(sizeof(float) * (size_t)(3 * P));
        float* m = (float*)malloc(sizeof(float) * (size_t)P);
        bxo_denormalize_u8(imgs + (int64_t)b * 3 * Pf, Hc, Wc, img_hw[2 * b], img_hw[2 * b + 1],
                           mean, std, to_rgb, full);
        bxo_pool_u8(full, Hc, Wc, stride, small);
        bxo_rgb2lab_u8(small, P, lab);
        bxo_image_mask(Hc, Wc, img_hw[2 * b], img_hw[2 * b + 1], rows_removed[b], stride, m);
        bxo_color_similarity(lab, m, h, w, size, dil, sim + (int64_t)b * K * P);
        free(full); free(small); free(lab); free(m);
    }
    {
        int g = 0;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < gt_count[b]; ++i, ++g) {
                img_of_gt[g] = b;
                bxo_box_bitmask(boxes + 4 * (int64_t)g, Hc, Wc, stride, bitm + (int64_t)g * P);
            }
    }

    /* loss (:1300-1302, :1315-1317): gather per-instance similarity and bitmask */
    losses[0] = losses[1] = 0.f;
    if (N > 0) {
        float* sim_n = (float*)malloc(sizeof(float) * (size_t)N * K * P);
        float* bit_n = (float*)malloc(sizeof(float) * (size_t)N * P);
        for (int n = 0; n < N; ++n) {
            int g = (int)gt_inds[n];
            memcpy(sim_n + (int64_t)n * K * P, sim + (int64_t)img_of_gt[g] * K * P, sizeof(float) * (size_t)(K * P));
            memcpy(bit_n + (int64_t)n * P, bitm + (int64_t)g * P, sizeof(float) * (size_t)P);
        }
        bxo_boxinst_loss_f32(logits, sim_n, bit_n, N, h, w, size, dil, color_thresh, warmup,
                             g_prj, g_pw, losses, g_logits);
        free(sim_n); free(bit_n);
    }
    if (!sim_out) free(sim);
    if (!bitmask_out) free(bitm);
    free(img_of_gt);
}
