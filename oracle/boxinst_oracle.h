/*
 * boxinst_oracle.h -- CPU restatement of the BoxInst box-supervised mask-loss path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (boxinstseg_amd/) never falls back to it.
 *
 * Every function cites the reference file:line it follows (paths relative to the upstream
 * LiWentomng/BoxInstSeg checkout; "condinst_head.py" = mmdet/models/dense_heads/condinst_head.py,
 * "pairwise.cu" = mmdet/ops/pairwise/csrc/pairwise/pairwise.cu).
 *
 * Parity status
 *   PINNED   (against the reference's own functions, AST-extracted from condinst_head.py and
 *             executed in the build container -- see tests/golden/make_golden.py):
 *             pairwise term fwd/bwd, projection term, unfold order, colour similarity from Lab,
 *             loss glue, get_targets / get_bitmasks_from_boxes control flow.
 *            The pairwise op is additionally pinned BIT FOR BIT (f32 and f64, forward and backward)
 *            to the reference's own pairwise.cu kernels, compiled from the reference tree and run
 *            on the CPU (oracle/ref_wrap/pairwise_kernels_wrap.cpp, tests/golden/pairwise_refk.npz).
 *   UNPINNED (third-party code that is NOT in the reference tree and not installed here):
 *             mmcv.image.tensor2imgs / imdenormalize (OpenCV arithmetic) and
 *             skimage.color.rgb2lab.  Both are restated from their published algorithms;
 *             rgb2lab is checked against textbook CIE-Lab values only.
 */
#ifndef BOXINST_ORACLE_H
#define BOXINST_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- image side (always float32 in, as in the reference) ------------------------------------ */

/* condinst_head.py:170-186 (get_original_image) + :1365-1374 (zero pad to the batch canvas).
 * img: one normalised image on the canvas, [3, Hc, Wc] f32.  out: [3, Hc, Wc] u8-valued RGB,
 * zero outside img_h x img_w.  mean/std in the order of the tensor's channels (img_norm_cfg).
 * mmcv imdenormalize = cv2.multiply(img, std_f64) then cv2.add(img, mean_f64): each is a
 * double-precision op rounded to f32; then astype(uint8) (C truncation). */
void bxo_denormalize_u8(const float* img, int Hc, int Wc, int img_h, int img_w,
                        const double mean[3], const double std[3], int to_rgb, uint8_t* out);

/* condinst_head.py:1403 (F.avg_pool2d(k=stride)) + :1413 (.byte()).  in [3,Hc,Wc] u8, out [3,h,w] u8. */
void bxo_pool_u8(const uint8_t* rgb, int Hc, int Wc, int stride, uint8_t* out);

/* skimage.color.rgb2lab as called at condinst_head.py:1413 (illuminant D65, observer 2),
 * float64 maths, result cast to float32 (:1415-1416).  rgb planar [3,n] u8 -> lab planar [3,n] f32. */
void bxo_rgb2lab_u8(const uint8_t* rgb, int64_t n, float* lab);
/* same maths for a single colour, double result (known-answer tests) */
void bxo_rgb2lab_one(uint8_t r, uint8_t g, uint8_t b, double lab[3]);

/* condinst_head.py:1354-1369 + :1405: validity mask sampled at [start::stride, start::stride].
 * out [h,w] f32 in {0,1}. */
void bxo_image_mask(int Hc, int Wc, int img_h, int img_w, int rows_removed, int stride, float* out);

/* condinst_head.py:220-246 (get_image_color_similarity) with :190-217 (unfold_wo_center):
 * lab [3,h,w] f32, mask [h,w] f32 -> sim [K,h,w] f32, K = size*size-1. */
void bxo_color_similarity(const float* lab, const float* mask, int h, int w, int size, int dilation,
                          float* sim);

/* condinst_head.py:1426-1432: per-box bitmask sampled at [start::stride].  Python slice semantics
 * for int(x1):int(x2)+1 (negative indices wrap).  box = x1,y1,x2,y2.  out [h,w] f32. */
void bxo_box_bitmask(const float box[4], int Hc, int Wc, int stride, float* out);

/* ---- loss side: f32 and f64 flavours -------------------------------------------------------- */

/* pairwise.cu:68-104 (forward kernel) with :27-50 device maths.  logits [N,H,W] -> out [N,K,H,W]. */
void bxo_pairwise_nlog_fwd_f32(const float* logits, int N, int H, int W, int size, int dil, float* out);
void bxo_pairwise_nlog_fwd_f64(const double* logits, int N, int H, int W, int size, int dil, double* out);

/* pairwise.cu:106-149 (backward kernel) with :52-66; atomics replaced by in-order scatter adds.
 * g_logits is overwritten (zero-initialised inside, pairwise.cu:186). */
void bxo_pairwise_nlog_bwd_f32(const float* logits, const float* pairwise, const float* g_pairwise,
                               int N, int H, int W, int size, int dil, float* g_logits);
void bxo_pairwise_nlog_bwd_f64(const double* logits, const double* pairwise, const double* g_pairwise,
                               int N, int H, int W, int size, int dil, double* g_logits);

/* condinst_head.py:117-143 (dice_coefficient, compute_project_term) on scores = sigmoid(logits)
 * (:1300).  bitmask [N,H,W] in {0,1}.  Returns loss_prj; if g_logits != NULL ADDS g_out *
 * d loss_prj / d logits into it (arg-max: first index). */
float  bxo_project_term_f32(const float* logits, const float* bitmask, int N, int H, int W,
                            float g_out, float* g_logits);
double bxo_project_term_f64(const double* logits, const double* bitmask, int N, int H, int W,
                            double g_out, double* g_logits);

/* condinst_head.py:1314-1332 loss glue.  sim [N,K,H,W] (already gathered per instance),
 * bitmask [N,H,W].  losses[0]=loss_prj, losses[1]=loss_pairwise.  g_logits (nullable) receives
 * d(g_prj*loss_prj + g_pw*loss_pairwise)/d logits.  N == 0 -> both losses 0 (documented deviation
 * from the reference, which produces NaN / an invalid launch). */
void bxo_boxinst_loss_f32(const float* logits, const float* sim, const float* bitmask,
                          int N, int H, int W, int size, int dil, float color_thresh, float warmup,
                          float g_prj, float g_pw, float losses[2], float* g_logits);
void bxo_boxinst_loss_f64(const double* logits, const double* sim, const double* bitmask,
                          int N, int H, int W, int size, int dil, double color_thresh, double warmup,
                          double g_prj, double g_pw, double losses[2], double* g_logits);

/* ---- whole path: condinst_head.py:1288-1343 + :1345-1448 ------------------------------------ */
/* imgs [B,3,Hc,Wc] f32; img_hw [B,2]; rows_removed [B]; boxes [G,4]; gt_count [B] (sum = G);
 * gt_inds [N] (index into the batch-concatenated GT list); logits [N,h,w], h=Hc/stride.
 * Optional outputs (nullable): sim_out [B,K,h,w], bitmask_out [G,h,w].
 * n_threads > 1 parallelises over instances / images with OpenMP when built with -fopenmp. */
void bxo_boxinst_path_f32(const float* imgs, int B, int Hc, int Wc, const int* img_hw,
                          const int* rows_removed, const double mean[3], const double std[3], int to_rgb,
                          const float* boxes, const int* gt_count, const int64_t* gt_inds,
                          const float* logits, int N, int stride, int size, int dil,
                          float color_thresh, float warmup, float g_prj, float g_pw,
                          float losses[2], float* g_logits, float* sim_out, float* bitmask_out);

int bxo_max_threads(void);
void bxo_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
