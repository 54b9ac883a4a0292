/*
 * boxinst_hip.h -- C ABI of libboxinst_hip.so: the BoxInst box-supervised mask-loss path
 * (projection term + colour-similarity pairwise term over CondInst mask logits) as hand-written
 * HIP kernels for gfx950 (MI355X / CDNA4).
 *
 * This is the drop-in boundary.  Every entry point names the interface of the reference
 * (LiWentomng/BoxInstSeg) it replaces; paths are relative to the upstream checkout:
 *   bind.cpp          = mmdet/ops/pairwise/csrc/pairwise/bind.cpp
 *   pairwise.cu       = mmdet/ops/pairwise/csrc/pairwise/pairwise.cu
 *   condinst_head.py  = mmdet/models/dense_heads/condinst_head.py
 *
 * Conventions
 *   - plain C types only; no torch / ATen types cross this boundary.
 *   - every pointer is a DEVICE pointer owned by the caller unless its name ends in `_host`.
 *   - allocation-free: outputs and workspace are supplied by the caller (size from *_workspace_bytes).
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); no host synchronisation, no hidden copies.  hipGraph capture: every entry point may be
 *     captured and the graph replayed with the CONTENTS of its buffers changed between replays -- nothing a
 *     replay must renew is a kernel argument (the evaluation's per-call tag lives in its workspace, section 3;
 *     a warm-up factor that changes per replay is read from `iter_counter`: pass warmup < 0).
 *   - return value: BXI_OK (0) or a negative bxi_status; never throws.  The reference reports the
 *     same conditions by TORCH_CHECK / AT_CUDA_CHECK (pairwise.cu:7-13,173,200); the Python
 *     shim turns a non-zero status into RuntimeError.
 *   - re-entrant, no process-wide mutable state, and nothing is read from the process environment: what varies is an
 *     argument (`flags` of the evaluation).  The few developer hooks (launch bracketing for benchmarks, a test switch of
 *     the tree filter) are declared in boxinst_hip_dev.h, not here; the A/B knobs of tools/ exist only in a -DBXI_DEV
 *     build.  One host thread per device is the expected use.
 *   - tensors are dense, row-major (NCHW like the reference), fp32 unless the name says _f64.
 */
#ifndef BOXINST_HIP_H
#define BOXINST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BXI_ABI_VERSION 7
#define BXI_MAX_IMAGES 64   /* images per call (per-image metadata travels in kernel arguments) */

typedef enum bxi_status {
    BXI_OK = 0,
    BXI_ERR_NULL_POINTER = -1,   /* a required pointer is NULL                                  */
    BXI_ERR_BAD_SHAPE = -2,      /* negative/zero/inconsistent dimension, or > 2^31-1 elements  */
    BXI_ERR_BAD_ARGUMENT = -3,   /* even pairwise_size, dilation < 1, stride < 1 ...            */
    BXI_ERR_UNSUPPORTED = -4,    /* valid for the reference op, outside this build's fast path  */
    BXI_ERR_WORKSPACE = -5,      /* workspace NULL / too small / misaligned                     */
    BXI_ERR_LAUNCH = -6,         /* hipGetLastError() != hipSuccess after a launch             */
    BXI_ERR_NO_DEVICE = -7       /* no HIP device / wrong architecture                          */
} bxi_status;

int bxi_abi_version(void);
const char* bxi_status_string(int status);
/* hipError_t of the most recent BXI_ERR_LAUNCH on this host thread (0 if none). */
int bxi_last_hip_error(void);
/* 0 if device `ordinal` exists and is gfx950, BXI_ERR_NO_DEVICE otherwise. */
int bxi_check_device(int ordinal);

/* ===========================================================================================
 * 1. Op level -- replaces the pybind11 module `pairwise_ext` (bind.cpp:15-36)
 * ===========================================================================================
 *
 * bxi_pairwise_nlog_forward_*  <->  pairwise_nlog_forward(int size, int dilation, Tensor& logits)
 *     bind.cpp:15-20 -> pairwiseNLogForwardCUDALauncher pairwise.cu:154-175, kernel :68-104.
 *     logits   [N,1,H,W]  (channel dim must be 1, pairwise.cu:93)
 *     pairwise [N,size*size-1,H,W] = -log P(y_p == y_q), q = p + (dy,dx), dy outer / dx inner in
 *     steps of `dilation`, centre skipped; an out-of-bounds neighbour gives 0.
 *     Differences from the reference: runs on `stream` instead of the legacy default stream
 *     (SURVEY 8a quirk 2); N == 0 is a no-op instead of an invalid launch.
 */
int bxi_pairwise_nlog_forward_f32(const float* logits, int N, int H, int W, int size, int dilation,
                                  float* pairwise, void* stream);
int bxi_pairwise_nlog_forward_f64(const double* logits, int N, int H, int W, int size, int dilation,
                                  double* pairwise, void* stream);

/* bxi_pairwise_nlog_backward_*  <->  pairwise_nlog_backward(size, dilation, logits, pairwise, g_pairwise)
 *     bind.cpp:22-29 -> pairwiseNLogBackwardCUDALauncher pairwise.cu:177-202, kernel :106-149.
 *     g_logits [N,1,H,W] is fully overwritten (the reference zero-fills then atomically adds,
 *     pairwise.cu:186,62-65).  Here each pixel GATHERS its 2*(size*size-1) contributions using
 *     f(x,y) = f(y,x) and channel K-1-k = the opposite offset, so the sum order is fixed and the
 *     result is deterministic.  `pairwise` (the saved forward output) is accepted for signature
 *     parity and may be NULL: the kernel recomputes the pair value.
 */
int bxi_pairwise_nlog_backward_f32(const float* logits, const float* pairwise, const float* g_pairwise,
                                   int N, int H, int W, int size, int dilation, float* g_logits,
                                   void* stream);
int bxi_pairwise_nlog_backward_f64(const double* logits, const double* pairwise, const double* g_pairwise,
                                   int N, int H, int W, int size, int dilation, double* g_logits,
                                   void* stream);

/* ===========================================================================================
 * 2. Target side -- replaces get_original_image (condinst_head.py:170-186),
 *    CondInstMaskHead.get_targets (:1345-1393), get_bitmasks_from_boxes (:1395-1448) and
 *    get_image_color_similarity (:220-246); no host round trip, no per-image / per-box loop.
 * ===========================================================================================
 */
typedef struct bxi_image_batch {
    const float* imgs;        /* [B,3,Hc,Wc] network input (normalised), canvas-padded           */
    int B, Hc, Wc;
    const int* img_h_host;    /* [B] img_metas[i]['img_shape'][0]                                 */
    const int* img_w_host;    /* [B] img_metas[i]['img_shape'][1]                                 */
    const int* rows_removed_host; /* [B] int(bottom_pixels_removed*img_h/ori_h), :1358-1361       */
    double mean[3], std[3];   /* img_norm_cfg, in the channel order of `imgs`                     */
    int to_rgb;               /* img_norm_cfg['to_rgb']                                           */
    const float* image_masks; /* optional [B,Hc,Wc] explicit validity masks (the padded_image_masks
                                 argument of get_bitmasks_from_boxes); NULL = derive from the
                                 img_h/img_w/rows_removed geometry as get_targets does            */
} bxi_image_batch;

/* Stage A (pool_rgb): de-normalise + truncate to uint8 (:170-186), stride x stride mean + .byte()
 *   (:1403,1413), rgb2lab (skimage algorithm, fp64 -> f32, :1413-1416):
 *   lab       [B,3,h,w] f32, h = Hc/stride   (required; also the scratch stage B reads)
 *   rgb_small [B,3,h,w] uint8                (nullable)
 * Stage B (affinity): colour similarity (:220-246):
 *   sim      [B,K,h,w] f32, K = size*size-1 (nullable: skip the 4*K bytes/pixel write)
 *   affinity [B,h,w]  bit k of the low K bits = (sim[k] >= color_thresh) (:1324); uint8 per pixel
 *            when K <= 8, uint32 when K <= 32 (nullable).
 * size must be odd; K <= 32 for `affinity`; any stride >= 1 dividing Hc and Wc.
 */
int bxi_color_affinity_f32(const bxi_image_batch* batch_host, int stride, int size, int dilation,
                           float color_thresh, float* lab, uint8_t* rgb_small, float* sim, void* affinity,
                           void* stream);

/* Per-box bitmasks (:1426-1432): out [G, Hc/stride, Wc/stride] f32 in {0,1},
 * out[g,r,c] = 1 iff (start+r*stride, start+c*stride) lies in the python slice
 * [int(y1):int(y2)+1, int(x1):int(x2)+1] of an Hc x Wc array.  stride=1,start=0 gives the
 * reference's `bitmasks_full`.  boxes_per_img_host: B device pointers to [G_i,4] xyxy f32. */
int bxi_box_bitmasks_f32(const float* const* boxes_per_img_host, const int* gt_count_host, int B,
                         int Hc, int Wc, int stride, int start, float* out, void* stream);

/* ===========================================================================================
 * 3. Loss -- replaces CondInstMaskHead.loss with boxinst_enabled (condinst_head.py:1288-1343):
 *    compute_project_term (:134-143) + pairwise_nlog + weights / normalise / warm-up
 *    (:1315-1332), forward AND backward to mask_logits in one pass over the logits.
 * ===========================================================================================
 */
typedef struct bxi_instances {
    const float* logits;      /* [N,1,h,w] mask logits                                            */
    int N, h, w;
    const int64_t* gt_inds;   /* [N] index into the batch-concatenated GT list (:1302,1316)       */
    const float* const* boxes_per_img_host; /* [B] device pointers to [G_i,4] xyxy boxes          */
    const int* gt_count_host; /* [B] G_i                                                          */
    int B;
    int Hc, Wc;               /* canvas size in pixels (h*stride, w*stride)                        */
    int stride;               /* out_stride                                                       */
    float* iter_counter;      /* optional (NULL = none): device float the evaluation entry points
                                 (bxi_boxinst_eval_f32, bxi_boxinst_head_eval_f32) add 1.0f to, inside
                                 their last launch -- `self._iter += 1`, condinst_head.py:1297, without
                                 a launch of its own.  Other entry points ignore it.                */
} bxi_instances;

size_t bxi_boxinst_loss_workspace_bytes(int N, int h, int w);
size_t bxi_boxinst_loss_state_bytes(int N, int h, int w);
/* byte offset, inside `state`, of two int32: {status (0 = fine; see bxi_boxinst_eval_f32), tile rows used}. */
size_t bxi_boxinst_loss_state_status_offset(int N, int h, int w);
/* byte offset, inside `state`, of one float: the warm-up factor bxi_boxinst_eval_f32 applied (given by value, or evaluated on the
 * device from iter_counter) -- what a second evaluation of the same iteration has to be given (a re-entrant backward). */
size_t bxi_boxinst_loss_state_warmup_offset(int N, int h, int w);

/* Forward (+ the bulk of the backward):
 * losses[0] = loss_prj, losses[1] = loss_pairwise (device, f32), complete when the call's work is done.
 * g_logits [N,1,h,w] (nullable: forward only): receives the UN-FINISHED gradient (zeros + the
 *   un-normalised pairwise gradient on the box tiles); bxi_boxinst_loss_backward_f32 finishes it in
 *   place.  Every element is written exactly once here.
 * affinity: the uint8 output of bxi_color_affinity_f32 for the same size/dilation/threshold.
 * warmup: min(_iter / pairwise_warmup, 1) (:1330-1331), evaluated on the host by the caller.
 * state (bxi_boxinst_loss_state_bytes, 256-B aligned): arg-max positions, unit projection gradients,
 *   box rectangles and the normaliser, for the backward; nullable when g_logits is NULL.
 * workspace (bxi_boxinst_loss_workspace_bytes, 256-B aligned): scratch, contents undefined after.
 * (The path for PRECOMPUTED affinity bits / explicit image masks; the evaluation from the network input is
 * bxi_boxinst_eval_f32 below.)  Only size == 3 (K = 8) is built into this fused path; other sizes return
 * BXI_ERR_UNSUPPORTED and the host composes section 1 + torch ops as the reference does.
 * N == 0 writes two zeros (documented deviation; the reference yields NaN). */
int bxi_boxinst_loss_fwd_bwd_f32(const bxi_instances* inst_host, const uint8_t* affinity,
                                 int size, int dilation, float warmup, float* losses, float* g_logits,
                                 void* state, void* workspace, size_t workspace_bytes, void* stream);

/* Backward: g_logits <- g_prj * d loss_prj/d logits + g_pw * d loss_pairwise/d logits, in place on
 * the buffer the forward filled (call exactly once per forward).  g_prj / g_pw are DEVICE scalars
 * (the upstream gradients autograd hands over), so there is no host sync. */
int bxi_boxinst_loss_backward_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw,
                                  int dilation, const void* state, float* g_logits, void* stream);

/* The evaluation proper: forward AND finished backward in one host call (csrc/fused_eval.hip), as ONE launch (eval1) at the
 * shipped configurations' shapes, otherwise as two (prep, pair):
 *   front half / launch 1: image side (stage A above: de-normalise, 4x4 pool, Lab) next to the logit streaming (row / column
 *             maxima, zero-fill of g_logits) and the per-instance table; nothing in it waits;
 *   back half / launch 2: projection term per instance (:117-143), pair weights and pairwise term per box tile (pairwise.cu:68-149
 *             semantics; colour affinity derived from Lab where it is needed, nothing of stage B is materialised), normaliser,
 *             both loss scalars; the gradient's two parts are ADDED to the zero-filled buffer (at most two additions per
 *             element: the result does not depend on their order).
 *   In the single-launch form the back half's workgroups follow the front half's in the grid and wait (bounded, loud) only for
 *   workgroups that precede them; what crosses workgroups carries a per-evaluation tag.
 * losses[0] = loss_prj, losses[1] = loss_pairwise (device, f32), complete when the call's work is done.
 * g_logits [N,1,h,w] (nullable: forward only) receives the FINISHED gradient
 *       up_prj * d loss_prj / d logits  +  up_pw * d loss_pairwise / d logits,
 *   every element written; up_prj / up_pw are DEVICE scalars read by the kernels (no host sync), NULL = 1 -- what
 *   `loss.backward()` seeds the two terms with (mmdet/core/optimizers hooks; a loss scale is known before the forward).
 *   Different factors later: bxi_boxinst_grad_rescale_f32.
 * warmup >= 0: min(_iter / pairwise_warmup, 1) (:1330-1331), evaluated on the host by the caller.
 * warmup <  0: -warmup is pairwise_warmup and the factor is evaluated ON THE DEVICE from inst_host->iter_counter (required then)
 *   as it stands after this call's `_iter += 1` (:1297): (float)min((double)(iter + 1.0f) / pairwise_warmup, 1.0) -- Python's
 *   arithmetic at :1330-1331.  No host mirror of the counter, and a captured graph ramps correctly.
 * flags: BXI_EVAL_* below, 0 = the library chooses.
 * state (bxi_boxinst_loss_state_bytes, 256-B aligned; required with g_logits): arg-max positions, unit projection
 *   gradients, box rectangles, normaliser, the factors applied, and a status word.  Status 0 = fine.  Non-zero = one of the
 *   bounded in-kernel waits ran out (tile waves wait for the predicate waves that precede them in the grid, the
 *   finisher for everybody; neither is expected to): BOTH LOSSES ARE NaN then (the reference surfaces launch failures through
 *   AT_CUDA_CHECK, pairwise.cu:173,200; here mmdet's CheckInvalidLossHook fires), and bxi_boxinst_grad_rescale_f32 poisons the
 *   gradient.
 * workspace (bxi_boxinst_eval_workspace_bytes, 256-B aligned): scratch incl. Lab.  ZERO IT ONCE after allocating it
 *   (bxi_boxinst_eval_workspace_init, or any memset ordered before the first evaluation) and never write to it again: it carries the
 *   tag counter ("epoch") that tells this evaluation's records from earlier ones', advanced on the device by each evaluation's last
 *   workgroup.  ONE workspace serves ONE canvas -- (B, Hc, Wc, stride) -- at ONE size: its layout is a function of those and of
 *   workspace_bytes alone (the per-instance regions are carved for the largest N the size admits), so evaluations with any N up to
 *   the N it was sized for share it, serialised on one stream; every word then only ever holds one kind of record, which is what makes
 *   the tags safe.  Another canvas, a larger N or overlapping evaluations (several streams) need a workspace of their own; to re-use
 *   the memory for another layout, or after an evaluation that reported a non-zero status, zero it again first.
 * batch_host->image_masks must be NULL (explicit masks: bxi_color_affinity_f32 + bxi_boxinst_loss_fwd_bwd_f32).
 * size == 3 and dilation <= 4 are built; others return BXI_ERR_UNSUPPORTED and the host composes section 1 + torch ops
 * as the reference does.  N == 0 writes two zeros (documented deviation; the reference yields NaN).
 * Strides other than 4 / unaligned canvases pool the image in launches of their own (same results, not the fast path).
 * The tag counter is 28 bits wide; the evaluation that draws its last value ends by returning the workspace to the all-zero state
 * itself (its last workgroup, after every other wave has arrived and drained its stores), so no caller ever has to count evaluations.
 * An ERROR return of an evaluation entry point after its first launch was enqueued (BXI_ERR_LAUNCH of a later launch) leaves a
 * stream-ordered memset of the whole workspace behind it: the workspace is usable again without further ado. */
size_t bxi_boxinst_eval_workspace_bytes(int B, int Hc, int Wc, int stride, int N);
/* byte offset, inside `workspace`, of the Lab image the evaluation leaves behind: [B, Hc/stride, Wc/stride] x float4 (L, a, b, 0)
 * -- what skimage.color.rgb2lab gives at condinst_head.py:1413-1416; exposed so that tests can compare it with scikit-image. */
size_t bxi_boxinst_eval_workspace_lab_offset(void);
/* hipMemsetAsync(workspace, 0, workspace_bytes) on `stream`: the one-time initialisation described above. */
int bxi_boxinst_eval_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
/* The image side alone, AHEAD of the evaluation -- replaces the `self.get_targets(gt_bboxes, gt_masks, imgs, img_metas)` call at the
 * top of CondInstMaskHead.loss (condinst_head.py:1298-1299; get_targets :1345-1393, get_bitmasks_from_boxes :1395-1448,
 * get_image_color_similarity :220-246).  Its inputs -- the network input and the GT boxes -- exist before the backbone runs
 * (mmdet/models/detectors/condinst.py:53 vs :73), so a caller can enqueue this on a side stream at the top of forward_train and take
 * the image -> Lab -> colour predicates -> pair-count chain off the loss's critical path.  Leaves in `workspace` (the evaluation's own,
 * same canvas, same size rules): Lab [B,h,w] float4, the predicate words [B,h,w], and PER GT BOX the count
 *   sum over the box's pixels p and the 8 neighbours k of [sim_k(p) >= color_thresh]   (:1324-1325 for one instance of that box),
 * so that the evaluation's normaliser sum W (:1327-1328) is a gather over gt_inds.  Two launches, no in-kernel wait (a kernel boundary
 * in between): makes progress next to anything.  Then bxi_boxinst_eval_f32(..., flags | BXI_EVAL_TARGETS_READY, ...) on the SAME
 * workspace, stream-ordered behind this call (same stream, or an event), with the same batch geometry / boxes / stride / window /
 * threshold: it launches only the logit stream, the leaders, the tiles and the finisher, reads no image (batch_host->imgs may be NULL)
 * and waits for nothing on the image side.  A digest of (canvas, stride, window, threshold, per image: shape, rows removed, box count) and
 * the cell rectangle of every GT box are kept with the targets and compared ON THE DEVICE by the evaluation (which maps its own boxes to
 * cells again): a mismatch, or targets overwritten by an evaluation without the flag (which computes its own), gives NaN losses and a
 * non-zero status -- never a plausible wrong number.  What the device cannot check is the IMAGE: the evaluation does not read it with the
 * targets ready, so that the targets were made from this batch's pixels is the caller's side of the contract.  Several evaluations may use one
 * set of targets (a re-entrant backward).  At most 1024 GT boxes per batch and color_thresh > 0, else BXI_ERR_UNSUPPORTED (call the
 * evaluation without the flag).  Results are bit-equal to the evaluation without the flag. */
int bxi_boxinst_targets_f32(const bxi_image_batch* batch_host, const float* const* boxes_per_img_host, const int* gt_count_host,
                            int stride, int size, int dilation, float color_thresh, void* workspace, size_t workspace_bytes, void* stream);

/* `flags` of the two evaluation entry points.  The forms give the same bits (tests run them against each other).
 * What the library runs by itself (flags == 0), at dilation <= 2 on a stride-4 aligned canvas:
 *   - ONE launch (4-row tiles, four workgroups per CU) while its stream workgroups -- instances x ceil(h / 32) -- fill at most half the GPU
 *     (up to 73 instances of 200 x 256 maps); while they fill at most a quarter (36 instances) they stay on as the launch's first tile workgroups;
 *   - beyond that TWO launches (table + logit stream + image pooling | predicates + leaders + tiles + finisher): 4-row tiles up to 95
 *     instances, 8-row tiles from 96 on.  Every in-kernel wait of this form is for a workgroup EARLIER in its grid;
 *   - with BXI_EVAL_TARGETS_READY the same two shapes without the image side (the one launch up to a quarter of the GPU: 36 instances), and
 *     from 96 instances on ONE launch with 8-row tiles (three workgroups per CU, nobody waits for a later workgroup).
 * Other dilations / canvases: two launches (+ the generic pooling launches).  ABI 7 removed the forms that lost their measurements
 * (BXI_EVAL_PRED_IN_PREP, the 8-row single launch with the image side in it) and folded BXI_EVAL_NO_STAY_ON into BXI_EVAL_SHARED_DEVICE. */
#define BXI_EVAL_SINGLE_LAUNCH   1u   /* the single-launch form wherever it is built, also where the library would not choose it (its stream
                                         workgroups fill more than half the GPU).  Not built -- two launches then, silently --: dilation > 2,
                                         threshold <= 0, generic pooling, and 8-row tiles without BXI_EVAL_TARGETS_READY                  */
#define BXI_EVAL_TWO_LAUNCHES    2u   /* always the two-launch form: it makes progress whatever else occupies the device; the host side
                                         switches to it after an evaluation that reported a non-zero status                              */
#define BXI_EVAL_TILE_ROWS_8     4u   /* 8-row tiles whatever the instance count                                                         */
#define BXI_EVAL_TILE_ROWS_4     8u   /* 4-row tiles whatever the instance count                                                         */
#define BXI_EVAL_SHARED_DEVICE  16u   /* other work (evaluations on other streams, collectives, other processes) may run on the device
                                         at the same time: nothing in the launch may hold execution slots while it waits for workgroups
                                         that come later in the grid (the single launch's stream workgroups do not stay on as tile
                                         workgroups; no 8-row single launch by default).  The library does not guess this               */
#define BXI_EVAL_TARGETS_READY  32u   /* the image side is in the workspace already: bxi_boxinst_targets_f32 above                       */
#define BXI_EVAL_WAITS_GIVE_UP  64u   /* TEST ONLY: every bounded in-kernel wait gives up at once, which makes the failure path
                                         observable (NaN losses, status word, poisoned gradient); zero the workspace afterwards          */
#define BXI_EVAL_ALL_FLAGS (1u | 2u | 4u | 8u | 16u | 32u | 64u)
int bxi_boxinst_eval_f32(const bxi_image_batch* batch_host, const bxi_instances* inst_host,
                         int size, int dilation, float color_thresh, float warmup,
                         const float* up_prj, const float* up_pw,
                         float* losses, float* g_logits, void* state,
                         void* workspace, size_t workspace_bytes, unsigned int flags, void* stream);

/* The same evaluation with the producer of the logits inside its first launch: replaces
 * CondInstMaskHead.forward (condinst_head.py:1139-1164) followed by CondInstMaskHead.loss (:1288-1343), the two calls
 * mmdet/models/detectors/condinst.py:71-74 makes back to back.  The first launch runs the dynamic mask head's tiles (arguments as
 * bxi_dynamic_mask_forward_f32; B = inst_host->B, N = inst_host->N) next to the image pooling; a tile leaves its logits, its
 * zero-filled gradient tile and its share of the row / column maxima, so nothing reads the logits back before the pair launch.
 * inst_host->logits is the buffer the logits are WRITTEN to ([N,1,2*Hs,2*Ws]; pair_kernel and the head's backward,
 * bxi_dynamic_mask_backward_f32 with g_logits, read them).  One launch, its kernel boundary and one 6.5 MB read fewer than
 * bxi_dynamic_mask_forward_f32 + bxi_boxinst_eval_f32 (20 us against 13.4 + 11.4 us at 2 x 800 x 1024 x 32).  Built for
 * factor == 2, C in {8, 16}, 16-byte aligned rows (w % 4 == 0), the 4x-pooled image path; anything else (and N == 0) returns
 * BXI_ERR_UNSUPPORTED: call the two entries instead (also with BXI_EVAL_TARGETS_READY: the head-fused launch computes the image side itself).
 * Workspace / state / upstream factors as bxi_boxinst_eval_f32. */
int bxi_boxinst_head_eval_f32(const bxi_image_batch* batch_host, const bxi_instances* inst_host,
                              const float* feat, int C, int Hs, int Ws, const float* params, const float* coors,
                              const int64_t* level_inds, const int64_t* img_inds, const float* sizes_of_interest, int n_levels,
                              int in_stride, int factor, int disable_rel_coors,
                              int size, int dilation, float color_thresh, float warmup,
                              const float* up_prj, const float* up_pw, float* losses, float* g_logits, void* state,
                              void* workspace, size_t workspace_bytes, unsigned int flags, void* stream);

/* g_logits finished by bxi_boxinst_eval_f32 for the factors recorded in `state`  ->  finished for (g_prj, g_pw)
 * (DEVICE scalars: the upstream gradients autograd hands over; no host sync).  The kernel returns at once when they
 * equal the recorded ones (the usual case).  At most one effective rescale per evaluation (the record is not updated);
 * a recorded up_pw of 0 cannot be rescaled. */
int bxi_boxinst_grad_rescale_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw, int dilation,
                                 const void* state, float* g_logits, void* stream);
/* The same from (N, h, w) alone -- all the kernel needs of the instances (the rectangles are in `state`): what an autograd node keeps
 * for its backward (torch.autograd.Function.backward of the reference's op, pairwise.py:17-26, likewise keeps tensors, not structs). */
int bxi_boxinst_grad_rescale_nhw_f32(int N, int h, int w, const float* g_prj, const float* g_pw, int dilation, const void* state,
                                     float* g_logits, void* stream);

/* ===========================================================================================
 * 4. The producer of mask_logits -- replaces CondInstMaskHead.forward (condinst_head.py:1139-1164):
 *    relative coordinates (:1142-1154), parse_dynamic_params (:1120-1137), the three per-instance
 *    grouped 1x1 convolutions + ReLU (:1156-1161) and aligned_bilinear (:146-167, :1163).
 * ===========================================================================================
 * feat   [B,C,H,W]  mask-branch features at `in_stride` (C = 8 or 16 built; others BXI_ERR_UNSUPPORTED: the host then
 *                   composes the layers as batched matrix products, boxinstseg_amd/mask_head.py:_composed_forward)
 * params [N,P]      per-instance dynamic parameters, P = (C+2)*8 + 64 + 8 + 8 + 8 + 1 (the
 *                   split_with_sizes order of the reference: w0, w1, w2, b0, b1, b2); C*8+... when
 *                   disable_rel_coors
 * coors  [N,2] f32 (x,y) ; level_inds [N] i64 ; img_inds [N] i64 ; sizes_of_interest [n_levels] f32
 * logits [N,1,H*factor,W*factor], factor = in_stride / out_stride.
 */
int bxi_dynamic_mask_forward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N,
                                 const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                 const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                 int disable_rel_coors, float* logits, void* stream);

/* Backward of the above: g_feat [B,C,H,W] and g_params [N,P] (both fully overwritten) from g_logits.
 * Deterministic (no atomics; the reference's feat[img_inds] backward is an atomic index_add).
 * workspace: bxi_dynamic_mask_backward_workspace_bytes, 256-B aligned. */
size_t bxi_dynamic_mask_backward_workspace_bytes(int B, int C, int H, int W, int N, int disable_rel_coors);
int bxi_dynamic_mask_backward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N,
                                  const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                  const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                  int disable_rel_coors, const float* g_logits, float* g_feat, float* g_params,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* The same head for EVERY shape the reference's constructor admits (condinst_head.py:1079-1089 leaves dynamic_convs, dynamic_channels
 * and in_channels free; the entries above are tuned for the shipped 3 layers x 8 channels on 8 / 16 feature channels):
 *   layers = dynamic_convs in 1..4, channels = dynamic_channels in 1..16, C + 2 (relative coordinates) <= 34, any factor.
 * params [N,P] as parse_dynamic_params (:1120-1137) splits them: all weights layer by layer (rows = output channels:
 *   [channels x (C+2)], [channels x channels] x (layers-2), [1 x channels]; layers == 1: [1 x (C+2)]), then all biases.
 * Other limits return BXI_ERR_UNSUPPORTED.  Backward: g_feat and g_params fully overwritten, no atomics, run-to-run identical. */
int bxi_dynamic_mask_generic_forward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N, int layers, int channels,
                                         const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                         const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                         int disable_rel_coors, float* logits, void* stream);
size_t bxi_dynamic_mask_generic_backward_workspace_bytes(int B, int C, int H, int W, int N, int layers, int channels, int disable_rel_coors);
int bxi_dynamic_mask_generic_backward_f32(const float* feat, int B, int C, int H, int W, const float* params, int N, int layers, int channels,
                                          const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                                          const float* sizes_of_interest, int n_levels, int in_stride, int factor,
                                          int disable_rel_coors, const float* g_logits, float* g_feat, float* g_params,
                                          void* workspace, size_t workspace_bytes, void* stream);

/* ===========================================================================================
 * 5. DiscoBox pseudo-label path -- mmdet/models/dense_heads/discobox_head.py:
 *    MeanField.__init__ (:591-613), MeanField.forward / simple_forward (:617-655),
 *    dice_loss (:542-550), mil_loss (:552-562).
 * ===========================================================================================*/

/* MeanField.__init__: the Gaussian-bilateral neighbourhood kernel of B images.
 * feat [B,C,H,W] (the resized, normalised image; C = 3 in the reference), ksize odd (3 or 5),
 * kernel [B,ksize*ksize,H,W]:
 *   kernel[b,k,p] = alpha0 * exp( sum_c -(F[c,q]-F[c,p])^2 / (2 theta0^2) - |delta_k|^2 / (2 theta1^2) ),
 *   F = feat + 10 inside the map, 0 outside (nn.Unfold's zero padding of the shifted map, :597-598). */
int bxi_meanfield_kernel_f32(const float* feat, int B, int C, int H, int W, int ksize, float alpha0, float theta0,
                             float theta1, float* kernel, void* stream);

/* MeanField.forward for N instances at once: `iters` mean-field updates under no_grad (one launch per update over
 * all instances, plus one before and one after; nothing is synchronised).
 * kernel [B,ksize^2,H,W] (above); x [N,H,W] f32; targets [N,H,W], f32 (targets_u8 = 0) or u8 (1), values 0/1;
 * img_inds [N] i64 (kernel plane of every instance) or NULL (all image 0: one MeanField object);
 * inter_img_mask [N,2,H,W] f32 or NULL (added times gamma, :647-648);
 * ret [N,H,W] f32 in {0,1} (the reference's [N,1,H,W]); valid [N] f32 (5 % <= foreground <= 95 %, :634-636).
 * workspace: bxi_meanfield_workspace_bytes(N,H,W) (one bit per pixel, three planes), 8-byte aligned. */
size_t bxi_meanfield_workspace_bytes(int N, int H, int W);
int bxi_meanfield_forward_f32(const float* kernel, int B, int H, int W, int ksize, const float* x, const void* targets,
                              int targets_u8, const int64_t* img_inds, int N, int iters, float base,
                              const float* inter_img_mask, float gamma, float* ret, float* valid, void* workspace,
                              size_t workspace_bytes, void* stream);

/* dice_loss (:542-550) of N rows of L elements: loss[n] = 1 - 2 a / (b + c), a = sum i t, b = sum i^2 + 0.001,
 * c = sum t^2 + 0.001.  target f32 (target_u8 = 0) or u8 (1).  sums [N,2] f32 receives (a, b + c) for the backward. */
int bxi_dice_loss_forward_f32(const float* input, const void* target, int target_u8, int N, int64_t L, float* loss,
                              float* sums, void* stream);
/* g_input[n,:] = g_loss[n] * ( -2 t / (b+c) + 4 a i / (b+c)^2 ) */
int bxi_dice_loss_backward_f32(const float* input, const void* target, int target_u8, int N, int64_t L,
                               const float* sums, const float* g_loss, float* g_input, void* stream);

/* mil_loss(dice_loss, input, _, target) (:552-562): row/column maxima of input and target [N,H,W], one dice term
 * per axis.  loss [N].  state (bxi_mil_loss_state_bytes, 16-byte aligned) keeps the arg-max positions and the unit
 * gradients of the H + W maxima, followed by the forward's scratch (per-band column maxima: the forward is two launches,
 * row bands over the whole GPU + one workgroup per instance); the backward writes g_input [N,H,W] densely (zeros
 * elsewhere), no atomics. */
size_t bxi_mil_loss_state_bytes(int N, int H, int W);
int bxi_mil_loss_forward_f32(const float* input, const void* target, int target_u8, int N, int H, int W, float* loss,
                             void* state, void* stream);
int bxi_mil_loss_backward_f32(int N, int H, int W, const void* state, const float* g_loss, float* g_input, void* stream);

/* ===========================================================================================
 * 6. Box2Mask / BoxLevelSet loss pieces:
 *    BoxProjectionLoss (mmdet/models/losses/box_projection_loss.py:5-43),
 *    LevelsetLoss / region_levelset (mmdet/models/losses/levelset_loss.py:7-45),
 *    LocalConsistencyModule (levelset_loss.py:63-126; LCM() :53-60 composes it with elementwise ops).
 * ===========================================================================================*/

/* BoxProjectionLoss.forward: loss[n] = loss_weight * (dice(max over rows) + dice(max over columns)), dice with
 * eps = 1e-5 on the union (:34-43).  mask_scores, box_bitmask [N,H,W] f32 (the reference's [N,1,H,W]).
 * state: bxi_mil_loss_state_bytes(N,H,W); backward: bxi_mil_loss_backward_f32 (loss_weight is in the state). */
int bxi_projection_loss_forward_f32(const float* mask_scores, const float* box_bitmask, int N, int H, int W, float loss_weight,
                                    float* loss, void* state, void* stream);

/* LevelsetLoss.forward (:13-18) = loss_weight * region_levelset(mask_score, target) / pixel_num.
 * mask_score [N,2,H,W] (foreground, background scores), target [N,C,H,W] (any C; channels beyond 8 cost one more launch of the partial sums per 8), pixel_num [N]; loss [N].
 * state (bxi_levelset_state_bytes) keeps the region sums / means for the backward. */
size_t bxi_levelset_state_bytes(int N, int C);
int bxi_levelset_loss_forward_f32(const float* mask_score, const float* target, const float* pixel_num, int N, int C, int H,
                                  int W, float loss_weight, float* loss, void* state, void* stream);
/* g_mask_score [N,2,H,W] and, if not NULL, g_target [N,C,H,W] (the deep-feature targets carry gradient, box2mask_head.py:319-328) */
int bxi_levelset_loss_backward_f32(const float* mask_score, const float* target, const float* pixel_num, int N, int C, int H,
                                   int W, float loss_weight, const void* state, const float* g_loss, float* g_mask_score,
                                   float* g_target, void* stream);

/* LocalConsistencyModule: affinity of the 8 dilated neighbours (replicate padding) from the image (:108-120),
 * imgs [N,C,h,w] -> aff [N,8,h,w]; */
int bxi_lcm_affinity_f32(const float* imgs, int N, int C, int h, int w, int dilation, float alpha, float* aff, void* stream);
/* `iters` refinement steps phi <- sum_k aff_k * phi(neighbour_k) (:122-126): phi [N,h,w] -> out [N,h,w].
 * transpose != 0 applies the adjoint operator instead (the backward: g_phi from g_out).
 * workspace: bxi_lcm_workspace_bytes(N,h,w) (one ping-pong plane; unused when a map fits LDS), 16-byte aligned. */
size_t bxi_lcm_workspace_bytes(int N, int h, int w);
int bxi_lcm_refine_f32(const float* aff, const float* phi, int N, int h, int w, int dilation, int iters, int transpose,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ===========================================================================================
 * 7. The `tree_filter` extension -- mmdet/ops/tree_filter: mst_forward (src/mst/mst.cu:93-118 +
 *    boruvka.cpp), bfs_forward (src/bfs/bfs.cu:92-135), refine_forward / refine_backward_feature /
 *    refine_backward_weight (src/refine/refine.cu:186-370), as bound by src/tree_filter.cpp.
 *    V <= 10200 vertices: everything a traversal touches is LDS-resident (Box2Mask's 96x96 maps).  Larger graphs (BoxLevelSet
 *    filters its 200x304 mask features, box_solov2_head.py:354-358) run the same algorithms with their arrays in the
 *    caller's workspace (the *_workspace_bytes functions return 0 extra when LDS suffices).
 * ===========================================================================================*/

/* Minimum spanning trees of B graphs: edge_index [B,E,2] i32, edge_weight [B,E] f32 >= 0 -> edge_out [B,V-1,2] i32.
 * The tree is the unique MST under the order (weight, edge index) -- the edges the reference's Boruvka selects --
 * listed in ascending edge order (the reference lists them in Boruvka's emission order).  On the GPU, no host copy.
 * workspace: bxi_mst_workspace_bytes(B, E, V), 16-byte aligned; its first B ints receive the number of tree edges per graph
 * (V-1 for a connected graph). */
size_t bxi_mst_workspace_bytes(int B, int E, int V);
int bxi_mst_forward_i32(const int* edge_index, const float* edge_weight, int B, int E, int V, int* edge_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Breadth-first order from vertex 0: tree_edges [B,V-1,2] -> sorted_index [B,V] (vertex at a position),
 * sorted_parent [B,V] (position of the parent), sorted_child [B,V,max_adj] (positions, 0 = none; zero-filled here).
 * Deterministic, children of a node contiguous.  levels [B,V+2] i32 = { D, off_0 = 0, ..., off_D = V }: the level
 * structure the refine kernels walk (an extra output; the reference keeps no such thing). */
size_t bxi_bfs_workspace_bytes(int B, int V);          /* 0 when V fits LDS; then workspace may be NULL */
int bxi_bfs_forward_i32(const int* tree_edges, int B, int V, int max_adj, int* sorted_index, int* sorted_parent, int* sorted_child,
                        int* levels, void* workspace, size_t workspace_bytes, void* stream);

/* refine_forward: feature_in [B,C,V] (vertex order), edge_weight [B,V] (sorted order, [0] unused) ->
 * feature_out, feature_aggr [B,C,V] (vertex order), feature_aggr_up [B,C,V] (sorted), weight_sum [B,V] (vertex),
 * weight_sum_up [B,V] (sorted) -- the five tensors of refine.cu:229-232.  sorted_* / levels from bxi_bfs_forward_i32
 * (a child order that is not contiguous yields NaN). */
size_t bxi_tree_refine_workspace_bytes(int B, int C, int V);   /* 0 when V fits LDS; then workspace may be NULL; 16-byte aligned */
int bxi_tree_refine_forward_f32(const float* feature_in, const float* edge_weight, const int* sorted_index, const int* sorted_child,
                                const int* levels, int B, int C, int V, int max_adj, float* feature_out, float* feature_aggr,
                                float* feature_aggr_up, float* weight_sum, float* weight_sum_up, void* workspace, size_t workspace_bytes,
                                void* stream);
/* refine_backward_feature / refine_backward_weight.  The weight gradient's first traversal IS the feature gradient, so
 * bxi_tree_refine_backward_weight_f32 also returns it when `grad_feature` [B,C,V] is non-null (one launch for both,
 * two concurrent traversals); workspace: bxi_tree_refine_backward_weight_workspace_bytes(B, C, V) (includes the refine
 * workspace of large graphs), 16-byte aligned. */
int bxi_tree_refine_backward_feature_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                         const int* sorted_child, const int* levels, const float* weight_sum, int B, int C, int V,
                                         int max_adj, float* grad_feature, void* workspace, size_t workspace_bytes, void* stream);
size_t bxi_tree_refine_backward_weight_workspace_bytes(int B, int C, int V);
int bxi_tree_refine_backward_weight_f32(const float* grad_out, const float* edge_weight, const int* sorted_index,
                                        const int* sorted_parent, const int* sorted_child, const int* levels, const float* feature_out,
                                        const float* feature_aggr, const float* feature_aggr_up, const float* weight_sum,
                                        const float* weight_sum_up, int B, int C, int V, int max_adj, float* grad_weight,
                                        float* grad_feature, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BOXINST_HIP_H */
