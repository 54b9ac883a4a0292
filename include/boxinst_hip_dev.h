/*
 * boxinst_hip_dev.h -- developer / test hooks of libboxinst_hip.so.  NOT part of the production ABI (include/boxinst_hip.h is
 * state-free: everything that varies is an argument); these two are process-wide switches for the benchmark harness and the test
 * suite, kept in atomics, and nothing in boxinstseg_amd/ calls them.
 */
#ifndef BOXINST_HIP_DEV_H
#define BOXINST_HIP_DEV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py): `hook(kernel_name, phase, stream, user)` is called on the host right before (phase 0) and right
 * after (phase 1) each kernel launch the library enqueues, so the caller can bracket individual kernels with hipEvents on the
 * launching stream -- the reference has no counterpart (its two launches, pairwise.cu:165-173,192-200, are timed from outside).
 * NULL removes the hook. */
typedef void (*bxi_launch_hook)(const char* kernel_name, int phase, void* stream, void* user);
void bxi_dev_set_launch_hook(bxi_launch_hook hook, void* user);

/* Test switch: bxi_bfs_forward_i32 / bxi_tree_refine_* walk large trees level by level (the reference's own order, bfs.cu:46-98,
 * refine.cu:70-199) instead of ranking their Euler tour / doubling; the two give the same bits and tests compare them. */
void bxi_dev_set_tree_level_walk(int on);

/* Measuring stick (bench.py `roofline.sol_us`): ONE launch with the single-launch evaluation's grid -- the same numbers of 256-thread
 * workgroups per role, four per CU -- that performs the evaluation's loads and stores (imgs [B,3,Hc,Wc] read, logits [N,1,h,w] read,
 * g_logits zero-filled and added to on the tile hulls, the Lab / predicate intermediates written and re-read in `workspace`:
 * >= 20 * B * h * w + 256 bytes) with no arithmetic and NO dependency between workgroups.  Requires Hc == 4 h, Wc == 4 w, w % 4 == 0, h >= 12, w >= 68.
 * imgs == NULL: without the image roles -- the bytes of an evaluation whose targets are ready (BXI_EVAL_TARGETS_READY).
 * Leaves garbage in g_logits / workspace.  csrc/sol_eval.hip. */
int bxi_dev_sol_eval_f32(const float* imgs, int B, int Hc, int Wc, const float* logits, int N, int h, int w, float* g_logits,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same for the op-level kernels (tools/bench_pairwise_op.py, bench.py `extras.pairwise_op.*_sol_us`): the bytes of
 * pairwise_nlog's forward (mode 0: logits [N,1,H,W] read, planes [N,8,H,W] written) or backward (mode 1: logits and planes read, out
 * [N,1,H,W] written; pairwise.cu:68-202 at size 3) moved by a copy -- 16-byte accesses, nothing computed.  N*H*W % 4 == 0. */
int bxi_dev_sol_pairwise_f32(const float* logits, float* planes, float* out, int N, int H, int W, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif
