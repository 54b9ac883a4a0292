// abi.hip -- library-level entry points of libboxinst_hip.so (see include/boxinst_hip.h).
#include "common.hpp"
#include "dynamic_head_device.hpp"
#include "../../include/boxinst_hip_dev.h"
#include <atomic>
#include <cstdlib>

namespace bxi {

static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }
std::atomic<bxi_launch_hook> g_hook{nullptr};         // developer hook (include/boxinst_hip_dev.h); not part of the production ABI
std::atomic<void*> g_hook_user{nullptr};

size_t loss_ws_bytes(int N, int h, int w);
size_t eval_ws_bytes(int B, int N, int h, int w);
bool fused_eval_supported(int dil);
void dev_set_tree_level_walk(int on);
int eval_ws_init(void* workspace, size_t bytes, void* stream);
int launch_fused_eval(const bxi_image_batch* batch, float color_thresh, const bxi_instances* in, int dil, float warmup, const float* up_prj,
                      const float* up_pw, float* losses, float* g_logits, void* state, void* workspace, size_t workspace_bytes, unsigned flags,
                      void* stream, const DynArgs* head = nullptr, int head_C = 0);
int launch_targets(const bxi_image_batch* batch, const float* const* boxes_per_img_host, const int* gt_count_host, int stride, int dil, float color_thresh,
                   void* workspace, size_t workspace_bytes, void* stream);
int launch_rescale(const bxi_instances* in, const float* g_prj, const float* g_pw, int dil, const void* state, float* g_logits, void* stream);
int launch_rescale_nhw(int N, int h, int w, const float* g_prj, const float* g_pw, int dil, const void* state, float* g_logits, void* stream);

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace bxi

extern "C" {

int bxi_abi_version(void) { return BXI_ABI_VERSION; }

const char* bxi_status_string(int status) {
    switch (status) {
        case BXI_OK: return "ok";
        case BXI_ERR_NULL_POINTER: return "a required pointer is NULL";
        case BXI_ERR_BAD_SHAPE: return "bad shape (negative, inconsistent or too large dimension)";
        case BXI_ERR_BAD_ARGUMENT: return "bad argument (pairwise_size must be odd, dilation/stride >= 1)";
        case BXI_ERR_UNSUPPORTED: return "configuration outside the fused fast path of this build";
        case BXI_ERR_WORKSPACE: return "workspace missing, too small or not 256-byte aligned";
        case BXI_ERR_LAUNCH: return "HIP kernel launch failed (see bxi_last_hip_error)";
        case BXI_ERR_NO_DEVICE: return "no gfx950 HIP device";
        default: return "unknown status";
    }
}

int bxi_last_hip_error(void) { return bxi::g_last_hip_error; }

void bxi_dev_set_launch_hook(bxi_launch_hook hook, void* user) {
    bxi::g_hook.store(nullptr, std::memory_order_release);          // never a new hook with the old user pointer
    bxi::g_hook_user.store(user, std::memory_order_release);
    bxi::g_hook.store(hook, std::memory_order_release);
}
void bxi_dev_set_tree_level_walk(int on) { bxi::dev_set_tree_level_walk(on); }

int bxi_check_device(int ordinal) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || ordinal < 0 || ordinal >= count) {
        (void)hipGetLastError();
        return BXI_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return BXI_ERR_NO_DEVICE;
    const char* arch = prop.gcnArchName;
    // "gfx950:sramecc+:xnack-"
    if (arch[0] == 'g' && arch[1] == 'f' && arch[2] == 'x' && arch[3] == '9' && arch[4] == '5' && arch[5] == '0')
        return BXI_OK;
    return BXI_ERR_NO_DEVICE;
}

size_t bxi_boxinst_eval_workspace_bytes(int B, int Hc, int Wc, int stride, int N) {
    if (B < 0 || Hc <= 0 || Wc <= 0 || stride < 1 || N < 0) return 0;
    const int h = Hc / stride, w = Wc / stride;
    if (h <= 0 || w <= 0) return 0;
    return bxi::eval_ws_bytes(B, N, h, w);
}

size_t bxi_boxinst_eval_workspace_lab_offset(void) { return 256; }      // behind the epoch word (fused_eval.hip: carve)

int bxi_boxinst_eval_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    return bxi::eval_ws_init(workspace, workspace_bytes, stream);
}

int bxi_boxinst_targets_f32(const bxi_image_batch* batch_host, const float* const* boxes_per_img_host, const int* gt_count_host, int stride, int size,
                            int dilation, float color_thresh, void* workspace, size_t workspace_bytes, void* stream) {
    if (!batch_host || !boxes_per_img_host || !gt_count_host) return BXI_ERR_NULL_POINTER;
    if (size < 1 || (size & 1) == 0 || dilation < 1 || stride < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || !bxi::fused_eval_supported(dilation)) return BXI_ERR_UNSUPPORTED;
    const size_t need = bxi_boxinst_eval_workspace_bytes(batch_host->B, batch_host->Hc, batch_host->Wc, stride, 0);
    if (!workspace || need == 0 || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    return bxi::launch_targets(batch_host, boxes_per_img_host, gt_count_host, stride, dilation, color_thresh, workspace, workspace_bytes, stream);
}

// the combinations of `flags` that contradict each other
static bool bad_eval_flags(unsigned int flags) {
    if (flags & ~(unsigned)BXI_EVAL_ALL_FLAGS) return true;
    if ((flags & BXI_EVAL_SINGLE_LAUNCH) && (flags & BXI_EVAL_TWO_LAUNCHES)) return true;
    if ((flags & BXI_EVAL_TILE_ROWS_8) && (flags & BXI_EVAL_TILE_ROWS_4)) return true;
    return false;
}

int bxi_boxinst_eval_f32(const bxi_image_batch* batch_host, const bxi_instances* inst_host, int size, int dilation,
                         float color_thresh, float warmup, const float* up_prj, const float* up_pw, float* losses,
                         float* g_logits, void* state, void* workspace, size_t workspace_bytes, unsigned int flags, void* stream) {
    if (!batch_host || !inst_host) return BXI_ERR_NULL_POINTER;
    if (size < 1 || (size & 1) == 0 || dilation < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || !bxi::fused_eval_supported(dilation)) return BXI_ERR_UNSUPPORTED;
    const int stride = inst_host->stride;
    if (stride < 1 || batch_host->Hc != inst_host->Hc || batch_host->Wc != inst_host->Wc ||
        batch_host->B != inst_host->B)
        return BXI_ERR_BAD_SHAPE;
    const size_t need = bxi_boxinst_eval_workspace_bytes(batch_host->B, batch_host->Hc, batch_host->Wc, stride,
                                                         inst_host->N);
    if (!workspace || need == 0 || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255))
        return BXI_ERR_WORKSPACE;
    if (bad_eval_flags(flags)) return BXI_ERR_BAD_ARGUMENT;
    return bxi::launch_fused_eval(batch_host, color_thresh, inst_host, dilation, warmup, up_prj, up_pw, losses, g_logits, state, workspace,
                                  workspace_bytes, flags, stream);
}

int bxi_boxinst_head_eval_f32(const bxi_image_batch* batch_host, const bxi_instances* inst_host, const float* feat, int C, int Hs,
                              int Ws, const float* params, const float* coors, const int64_t* level_inds, const int64_t* img_inds,
                              const float* sizes_of_interest, int n_levels, int in_stride, int factor, int disable_rel_coors,
                              int size, int dilation, float color_thresh, float warmup, const float* up_prj, const float* up_pw,
                              float* losses, float* g_logits, void* state, void* workspace, size_t workspace_bytes, unsigned int flags,
                              void* stream) {
    if (!batch_host || !inst_host) return BXI_ERR_NULL_POINTER;
    if (bad_eval_flags(flags)) return BXI_ERR_BAD_ARGUMENT;      // (the forms the head-fused first launch is not built in are not taken: two launches)
    if (flags & BXI_EVAL_TARGETS_READY) return BXI_ERR_UNSUPPORTED;   // (targets ahead: bxi_dynamic_mask_forward_f32 + bxi_boxinst_eval_f32)
    if (size < 1 || (size & 1) == 0 || dilation < 1) return BXI_ERR_BAD_ARGUMENT;
    if (size != 3 || !bxi::fused_eval_supported(dilation)) return BXI_ERR_UNSUPPORTED;
    const int stride = inst_host->stride;
    if (stride < 1 || batch_host->Hc != inst_host->Hc || batch_host->Wc != inst_host->Wc || batch_host->B != inst_host->B)
        return BXI_ERR_BAD_SHAPE;
    if (inst_host->N == 0) return BXI_ERR_UNSUPPORTED;      // nothing to fuse: bxi_boxinst_eval_f32 writes the two zeros
    if (!inst_host->logits) return BXI_ERR_NULL_POINTER;
    bxi::DynArgs da;
    int rc = bxi::fill_dyn(feat, inst_host->B, C, Hs, Ws, params, inst_host->N, coors, level_inds, img_inds, sizes_of_interest, n_levels,
                           in_stride, factor, disable_rel_coors, da);
    if (rc != BXI_OK) return rc;
    const size_t need = bxi_boxinst_eval_workspace_bytes(batch_host->B, batch_host->Hc, batch_host->Wc, stride, inst_host->N);
    if (!workspace || need == 0 || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    return bxi::launch_fused_eval(batch_host, color_thresh, inst_host, dilation, warmup, up_prj, up_pw, losses, g_logits, state, workspace,
                                  workspace_bytes, flags, stream, &da, C);
}

int bxi_boxinst_grad_rescale_f32(const bxi_instances* inst_host, const float* g_prj, const float* g_pw, int dilation,
                                 const void* state, float* g_logits, void* stream) {
    return bxi::launch_rescale(inst_host, g_prj, g_pw, dilation, state, g_logits, stream);
}
int bxi_boxinst_grad_rescale_nhw_f32(int N, int h, int w, const float* g_prj, const float* g_pw, int dilation, const void* state,
                                     float* g_logits, void* stream) {
    return bxi::launch_rescale_nhw(N, h, w, g_prj, g_pw, dilation, state, g_logits, stream);
}

}  // extern "C"
