// fused_eval.hip -- the BoxInst loss evaluation (forward AND finished backward) on gfx950: ONE launch (eval1_kernel, the roles below in
// one grid) where the shipped shapes allow it, otherwise TWO (prep_kernel, pair_kernel).
//
// Replaces (reference, LiWentomng/BoxInstSeg) CondInstMaskHead.loss with boxinst_enabled,
// condinst_head.py:1288-1343, together with everything it calls:
//   get_targets / get_original_image / get_bitmasks_from_boxes   :170-186, :1345-1448   (image side)
//   get_image_color_similarity + unfold_wo_center                :190-246
//   compute_project_term + dice_coefficient                      :117-143
//   pairwise_nlog (CUDA op, pairwise.cu:68-149) + weights / normalise / warm-up   :1315-1332
// and what autograd does behind them, with the upstream factors folded in.
//
//   launch 1  prep_kernel   256-thread workgroups, three roles, nothing waits                              HBM stream
//     table waves   per-instance table (tile prefix, box cells, image, valid-cell limits): 16 bytes per instance; zeroes the
//                   words the next launch polls
//     stream blocks 4 waves x 8 rows of one instance map: zero-fill of g_logits (written through), row maxima, column maxima
//                   of the block's 32 rows -> partials for the leaders of the next launch
//     pool blocks   the 4 input rows of 64 pooled pixels -> de-normalise, truncate, 4x4 mean, Lab (fp64) -> ONE 16-byte
//                   store per pooled pixel
//   launch 2  pair_kernel   [predicate blocks][reducer][leaders][tile blocks][finisher]
//     leaders       one block per instance: partial maxima -> maxima -> sigmoid -> dice -> unit projection gradients, ADDED
//                   (float atomic) at the arg-max positions of the zero-filled gradient.  Nobody waits for a leader but the finisher.
//     predicate waves  one wave64 per pooled row segment (64 pixels) of an image: the four colour predicates per pixel (one byte)
//                   -- each unordered pair ONCE PER IMAGE, not once per instance and tile -- and the segment's share of the pair
//                   weights' sum (a function of the image and the boxes only, :1324-1328) -> one packed integer atomic per workgroup;
//                   the reducer (one wave, right behind them in the grid) adds the 64 count words up and publishes ONE word (1 << 63 | sum W)
//     tile waves    one wave64 per box tile (no LDS, no barrier): logits tile + halo in registers, every unordered pair
//                   evaluated once; g_pw warm/max(sum W,1) d pw is ADDED (float atomic) to the gradient -- an element receives
//                   at most two additions onto 0 (its tile's and its leader's), so the sum does not depend on their order;
//                   the tile's share of sum W pw goes to an integer accumulator by an atomic without return.  Its two waits:
//                   its own predicate bytes (bit 7 = evaluated) before the pair loop, the published sum W (the global
//                   normaliser) after it; both are produced by workgroups that precede it in the grid and never wait
//                   for a tile.
//     finisher      the last workgroup: polls the accumulators (every tile wave arrives exactly once, with or without tiles), writes the two
//                   loss values and, as its last act, advances the workspace's epoch.
// What one workgroup hands to another inside a launch carries the evaluation's TAG = epoch + 1; the epoch is word 0 of the workspace,
// read on the device by every kernel of an evaluation (with_tag) -- nothing about it is a kernel argument, so a captured launch replayed
// from a hipGraph is as correct as an eager one; the warm-up factor likewise (resolve_warmup).  The workspace is zeroed once and keeps ONE
// layout (a function of the canvas and of its size), so a tag field only ever holds tags.  The launch's form is the caller's `flags`
// (include/boxinst_hip.h: BXI_EVAL_*): the library keeps no process-wide state and guesses nothing about what else runs on the device.
// Every wait is bounded and running out of it is loud: NaN losses, a status word, a poisoned gradient (rescale_kernel).
// Table entries instead of a work list: a tile wave finds its tile from 16 bytes per instance that every wave reads (the same
// few cache lines), not from a record of its own behind a list length (two dependent misses right after the kernel boundary).
// Data layout in HBM: everything NCHW / row-major as the reference hands it over; intermediates: Lab [B,h,w] float4 (1.6 MB at
// 2x800x1024), column / row partial maxima, 16-byte table entries.
#include "loss_common.hpp"
#include "dynamic_head_device.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <ctime>

namespace bxi {

constexpr int kWaves = 4;                       // waves per workgroup in both launches
constexpr int kSRows = 8;                       // rows per stream wave
constexpr int kSBlk = kWaves * kSRows;          // rows per stream workgroup
constexpr int kChunkC = 256;                    // columns per pass of a stream wave: 64 lanes x float4
constexpr int kMaxDilFused = 4;
#ifndef BXI_SPIN_LIMIT
#define BXI_SPIN_LIMIT 4000000
#endif
constexpr int kSpinLimit = BXI_SPIN_LIMIT;             // bounded waits (0.3 - 1 us per poll: seconds): a bound, not a schedule -- a transient stall (another
                                                // process time-slicing the GPU, a long kernel on another stream while stream workgroups stay on) must
                                                // not turn an iteration's losses into NaN; running out is loud (NaN losses, status word) and the host
                                                // side then takes the two-launch form, whose every wait is for an EARLIER workgroup
// bits of the `flags` argument of bxi_boxinst_eval_f32 (include/boxinst_hip.h: BXI_EVAL_*); per call, no process-wide state
constexpr unsigned kFlagSingle = BXI_EVAL_SINGLE_LAUNCH, kFlagTwo = BXI_EVAL_TWO_LAUNCHES, kFlagRows8 = BXI_EVAL_TILE_ROWS_8, kFlagRows4 = BXI_EVAL_TILE_ROWS_4,
                   kFlagShared = BXI_EVAL_SHARED_DEVICE, kFlagTargetsReady = BXI_EVAL_TARGETS_READY, kFlagGiveUp = BXI_EVAL_WAITS_GIVE_UP;
constexpr int kBoxCap = 1024;                   // GT boxes per batch bxi_boxinst_targets_f32 keeps pair counts for
constexpr int kBoxSplit = 8;                    // count words per box (each in its own 128 bytes): arrivals on one word are performed one after the other
constexpr int kNtStreamFromMB = 20;             // logit maps of this many MB and more are streamed past the L2 (non-temporal loads)
constexpr int kLongFrom = 96;                   // single launch, long form (8-row tiles) from this many instances on
constexpr unsigned int kMaxTag = 0x0fffffffu;   // tags are 28 bits (a predicate word is tag << 4 | bits)
constexpr int kAcc2Split = 8, kAcc2Stride = 16; // tile arrivals: eight words per instance, each in its own 128 bytes
constexpr int kAcc1Words = 64;                  // count-wave arrivals + sum W: 64 words, each in its own 128 bytes
constexpr int kMaxInst = 65536;
#ifndef BXI_ONE_OCC
#define BXI_ONE_OCC 4
#endif
#ifndef BXI_LONG_OCC
#define BXI_LONG_OCC 3      // workgroups per CU of the 8-row single-launch forms (138 VGPRs; a developer build may force 4: profiles/NOTES.md R6-3)
#endif
// developer builds (-DBXI_WAITLOG): the longest wait of every bounded in-grid wait, by site, in polls -- which wait a slow launch sat in
#ifdef BXI_WAITLOG
static __device__ unsigned int g_waitlog[16];
#define BXI_WL(site, spins) do { if ((spins) > 1000 && (threadIdx.x & 63) == 0) atomicMax(&g_waitlog[site], (unsigned int)(spins)); } while (0)
#else
#define BXI_WL(site, spins) do {} while (0)
#endif
constexpr int kOneOcc = BXI_ONE_OCC;           // workgroups per CU of the single-launch form (<= 128 VGPRs: the tile role's budget)
constexpr unsigned kFaultCounts = 1u, kFaultFinisher = 2u;
// A wave whose bounded wait ran out says so on the evaluation's fault word (zeroed by the first table wave before the entries every waiter checks),
// with a returning atomic it waits for BEFORE its arrival: the round in which the finisher sees the last arrival reads the fault word too.  The sum W
// word and a leader's dice word carry their own fault bit (one writer each); the arrival words of tile waves and predicate workgroups do not any
// more -- a flag ADDED to an arrival carries into the arrival count from the second (predicate) / fourth (tile) fault on one word on, and the
// finisher then waits kSpinLimit polls for a count that cannot come (4.5 s per evaluation with foreign targets, where every tile wave is "bad").
constexpr unsigned long long kCountFault = 1ull << 39, kSumwFault = 1ull << 62, kDiceFault = 1ull << 33;      // (bits 50 / 51 of an arrival word: reserved, checked by the finisher, set by nobody since R6-3)

#ifdef BXI_TRACE
#ifdef BXI_TRACE_LIGHT      // only the first and the last stamp of a wave: two stores per wave instead of eight (the full trace lengthens the launch by half)
#define BXI_TW(kid, idx, ph)                                                                                  \
    do {                                                                                                      \
        if (((ph) == 0 || (ph) == 7 || (kid) >= 2) && (threadIdx.x & 63) == 0 && g_trace && (idx) >= 0 && (idx) < ::bxi::kTraceBlocks) \
            g_trace[((size_t)(kid) * ::bxi::kTraceBlocks + (idx)) * ::bxi::kTracePhases + (ph)] = wall_clock64(); \
    } while (0)
#else
#define BXI_TW(kid, idx, ph)                                                                                  \
    do {                                                                                                      \
        if ((threadIdx.x & 63) == 0 && g_trace && (idx) >= 0 && (idx) < ::bxi::kTraceBlocks)                  \
            g_trace[((size_t)(kid) * ::bxi::kTraceBlocks + (idx)) * ::bxi::kTracePhases + (ph)] = wall_clock64(); \
    } while (0)
#endif
#else
#define BXI_TW(kid, idx, ph) do {} while (0)
#endif

// developer build (-DBXI_ABLATE, tools/ablate.py): switch parts of the evaluation OFF (results become wrong) to see, without the
// distortion of a trace, what the step time is sensitive to.  Never in the shipped library: BXI_AB(x) is the constant false there.
#ifdef BXI_ABLATE
static __device__ int g_ablate = 0;
#define BXI_AB(bit) ((g_ablate & (bit)) != 0)
#else
#define BXI_AB(bit) false
#endif

// back-off between the polls of the bounded in-grid waits, in units of 64 clocks (tuning knobs; A/B of 1 .. 32 on one box moved the step by +-0.15 us at most: the waits are not what the launch ends on)
#ifndef BXI_SLEEP_TAB
#define BXI_SLEEP_TAB 16
#endif
#ifndef BXI_SLEEP_PRED
#define BXI_SLEEP_PRED 16
#endif
#ifndef BXI_SLEEP_WORDS
#define BXI_SLEEP_WORDS 8
#endif
#ifndef BXI_SLEEP_SUMW
#define BXI_SLEEP_SUMW 8
#endif
#ifndef BXI_SLEEP_LEAD
#define BXI_SLEEP_LEAD 4
#endif
#ifndef BXI_SLEEP_FIN
#define BXI_SLEEP_FIN 2
#endif

#define BXI_RLX __ATOMIC_RELAXED
#define BXI_AGENT __HIP_MEMORY_SCOPE_AGENT

// float add at the L2 without return (global_atomic_add_f32): the gradient is zero-filled by launch 1 and every element
// receives at most two additions, so the result does not depend on their order
__device__ __forceinline__ void add_f32(float* p, float v) {
    (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v);
}

// 16-byte load past the caches (sc1 = agent scope): what another workgroup of the SAME launch stored with store4_through /
// store_u64x2_through.  One instruction per datum, so a tagged 16-byte record is seen whole or not at all.  The asm form is
// invisible to the compiler's vmcnt bookkeeping, hence the wait inside the statement.
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4v load16_past(const void* p) {
    u4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void load16_past_x4(const void* p0, const void* p1, const void* p2, const void* p3, u4v& a, u4v& b, u4v& c, u4v& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// two 8-byte words past the caches, one round trip
__device__ __forceinline__ void load8_past_x2(const void* p0, const void* p1, unsigned long long& a, unsigned long long& b) {
    asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void load16_past_x3(const void* p0, const void* p1, const void* p2, u4v& a, u4v& b, u4v& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
__device__ __forceinline__ void load16_past_x5(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, u4v& a, u4v& b, u4v& c, u4v& d,
                                               u4v& e) {
    asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                 "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
}
__device__ __forceinline__ u4v load16_past_epoch(const void* p, const unsigned int* epoch, unsigned int& ep_word) {
    u4v v;
    asm volatile("global_load_dword %1, %3, %4\n\tglobal_load_dwordx4 %0, %2, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v), "=&v"(ep_word) : "v"(p), "v"(0), "s"(epoch) : "memory");
    return v;
}
// ... and the workspace's epoch word in the same round trip (a wave that does not know the evaluation's tag yet: with_tag)
__device__ __forceinline__ void load16_past_x5_epoch(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const unsigned int* epoch,
                                                     u4v& a, u4v& b, u4v& c, u4v& d, u4v& e, unsigned int& ep_word) {
    asm volatile("global_load_dword %5, %11, %12\n\tglobal_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\t"
                 "global_load_dwordx4 %2, %8, off sc1\n\tglobal_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(ep_word)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(0), "s"(epoch)
                 : "memory");
}
// N words past the caches in ONE round trip: `base` (a scalar register pair) + a 32-bit byte offset per lane and word.  As a sequence of
// __hip_atomic_load the compiler issues them one at a time, an s_waitcnt vmcnt(0) behind each (atomics are not reordered against each other and
// each one's consumer is scheduled right behind it): R + D dependent trips to the L2 in front of every tile's pair loop -- six at 4-row tiles,
// ten at 8-row tiles, the "pred + masks" phase of the per-wave traces (1.4 / 3.3 us).  profiles/NOTES.md R6-11.
#define BXI_PW_LD(i) "global_load_dword %[v" #i "], %[o" #i "], %[b] sc1\n\t"
#define BXI_PW_OUT(i) [v##i] "=&v"(v[i])
#define BXI_PW_IN(i) [o##i] "v"(off[i])
template <int N>
__device__ __forceinline__ void load_words_past(const unsigned int* base_in, const uint32_t (&off)[N], uint32_t (&v)[N]) {
    static_assert(N >= 5 && N <= 12, "R + D of the tile kernels");
    // the base is wave-uniform by construction (a tile's image); said so explicitly: where the compiler cannot prove it (an ablation build did not)
    // an "s" operand is handed a VGPR pair and the assembler rejects the statement.  Two v_readfirstlane at most, none when the value is scalar already.
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(base_in);
    const unsigned int* base = reinterpret_cast<const unsigned int*>(
        ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)b64));
    if constexpr (N == 5)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), [b] "s"(base) : "memory");
    else if constexpr (N == 6)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), [b] "s"(base) : "memory");
    else if constexpr (N == 7)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), [b] "s"(base) : "memory");
    else if constexpr (N == 8)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) BXI_PW_LD(7) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6), BXI_PW_OUT(7)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), BXI_PW_IN(7), [b] "s"(base) : "memory");
    else if constexpr (N == 9)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) BXI_PW_LD(7) BXI_PW_LD(8) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6), BXI_PW_OUT(7), BXI_PW_OUT(8)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), BXI_PW_IN(7), BXI_PW_IN(8), [b] "s"(base) : "memory");
    else if constexpr (N == 10)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) BXI_PW_LD(7) BXI_PW_LD(8) BXI_PW_LD(9) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6), BXI_PW_OUT(7), BXI_PW_OUT(8), BXI_PW_OUT(9)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), BXI_PW_IN(7), BXI_PW_IN(8), BXI_PW_IN(9), [b] "s"(base) : "memory");
    else if constexpr (N == 11)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) BXI_PW_LD(7) BXI_PW_LD(8) BXI_PW_LD(9) BXI_PW_LD(10) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6), BXI_PW_OUT(7), BXI_PW_OUT(8), BXI_PW_OUT(9), BXI_PW_OUT(10)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), BXI_PW_IN(7), BXI_PW_IN(8), BXI_PW_IN(9), BXI_PW_IN(10), [b] "s"(base) : "memory");
    else if constexpr (N == 12)
        asm volatile(BXI_PW_LD(0) BXI_PW_LD(1) BXI_PW_LD(2) BXI_PW_LD(3) BXI_PW_LD(4) BXI_PW_LD(5) BXI_PW_LD(6) BXI_PW_LD(7) BXI_PW_LD(8) BXI_PW_LD(9) BXI_PW_LD(10) BXI_PW_LD(11) "s_waitcnt vmcnt(0)"
                     : BXI_PW_OUT(0), BXI_PW_OUT(1), BXI_PW_OUT(2), BXI_PW_OUT(3), BXI_PW_OUT(4), BXI_PW_OUT(5), BXI_PW_OUT(6), BXI_PW_OUT(7), BXI_PW_OUT(8), BXI_PW_OUT(9), BXI_PW_OUT(10), BXI_PW_OUT(11)
                     : BXI_PW_IN(0), BXI_PW_IN(1), BXI_PW_IN(2), BXI_PW_IN(3), BXI_PW_IN(4), BXI_PW_IN(5), BXI_PW_IN(6), BXI_PW_IN(7), BXI_PW_IN(8), BXI_PW_IN(9), BXI_PW_IN(10), BXI_PW_IN(11), [b] "s"(base) : "memory");
}
#undef BXI_PW_LD
#undef BXI_PW_OUT
#undef BXI_PW_IN
__device__ __forceinline__ float4 f4_of(const u4v& v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }

// ---- workspace ---------------------------------------------------------------------------------------------------------
struct Ws {
    float4* lab4;                               // [B,h,w] (L, a, b, 0)
    float* lab_planar;                          // [B,3,h,w] only the generic pooling path (other strides, unaligned canvases) fills it
    unsigned int* pred;                         // [B,h,w] epoch << 4 | bits; bit d = colour predicate of pair direction d with this pixel as the step pixel
    unsigned long long* colpart;                // [N,n_cb,w] packed (max logit, first row) of a band of rows
    unsigned long long* rowkey;                 // [N,n_rp,h] packed (max logit, first column)
    int n_cb, n_rp;
    int4* tab;                                  // [N+1] {tile prefix | img << 24, r0 | r1 << 16, c0 | c1 << 16, epoch}; [N].x = tiles
    unsigned int* bandflag;                     // [N,n_cb] epoch once a stream block's zero-fill and partial maxima are in memory (single-launch form)
    unsigned int* epoch;                        // [1] tag of the last evaluation FINISHED on this workspace (0 after the one-time zeroing); only the
                                                // finisher writes it, as its last act
    unsigned int ep;                            // this evaluation's tag (1 .. 2^28 - 1) = epoch + 1, read ON THE DEVICE by every kernel (with_tag):
                                                // data another workgroup of the SAME launch reads carries it.  Nothing about it is a kernel
                                                // argument, so a captured launch replayed from a hipGraph draws a fresh tag every time
    // words polled inside pair_kernel; zeroed by prep_kernel's table waves, i.e. before a kernel boundary
    unsigned long long* acc1;                   // [kAcc1Words] (one per 128 B) predicate workgroups: segments evaluated << 40 | sum W
    unsigned long long* sumw;                   // [1]   1 << 63 | sum W, published by the reducer wave once every segment is in (0 = not yet)
    unsigned long long* acc2;                   // [N][kAcc2Split] (one per 128 B) tile waves: arrivals << 52 | sum (W pw + 1) in 2^-24 units
    unsigned long long* dice;                   // [N]   leader: 1 << 32 | bits of the instance's dice loss (0 = not published)
    unsigned int* fault;                        // [1]   bit mask of waits that ran out (never expected)
    // what bxi_boxinst_targets_f32 leaves for evaluations with BXI_EVAL_TARGETS_READY (next to lab4 / pred)
    // (ONE pointer for the three regions: every field of this structure is a pair of scalar registers in every role of every kernel)
    unsigned char* tgt;                         // +0: tkey [1] u32, digest of the geometry / window / threshold the targets were computed for (0 = none)
                                                // +256: boxtab [kBoxCap] int4 per GT box {img << 24, r0 | r1 << 16, c0 | c1 << 16, 0}: what its predicate waves count against
                                                // +256 + 16 kBoxCap: boxcnt [kBoxCap][kBoxSplit] u64 (one per 128 B) per GT box: sum over its pixels p and the 8
                                                //   neighbours k of [sim_k(p) >= thresh]
    unsigned int pred_any;                      // 1: lab4 / pred come from bxi_boxinst_targets_f32 (an earlier launch): their tag field is not this evaluation's
    unsigned int ws_n16;                        // size of the workspace in 16-byte units (the finisher zeroes all of it when the tag counter is about to wrap)
    __host__ __device__ __forceinline__ unsigned int* tkey() const { return reinterpret_cast<unsigned int*>(tgt); }
    __host__ __device__ __forceinline__ int4* boxtab() const { return reinterpret_cast<int4*>(tgt + 256); }
    __host__ __device__ __forceinline__ unsigned long long* boxcnt() const { return reinterpret_cast<unsigned long long*>(tgt + 256 + 16 * (size_t)kBoxCap); }
};

__device__ __forceinline__ unsigned long long* acc2_word(unsigned long long* acc2, int n, int sub) {
    return acc2 + ((size_t)n * kAcc2Split + (sub & (kAcc2Split - 1))) * kAcc2Stride;
}

// This evaluation's tag: one more than the tag of the last evaluation that FINISHED on this workspace.  Every wave of an evaluation
// reads the word; only the finisher -- the last workgroup, which has by then seen every other wave of the launch arrive (each tile
// wave arrives exactly once, with or without tiles) -- writes it.  Evaluations that share a workspace are serialised by their stream,
// so the word is stable while anybody reads it, and a reader is always a LATER kernel than the writer: a scalar load (constant cache,
// invalidated at every dispatch) on one side, a plain store on the other.  What it costs is WHERE it is read: a wave that reads it
// first thing starts one dependent memory round trip late -- the whole launch with it (18.1 against 17.4 us with the tag as a kernel
// argument, same box).  The roles that head the launch's dependency chain (stream, pool) therefore read it behind their first loads
// (`after_loads`), where the round trip hides; the others wait for somebody anyway.  Likewise the finisher's store: written through
// (sc1) it is acknowledged ~0.3 us later than a plain one, and the launch ends on it.
__device__ __forceinline__ unsigned int next_tag(unsigned int e) { const unsigned int t = (e + 1u) & 0x0fffffffu; return t ? t : 1u; }
__device__ __forceinline__ Ws with_tag(Ws ws) {
    unsigned int e;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e) : "s"(ws.epoch) : "memory");
    ws.ep = next_tag(e);
    return ws;
}
// min(_iter / pairwise_warmup, 1) (condinst_head.py:1330-1331).  warmup >= 0: the caller's value.  warmup < 0: -warmup is
// pairwise_warmup and the factor comes from the device counter as it stands after this call's `self._iter += 1` (:1297; the finisher
// adds the 1 at the very end, behind every reader): a float32 add, then Python's double division, then the f32 operand of the multiply.
__device__ __forceinline__ float resolve_warmup(float warmup, const float* iter) {
    if (warmup >= 0.f) return warmup;
    const float it = __fadd_rn(__hip_atomic_load(iter, BXI_RLX, BXI_AGENT), 1.0f);
    return (float)fmin((double)it / (double)(-warmup), 1.0);
}

static inline int tile_width(int dil) { return 64 - 2 * dil; }
static inline int64_t eval_cap(int N, int h, int w, int dil, int R) {
    const int tw = tile_width(dil);
    return (int64_t)(N > 0 ? N : 1) * ((h + R - 1) / R) * ((w + tw - 1) / tw);
}

static size_t carve(void* base, int B, int N, int h, int w, Ws* ws) {
    const int N1 = N > 0 ? N : 1;
    const size_t Sn = (size_t)(h + kSBlk - 1) / kSBlk;
    const size_t cb_max = (size_t)(h + kYR * 2 - 1) / (kYR * 2), rp_max = (size_t)(w + kYC * 2 - 1) / (kYC * 2);   // the head-fused launch's tiles
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? p + o : nullptr; };
    Ws t;
    const size_t P = (size_t)h * w, B1 = B > 0 ? B : 1;
    // the epoch word FIRST, at offset 0 whatever the shape: evaluations of different shapes share a workspace (one per stream), and the
    // tag counter that tells their records apart must be the same word for all of them (everything behind it moves with the shape)
    t.epoch = (unsigned int*)take(4);
    t.lab4 = (float4*)take(16 * B1 * P);
    t.lab_planar = (float*)take(12 * B1 * P);
    t.pred = (unsigned int*)take(4 * B1 * P);
    t.tgt = (unsigned char*)take(256 + 16 * (size_t)kBoxCap + 8 * (size_t)kBoxCap * kBoxSplit * kAcc2Stride);
    t.colpart = (unsigned long long*)take(8 * (size_t)N1 * (cb_max > Sn ? cb_max : Sn) * w);
    t.rowkey = (unsigned long long*)take(8 * (size_t)N1 * h * (rp_max > 1 ? rp_max : 1));
    t.n_cb = (int)Sn; t.n_rp = 1;
    t.tab = (int4*)take(16 * (size_t)(N1 + 1));
    t.bandflag = (unsigned int*)take(4 * (size_t)N1 * (cb_max > Sn ? cb_max : Sn));
    t.ep = 0u; t.pred_any = 0u; t.ws_n16 = 0u;
    t.acc1 = (unsigned long long*)take(8 * (size_t)kAcc1Words * kAcc2Stride);
    t.sumw = (unsigned long long*)take(8);
    t.acc2 = (unsigned long long*)take(8 * (size_t)N1 * kAcc2Split * kAcc2Stride);
    t.dice = (unsigned long long*)take(8 * (size_t)N1);
    t.fault = (unsigned int*)take(4);
    if (ws) *ws = t;
    return off;
}

// ================================================================================================
// launch 1
// ================================================================================================
// ---- role 1: table waves (one wave per 64 table entries) --------------------------------------------------------------
struct LaneBox { int r0, r1, c0, c1, img, cnt; };
__device__ __forceinline__ int valid_cells(int limit_px, int stride, int n) {     // cells r with r*stride + stride/2 < limit_px
    const int half = stride / 2;
    const int v = limit_px - half <= 0 ? 0 : (limit_px - half + stride - 1) / stride;
    return min(v, n);
}
__device__ __forceinline__ LaneBox lane_box(const InstArgs& a, const ImageMeta& meta, int dil, int R, int m) {
    LaneBox lb = {0, 0, 0, 0, 0, 0};
    const int64_t g = a.gt_inds[m];
    const float* bp = nullptr;
    for (int b = 0; b < a.gt.B; ++b)      // uniform loop: the by-value kernel arguments are never indexed per lane
        if (g >= a.gt.first[b] && g < a.gt.first[b + 1]) {
            bp = a.gt.boxes[b] + 4 * (g - a.gt.first[b]); lb.img = b;
        }
    if (!bp) return lb;
    const Rect rc = box_rect(bp, a.Hc, a.Wc, a.stride, a.stride / 2, a.h, a.w);
    if (rc.r1 <= rc.r0 || rc.c1 <= rc.c0) return lb;
    lb.r0 = rc.r0; lb.r1 = rc.r1; lb.c0 = rc.c0; lb.c1 = rc.c1;
    const int r0 = max(rc.r0 - dil, 0), r1 = min(rc.r1 + dil, a.h);
    const int hc0 = max(rc.c0 - dil, 0), hc1 = min(rc.c1 + dil, a.w);
    const int tw = 64 - 2 * dil;
    lb.cnt = ((r1 - 1) / R - r0 / R + 1) * ((hc1 - hc0 + tw - 1) / tw);
    return lb;
}

__device__ __forceinline__ void publish_gathered_sumw(const InstArgs& a, const Ws& ws, int G, unsigned int key);
// `ready` != 0 (BXI_EVAL_TARGETS_READY; the value is the number of GT boxes + 1): the image side was evaluated by an earlier call (bxi_boxinst_targets_f32); sum W is then a GATHER --
// sum over the instances of their GT box's pair count (:1324-1328: the weights of instance n are its box's bitmask times the image's
// affinity mask, a function of the box and the image only) -- that the first table wave does behind its entries, instead of the
// predicate -> count -> reducer chain.  `key`: the digest of what the targets were computed for; a mismatch is a fault (loud).
__device__ __forceinline__ void table_wave(const InstArgs& a, const ImageMeta& meta, int dil, int R, const Ws& ws, const LossState& st, int k,
                                           bool write_status, int ready = 0, unsigned int key = 0u) {
    const int lane = threadIdx.x & 63;
    int base = 0, prefix = 0;
    LaneBox mine = {0, 0, 0, 0, 0, 0};
    for (int m0 = 0; m0 <= 64 * k; m0 += 64) {       // exclusive scan of the tile counts: deterministic offsets, no atomics
        const int m = m0 + lane;
        LaneBox lb = {0, 0, 0, 0, 0, 0};
        if (m < a.N) lb = lane_box(a, meta, dil, R, m);
        int incl = lb.cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (m0 == 64 * k) { prefix = base + incl - lb.cnt; mine = lb; }
        base += __shfl(incl, 63, 64);
    }
    const int m = 64 * k + lane;
    // the words that are polled later: zeroed here, written through, and DRAINED before the table entries that announce them go
    // out -- whoever holds a tagged entry m (entry 0) may use instance m's accumulators (the global ones).  No hipMemsetAsync, no
    // initialisation contract: in the two-launch form a kernel boundary follows anyway, in the single-launch form the tag orders it.
    if (m < a.N) {
        if (st.inst) { InstRec rc; rc.r0 = mine.r0; rc.r1 = mine.r1; rc.c0 = mine.c0; rc.c1 = mine.c1; rc.img = mine.img; rc.pad0 = rc.pad1 = rc.pad2 = 0; st.inst[m] = rc; }
#pragma unroll
        for (int sub = 0; sub < kAcc2Split; ++sub) __hip_atomic_store(acc2_word(ws.acc2, m, sub), 0ull, BXI_RLX, BXI_AGENT);
        __hip_atomic_store(&ws.dice[m], 0ull, BXI_RLX, BXI_AGENT);
    }
    if (k == 0) {
        __hip_atomic_store(&ws.acc1[(size_t)lane * kAcc2Stride], 0ull, BXI_RLX, BXI_AGENT);
        if (lane == 0) __hip_atomic_store(ws.sumw, 0ull, BXI_RLX, BXI_AGENT);
        if (lane == 0) __hip_atomic_store(ws.fault, 0u, BXI_RLX, BXI_AGENT);
        if (lane == 0 && st.status && write_status) { st.status[0] = 0; st.status[1] = R; }
    }
    drain_vmem();
    if (m < a.N)
        store_u64x2_through(reinterpret_cast<unsigned long long*>(ws.tab + m),
                            (unsigned long long)(unsigned int)(prefix | (mine.img << 24)) | ((unsigned long long)(unsigned int)(mine.r0 | (mine.r1 << 16)) << 32),
                            (unsigned long long)(unsigned int)(mine.c0 | (mine.c1 << 16)) | ((unsigned long long)ws.ep << 32));
    else if (m == a.N)
        store_u64x2_through(reinterpret_cast<unsigned long long*>(ws.tab + m), (unsigned long long)(unsigned int)prefix, (unsigned long long)ws.ep << 32);
    if (k != 0 || ready < 0) return;                // (ready < 0: targets ready, sum W is gathered by the reducer workgroup -- the single-launch form)
    if (!ready) {       // an evaluation that computes the image side itself overwrites lab4 / pred: targets an earlier call left are gone
        if (lane == 0) *ws.tkey() = 0u;
        return;
    }
    publish_gathered_sumw(a, ws, ready - 1, key);
}

// sum W = sum over the instances of their GT box's pair count (bxi_boxinst_targets_f32 left the counts): one wave.
// The targets must be THIS call's: the digest `key` covers the geometry and the box COUNTS (host data); the box COORDINATES are device data, so
// every instance's box is mapped to its cells again here -- from the evaluation's own boxes -- and compared with the rectangle the targets call
// recorded for that box (the counts were taken inside it).  A mismatch is a fault: NaN losses and a status word, never the old boxes' normaliser
// under the new boxes' rectangles.  (The IMAGE's pixels are not compared -- the evaluation does not read them with the targets ready: that the
// targets were made from this batch's images is the caller's side of the contract, include/boxinst_hip.h.)
__device__ __forceinline__ void publish_gathered_sumw(const InstArgs& a, const Ws& ws, int G, unsigned int key) {
    const int lane = threadIdx.x & 63;
    const unsigned int have = __hip_atomic_load(ws.tkey(), BXI_RLX, BXI_AGENT);
    double tot = 0.0;
    bool other_boxes = false;
    for (int m0 = 0; m0 < a.N; m0 += 64) {
        const int mm = m0 + lane;
        const int64_t g = mm < a.N ? a.gt_inds[mm] : -1;
        if (g >= 0 && g < G && g < kBoxCap) {
            unsigned long long c[kBoxSplit];
#pragma unroll
            for (int j = 0; j < kBoxSplit; ++j) c[j] = __hip_atomic_load(ws.boxcnt() + ((size_t)g * kBoxSplit + j) * kAcc2Stride, BXI_RLX, BXI_AGENT);
            const int4 rec = ws.boxtab()[g];                        // (img << 24, r0 | r1 << 16, c0 | c1 << 16, 0): written by an earlier launch
            const float* bp = nullptr;
            int img = 0;
            for (int b = 0; b < a.gt.B; ++b)                         // uniform loop: the by-value kernel arguments are never indexed per lane
                if (g >= a.gt.first[b] && g < a.gt.first[b + 1]) { bp = a.gt.boxes[b] + 4 * (g - a.gt.first[b]); img = b; }
            Rect rc = {0, 0, 0, 0};
            if (bp) rc = box_rect(bp, a.Hc, a.Wc, a.stride, a.stride / 2, a.h, a.w);
            other_boxes |= !bp || rec.x != (img << 24) || rec.y != (rc.r0 | (rc.r1 << 16)) || rec.z != (rc.c0 | (rc.c1 << 16));
#pragma unroll
            for (int j = 0; j < kBoxSplit; ++j) tot += (double)c[j];
        }
    }
    tot = wave_total_f64(tot);                                             // exact: integers far below 2^53
    const bool bad = __any(other_boxes) || have != key || key == 0u;
    if (lane == 0)
        __hip_atomic_store(ws.sumw, (1ull << 63) | (bad ? kSumwFault : 0ull) | (unsigned long long)tot, BXI_RLX, BXI_AGENT);
}

// This lane's table entry m (m <= N; `want` false: nothing).  Two-launch form: a plain load behind the kernel boundary.  Single-launch
// form (ONE): read past the caches until every wanted entry carries this evaluation's tag -- the table workgroup is the first of
// the grid and waits for nobody, so this is a wait for a workgroup that precedes the asker.  false = the bounded wait ran out.
template <bool ONE>
__device__ __forceinline__ bool tab_entry(const Ws& ws, int m, bool want, int spin_limit, int4& e) {
    if (!ONE) { e = want ? ws.tab[m] : make_int4(0, 0, 0, 0); return true; }
    for (int spins = 0; spins <= spin_limit; ++spins) {
        const u4v v = load16_past(ws.tab + (want ? m : 0));
        if (__all(!want || v.w == ws.ep)) {
            e = want ? make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w) : make_int4(0, 0, 0, 0);
            BXI_WL(1, spins);
            return true;
        }
        __builtin_amdgcn_s_sleep(BXI_SLEEP_TAB);
    }
    e = make_int4(0, 0, 0, 0);
    return false;
}
// every entry 0..N tagged = every polled word of this evaluation zeroed (finisher, reducer)
template <bool ONE>
__device__ __forceinline__ bool table_complete(const Ws& ws, int N, int spin_limit) {
    if (!ONE) return true;
    const int lane = threadIdx.x & 63;
    int4 e;
    for (int m0 = 0; m0 <= N; m0 += 64)
        if (!tab_entry<true>(ws, m0 + lane, m0 + lane <= N, spin_limit, e)) return false;
    return true;
}

// ---- role 2: stream block = 4 waves x 8 rows of one instance map ---------------------------------------------------------
struct LogitRows {
    const float* L; int w, vec, nt;
    __device__ __forceinline__ float4 operator()(int r, int c) const {
        if (vec && nt) {      // non-temporal (launch_fused_eval decides: maps that outgrow the L2)
            typedef float f4n __attribute__((ext_vector_type(4)));
            const f4n t_ = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(L + (int64_t)r * w + c));
            return make_float4(t_.x, t_.y, t_.z, t_.w);
        }
        return load4(L + (int64_t)r * w, c, w, vec);
    }
};

struct NoHook { __device__ __forceinline__ void operator()(Ws&) const {} };

// `after_loads(ws)` runs once the zero-fill stores and the first loads are issued: the place for work whose latency should hide
// behind them (the evaluation's tag, read from the device: with_tag)
template <bool ONE, typename Src, typename Hook = NoHook>
__device__ __forceinline__ void stream_block(const InstArgs& a, Ws& ws, float* __restrict__ g_logits, int vec, int sb,
                                             unsigned long long* colp /* LDS [kWaves][w] */, const Src& src, int tix, const Hook& after_loads = Hook()) {
    const int h = a.h, w = a.w;
    const int Sn = (h + kSBlk - 1) / kSBlk;
    const int n = sb / Sn, s = sb % Sn;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = s * kSBlk + wv * kSRows, r1 = min(h, r0 + kSRows);     // may be empty
    const int64_t P = (int64_t)h * w;
    float* G = g_logits ? g_logits + (int64_t)n * P : nullptr;
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);

    if (G && !BXI_AB(16))   // zero-fill of d loss / d logits (depends on nothing); written through: drains while the launch is still reading
        for (int cb = 0; cb < w; cb += kChunkC) {
            const int c = cb + lane * 4;
            if (c < w) {
#pragma unroll
                for (int i = 0; i < kSRows; ++i)
                    if (r0 + i < r1) {
                        if (vec) store4_through(G + (int64_t)(r0 + i) * w + c, 0.f, 0.f, 0.f, 0.f);
                        else
                            for (int j = 0; j < 4; ++j)
                                if (c + j < w) __hip_atomic_store(G + (int64_t)(r0 + i) * w + c + j, 0.f, BXI_RLX, BXI_AGENT);
                    }
            }
        }
    float4 v[kSRows];
    {
        const int c = lane * 4;
#pragma unroll
        for (int i = 0; i < kSRows; ++i) v[i] = (r0 + i < r1 && c < w && !BXI_AB(64)) ? src(r0 + i, c) : ninf;
    }
    after_loads(ws);
    BXI_TW(0, tix, 1);
    float rmax[kSRows]; int rcol[kSRows];
#pragma unroll
    for (int i = 0; i < kSRows; ++i) { rmax[i] = -INFINITY; rcol[i] = 0; }
    for (int cb = 0;;) {
        const int c = cb + lane * 4;
        if (c < w) {
            float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int crow[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < kSRows; ++i) {
                if (r0 + i < r1) {
                    float m = v[i].x; int mc = c;                       // first column wins ties
                    if (v[i].y > m) { m = v[i].y; mc = c + 1; }
                    if (v[i].z > m) { m = v[i].z; mc = c + 2; }
                    if (v[i].w > m) { m = v[i].w; mc = c + 3; }
                    if (m > rmax[i]) { rmax[i] = m; rcol[i] = mc; }     // chunks ascend: strict > keeps the first
                    if (v[i].x > cmax[0]) { cmax[0] = v[i].x; crow[0] = i; }   // ascending row, strict >: first row wins
                    if (v[i].y > cmax[1]) { cmax[1] = v[i].y; crow[1] = i; }
                    if (v[i].z > cmax[2]) { cmax[2] = v[i].z; crow[2] = i; }
                    if (v[i].w > cmax[3]) { cmax[3] = v[i].w; crow[3] = i; }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < w) colp[(size_t)wv * w + c + j] = pack_max(cmax[j], (uint32_t)(r0 + crow[j]));   // absolute row
        }
        cb += kChunkC;
        if (cb >= w) break;
        const int c2 = cb + lane * 4;
#pragma unroll
        for (int i = 0; i < kSRows; ++i) v[i] = (r0 + i < r1 && c2 < w) ? src(r0 + i, c2) : ninf;
    }
    BXI_TW(0, tix, 2);
    float wmax[kSRows];
#pragma unroll
    for (int i = 0; i < kSRows; ++i) wmax[i] = rmax[i];
    // eight maxima over the wave side by side: within rows of 16 lanes by DPP, the four rows by v_readlane (no LDS crossbar)
    wave_total_steps([&](int c) {
        float o[kSRows];
#pragma unroll
        for (int i = 0; i < kSRows; ++i) o[i] = __int_as_float(dpp_i32(__float_as_int(wmax[i]), c));
#pragma unroll
        for (int i = 0; i < kSRows; ++i) wmax[i] = fmaxf(wmax[i], o[i]);
    });
#pragma unroll
    for (int i = 0; i < kSRows; ++i) {
        const int b = __float_as_int(wmax[i]);
        wmax[i] = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                        fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
    }
    unsigned long long mine = 0ull;
#pragma unroll
    for (int i = 0; i < kSRows; ++i) {
        const unsigned long long who = __ballot(rmax[i] == wmax[i]);
        const int first = who ? __ffsll((long long)who) - 1 : 0;
        const int col = __builtin_amdgcn_readlane(rcol[i], first);
        if (lane == i) mine = pack_max(wmax[i], (uint32_t)col);
    }
    if (lane < kSRows && r0 + lane < r1) {
        if (ONE) __hip_atomic_store(&ws.rowkey[(int64_t)n * h + r0 + lane], mine, BXI_RLX, BXI_AGENT);     // written through: read by a leader of this launch
        else ws.rowkey[(int64_t)n * h + r0 + lane] = mine;
    }
    BXI_TW(0, tix, 3);
    lds_barrier();
    BXI_TW(0, tix, 4);
    for (int c = threadIdx.x; c < w; c += kWaves * 64) {
        unsigned long long k = colp[c];
#pragma unroll
        for (int u = 1; u < kWaves; ++u) { const unsigned long long o = colp[(size_t)u * w + c]; k = o > k ? o : k; }
        if (ONE) __hip_atomic_store(&ws.colpart[((int64_t)n * Sn + s) * w + c], k, BXI_RLX, BXI_AGENT);
        else ws.colpart[((int64_t)n * Sn + s) * w + c] = k;       // larger value, then smaller row
    }
    if (ONE) {
        // single-launch form: this band's zero-fill and partial maxima are in memory (every wave drains its own stores, the
        // workgroup meets) before the band's flag says so to the instance's leader and to the tile waves that add onto these rows
        drain_vmem();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&ws.bandflag[(int64_t)n * ws.n_cb + s], ws.ep, BXI_RLX, BXI_AGENT);
    }
}

// ---- role 3: pool block = the 4 input rows of 64 pooled pixels ---------
__device__ __forceinline__ double lab_f(const double* lut, int i, int r8, int g8, int b8) {
    const double r = lut[r8], g = lut[g8], b = lut[b8];
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    const double m0 = i == 0 ? M[0][0] : (i == 1 ? M[1][0] : M[2][0]);
    const double m1 = i == 0 ? M[0][1] : (i == 1 ? M[1][1] : M[2][1]);
    const double m2 = i == 0 ? M[0][2] : (i == 1 ? M[1][2] : M[2][2]);
    const double wt = i == 0 ? white[0] : (i == 1 ? white[1] : white[2]);
    const double acc = __dadd_rn(__dadd_rn(__dmul_rn(m0, r), __dmul_rn(m1, g)), __dmul_rn(m2, b));
    const double v = acc / wt;
    return v > 0.008856 ? cbrt(v) : __dadd_rn(__dmul_rn(7.787, v), 16.0 / 116.0);
}

__device__ __forceinline__ void pool_load(const PoolArgs& pa, int item, int segs, int h, int w, float4 (&v)[3]) {
    const int seg = item % segs, r = (item / segs) % h, b = item / (segs * h);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = seg * 64 + lane;
    const int64_t plane = (int64_t)pa.Hc * pa.Wc;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < w && !BXI_AB(32)) {
        const float* base = pa.imgs + (int64_t)b * 3 * plane + (int64_t)(4 * r + wv) * pa.Wc + 4 * c;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            // non-temporal: 19.7 MB at 2 x 800 x 1024 that nobody reads twice -- kept out of the L2's way they leave it to the logits, the Lab records
            // and the predicate words the rest of the launch asks for again: 17.39 -> 16.92 us per evaluation at 32 instances, 22.4 -> 21.9 at 64,
            // 37.4 -> 36.7 at 128 (same box, interleaved three times; profiles/NOTES.md R6-7)
            typedef float f4n __attribute__((ext_vector_type(4)));
            const f4n t_ = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(base + pa.dn.src_ch[ch] * plane));
            v[ch] = make_float4(t_.x, t_.y, t_.z, t_.w);
        }
    }
}

__device__ __forceinline__ float n2_of(float L0, float A0, float B0, float L1, float A1, float B1) {
    const float dL = L0 - L1, dA = A0 - A1, dB = B0 - B1;     // un-fused: the decision must equal get_image_color_similarity's (:237)
    return __fadd_rn(__fadd_rn(__fmul_rn(dL, dL), __fmul_rn(dA, dA)), __fmul_rn(dB, dB));
}

// items first, first + step, ... < n_items
template <typename Hook = NoHook>
__device__ __forceinline__ void pool_block(const PoolArgs& pa, Ws& ws, int first, int step, int n_items, double* lut /*[256]*/,
                                           int* part /*[4][3][64]*/, double* fch /*[3][64]*/, int tix, const Hook& after_loads = Hook()) {
    const int h = pa.Hc >> 2, w = pa.Wc >> 2;
    const int segs = (w + 63) >> 6;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 v[3], nx[3];
    pool_load(pa, first, segs, h, w, v);
    after_loads(ws);
    lut[threadIdx.x] = kSrgbLut[threadIdx.x];            // staged while the image loads fly
    for (int item = first; item < n_items; item += step) {
        const bool more = item + step < n_items;         // workgroup-uniform
        if (more) pool_load(pa, item + step, segs, h, w, nx);
        const int seg = item % segs, r = (item / segs) % h, b = item / (segs * h);
        const int c = seg * 64 + lane;
        const int y = 4 * r + wv;
        const bool act = c < w;
        const int ih = pa.meta.img_h[b], iw = pa.meta.img_w[b];
        const int x0 = 4 * c;
        const bool yin = y < ih;
        int sum[3];
        if (__all(!act || (yin && x0 + 3 < iw))) {       // wave-uniform: the whole row segment is image, not canvas padding
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
                sum[ch] = denorm_u8(v[ch].x, s, m) + denorm_u8(v[ch].y, s, m) + denorm_u8(v[ch].z, s, m) + denorm_u8(v[ch].w, s, m);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const double s = pa.dn.stdv[pa.dn.src_ch[ch]], m = pa.dn.mean[pa.dn.src_ch[ch]];
                int t = 0;
                t += (yin && x0 + 0 < iw) ? denorm_u8(v[ch].x, s, m) : 0;
                t += (yin && x0 + 1 < iw) ? denorm_u8(v[ch].y, s, m) : 0;
                t += (yin && x0 + 2 < iw) ? denorm_u8(v[ch].z, s, m) : 0;
                t += (yin && x0 + 3 < iw) ? denorm_u8(v[ch].w, s, m) : 0;
                sum[ch] = t;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) part[(wv * 3 + ch) * 64 + lane] = sum[ch];
        BXI_TW(0, tix, 1);
        lds_barrier();
        BXI_TW(0, tix, 2);
        if (wv < 3) {                                     // wave-uniform: wave i takes channel i of XYZ -> f_i
            int px[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                px[ch] = (part[(0 * 3 + ch) * 64 + lane] + part[(1 * 3 + ch) * 64 + lane] + part[(2 * 3 + ch) * 64 + lane] +
                          part[(3 * 3 + ch) * 64 + lane]) >> 4;
            fch[wv * 64 + lane] = BXI_AB(1) ? 0.001 * (double)(px[0] + 2 * px[1] + 3 * px[2] + wv) : lab_f(lut, wv, px[0], px[1], px[2]);
        }
        BXI_TW(0, tix, 3);
        lds_barrier();
        BXI_TW(0, tix, 4);
        if (wv == 3 && act) {       // one 16-byte store per pooled pixel (the wave that had no channel to compute)
            const double f0 = fch[lane], f1 = fch[64 + lane], f2 = fch[128 + lane];
            // the fourth component is this evaluation's tag: a predicate wave of the SAME launch (single-launch form) re-reads a pixel
            // until it carries it; the record is one 16-byte store, written through
            store4_through(reinterpret_cast<float*>(ws.lab4 + ((int64_t)b * h + r) * w + c), (float)__dadd_rn(__dmul_rn(116.0, f1), -16.0),
                           (float)__dmul_rn(500.0, __dadd_rn(f0, -f1)), (float)__dmul_rn(200.0, __dadd_rn(f1, -f2)), __uint_as_float(ws.ep));
        }
        // the next trip's `part` writes come after this barrier; its `fch` writes after the next one, which wave 3 reaches only
        // after it has read `fch` here: no extra barrier needed
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = nx[ch];
    }
}

// (prep_kernel, the first launch of the two-launch form, follows the roles of the second launch below: its folded form runs two of them)

// ---- head-fused first launch ----------------------------------------------------------------------------
// CondInstMaskHead.forward (condinst_head.py:1139-1164) and the evaluation's first launch as ONE grid of independent roles:
//   [table blocks][pool blocks][head tiles: instance x 8 x 32 tiles of y -> 16 x 64 logits]
// A head tile does the stream role's job on the tile it just produced: zero-filled gradient tile (written through), per-row and
// per-column (value, first index) maxima as partials for the leaders.  Nothing in the launch waits for anything else in it.
template <int C, bool REL>
__global__ __launch_bounds__(256, 7) void head_prep_kernel(PoolArgs pa, int n_pool, int n_items, InstArgs a, int dil, int R, Ws ws_in, LossState st,
                                                            float* __restrict__ g_logits, DynArgs da, const float* __restrict__ params,
                                                            float* __restrict__ logits_out, int ready, unsigned int key) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ws ws = with_tag(ws_in);
    const int n_tab = ((a.N + 64) / 64 + kWaves - 1) / kWaves;
    const int blk = (int)blockIdx.x;
    const int tix = blk * kWaves + (int)(threadIdx.x >> 6);
    (void)tix;
    if (blk < n_tab) {
        const int k = blk * kWaves + (int)(threadIdx.x >> 6);
        if (64 * k <= a.N) table_wave(a, pa.meta, dil, R, ws, st, k, true, ready, key);
    } else if (blk < n_tab + n_pool) {
        double* lut = reinterpret_cast<double*>(smem);
        double* fch = lut + 256;
        int* part = reinterpret_cast<int*>(fch + 3 * 64);
        pool_block(pa, ws, blk - n_tab, n_pool, n_items, lut, part, fch, tix);
    } else {
        const int tiles_x = (da.W + kYC - 1) / kYC, tiles_y = (da.H + kHeadR - 1) / kHeadR;
        int t = blk - n_tab - n_pool;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        unsigned long long* ckeys = reinterpret_cast<unsigned long long*>(smem);          // [4][64]
        float* otile = reinterpret_cast<float*>(ckeys + 4 * 64);                          // [2 kHeadR][64]
        float* ytile = otile + 2 * kHeadR * 64;                                           // [(kHeadR+1)*(kYC+1)]
        const DynEpi ep = {ws.colpart, ws.rowkey, g_logits, ws.n_cb, ws.n_rp, 0};
        dyn_tile_forward<C, REL, 2, true, kHeadR, kYC>(da, params, logits_out, n, ty, tx, ytile, otile, ckeys, ep);
    }
}

// ---- the image side for strides other than 4 / unaligned canvases: launches of their own (pool_rgb_generic of
// color_affinity.hip -> Lab planes, then this repacking) -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_lab4_kernel(const float* __restrict__ lab, float4* __restrict__ lab4, const unsigned int* __restrict__ epoch, int B,
                                                         int64_t P) {
    const unsigned int ep = next_tag(*epoch);          // as with_tag: the evaluation's tag is device state, never a kernel argument
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * P; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / P, p = i - b * P;
        const float* src = lab + b * 3 * P + p;
        lab4[i] = make_float4(src[0], src[P], src[2 * P], __uint_as_float(ep));
    }
}

// ================================================================================================
// launch 2
// ================================================================================================
template <int D, int R> struct TG { static constexpr int RD = R + 2 * D, TW = 64 - 2 * D; };

struct Tile {                                   // wave-uniform (SGPRs)
    int r0, r1, c0, c1;                         // cells whose sample lies in the GT box (bitmask == 1)
    int img, n, tile_r0, tile_c0;
    int vrow, vcol;                             // valid(q) <=> row(q) < vrow && col(q) < vcol
    int hc1;                                    // end column of the instance's tile hull (= dilated box)
};

__device__ __forceinline__ uint32_t row_bits(int lo, int hi, int base, int n) {   // bits j in [0,n) with lo <= base + j < hi
    const int a = max(lo - base, 0), b = min(hi - base, n);
    if (b <= a) return 0u;
    return ((1u << b) - 1u) & ~((1u << a) - 1u);              // n <= 16
}

// The four pair directions of a step i (j = i + D), every one between this lane and the lane D to its right or itself, so
// that only right-neighbour values are ever fetched:
//   0: A = (i, l)  B = (i, l + D)   |   1: A = (j, l)  B = (i, l + D)   |   2: A = (i, l)  B = (j, l)   |   3: A = (i, l)  B = (j, l + D)
// Pair weights, per step i:  W[k, A] and W[7 - k, B] -- [A in the GT box][B a valid image pixel][colour predicate of the pair] and the mirror --,
// and the same restricted to pixels this tile owns (math_tile builds them as bytes).
__device__ __forceinline__ uint32_t spread4(uint32_t x4) { return (x4 * 0x00204081u) & 0x01010101u; }   // bits 0..3 -> bytes 0..3
// the two 16-bit halves of a word times those of another (v_pk_mul_lo_u16 / v_pk_mad_u16: full rate, where a 32-bit multiply is a quarter-rate
// instruction and the 24-bit one loses the fourth byte)
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2v)(__builtin_bit_cast(us2v, a) * __builtin_bit_cast(us2v, b))); }
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, (us2v)(__builtin_bit_cast(us2v, a) * __builtin_bit_cast(us2v, b) + __builtin_bit_cast(us2v, c)));
}

template <int D>
__device__ __forceinline__ float lane_plus(float v) {
    int x = __float_as_int(v);
#pragma unroll
    for (int s = 0; s < D; ++s) x = __builtin_amdgcn_mov_dpp(x, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
    return __int_as_float(x);
}
template <int D>
__device__ __forceinline__ float lane_minus(float v) {
    int x = __float_as_int(v);
#pragma unroll
    for (int s = 0; s < D; ++s) x = __builtin_amdgcn_mov_dpp(x, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
    return __int_as_float(x);
}

// Generic (slow) evaluation of one tile: ordered pairs per owned pixel straight from global memory, pair value and
// gradient in log space exactly as pairwise.cu:38-61.  Taken for thresh <= 0 (zero_bit: padded / masked-out neighbours
// weigh 1) and for tiles with saturated logits (S underflows).  Gradients -> gout, the lane's sum W pw -> gout[R].
template <int D, int R, bool ONE>
__device__ __forceinline__ void slow_tile(const float* __restrict__ Lg, const float4* __restrict__ lab4, const Tile& t, float n2max, int zero_bit,
                                          int h, int w, int lane, float* gout /* LDS [R + 1][64] */) {
    const int c = t.tile_c0 - D + lane;
    const bool col_owned = lane >= D && lane < 64 - D && c < t.hc1;
    float num = 0.f;
    const float4* L0p = lab4 + (int64_t)t.img * h * w;
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
        const int r = t.tile_r0 + j;
        float gacc = 0.f;
        if (col_owned && r < h) {
            const bool in_p = r >= t.r0 && r < t.r1 && c >= t.c0 && c < t.c1;
            const bool val_p = r < t.vrow && c < t.vcol;
            const int64_t pi = (int64_t)r * w + c;
            const float4 lp = ONE ? f4_of(load16_past(L0p + pi)) : L0p[pi];      // single-launch form: written by this launch, read past the caches
            const float xa = Lg[pi];
            const float ax = logsig(xa), bx = logsig(-xa);
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                const int kk = k < 4 ? k : k + 1;
                const int r2 = r + (kk / 3 - 1) * D, c2 = c + (kk % 3 - 1) * D;
                const bool inq = r2 >= 0 && r2 < h && c2 >= 0 && c2 < w;
                uint32_t pn = 0u;
                int64_t qi = 0;
                if (inq) {
                    qi = (int64_t)r2 * w + c2;
                    const float4 lq = ONE ? f4_of(load16_past(L0p + qi)) : L0p[qi];
                    pn = n2_of(lp.x, lp.y, lp.z, lq.x, lq.y, lq.z) <= n2max ? 1u : 0u;
                }
                const bool val_q = inq && r2 < t.vrow && c2 < t.vcol;
                const bool in_q = inq && r2 >= t.r0 && r2 < t.r1 && c2 >= t.c0 && c2 < t.c1;
                const uint32_t wp = in_p ? (val_q ? pn : (uint32_t)zero_bit) : 0u;
                const uint32_t wq = in_q ? (val_p ? pn : (uint32_t)zero_bit) : 0u;
                if (inq && (wp + wq)) {
                    const float xb = Lg[qi];
                    const float ay = logsig(xb), by = logsig(-xb);
                    const float e1 = ax + ay, e0 = bx + by;
                    const float nl2 = logsig(fabsf(e1 - e0)) - fmaxf(e1, e0);
                    num += (float)wp * nl2;
                    gacc += (float)(wp + wq) * (-(expf(ay) - expf(by)) * expf(ax + bx + nl2));
                }
            }
        }
        gout[j * 64 + lane] = gacc;
    }
    gout[R * 64 + lane] = num;
}

template <int D, int R>
__device__ __forceinline__ void load_plane(const float* __restrict__ plane, const Tile& t, int h, int w, int lane, float (&v)[R + 2 * D]) {
    const uint32_t cc4 = (uint32_t)min(max(t.tile_c0 - D + lane, 0), w - 1) * 4u;
    const char* pb = reinterpret_cast<const char*>(plane);                       // scalar base + 32-bit byte offset (one plane < 2^31 bytes)
#pragma unroll
    for (int j = 0; j < R + 2 * D; ++j) {
        const uint32_t rr = (uint32_t)min(max(t.tile_r0 - D + j, 0), h - 1);       // clamped: pairs with a pixel outside the map weigh 0
        v[j] = *reinterpret_cast<const float*>(pb + (rr * (uint32_t)w * 4u + cc4));
    }
}

// ---- predicate wave: one pooled row segment (64 pixels) of one image ------------------------------------------------------
// The colour pairs whose step row is pooled row r of segment `seg` of image b -- directions (get_image_color_similarity :220-246
// through unfold_wo_center's offsets :190-217, each unordered pair ONCE PER IMAGE, not once per instance and tile):
//   0: (r, c) - (r, c+D)    1: (r+D, c) - (r, c+D)    2: (r, c) - (r+D, c)    3: (r, c) - (r+D, c+D)
// -> one predicate byte per pixel (bit d: squared Lab distance <= n2max, i.e. sim >= thresh for a valid neighbour), and the
// segment's share of  sum W = sum_n sum_{p in box n} sum_k [sim_k(p) >= thresh]  (:1324-1328): a pair (p, q) weighs
// [p in box n][q valid] + [q in box n][p valid] for every instance n of the image (returned per lane; the workgroup arrives
// once with its total).  A byte carries its own "evaluated" bit: a tile wave re-reads the few bytes it needs until they have it.
__device__ __forceinline__ float lane_plus_n(float v, int d) {
    int x = __float_as_int(v);
    for (int s = 0; s < d; ++s) x = __builtin_amdgcn_mov_dpp(x, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
    return __int_as_float(x);
}
// Inclusive prefix sum over the 64 lanes on the DPP path: Hillis-Steele inside a row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source add 0),
// then the rows' totals by row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3).  Ten VALU instructions, no LDS.
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}
struct ValidCells { int vrow[BXI_MAX_IMAGES], vcol[BXI_MAX_IMAGES]; };   // per image: valid(q) <=> row(q) < vrow && col(q) < vcol (host: :1354-1369,:1405)
// PER_BOX (bxi_boxinst_targets_f32's second launch; never ONE): the rectangles are the GT BOXES' (ws.boxtab, n_ent of them) instead of the
// instances', and what a box containing a site adds is kept PER BOX (`boxacc`, LDS of the workgroup, one counter per box) instead of
// summed -- the evaluation that follows gathers sum W from its instances' boxes.  No tag is read or written (the consumer is a later launch).
template <bool ONE, bool PER_BOX = false>
__device__ __forceinline__ int pred_item(int h, int w, int n_ent, const ValidCells& vc, Ws& ws, int D, float n2max, int item, int segs, int spin_limit, bool& ok,
                                         int* boxacc = nullptr) {
    const int lane = threadIdx.x & 63;
    const int seg = item % segs, r = (item / segs) % h, b = item / (segs * h);
    const int c = seg * 64 + lane, cn = c + D;
    const bool rowD = r + D < h;                                  // wave-uniform
    const float4* L4 = ws.lab4 + (int64_t)b * h * w;
    const int cc = min(c, w - 1), cx = min(lane >= 64 - D ? cn : c, w - 1), rD = min(r + D, h - 1);
    // this row, the row D below, and for the last D lanes their right neighbours (they live in the next segment)
    float4 o0, oD, x0, xD;
    // lane n: instance n's table entry (box cells, image), requested with the Lab
    int4 rect, rect1 = make_int4(-1, 0, 0, 0);
    if (ONE) {
        // single-launch form: the pool workgroups of THIS launch write these pixels (16-byte records carrying the evaluation's
        // tag, written through); they precede this wave in the grid and wait for nobody
        bool got = false;
        for (int spins = 0; spins <= spin_limit; ++spins) {
            u4v q0, q1, q2, q3, qe;      // the table entry travels with the pixels: one round trip
            if (ws.ep == 0u) {           // wave-uniform: this wave's first poll -- the evaluation's tag travels with it too (with_tag)
                unsigned int e;
                load16_past_x5_epoch(L4 + (int64_t)r * w + cc, L4 + (int64_t)rD * w + cc, L4 + (int64_t)r * w + cx, L4 + (int64_t)rD * w + cx,
                                     ws.tab + (lane < n_ent ? lane : 0), ws.epoch, q0, q1, q2, q3, qe, e);
                ws.ep = next_tag((unsigned int)__builtin_amdgcn_readfirstlane((int)e));
            } else
                load16_past_x5(L4 + (int64_t)r * w + cc, L4 + (int64_t)rD * w + cc, L4 + (int64_t)r * w + cx, L4 + (int64_t)rD * w + cx,
                               ws.tab + (lane < n_ent ? lane : 0), q0, q1, q2, q3, qe);
            if (__all(q0.w == ws.ep && q1.w == ws.ep && q2.w == ws.ep && q3.w == ws.ep && qe.w == ws.ep)) {
                o0 = f4_of(q0); oD = f4_of(q1); x0 = f4_of(q2); xD = f4_of(q3);
                rect = lane < n_ent ? make_int4((int)qe.x, (int)qe.y, (int)qe.z, (int)qe.w) : make_int4(-1, 0, 0, 0);
                got = true;
                BXI_WL(2, spins);
                break;
            }
            __builtin_amdgcn_s_sleep(BXI_SLEEP_PRED);
        }
        if (!got) { ok = false; return 0; }
    } else {
        const bool untagged = !PER_BOX && ws.ep == 0u;               // wave-uniform: this wave's first item -- the epoch word rides with its loads
        unsigned int ew = 0u;
        if (untagged) ew = *ws.epoch;                                // (a plain load: the word was written by an earlier kernel)
        o0 = L4[(int64_t)r * w + cc]; oD = L4[(int64_t)rD * w + cc]; x0 = L4[(int64_t)r * w + cx]; xD = L4[(int64_t)rD * w + cx];
        rect = lane < n_ent ? (PER_BOX ? ws.boxtab()[lane] : ws.tab[lane]) : make_int4(-1, 0, 0, 0);
        // (entries 64..127 ride with the same round trip: a load per 64-entry chunk BEHIND the first chunk's arithmetic was a second dependent trip
        // in every item of an evaluation of more than 64 instances)
        if (n_ent > 64) rect1 = 64 + lane < n_ent ? (PER_BOX ? ws.boxtab()[64 + lane] : ws.tab[64 + lane]) : make_int4(-1, 0, 0, 0);
        if (untagged) ws.ep = next_tag((unsigned int)__builtin_amdgcn_readfirstlane((int)ew));
    }
    float nL = lane_plus_n(o0.x, D), nA = lane_plus_n(o0.y, D), nB = lane_plus_n(o0.z, D);
    float mL = lane_plus_n(oD.x, D), mA = lane_plus_n(oD.y, D), mB = lane_plus_n(oD.z, D);
    if (lane >= 64 - D) { nL = x0.x; nA = x0.y; nB = x0.z; mL = xD.x; mA = xD.y; mB = xD.z; }
    const bool cin = c < w, nin = cn < w;
    const bool p0 = cin && nin && n2_of(o0.x, o0.y, o0.z, nL, nA, nB) <= n2max;
    const bool p1 = cin && nin && rowD && n2_of(oD.x, oD.y, oD.z, nL, nA, nB) <= n2max;
    const bool p2 = cin && rowD && n2_of(o0.x, o0.y, o0.z, oD.x, oD.y, oD.z) <= n2max;
    const bool p3 = cin && nin && rowD && n2_of(o0.x, o0.y, o0.z, mL, mA, mB) <= n2max;
    if (cin) __hip_atomic_store(ws.pred + ((int64_t)b * h + r) * w + c, (ws.ep << 4) | (p0 ? 1u : 0u) | (p1 ? 2u : 0u) | (p2 ? 4u : 0u) | (p3 ? 8u : 0u),
                                BXI_RLX, BXI_AGENT);     // the evaluation's tag = "evaluated"; written through (sc1), read past the caches
    const int vrow = vc.vrow[b], vcol = vc.vcol[b];
    const bool v00 = r < vrow && c < vcol, v0n = r < vrow && cn < vcol, vD0 = r + D < vrow && c < vcol, vDn = r + D < vrow && cn < vcol;
    // what a box containing the site adds:  (r, c)  (r, c+D)  (r+D, c)  (r+D, c+D)
    const int s00 = (p0 && v0n) + (p2 && vD0) + (p3 && vDn), s0n = (p0 && v00) + (p1 && vD0), sD0 = (p1 && v0n) + (p2 && v00), sDn = (p3 && v00) ? 1 : 0;
    int cnt = 0;
    // LANE = RECTANGLE: what rectangle [r0, r1) x [c0, c1) collects from this row segment is a sum of the four site values over a RANGE of lanes
    //   rows r:      s00 over lanes [c0 - base, c1 - base)  +  s0n over lanes [c0 - D - base, c1 - D - base)        (base = the segment's first column)
    //   rows r + D:  sD0 over the first range               +  sDn over the second
    // so ONE prefix sum over the lanes -- the four values packed into the bytes of a word: a segment's sums are <= 192, 128, 128, 64, no byte
    // carries -- and four crossbar reads per lane serve 64 rectangles at once.  ~60 instructions per 64 rectangles where rounds 2-6 walked the
    // rectangles that reach the row one after the other (ballot, readlane, four range tests: ~22 instructions each -- a handful at 32 instances, 20-30
    // of an image's 64 at 128 instances, where the predicate workgroups hold the slots the tile workgroups are waiting for).  The same integers,
    // added in another order.  profiles/NOTES.md R6-9
    const uint32_t incl = wave_scan_incl_u32((uint32_t)s00 | ((uint32_t)s0n << 8) | ((uint32_t)sD0 << 16) | ((uint32_t)sDn << 24));
    const int base = seg * 64;
    for (int m0 = 0; m0 < n_ent; m0 += 64) {
        if (m0 == 64 && !ONE) rect = rect1;
        else if (m0) {
            if (PER_BOX) rect = m0 + lane < n_ent ? ws.boxtab()[m0 + lane] : make_int4(-1, 0, 0, 0);
            else if (!tab_entry<ONE>(ws, m0 + lane, m0 + lane < n_ent, spin_limit, rect)) { ok = false; return 0; }
            if (m0 + lane >= n_ent) rect = make_int4(-1, 0, 0, 0);
        }
        const int r0 = rect.y & 0xffff, r1 = (int)((unsigned int)rect.y >> 16), c0 = rect.z & 0xffff, c1 = (int)((unsigned int)rect.z >> 16);
        const bool mine = m0 + lane < n_ent && (int)((unsigned int)rect.x >> 24) == b;
        const bool rr = mine && r >= r0 && r < r1, rD2 = mine && r + D >= r0 && r + D < r1;
        if (!__any(rr || rD2)) continue;                       // wave-uniform
        const int i0 = min(max(c0 - base, 0), 64), i1 = min(max(c1 - base, 0), 64);
        const int j0 = min(max(c0 - D - base, 0), 64), j1 = min(max(c1 - D - base, 0), 64);
        // sum over the lanes below i (i in [0, 64]): the inclusive sum of lane i - 1
        const uint32_t ei0 = (uint32_t)__builtin_amdgcn_ds_bpermute(((i0 - 1) & 63) << 2, (int)incl), ei1 = (uint32_t)__builtin_amdgcn_ds_bpermute(((i1 - 1) & 63) << 2, (int)incl);
        const uint32_t ej0 = (uint32_t)__builtin_amdgcn_ds_bpermute(((j0 - 1) & 63) << 2, (int)incl), ej1 = (uint32_t)__builtin_amdgcn_ds_bpermute(((j1 - 1) & 63) << 2, (int)incl);
        const uint32_t X = (i1 > 0 ? ei1 : 0u) - (i0 > 0 ? ei0 : 0u), Y = (j1 > 0 ? ej1 : 0u) - (j0 > 0 ? ej0 : 0u);      // bytewise monotone: no borrows
        const int add = (rr ? (int)((X & 255u) + ((Y >> 8) & 255u)) : 0) + (rD2 ? (int)(((X >> 16) & 255u) + (Y >> 24)) : 0);
        if (PER_BOX) { if (add) atomicAdd(&boxacc[m0 + lane], add); }       // LDS; flushed once per workgroup (targets_pred_kernel)
        else cnt += add;
    }
    return cnt;
}

// sum W, once every pooled row segment has been evaluated: ONE word for the (hundreds of) askers; the reducer -- one wave of the
// finisher workgroup -- watches the 64 count words and publishes it.
__device__ __forceinline__ bool counts_complete(const Ws& ws, int n_items, double* total, bool* fault) {
    (void)n_items;
    const unsigned long long x = __hip_atomic_load(ws.sumw, BXI_RLX, BXI_AGENT);
    *total = (double)(x & (kSumwFault - 1ull));                         // exact: an integer far below 2^53
    if ((x >> 63) != 0ull && (x & kSumwFault)) *fault = true;
    return (x >> 63) != 0ull;
}
__device__ __forceinline__ bool reduce_counts(const Ws& ws, int n_items, int spin_limit) {
    for (int spins = 0; spins <= spin_limit; ++spins) {
        const unsigned long long x = __hip_atomic_load(&ws.acc1[(size_t)(threadIdx.x & 63) * kAcc2Stride], BXI_RLX, BXI_AGENT);
        const int arrived = wave_total_i32((int)(x >> 40));
        const double tot = wave_total_f64((double)(x & (kCountFault - 1ull)));        // exact
        const bool flt = __any((x & kCountFault) != 0ull);
        if (arrived == n_items) {
            if ((threadIdx.x & 63) == 0)
                __hip_atomic_store(ws.sumw, (1ull << 63) | (flt ? kSumwFault : 0ull) | (unsigned long long)tot, BXI_RLX, BXI_AGENT);
            BXI_WL(3, spins);
            return true;
        }
    }
    return false;
}
// thresh <= 0: every pair (padded ones too) weighs 1 (:1324), sum W = 8 x the box areas; no predicate waves then
__device__ __forceinline__ double total_weight_all_pairs(const InstArgs& a, const Ws& ws) {
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int m0 = 0; m0 < a.N; m0 += 64) {
        const int m = m0 + lane;
        if (m < a.N) {
            const int4 e = ws.tab[m];
            const int r0 = e.y & 0xffff, r1 = (int)((unsigned int)e.y >> 16), c0 = e.z & 0xffff, c1 = (int)((unsigned int)e.z >> 16);
            s += 8.0 * (double)((r1 - r0) * (int64_t)(c1 - c0));
        }
    }
    return wave_total_f64(s);
}

// ---- tile wave (wave64, no LDS, no barrier) ------------------------------------------------------------------------------
// Every UNORDERED pair is evaluated once and feeds both of its pixels: f(p,q) = f(q,p), the two weights W[k,p] + W[7-k,q]
// share the colour predicate.  Per pixel (a, b) = (sigmoid(x), sigmoid(-x)), t = a - b, u = a b.  Per pair (p, q):
//   S = a_p a_q + b_p b_q ; pw = -log S ; d pw / d x_p = -t_q u_p / S ; d pw / d x_q = -t_p u_q / S      (pairwise.cu:38-61)
// S cannot underflow while every |x| <= 34; tiles with a larger logit take the log-space path.
// Its waits: the predicate bytes of its own pixels (bit 7 set), when the logits have arrived and the per-pixel quantities are
// computed; and, before the gradient goes out, sum W (the global normaliser, :1327-1328) = every predicate wave's arrival.  The
// predicate waves precede the tile waves in the grid and never wait; by the time a tile wave asks they are normally done.
// A tile wave's own few predicate words (written through by the predicate waves, which precede it in the grid), read past the
// caches until every one carries this evaluation's tag; usually they are there at once.
// BATCH: the words in one asm statement (one round trip).  Not in the short single-launch kernels (eval1_kernel<D, 4, *>): there a tile wave is resident
// before the predicate waves start and polls anyway -- the trips hide in that wait (17.15 us per evaluation either way, R6-11) --, and the statement's
// twelve early-clobber outputs leave the kernel with a 36-byte private segment that nothing ever touches.
template <int D, int R, bool BATCH>
__device__ __forceinline__ bool pred_words(const Ws& ws, const Tile& t, int h, int w, int c, int spin_limit, uint32_t (&pbyte)[R + D]) {
    const unsigned int* pp = ws.pred + (int64_t)t.img * h * w;            // scalar base + 32-bit byte offsets (one plane < 2^31 bytes)
    const uint32_t cc = (uint32_t)min(max(c, 0), w - 1);
    const unsigned int want = ws.pred_any ? 0u : ws.ep;       // words an earlier launch left (bxi_boxinst_targets_f32) carry tag 0: no tag of this evaluation
    uint32_t off[R + D];
#pragma unroll
    for (int i = 0; i < R + D; ++i) off[i] = ((uint32_t)min(max(t.tile_r0 - D + i, 0), h - 1) * (uint32_t)w + cc) * 4u;
    bool ok = false;
    for (int spins = 0; spins <= spin_limit; ++spins) {
        bool all = true;
        if constexpr (BATCH) {
            load_words_past<R + D>(pp, off, pbyte);
#pragma unroll
            for (int i = 0; i < R + D; ++i) all = all && (pbyte[i] >> 4) == want;
        } else {
#pragma unroll
            for (int i = 0; i < R + D; ++i) {
                pbyte[i] = __hip_atomic_load(pp + off[i] / 4u, BXI_RLX, BXI_AGENT);
                all = all && (pbyte[i] >> 4) == want;
            }
        }
        if (__all(all)) { ok = true; BXI_WL(4, spins); break; }
        if (ws.pred_any) break;        // targets ready: the words are an EARLIER launch's -- what is not there now will not come (foreign or overwritten targets: loud at once, not after kSpinLimit polls)
        __builtin_amdgcn_s_sleep(BXI_SLEEP_WORDS);
    }
    return ok;        // false: the caller's arrival says so, and the finisher turns both losses into NaN
}

template <int D, int R, bool ONE>
__device__ __forceinline__ void math_tile(const InstArgs& a, const Ws& ws, const Tile& t, float upw_warm, float n2max, int zero_bit, int n_items,
                                          int spin_limit, float& scale, bool& have_scale, float* __restrict__ g_logits, float* gbuf /* LDS [R + 1][64] of this wave */,
                                          int tix, long long& fx_sum, bool& bad_out) {
    constexpr int RD = TG<D, R>::RD;
    const int lane = threadIdx.x & 63;
    const int h = a.h, w = a.w, n = t.n;
    const int64_t P = (int64_t)h * w;
    const float* Lg = a.logits + (int64_t)n * P;
    const int c = t.tile_c0 - D + lane;
    const bool col_owned = g_logits && lane >= D && lane < 64 - D && c < t.hc1;
    float x[RD];
    load_plane<D, R>(Lg, t, h, w, lane, x);
    // targets ready: the predicate words are an earlier launch's -- asked for WITH the logits (one round trip instead of two; the per-pixel
    // arithmetic below runs while they fly), looked at once where the other forms start polling
    // (Measured and dropped, R6-13: the same early look in the two-launch form, where most tile waves get their slots behind the predicate workgroups --
    // 31.2 vs 30.6 us at 128 instances, 23.4 vs 22.8 with 4-row tiles at 64: the waves that come too early pay ten wasted loads and poll anyway.)
    uint32_t pearly[R + D];
    const bool early = ws.pred_any != 0u && zero_bit == 0;             // wave-uniform
    if (early) {
        const unsigned int* pp = ws.pred + (int64_t)t.img * h * w;
        const uint32_t cc = (uint32_t)min(max(c, 0), w - 1);
#pragma unroll
        for (int i = 0; i < R + D; ++i) pearly[i] = __hip_atomic_load(pp + (uint32_t)min(max(t.tile_r0 - D + i, 0), h - 1) * (uint32_t)w + cc, BXI_RLX, BXI_AGENT);
    }
    float g[R];
    float num = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) g[j] = 0.f;
    // per-pixel quantities of this lane and of the lane D to its right: before the wait, they need the logits only
#ifndef BXI_PK_PAIRS
#define BXI_PK_PAIRS 1
#endif
    // PK (even dilation): rows 2k, 2k + 1 ride in the two halves of packed FP32 instructions (v_pk_mul / v_pk_fma: two pairs per instruction;
    // the conversions and the two transcendentals per pair stay scalar).  Every row's gradient receives the same terms in the same order.
    // 122 -> 106 registers at <2, 4>, 159 -> 138 at <2, 8>; 17.43 -> 17.07 us per evaluation at 32 instances (same box).  (The 8-row role
    // fits 113 registers when t and u are made again per pair -- four workgroups per CU --: slower, and the targets-ready long form then
    // stalled for seconds at 128 instances with every slot of the device taken from the start: profiles/NOTES.md R5-7.  Not built.)
    constexpr bool PK = BXI_PK_PAIRS && D % 2 == 0 && RD % 2 == 0;
    typedef float v2 __attribute__((ext_vector_type(2)));
    float pa_[PK ? 1 : RD], pb_[PK ? 1 : RD], pt_[PK ? 1 : RD], pu_[PK ? 1 : RD], aR[PK ? 1 : RD], bR[PK ? 1 : RD], tR[PK ? 1 : RD], uR[PK ? 1 : RD];
    v2 pa2[PK ? RD / 2 : 1], pb2[PK ? RD / 2 : 1], pt2[PK ? RD / 2 : 1], pu2[PK ? RD / 2 : 1], aR2[PK ? RD / 2 : 1], bR2[PK ? RD / 2 : 1],
        tR2[PK ? RD / 2 : 1], uR2[PK ? RD / 2 : 1];
    bool sat = false;
    if constexpr (PK) {
#pragma unroll
        for (int k = 0; k < RD / 2; ++k) {
            sat |= !(fabsf(x[2 * k]) <= 34.f) || !(fabsf(x[2 * k + 1]) <= 34.f);
            const float2 s0 = sig_pair(x[2 * k]), s1 = sig_pair(x[2 * k + 1]);
            pa2[k] = v2{s0.x, s1.x}; pb2[k] = v2{s0.y, s1.y};
            aR2[k] = v2{lane_plus<D>(s0.x), lane_plus<D>(s1.x)}; bR2[k] = v2{lane_plus<D>(s0.y), lane_plus<D>(s1.y)};
            pt2[k] = pa2[k] - pb2[k]; pu2[k] = pa2[k] * pb2[k]; tR2[k] = aR2[k] - bR2[k]; uR2[k] = aR2[k] * bR2[k];
        }
    } else {
#pragma unroll
    for (int j = 0; j < RD; ++j) {
        sat |= !(fabsf(x[j]) <= 34.f);
        const float2 s = sig_pair(x[j]); pa_[j] = s.x; pb_[j] = s.y; pt_[j] = s.x - s.y; pu_[j] = s.x * s.y;
        aR[j] = lane_plus<D>(pa_[j]); bR[j] = lane_plus<D>(pb_[j]); tR[j] = aR[j] - bR[j]; uR[j] = aR[j] * bR[j];
    }
    }
    const bool slow = zero_bit != 0 || __any(sat);
    bool bad = false;          // a bounded wait of this wave ran out (never expected): its arrival carries the fact to the finisher
    const int band0 = t.tile_r0 / kSBlk, band1 = (min(t.tile_r0 + R, h) - 1) / kSBlk;
    const bool look_early = !slow && (!have_scale || (ONE && g_logits));       // wave-uniform
    unsigned long long sw_early = 0ull;
    unsigned int f0e = 0u, f1e = 0u;
    BXI_TW(1, tix, 2);
    if (!slow) {
        uint32_t pbyte[R + D];
        if (BXI_AB(8)) { for (int i = 0; i < R + D; ++i) pbyte[i] = 0xfu; }
        else if (early) {
            bool all = true;
#pragma unroll
            for (int i = 0; i < R + D; ++i) { pbyte[i] = pearly[i]; all = all && (pbyte[i] >> 4) == 0u; }      // (words an earlier launch left carry tag 0)
            bad |= !__all(all);
        } else bad |= !pred_words<D, R, (!ONE || R == 8)>(ws, t, h, w, c, spin_limit, pbyte);
        float gq[PK ? 1 : RD], gR[PK ? 1 : RD];      // gradient of this lane's pixels / of lane + D's
        v2 gq2[PK ? RD / 2 : 1], gR2[PK ? RD / 2 : 1];
        if constexpr (PK) {
#pragma unroll
            for (int k = 0; k < RD / 2; ++k) { gq2[k] = v2{0.f, 0.f}; gR2[k] = v2{0.f, 0.f}; }
        } else {
#pragma unroll
            for (int j = 0; j < RD; ++j) { gq[j] = 0.f; gR[j] = 0.f; }
        }
        // pair weights as bytes, four rows per word: cw = W[k,A] + W[7-k,B] (gradient), dw = the same restricted to
        // pixels this tile owns (loss sum)
        constexpr int NQ = (R + D + 3) / 4;
        uint32_t cw[4][NQ], dw[4][NQ];
        // Every mask of dir_masks is (a 0/1 of the LANE: its column in the box / valid / owned) x (a row range of the TILE: wave-uniform) x (the colour
        // predicate), so the weights are made in the byte domain at once: the predicate nibbles of four rows packed into a word (bytes = rows), one
        // shift + AND per direction, an AND with the row range's byte mask (scalar registers, made on the scalar unit) and a packed 16-bit multiply by the
        // lane's 0 / 1 / 2.  ~160 vector instructions per tile where the bit-mask form (transpose to row bits, AND the flag words, spread nibble by
        // nibble: rounds 3-6, ~300) stood next to ~500 of the pair loop itself.  The same bytes.  profiles/NOTES.md R6-11
        {
            uint32_t spb[4][NQ];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) {
                uint32_t W = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * q4 + k < R + D) W |= (pbyte[4 * q4 + k] & 15u) << (8 * k);
#pragma unroll
                for (int d = 0; d < 4; ++d) spb[d][q4] = (W >> d) & 0x01010101u;
            }
            const int base = t.tile_r0 - D;
            const uint32_t rows_box = row_bits(t.r0, t.r1, base, RD), rows_val = row_bits(0, min(h, t.vrow), base, RD);
            const uint32_t rows_own = row_bits(t.tile_r0, min(t.tile_r0 + R, h), base, RD);
            const uint32_t X1 = rows_box & rows_val, X2 = (rows_box >> D) & rows_val, X3 = rows_box & (rows_val >> D);
            const uint32_t X1O = X1 & rows_own, X2OD = X2 & (rows_own >> D), X3O = X3 & rows_own;
            const int cl = t.tile_c0 - D + lane, cr = cl + D, cv = min(w, t.vcol);
            const bool inR = lane + D < 64;          // lanes without a right neighbour: every pair weight 0 (they receive some other lane's data)
            const uint32_t fa = cl >= t.c0 && cl < t.c1, fv = cl >= 0 && cl < cv, fo = lane >= D && lane < 64 - D && cl < t.hc1;
            const uint32_t faR = inR && cr >= t.c0 && cr < t.c1, fvR = inR && cr >= 0 && cr < cv, foR = inR && lane < 64 - 2 * D && cr < t.hc1;
            // the lane's multipliers, one per 16-bit half
            const uint32_t p1 = fa & fvR, q1 = faR & fv, s1 = fa & fv;
            const uint32_t kp = p1 * 0x10001u, kq = q1 * 0x10001u, ks = s1 * 0x10001u, kpq = kp + kq;
            const uint32_t kpo = (p1 & fo) * 0x10001u, kqo = (q1 & foR) * 0x10001u, kso = (s1 & fo) * 0x10001u, kpqo = kpo + kqo;
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) {
                const uint32_t b1 = spread4((X1 >> (4 * q4)) & 15u), b2 = spread4((X2 >> (4 * q4)) & 15u), b3 = spread4((X3 >> (4 * q4)) & 15u);
                const uint32_t b1o = spread4((X1O >> (4 * q4)) & 15u), b2o = spread4((X2OD >> (4 * q4)) & 15u), b3o = spread4((X3O >> (4 * q4)) & 15u);
                cw[0][q4] = pk_mul_u16(spb[0][q4] & b1, kpq);
                cw[1][q4] = pk_mad_u16(spb[1][q4] & b2, kp, pk_mul_u16(spb[1][q4] & b3, kq));
                cw[2][q4] = pk_mul_u16((spb[2][q4] & b3) + (spb[2][q4] & b2), ks);
                cw[3][q4] = pk_mad_u16(spb[3][q4] & b3, kp, pk_mul_u16(spb[3][q4] & b2, kq));
                dw[0][q4] = pk_mul_u16(spb[0][q4] & b1o, kpqo);
                dw[1][q4] = pk_mad_u16(spb[1][q4] & b2o, kpo, pk_mul_u16(spb[1][q4] & b3o, kqo));
                dw[2][q4] = pk_mul_u16((spb[2][q4] & b3o) + (spb[2][q4] & b2o), kso);
                dw[3][q4] = pk_mad_u16(spb[3][q4] & b3o, kpo, pk_mul_u16(spb[3][q4] & b2o, kqo));
            }
        }
        BXI_TW(1, tix, 3);
        // the first look at sum W (and, single-launch form, at the band flags of the rows this tile adds onto) goes out BEFORE the pair loop and is
        // evaluated behind it: the round trip hides under ~2 us of arithmetic; what is not there yet is polled for as before
        if (look_early) {
            if (!have_scale) sw_early = __hip_atomic_load(ws.sumw, BXI_RLX, BXI_AGENT);
            if (ONE && g_logits) {
                f0e = __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + band0], BXI_RLX, BXI_AGENT);
                f1e = __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + band1], BXI_RLX, BXI_AGENT);
            }
        }
        // one unordered pair: A = (row ra, this lane) ; B = (row rb of the lane `q` names) ; num collects -log2 S
#define BXI_PAIR(i, ra, rb, qa, qb, qt, qu, dir, GA, GB)                                                            \
        {                                                                                                           \
            const float gw = (float)((cw[dir][(i) >> 2] >> (8 * ((i) & 3))) & 255u);                                \
            const float nw = (float)((dw[dir][(i) >> 2] >> (8 * ((i) & 3))) & 255u);                                \
            const float S = pa_[ra] * qa[rb] + pb_[ra] * qb[rb];                    /* P(y_A == y_B) */            \
            num -= nw * __builtin_amdgcn_logf(S);                                   /* v_log_f32 = log2 */         \
            const float mm = gw * __builtin_amdgcn_rcpf(S);                                                         \
            GA -= mm * qt[rb] * pu_[ra];                                                                            \
            GB -= mm * pt_[ra] * qu[rb];                                                                            \
        }
        if constexpr (PK) {
            v2 num2 = {0.f, 0.f};
            // two unordered pairs: rows (2 ka, 2 ka + 1) of this lane against rows (2 kb, 2 kb + 1) of the lane `q` names; bytes 2 ip, 2 ip + 1 of the weights
#define BXI_PAIR2(ip, ka, kb, qa, qb, qt, qu, dir, GA, GB)                                                          \
            {                                                                                                       \
                const uint32_t cwd = cw[dir][(2 * (ip)) >> 2] >> (8 * ((2 * (ip)) & 3));                              \
                const uint32_t dwd = dw[dir][(2 * (ip)) >> 2] >> (8 * ((2 * (ip)) & 3));                              \
                const v2 gw = {(float)(cwd & 255u), (float)((cwd >> 8) & 255u)};                                      \
                const v2 nw = {(float)(dwd & 255u), (float)((dwd >> 8) & 255u)};                                      \
                const v2 S = pa2[ka] * qa[kb] + pb2[ka] * qb[kb];                                                     \
                const v2 lg = {__builtin_amdgcn_logf(S.x), __builtin_amdgcn_logf(S.y)};                               \
                num2 -= nw * lg;                                                                                      \
                const v2 rc = {__builtin_amdgcn_rcpf(S.x), __builtin_amdgcn_rcpf(S.y)};                               \
                const v2 mm = gw * rc;                                                                                \
                GA -= mm * qt[kb] * pu2[ka];                                                                          \
                GB -= mm * pt2[ka] * qu[kb];                                                                          \
            }
#pragma unroll
            for (int ip = 0; ip < (R + D) / 2; ++ip) {
                const int jp = ip + D / 2;
                if (2 * ip >= D) BXI_PAIR2(ip, ip, ip, aR2, bR2, tR2, uR2, 0, gq2[ip], gR2[ip])
                BXI_PAIR2(ip, jp, ip, aR2, bR2, tR2, uR2, 1, gq2[jp], gR2[ip])
                BXI_PAIR2(ip, ip, jp, pa2, pb2, pt2, pu2, 2, gq2[ip], gq2[jp])
                BXI_PAIR2(ip, ip, jp, aR2, bR2, tR2, uR2, 3, gq2[ip], gR2[jp])
                if (2 * ip >= D) {
                    const float fromL0 = lane_minus<D>(gR2[ip].x), fromL1 = lane_minus<D>(gR2[ip].y);
                    g[2 * ip - D] = gq2[ip].x + (lane >= D ? fromL0 : 0.f);
                    g[2 * ip + 1 - D] = gq2[ip].y + (lane >= D ? fromL1 : 0.f);
                }
            }
#undef BXI_PAIR2
            num = num2.x + num2.y;
        } else {
#pragma unroll
        for (int i = 0; i < R + D; ++i) {
            if (BXI_AB(2)) break;
            const int j = i + D;
            if (i >= D) BXI_PAIR(i, i, i, aR, bR, tR, uR, 0, gq[i], gR[i])
            BXI_PAIR(i, j, i, aR, bR, tR, uR, 1, gq[j], gR[i])
            BXI_PAIR(i, i, j, pa_, pb_, pt_, pu_, 2, gq[i], gq[j])
            BXI_PAIR(i, i, j, aR, bR, tR, uR, 3, gq[i], gR[j])
            if (i >= D) {     // row i is complete: collect what the lane D to the left computed for it
                const float fromL = lane_minus<D>(gR[i]);
                g[i - D] = gq[i] + (lane >= D ? fromL : 0.f);
            }
        }
        }
        num *= 0.69314718055994531f;
#undef BXI_PAIR
    } else {         // wave-uniform; rare
        if (ONE) {   // single-launch form: the tile's predicate words vouch for the Lab pixels the log-space path reads
            uint32_t pbyte[R + D];
            bad |= !pred_words<D, R, (!ONE || R == 8)>(ws, t, h, w, c, spin_limit, pbyte);
        }
        slow_tile<D, R, ONE>(Lg, ws.lab4, t, n2max, zero_bit, h, w, lane, gbuf);
        num = gbuf[R * 64 + lane];
#pragma unroll
        for (int j = 0; j < R; ++j) g[j] = gbuf[j * 64 + lane];
    }
    BXI_TW(1, tix, 5);
    num = wave_total_f32(num);
    fx_sum += (long long)(num * kNumScale);                                    // this tile's share of sum W pw, fixed point: integer adds commute
    // single-launch form: the rows this tile adds onto were zero-filled by stream workgroups of THIS launch; their band flags are
    // asked for in the same round as sum W
    bool bands_ok = !ONE || !g_logits;
    if (look_early) {
        if (!have_scale && (sw_early >> 63) != 0ull) {
            if (sw_early & kSumwFault) bad = true;
            have_scale = true;
            scale = upw_warm / fmaxf((float)(double)(sw_early & (kSumwFault - 1ull)), 1.f);
        }
        if (ONE && g_logits) bands_ok = f0e == ws.ep && f1e == ws.ep;
    }
    if (BXI_AB(4)) { bands_ok = true; if (!have_scale) { have_scale = true; scale = 1e-6f; } }
    if (!have_scale || !bands_ok) {           // wave-uniform
        double total_w = 0.0;
        bool ok = false;
        for (int spins = 0; spins <= spin_limit; ++spins) {
            unsigned int f0 = ws.ep, f1 = ws.ep;
            if (!bands_ok) {
                f0 = __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + band0], BXI_RLX, BXI_AGENT);
                f1 = __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + band1], BXI_RLX, BXI_AGENT);
            }
            if (!have_scale) {
                if (zero_bit) { total_w = total_weight_all_pairs(a, ws); have_scale = true; }
                else have_scale = counts_complete(ws, n_items, &total_w, &bad);
                if (have_scale) scale = upw_warm / fmaxf((float)total_w, 1.f);
            }
            bands_ok = f0 == ws.ep && f1 == ws.ep;
            if (have_scale && bands_ok) { ok = true; BXI_WL(5, spins); break; }
            __builtin_amdgcn_s_sleep(BXI_SLEEP_SUMW);
        }
        bad |= !ok;
        have_scale = true;
    }
    BXI_TW(1, tix, 4);
    if (g_logits) {
        char* G = reinterpret_cast<char*>(g_logits + (int64_t)n * P);      // scalar base + 32-bit byte offset
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = t.tile_r0 + j;
            if (col_owned && r < h) add_f32(reinterpret_cast<float*>(G + (uint32_t)(r * w + c) * 4u), g[j] * scale);
        }
    }
    BXI_TW(1, tix, 6);
    bad_out |= bad;
}

// A tile wave's ONE arrival, with or without tiles: its share of sum W pw (+ 1.0: keeps the packed field non-negative -- S may exceed 1 by
// a rounding) and whether one of its bounded waits ran out, as one atomic without return on one of the N x 8 arrival words (each in its
// own 128 bytes).  The finisher counts WAVES, so its last act -- advancing the workspace's epoch -- comes after every tile wave of
// the launch has read the epoch (an idle wave that started late could otherwise draw the NEXT evaluation's tag and wait for nobody).
// WHICH word: one of an instance whose table entry this wave has SEEN tagged -- the table wave of instances 64 k .. 64 k + 63 zeroes their arrival
// words and drains before it writes their entries, and a tile wave checks entries 0 .. 63 and N only (tile_role).  Rounds 3-5 spread the arrivals
// over all N x 8 words: in the single-launch forms a tile wave could then arrive on a word of instances 64 .. N - 1 that the SECOND table wave --
// draining its written-through zeroes under the logit stream's traffic -- had not zeroed yet; the zero wiped the arrival, the finisher never saw its
// count, ran out (status 2, NaN losses for that evaluation) after kSpinLimit polls = 4.1 s.  Seen five times in 4800 evaluations with the 8-row
// kernels at four workgroups per CU and 128 instances, where the tile workgroups start just as the stream workgroups' traffic lets the table's
// drains complete (profiles/NOTES.md R5-7, R6-3: the stall's length follows kSpinLimit, the wait that runs out is the finisher's).
// A wave whose bounded wait ran out (or that saw a fault word) says so on the evaluation's fault word BEFORE it arrives -- a returning atomic, waited
// for -- so that the round in which the finisher sees the last arrival sees the fault too.  (Rounds 3-5 added a flag bit to the arrival itself: with
// six arrivals per word four faults carry into the arrival count, the finisher never sees the count it waits for and the -- already loud -- error path
// takes kSpinLimit polls: 4.5 s for an evaluation whose targets are somebody else's.)
__device__ __forceinline__ void tile_wave_arrives(const Ws& ws, int N, int wid, long long fx_sum, bool bad) {
    if ((threadIdx.x & 63) == 0) {
        if (bad) {
            const unsigned int seen = __hip_atomic_fetch_or(ws.fault, kFaultCounts, BXI_RLX, BXI_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
        }
        __hip_atomic_fetch_add(ws.acc2 + (size_t)(wid % ((N < 64 ? N : 64) * kAcc2Split)) * kAcc2Stride,
                               (1ull << 52) + (unsigned long long)(fx_sum + (1ll << 24)), BXI_RLX, BXI_AGENT);
    }
}

__device__ __forceinline__ void block_sum4(float (&v)[4], float* red /*[16]*/) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_total_f32(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[(threadIdx.x >> 6) * 4 + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
}
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// ---- leader workgroup (one per instance) -------------------------------------------------------------------------------
//   partial maxima -> maxima -> sigmoid on those only -> both dice terms (:117-143) -> unit projection gradients, recorded as
//   one 8-byte word per column / row (gradient bits << 32 | arg-max index) for bxi_boxinst_grad_rescale_f32 and ADDED to the
//   gradient at the arg-max positions.  Nobody in this launch reads what a leader writes except the finisher (its dice loss).
template <bool ONE>
__device__ __forceinline__ void leader_block(const InstArgs& a, int dil, Ws ws /* .ep == 0: the first poll fetches the tag */, const LossState& st, int n, float upp,
                                             float* __restrict__ g_logits, unsigned char* smem, float* red, int spin_limit) {
    const int h = a.h, w = a.w, tid = threadIdx.x;
    float* xs = reinterpret_cast<float*>(smem);   // [w] sigmoid of the column maxima, then their unit gradients
    float* ys = xs + w;                           // [h]
    int* carg = reinterpret_cast<int*>(ys + h);   // [w]
    int* rarg = carg + w;                         // [h]
    int4 e;
    bool waited;
    if (ONE) {
        // single-launch form: the instance's partial maxima and the zero-fill of its map come from stream workgroups of THIS
        // launch (earlier in the grid, waiting for nobody); one flag per band says they are in memory.  The table entry and the
        // (first 64) band flags are asked for in ONE round trip -- both are self-announcing words, and by the time a leader gets a
        // slot both are normally there; the partial maxima themselves are only asked for once their flags have been seen.
        const int lane = tid & 63;
        waited = false;
        for (int spins = 0; spins <= spin_limit; ++spins) {
            unsigned int f = lane < ws.n_cb ? __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + lane], BXI_RLX, BXI_AGENT) : 0u;
            u4v v;                                                  // (its wait covers the flag load issued before it)
            if (ws.ep == 0u) {                                      // wave-uniform: the first poll brings the evaluation's tag along (with_tag)
                unsigned int ew;
                v = load16_past_epoch(ws.tab + n, ws.epoch, ew);
                ws.ep = next_tag((unsigned int)__builtin_amdgcn_readfirstlane((int)ew));
            } else
                v = load16_past(ws.tab + n);
            if (lane >= ws.n_cb) f = ws.ep;
            if (__all(f == ws.ep && v.w == ws.ep)) { e = make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w); waited = true; BXI_WL(6, spins); break; }
            __builtin_amdgcn_s_sleep(BXI_SLEEP_LEAD);
        }
        for (int b0 = 64; b0 < ws.n_cb && waited; b0 += 64) {
            bool got = false;
            for (int spins = 0; spins <= spin_limit; ++spins) {
                const unsigned int f = b0 + lane < ws.n_cb ? __hip_atomic_load(&ws.bandflag[(int64_t)n * ws.n_cb + b0 + lane], BXI_RLX, BXI_AGENT) : ws.ep;
                if (__all(f == ws.ep)) { got = true; BXI_WL(7, spins); break; }
                __builtin_amdgcn_s_sleep(BXI_SLEEP_LEAD);
            }
            waited = got;
        }
        if (!waited) {        // loud; nothing is computed from partial maxima that may be stale (their indices address the gradient)
            if (tid == 0) __hip_atomic_store(&ws.dice[n], (1ull << 32) | kDiceFault, BXI_RLX, BXI_AGENT);
            return;
        }
    } else {
        waited = tab_entry<false>(ws, n, true, spin_limit, e);
    }
    const int br0 = e.y & 0xffff, br1 = (int)((unsigned int)e.y >> 16), bc0 = e.z & 0xffff, bc1 = (int)((unsigned int)e.z >> 16);
    const bool any = br1 > br0 && bc1 > bc0;
    (void)dil;
    float sums[4] = {0.f, 0.f, 0.f, 0.f};   // I_x, U_x, I_y, U_y
    // the partial maxima of a column / row: up to eight loads in flight at once
    auto best_key = [](const unsigned long long* __restrict__ part, int n_part, int64_t stride) {
        unsigned long long k = 0ull;
        for (int s0 = 0; s0 < n_part; s0 += 8) {
            unsigned long long o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                o[u] = ONE ? __hip_atomic_load(part + (int64_t)min(s0 + u, n_part - 1) * stride, BXI_RLX, BXI_AGENT) : part[(int64_t)min(s0 + u, n_part - 1) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) k = o[u] > k ? o[u] : k;
        }
        return k;
    };
    for (int i = tid; i < max(w, h); i += 256) {
        const bool is_c = i < w, is_r = i < h;
        // both keys requested before either is used
        const unsigned long long kc = best_key(ws.colpart + (int64_t)n * ws.n_cb * w + (is_c ? i : 0), ws.n_cb, w);
        const unsigned long long kr = best_key(ws.rowkey + (int64_t)n * ws.n_rp * h + (is_r ? i : 0), ws.n_rp, h);
        if (is_c) {
            const int c = i;
            const float X = sigmoid_acc(unpack_val(kc));
            const float TX = (any && c >= bc0 && c < bc1) ? 1.f : 0.f;
            xs[c] = X; carg[c] = (int)unpack_idx(kc);
            sums[0] += X * TX; sums[1] += X * X + TX * TX;
        }
        if (is_r) {
            const int r = i;
            const float Y = sigmoid_acc(unpack_val(kr));
            const float TY = (any && r >= br0 && r < br1) ? 1.f : 0.f;
            ys[r] = Y; rarg[r] = (int)unpack_idx(kr);
            sums[2] += Y * TY; sums[3] += Y * Y + TY * TY;
        }
    }
    BXI_TW(3, 1 + n, 1);
    block_sum4(sums, red);
    const float Ix = sums[0], Ux = sums[1] + 1e-5f, Iy = sums[2], Uy = sums[3] + 1e-5f;
    if (tid == 0)   // :130, summed over both axes :143; the datum is its own flag
        __hip_atomic_store(&ws.dice[n], (1ull << 32) | (unsigned long long)__float_as_uint((1.f - 2.f * Ix / Ux) + (1.f - 2.f * Iy / Uy)),
                           BXI_RLX, BXI_AGENT);
    BXI_TW(3, 1 + n, 2);
    if (g_logits) {
        // dice = 1 - 2I/U ; d dice/d u_j = (-2 t_j U + 4 I u_j) / U^2 ; chain through sigmoid ; mean over N
        const float invN = 1.f / (float)a.N;
        for (int c = tid; c < w; c += 256) {
            const float X = xs[c];
            const float TX = (any && c >= bc0 && c < bc1) ? 1.f : 0.f;
            const float gv = invN * ((-2.f * TX * Ux + 4.f * Ix * X) / (Ux * Ux)) * X * (1.f - X);
            xs[c] = gv;
            st.colk[(int64_t)n * w + c] = ((unsigned long long)__float_as_uint(gv) << 32) | (unsigned int)carg[c];
        }
        for (int r = tid; r < h; r += 256) {
            const float Y = ys[r];
            const float TY = (any && r >= br0 && r < br1) ? 1.f : 0.f;
            const float gv = invN * ((-2.f * TY * Uy + 4.f * Iy * Y) / (Uy * Uy)) * Y * (1.f - Y);
            ys[r] = gv;
            st.rowk[(int64_t)n * h + r] = ((unsigned long long)__float_as_uint(gv) << 32) | (unsigned int)rarg[r];
        }
        lds_barrier();
        // xs / ys now hold the gradients for every thread (LDS only: the record stores above need not have landed)
        // one addition per arg-max position (a pixel that is its column's AND its row's arg-max gets their sum in one)
        float* G = g_logits + (int64_t)n * h * w;
        for (int c = tid; c < w; c += 256) {
            const int r = carg[c];
            float v = xs[c];
            if (rarg[r] == c) v += ys[r];
            add_f32(G + (int64_t)r * w + c, v * upp);
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg[r];
            if (carg[c] != r) add_f32(G + (int64_t)r * w + c, ys[r] * upp);
        }
    }
    BXI_TW(3, 1 + n, 3);
}

__device__ __forceinline__ Tile tile_of(const int4& e, const ValidCells& vc, int D, int R, int TW, int n, int idx, int h, int w) {   // e: the instance's table entry (uniform)
    Tile t;
    t.r0 = e.y & 0xffff; t.r1 = (int)((unsigned int)e.y >> 16); t.c0 = e.z & 0xffff; t.c1 = (int)((unsigned int)e.z >> 16);
    t.img = (int)((unsigned int)e.x >> 24); t.n = n;
    t.vrow = vc.vrow[t.img]; t.vcol = vc.vcol[t.img];
    const int dr0 = max(t.r0 - D, 0), hc0 = max(t.c0 - D, 0);
    t.hc1 = min(t.c1 + D, w);
    const int ntc = (t.hc1 - hc0 + TW - 1) / TW;
    const int ti = idx / ntc, tj = idx - ti * ntc;
    t.tile_r0 = (dr0 / R + ti) * R;
    t.tile_c0 = hc0 + tj * TW;
    (void)h;
    return t;
}

// The finisher's rounds.  Leaders: the dice losses of instances [b0, b0 + 64) (self-flagging words).
__device__ __forceinline__ bool dice_round(const Ws& ws, int N, int b0, float* dsum, bool* fault) {
    const int lane = threadIdx.x & 63, i = b0 + lane;
    const unsigned long long dg = i < N ? __hip_atomic_load(&ws.dice[i], BXI_RLX, BXI_AGENT) : (1ull << 32);
    if (!__all((dg >> 32) != 0ull)) return false;
    if (__any((dg & kDiceFault) != 0ull)) *fault = true;
    const float dv = i < N ? __uint_as_float((unsigned int)dg) : 0.f;
    const int m = min(64, N - b0);
    for (int k = 0; k < m; ++k) *dsum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), k));   // index order: run-to-run identical
    return true;
}

// The tile of list position `ti`: the instance whose tile range holds it (table entries: 16 bytes per instance, the same lines
// for every wave), then the tile's place inside the instance's hull.  e0 = this lane's entry of the first 64 (N < 64: all).
template <int D, int R, bool ONE>
__device__ __forceinline__ bool locate_tile(const Ws& ws, const ValidCells& vc, int N, const int4& e0, const int4& e1, int ti, int h, int w, int spin_limit, Tile& out) {
    const int lane = threadIdx.x & 63;
    int n = 0;
    int4 e = make_int4(0, 0, 0, 0);
    if (N < 64) {
        const unsigned long long mask = __ballot(lane < N && (e0.x & 0xffffff) <= ti);
        n = __popcll(mask) - 1;
        e.x = __builtin_amdgcn_readlane(e0.x, n); e.y = __builtin_amdgcn_readlane(e0.y, n);
        e.z = __builtin_amdgcn_readlane(e0.z, n); e.w = __builtin_amdgcn_readlane(e0.w, n);
    } else {
        // (entries 0..63 and 64..127 came with the wave's first round trip -- tile_role --: up to 128 instances the search asks memory for nothing.
        //  Rounds 3-5 loaded chunk after chunk here, two dependent round trips in front of every tile of instances 64.. : the "locate" phase
        //  of a tile wave, 2.0 us at 128 instances)
        for (int m0 = 0; m0 < N; m0 += 64) {
            int4 em = m0 == 0 ? e0 : e1;
            if (m0 >= 128 && !tab_entry<ONE>(ws, m0 + lane, m0 + lane < N, spin_limit, em)) return false;
            const unsigned long long mask = __ballot(m0 + lane < N && (em.x & 0xffffff) <= ti);
            const int cntm = __popcll(mask);
            if (cntm == 0) break;
            n = m0 + cntm - 1;
            e.x = __builtin_amdgcn_readlane(em.x, cntm - 1); e.y = __builtin_amdgcn_readlane(em.y, cntm - 1);
            e.z = __builtin_amdgcn_readlane(em.z, cntm - 1); e.w = __builtin_amdgcn_readlane(em.w, cntm - 1);
            if (cntm < 64) break;
        }
    }
    out = tile_of(e, vc, D, R, TG<D, R>::TW, n, ti - (e.x & 0xffffff), h, w);
    return true;
}

// ---- the roles of the second launch (two-launch form) / of the back half of the single launch ----------------------------------
// predicate workgroup `pblk` of n_pb: 4 independent waves striding through the pooled row segments
template <bool ONE>
__device__ __forceinline__ void pred_role(const InstArgs& a, const ValidCells& vc, Ws ws /* .ep == 0: the first poll fetches the tag */, int D, float n2max, int pblk, int n_pb, int n_items, int spin_limit,
                                          bool high_prio = true) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int segs = (a.w + 63) >> 6, pid = pblk * kWaves + wave;
    BXI_TW(2, pid, 0);
    // short, and the tile waves will ask for these words -- except in the single launch WITHOUT the staying-on (37 .. 73 instances), where the predicate
    // waves mostly wait for Lab records and their priority only takes issue slots from the pool waves they wait for (64 instances: 21.2 -> 21.05 us, R6-27)
    if (high_prio) __builtin_amdgcn_s_setprio(3);
    int cnt = 0, segments = 0;
    bool ok = true;
    for (int item = pid; item < n_items && ok; item += n_pb * kWaves) { cnt += pred_item<ONE>(a.h, a.w, a.N, vc, ws, D, n2max, item, segs, spin_limit, ok); ++segments; }
    // the evaluation whose tag is the last one: the finisher zeroes the workspace behind it (the tag counter starts again), so every store of
    // this evaluation must have landed before its arrival can be seen (the predicate words otherwise announce themselves)
    if (ws.ep == kMaxTag) drain_vmem();
    cnt = wave_total_i32(cnt);
    // ONE arrival per workgroup: arrivals on one word are performed one after the other (~0.15 us each), and the tile waves
    // need the last one
    __shared__ int pred_cnt[kWaves], pred_seg[kWaves], pred_bad[kWaves];
    if (lane == 0) { pred_cnt[wave] = cnt; pred_seg[wave] = segments; pred_bad[wave] = ok ? 0 : 1; }
    // an LDS-only barrier: __syncthreads() would also wait for this wave's predicate-word stores to be acknowledged (~1 us) before the
    // count -- which the tile waves' normaliser hangs on -- could leave; the words announce themselves, nobody infers them from the count
    lds_barrier();
    if (threadIdx.x == 0) {  // (segments evaluated, sum W); integer adds commute: run-to-run identical
        if ((pred_bad[0] | pred_bad[1]) | (pred_bad[2] | pred_bad[3])) {      // loud: on the fault word, ahead of the arrival (a flag bit ADDED to the
            const unsigned int seen = __hip_atomic_fetch_or(ws.fault, kFaultCounts, BXI_RLX, BXI_AGENT);    // arrival carries into its count from the second fault on)
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
        }
        __hip_atomic_fetch_add(&ws.acc1[(size_t)(pblk & (kAcc1Words - 1)) * kAcc2Stride],
                               ((unsigned long long)(unsigned int)((pred_seg[0] + pred_seg[1]) + (pred_seg[2] + pred_seg[3])) << 40) |
                                   (unsigned long long)(unsigned int)((pred_cnt[0] + pred_cnt[1]) + (pred_cnt[2] + pred_cnt[3])),
                               BXI_RLX, BXI_AGENT);
    }
    BXI_TW(2, pid, 1);
}

// the reducer: ONE wave, in a workgroup of its own right behind the predicate workgroups -- EARLIER in the grid than every tile
// workgroup that waits for what it publishes (it used to be a wave of the finisher, the LAST workgroup: on a stream with fewer
// slots than tile workgroups the finisher could not start while the tile waves, holding every slot, waited for it).  It waits
// only for predicate workgroups.  (Single-launch form: the count words are this evaluation's only once the table says so --
// before that they hold the previous evaluation's complete counts.)
template <bool ONE>
__device__ __forceinline__ void reducer_role(const Ws& ws, int zero_bit, int n_items, int spin_limit) {
    if (threadIdx.x >= 64 || zero_bit || n_items <= 0) return;      // n_items <= 0: sum W is already published (an earlier launch, or the table wave)
    if (!(table_complete<ONE>(ws, 0, spin_limit) && reduce_counts(ws, n_items, spin_limit)) && threadIdx.x == 0) atomicOr(ws.fault, kFaultCounts);
}

// the last workgroup: waits only for workgroups that never wait for it -- the leaders and the predicate waves (done early), then
// the tile waves -- and writes the two loss values
template <bool ONE>
__device__ __forceinline__ void finisher_role(const InstArgs& a, const Ws& ws, const LossState& st, float upp, float upw, float warmup, int zero_bit, int n_items,
                                              int spin_limit, int R, int n_tile_waves, float* __restrict__ losses) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int N = a.N;
    // the launch ends on this workgroup's polls and its last few instructions: they go ahead of whatever else the CU holds (128 instances, two or
    // three tile waves on every SIMD: 30.5 -> 29.85 us; 24.9 -> 24.65 with the targets ready; nothing at 32 / 64.  The reducer and the leaders at a
    // higher priority: nothing.  profiles/NOTES.md R6-23)
    __builtin_amdgcn_s_setprio(3);
    BXI_TW(3, 0, 0);
    __shared__ double fin_d[kWaves];
    __shared__ int fin_i[kWaves];
    __shared__ float fin_f;
    __shared__ int fin_ok, fin_flt, fin_b[kWaves];
    bool ok = true, flt0 = false;
    double total_w = 0.0;
    float dsum = 0.f;
    int spins = 0;
    if (spin_limit < 0) ok = false;
    if (wave == 0) {
        if (!table_complete<ONE>(ws, N, spin_limit)) ok = false;      // every polled word of this evaluation is zeroed from here on
        for (int b0 = 0; b0 < N && ok; b0 += 64) {
            while (!dice_round(ws, N, b0, &dsum, &flt0)) {
                if (++spins > spin_limit) { ok = false; break; }
                __builtin_amdgcn_s_sleep(BXI_SLEEP_FIN);
            }
        }
        if (zero_bit) total_w = total_weight_all_pairs(a, ws);
        else
            while (ok && !counts_complete(ws, n_items, &total_w, &flt0)) {
                if (++spins > spin_limit) ok = false;
                __builtin_amdgcn_s_sleep(BXI_SLEEP_FIN);
            }
        BXI_WL(8, spins);
        if (lane == 0) { fin_f = dsum; fin_d[0] = total_w; fin_ok = ok ? 1 : 0; fin_flt = flt0 ? 1 : 0; }
    }
    __syncthreads();
    ok = fin_ok != 0; dsum = fin_f; total_w = fin_d[0];
    __syncthreads();
    // every thread watches its own arrival words (N * 8 / 256 each: one at the headline size); the launch ends on this loop
    long long mine = 0;
    unsigned int fault_seen = (ok ? 0u : kFaultFinisher) | (fin_flt ? kFaultCounts : 0u);
    spins = 0;
    for (; ok;) {
        mine = 0;
        int arrived = 0;
        bool flt = false;
        // (the words tile waves arrive on: those of instances 0 .. min(N, 64) - 1 -- tile_wave_arrives --, at most two per thread, asked for in one
        // round trip.  Rounds 3-6 walked all N x 8 words with one atomic load each: four dependent trips per poll at 128 instances, two of them for
        // words nobody arrives on.)
        {
            const int n_words = (N < 64 ? N : 64) * kAcc2Split;
            const int i0 = threadIdx.x, i1 = threadIdx.x + 256;
            unsigned long long x0 = 0ull, x1 = 0ull;
            if (n_words > 256) load8_past_x2(ws.acc2 + (size_t)(i0 < n_words ? i0 : 0) * kAcc2Stride, ws.acc2 + (size_t)(i1 < n_words ? i1 : 0) * kAcc2Stride, x0, x1);
            else x0 = __hip_atomic_load(ws.acc2 + (size_t)(i0 < n_words ? i0 : 0) * kAcc2Stride, BXI_RLX, BXI_AGENT);
            if (i0 >= n_words) x0 = 0ull;
            if (i1 >= n_words) x1 = 0ull;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned long long x = k ? x1 : x0;
                arrived += (int)(x >> 52);
                mine += (long long)(x & ((1ull << 52) - 1ull)) - ((long long)(x >> 52) << 24);       // the +1 per tile
                flt |= (x & (3ull << 50)) != 0ull;                                                   // a tile wave's wait ran out
            }
        }
        // the fault word (waves that gave up WITHOUT arriving set it; the finisher then runs out itself) rides in the same round
        if (threadIdx.x == 0) fault_seen |= __hip_atomic_load(ws.fault, BXI_RLX, BXI_AGENT);
        arrived = wave_total_i32(arrived);
        const bool anyflt = __any(flt);
        if (lane == 0) { fin_i[wave] = arrived; fin_b[wave] = anyflt ? 1 : 0; }
        __syncthreads();
        const bool all = (fin_i[0] + fin_i[1]) + (fin_i[2] + fin_i[3]) == n_tile_waves || BXI_AB(128);    // every tile wave arrives once, tiles or not
        if ((fin_b[0] | fin_b[1]) | (fin_b[2] | fin_b[3])) fault_seen |= kFaultCounts;
        __syncthreads();
        if (all) break;
        if (++spins > spin_limit) { ok = false; break; }           // workgroup-uniform: the same count in every thread
    }
    BXI_WL(10, spins);
    const double wsum = wave_total_f64((double)mine);                // exact; fixed order: run-to-run identical
    if (lane == 0) fin_d[wave] = wsum;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const double num = (fin_d[0] + fin_d[1]) + (fin_d[2] + fin_d[3]);
    const unsigned int status = (unsigned int)__builtin_amdgcn_readfirstlane((int)(fault_seen | (ok ? 0u : kFaultFinisher)));
    if (lane == 0) {
        const float denom = fmaxf((float)total_w, 1.f);                      // weights.sum().clamp(min=1.0), :1328
        float l0 = dsum / (float)N;                                          // .mean(), :143
        float l1 = (float)((num / (double)kNumScale) / (double)denom) * warmup;   // :1327-1332
        if (status) { l0 = __int_as_float(0x7fc00000); l1 = l0; }            // loud: mmdet's CheckInvalidLossHook fires
        losses[0] = l0; losses[1] = l1;
        if (st.scale) { st.scale[0] = warmup / denom; st.scale[1] = warmup; st.applied[0] = upp; st.applied[1] = upw; }   // [1]: the warm-up factor applied
        if (st.status) { st.status[0] = (int)status; if (ONE) st.status[1] = R; }
        if (st.iter) atomicAdd(st.iter, 1.0f);                               // self._iter += 1, condinst_head.py:1297
        // the evaluation is over: every other wave of it has been seen to arrive, so nobody reads the epoch any more
        if (ws.ep != kMaxTag) *ws.epoch = ws.ep;      // (a plain store: see with_tag)
    }
    if (ws.ep == kMaxTag) {
        // The tag counter is about to wrap: records of 2^28 evaluations ago would pass for fresh ones (table entries and arrival words of
        // instances beyond the current count keep their tags until an evaluation that large comes again).  So this evaluation ends by
        // returning the workspace to its initial state -- all zero, epoch 0 -- as bxi_boxinst_eval_workspace_init does: every other wave
        // has arrived, and in this evaluation every wave drains its stores before it arrives (pred_role; the others always do).
        // Once per 2^28 - 1 evaluations, ~3 MB by one wave.  (Targets an earlier bxi_boxinst_targets_f32 left in the workspace go too:
        // an evaluation that counts on them finds key 0 and says so, loud.)
        uint4* z = reinterpret_cast<uint4*>(ws.epoch);
        const size_t n16 = ws.ws_n16;
        for (size_t i = lane; i < n16; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    BXI_TW(3, 0, 1);
}

// tile workgroup: 4 independent waves striding through the tile list (its length is device data)
template <int D, int R, bool ONE>
__device__ __forceinline__ void tile_role(const InstArgs& a, const ValidCells& vc, const Ws& ws, float upw_warm, float n2max, int zero_bit, int n_items, int spin_limit,
                                          float* __restrict__ g_logits, unsigned char* smem, int tblk, int n_tb) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const int N = a.N;
    const int wid = tblk * kWaves + wave, nwaves = n_tb * kWaves;
    // (priority 0 like the stream, pool and leader waves.  Rounds 3-6 ran the tile waves at priority 2 -- "the launch ends on the tile waves, not on
    // the leaders next to them" --, and with the targets ready that starved what they themselves wait for: the stream waves whose band flags gate their
    // adds.  At 0: targets ready 14.1 -> 13.7 us at 32 instances, 18.2 -> 17.5 at 64, 21.5 -> 20.9 at 96, 26.0 -> 24.8 at 128; the un-split evaluation
    // 21.5 -> 21.2 at 64, unchanged at 32 / 96 / 128.  Priority 1 loses all of it; the predicate waves' 3 is worth 0.1-0.2 us at 32 instances.  R6-22)
    BXI_TW(1, wid, 0);
    int4 e0, e1 = make_int4(0, 0, 0, 0), eN = make_int4(0, 0, 0, 0);
    bool ok;
    if (ONE && R == 8 && N >= 64) {
        // the long single-launch form (64 instances or more; the short form runs 64..73 instances with the three calls below: see pred_words): entries 0..63, N and 64..127 polled for in ONE round trip (three tab_entry calls are three
        // statements with a wait each: three dependent trips in front of every tile of the long form)
        ok = false;
        const bool want1 = 64 + lane < N;
        for (int spins = 0; spins <= spin_limit; ++spins) {
            u4v v0, vN, v1;
            load16_past_x3(ws.tab + lane, ws.tab + N, ws.tab + (want1 ? 64 + lane : N), v0, vN, v1);
            if (__all(v0.w == ws.ep && vN.w == ws.ep && v1.w == ws.ep)) {
                e0 = make_int4((int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w); eN = make_int4((int)vN.x, (int)vN.y, (int)vN.z, (int)vN.w);
                if (want1) e1 = make_int4((int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w);
                ok = true;
                BXI_WL(1, spins);
                break;
            }
            __builtin_amdgcn_s_sleep(BXI_SLEEP_TAB);
        }
    } else {
        ok = tab_entry<ONE>(ws, lane, lane <= N, spin_limit, e0);
        if (N >= 64) ok = ok && tab_entry<ONE>(ws, N, true, spin_limit, eN);
        if (N > 64) ok = ok && tab_entry<ONE>(ws, 64 + lane, 64 + lane < N, spin_limit, e1);      // (with the two above: one round trip in the two-launch form)
    }
    // (a wave whose table wait ran out still arrives, saying so: the finisher then ends at once, loud, instead of running out itself.
    // The table's own zeroing of the arrival words precedes its entries, so without an entry the arrival may be wiped -- then the finisher
    // does run out: as loud)
    if (!ok) { tile_wave_arrives(ws, N, wid, 0, true); return; }
    const int total = N < 64 ? __builtin_amdgcn_readlane(e0.x, N < 64 ? N : 0) : __builtin_amdgcn_readfirstlane(eN.x);
    float* gbuf = reinterpret_cast<float*>(smem) + wave * ((R + 1) * 64);
    float scale = 0.f;
    bool have_scale = false, bad = false;
    long long fx_sum = 0;
    // (tiles are dealt wave by wave: the first workgroups' four waves each take one, the last quarter of the workgroups at 128 instances none.
    // Dealt workgroup by workgroup -- three per workgroup, nine per CU instead of twelve or eight -- the launch is SLOWER: 38.4 vs 37.2 us at 128
    // instances, 32.1 vs 31.1 at 96, targets ready 30.3 vs 28.7: the early workgroups' waves start their chains first.  profiles/NOTES.md R6-5)
    for (int ti = wid; ti < total && !bad; ti += nwaves) {
        if (BXI_AB(256) && (ti & 7) == 7) continue;              // (ablation only: an eighth of the tiles gone -- what fewer tile waves would be worth)
        Tile t;
        if (!locate_tile<D, R, ONE>(ws, vc, N, e0, e1, ti, a.h, a.w, spin_limit, t)) { bad = true; break; }
        BXI_TW(1, wid, 1);
        math_tile<D, R, ONE>(a, ws, t, upw_warm, n2max, zero_bit, n_items, spin_limit, scale, have_scale, g_logits, gbuf, wid, fx_sum, bad);
    }
    tile_wave_arrives(ws, N, wid, fx_sum, bad);
    BXI_TW(1, wid, 7);
}

// ---- launch 1 of the two-launch form ------------------------------------------------------------------------------------------
// what the first launch needs beyond its streams
struct PrepTail {
    int ready;               // BXI_EVAL_TARGETS_READY (the number of GT boxes + 1): no pool workgroups; the first table wave gathers sum W from the boxes' pair counts
    unsigned int key;        // ... and checks that the targets in the workspace are the ones this call means
};

// grid: [table blocks][pool blocks][stream blocks] (pool_first) or [table][stream][pool].  Nobody in this launch waits for anybody.
// (Round 5 also built a FOLDED form -- predicate workgroups and the reducer at this launch's tail, under its logit stream -- and measured it
// slower at every instance count, 37.9 vs 36.5 us at 128: profiles/NOTES.md R5-1.  Gone with ABI 7.)
__global__ __launch_bounds__(256, 5) void prep_kernel(PoolArgs pa, int n_pool, int n_items, InstArgs a, int dil, int R, Ws ws_in, LossState st,
                                                       float* __restrict__ g_logits, int vec, int pool_first, PrepTail tl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ws ws = ws_in;                                        // the tag is read where a role needs it (with_tag), behind its loads
    const int n_tab = ((a.N + 64) / 64 + kWaves - 1) / kWaves;
    const int Sn = (a.h + kSBlk - 1) / kSBlk;
    const int n_stream = a.N * Sn;
    const int blk = (int)blockIdx.x;
    const int tix = blk * kWaves + (int)(threadIdx.x >> 6);
    (void)tix;
    BXI_TW(0, tix, 0);
    int role = 0, idx = blk;                              // 0 table, 1 pool, 2 stream
    if (blk >= n_tab) {
        idx = blk - n_tab;
        const int n_a = pool_first ? n_pool : n_stream, n_b = pool_first ? n_stream : n_pool;
        (void)n_b;
        if (idx < n_a) role = pool_first ? 1 : 2;
        else { idx -= n_a; role = pool_first ? 2 : 1; }
    }
    if (role == 0) {
        const int k = blk * kWaves + (int)(threadIdx.x >> 6);
        if (64 * k <= a.N) { ws = with_tag(ws); table_wave(a, pa.meta, dil, R, ws, st, k, true, tl.ready, tl.key); }
    } else if (role == 2) {
        const LogitRows rows = {a.logits + (int64_t)(idx / Sn) * a.h * a.w, a.w, vec & 1, vec >> 1};
        stream_block<false>(a, ws, g_logits, vec & 1, idx, reinterpret_cast<unsigned long long*>(smem), rows, tix);     // (no tagged record in this form)
    } else {
        double* lut = reinterpret_cast<double*>(smem);
        double* fch = lut + 256;
        int* part = reinterpret_cast<int*>(fch + 3 * 64);
        pool_block(pa, ws, idx, n_pool, n_items, lut, part, fch, tix, [&](Ws& w_) { w_ = with_tag(w_); });
    }
    BXI_TW(0, tix, 7);
}

// grid: [n_pb predicate blocks][reducer][N leaders][n_tb tile blocks][finisher].  The only waits: a tile wave for the predicate waves
// (earlier in the grid, never waiting themselves), the finisher for everybody (nobody waits for it).  Every wait is bounded, and
// running out of it is loud: NaN losses, status word, poisoned gradient (the reference surfaces launch failures through
// AT_CUDA_CHECK, pairwise.cu:173,200).
template <int D, int R>
// (four workgroups per CU only where the tile role fits 128 VGPRs without spilling: dilation <= 2 -- every shipped configuration uses 2;
// dilation 3 needs 10-row register arrays and ran with 9 spilled VGPRs at four per CU)
__global__ __launch_bounds__(256, (R == 4 ? (D <= 2 ? 4 : 3) : (D <= 2 ? 3 : 2))) void pair_kernel(const float* __restrict__ up_prj, const float* __restrict__ up_pw, float warmup,
                                                       float n2max, int zero_bit, int n_pb, int n_items, int spin_limit, ValidCells vc, float* __restrict__ losses,
                                                       float* __restrict__ g_logits, InstArgs a, Ws ws_in, LossState st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[16];
    const int blk = (int)blockIdx.x;
    const int N = a.N;
    if (blk < n_pb) {                                                // ---- predicate waves: first in the grid, everybody asks for their words
        if (zero_bit) return;                                          // every pair weighs 1: sum W has a closed form, the tiles take the log-space path
        pred_role<false>(a, vc, ws_in /* no tag yet: it comes with the wave's first loads (pred_item) */, D, n2max, blk, n_pb, n_items, spin_limit);
        return;
    }
    const Ws ws = with_tag(ws_in);
    const float upp = up_prj ? *up_prj : 1.f, upw = up_pw ? *up_pw : 1.f;
    // (grid order [predicate][reducer][leaders][tiles][finisher].  Measured and dropped -- profiles/NOTES.md R6-10 --: the tile workgroups AHEAD of the leaders,
    // so that 128 more of them are resident from the start: 30.7 vs 28.4 us at 96 instances, 35.0 vs 33.0 at 128 -- the leaders then run last and the
    // finisher waits for their dice words)
    const int n_tb = (int)gridDim.x - 2 - N - n_pb;
    const int lead0 = n_pb + 1, tile0 = n_pb + 1 + N;
    if (blk == n_pb) {                                          // ---- the reducer
        reducer_role<false>(ws, zero_bit, n_pb > 0 ? n_items : 0, spin_limit);      // (no predicate workgroups here: sum W is in memory already)
    } else if (blk == (int)gridDim.x - 1) {
        finisher_role<false>(a, ws, st, upp, upw, resolve_warmup(warmup, st.iter), zero_bit, n_items, spin_limit, R, n_tb * kWaves, losses);
    } else if (blk >= lead0 && blk < lead0 + N) {                      // ---- leader of an instance
        BXI_TW(3, 1 + blk - lead0, 0);
        leader_block<false>(a, D, ws, st, blk - lead0, upp, g_logits, smem, red, spin_limit);
    } else {
        tile_role<D, R, false>(a, vc, ws, upw * resolve_warmup(warmup, st.iter), n2max, zero_bit, n_items, spin_limit, g_logits, smem, blk - tile0, n_tb);
    }
}

// ---- the single-launch form ---------------------------------------------------------------------------------------------------
// All roles in ONE grid, in this order:  [stream blocks (their first waves write the table)][pool blocks][predicate blocks][reducer][N leaders][tile blocks][finisher].
// Workgroups are dispatched in grid order and every wait is for a workgroup EARLIER in the grid:
//   table, stream, pool   wait for nobody;
//   leader n              for the table entry n and the band flags of instance n (stream blocks);
//   predicate wave        for the Lab pixels it reads (pool blocks; 16-byte records carrying the evaluation's tag) and the table;
//   tile wave             for the table, its predicate words (tagged), sum W (reducer <- predicate blocks) and the band flags of
//                         the rows it adds onto (stream blocks);
//   finisher              for everybody.
// So no waiter can hold a slot that a workgroup it waits for still needs.  What crosses workgroups is written through (sc1) and
// read past the caches; what a flag announces is drained (s_waitcnt vmcnt(0)) before the flag goes out; what announces itself is
// one 16-byte (or 4-byte) record written by one store.  Four workgroups per CU (<= 128 VGPRs: the tile role's budget): at the
// headline size the table + stream + pool workgroups fill the GPU once, and the back half flows into the slots they leave -- the
// kernel boundary of the two-launch form (~2.2 us) is gone.  The two-launch form stays for 8-row tiles (> 96 instances: 2
// workgroups per CU would starve the front half), dilation 4, the head-fused first launch and the generic pooling path.
// R = 8 ("the long form", three workgroups per CU): the same grid for MANY instances.  With 4-row tiles 128 instances are ~4600 tiles on at most
// half the slots' waves -- every tile wave then walks two or three tiles, each a ~7 us dependent chain, one after the other (45 us per
// evaluation).  8-row tiles are ~2300, one per wave, and the whole back half runs while the front half still streams: no kernel boundary,
// the tile waves' arithmetic under the logit stream.  Nothing here waits for a workgroup LATER in the grid (no staying-on: `merge` = 0),
// so the form makes progress at any residency; the pool workgroups go first (`pool_first`), the image side being the longer chain.
// READY (BXI_EVAL_TARGETS_READY): an instantiation of its own -- no pool / predicate role, table workgroups at the head of the grid, sum W
// gathered by the reducer workgroup -- so that the un-split kernel stays what it was (the same registers, no extra argument).
template <int D, int R, bool READY>
__global__ __launch_bounds__(256, (R == 4 ? kOneOcc : BXI_LONG_OCC)) void eval1_kernel(PoolArgs pa, int n_pool, int n_items, int n_pb, int n_tb, InstArgs a, Ws ws_in, LossState st, ValidCells vc,
                                                        const float* __restrict__ up_prj, const float* __restrict__ up_pw, float warmup, float n2max, int spin_limit,
                                                        float* __restrict__ losses, float* __restrict__ g_logits, int vec, int merge, int ready_in, unsigned int key,
                                                        int n_tabw_in) {
    const int ready = READY ? ready_in : 0, n_tabw = READY ? n_tabw_in : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[16];
    Ws ws = ws_in;
    const int N = a.N;
    constexpr int n_tab = 0;                                  // (trace index layout: table, stream, pool)
    const int Sn = (a.h + kSBlk - 1) / kSBlk;
    const int n_stream = N * Sn;
    const int blk = (int)blockIdx.x;
    // n_tabw > 0 (targets ready): the table has workgroups of its own at the head of the grid.  With the image side in memory the table IS the
    // head of every chain (tile waves, leaders), and as a duty of a stream wave behind its zero-fill and loads it came ~5 us into the launch
    // (its drain waits for that wave's 8 KB of written-through stores) and made the first instance's band the last one.
    if (READY && blk < n_tabw) {
        const int k = blk * kWaves + (int)(threadIdx.x >> 6);
        if (64 * k <= N) table_wave(a, pa.meta, D, R, with_tag(ws), st, k, false, -1, 0u);
        return;
    }
    // role of this workgroup: 0 stream, 1 pool, 2 leader, 3 predicate, 4 tile, 5 finisher, 6 reducer
    int role, idx = blk - n_tabw;
    constexpr bool pool_first = R == 8;                      // the long form (a run-time switch here costs the short form a stack slot)
    const int n_a = pool_first ? n_pool : n_stream, n_b = pool_first ? n_stream : n_pool;      // grid: [stream][pool] or [pool][stream], then the back half
    if (idx < n_a) role = pool_first ? 1 : 0;
    else if ((idx -= n_a) < n_b) role = pool_first ? 0 : 1;
    else if ((idx -= n_b) < n_pb) role = 3;
    else if ((idx -= n_pb) < 1) role = 6;                              // the reducer
    else if ((idx -= 1) < N) role = 2;
    else if ((idx -= N) < n_tb) role = 4;
    else role = 5;
    const int tix = (n_tab + (role == 0 ? idx : n_stream + idx)) * kWaves + (int)(threadIdx.x >> 6);
    (void)tix;
    bool stayed = false;
    if (role == 0) {
        BXI_TW(0, tix, 0);
        // the table is the first duty of the first stream workgroups' first waves (wave k of the table in workgroup k): a workgroup
        // of its own would be the one workgroup too many for the front half to be resident at once at the headline size
        const LogitRows rows = {a.logits + (int64_t)(idx / Sn) * a.h * a.w, a.w, vec & 1, vec >> 1};
        stream_block<true>(a, ws, g_logits, vec & 1, idx, reinterpret_cast<unsigned long long*>(smem), rows, tix, [&](Ws& w_) {
            w_ = with_tag(w_);
            if (!READY && (threadIdx.x >> 6) == 0 && 64 * idx <= N) table_wave(a, pa.meta, D, R, w_, st, idx, false, 0, 0u);
        });
        BXI_TW(0, tix, 7);
        if (!merge) return;
        // ... and stays as a tile workgroup: its four waves are the first tile waves, resident since the start of the launch, so their
        // table -> tile -> logits -> per-pixel chain runs while the pool workgroups finish instead of behind a slot that has to come
        // free first.  (It now waits for predicate workgroups LATER in the grid.  Those wait only for pool workgroups, which wait for
        // nobody, and the host launches this form only while the stream workgroups leave at least half of the slots free: a
        // predicate workgroup always finds a slot.)
        __syncthreads();                                                   // the column-partial LDS becomes the tile waves' scratch
        role = 4;
        stayed = true;
        idx -= n_stream;                                                   // tile workgroup index idx + n_stream below
    }
    if (!READY && role == 1) {
        BXI_TW(0, tix, 0);
        double* lut = reinterpret_cast<double*>(smem);
        double* fch = lut + 256;
        int* part = reinterpret_cast<int*>(fch + 3 * 64);
        pool_block(pa, ws, idx, n_pool, n_items, lut, part, fch, tix, [&](Ws& w_) { w_ = with_tag(w_); });
        BXI_TW(0, tix, 7);
        return;
    }
    // (a stream workgroup that stays on has its tag already; predicate waves and leaders get it with their first poll)
    if (!stayed && role != 3 && role != 2) ws = with_tag(ws);
    const float upp = up_prj ? *up_prj : 1.f, upw = up_pw ? *up_pw : 1.f;
    if (role == 2) {
        BXI_TW(3, 1 + idx, 0);
        leader_block<true>(a, D, ws, st, idx, upp, g_logits, smem, red, spin_limit);
        return;
    }
    if (!READY && role == 3) { pred_role<true>(a, vc, ws, D, n2max, idx, n_pb, n_items, spin_limit, merge != 0); return; }
    if (role == 6) {
        if (!READY) reducer_role<true>(ws, 0, n_items, spin_limit);
        else if (threadIdx.x < 64) {
            // targets ready: sum W is a gather over the instances' boxes, published once the table says this evaluation's polled words are zeroed
            // (the table wave is a wave of the first stream workgroup: earlier in the grid, waiting for nobody)
            if (table_complete<true>(ws, 0, spin_limit)) publish_gathered_sumw(a, ws, ready - 1, key);
            else if (threadIdx.x == 0) atomicOr(ws.fault, kFaultCounts);
        }
        return;
    }
    if (role == 4) {          // ONE call site for the stream workgroups that stay on and for the tile workgroups proper
        const int shift = merge ? n_stream : 0;
        tile_role<D, R, true>(a, vc, ws, upw * resolve_warmup(warmup, st.iter), n2max, 0, n_items, spin_limit, g_logits, smem, idx + shift, n_tb + shift);
        return;
    }
    finisher_role<true>(a, ws, st, upp, upw, resolve_warmup(warmup, st.iter), 0, n_items, spin_limit, R, (n_tb + (merge ? n_stream : 0)) * kWaves, losses);
}

// ---- bxi_boxinst_targets_f32: the image side alone, ahead of the evaluation ------------------------------------------------------
// CondInstMaskHead.loss computes its targets from `imgs` and `gt_bboxes` alone (get_targets, condinst_head.py:1298-1299, :1345-1448) --
// both exist before the backbone runs (condinst.py:53 vs :73).  Two launches with NO in-kernel wait (a kernel boundary between them), so the
// call makes progress next to anything: launch 1 = the pool workgroups of prep_kernel (Lab records, tag field 0) + one workgroup that writes
// the box table and zeroes the boxes' count words; launch 2 = the predicate waves over the box table, per-box pair counts.
// (seven workgroups per CU -- the pool role needs 66 registers --: the 1600 items of a 2 x 800 x 1024 batch are resident at once, one item each)
__global__ __launch_bounds__(256, 7) void targets_pool_kernel(PoolArgs pa, int n_pool, int n_items, GtTable gt, int Hc, int Wc, int stride, int h, int w, Ws ws,
                                                               unsigned int key) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x == 0) {
        const int G = gt.first[gt.B];
        for (int g = threadIdx.x; g < G; g += 256) {
            int img = 0;
            const float* bp = gt_box(gt, g, img);
            const Rect rc = box_rect(bp, Hc, Wc, stride, stride / 2, h, w);
            ws.boxtab()[g] = make_int4(img << 24, rc.r0 | (rc.r1 << 16), rc.c0 | (rc.c1 << 16), 0);
#pragma unroll
            for (int j = 0; j < kBoxSplit; ++j) ws.boxcnt()[((size_t)g * kBoxSplit + j) * kAcc2Stride] = 0ull;
        }
        if (threadIdx.x == 0) *ws.tkey() = key;
        return;
    }
    double* lut = reinterpret_cast<double*>(smem);
    double* fch = lut + 256;
    int* part = reinterpret_cast<int*>(fch + 3 * 64);
    pool_block(pa, ws, (int)blockIdx.x - 1, n_pool, n_items, lut, part, fch, 0);
}

__global__ __launch_bounds__(256) void targets_pred_kernel(int h, int w, int G, ValidCells vc, Ws ws, int D, float n2max, int n_pb, int n_items) {
    __shared__ int boxacc[kBoxCap];
    for (int g = threadIdx.x; g < G; g += 256) boxacc[g] = 0;
    __syncthreads();
    const int wave = (int)(threadIdx.x >> 6);
    const int segs = (w + 63) >> 6;
    bool ok = true;
    for (int item = (int)blockIdx.x * kWaves + wave; item < n_items; item += n_pb * kWaves)
        (void)pred_item<false, true>(h, w, G, vc, ws, D, n2max, item, segs, 0, ok, boxacc);
    __syncthreads();
    // one arrival per (workgroup, box it met): integer adds commute -- run-to-run identical
    for (int g = threadIdx.x; g < G; g += 256) {
        const int v = boxacc[g];
        if (v) __hip_atomic_fetch_add(ws.boxcnt() + ((size_t)g * kBoxSplit + (blockIdx.x & (kBoxSplit - 1))) * kAcc2Stride, (unsigned long long)v, BXI_RLX, BXI_AGENT);
    }
}

// ---- rescale: g_logits finished for the factors recorded in `state` -> finished for (g_prj, g_pw) ----------------
// grid (8, N).  No-op when the factors are the recorded ones (the usual case: loss.backward() seeds both terms with 1).
// An evaluation whose status word is set has no gradient: it is poisoned here, next to the NaN losses.
__global__ __launch_bounds__(256) void rescale_kernel(InstArgs a, int dil, LossState st, const float* __restrict__ g_prj,
                                                       const float* __restrict__ g_pw, float* __restrict__ g_logits) {
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int h = a.h, w = a.w;
    float* G = g_logits + (int64_t)n * h * w;
    if (st.status[0] != 0) {
        const int per = (h + gridDim.x - 1) / gridDim.x;
        const int ra = s * per, rb = min(h, ra + per);
        for (int i = tid; i < (rb - ra) * w; i += 256) G[(int64_t)ra * w + i] = __int_as_float(0x7fc00000);
        return;
    }
    const float np = *g_prj, nw = *g_pw, op = st.applied[0], ow = st.applied[1];
    if (np == op && nw == ow) return;
    const int R = st.status[1];
    const InstRec rec = st.inst[n];
    const InstBox ib = inst_from_rec(rec, dil, h, w);
    const int hr0 = ib.any ? (ib.dil.r0 / R) * R : 0, hr1 = ib.any ? min(h, ((ib.dil.r1 + R - 1) / R) * R) : 0;
    const int hc0 = ib.dil.c0, hc1 = ib.any ? ib.dil.c1 : 0;
    const float ratio = nw / ow;                      // recorded g_pw == 0 cannot be rescaled (documented)
    const unsigned long long* ckp = st.colk + (int64_t)n * w; const unsigned long long* rkp = st.rowk + (int64_t)n * h;
    auto carg = [&](int c) { return (int)(unsigned int)ckp[c]; };
    auto rarg = [&](int r) { return (int)(unsigned int)rkp[r]; };
    auto gcol = [&](int c) { return __uint_as_float((unsigned int)(ckp[c] >> 32)); };
    auto grow = [&](int r) { return __uint_as_float((unsigned int)(rkp[r] >> 32)); };
    const int cw = hc1 - hc0, rows = hr1 - hr0;
    const int per = (rows + gridDim.x - 1) / gridDim.x;
    const int ra = hr0 + s * per, rb = min(hr1, ra + per);
    const int npx = cw > 0 && rb > ra ? (rb - ra) * cw : 0;
    for (int i = tid; i < npx; i += 256) {            // G = ow*s*d + op*sp  ->  nw*s*d + np*sp
        const int r = ra + i / cw, c = hc0 + i % cw;
        float sp = 0.f;
        if (carg(c) == r) sp += gcol(c);
        if (rarg(r) == c) sp += grow(r);
        const float v = G[(int64_t)r * w + c];
        G[(int64_t)r * w + c] = (v - op * sp) * ratio + np * sp;
    }
    if (s == 0) {                                     // arg-max positions outside the hull hold op * sp
        for (int c = tid; c < w; c += 256) {
            const int r = carg(c);
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t) {
                float v = gcol(c);
                if (rarg(r) == c) v += grow(r);
                G[(int64_t)r * w + c] = v * np;
            }
        }
        for (int r = tid; r < h; r += 256) {
            const int c = rarg(r);
            const bool in_t = r >= hr0 && r < hr1 && c >= hc0 && c < hc1;
            if (!in_t && carg(c) != r) G[(int64_t)r * w + c] = grow(r) * np;
        }
    }
}

__global__ void zero_losses2_kernel(float* losses, float* iter) { losses[0] = 0.f; losses[1] = 0.f; if (iter) atomicAdd(iter, 1.0f); }

// ---- host side ---------------------------------------------------------------------------------------------------
// Developer knobs (tools/ A/B scripts): read from the environment ONLY in a -DBXI_DEV build.  The shipped library reads nothing from the
// process environment: what varies is an argument (`flags`), as include/boxinst_hip.h promises.
#ifdef BXI_DEV
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define BXI_KNOB(name, dflt) ([&]() -> int { static const int set = getenv(name) ? 1 : 0; static const int v = env_int(name, 0); return set ? v : (dflt); }())
#else
#define BXI_KNOB(name, dflt) (dflt)
#endif

// A stream that is being captured into a hipGraph: the launch is recorded, not run; what the host decides here is frozen into the graph
static bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Compute units the stream may use: a CU mask (hipExtStreamCreateWithCUMask, ROC_GLOBAL_CU_MASK) leaves fewer than the device has.
// The single-launch form needs to know (its stream workgroups must leave free slots for the workgroups they wait for).  Cached per
// stream handle (a handful of streams per process); a failing query counts as "the whole device".
static int stream_cus(hipStream_t s, int device_total) {
    struct Entry { std::atomic<uintptr_t> key; std::atomic<int> cus; };
    static Entry cache[8];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const uintptr_t k = (reinterpret_cast<uintptr_t>(s) + 1) ^ ((uintptr_t)(unsigned)dev << 56);   // +1: the null stream is a key too; per device
    for (auto& e : cache)
        if (e.key.load(std::memory_order_acquire) == k) return e.cus.load(std::memory_order_relaxed);
    uint32_t mask[32] = {};
    int n = device_total;
    if (hipExtStreamGetCUMask(s, 32, mask) == hipSuccess) {
        int bits = 0;
        for (uint32_t m : mask) bits += __builtin_popcount(m);
        if (bits > 0 && bits < n) n = bits;
    } else {
        (void)hipGetLastError();
    }
    static std::atomic<unsigned> next{0};
    Entry& e = cache[next.fetch_add(1, std::memory_order_relaxed) % 8];
    e.cus.store(n, std::memory_order_relaxed);
    e.key.store(k, std::memory_order_release);
    return n;
}

// Rows per tile: 4 up to ~95 instances, 8 from there on (two launches; dilation <= 2).  With 4-row tiles 128 instances are ~4600
// tiles on ~2500 tile waves: two rounds of a ~7 us dependent chain (table -> logits -> predicate words -> pair loop -> sum W -> adds).
// 8-row tiles halve the count; at 157 - 161 VGPRs they run three workgroups per CU (round 3 ran them at two, where they lost): one
// round.  Measured (two launches, 200 x 256 maps, same box): 96 instances 29.4 vs 30.1 us, 128: 35.9 vs 38.3, 256: 64.0 vs 69.7;
// 64 instances: 24.2 vs 23.8 (single launch) -- hence the threshold.  BXI_TILE_ROWS / BXI_EVAL_TILE_ROWS_8 override.
static int tile_rows_for(int N, int dil) { return (N >= 96 && dil <= 2) ? 8 : 4; }

// digest of what targets are computed FOR -- canvas, stride, window, threshold, per image its shape, rows removed and box count -- kept in
// the workspace by bxi_boxinst_targets_f32 and compared on the device by an evaluation that counts on them (FNV-1a; never 0)
static unsigned int targets_key(const PoolArgs& pa, int B, int Hc, int Wc, int stride, int dil, float n2max, const int* first /* [B + 1] */) {
    unsigned int hsh = 2166136261u;
    auto mix = [&](unsigned int v) { for (int i = 0; i < 4; ++i) { hsh ^= (v >> (8 * i)) & 255u; hsh *= 16777619u; } };
    unsigned int nb;
    memcpy(&nb, &n2max, 4);
    mix((unsigned)B); mix((unsigned)Hc); mix((unsigned)Wc); mix((unsigned)stride); mix((unsigned)dil); mix(nb);
    for (int b = 0; b < B; ++b) { mix((unsigned)pa.meta.img_h[b]); mix((unsigned)pa.meta.img_w[b]); mix((unsigned)pa.meta.first_removed[b]); mix((unsigned)(first[b + 1] - first[b])); }
    return hsh ? hsh : 1u;
}
static void valid_cells_of(const PoolArgs& pa, int B, int stride, int h, int w, ValidCells& vc) {
    for (int b = 0; b < B; ++b) {                  // the device formula (valid_cells), evaluated here once per image
        const int half = stride / 2;
        auto cells = [&](int limit, int n) { const int v = limit - half <= 0 ? 0 : (limit - half + stride - 1) / stride; return v < n ? v : n; };
        vc.vrow[b] = cells(pa.meta.img_h[b] < pa.meta.first_removed[b] ? pa.meta.img_h[b] : pa.meta.first_removed[b], h);
        vc.vcol[b] = cells(pa.meta.img_w[b], w);
    }
}

// (sim >= thresh) for a valid neighbour as a compare on the squared Lab distance: exp(-0.5 * sqrt(n2)) >= thresh  <=>  n2 <= n2max
// (get_image_color_similarity :237 + the threshold of loss() :1324), n2max found by bisecting the f32 expression over the float
// bit patterns (it is non-increasing in n2 >= 0, and positive floats order like their bit patterns).  Host arithmetic: sqrtf is
// correctly rounded everywhere; expf is the C library's, as in the CPU reference path.
struct HostPred { float n2max; int zero_bit; };
static bool host_sim_pred(float n2, float thresh) { return expf(-sqrtf(n2) * 0.5f) >= thresh; }
static HostPred host_pred(float thresh) {
    static std::atomic<uint64_t> cache{~0ull};                          // (thresh bits << 32 | n2max bits) of the last call
    uint32_t tb, nb;
    memcpy(&tb, &thresh, 4);
    const uint64_t c = cache.load(std::memory_order_relaxed);
    HostPred p;
    p.zero_bit = (0.f >= thresh) ? 1 : 0;                               // weight of a padded / masked-out neighbour (sim == 0)
    if ((uint32_t)(c >> 32) == tb && c != ~0ull) { nb = (uint32_t)c; memcpy(&p.n2max, &nb, 4); return p; }
    if (!host_sim_pred(0.f, thresh)) p.n2max = -1.f;                    // thresh > 1: never
    else if (host_sim_pred(3.0e38f, thresh)) p.n2max = INFINITY;        // thresh <= 0 (exp underflows to 0): always
    else {
        uint32_t lo = 0u, hi;
        const float big = 3.0e38f;
        memcpy(&hi, &big, 4);                                           // pred(lo) true, pred(hi) false
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float fm;
            memcpy(&fm, &mid, 4);
            if (host_sim_pred(fm, thresh)) lo = mid; else hi = mid;
        }
        memcpy(&p.n2max, &lo, 4);
    }
    memcpy(&nb, &p.n2max, 4);
    cache.store(((uint64_t)tb << 32) | nb, std::memory_order_relaxed);
    return p;
}

template <int D, int R>
static void launch_pair(hipStream_t s, int grid, size_t lds, const InstArgs& a, float warmup, float n2max, int zero_bit, int n_pb, int n_items, int spin_limit,
                        const ValidCells& vc, const Ws& ws, const LossState& st, float* losses, float* g_logits, const float* up_prj, const float* up_pw) {
    BXI_LAUNCH(n_pb > 0 ? "pair" : "pair_tiles", s, (pair_kernel<D, R>), dim3((unsigned)grid), dim3(256), lds, s, up_prj, up_pw, warmup, n2max, zero_bit,
               n_pb, n_items, spin_limit, vc, losses, g_logits, a, ws, st);
}

size_t eval_ws_bytes(int B, int N, int h, int w) { return carve(nullptr, B, N, h, w, nullptr); }
bool fused_eval_supported(int dil) { return dil >= 1 && dil <= kMaxDilFused; }
// the one-time initialisation of a workspace: all zero (epoch 0, no record carries a tag an evaluation will draw)
int eval_ws_init(void* workspace, size_t bytes, void* stream) {
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    if (hipMemsetAsync(workspace, 0, bytes, as_stream(stream)) != hipSuccess) { set_last_hip_error((int)hipGetLastError()); return BXI_ERR_LAUNCH; }
    return BXI_OK;
}

// the per-instance capacity a workspace of this size admits for this canvas (the layout is a function of both, never of a call's N)
static int ws_capacity(int B, int h, int w, int N, size_t workspace_bytes) {
    // (a pure function of (B, h, w, size): the last answer is kept per host thread -- a training loop asks the same question every
    // iteration, and the search is sixteen layouts)
    struct Last { int B, h, w, n_cap; size_t bytes; };
    static thread_local Last last = {0, 0, 0, 0, 0};
    if (last.bytes == workspace_bytes && last.B == B && last.h == h && last.w == w && last.n_cap >= N) return last.n_cap;
    int lo = N, hi = kMaxInst - 1;           // carve() is non-decreasing in N
    while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        if (carve(nullptr, B, mid, h, w, nullptr) <= workspace_bytes) lo = mid; else hi = mid - 1;
    }
    last = Last{B, h, w, lo, workspace_bytes};
    return lo;
}

// bxi_boxinst_targets_f32: Lab, predicate words and per-GT-box pair counts into the workspace, ahead of the evaluation
int launch_targets(const bxi_image_batch* batch, const float* const* boxes_per_img_host, const int* gt_count_host, int stride, int dil, float color_thresh,
                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!batch || !workspace) return BXI_ERR_NULL_POINTER;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (batch->B <= 0 || batch->B > BXI_MAX_IMAGES || stride < 1 || batch->Hc % stride || batch->Wc % stride) return BXI_ERR_BAD_SHAPE;
    if (!batch->imgs) return BXI_ERR_NULL_POINTER;
    if (batch->image_masks) return BXI_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    PoolArgs pa = {};
    int rc = fill_pool_args(batch, nullptr, nullptr, pa);
    if (rc != BXI_OK) return rc;
    GtTable gt;
    int G = 0;
    rc = fill_gt_table(boxes_per_img_host, gt_count_host, batch->B, gt, G);
    if (rc != BXI_OK) return rc;
    if (G > kBoxCap) return BXI_ERR_UNSUPPORTED;           // the evaluation then computes its targets itself (no BXI_EVAL_TARGETS_READY)
    const int h = batch->Hc / stride, w = batch->Wc / stride;
    if (h > 65535 || w > 65535) return BXI_ERR_BAD_SHAPE;
    const HostPred pr = host_pred(color_thresh);
    if (pr.zero_bit) return BXI_ERR_UNSUPPORTED;           // every pair weighs 1: nothing of the image is needed ahead
    const size_t need = carve(nullptr, batch->B, 0, h, w, nullptr);
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    Ws ws;
    carve(workspace, batch->B, ws_capacity(batch->B, h, w, 0, workspace_bytes), h, w, &ws);
    ws.ws_n16 = (unsigned int)(workspace_bytes / 16);
    ValidCells vc = {};
    valid_cells_of(pa, batch->B, stride, h, w, vc);
    const int64_t n_items64 = (int64_t)batch->B * h * ((w + 63) / 64);
    if (n_items64 > 0x7fffffffLL) return BXI_ERR_BAD_SHAPE;
    const int n_items = (int)n_items64;
    const unsigned int key = targets_key(pa, batch->B, batch->Hc, batch->Wc, stride, dil, pr.n2max, gt.first);
    const bool pooled_in_launch = pool_vec_ok(batch, stride);
    const int slots = 7 * device_cus();
    int n_pool = 0;
    if (pooled_in_launch) {
        const int per = (n_items + slots - 1) / slots;
        n_pool = (n_items + (per < 1 ? 1 : per) - 1) / (per < 1 ? 1 : per);
    }
    const size_t lds1 = sizeof(double) * (256 + 3 * 64) + sizeof(int) * 4 * 3 * 64;
    BXI_LAUNCH("targets_pool", s, targets_pool_kernel, dim3((unsigned)(1 + n_pool)), dim3(256), lds1, s, pa, n_pool, n_items, gt, batch->Hc, batch->Wc, stride, h, w,
               ws, key);
    rc = check_launch();
    if (rc != BXI_OK) return rc;
    if (!pooled_in_launch) {       // other strides / unaligned canvases: the generic pooling kernels, then the repacking into 16-byte records
        rc = launch_pool(batch, stride, nullptr, ws.lab_planar, s);
        if (rc != BXI_OK) return rc;
        const int64_t BP = (int64_t)batch->B * h * w;
        BXI_LAUNCH("pack_lab4", s, pack_lab4_kernel, dim3((unsigned)((BP + 255) / 256 > 2048 ? 2048 : (BP + 255) / 256)), dim3(256), 0, s,
                   (const float*)ws.lab_planar, ws.lab4, (const unsigned int*)ws.epoch, batch->B, (int64_t)h * w);
        rc = check_launch();
        if (rc != BXI_OK) return rc;
    }
    int n_pb = (n_items + kWaves - 1) / kWaves;
    if (n_pb > 8 * device_cus()) n_pb = 8 * device_cus();
    BXI_LAUNCH("targets_pred", s, targets_pred_kernel, dim3((unsigned)n_pb), dim3(256), 0, s, h, w, G, vc, ws, dil, pr.n2max, n_pb, n_items);
    return check_launch();
}

// One evaluation: one launch (eval1) or two (prep, pair).
int launch_fused_eval(const bxi_image_batch* batch, float color_thresh, const bxi_instances* in, int dil, float warmup, const float* up_prj,
                 const float* up_pw, float* losses, float* g_logits, void* state, void* workspace, size_t workspace_bytes, unsigned flags,
                 void* stream, const DynArgs* head, int head_C) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (!losses || !batch) return BXI_ERR_NULL_POINTER;
    hipStream_t s = as_stream(stream);
    PoolArgs pa = {};
    if (batch->Hc != in->Hc || batch->Wc != in->Wc || batch->B != in->B) return BXI_ERR_BAD_SHAPE;
    // the image side is in the workspace (bxi_boxinst_targets_f32): `imgs` is not read.  (The kernels get the number of GT boxes + 1.)
    const int ready = (flags & kFlagTargetsReady) ? a.gt.first[a.gt.B] + 1 : 0;
    rc = fill_pool_args(batch, nullptr, nullptr, pa);
    if (rc != BXI_OK) return rc;
    if (batch->B > 0 && !batch->imgs && !ready) return BXI_ERR_NULL_POINTER;
    if (batch->image_masks) return BXI_ERR_UNSUPPORTED;   // explicit masks: use bxi_color_affinity_f32 + bits
    if (warmup < 0.f && !in->iter_counter) return BXI_ERR_NULL_POINTER;     // the factor is to come from the device counter
    if (!(warmup == warmup)) return BXI_ERR_BAD_ARGUMENT;
    if (a.N == 0) {
        BXI_LAUNCH("zero_losses", s, zero_losses2_kernel, dim3(1), dim3(1), 0, s, losses, in->iter_counter);
        return check_launch();
    }
    if (a.N >= kMaxInst || a.h > 65535 || a.w > 65535) return BXI_ERR_BAD_SHAPE;
    if (batch->B <= 0) return BXI_ERR_BAD_SHAPE;
    if (g_logits && !state) return BXI_ERR_NULL_POINTER;
    const bool pooled_in_launch = ready || pool_vec_ok(batch, a.stride);     // else: the generic pooling kernels in launches of their own
    const size_t need = carve(nullptr, batch->B, a.N, a.h, a.w, nullptr);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) return BXI_ERR_WORKSPACE;
    // The layout is a function of (B, h, w) and of the workspace's SIZE, not of this call's instance count: the per-instance regions are
    // carved for the largest count the size admits, so every evaluation of this canvas on this workspace finds every kind of record at
    // the same address, whatever N it has.  A word that ever held a tag then only ever holds tags of the same kind (or the zero of the
    // one-time initialisation), and tags grow monotonically -- a stale word can never pass for a fresh one.  (With the layout moving
    // with N, a small tag could meet an old PAYLOAD word of the same value -- a predicate word is 16 * tag + bits -- found by
    // tools/extended_fuzz.py: intermittent wrong results, status 0.)
    Ws ws;
    carve(workspace, batch->B, ws_capacity(batch->B, a.h, a.w, a.N, workspace_bytes), a.h, a.w, &ws);
    ws.ws_n16 = (unsigned int)(workspace_bytes / 16);
    ws.pred_any = (ready ? 1u : 0u);
    LossState st = {};
    if (state) {
        if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
        carve_state(state, a.N, a.h, a.w, &st);
    }
    st.iter = in->iter_counter;
    // bit 0: 16-byte rows; bit 1: the logit stream is read non-temporally -- where the maps outgrow the L2 anyway (128 instances: 52 MB) the
    // stream's lines only push the Lab / predicate / table lines out of it: 37.4 -> 34.3 us per evaluation at 128 instances; at 32 (6.5 MB, which
    // the tile waves find in the L2 again) the hint costs 0.4 us.  profiles/NOTES.md R6-7
    const int nt_stream = (int64_t)a.N * a.h * a.w * 4 >= ((int64_t)BXI_KNOB("BXI_NT_FROM_MB", kNtStreamFromMB) << 20) ? 2 : 0;
    const int vec = (((a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0 &&
                      (!g_logits || (reinterpret_cast<uintptr_t>(g_logits) & 15) == 0)) ? 1 : 0) | nt_stream;
    const int env_rows = BXI_KNOB("BXI_TILE_ROWS", 0);            // developer knobs (-DBXI_DEV builds only)
    const int env_pool_first = BXI_KNOB("BXI_POOL_FIRST", 0);
    const int env_pool_wgs = BXI_KNOB("BXI_POOL_WGS_PER_CU", 5);
    const int force_rows = (flags & kFlagRows8) ? 8 : ((flags & kFlagRows4) ? 4 : env_rows);
    const int R = force_rows == 4 || force_rows == 8 ? force_rows : tile_rows_for(a.N, dil);
    if (eval_cap(a.N, a.h, a.w, dil, R) >= (1 << 24)) return BXI_ERR_BAD_SHAPE;     // the table packs a tile prefix into 24 bits
    const HostPred pr = host_pred(color_thresh);
    if (ready && pr.zero_bit) return BXI_ERR_UNSUPPORTED;       // (bxi_boxinst_targets_f32 refuses thresholds <= 0 as well)
    ValidCells vc = {};
    valid_cells_of(pa, batch->B, a.stride, a.h, a.w, vc);
    const int64_t n_items64 = (int64_t)batch->B * a.h * ((a.w + 63) / 64);
    if (n_items64 > 0x7fffffffLL) return BXI_ERR_BAD_SHAPE;
    const int n_items = (int)n_items64;
    const int spin_limit = (flags & kFlagGiveUp) ? -1 : kSpinLimit;
    const unsigned int key = ready ? targets_key(pa, batch->B, batch->Hc, batch->Wc, a.stride, dil, pr.n2max, a.gt.first) : 0u;

    // ---- the single-launch form ---------------------------------------------------------------------------------------
    const int env_one = BXI_KNOB("BXI_ONE_LAUNCH", 1);            // developer knob: 0 = always two launches
    // The single launch pays off while its front half (stream + pool workgroups) is resident at once: measured 24.8 vs 27.0 us at
    // 64 instances, 35.1 vs 33.9 at 96, 45.1 vs 39.4 at 128 (200 x 256 maps) -- hence: stream workgroups <= half the slots.
    const int one_slots = kOneOcc * device_cus();
    const bool one_fits = 2 * (int64_t)a.N * ((a.h + kSBlk - 1) / kSBlk) <= one_slots && stream_cus(s, device_cus()) >= device_cus();
    // (built for dilation <= 2: the single launch needs four workgroups per CU, and at dilation 3 the tile role does not fit 128 VGPRs)
    // Two shapes of the single launch: the SHORT form (4-row tiles, four workgroups per CU, the stream workgroups staying on as the first tile
    // workgroups) while its front half is resident at once; the LONG form (8-row tiles, three per CU, pool workgroups first, nobody waits for a
    // later workgroup) for many instances, where the short form's tile waves would each walk several tiles one after the other.
    const bool whole_device = stream_cus(s, device_cus()) >= device_cus();
    // Measured (2 x 800 x 1024, us per evaluation, one box): 128 instances long form 36.9 vs two launches 37.1, 96: 32.3 vs 31.6 -- no gain
    // (three workgroups per CU slow the front half down by what the kernel boundary costs), so the library takes the long form only where it
    // is asked to (BXI_EVAL_SINGLE_LAUNCH with 8-row tiles) and, with the targets ready (no front half to slow down), from kLongFrom on.
    // With the targets ready the short form has table workgroups of its own at the head of the grid (eval1_kernel<D, R, true>): 14.4 vs 14.6 us
    // for two launches at 32 instances, 18.6 vs 19.0 at 64 (as a duty of a stream wave behind its loads the table came ~5 us into the launch:
    // 15.1 / 19.9).
    // (Measured and dropped: a second launch that reads its predicate words by plain loads ahead of the logits and has sum W up front --
    // 14.85 vs 14.65 at 32 instances, 31.9 vs 31.0 at 128: the tile role is bound by its arithmetic and memory pipeline, not by that hop.)
    // (the long form exists with the targets ready only: with the image side in the launch it measured 36.9 vs 37.1 us for two launches at 128 instances,
    // 32.3 vs 31.6 at 96 -- no gain -- and left the library with ABI 7: BXI_EVAL_SINGLE_LAUNCH | BXI_EVAL_TILE_ROWS_8 without targets runs two launches)
    // (again in round 6, after the second launch had lost 6 us: the long form with the image side 31.0 vs 30.6 us at 128 instances, 27.8 vs 27.2 at 96 -- R6-14)
    const bool long_form = R == 8 && dil <= 2 && ready && ((flags & kFlagSingle) || (a.N >= BXI_KNOB("BXI_LONG_FROM", kLongFrom) && whole_device && !(flags & kFlagShared)));
    // (targets ready: the short single launch while its stream workgroups are a quarter of the slots -- 14.1 vs 14.4-14.9 us for two launches at 32
    // instances; at 64 two launches take 18.3 against 18.7 us, and from kLongFrom on the long form 21.6 against 23.9: R6-14)
    const bool short_ok = one_fits && !(ready && (BXI_KNOB("BXI_READY_TWO", 0) || 4 * (int64_t)a.N * ((a.h + kSBlk - 1) / kSBlk) > one_slots));
    if (env_one && !(flags & kFlagTwo) && (short_ok || env_one == 2 || (flags & kFlagSingle) || long_form) && !head && pooled_in_launch &&
        (R == 4 || long_form) && dil <= 2 && !pr.zero_bit) {
        const int env_one_pool = BXI_KNOB("BXI_ONE_POOL_WGS", 0);
        const int Sn = (a.h + kSBlk - 1) / kSBlk;
        const int n_stream = a.N * Sn;
        const int slots = long_form ? BXI_LONG_OCC * device_cus() : one_slots;
        // the front half (table, stream, pool) should fill the GPU exactly once: a pool workgroup takes several items
        // (measured and dropped: pool workgroups alone filling the GPU first, predicate and stream workgroups behind them -- the
        // stream workgroups, and with them the band flags and the leaders, then end 8 us late: 22.9 us per evaluation against 18.3)
        // (also measured and dropped: stream waves that hold their loads back for 1 - 4 us so that the pool workgroups' reads go
        // first: 18.0 / 18.3 / 20.3 us against 17.96; pool workgroups of one item each: 18.3)
        const int front = slots - n_stream;
        const int room = env_one_pool > 0 ? env_one_pool : (front > slots / 4 ? front : slots / 4);
        int per = (n_items + room - 1) / room;
        if (long_form && per > 2) per = 2;        // (never more than two items per pool workgroup: launch_fused_eval's first launch below)
        const int n_pool = ready ? 0 : (n_items + (per < 1 ? 1 : per) - 1) / (per < 1 ? 1 : per);
        int n_pb = ready ? 0 : (n_items + kWaves - 1) / kWaves;
        if (n_pb > slots / 2) n_pb = slots / 2;
        int64_t n_tb = (eval_cap(a.N, a.h, a.w, dil, R) + kWaves - 1) / kWaves;
        // One tile per wave while the slots last: a wave that walks two tiles runs two ~7 us dependent chains one after the other.  (Round 4
        // capped the tile workgroups at half the slots; they are the LAST workgroups of the grid and wait only for earlier ones, so nothing
        // depends on their number -- 64 instances: 24.1 -> 22.9 us per evaluation, 32 instances unchanged: their 289 were below the cap.)
        const int tb_cap = BXI_KNOB("BXI_ONE_TB_CAP", slots);
        if (n_tb > tb_cap) n_tb = tb_cap;
        // the stream workgroups stay on as the first tile workgroups (only while they leave half of the slots to the rest of the grid)
        // ... unless evaluations run on SEVERAL streams at once: each would hold its stream workgroups' slots while waiting, and three
        // or four of them leave no room for anybody's pool workgroups (measured: 2.4 ms per evaluation with four streams in flight,
        // against 10 us without the staying-on).  The library cannot see what else runs on the device and does not guess: the CALLER
        // says so (BXI_EVAL_SHARED_DEVICE; boxinstseg_amd/functional.py sets it once a second stream has been
        // seen on the device).  A launch that is being captured into a graph may be replayed next to anything: no staying-on either.
        const int env_merge = BXI_KNOB("BXI_ONE_MERGE", 1);
        // ... and only while they are at most a QUARTER of the slots: at 64 instances (448 of 1024) the staying-on costs 0.3 us (21.65 vs 21.35 us, their
        // slots are what the pool workgroups -- three items each then -- are short of); at 32 instances it is worth 0.05 us (R6-12)
        const int merge = env_merge && !long_form && one_fits && 4 * n_stream <= slots && !(flags & kFlagShared) && !stream_is_capturing(s) ? 1 : 0;
        if (merge) n_tb = n_tb > n_stream ? n_tb - n_stream : 0;
        size_t lds = sizeof(double) * (256 + 3 * 64) + sizeof(int) * 4 * 3 * 64;
        if (lds < 8 * (size_t)kWaves * a.w) lds = 8 * (size_t)kWaves * a.w;
        if (lds < sizeof(float) * (size_t)kWaves * (R + 1) * 64) lds = sizeof(float) * (size_t)kWaves * (R + 1) * 64;
        if (lds < 2 * sizeof(float) * (size_t)(a.h + a.w) + 16) lds = 2 * sizeof(float) * (size_t)(a.h + a.w) + 16;
        if (lds <= 36 * 1024) {                             // four workgroups per CU must fit
            const int n_tabw = ready ? ((a.N + 64) / 64 + kWaves - 1) / kWaves : 0;
            const unsigned grid = (unsigned)(n_tabw + n_stream + n_pool + n_pb + 1 + a.N + (int)n_tb + 1);
#define BXI_ONE_CASE(DD)                                                                                                                    \
            case DD:                                                                                                                        \
                if (long_form)                                                                                                              \
                    BXI_LAUNCH("eval1_ready", s, (eval1_kernel<DD, 8, true>), dim3(grid), dim3(256), lds, s, pa, n_pool, n_items, n_pb, (int)n_tb, a, ws, st, vc, \
                               up_prj, up_pw, warmup, pr.n2max, spin_limit, losses, g_logits, vec, merge, ready, key, n_tabw);                  \
                else if (ready)                                                                                                             \
                    BXI_LAUNCH("eval1_ready", s, (eval1_kernel<DD, 4, true>), dim3(grid), dim3(256), lds, s, pa, n_pool, n_items, n_pb, (int)n_tb, a, ws, st, vc, \
                               up_prj, up_pw, warmup, pr.n2max, spin_limit, losses, g_logits, vec, merge, ready, key, n_tabw);                  \
                else                                                                                                                        \
                    BXI_LAUNCH("eval1", s, (eval1_kernel<DD, 4, false>), dim3(grid), dim3(256), lds, s, pa, n_pool, n_items, n_pb, (int)n_tb, a, ws, st, vc, \
                               up_prj, up_pw, warmup, pr.n2max, spin_limit, losses, g_logits, vec, merge, ready, key, n_tabw);                  \
                break;
            switch (dil) { BXI_ONE_CASE(1) BXI_ONE_CASE(2) default: return BXI_ERR_UNSUPPORTED; }
#undef BXI_ONE_CASE
            return check_launch();
        }
    }

    // ---- launch 1 --------------------------------------------------------------------------------------------------
    const int n_tab = ((a.N + 64) / 64 + kWaves - 1) / kWaves;
    const int Sn = (a.h + kSBlk - 1) / kSBlk;
    const int n_stream = a.N * Sn;
    // one item = the 4 input rows of 64 pooled pixels.  The whole launch should be resident at once (5 workgroups per CU at
    // <= 96 VGPRs): a pool workgroup takes several items, the next one's loads in flight, when it is not.
    const int room = env_pool_wgs * device_cus() - n_tab - (head ? 0 : n_stream);
    // ... but never more than two items per pool workgroup: its items are a dependent chain (load -> sums -> Lab -> store, ~2.5 us each), and
    // with many instances -- 896 stream workgroups at 128 -- the few pool workgroups the slots leave would each drag four or five of them
    // behind the HBM stream (the first launch's last 5 us at 128 instances).  Then the launch exceeds the slots and its tail workgroups
    // take them as they come free; the pool workgroups go FIRST in that case, the image side being the longer chain.  Measured: 40.1 ->
    // 38.5 us per evaluation at 128 instances, 75.8 -> 70.3 at 256 (BXI_PREP_ITEMS=<n>: developer override)
    const int env_prep_items = BXI_KNOB("BXI_PREP_ITEMS", 0);
    int per = env_prep_items > 0 ? env_prep_items : (room > 0 ? (n_items + room - 1) / room : 8);
    if (env_prep_items <= 0 && per > 2) per = 2;
    const int n_pool = pooled_in_launch && !ready ? (n_items + (per < 1 ? 1 : per) - 1) / (per < 1 ? 1 : per) : 0;
    PrepTail tl = {};
    tl.ready = ready; tl.key = key;
    const int pool_first = env_pool_first || (!head && n_tab + n_stream + n_pool > env_pool_wgs * device_cus()) ? 1 : 0;
    size_t lds1 = sizeof(double) * (256 + 3 * 64) + sizeof(int) * 4 * 3 * 64;
    // every refusal comes BEFORE the first launch: a refused call has enqueued nothing (callers fall back to other entry points)
    size_t lds2 = sizeof(float) * (size_t)kWaves * (R + 1) * 64;
    const size_t lds_leader = 2 * sizeof(float) * (size_t)(a.h + a.w) + 16;
    if (lds2 < lds_leader) lds2 = lds_leader;
    if (lds2 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
    if (head && head_C != 8 && head_C != 16) return BXI_ERR_UNSUPPORTED;
    if (head) {
        // the head-fused first launch (factor 2, vector rows): tables, pool blocks, head tiles
        if (head->factor != 2 || !(vec & 1) || head->H * 2 != a.h || head->W * 2 != a.w || head->N != a.N || head->B != in->B || !pooled_in_launch)
            return BXI_ERR_UNSUPPORTED;
        const int tiles = ((head->H + kHeadR - 1) / kHeadR) * ((head->W + kYC - 1) / kYC);
        ws.n_cb = (head->H + kHeadR - 1) / kHeadR;
        ws.n_rp = (head->W + kYC - 1) / kYC;
        const size_t lds_head = 8 * 4 * 64 + sizeof(float) * (2 * kHeadR * 64 + 512);
        if (lds1 < lds_head) lds1 = lds_head;
        if (lds1 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
        float* logits_out = const_cast<float*>(a.logits);
        const unsigned grid1 = (unsigned)(n_tab + n_pool + a.N * tiles);
#define BXI_HEAD_LAUNCH(CC, RR)                                                                                                          \
        BXI_LAUNCH("head_prep", s, (head_prep_kernel<CC, RR>), dim3(grid1), dim3(256), lds1, s, pa, n_pool, n_items, a, dil, R, ws, st,     \
                   g_logits, *head, head->params, logits_out, ready, key)
        if (head_C == 16 && head->rel) BXI_HEAD_LAUNCH(16, true);
        else if (head_C == 16) BXI_HEAD_LAUNCH(16, false);
        else if (head_C == 8 && head->rel) BXI_HEAD_LAUNCH(8, true);
        else if (head_C == 8) BXI_HEAD_LAUNCH(8, false);
        else return BXI_ERR_UNSUPPORTED;
#undef BXI_HEAD_LAUNCH
    } else {
        if (lds1 < 8 * (size_t)kWaves * a.w) lds1 = 8 * (size_t)kWaves * a.w;
        if (lds1 > 64 * 1024) return BXI_ERR_UNSUPPORTED;
        BXI_LAUNCH(ready ? "prep_ready" : "prep", s, prep_kernel, dim3((unsigned)(n_tab + n_stream + n_pool)), dim3(256), lds1, s, pa, n_pool, n_items, a, dil,
                   R, ws, st, g_logits, vec, pool_first, tl);
    }
    rc = check_launch();
    if (rc != BXI_OK) return rc;
    // From here on a launch of this evaluation is enqueued: records carrying its tag are (or will be) in the workspace while the epoch only
    // moves with the finisher.  An error return below would leave them behind an unchanged epoch -- the next evaluation would draw the same
    // tag and take them for its own -- so the workspace is returned to its initial state (stream-ordered) on that path.
    auto fail = [&](int code) { (void)hipMemsetAsync(workspace, 0, workspace_bytes, s); (void)hipGetLastError(); return code; };
    if (!pooled_in_launch) {
        rc = launch_pool(batch, a.stride, nullptr, ws.lab_planar, s);
        if (rc != BXI_OK) return fail(rc);
        const int64_t BP = (int64_t)batch->B * a.h * a.w;
        BXI_LAUNCH("pack_lab4", s, pack_lab4_kernel, dim3((unsigned)((BP + 255) / 256 > 2048 ? 2048 : (BP + 255) / 256)), dim3(256), 0, s,
                   (const float*)ws.lab_planar, ws.lab4, (const unsigned int*)ws.epoch, batch->B, (int64_t)a.h * a.w);
        rc = check_launch();
        if (rc != BXI_OK) return fail(rc);
    }

    // ---- launch 2 --------------------------------------------------------------------------------------------------
    const int64_t cap = eval_cap(a.N, a.h, a.w, dil, R);
    int64_t n_tb = (cap + kWaves - 1) / kWaves;
    // the tile list's length is device data: the tile waves stride through it.  The launch should be resident in one round:
    // 4 (R = 4: <= 128 VGPRs) or 2 (R = 8) workgroups per CU; the predicate waves are short-lived.
    const int env_pair_wgs = BXI_KNOB("BXI_PAIR_WGS_PER_CU", 0);
    const int occ = env_pair_wgs > 0 ? env_pair_wgs : (R == 4 ? (dil <= 2 ? 4 : 3) : (dil <= 2 ? 3 : 2));
    // (the leaders are short-lived and are not counted; with them subtracted, 512 instances at two workgroups per CU left ONE
    // predicate workgroup for the whole image side: 4.4 ms per evaluation)
    const int cus2 = stream_cus(s, device_cus());                      // a CU-masked stream has fewer
    const int slots = occ * cus2 > 64 ? occ * cus2 : 64;
    int n_pb = ready ? 0 : (n_items + kWaves - 1) / kWaves;  // (targets ready: the image side is in memory at this kernel's start)
    if (n_pb > slots / 2) n_pb = slots / 2;     // (5 / 8 of the slots -- every item a wave of its own at 2 x 800 x 1024 and three per CU --: no difference, R6-10)
    // (the predicate workgroups are short-lived: the tile workgroups behind them in the grid take their slots as they leave, so the
    // tile workgroups are sized for the slots, not for what the predicate workgroups leave over -- BXI_PAIR_TB_FULL=0: the round-3 sizing)
    const int env_tb_full = BXI_KNOB("BXI_PAIR_TB_FULL", 1);
    if (n_tb > (env_tb_full ? slots : slots - n_pb)) n_tb = env_tb_full ? slots : slots - n_pb;
    const int grid = n_pb + 1 + a.N + (int)n_tb + 1;      // predicate blocks + the reducer + leaders + tile blocks + the finisher
#define BXI_PAIR_CASE(DD)                                                                                                                  \
    case DD:                                                                                                                               \
        if (R == 4) launch_pair<DD, 4>(s, grid, lds2, a, warmup, pr.n2max, pr.zero_bit, n_pb, n_items, spin_limit, vc, ws, st, losses, g_logits, up_prj, up_pw); \
        else launch_pair<DD, 8>(s, grid, lds2, a, warmup, pr.n2max, pr.zero_bit, n_pb, n_items, spin_limit, vc, ws, st, losses, g_logits, up_prj, up_pw);        \
        break;
    switch (dil) {
        BXI_PAIR_CASE(1) BXI_PAIR_CASE(2) BXI_PAIR_CASE(3) BXI_PAIR_CASE(4)
        default: return fail(BXI_ERR_UNSUPPORTED);
    }
#undef BXI_PAIR_CASE
    rc = check_launch();
    return rc == BXI_OK ? rc : fail(rc);
}

// the same from the three numbers the kernel needs of the instances (the autograd node's backward keeps those, not the structs)
int launch_rescale_nhw(int N, int h, int w, const float* g_prj, const float* g_pw, int dil, const void* state, float* g_logits, void* stream) {
    if (N < 0 || h <= 0 || w <= 0) return BXI_ERR_BAD_SHAPE;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (N == 0) return BXI_OK;
    if (!g_prj || !g_pw || !state || !g_logits) return BXI_ERR_NULL_POINTER;
    if (N > 65535) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
    InstArgs a = {};
    a.N = N; a.h = h; a.w = w;
    LossState st = {};
    carve_state(const_cast<void*>(state), N, h, w, &st);
    hipStream_t s = as_stream(stream);
    BXI_LAUNCH("rescale", s, rescale_kernel, dim3(8, N), dim3(256), 0, s, a, dil, st, g_prj, g_pw, g_logits);
    return check_launch();
}

int launch_rescale(const bxi_instances* in, const float* g_prj, const float* g_pw, int dil, const void* state, float* g_logits, void* stream) {
    InstArgs a;
    int rc = fill_inst(in, a);
    if (rc != BXI_OK) return rc;
    if (!fused_eval_supported(dil)) return BXI_ERR_UNSUPPORTED;
    if (a.N == 0) return BXI_OK;
    if (!g_prj || !g_pw || !state || !g_logits) return BXI_ERR_NULL_POINTER;
    if (a.N > 65535) return BXI_ERR_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(state) & 255) return BXI_ERR_WORKSPACE;
    LossState st = {};
    carve_state(const_cast<void*>(state), a.N, a.h, a.w, &st);
    hipStream_t s = as_stream(stream);
    BXI_LAUNCH("rescale", s, rescale_kernel, dim3(8, a.N), dim3(256), 0, s, a, dil, st, g_prj, g_pw, g_logits);
    return check_launch();
}

}  // namespace bxi

#ifdef BXI_WAITLOG
extern "C" int bxi_debug_waitlog(unsigned int* out16, int reset) {   // developer builds only
    int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(bxi::g_waitlog), 16 * sizeof(unsigned int));
    if (rc == 0 && reset) { unsigned int z[16] = {0}; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_waitlog), z, sizeof(z)); }
    return rc;
}
#endif
#ifdef BXI_ABLATE
extern "C" int bxi_debug_set_ablate(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_ablate), &bits, sizeof(bits)); }
#endif
#ifdef BXI_TRACE
extern "C" int bxi_debug_set_trace2(void* buf) {   // developer builds only
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bxi::g_trace), &buf, sizeof(buf));
}
#endif
