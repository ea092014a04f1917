// a3_api.hip -- host side of libaruco3_hip.so: context, device pools, the batch pipeline and the C ABI
// declared in include/aruco3_hip.h.  No CPU fallback of any stage lives here: without a HIP device
// a3_create fails with A3_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "a3_common.h"
#include "a3_internal.h"

namespace a3 {
// k_threshold.hip
hipError_t launch_grey_threshold(hipStream_t, const uint8_t*, int, size_t, size_t, int, int, uint32_t, uint32_t, uint8_t*, uint64_t*, uint16_t*);
void set_k1_waves(int);
bool threshold_writes_grey_plane(uint32_t radius, const uint8_t* pixels, size_t row_stride, size_t frame_stride, int W);
void set_k1_cus(int);
bool k1_build_is_default();
// k_contours.hip
hipError_t launch_dart_count(hipStream_t, const uint64_t*, int, int, uint32_t, uint32_t, unsigned long long*, uint32_t*, uint64_t, uint32_t*, uint32_t*, void*, size_t);
size_t tile_darts_bytes(uint32_t W, uint32_t H, uint32_t n_frames);
size_t tile_off_offset(uint32_t W, uint32_t H, uint32_t n_frames);
size_t tile_mask_offset_bytes(uint32_t W, uint32_t H, uint32_t n_frames);
hipError_t launch_dart_build(hipStream_t, const uint64_t*, int, int, uint32_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t*, const uint32_t*,
                             uint64_t*, uint32_t*, uint32_t, const uint32_t*, int, const unsigned long long*);
hipError_t launch_zero(hipStream_t, void*, size_t);
size_t entry_state_bytes();
size_t fin_state_bytes();
size_t entry_slots(uint32_t);
size_t leader_list_bytes(uint32_t);
hipError_t launch_rank_cycles(hipStream_t, uint32_t, int, const uint64_t*, const uint32_t*, JumpState*,
                              uint32_t*, unsigned int*, void*, void*, void*, uint32_t*, unsigned int*, int, DeviceCounters*, const uint32_t*, int,
                              const uint32_t*, uint32_t*, uint32_t, int, uint32_t, unsigned int*, int);
hipError_t launch_resolve(hipStream_t, const JumpState*, const void*, uint32_t, int, const uint64_t*, const uint32_t*, const unsigned int*, uint64_t*, uint64_t*,
                          DeviceCounters*, int, const uint32_t*);
hipError_t launch_select_scatter(hipStream_t, const JumpState*, const void*, uint32_t, const uint32_t*, const unsigned int*, const uint32_t*, const uint64_t*,
                                 const uint32_t*, uint32_t, uint32_t,
                                 uint32_t, double, double, ContourRec*, uint32_t*, uint32_t, uint64_t, DeviceCounters*,
                                 const uint64_t*, uint32_t*, const uint32_t*, int, uint32_t*, int);
hipError_t launch_debug_clockwise(hipStream_t, const int32_t*, uint32_t, int32_t*);
hipError_t launch_unpack_bits(hipStream_t, const uint64_t*, int, int, uint8_t*);
hipError_t launch_contour_quads(hipStream_t, const ContourRec*, const DeviceCounters*, uint32_t, const uint32_t*, double, uint32_t, uint32_t,
                                uint32_t, CandRec*, uint32_t*, unsigned int*, int, uint32_t);
// k_decode.hip
size_t decode_out_bytes();
hipError_t launch_frame_candidates(hipStream_t, const CandRec*, const uint32_t*, uint32_t, uint32_t, float, uint16_t*, uint16_t*, uint32_t*,
                                   uint32_t*, unsigned int*, uint32_t, void*, float*);
uint32_t frame_cand_lds_slots();
size_t proj_rec_bytes();
size_t weight_table_bytes();
hipError_t launch_weight_table(hipStream_t, uint32_t, uint32_t, uint32_t, float*);
hipError_t launch_decode(hipStream_t, PixelSrc, int, int, uint32_t, const uint16_t*, const uint32_t*, const unsigned int*, uint32_t,
                         uint32_t, uint32_t, uint32_t, const uint64_t*, uint32_t, uint32_t, int, void*, const float*, void*, uint8_t*, uint32_t, uint32_t*, int, int, int);
hipError_t launch_compact_markers(hipStream_t, const void*, const uint16_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, a3_marker*,
                                  uint32_t, uint32_t*, unsigned int*, unsigned int*, const uint32_t*, unsigned int*);
hipError_t launch_pack_detections(hipStream_t, const a3_marker*, const a3_pose*, const uint32_t*, uint32_t, uint32_t, uint32_t, void*, unsigned int*);
hipError_t launch_debug_rotate_bits(hipStream_t, const uint8_t*, uint32_t, uint32_t, uint8_t*);
hipError_t launch_gather_patches(hipStream_t, const void*, uint32_t, const uint8_t*, uint32_t, uint32_t, uint8_t*, unsigned int*);
hipError_t launch_pose(hipStream_t, const uint32_t*, uint32_t, const float*, uint32_t, const unsigned int*, int, float, float, float, float,
                       float, float, float, a3_pose*);
hipError_t launch_find_nearest(hipStream_t, const uint64_t*, uint32_t, const uint64_t*, uint32_t, uint32_t*, uint8_t*);
hipError_t launch_calc_tau(hipStream_t, const uint64_t*, uint32_t, unsigned int*);
hipError_t launch_synth_render(hipStream_t, const a3_synth_frame*, uint32_t, const a3_synth_marker*, uint32_t, uint32_t, int, float, float, int,
                               uint8_t*, size_t, size_t);
hipError_t launch_spin(hipStream_t, int, int, int, uint32_t*);
hipError_t launch_selftest(hipStream_t, const double*, const double*, uint32_t, double*, double*, float*, float*);
}  // namespace a3

using namespace a3;

namespace {

// message of the last failed context-less call (a3_create, a3_calculate_tau) ON THIS THREAD: contexts for different GPUs may be
// created concurrently from different threads
thread_local std::string g_create_error;

// Quad candidates kept per frame.  The reference is unbounded (src/aruco.rs:124-166 pushes into a Vec); here the tables start at
// kMaxCandDefault per frame and a batch that overflows them is re-run with tables twice the size (1024, 2048, 4096, 6144 -- as far as the
// per-frame ordering + discard_too_near kernel keeps a frame's candidates in LDS, 21 bytes each -- then 12 288 ... through memory), up
// to kMaxCandLimit = 65 536, where a3_marker.candidate_index (16 bits) ends; beyond that: A3_ERR_LIMIT.
constexpr uint32_t kMaxCandDefault = 1024, kMaxCandLimit = A3_MAX_CANDIDATES_PER_FRAME;
static_assert(kMaxCandLimit == 65536, "a3_marker.candidate_index is a uint16_t");
constexpr uint32_t kMaxContoursDefault = 1u << 20;
constexpr uint64_t kMaxDartsDefault = 48ull << 20;
constexpr uint64_t kMaxPointsDefault = 64ull << 20;
constexpr uint64_t kHardMaxDarts = 3ull << 30;   // 32-bit dart indices
constexpr uint64_t kHardMaxPoints = 3ull << 30;
constexpr int kResolveItersMax = 16;        // == DeviceCounters::resolve_changed slots
// debug taps: warped patches kept per batch (one per candidate that reaches the decode stage).  The tap holds a patch for every
// candidate the batch can have (frames x kMaxCand) up to kPatchCapMax patches (2.4 GB at 49 x 49); a binding that populates
// Detection.homographies for more than 1024 frames per call splits the call (integration/aruco3_hip.rs does).
constexpr uint32_t kPatchCapMin = 32768, kPatchCapMax = 1u << 20;

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Chunk { uint32_t first, count; uint64_t darts; uint32_t max_frame_darts; };
// The second half of a batch (candidates -> markers -> read-back), kept aside when its enqueue is deferred: everything
// enqueue_back needs besides the context's buffers.
struct BackArgs {
    uint32_t n = 0, W = 0, H = 0, S = 0, max_cand = 0, patch_cap = 0, marker_cap = 0, guess = 0;
    float min_corner_separation = 0.0f;
    PixelSrc src{};
    size_t head_bytes = 0, pose_bytes = 0;
    bool taps = false, want_pose = false, pose_has_intr = false;
    float pose_size_mm = 0.0f;
    a3_intrinsics pose_intr{};
    int profiling = 0;
};

// what finish_batch needs to know about the batch enqueue_batch put on the stream
struct Pending {
    bool active = false, device_plan = false;
    size_t n_chunks = 0, ctr_bytes = 0, head_pad = 0, pose_bytes = 0;
    uint64_t chunk0_darts = 0;
    uint32_t marker_cap = 0, guess = 0, n = 0, W = 0, H = 0;
    int rounds_max = 0, profiling = 0;
    bool taps = false;   // debug taps were on: the per-frame candidate counts came back with the results
    // the submitted call, for the synchronous re-run when the device asks for one
    const uint8_t* pixels = nullptr; int fmt = 0; size_t row_stride = 0, frame_stride = 0;
    bool want_pose = false;   // a3_detect_batch_pose_submit: the re-run must solve the poses again
};

}  // namespace

struct a3_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // host frames (A3_MEM_HOST) are copied on the device's copy stream (shared by the contexts of a device, see DeviceStreams), the
    // compute stream waits for the copy through ev_in: with two contexts in flight (submit / collect) the H2D of batch i+1 runs
    // under the kernels of batch i
    hipEvent_t ev_in = nullptr;
    // Deferred decode (submit / collect with more than one context, see enqueue_batch): the decode stage of a submitted batch
    // runs on the device's decode stream, released from inside the launch sequence of the NEXT submitted batch, so that it shares
    // the GPU with that batch's contour stage (both are latency-bound and leave the chip mostly idle) instead of standing in line.
    hipEvent_t ev_contours = nullptr, ev_k1 = nullptr, ev_k1_ready = nullptr, ev_k1_done = nullptr, ev_gate = nullptr;
    bool k1_marked = false;          // ev_k1_done was recorded behind the threshold kernel of the batch in flight
    // Bursts (a3_order_after): a context that declared gates since its last submit is a member of a burst that is not the last
    // one: its submit enqueues the threshold kernel only and HOLDS the rest (contour stage ... read-back) until the burst's last
    // member -- the first submit without gates on the device -- has enqueued its threshold kernel; the held chains are then
    // enqueued behind that kernel.  The threshold kernels of a burst so run back to back with nothing in between.
    bool gates_declared = false;     // a3_order_after was called since the last submit
    bool rest_held = false;          // guarded by g_defer_mu: the chain of the submitted batch has not been enqueued yet
    int held_rc = 0;                 // guarded by g_defer_mu: what enqueueing the held chain returned, whoever did it
    int front_prof = 0;              // profiling level in force for the batch whose front half has been enqueued
    size_t held_out_cap = 0;
    bool back_deferred = false;      // guarded by g_defer_mu
    int back_rc = 0;                 // a failed launch of the deferred half, whoever enqueued it (guarded by g_defer_mu): collect reports it
    bool allow_defer = false;        // set by the submit entry points for the batch being enqueued
    int batch_mode = 0;              // where the decode stage of the batch being submitted is released (batch_mode_of, fixed at submit)
    std::atomic<uint32_t> stepping{0};   // A3_STEP_* of the batch in flight / last finished (a3_stats.stepping); another thread's submit may release this context's held chain
    uint32_t released_others = 0;    // held chains of other contexts this batch's submit released (the burst's last member)
    uint32_t reruns = 0;             // synchronous re-runs the device asked for while the last call's batch was produced (pool growth, more passes, host plan)
    BackArgs back;
    a3_config cfg{};
    uint8_t num_bits = 0, tau = 0;
    uint32_t n_codes = 0, mark_size = 0;
    std::string err;

    uint64_t max_darts = kMaxDartsDefault, max_points = kMaxPointsDefault;
    uint32_t max_contours = kMaxContoursDefault;
    uint32_t max_cand = kMaxCandDefault;   // candidate slots per frame (grows on overflow, see kMaxCandLimit)
    // launch-count hints (every pass past convergence is an empty launch of ~5 us): start low, retry the batch with the
    // maximum if a pass count turns out too small
    int jump_rounds_hint = 10, resolve_iters_hint = 4;
    uint32_t dbg_nd = 0, dbg_frames = 0, dbg_chunks = 0;
    PixelSrc dbg_src{};
    // a3_debug_inject_candidates (tests only): quads that replace frame 0's candidate list of the next batch, between the contour
    // stage and k_frame_candidates (quirk Q4: a degenerate quad cannot come out of a convex hull)
    std::vector<CandRec> inject;
    bool inject_armed = false;
    uint32_t inject_count = 0;
    Pending pending;
    bool pending_trivial = false;   // a submitted batch with no frames / empty images   // a3_debug_kernel_time: shape of the last batch's contour graph
    int resolve_full_ttl = 0;
    int entry_global_ttl = 0;   // > 0: a recent batch had a frame whose entry list did not fit LDS: use the global doubling rounds
    // device-side planning: the previous batch of this shape fitted one chunk with plan_darts darts, so this one is enqueued
    // without reading the dart counts back first (k_plan); an overflow falls back to the host plan once (force_host_plan)
    bool plan_valid = false, force_host_plan = false;
    uint32_t plan_n = 0, plan_W = 0, plan_H = 0;
    uint64_t plan_darts = 0;   // > 0: launch the fixpoint passes over all darts too (a recent batch needed them); else only k_resolve_fast
    // a3_detect_batch_pose: poses of every marker are computed on the device right after detection
    bool want_pose = false;
    float pose_size_mm = 0.0f;
    bool pose_has_intr = false;
    a3_intrinsics pose_intr{};
    a3_pose* pose_out = nullptr;
    bool debug_taps = false;
    uint32_t patch_cap = 0;    // patches the tap of the last tapped batch could hold
    bool grey_valid = false;   // the last batch wrote the grey plane
    // a3_download_contours: the last batch ran with debug taps in one chunk, so its contour table and point pool are whole
    bool contours_valid = false;
    uint32_t tap_contours = 0; uint64_t tap_points = 0;
    // a3_pack_detections: the marker list of the last finished batch is still on the device
    bool markers_valid = false;
    uint32_t last_n = 0, last_max_per_frame = 0;
    int profiling = 0;   // 0 off, 1 threshold stage only, 2 every stage (an event record between kernels costs ~6 us of device time)
    int profile_every = 1;   // threshold-only mode: time every k-th batch (A3_PROFILE_THRESHOLD_SAMPLED: 4)
    uint32_t batch_seq = 0;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double prof_ms[A3_STAGE_COUNT] = {0, 0, 0};
    uint64_t prof_n[A3_STAGE_COUNT] = {0, 0, 0};
    a3_stats stats{};

    // last batch geometry (for the debug downloads)
    uint32_t W = 0, H = 0, frames = 0;

    DevBuf dict, in, grey, bin, frame_darts, frame_darts_dev, frame_base, pix_base, tile_darts;
    DevBuf d_xy, d_succ, stA, stB, t_cur, t_next;
    DevBuf leader_list, leader_keep, entry_list, es_a, es_b;
    DevBuf contours, cyc_start_off, points;
    DevBuf cands, pre_xy, fin_xy, fin_count, work, outs, proj, patches, cand_big;
    // one allocation zeroed by one memset per batch and read back with one copy: [scratch 256 B | counters | per_frame | frame_cursor | cand_count]
    DevBuf zero_blk;
    a3_marker* markers_ptr = nullptr;                // the compacted marker list, right behind the read-back head in the zero block
    unsigned long long* frame_darts_ptr = nullptr;   // frame_darts_dev (device plan: kept zero by its reader) or the frame_darts buffer (host plan)
    unsigned int* scratch_u32 = nullptr; DeviceCounters* counters = nullptr; uint32_t* per_frame = nullptr; uint32_t* frame_cursor = nullptr; uint32_t* cand_count = nullptr;
    uint32_t last_marker_total = 0;   // sizes the speculative marker read-back of the next batch
    DevBuf tmp_a, tmp_b, tmp_c, tmp_d;
    DevBuf hsum;                // row sums of the grey plane (u16): the separable threshold path only (windows above 15)
    DevBuf wtab;                // triangle-resize weights of a full patch (sample -> mark_size), written once at a3_create
    DevBuf pose_buf;            // a3_detect_batch_pose: both poses of every marker of the last batch (kept for a3_pack_detections)
    bool poses_valid = false;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // debug taps: per-frame candidate counts of the last batch (before / after discard_too_near), read back with the results so
    // that a3_candidate_count and the a3_download_* calls that start with it need no device round trip of their own
    void* pinned_counts = nullptr;
    size_t pinned_counts_cap = 0;
    std::vector<uint32_t> h_cand_pre, h_cand_fin;
    bool counts_valid = false;
};

namespace {

int fail(a3_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define A3_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return fail(ctx, A3_ERR_HIP, #call, e_);      \
    } while (0)

// Waiting for a batch that takes about a millisecond: a blocking hipStreamSynchronize wakes the host tens of microseconds
// late, so poll first (with the CPU's spin-wait hint between polls, which leaves the core's other hardware thread its share)
// and block once the work has proved long: after 2 ms the wake-up delay no longer matters.
inline void spin_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
constexpr auto kSpinBudget = std::chrono::milliseconds(2);

hipError_t wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > kSpinBudget) return hipStreamSynchronize(st);
        spin_pause();
    }
}

hipError_t wait_event(hipEvent_t ev, hipStream_t st) {
    (void)hipStreamQuery(st);   // makes the runtime hand everything queued so far to the GPU; the event poll alone may not
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > kSpinBudget) return hipEventSynchronize(ev);
        spin_pause();
    }
}

int ensure_pinned(a3_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_cap) return A3_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr; ctx->pinned_cap = 0;
    A3_HIP(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_cap = bytes;
    return A3_OK;
}

uint32_t mark_size_of(uint8_t num_bits) {  // src/dictionaries.rs:154-156
    return (uint32_t)((uint8_t)std::ceil(std::sqrt((float)num_bits)) + 2);
}

int ensure_dart_pool(a3_ctx* ctx, uint64_t darts) {
    A3_HIP(ctx->d_xy.ensure(darts * 8));   // dart records (dart_rec)
    A3_HIP(ctx->d_succ.ensure(darts * 4));
    A3_HIP(ctx->stA.ensure(darts * sizeof(JumpState)));
    A3_HIP(ctx->stB.ensure(darts * fin_state_bytes()));   // the 8-byte final states (leader, hops | flags) of k_jump_finalize
    A3_HIP(ctx->t_cur.ensure(darts * 8));
    A3_HIP(ctx->t_next.ensure(darts * 8));
    A3_HIP(ctx->leader_list.ensure(leader_list_bytes((uint32_t)darts)));   // leaders of cycles with a start event, 16 shards
    A3_HIP(ctx->leader_keep.ensure(leader_list_bytes((uint32_t)darts)));   // k_cycle_select: pass-1 verdict per leader slot
    const size_t eslots = entry_slots((uint32_t)darts);   // sharded slot space: darts + at most 16 tiles of padding
    A3_HIP(ctx->entry_list.ensure(eslots * 4));
    // entries are darts whose predecessor lies in another 2048-dart tile; the bound darts is never reached in practice,
    // but an adversarial image can come close, so size for it
    A3_HIP(ctx->es_a.ensure(eslots * entry_state_bytes()));
    A3_HIP(ctx->es_b.ensure(eslots * entry_state_bytes()));
    return A3_OK;
}

// ---- streams ----
// A process has few hardware queues (the runtime multiplexes its streams onto four by default), and two streams that land on one
// queue run in order whatever their events say: a decode stage "released beside the next batch" then simply stands in that
// batch's line -- measured: a second set of contexts, each with three streams of its own, lost 10 % where the first set gained
// 7 %.  So streams are few: ONE decode stream and ONE copy stream per device, shared by all contexts (their work never wants to
// overlap with itself), created on first use; a context's own stream exists only if the caller never passed one (a3_set_stream).
struct DeviceStreams { hipStream_t decode = nullptr, copy = nullptr, k1 = nullptr; };
std::mutex g_streams_mu;
DeviceStreams g_dev_streams[64];
bool g_decode_low_prio = false;   // (see a3_debug_set_overlap)
// CU partition (a3_internal.h: a3_debug_set_partition): the threshold kernel of every batch on a device-wide stream restricted to
// g_part_k1_cus compute units, everything else on streams restricted to the others.  0 = off.
int g_part_k1_cus = 0, g_part_pattern = 0;
// a3_debug_set_k1_stream (measurement aid): 0 off; 1 / 2: the threshold kernel of every batch on a device-wide stream of the LOWEST /
// HIGHEST priority (no CU mask), ordered against the context's stream by two events
int g_k1_stream_prio = 0;
bool g_hold_rests = true;        // a3_debug_set_hold: bursts hold their chains back (see submit_common); 0 for A/B
bool g_mark_threshold = false;   // a3_debug_set_mark_threshold: record an event behind every threshold kernel (costs ~2 % of a step: tools/spin_probe.py)
enum { kStreamCopy = 0, kStreamDecode = 1, kStreamK1 = 2 };

// CU masks of the partition: 256 bits, bit i = compute unit i as the runtime numbers them
void partition_masks(uint32_t k1[8], uint32_t rest[8]) {
    for (int i = 0; i < 8; i++) { k1[i] = 0u; rest[i] = 0u; }
    for (int cu = 0; cu < 256; cu++) {
        bool to_k1;
        if (g_part_pattern == 0) to_k1 = cu < g_part_k1_cus;                                   // the first k
        else to_k1 = ((cu % 16) * g_part_k1_cus) / 256 != (((cu % 16) + 1) * g_part_k1_cus) / 256;   // k/256 of every group of 16
        (to_k1 ? k1 : rest)[cu >> 5] |= 1u << (cu & 31);
    }
}

hipError_t create_stream(hipStream_t* st, int role /* 0 rest, 1 k1 */, int priority) {
    if (g_part_k1_cus > 0) {
        uint32_t k1[8], rest[8];
        partition_masks(k1, rest);
        return hipExtStreamCreateWithCUMask(st, 8, role ? k1 : rest);
    }
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, priority);
}

hipError_t device_stream(int device, int kind, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    DeviceStreams& ds = g_dev_streams[device & 63];
    hipStream_t& st = kind == kStreamDecode ? ds.decode : (kind == kStreamK1 ? ds.k1 : ds.copy);
    if (!st) {
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int prio = kind == kStreamDecode && g_decode_low_prio ? lo : (kind == kStreamK1 && g_k1_stream_prio == 1 ? lo : (kind == kStreamK1 && g_k1_stream_prio == 2 ? hi : 0));
        if (e == hipSuccess) e = create_stream(&st, kind == kStreamK1 ? 1 : 0, prio);
        if (e != hipSuccess) { st = nullptr; return e; }
    }
    *out = st;
    return hipSuccess;
}

// the stream a context enqueues on: the caller's (a3_set_stream) or, created on first need, its own
int need_stream(a3_ctx* ctx);

// ---- deferred decode: contexts whose submitted batch has its contour stage enqueued and its decode stage not yet ----
std::mutex g_defer_mu;
std::vector<a3_ctx*> g_deferred;
// 0: no deferral; 1: release a waiting decode stage behind the next batch's threshold kernel; 2 (default): behind the next batch's
// k_local_contract -- the kernels that follow it (entry resolution, finalize, scatter, quads) are latency-bound like the decode
// stage and share the chip with it, whereas the dart kernels before it are bound by VALU and LDS throughput and only get slower
// in company.  Measured in one process (tools/attic/ab_overlap.py, BASELINE config 2, two contexts): 0.820 / 0.769 / 0.762 ms per step
// for modes 0 / 1 / 2 (0.794 / 0.744 / 0.741 on another box); smaller decode grids (2048 ... 512 workgroups) only lose.  Deferring the second half of the contour stage as well (entry resolution ... quads, released with the
// decode stage behind the next threshold kernel) was built and measured: 0.777 with two contexts, 0.821 with three -- dropped.
// (a3_debug_set_overlap in a3_internal.h switches modes for the A/B measurements of tools/.)
// Since round 5 the mode is decided PER BATCH, by the library, from what the caller did through the public header -- no switch
// selects the stepping any more (g_overlap_force == -1, the default):
//   * a context whose stream no other context of the device uses never defers (mode 0): consecutive batches on such contexts
//     overlap by themselves, and a context that declared burst gates (a3_order_after) holds its chain back (submit_common);
//   * contexts that SHARE one stream (a3_set_stream with the same caller stream) run in order there, whatever they do: the
//     deferred decode stage -- mode 2 -- is the only overlap there is, so they get it.
// a3_debug_set_overlap(0 | 1 | 2) forces one mode on every batch of the process for the A/B measurements of tools/; forced modes
// 1 and 2 also switch the burst hold off, which is what the library did by default until round 4.
std::atomic<int> g_overlap_force{-1};
// the mode of the batch `ctx` is about to submit
std::vector<a3_ctx*> g_contexts;   // every live context of the process (guarded by g_streams_mu)
int batch_mode_of(const a3_ctx* ctx) {
    const int f = g_overlap_force.load(std::memory_order_relaxed);
    if (f >= 0) return f;
    if (!ctx->stream || ctx->stream == ctx->own_stream) return 0;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    for (const a3_ctx* o : g_contexts)
        if (o != ctx && o->device == ctx->device && o->stream == ctx->stream) return 2;
    return 0;
}
// where a batch being enqueued releases the decode stages other contexts have deferred
int release_mode() {
    const int f = g_overlap_force.load(std::memory_order_relaxed);
    return f >= 0 ? f : 2;
}
// (The decode stream has default priority: the lowest one measured the same, and is the wrong thing to hold when two processes
// share a GPU.)

// candidates -> markers -> read-back of one batch, on stream `st` (the context's stream, or its decode stream when deferred)
int enqueue_back(a3_ctx* ctx, hipStream_t st, const BackArgs& b) {
    unsigned int* d_work_count = ctx->scratch_u32 + 0;
    unsigned int* d_marker_total = ctx->scratch_u32 + 1;
    unsigned int* d_err = ctx->scratch_u32 + 4;
    if (ctx->inject_armed) {   // (test hook; never set by a caller of the public header)
        ctx->inject_armed = false;
        const uint32_t cnt = (uint32_t)ctx->inject.size();
        if (cnt) A3_HIP(hipMemcpyAsync(ctx->cands.p, ctx->inject.data(), cnt * sizeof(CandRec), hipMemcpyHostToDevice, st));
        ctx->inject_count = cnt;
        A3_HIP(hipMemcpyAsync(ctx->cand_count, &ctx->inject_count, 4, hipMemcpyHostToDevice, st));
    }
    A3_HIP(launch_frame_candidates(st, ctx->cands.as<CandRec>(), ctx->cand_count, b.n, b.max_cand, b.min_corner_separation,
                                   ctx->pre_xy.as<uint16_t>(), ctx->fin_xy.as<uint16_t>(), ctx->fin_count.as<uint32_t>(),
                                   ctx->work.as<uint32_t>(), d_work_count, b.S, ctx->proj.p, ctx->cand_big.as<float>()));
    // few frames (small batches): all four waves of a workgroup run the stages behind the sampling.  (Round 5 tried folding the marker
    // gather into k_decode for one-frame calls -- its last workgroup, found by a ticket -- to save a launch: the call got 5 us SLOWER,
    // 166 against 161 us in alternating runs, a fence + ticket per workgroup and a serial tail costing more than the launch; removed.)
    const int few = b.n <= 64u ? 1 : 0;
    A3_HIP(launch_decode(st, b.src, (int)b.W, (int)b.H, 0, ctx->fin_xy.as<uint16_t>(), ctx->work.as<uint32_t>(), d_work_count,
                         b.max_cand, b.S, ctx->mark_size, b.S, ctx->dict.as<uint64_t>(), ctx->n_codes, ctx->tau, ctx->cfg.filter_high_bit_errors,
                         ctx->proj.p, ctx->wtab.as<float>(), ctx->outs.p, b.taps ? ctx->patches.as<uint8_t>() : nullptr, b.patch_cap, ctx->per_frame,
                         (int)std::min<uint32_t>(4096u, b.n * 128u) /* (grid-stride over the work list; 4096 workgroups that find nothing cost a one-frame call ~4 us) */, 0,
                         few));
    A3_HIP(launch_compact_markers(st, ctx->outs.p, ctx->fin_xy.as<uint16_t>(), ctx->fin_count.as<uint32_t>(), b.n, 0, b.max_cand,
                                  ctx->markers_ptr, b.marker_cap, ctx->per_frame, d_marker_total, d_err, ctx->cand_count, ctx->scratch_u32 + 2));
    if (b.want_pose) {   // IPPE on the device-resident marker list (src/pose.rs:52-81), no extra round trip
        const a3_intrinsics& in = b.pose_intr;
        A3_HIP(launch_pose(st, reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(ctx->markers_ptr) + offsetof(a3_marker, corners)),
                           (uint32_t)(sizeof(a3_marker) / 4), nullptr, b.marker_cap, d_marker_total, b.pose_has_intr ? 1 : 0, b.pose_size_mm,
                           (float)b.W, (float)b.H, in.focal_x, in.focal_y, in.principal_x, in.principal_y, ctx->pose_buf.as<a3_pose>()));
    }
    if (b.profiling >= 2) A3_HIP(hipEventRecord(ctx->ev[3], st));
    // ---- results: one copy of [scratch | counters | per-frame counts | `guess` markers], then the poses and (taps) the counts ----
    uint8_t* hp = (uint8_t*)ctx->pinned;
    a3_pose* h_poses = reinterpret_cast<a3_pose*>(hp + b.head_bytes + (size_t)b.guess * sizeof(a3_marker));
    A3_HIP(hipMemcpyAsync(hp, ctx->scratch_u32, b.head_bytes + (size_t)b.guess * sizeof(a3_marker), hipMemcpyDeviceToHost, st));
    if (b.pose_bytes) A3_HIP(hipMemcpyAsync(h_poses, ctx->pose_buf.p, (size_t)b.guess * b.pose_bytes, hipMemcpyDeviceToHost, st));
    if (b.taps) {   // Detection.candidates / .homographies will be asked for frame by frame: their counts travel now
        A3_HIP(hipMemcpyAsync(ctx->pinned_counts, ctx->cand_count, (size_t)b.n * 4, hipMemcpyDeviceToHost, st));
        A3_HIP(hipMemcpyAsync((uint8_t*)ctx->pinned_counts + (size_t)b.n * 4, ctx->fin_count.p, (size_t)b.n * 4, hipMemcpyDeviceToHost, st));
    }
    A3_HIP(hipEventRecord(ctx->ev[4], st));
    return A3_OK;
}

// Enqueue the deferred second half of `ctx`'s batch on its decode stream: after its own contour stage and, when `after` is
// given, after that event (the threshold kernel of the batch another context has just submitted).  g_defer_mu is held.
int flush_deferred_impl(a3_ctx* ctx, hipEvent_t after);
int flush_deferred_locked(a3_ctx* ctx, hipEvent_t after) {
    const int rc = flush_deferred_impl(ctx, after);
    if (rc) ctx->back_rc = rc;   // the owner may be another thread's context: its collect must not read a batch that never ran
    return rc;
}
int flush_deferred_impl(a3_ctx* ctx, hipEvent_t after) {
    if (!ctx->back_deferred) return A3_OK;
    ctx->back_deferred = false;
    for (size_t i = 0; i < g_deferred.size(); i++)
        if (g_deferred[i] == ctx) { g_deferred.erase(g_deferred.begin() + (long)i); break; }
    hipStream_t ds = nullptr;
    A3_HIP(device_stream(ctx->device, kStreamDecode, &ds));
    A3_HIP(hipStreamWaitEvent(ds, ctx->ev_contours, 0));
    if (after) A3_HIP(hipStreamWaitEvent(ds, after, 0));
    if (int rc = enqueue_back(ctx, ds, ctx->back)) return rc;
    (void)hipStreamQuery(ds);   // hands what was just queued to the GPU now (the owner polls an event, not this stream)
    return A3_OK;
}

int need_stream(a3_ctx* ctx) {
    if (ctx->stream) return A3_OK;
    if (!ctx->own_stream) {
        A3_HIP(hipSetDevice(ctx->device));
        A3_HIP(create_stream(&ctx->own_stream, 0, 0));
    }
    {   // (batch_mode_of reads every live context's stream under this lock, possibly from another thread's submit)
        std::lock_guard<std::mutex> lk(g_streams_mu);
        ctx->stream = ctx->own_stream;
    }
    return A3_OK;
}

// ---- bursts: contexts whose submitted batch has its threshold kernel enqueued and the rest held back (see a3_ctx::rest_held) ----
std::vector<a3_ctx*> g_held;   // guarded by g_defer_mu, in submission order
int enqueue_batch(a3_ctx* ctx, const uint8_t* pixels, int fmt, uint32_t W, uint32_t H, size_t row_stride, size_t frame_stride, uint32_t n,
                  size_t out_cap, int phase, bool defer_locked = false);
// Enqueue the held chain of `ctx`'s batch on its stream, behind `after` when given (the threshold kernel of the burst's last
// member).  g_defer_mu is held; the owner may be another thread's context, so the verdict is kept for its collect.
int flush_held_locked(a3_ctx* ctx, hipEvent_t after, bool by_last_member = false) {
    if (!ctx->rest_held) return A3_OK;
    ctx->rest_held = false;
    ctx->stepping = by_last_member ? A3_STEP_HELD_RELEASED_BY_LAST : A3_STEP_HELD_RELEASED_EARLY;
    for (size_t i = 0; i < g_held.size(); i++)
        if (g_held[i] == ctx) { g_held.erase(g_held.begin() + (long)i); break; }
    const Pending pd = ctx->pending;
    const bool wp = ctx->want_pose;
    ctx->want_pose = pd.want_pose;
    int rc = A3_OK;
    if (after && hipStreamWaitEvent(ctx->stream, after, 0) != hipSuccess) rc = fail(ctx, A3_ERR_HIP, "hipStreamWaitEvent (burst gate)");
    if (rc == A3_OK) rc = enqueue_batch(ctx, pd.pixels, pd.fmt, pd.W, pd.H, pd.row_stride, pd.frame_stride, pd.n, ctx->held_out_cap, 2, /*defer_locked=*/true);
    ctx->want_pose = wp;
    ctx->held_rc = rc;
    (void)hipStreamQuery(ctx->stream);   // hands what was just queued to the GPU now (the owner may be polling an event)
    return rc;
}

// ---- what a batch's chain needs before it can be enqueued without allocating ----
// A batch shaped like the previous one, which fitted one chunk, is planned on the device (no read-back of the dart counts before
// the contour stage).  -> the dart capacity such a batch is launched with; 0: this batch must be planned by the host.
uint64_t device_plan_capacity(const a3_ctx* ctx, uint32_t n, uint32_t W, uint32_t H) {
    if (!(ctx->plan_valid && !ctx->force_host_plan && ctx->plan_n == n && ctx->plan_W == W && ctx->plan_H == H && n <= kMaxChunkFrames)) return 0;
    const uint64_t cap_d = ctx->plan_darts + ctx->plan_darts / 4 + 65536;
    return cap_d > ctx->max_darts ? 0 : cap_d;
}
// One allocation for everything small: [frame_cursor | cand_count | (device plan: frame_darts) | HEAD | markers], HEAD =
// [scratch 256 B | counters | per_frame]: what the host reads back ahead of the markers.
struct ZeroLayout { size_t ctr_bytes, head_bytes, head_off, total; };
ZeroLayout zero_layout(size_t n_chunks, uint32_t chunk_frames, uint32_t n, uint32_t marker_cap) {
    ZeroLayout z;
    z.ctr_bytes = sizeof(DeviceCounters) * n_chunks;
    z.head_bytes = (256 + z.ctr_bytes + (size_t)n * 4 + 7) & ~(size_t)7;
    const size_t fd_off = ((size_t)chunk_frames * 4 + (size_t)n * 4 + 15) & ~(size_t)15;
    z.head_off = (fd_off + 255) & ~(size_t)255;
    z.total = z.head_off + z.head_bytes + (size_t)marker_cap * sizeof(a3_marker);
    return z;
}
uint32_t marker_cap_of(const a3_ctx* ctx, uint32_t n, size_t out_cap) {
    return (uint32_t)std::min<size_t>(std::max<size_t>(out_cap, 1), (size_t)n * ctx->max_cand);
}
uint32_t marker_guess_of(const a3_ctx* ctx, uint32_t marker_cap) {
    return (uint32_t)std::min<size_t>(marker_cap, (size_t)ctx->last_marker_total + ctx->last_marker_total / 4 + 64);
}
// a single-chunk batch is followed by device-planned ones sized darts * 1.25 + 64k: the pool is allocated for that at once
uint64_t pool_darts_of(const a3_ctx* ctx, uint64_t max_chunk_darts, size_t n_chunks) {
    uint64_t pool_darts = std::max<uint64_t>(max_chunk_darts, 1);
    if (n_chunks == 1) {
        pool_darts = std::min<uint64_t>(std::max<uint64_t>(ctx->max_darts, pool_darts), pool_darts + pool_darts / 2 + 131072);
        // in steps of an eighth of an octave: a stream of batches whose graphs differ by a percent or two (consecutive batches of one
        // camera) must not re-allocate the pool -- a dozen hipFree + hipMalloc in the middle of a batch, ~2 ms -- at every new maximum
        uint64_t step = 1ull << 17;
        while (step * 16 < pool_darts) step <<= 1;
        pool_darts = std::min<uint64_t>((pool_darts + step - 1) / step * step, std::max<uint64_t>(kHardMaxDarts, pool_darts));
    }
    return pool_darts;
}
// Every buffer the chain (contour stage ... read-back) of a DEVICE-PLANNED batch uses, allocated now: a chain that is held back for
// a burst is enqueued later -- by whichever thread submits the burst's last member, under the process-wide lock -- and must then
// find nothing left to allocate (hipMalloc / hipHostMalloc synchronise the device).  The same calls stand in enqueue_batch's chain
// half, where they are no-ops afterwards (the buffers only grow).
int ensure_chain_buffers(a3_ctx* ctx, hipStream_t st, uint32_t n, uint32_t W, uint32_t H, uint64_t cap_d, size_t out_cap) {
    const size_t npx = (size_t)W * H;
    const uint32_t marker_cap = marker_cap_of(ctx, n, out_cap);
    const ZeroLayout z = zero_layout(1, n, n, marker_cap);
    A3_HIP(ctx->tile_darts.ensure(tile_darts_bytes(W, H, n)));
    A3_HIP(ctx->zero_blk.ensure(z.total));
    if (ctx->frame_darts_dev.cap < (size_t)n * 8) {
        A3_HIP(ctx->frame_darts_dev.ensure((size_t)n * 8));
        A3_HIP(hipMemsetAsync(ctx->frame_darts_dev.p, 0, ctx->frame_darts_dev.cap, st));
    }
    A3_HIP(ctx->frame_base.ensure((size_t)(n + 1) * 4));
    if (int rc = ensure_dart_pool(ctx, pool_darts_of(ctx, cap_d, 1))) return rc;
    A3_HIP(ctx->pix_base.ensure((size_t)n * npx * 4));
    A3_HIP(ctx->contours.ensure((size_t)ctx->max_contours * sizeof(ContourRec)));
    A3_HIP(ctx->cyc_start_off.ensure((size_t)ctx->max_contours * 4));
    A3_HIP(ctx->points.ensure(ctx->max_points * 4));
    if (ctx->want_pose) A3_HIP(ctx->pose_buf.ensure((size_t)marker_cap * 2 * sizeof(a3_pose)));
    if (int rc = ensure_pinned(ctx, z.head_bytes + (size_t)marker_guess_of(ctx, marker_cap) * (sizeof(a3_marker) + 2 * sizeof(a3_pose)) + (1 << 16))) return rc;
    if (ctx->debug_taps && ctx->pinned_counts_cap < (size_t)n * 8) {
        if (ctx->pinned_counts) (void)hipHostFree(ctx->pinned_counts);
        ctx->pinned_counts = nullptr; ctx->pinned_counts_cap = 0;
        A3_HIP(hipHostMalloc(&ctx->pinned_counts, (size_t)n * 8, hipHostMallocDefault));
        ctx->pinned_counts_cap = (size_t)n * 8;
    }
    return A3_OK;
}

// the whole pipeline for one batch; `pixels` is a device pointer here
// One batch = enqueue_batch (every launch and the read-back copies, then an event) + finish_batch (wait for the event, check
// the device's verdict, hand out the markers).  a3_detect_batch runs them back to back; a3_detect_batch_submit / _collect
// let the caller enqueue the next batch (on another context) before collecting this one, so the GPU never waits for the host.
// phase 0: the whole batch; 1: the front half only (buffers + threshold kernel + an event behind it); 2: everything after the
// threshold kernel of a batch whose front half phase 1 enqueued (same arguments).  `defer_locked`: the caller holds g_defer_mu
// (a held chain released by another context's submit, by a gate, by collect): nothing in here may take that lock again, so such a
// batch neither releases other contexts' deferred decode stages nor defers its own -- and, planned on the device with every buffer
// allocated by its phase 1 (ensure_chain_buffers), it neither allocates nor waits for the device.
int enqueue_batch(a3_ctx* ctx, const uint8_t* pixels, int fmt, uint32_t W, uint32_t H, size_t row_stride, size_t frame_stride, uint32_t n,
                  size_t out_cap, int phase, bool defer_locked) {
    hipStream_t st = ctx->stream;
    const int rel_mode = defer_locked ? 0 : release_mode();
    const size_t npx = (size_t)W * H;
    const uint32_t minwh = W < H ? W : H;
    const uint32_t min_edge_length = (uint32_t)((float)minwh * ctx->cfg.min_side_length_factor);   // src/aruco.rs:55
    const float min_corner_separation = (float)minwh * ctx->cfg.min_corner_separation_factor;       // src/aruco.rs:56
    const uint32_t S = ctx->cfg.homography_sample_size;
    const uint32_t kMaxCand = ctx->max_cand;

    // the grey plane is materialised only for readers outside the fused path: Detection.grey (debug taps) and the generic
    // threshold kernels (windows above 15); the decode stage otherwise samples the caller's frames directly
    const bool big_window = threshold_writes_grey_plane(ctx->cfg.threshold_window, pixels, row_stride, frame_stride, (int)W);
    const bool need_grey = ctx->debug_taps || big_window;
    if (need_grey) A3_HIP(ctx->grey.ensure(npx * n));
    ctx->grey_valid = need_grey;
    if (big_window) A3_HIP(ctx->hsum.ensure(npx * n * 2));
    const size_t bits_per_frame = (size_t)words_per_row(W) * 8 * H;   // packed thresholded image
    A3_HIP(ctx->bin.ensure(bits_per_frame * n));
    A3_HIP(ctx->frame_darts.ensure((size_t)n * 8));
    A3_HIP(ctx->cands.ensure((size_t)n * kMaxCand * sizeof(CandRec)));
    A3_HIP(ctx->pre_xy.ensure((size_t)n * kMaxCand * 16));
    A3_HIP(ctx->fin_xy.ensure((size_t)n * kMaxCand * 16));
    A3_HIP(ctx->fin_count.ensure((size_t)n * 4));
    A3_HIP(ctx->work.ensure((size_t)n * kMaxCand * 4));
    if (kMaxCand > frame_cand_lds_slots()) A3_HIP(ctx->cand_big.ensure((size_t)n * kMaxCand * 4));   // keys / perimeters of k_frame_candidates' through-memory form
    A3_HIP(ctx->outs.ensure((size_t)n * kMaxCand * decode_out_bytes()));
    A3_HIP(ctx->proj.ensure((size_t)n * kMaxCand * proj_rec_bytes()));
    const uint32_t marker_cap = marker_cap_of(ctx, n, out_cap);
    const uint32_t patch_cap = (uint32_t)std::min<uint64_t>(kPatchCapMax, std::max<uint64_t>(kPatchCapMin, (uint64_t)n * kMaxCand));
    if (ctx->debug_taps) { A3_HIP(ctx->patches.ensure((size_t)patch_cap * S * S)); ctx->patch_cap = patch_cap; }
    int prof = ctx->front_prof;
    if (phase != 2) {
    ctx->W = W; ctx->H = H; ctx->frames = n;
    ctx->stats = a3_stats{};
    ctx->contours_valid = false; ctx->markers_valid = false; ctx->poses_valid = false;

    // ---- K1 ----
    // (level in force for THIS batch: the sampled threshold-only mode times one batch in profile_every)
    prof = ctx->profiling == 1 && (ctx->batch_seq++ % (uint32_t)ctx->profile_every) != 0 ? 0 : ctx->profiling;
    ctx->front_prof = prof;
    hipStream_t k1st = st;
    if (g_part_k1_cus > 0 || g_k1_stream_prio > 0) {   // CU partition / priority probe: the threshold kernel runs on the device's K1 stream, between two events
        A3_HIP(device_stream(ctx->device, kStreamK1, &k1st));
        A3_HIP(hipEventRecord(ctx->ev_k1_ready, st));
        A3_HIP(hipStreamWaitEvent(k1st, ctx->ev_k1_ready, 0));
    }
    if (prof) A3_HIP(hipEventRecord(ctx->ev[0], k1st));
    A3_HIP(launch_grey_threshold(k1st, pixels, fmt, row_stride, frame_stride, (int)W, (int)H, n, ctx->cfg.threshold_window,
                                 need_grey ? ctx->grey.as<uint8_t>() : nullptr, ctx->bin.as<uint64_t>(), big_window ? ctx->hsum.as<uint16_t>() : nullptr));
    if (prof) A3_HIP(hipEventRecord(ctx->ev[1], k1st));
    if (k1st != st) {
        A3_HIP(hipEventRecord(ctx->ev_k1_done, k1st));
        A3_HIP(hipStreamWaitEvent(st, ctx->ev_k1_done, 0));
    } else if (g_mark_threshold || phase == 1) A3_HIP(hipEventRecord(ctx->ev_k1_done, st));   // (held chains of a burst wait for the last member's)
    ctx->k1_marked = g_mark_threshold || phase == 1 || k1st != st;
    if (phase == 1) {   // the chain is held back: what it will need is allocated now (submit_common holds device-planned batches only)
        const uint64_t cap1 = device_plan_capacity(ctx, n, W, H);
        if (cap1 == 0) return fail(ctx, A3_ERR_INTERNAL, "a chain was held back for a batch that needs a host-side plan");
        return ensure_chain_buffers(ctx, st, n, W, H, cap1, out_cap);
    }
    }
    // batches of OTHER contexts (same device) that wait with their decode stage are released from inside this batch's launch
    // sequence (see release_mode()): `release_waiting()` records the event they wait for and enqueues them
    bool released = false;
    auto release_waiting = [&]() -> int {
        if (released) return A3_OK;
        std::lock_guard<std::mutex> lk(g_defer_mu);
        bool any = false;
        for (a3_ctx* o : g_deferred) any |= (o != ctx && o->device == ctx->device);
        if (!any) return A3_OK;
        released = true;
        A3_HIP(hipEventRecord(ctx->ev_k1, st));
        const std::vector<a3_ctx*> list = g_deferred;   // (flush edits g_deferred)
        for (a3_ctx* o : list)
            if (o != ctx && o->device == ctx->device)
                if (int rc = flush_deferred_locked(o, ctx->ev_k1)) { ctx->err = "deferred decode of another context: " + o->err; return rc; }
        return A3_OK;
    };
    if (rel_mode == 1) { if (int rc = release_waiting()) return rc; }

    // ---- contour graph size per frame -> chunk plan ----
    // A batch shaped like the previous one is planned on the device: no read-back, no idle GPU while the host thinks.
    const uint64_t cap_d = device_plan_capacity(ctx, n, W, H);
    const bool device_plan = cap_d != 0;
    // The zero block (zero_layout): one memset zeroes it up to the end of HEAD; HEAD and the marker list that follows it come back
    // to the host in one copy.
    size_t ctr_bytes = 0, head_bytes = 0, head_off = 0;
    void* zero_p = nullptr; size_t zero_bytes = 0;
    auto layout_zero_block = [&](size_t n_chunks, uint32_t chunk_frames, bool launch) -> hipError_t {
        const ZeroLayout zl = zero_layout(n_chunks, chunk_frames, n, marker_cap);
        ctr_bytes = zl.ctr_bytes; head_bytes = zl.head_bytes; head_off = zl.head_off;
        const hipError_t e = ctx->zero_blk.ensure(zl.total);
        if (e != hipSuccess) return e;
        uint8_t* z = ctx->zero_blk.as<uint8_t>();
        ctx->frame_cursor = reinterpret_cast<uint32_t*>(z);
        ctx->cand_count = ctx->frame_cursor + chunk_frames;
        ctx->scratch_u32 = reinterpret_cast<unsigned int*>(z + head_off);
        ctx->counters = reinterpret_cast<DeviceCounters*>(z + head_off + 256);
        ctx->per_frame = reinterpret_cast<uint32_t*>(z + head_off + 256 + ctr_bytes);
        ctx->markers_ptr = reinterpret_cast<a3_marker*>(z + head_off + head_bytes);
        zero_p = z; zero_bytes = (head_off + head_bytes + 15) & ~(size_t)15;   // (may run a few bytes into the marker area: not yet written)
        return launch ? launch_zero(st, zero_p, zero_bytes) : hipSuccess;
    };
    std::vector<Chunk> chunks;
    std::vector<uint64_t> fd;
    A3_HIP(ctx->tile_darts.ensure(tile_darts_bytes(W, H, n)));
    if (device_plan) {
        chunks.push_back(Chunk{0, n, cap_d, (uint32_t)std::min<uint64_t>(cap_d, 0xFFFFFFFFu)});
        // The plan workgroup (last of k_tile_scan's launch) zeroes the block before it writes the plan into it: nothing earlier
        // touches it, and a launch of its own costs 4 us for a few KB.  The per-frame dart totals k_dart_count adds up live in a
        // buffer of their own that the plan workgroup hands back zeroed.
        A3_HIP(layout_zero_block(1, n, false));
        if (ctx->frame_darts_dev.cap < (size_t)n * 8) {
            A3_HIP(ctx->frame_darts_dev.ensure((size_t)n * 8));
            A3_HIP(hipMemsetAsync(ctx->frame_darts_dev.p, 0, ctx->frame_darts_dev.cap, st));
        }
        ctx->frame_darts_ptr = ctx->frame_darts_dev.as<unsigned long long>();
    } else {
        ctx->frame_darts_ptr = ctx->frame_darts.as<unsigned long long>();
        A3_HIP(hipMemsetAsync(ctx->frame_darts.p, 0, (size_t)n * 8, st));
    }
    // device plan: frame bases and the dart total come out of the same launch sequence (scratch words 8..11, read back with the results)
    if (device_plan) A3_HIP(ctx->frame_base.ensure((size_t)(n + 1) * 4));
    A3_HIP(launch_dart_count(st, ctx->bin.as<uint64_t>(), (int)W, (int)H, 0, n, ctx->frame_darts_ptr, ctx->tile_darts.as<uint32_t>(), cap_d,
                             device_plan ? ctx->frame_base.as<uint32_t>() : nullptr, device_plan ? ctx->scratch_u32 + 8 : nullptr,
                             device_plan ? zero_p : nullptr, device_plan ? zero_bytes : 0));
    if (!device_plan) {
        if (int rc = ensure_pinned(ctx, std::max<size_t>((size_t)n * 8, 1 << 16))) return rc;
        A3_HIP(hipMemcpyAsync(ctx->pinned, ctx->frame_darts.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        A3_HIP(wait_stream(st));
        fd.assign((uint64_t*)ctx->pinned, (uint64_t*)ctx->pinned + n);
        uint64_t biggest = 0;
        for (uint64_t v : fd) { biggest = std::max(biggest, v); ctx->stats.darts += v; }
        if (biggest > kHardMaxDarts) return fail(ctx, A3_ERR_LIMIT, "a frame needs more contour-graph nodes than 32-bit indices allow");
        if (biggest > ctx->max_darts) ctx->max_darts = biggest;  // one frame must fit; grow the pool
        Chunk c{0, 0, 0, 0};
        for (uint32_t f = 0; f < n; f++) {
            if (c.count && (c.darts + fd[f] > ctx->max_darts || c.count >= kMaxChunkFrames)) { chunks.push_back(c); c = Chunk{f, 0, 0, 0}; }
            c.count++; c.darts += fd[f]; c.max_frame_darts = (uint32_t)std::max<uint64_t>(c.max_frame_darts, fd[f]);
        }
        if (c.count) chunks.push_back(c);
    }
    uint32_t max_chunk_frames = 0; uint64_t max_chunk_darts = 0;
    for (auto& c : chunks) { max_chunk_frames = std::max(max_chunk_frames, c.count); max_chunk_darts = std::max(max_chunk_darts, c.darts); }
    ctx->stats.chunks = (uint32_t)chunks.size();
    if (int rc = ensure_dart_pool(ctx, pool_darts_of(ctx, max_chunk_darts, chunks.size()))) return rc;
    A3_HIP(ctx->pix_base.ensure((size_t)max_chunk_frames * npx * 4));
    A3_HIP(ctx->frame_base.ensure((size_t)(max_chunk_frames + 1) * 4 * chunks.size()));
    if (!device_plan) A3_HIP(layout_zero_block(chunks.size(), max_chunk_frames, true));
    A3_HIP(ctx->contours.ensure((size_t)ctx->max_contours * sizeof(ContourRec)));
    A3_HIP(ctx->cyc_start_off.ensure((size_t)ctx->max_contours * 4));
    A3_HIP(ctx->points.ensure(ctx->max_points * 4));

    unsigned int* d_err = ctx->scratch_u32 + 4;             // ([0] work count, [1] marker total: enqueue_back)
    unsigned int* d_entry_count = ctx->scratch_u32 + 32;    // [32..47]
    unsigned int* d_leader_count = ctx->scratch_u32 + 16;   // [16..31]
    unsigned int* d_dead_count = ctx->scratch_u32 + 48;     // [48..63]: borders k_local_contract finished with (traced, never listed), 16 shards, all chunks

    const uint32_t* n_live = nullptr;
    if (device_plan) {
        n_live = ctx->scratch_u32 + 8;   // written by the plan workgroup of launch_dart_count
    } else {
        // frame bases of every chunk, uploaded once
        std::vector<uint32_t> bases;
        for (auto& c : chunks) {
            uint32_t acc = 0;
            for (uint32_t i = 0; i < max_chunk_frames + 1; i++) {
                bases.push_back(acc);
                if (i < c.count) acc += (uint32_t)fd[c.first + i];
            }
        }
        // the pinned buffer still holds fd; stage the bases behind it
        if (int rc = ensure_pinned(ctx, (size_t)n * 8 + bases.size() * 4 + (1 << 16))) return rc;
        uint32_t* h_bases = reinterpret_cast<uint32_t*>((uint8_t*)ctx->pinned + (size_t)n * 8);
        memcpy(h_bases, bases.data(), bases.size() * 4);
        A3_HIP(hipMemcpyAsync(ctx->frame_base.p, h_bases, bases.size() * 4, hipMemcpyHostToDevice, st));
    }

    // ---- contour stage, chunk by chunk ----
    const uint64_t* d_bin = ctx->bin.as<uint64_t>();
    const double image_diag = std::sqrt((double)W * W + (double)H * H);
    int rounds_max = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        const Chunk& c = chunks[ci];
        DeviceCounters* ctr = ctx->counters + ci;
        const uint32_t* fb = ctx->frame_base.as<uint32_t>() + ci * (max_chunk_frames + 1);
        const uint32_t nd = (uint32_t)c.darts;
        if (nd == 0) continue;
        if (ci > 0) {   // the batch-wide memset covered chunk 0
            A3_HIP(hipMemsetAsync(d_leader_count, 0, 4 * 32, st));   // leader + entry counters, adjacent
            A3_HIP(hipMemsetAsync(ctx->frame_cursor, 0, (size_t)c.count * 4, st));   // per-frame entry counts
        }
        const uint32_t* tile_off = ctx->tile_darts.as<uint32_t>() + tile_off_offset(W, H, n);
        A3_HIP(launch_dart_build(st, d_bin, (int)W, (int)H, c.first, c.count, fb, tile_off, ctx->pix_base.as<uint32_t>(),
                                 ctx->tile_darts.as<uint32_t>(), ctx->d_xy.as<uint64_t>(), ctx->d_succ.as<uint32_t>(), nd, n_live, 0,
                                 reinterpret_cast<const unsigned long long*>(ctx->tile_darts.as<uint8_t>() + tile_mask_offset_bytes(W, H, n))));
        int rounds = 1;
        while ((1ull << rounds) < (uint64_t)c.max_frame_darts && rounds < 31) rounds++;
        rounds += 1;  // the round that observes "nothing moved"
        rounds = std::min(rounds, ctx->jump_rounds_hint);
        rounds_max = std::max(rounds_max, rounds);
        uint32_t* const frame_entries = ctx->entry_global_ttl > 0 ? nullptr : ctx->frame_cursor;   // per-frame entry counts
        const double eps_factor = ctx->cfg.contour_simplification_epsilon;
        const uint32_t max_contours = ctx->max_contours;
        const uint64_t max_points = ctx->max_points;
        const int resolve_iters = ctx->resolve_full_ttl > 0 ? ctx->resolve_iters_hint : 0;
        const int inline_resolve_W = ctx->resolve_full_ttl > 0 ? 0 : (int)W;
        const int keep_all = ctx->debug_taps ? 1 : 0;
        // Short borders are finished with inside k_local_contract (kDead) on DENSE graphs only -- noise-like frames, where nine borders
        // in ten die of their length: there it saves a sixth of the contour stage; on clean frames (a dart per hundred pixels, a few
        // dozen borders per frame) it would only cost its bookkeeping.  Results are the same either way.
        const bool dense_graph = (uint64_t)nd * 10u >= (uint64_t)c.count * npx;
        unsigned int* const dead_ctr = (keep_all || !dense_graph) ? nullptr : d_dead_count;
        const Chunk cc = c;
        // first half: the doubling rounds inside LDS tiles
        A3_HIP(launch_rank_cycles(st, nd, (int)W, ctx->d_xy.as<uint64_t>(), ctx->d_succ.as<uint32_t>(), ctx->stA.as<JumpState>(),
                                  ctx->entry_list.as<uint32_t>(), d_entry_count, ctx->es_a.p, ctx->es_b.p,
                                  ctx->stB.p, ctx->leader_list.as<uint32_t>(), d_leader_count, rounds, ctr, n_live, 0, fb,
                                  frame_entries, cc.count, 1, min_edge_length, dead_ctr, inline_resolve_W > 0 ? 1 : 0));
        if (rel_mode == 2) { if (int rc = release_waiting()) return rc; }   // waiting decode stages go out behind this k_local_contract
        ctx->dbg_nd = nd; ctx->dbg_frames = c.count; ctx->dbg_chunks = (uint32_t)chunks.size();
        // second half: entry resolution, final states (+ border selection), point scatter, quads -- on `s2`
        auto chunk_back = [=](hipStream_t s2) -> int {
            A3_HIP(launch_rank_cycles(s2, nd, (int)W, ctx->d_xy.as<uint64_t>(), ctx->d_succ.as<uint32_t>(), ctx->stA.as<JumpState>(),
                                      ctx->entry_list.as<uint32_t>(), d_entry_count, ctx->es_a.p, ctx->es_b.p,
                                      ctx->stB.p /* the 8-byte final states */, ctx->leader_list.as<uint32_t>(), d_leader_count, rounds, ctr,
                                      n_live, 0, fb, frame_entries, cc.count, 2, min_edge_length, dead_ctr, inline_resolve_W > 0 ? 1 : 0));
            const JumpState* loc = ctx->stA.as<JumpState>();
            const void* fin = ctx->stB.p;
            A3_HIP(launch_resolve(s2, loc, fin, nd, (int)W, ctx->d_xy.as<uint64_t>(), ctx->leader_list.as<uint32_t>(), d_leader_count,
                                  ctx->t_cur.as<uint64_t>(), ctx->t_next.as<uint64_t>(), ctr, resolve_iters, n_live));
            A3_HIP(launch_select_scatter(s2, loc, fin, nd, ctx->leader_list.as<uint32_t>(), d_leader_count, ctx->d_succ.as<uint32_t>(), ctx->t_cur.as<uint64_t>(), fb,
                                         cc.count, cc.first, min_edge_length, eps_factor, image_diag,
                                         ctx->contours.as<ContourRec>(), ctx->cyc_start_off.as<uint32_t>(), max_contours, max_points, ctr,
                                         ctx->d_xy.as<uint64_t>(), ctx->points.as<uint32_t>(), n_live, inline_resolve_W, ctx->leader_keep.as<uint32_t>(),
                                         keep_all));
            A3_HIP(launch_contour_quads(s2, ctx->contours.as<ContourRec>(), ctr, max_contours, ctx->points.as<uint32_t>(),
                                        eps_factor, min_edge_length, cc.first, kMaxCand,
                                        ctx->cands.as<CandRec>() + (size_t)cc.first * kMaxCand, ctx->cand_count + cc.first, d_err,
                                        W <= 16384u && H <= 16384u ? 1 : 0, nd));
            return A3_OK;
        };
        if (int rc = chunk_back(st)) return rc;
    }
    if (rel_mode != 0) { if (int rc = release_waiting()) return rc; }   // (a batch without a contour graph releases here)
    if (prof >= 2) A3_HIP(hipEventRecord(ctx->ev[2], st));

    // ---- candidates -> markers -> read-back: enqueued now, or deferred behind the next submitted batch's threshold kernel ----
    const PixelSrc src = need_grey ? PixelSrc{ctx->grey.as<uint8_t>(), W, (unsigned long long)npx, kFmtGreyPlane}
                                   : PixelSrc{pixels, row_stride, frame_stride, fmt};
    ctx->dbg_src = src;
    const size_t pose_bytes = ctx->want_pose ? 2 * sizeof(a3_pose) : 0;
    const uint32_t guess = marker_guess_of(ctx, marker_cap);
    const size_t head_pad = head_bytes;   // the markers follow the head directly, on the device and in the staging buffer
    // every allocation of the second half happens here, at submit time: pose buffer, pinned staging for the head and `guess`
    // markers (+ poses; a longer list is fetched by finish_batch after growing it), pinned staging for the tap counts
    if (ctx->want_pose) A3_HIP(ctx->pose_buf.ensure((size_t)marker_cap * 2 * sizeof(a3_pose)));
    if (int rc = ensure_pinned(ctx, head_pad + (size_t)guess * (sizeof(a3_marker) + 2 * sizeof(a3_pose)) + (1 << 16))) return rc;
    ctx->counts_valid = false;
    if (ctx->debug_taps && ctx->pinned_counts_cap < (size_t)n * 8) {
        if (ctx->pinned_counts) (void)hipHostFree(ctx->pinned_counts);
        ctx->pinned_counts = nullptr; ctx->pinned_counts_cap = 0;
        A3_HIP(hipHostMalloc(&ctx->pinned_counts, (size_t)n * 8, hipHostMallocDefault));
        ctx->pinned_counts_cap = (size_t)n * 8;
    }
    BackArgs& bk = ctx->back;
    bk.n = n; bk.W = W; bk.H = H; bk.S = S; bk.max_cand = kMaxCand; bk.patch_cap = patch_cap; bk.marker_cap = marker_cap; bk.guess = guess;
    bk.min_corner_separation = min_corner_separation; bk.src = src; bk.head_bytes = head_bytes; bk.pose_bytes = pose_bytes;
    bk.taps = ctx->debug_taps; bk.want_pose = ctx->want_pose; bk.pose_has_intr = ctx->pose_has_intr; bk.pose_size_mm = ctx->pose_size_mm;
    bk.pose_intr = ctx->pose_intr; bk.profiling = prof;
    // Deferral: only for submitted batches (somebody will submit again or collect), and not while every stage is being timed
    // (the stage times are those of stages that run alone).  The decode stage then waits on the context's decode stream until
    // (a) another context submits a batch -- it is released behind that batch's threshold kernel and shares the GPU with its
    // contour stage -- or (b) this batch is collected first.
    if (phase == 0) ctx->stepping = A3_STEP_WHOLE;
    if (ctx->allow_defer && !defer_locked && ctx->profiling < 2 && ctx->batch_mode != 0) {
        A3_HIP(hipEventRecord(ctx->ev_contours, st));
        std::lock_guard<std::mutex> lk(g_defer_mu);
        ctx->back_deferred = true;
        ctx->back_rc = 0;
        ctx->stepping = A3_STEP_DECODE_DEFERRED;
        g_deferred.push_back(ctx);
    } else if (int rc = enqueue_back(ctx, st, bk)) return rc;
    Pending& pd = ctx->pending;
    pd.active = true; pd.n_chunks = chunks.size(); pd.chunk0_darts = chunks.empty() ? 0 : chunks[0].darts; pd.ctr_bytes = ctr_bytes;
    pd.head_pad = head_pad; pd.marker_cap = marker_cap; pd.guess = guess; pd.pose_bytes = pose_bytes; pd.device_plan = device_plan;
    pd.rounds_max = rounds_max; pd.n = n; pd.W = W; pd.H = H; pd.profiling = prof; pd.taps = ctx->debug_taps;
    return A3_OK;
}

int finish_batch(a3_ctx* ctx, a3_marker* out, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    Pending& pd = ctx->pending;
    if (!pd.active) return fail(ctx, A3_ERR_INVALID, "no batch was submitted");
    {   // a chain still held back (no later member of its burst was submitted): it goes out now
        std::lock_guard<std::mutex> lk(g_defer_mu);
        (void)flush_held_locked(ctx, nullptr);
        if (const int rc = ctx->held_rc) { ctx->held_rc = 0; pd.active = false; return rc; }
    }
    pd.active = false;
    hipStream_t st = ctx->stream;
    const size_t ctr_bytes = pd.ctr_bytes, head_pad = pd.head_pad, pose_bytes = pd.pose_bytes, n_chunks = pd.n_chunks;
    const uint32_t marker_cap = pd.marker_cap, guess = pd.guess, n = pd.n, W = pd.W, H = pd.H;
    const bool device_plan = pd.device_plan;
    const int rounds_max = pd.rounds_max;
    uint8_t* hp = (uint8_t*)ctx->pinned;
    a3_marker* h_markers = reinterpret_cast<a3_marker*>(hp + head_pad);
    a3_pose* h_poses = reinterpret_cast<a3_pose*>(hp + head_pad + (size_t)guess * sizeof(a3_marker));
    (void)marker_cap;
    {   // nobody submitted behind this batch: its decode stage goes out now
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (int rc = flush_deferred_locked(ctx, nullptr)) return rc;
        if (const int rc = ctx->back_rc) { ctx->back_rc = 0; return rc; }   // (released by another context's submit, and that failed)
    }
    A3_HIP(wait_event(ctx->ev[4], st));
    const unsigned int* hs = reinterpret_cast<const unsigned int*>(hp);
    const DeviceCounters* hc = reinterpret_cast<const DeviceCounters*>(hp + 256);
    if (device_plan) {
        if (hs[9]) { ctx->force_host_plan = true; return 1; }   // the graph outgrew the hint: plan on the host this once
        ctx->stats.darts = hs[8];
        ctx->dbg_nd = hs[8];
        ctx->plan_darts = hs[8];
    } else {
        ctx->plan_valid = n_chunks == 1;
        ctx->plan_n = n; ctx->plan_W = W; ctx->plan_H = H;
        ctx->plan_darts = n_chunks == 1 ? pd.chunk0_darts : 0;
    }
    unsigned int flags = hs[4];
    for (int sh = 0; sh < 16; sh++) ctx->stats.contours_traced += hs[48 + sh];   // borders finished inside k_local_contract (kDead)
#ifdef A3_TUNING
    if (tuning_knob("A3_PRINT_DEAD", 0)) { uint64_t dead = 0; for (int sh = 0; sh < 16; sh++) dead += hs[48 + sh]; fprintf(stderr, "[a3] borders finished in k_local_contract: %llu\n", (unsigned long long)dead); }
#endif
    uint64_t need_points = 0; uint32_t need_contours = 0;
    bool jump_short = false, resolve_needed = false, entry_overflow = false;
    for (size_t ci = 0; ci < n_chunks; ci++) {
        flags |= hc[ci].err_flags;
        resolve_needed |= hc[ci].resolve_needed != 0;
        entry_overflow |= hc[ci].entry_overflow != 0;
        need_points = std::max<uint64_t>(need_points, hc[ci].points);
        need_contours = std::max(need_contours, hc[ci].contours);
        ctx->stats.contours_traced += hc[ci].traced;
        ctx->stats.contours_materialised += hc[ci].contours;
        for (int r = 0; r < 32; r++) if (hc[ci].jump_changed[r]) ctx->stats.jump_rounds = std::max<uint32_t>(ctx->stats.jump_rounds, r + 1);
        uint32_t it = hc[ci].resolve_needed ? 1 : 0;  // 0: k_resolve_fast confirmed the natural starts; pass k+1 ran iff pass k moved something
        for (int r = 0; r < kResolveItersMax - 1; r++) if (hc[ci].resolve_changed[r]) it = r + 2;
        ctx->stats.resolve_iterations = std::max(ctx->stats.resolve_iterations, it);
        if (rounds_max > 0 && rounds_max < 32 && hc[ci].jump_changed[rounds_max - 1] != 0) jump_short = true;
    }
    if (entry_overflow) { ctx->entry_global_ttl = 64; ctx->jump_rounds_hint = std::max(ctx->jump_rounds_hint, 12); return 1; }     // a frame's entry list outgrew LDS: re-run with the global rounds
    if (ctx->entry_global_ttl > 0) ctx->entry_global_ttl--;
    if (jump_short && ctx->jump_rounds_hint < 32) { ctx->jump_rounds_hint = 32; return 1; }             // re-run with all rounds
    if (resolve_needed) {
        const bool ran = ctx->resolve_full_ttl > 0;
        ctx->resolve_full_ttl = 64;        // keep the full passes in the launch sequence for the next batches
        if (!ran) return 1;                // they were not launched this time: re-run
    } else if (ctx->resolve_full_ttl > 0) ctx->resolve_full_ttl--;
    if ((flags & kErrResolve) && ctx->resolve_iters_hint < kResolveItersMax) { ctx->resolve_iters_hint = kResolveItersMax; return 1; }
    ctx->jump_rounds_hint = std::max(4, std::min(32, (int)ctx->stats.jump_rounds + 2));   // follow the workload, both ways (one round of slack: a short launch costs a re-run)
    if (flags & (kErrPointPool | kErrContourTable)) {
        // grow and let the caller loop re-run the batch
        if (need_points > ctx->max_points) ctx->max_points = std::min<uint64_t>(kHardMaxPoints, std::max(need_points, ctx->max_points * 2));
        if (need_contours > ctx->max_contours) ctx->max_contours = std::max(need_contours, ctx->max_contours * 2);
        return 1;  // retry
    }
    if (flags & kErrBrokenEvent) return fail(ctx, A3_ERR_INTERNAL, "contour graph: a start event lies on an open chain");
    if (flags & kErrResolve) return fail(ctx, A3_ERR_INTERNAL, "contour start resolution did not converge");
    if (flags & kErrCandTable) {   // a frame has more quad candidates than its table: twice the table and again
        // straight to the table that holds the fullest frame (hs[5], from the marker gather): 2048, 4096, 6144 (the last that
        // k_frame_candidates works in LDS), 12 288, 24 576, 49 152, 65 536
        const uint32_t lds_slots = frame_cand_lds_slots(), need = std::max(hs[5], ctx->max_cand + 1u);
        if (need > kMaxCandLimit) return fail(ctx, A3_ERR_LIMIT, "a frame holds more than 65536 quad candidates (A3_MAX_CANDIDATES_PER_FRAME)");
        uint32_t next = ctx->max_cand;
        while (next < need) next = next < lds_slots ? std::min(lds_slots, next * 2) : std::min(kMaxCandLimit, next * 2);
        if ((uint64_t)n * next > 0xFFFFFFFFull) return fail(ctx, A3_ERR_LIMIT, "frames x candidate slots per frame exceeds 2^32: fewer frames per call");
        {   // Every frame of the batch gets the table of the fullest one: what that costs is known before anything is allocated, and a
            // batch whose tables cannot fit is a limit of this implementation (fewer frames per call cure it), not a HIP failure.
            const size_t per_slot = sizeof(CandRec) + 16 + 16 + 4 + (next > lds_slots ? 4 : 0) + decode_out_bytes() + proj_rec_bytes();
            const size_t had = ctx->cands.cap + ctx->pre_xy.cap + ctx->fin_xy.cap + ctx->work.cap + ctx->cand_big.cap + ctx->outs.cap + ctx->proj.cap;
            const size_t want = (size_t)n * next * per_slot;
            size_t free_b = 0, total_b = 0;
            if (want > had && hipMemGetInfo(&free_b, &total_b) == hipSuccess && want - had > free_b)
                return fail(ctx, A3_ERR_LIMIT, "candidate tables of this batch do not fit the device (every frame gets the fullest frame's table): fewer frames per call");
        }
        ctx->max_cand = next;
        return 1;
    }
    if (flags & kErrMarkerCap) return fail(ctx, A3_ERR_CAPACITY, "out_cap is smaller than the number of markers found");
    const uint32_t total = hs[1];
    if (total > out_cap) return fail(ctx, A3_ERR_CAPACITY, "out_cap is smaller than the number of markers found");
    const uint32_t* hpf = reinterpret_cast<const uint32_t*>(hp + 256 + ctr_bytes);
    if (per_frame_count) memcpy(per_frame_count, hpf, (size_t)n * 4);
    uint32_t max_per_frame = 0;   // (read now: the staging buffer may be re-allocated below)
    for (uint32_t f = 0; f < n; f++) max_per_frame = std::max(max_per_frame, hpf[f]);
    const uint32_t n_work = hs[0], n_pre = hs[2];
    const uint32_t tap_contours = n_chunks ? hc[0].contours : 0u; const uint64_t tap_points = n_chunks ? hc[0].points : 0ull;
    if (total > guess) {   // the guess was short: the staging area grows (the head has been consumed) and the whole list is fetched
        if (int rc = ensure_pinned(ctx, (size_t)total * (sizeof(a3_marker) + 2 * sizeof(a3_pose)) + (1 << 16))) return rc;
        h_markers = reinterpret_cast<a3_marker*>(ctx->pinned);
        h_poses = reinterpret_cast<a3_pose*>((uint8_t*)ctx->pinned + (size_t)total * sizeof(a3_marker));
        A3_HIP(hipMemcpyAsync(h_markers, ctx->markers_ptr, (size_t)total * sizeof(a3_marker), hipMemcpyDeviceToHost, st));
        if (pose_bytes) A3_HIP(hipMemcpyAsync(h_poses, ctx->pose_buf.p, (size_t)total * pose_bytes, hipMemcpyDeviceToHost, st));
        A3_HIP(hipStreamSynchronize(st));
    }
    if (total) {
        memcpy(out, h_markers, (size_t)total * sizeof(a3_marker));
        if (pose_bytes && ctx->pose_out) memcpy(ctx->pose_out, h_poses, (size_t)total * pose_bytes);
    }
    if (pd.taps) {
        const uint32_t* hc32 = reinterpret_cast<const uint32_t*>(ctx->pinned_counts);
        ctx->h_cand_pre.assign(hc32, hc32 + n);
        ctx->h_cand_fin.assign(hc32 + n, hc32 + 2 * (size_t)n);
        for (auto& v : ctx->h_cand_pre) v = std::min(v, ctx->max_cand);
        ctx->counts_valid = true;
    }
    ctx->last_marker_total = total;
    *out_n = total;
    ctx->stats.markers = total;
    ctx->stats.candidates = n_work;      // work items = quads that survived discard_too_near
    ctx->stats.candidates_pre = n_pre;   // quads after contours_to_candidates (k_compact_markers sums the per-frame counts)
    ctx->markers_valid = true; ctx->last_n = n; ctx->last_max_per_frame = max_per_frame;
    ctx->poses_valid = pose_bytes != 0;
    ctx->contours_valid = ctx->debug_taps && n_chunks == 1;
    if (ctx->contours_valid) { ctx->tap_contours = tap_contours; ctx->tap_points = tap_points; }
    if (pd.profiling >= 1) {   // the level in force when the batch was enqueued
        float ms;
        A3_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); ctx->prof_ms[A3_STAGE_THRESHOLD] += ms; ctx->prof_n[A3_STAGE_THRESHOLD]++;
        if (pd.profiling >= 2) {
            A3_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2])); ctx->prof_ms[A3_STAGE_CONTOUR] += ms; ctx->prof_n[A3_STAGE_CONTOUR]++;
            A3_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->prof_ms[A3_STAGE_DECODE] += ms; ctx->prof_n[A3_STAGE_DECODE]++;
        }
    }
    return A3_OK;
}

int run_batch(a3_ctx* ctx, const uint8_t* pixels, int fmt, uint32_t W, uint32_t H, size_t row_stride, size_t frame_stride, uint32_t n,
              a3_marker* out, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    if (int rc = enqueue_batch(ctx, pixels, fmt, W, H, row_stride, frame_stride, n, out_cap, 0)) return rc;
    return finish_batch(ctx, out, out_cap, per_frame_count, out_n);
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int a3_abi_version(void) { return A3_ABI_VERSION; }

void a3_default_config(a3_config* cfg) {  // src/aruco.rs:32-43
    if (!cfg) return;
    cfg->threshold_window = 7;
    cfg->contour_simplification_epsilon = 0.05;
    cfg->min_side_length_factor = 0.2f;
    cfg->min_corner_separation_factor = 0.1f;
    cfg->homography_sample_size = 49;
    cfg->filter_high_bit_errors = 1;
}

const char* a3_last_error(const a3_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int a3_calculate_tau(int device, const uint64_t* codes, size_t n_codes, uint8_t* tau) {
    a3_ctx* ctx = nullptr;
    if (!codes || !tau) return fail(ctx, A3_ERR_INVALID, "a3_calculate_tau: null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(ctx, A3_ERR_NO_DEVICE, "no HIP device");
    A3_HIP(hipSetDevice(device));
    uint64_t* d = nullptr; unsigned int* dt = nullptr;
    unsigned int init = 255, res = 255;
    hipError_t e = hipMalloc(&d, std::max<size_t>(n_codes, 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&dt, 4);
    if (e == hipSuccess && n_codes) e = hipMemcpy(d, codes, n_codes * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dt, &init, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_calc_tau(nullptr, d, (uint32_t)n_codes, dt);
    if (e == hipSuccess) e = hipMemcpy(&res, dt, 4, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);     // on every path
    if (dt) (void)hipFree(dt);
    if (e != hipSuccess) return fail(ctx, A3_ERR_HIP, "a3_calculate_tau", e);
    *tau = (uint8_t)res;
    return A3_OK;
}

int a3_create(int device, const a3_config* cfg, const uint64_t* codes, size_t n_codes, uint8_t num_bits, uint8_t tau, a3_ctx** out) {
    a3_ctx* ctx = nullptr;
    if (!cfg || !out || (!codes && n_codes)) return fail(ctx, A3_ERR_INVALID, "a3_create: null argument");
    if (cfg->threshold_window == 0) return fail(ctx, A3_ERR_INVALID, "threshold_window must be > 0 (imageproc asserts block_radius > 0)");
    if (!(cfg->contour_simplification_epsilon > 0.0)) return fail(ctx, A3_ERR_INVALID, "contour_simplification_epsilon must be > 0");
    if (cfg->homography_sample_size == 0 || cfg->homography_sample_size > 200)
        return fail(ctx, A3_ERR_INVALID, "homography_sample_size must be in 1..200");
    if (num_bits == 0 || num_bits > 64) return fail(ctx, A3_ERR_INVALID, "num_bits must be in 1..64");
    if (n_codes > 0xFFFFFFFFull) return fail(ctx, A3_ERR_INVALID, "dictionary too large");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(ctx, A3_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= count) return fail(ctx, A3_ERR_INVALID, "device index out of range");
    A3_HIP(hipSetDevice(device));
    a3_ctx* c = new a3_ctx();
    c->device = device;
    c->cfg = *cfg;
    c->num_bits = num_bits;
    c->n_codes = (uint32_t)n_codes;
    c->mark_size = mark_size_of(num_bits);
    ctx = c;
    hipError_t e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_contours, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k1, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k1_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k1_done, hipEventDisableTiming);
    if (e != hipSuccess) { a3_destroy(c); ctx = nullptr; return fail(ctx, A3_ERR_HIP, "hipEventCreate", e); }
    for (auto& ev : c->ev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) { a3_destroy(c); ctx = nullptr; return fail(ctx, A3_ERR_HIP, "hipEventCreate", e); }
    }
    e = c->dict.ensure(std::max<size_t>(n_codes, 1) * 8);
    if (e == hipSuccess && n_codes) e = hipMemcpy(c->dict.p, codes, n_codes * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { a3_destroy(c); ctx = nullptr; return fail(ctx, A3_ERR_HIP, "dictionary upload", e); }
    e = c->wtab.ensure(weight_table_bytes());
    // (on the legacy default stream: a stream of the context's own is only created if the caller never passes one)
    if (e == hipSuccess) e = launch_weight_table(nullptr, cfg->homography_sample_size, c->mark_size, cfg->homography_sample_size, c->wtab.as<float>());
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { a3_destroy(c); ctx = nullptr; return fail(ctx, A3_ERR_HIP, "resize weight table", e); }
    if (tau == 0) {  // src/dictionaries.rs:124
        uint8_t t = 255;
        int rc = a3_calculate_tau(device, codes, n_codes, &t);
        if (rc) { a3_destroy(c); return rc; }
        tau = t;
    }
    c->tau = tau;
    { std::lock_guard<std::mutex> lk(g_streams_mu); g_contexts.push_back(c); }
    *out = c;
    return A3_OK;
}

void a3_destroy(a3_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        for (size_t i = 0; i < g_contexts.size(); i++)
            if (g_contexts[i] == ctx) { g_contexts.erase(g_contexts.begin() + (long)i); break; }
    }
    {   // a submitted batch that was never collected: its deferred half is dropped
        std::lock_guard<std::mutex> lk(g_defer_mu);
        for (size_t i = 0; i < g_deferred.size(); i++)
            if (g_deferred[i] == ctx) { g_deferred.erase(g_deferred.begin() + (long)i); break; }
        ctx->back_deferred = false;
        for (size_t i = 0; i < g_held.size(); i++)
            if (g_held[i] == ctx) { g_held.erase(g_held.begin() + (long)i); break; }
        ctx->rest_held = false;
    }
    if (ctx->stream && ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    {   // the device's shared streams may still hold work of this context: handles copied under the mutex, waited for outside it
        DeviceStreams ds;   // (a destroy must not stall every other thread's first use of a stream for the length of the queued work)
        { std::lock_guard<std::mutex> lk(g_streams_mu); ds = g_dev_streams[ctx->device & 63]; }
        if (ds.decode) (void)hipStreamSynchronize(ds.decode);
        if (ds.copy) (void)hipStreamSynchronize(ds.copy);
        if (ds.k1) (void)hipStreamSynchronize(ds.k1);
    }
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    DevBuf* bufs[] = {&ctx->dict, &ctx->in, &ctx->grey, &ctx->bin, &ctx->frame_darts, &ctx->frame_darts_dev, &ctx->frame_base, &ctx->pix_base,
                      &ctx->tile_darts, &ctx->d_xy, &ctx->d_succ, &ctx->stA, &ctx->stB, &ctx->t_cur, &ctx->t_next,
                      &ctx->leader_list, &ctx->leader_keep, &ctx->entry_list, &ctx->es_a, &ctx->es_b,
                      &ctx->contours, &ctx->cyc_start_off, &ctx->points, &ctx->zero_blk, &ctx->cands,
                      &ctx->pre_xy, &ctx->fin_xy, &ctx->fin_count, &ctx->work, &ctx->outs, &ctx->proj, &ctx->patches, &ctx->cand_big,
                      &ctx->tmp_a, &ctx->tmp_b, &ctx->tmp_c, &ctx->tmp_d, &ctx->hsum, &ctx->pose_buf, &ctx->wtab};
    for (DevBuf* b : bufs) b->release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_counts) (void)hipHostFree(ctx->pinned_counts);
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->ev_contours) (void)hipEventDestroy(ctx->ev_contours);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->ev_gate) (void)hipEventDestroy(ctx->ev_gate);
    if (ctx->ev_k1_ready) (void)hipEventDestroy(ctx->ev_k1_ready);
    if (ctx->ev_k1_done) (void)hipEventDestroy(ctx->ev_k1_done);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int a3_set_stream(a3_ctx* ctx, void* hip_stream) {
    if (!ctx) return A3_ERR_INVALID;
    {   // a chain held back for a burst belongs on the stream its threshold kernel went to: it goes out before the stream changes
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (ctx->rest_held) (void)flush_held_locked(ctx, nullptr);
    }
    {   // (batch_mode_of reads every live context's stream under this lock, possibly from another thread's submit)
        std::lock_guard<std::mutex> lk(g_streams_mu);
        ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;   // (may be null until first needed)
    }
    return A3_OK;
}

// pinned host memory for frames: from it the H2D copy of a3_detect_batch(_submit) is asynchronous and runs at the link's rate
int a3_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return A3_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, A3_ERR_HIP, "hipHostMalloc", e);
    return A3_OK;
}
int a3_host_free(void* p) {
    if (!p) return A3_OK;
    const hipError_t e = hipHostFree(p);
    return e == hipSuccess ? A3_OK : fail(nullptr, A3_ERR_HIP, "hipHostFree", e);
}
int a3_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return A3_ERR_INVALID;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    return e == hipSuccess ? A3_OK : fail(nullptr, A3_ERR_HIP, "hipHostRegister", e);
}
int a3_host_unregister(void* p) {
    if (!p) return A3_ERR_INVALID;
    const hipError_t e = hipHostUnregister(p);
    return e == hipSuccess ? A3_OK : fail(nullptr, A3_ERR_HIP, "hipHostUnregister", e);
}

// The next batch submitted on `ctx` starts on the device only after everything enqueued so far on `other` (its batch in flight
// included) has finished.  A scheduling hint for callers that keep several contexts in flight (include/aruco3_hip.h, "bursts"):
// results do not depend on it.
int a3_order_after(a3_ctx* ctx, a3_ctx* other) {
    if (!ctx || !other) return A3_ERR_INVALID;
    if (ctx == other || ctx->device != other->device) return A3_OK;   // (a context's own batches are in order anyway; other devices do not share a chip)
    A3_HIP(hipSetDevice(ctx->device));
    if (int rc = need_stream(ctx)) return rc;
    if (!other->stream) return A3_OK;   // never used: nothing in flight
    {   // a decode stage still held back would not be covered by an event on the owner's stream: it goes out now
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (other->back_deferred) { a3_ctx* o = other; if (int rc = flush_deferred_locked(o, nullptr)) { ctx->err = o->err; return rc; } }
        if (other->rest_held) (void)flush_held_locked(other, nullptr);   // (likewise a held chain; its verdict is its owner's)
    }
    if (other->stream == ctx->stream) return A3_OK;   // one stream: already ordered
    A3_HIP(hipEventRecord(other->ev_gate, other->stream));
    A3_HIP(hipStreamWaitEvent(ctx->stream, other->ev_gate, 0));
    ctx->gates_declared = true;
    return A3_OK;
}

// internal (a3_internal.h, tools/spin_probe.py): work enqueued on `hip_stream` after this call starts only once the threshold kernel
// of ctx's batch in flight has finished; needs a3_debug_set_mark_threshold(1) before the submit
int a3_debug_stream_wait_threshold(a3_ctx* ctx, void* hip_stream) {
    if (!ctx) return A3_ERR_INVALID;
    if (!ctx->pending.active || !ctx->k1_marked) return A3_OK;   // nothing in flight (or a synchronous call): nothing to wait for
    A3_HIP(hipSetDevice(ctx->device));
    A3_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), ctx->ev_k1_done, 0));
    return A3_OK;
}

int a3_get_stream(const a3_ctx* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return A3_ERR_INVALID;
    if (int rc = need_stream(const_cast<a3_ctx*>(ctx))) return rc;
    *hip_stream = reinterpret_cast<void*>(ctx->stream);
    return A3_OK;
}

int a3_set_pool_limits(a3_ctx* ctx, uint64_t max_darts, uint64_t max_points) {
    if (!ctx) return A3_ERR_INVALID;
    if (max_darts) ctx->max_darts = std::min<uint64_t>(max_darts, kHardMaxDarts);
    if (max_points) ctx->max_points = std::min<uint64_t>(max_points, kHardMaxPoints);
    return A3_OK;
}

int a3_set_debug_taps(a3_ctx* ctx, int enabled) {
    if (!ctx) return A3_ERR_INVALID;
    ctx->debug_taps = enabled != 0;
    return A3_OK;
}

int a3_get_tau(const a3_ctx* ctx, uint8_t* tau) {
    if (!ctx || !tau) return A3_ERR_INVALID;
    *tau = ctx->tau;
    return A3_OK;
}

// argument checks + H2D staging shared by the synchronous and the split entry points.  -> A3_OK, an error, or
// kNothingToDo (no frames / empty images: the answer is "no markers").
static constexpr int kNothingToDo = 2;
static int stage_input(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t* row_stride,
                       size_t* frame_stride, uint32_t n_frames, const uint8_t** d_pixels) {
    if (n_frames == 0) return kNothingToDo;
    if (!pixels) return fail(ctx, A3_ERR_INVALID, "a3_detect_batch: null pixels");
    if (fmt != A3_FMT_RGB8 && fmt != A3_FMT_RGBA8 && fmt != A3_FMT_L8 && fmt != A3_FMT_BGRA8) return fail(ctx, A3_ERR_INVALID, "unknown pixel format");
    if (width == 0 || height == 0) return kNothingToDo;   // an empty image has no contours
    if (n_frames > 65535) return fail(ctx, A3_ERR_INVALID, "more than 65535 frames in one call (split the batch)");
    if (width > 65535 || height > 65535 || (uint64_t)width * height >= (1ull << 30))
        return fail(ctx, A3_ERR_INVALID, "image dimensions above 65535 (or 2^30 pixels) are not supported");
    const size_t bpp = fmt == A3_FMT_RGB8 ? 3 : (fmt == A3_FMT_L8 ? 1 : 4);
    if (*row_stride == 0) *row_stride = (size_t)width * bpp;
    if (*row_stride < (size_t)width * bpp) return fail(ctx, A3_ERR_INVALID, "row_stride smaller than a row");
    if (*frame_stride == 0) *frame_stride = *row_stride * height;
    if (*frame_stride < *row_stride * (height - 1) + (size_t)width * bpp) return fail(ctx, A3_ERR_INVALID, "frame_stride smaller than a frame");
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    *d_pixels = reinterpret_cast<const uint8_t*>(pixels);
    if (memory == A3_MEM_HOST) {
        const size_t bytes = *frame_stride * (n_frames - 1) + *row_stride * (height - 1) + (size_t)width * bpp;
        A3_HIP(ctx->in.ensure(bytes));
        // On the copy stream, not the compute stream: the kernels of another context's batch (submit / collect with two
        // contexts) keep running while these frames cross the link.  Pageable memory is staged by the runtime and the call
        // returns when the caller's buffer has been read; pinned memory (a3_host_alloc / a3_host_register) makes the copy
        // asynchronous and the buffer must then stay untouched until the batch is collected.
        hipStream_t cs = nullptr;
        A3_HIP(device_stream(ctx->device, kStreamCopy, &cs));
        A3_HIP(hipMemcpyAsync(ctx->in.p, pixels, bytes, hipMemcpyHostToDevice, cs));
        A3_HIP(hipEventRecord(ctx->ev_in, cs));
        A3_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
        *d_pixels = ctx->in.as<uint8_t>();
    } else if (memory != A3_MEM_DEVICE) return fail(ctx, A3_ERR_INVALID, "memory must be A3_MEM_HOST or A3_MEM_DEVICE");
    return A3_OK;
}

static int run_batch_with_retries(a3_ctx* ctx, const uint8_t* d_pixels, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                                  size_t frame_stride, uint32_t n_frames, a3_marker* out, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    for (int attempt = 0; attempt < 8; attempt++) {
        const int rc = run_batch(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out, out_cap, per_frame_count, out_n);
        if (rc != 1) return rc;
        ctx->reruns++;
    }
    return fail(ctx, A3_ERR_CAPACITY, "contour pools kept overflowing");
}

int a3_detect_batch(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                    size_t frame_stride, uint32_t n_frames, a3_marker* out, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    if (!ctx) return A3_ERR_INVALID;
    if (!out_n || (!out && out_cap)) return fail(ctx, A3_ERR_INVALID, "a3_detect_batch: null output");
    *out_n = 0;
    if (ctx->pending.active) return fail(ctx, A3_ERR_INVALID, "a submitted batch has not been collected");
    const uint8_t* d_pixels = nullptr;
    const int rc = stage_input(ctx, pixels, memory, fmt, width, height, &row_stride, &frame_stride, n_frames, &d_pixels);
    if (rc == kNothingToDo) {
        if (per_frame_count && n_frames) memset(per_frame_count, 0, (size_t)n_frames * 4);
        return A3_OK;
    }
    if (rc != A3_OK) return rc;
    ctx->force_host_plan = false;
    ctx->reruns = 0; ctx->released_others = 0;
    return run_batch_with_retries(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out, out_cap, per_frame_count, out_n);
}

static int submit_common(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                         size_t frame_stride, uint32_t n_frames, size_t out_cap, bool want_pose) {
    if (ctx->pending.active || ctx->pending_trivial) return fail(ctx, A3_ERR_INVALID, "a submitted batch has not been collected");
    const uint8_t* d_pixels = nullptr;
    const int rc = stage_input(ctx, pixels, memory, fmt, width, height, &row_stride, &frame_stride, n_frames, &d_pixels);
    if (rc == kNothingToDo) { ctx->pending_trivial = true; ctx->pending.n = n_frames; return A3_OK; }
    if (rc != A3_OK) return rc;
    ctx->force_host_plan = false;
    ctx->want_pose = want_pose;
    Pending& pd = ctx->pending;
    pd.pixels = d_pixels; pd.fmt = fmt; pd.row_stride = row_stride; pd.frame_stride = frame_stride; pd.want_pose = want_pose;
    // Bursts: a context that declared gates (a3_order_after) since its last submit holds its chain back behind its threshold
    // kernel; a submit without gates is the last member of its burst and releases every held chain of the device behind ITS
    // threshold kernel.  This is the library's behaviour behind the public header -- no switch selects it.  Exceptions, all of
    // them "enqueue the whole batch now": every stage is being timed (the stage times are those of stages that run alone); the
    // batch needs a host-side plan (first batch of a shape, or a graph that outgrew the previous plan: the plan waits for the
    // device, and a chain enqueued later by another thread must not); the context SHARES its stream with another context
    // (batch_mode_of != 0: they are in order there anyway, a3_order_after is a no-op, the decode stage is deferred instead);
    // a forced mode 1 / 2 (a3_debug_set_overlap: round 4's default, for A/B).  A context alone on a caller's stream IS held like
    // one on a stream of its own: its chain then lands on that stream behind whatever the caller queued after the submit
    // (stated in the header).
    const bool gated = ctx->gates_declared;
    ctx->gates_declared = false;
    ctx->batch_mode = batch_mode_of(ctx);
    ctx->released_others = 0; ctx->reruns = 0;
    const bool bursts = ctx->batch_mode == 0 && g_overlap_force.load(std::memory_order_relaxed) <= 0 && g_hold_rests && ctx->profiling < 2;
    if (bursts && gated && device_plan_capacity(ctx, n_frames, width, height) != 0) {
        pd.n = n_frames; pd.W = width; pd.H = height;
        ctx->held_out_cap = out_cap;
        if (int erc = enqueue_batch(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, 1)) return erc;
        (void)hipStreamQuery(ctx->stream);
        std::lock_guard<std::mutex> lk(g_defer_mu);
        pd.active = true;
        ctx->rest_held = true; ctx->held_rc = 0;
        ctx->stepping = A3_STEP_HELD;   // (until the chain goes out: flush_held_locked says how)
        g_held.push_back(ctx);
        return A3_OK;
    }
    bool any_held = false;
    if (bursts) {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        for (a3_ctx* o : g_held) any_held |= (o != ctx && o->device == ctx->device);
    }
    if (any_held) {   // the last member: threshold kernel, then the held chains of the others behind it, then this batch's own
        // A last member that needs a host-side plan cannot be split (phase 1 refuses it): the others are then released ungated,
        // ahead of its whole batch.
        const bool planned = device_plan_capacity(ctx, n_frames, width, height) != 0;
        if (planned) { if (int erc = enqueue_batch(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, 1)) return erc; }
        uint32_t released = 0;
        {
            std::lock_guard<std::mutex> lk(g_defer_mu);
            const std::vector<a3_ctx*> list = g_held;   // (flush edits g_held)
            for (a3_ctx* o : list)
                if (o != ctx && o->device == ctx->device) { (void)flush_held_locked(o, planned ? ctx->ev_k1_done : nullptr, true); released++; }   // (a failure is o's: its collect reports it)
        }
        const int erc = enqueue_batch(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, planned ? 2 : 0);
        ctx->stepping = A3_STEP_BURST_LAST; ctx->released_others = released;
        return erc;
    }
    ctx->allow_defer = true;    // (a synchronous call, or a re-run, enqueues both halves at once)
    const int erc = enqueue_batch(ctx, d_pixels, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, 0);
    ctx->allow_defer = false;
    return erc;
}

static int collect_common(a3_ctx* ctx, a3_marker* out, a3_pose* poses, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    *out_n = 0;
    if (ctx->pending_trivial) {
        ctx->pending_trivial = false;
        if (per_frame_count && ctx->pending.n) memset(per_frame_count, 0, (size_t)ctx->pending.n * 4);
        return A3_OK;
    }
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    const Pending pd = ctx->pending;   // finish_batch clears .active
    ctx->want_pose = pd.want_pose;     // (the other half of the pair may have been a different kind of call on this context)
    ctx->pose_out = pd.want_pose ? poses : nullptr;
    int rc = finish_batch(ctx, out, out_cap, per_frame_count, out_n);
    const uint32_t stepping = ctx->stepping;   // how the SUBMITTED batch was stepped (a re-run below is a synchronous call of its own)
    // the device asked for a re-run (pool growth, more passes, host-side plan): do it synchronously
    if (rc == 1) { ctx->reruns++; rc = run_batch_with_retries(ctx, pd.pixels, pd.fmt, pd.W, pd.H, pd.row_stride, pd.frame_stride, pd.n, out, out_cap, per_frame_count, out_n); }
    ctx->stepping = stepping;
    ctx->want_pose = false;
    ctx->pose_out = nullptr;
    return rc;
}

int a3_detect_batch_submit(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                           size_t frame_stride, uint32_t n_frames, size_t out_cap) {
    if (!ctx) return A3_ERR_INVALID;
    return submit_common(ctx, pixels, memory, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, false);
}

int a3_detect_batch_collect(a3_ctx* ctx, a3_marker* out, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    if (!ctx) return A3_ERR_INVALID;
    if (!out_n || (!out && out_cap)) return fail(ctx, A3_ERR_INVALID, "a3_detect_batch_collect: null output");
    return collect_common(ctx, out, nullptr, out_cap, per_frame_count, out_n);
}

// detect + pose in two halves (BASELINE config 5 pipelined like config 2): the pose request travels with the submitted batch
int a3_detect_batch_pose_submit(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                                size_t frame_stride, uint32_t n_frames, float marker_size_mm, const a3_intrinsics* intr, size_t out_cap) {
    if (!ctx) return A3_ERR_INVALID;
    if (ctx->pending.active || ctx->pending_trivial) return fail(ctx, A3_ERR_INVALID, "a submitted batch has not been collected");
    ctx->pose_size_mm = marker_size_mm;
    ctx->pose_has_intr = intr != nullptr;
    if (intr) ctx->pose_intr = *intr;
    const int rc = submit_common(ctx, pixels, memory, fmt, width, height, row_stride, frame_stride, n_frames, out_cap, true);
    ctx->want_pose = false;   // (enqueue_batch has read it; a plain call in between must not inherit it)
    return rc;
}

int a3_detect_batch_pose_collect(a3_ctx* ctx, a3_marker* out, a3_pose* poses, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    if (!ctx) return A3_ERR_INVALID;
    if (!out_n || (!out && out_cap) || (!poses && out_cap)) return fail(ctx, A3_ERR_INVALID, "a3_detect_batch_pose_collect: null output");
    if (!ctx->pending_trivial && ctx->pending.active && !ctx->pending.want_pose)
        return fail(ctx, A3_ERR_INVALID, "a3_detect_batch_pose_collect: the submitted batch was not an a3_detect_batch_pose_submit call");
    return collect_common(ctx, out, poses, out_cap, per_frame_count, out_n);
}

int a3_detect_batch_pose(a3_ctx* ctx, const void* pixels, int memory, int fmt, uint32_t width, uint32_t height, size_t row_stride,
                         size_t frame_stride, uint32_t n_frames, float marker_size_mm, const a3_intrinsics* intr, a3_marker* out,
                         a3_pose* poses, size_t out_cap, uint32_t* per_frame_count, size_t* out_n) {
    if (!ctx || !poses) return A3_ERR_INVALID;
    ctx->want_pose = true;
    ctx->pose_size_mm = marker_size_mm;
    ctx->pose_has_intr = intr != nullptr;
    if (intr) ctx->pose_intr = *intr;
    ctx->pose_out = poses;
    const int rc = a3_detect_batch(ctx, pixels, memory, fmt, width, height, row_stride, frame_stride, n_frames, out, out_cap, per_frame_count, out_n);
    ctx->want_pose = false;
    ctx->pose_out = nullptr;
    return rc;
}

// Re-runs one contour kernel on the buffers of the last batch (single-chunk batches only) and returns its average device
// time; dbg selects a truncated variant (see the kernels).  The contour graph buffers hold garbage afterwards, which the
// next a3_detect_batch overwrites; nothing reads them in between.
int a3_debug_kernel_time(a3_ctx* ctx, int kernel, int dbg, int reps, float* avg_ms) {
    if (!ctx || !avg_ms || reps <= 0) return A3_ERR_INVALID;
    if (ctx->dbg_chunks != 1 || ctx->dbg_nd == 0) return fail(ctx, A3_ERR_INVALID, "needs a preceding single-chunk batch");
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    hipStream_t st = ctx->stream;
    // the events are owned by a guard: every early return below (A3_HIP) destroys them
    struct EventPair {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } evp;
    A3_HIP(hipEventCreate(&evp.e0)); A3_HIP(hipEventCreate(&evp.e1));
    const hipEvent_t e0 = evp.e0, e1 = evp.e1;
    double total = 0.0;
    unsigned int* d_entry_count = ctx->scratch_u32 + 32;
    unsigned int* d_leader_count = ctx->scratch_u32 + 16;
    for (int r = 0; r < reps; r++) {
        A3_HIP(hipMemsetAsync(d_leader_count, 0, 4 * 32, st));
        A3_HIP(hipEventRecord(e0, st));
        if (kernel == 0) {
            A3_HIP(hipMemsetAsync(ctx->frame_darts.p, 0, (size_t)ctx->frames * 8, st));
            A3_HIP(launch_dart_count(st, ctx->bin.as<uint64_t>(), (int)ctx->W, (int)ctx->H, 0, ctx->frames, ctx->frame_darts.as<unsigned long long>(),
                                     ctx->tile_darts.as<uint32_t>(), 0, nullptr, nullptr, nullptr, 0));
        } else if (kernel == 1) {
            A3_HIP(launch_dart_build(st, ctx->bin.as<uint64_t>(), (int)ctx->W, (int)ctx->H, 0, ctx->dbg_frames, ctx->frame_base.as<uint32_t>(),
                                     ctx->tile_darts.as<uint32_t>() + tile_off_offset(ctx->W, ctx->H, ctx->frames), ctx->pix_base.as<uint32_t>(),
                                     ctx->tile_darts.as<uint32_t>(), ctx->d_xy.as<uint64_t>(),
                                     ctx->d_succ.as<uint32_t>(), ctx->dbg_nd, nullptr, dbg ? dbg : 5,
                                     reinterpret_cast<const unsigned long long*>(ctx->tile_darts.as<uint8_t>() + tile_mask_offset_bytes(ctx->W, ctx->H, ctx->frames))));
        } else if (kernel == 2) {
            A3_HIP(launch_rank_cycles(st, ctx->dbg_nd, (int)ctx->W, ctx->d_xy.as<uint64_t>(), ctx->d_succ.as<uint32_t>(), ctx->stA.as<JumpState>(),
                                      ctx->entry_list.as<uint32_t>(),
                                      d_entry_count, ctx->es_a.p, ctx->es_b.p, ctx->stB.p,
                                      ctx->leader_list.as<uint32_t>(), d_leader_count, 0, ctx->counters, nullptr, dbg ? dbg : 11, ctx->frame_base.as<uint32_t>(), nullptr, ctx->dbg_frames, 0,
                                      0u, nullptr, 0));
        } else if (kernel == 3 || kernel == 4) {   // dbg < 0: k_decode alone (variant -dbg; -5: the whole kernel), dbg >= 0: k_projection + k_decode
            if (kernel == 4) {   // COLD frames, as the pipeline meets them (1.6 GB went through the threshold kernel, the contour stage's buffers since):
                                 // half a gigabyte of the pixel-base plane is overwritten first (garbage after a batch anyway), outside the timed span
                const size_t sweep = std::min<size_t>(ctx->pix_base.cap & ~(size_t)15, (size_t)512 << 20);
                if (sweep) A3_HIP(launch_zero(st, ctx->pix_base.p, sweep));
                A3_HIP(hipEventRecord(e0, st));
            }
            A3_HIP(launch_decode(st, ctx->dbg_src, (int)ctx->W, (int)ctx->H, 0, ctx->fin_xy.as<uint16_t>(), ctx->work.as<uint32_t>(), ctx->scratch_u32,
                                 ctx->max_cand, ctx->cfg.homography_sample_size, ctx->mark_size, ctx->cfg.homography_sample_size, ctx->dict.as<uint64_t>(),
                                 ctx->n_codes, ctx->tau, ctx->cfg.filter_high_bit_errors, ctx->proj.p, ctx->wtab.as<float>(), ctx->outs.p, nullptr, 0u, nullptr, 4096,
                                 dbg == 0 ? -1000 : dbg, ctx->frames <= 64u ? 1 : 0));   // 0: k_projection + k_decode, < 0: k_decode alone (variant -dbg)
        } else return fail(ctx, A3_ERR_INVALID, "kernel: 0 dart_count, 1 dart_assign, 2 local_contract, 3 decode, 4 decode on cold frames");
        A3_HIP(hipEventRecord(e1, st));
        A3_HIP(hipStreamSynchronize(st));
        float ms = 0;
        A3_HIP(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    *avg_ms = (float)(total / reps);
    return A3_OK;
}

// internal (a3_internal.h): where a submitted batch's decode stage is released (0 not deferred, 1 behind the next batch's
// threshold kernel, 2 behind its k_local_contract); process-wide, for A/B measurements
int a3_debug_set_overlap(int mode) {   // -1: the library decides per batch (default); bits 0-7: forced mode; bit 8: a decode stream created from now on gets the LOWEST priority
    if (mode != -1 && (mode < 0 || (mode & 0xFF) > 2)) return A3_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!g_held.empty() || !g_deferred.empty()) return A3_ERR_INVALID;   // batches in flight were submitted under the old mode: collect them first
    g_overlap_force.store(mode == -1 ? -1 : (mode & 0xFF), std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk2(g_streams_mu); g_decode_low_prio = mode != -1 && (mode & 0x100) != 0; }
    return A3_OK;
}

// a kernel of `workgroups` x `threads` that stays resident for `usec` microseconds on `hip_stream` (stand-in for a collective's
// channel kernels while only one GPU is at hand); returns at once
int a3_debug_spin(void* hip_stream, int workgroups, int threads, int usec) {
    if (workgroups < 1 || workgroups > 1024 || threads < 64 || threads > 512 || threads % 64 || usec < 1 || usec > 100000) return A3_ERR_INVALID;
    static uint32_t* sink = nullptr;
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!sink) {
            if (hipMalloc(&sink, 4096 * 4) != hipSuccess) return A3_ERR_HIP;
            if (hipMemset(sink, 0, 4096 * 4) != hipSuccess) return A3_ERR_HIP;
        }
    }
    return launch_spin(reinterpret_cast<hipStream_t>(hip_stream), workgroups, threads, usec, sink) == hipSuccess ? A3_OK : A3_ERR_HIP;
}

// bit 0: built with -DA3_TUNING (environment knobs are read), bit 1: any other non-default build flag of the kernels
int a3_debug_build_flags(void) {
    int f = 0;
#ifdef A3_TUNING
    f |= 1;
#endif
    f |= k1_build_is_default() ? 0 : 2;
    return f;
}

int a3_debug_set_partition(int k1_cus, int pattern) {   // before the first context of the process is used
    if (k1_cus < 0 || k1_cus > 248) return A3_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    g_part_k1_cus = k1_cus; g_part_pattern = pattern;
    set_k1_cus(k1_cus > 0 ? k1_cus : 256);
    return A3_OK;
}

// the threshold kernel alone on the context's stream, asynchronously (buffers of a preceding batch of the same shape are re-used)
int a3_debug_launch_threshold(a3_ctx* ctx, const void* pixels_device, int fmt, uint32_t width, uint32_t height, uint32_t n_frames) {
    if (!ctx || !pixels_device || n_frames == 0) return A3_ERR_INVALID;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rc = need_stream(ctx)) return rc;
    const size_t bpp = fmt == A3_FMT_RGB8 ? 3 : (fmt == A3_FMT_L8 ? 1 : 4);
    if (threshold_writes_grey_plane(ctx->cfg.threshold_window, reinterpret_cast<const uint8_t*>(pixels_device), (size_t)width * bpp, (size_t)width * bpp * height, (int)width))
        return fail(ctx, A3_ERR_INVALID, "a3_debug_launch_threshold: this threshold_window / frame layout takes the separable path, which needs a grey plane");
    A3_HIP(ctx->bin.ensure((size_t)words_per_row(width) * 8 * height * n_frames));
    A3_HIP(launch_grey_threshold(ctx->stream, reinterpret_cast<const uint8_t*>(pixels_device), fmt, (size_t)width * bpp, (size_t)width * bpp * height,
                                 (int)width, (int)height, n_frames, ctx->cfg.threshold_window, nullptr, ctx->bin.as<uint64_t>(), nullptr));
    return A3_OK;
}

int a3_debug_set_mark_threshold(int on) { g_mark_threshold = on != 0; return A3_OK; }
int a3_debug_set_k1_stream(int mode) {   // before the first context of the process is used (the device-wide stream is created once)
    if (mode < 0 || mode > 2) return A3_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    g_k1_stream_prio = mode;
    return A3_OK;
}
int a3_debug_set_hold(int on) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!g_held.empty()) return A3_ERR_INVALID;   // (chains held under the old setting: collect them first)
    g_hold_rests = on != 0;
    return A3_OK;
}

int a3_debug_set_k1_waves(int waves_per_simd) { set_k1_waves(waves_per_simd); return A3_OK; }

int a3_synth_render(int device, void* hip_stream, const a3_synth_frame* frames, uint32_t n_frames, const a3_synth_marker* markers,
                    uint32_t n_markers, uint32_t width, uint32_t height, int paper, float black, float white, int supersample,
                    void* out_rgb_device, size_t row_stride, size_t frame_stride) {
    if (!frames || !out_rgb_device || (n_markers && !markers) || supersample < 1 || supersample > 8) return A3_ERR_INVALID;
    if (n_frames == 0 || width == 0 || height == 0) return A3_OK;
    if (row_stride == 0) row_stride = (size_t)width * 3;
    if (frame_stride == 0) frame_stride = row_stride * height;
    if (row_stride < (size_t)width * 3 || frame_stride < row_stride * (height - 1) + (size_t)width * 3 || height > 65535 || n_frames > 65535)
        return A3_ERR_INVALID;
    for (uint32_t f = 0; f < n_frames; f++)
        if ((uint64_t)frames[f].first_marker + frames[f].n_markers > n_markers) return A3_ERR_INVALID;
    for (uint32_t m = 0; m < n_markers; m++)
        if (markers[m].n == 0 || markers[m].n > 11) return A3_ERR_INVALID;   // 121 cells in two words (CHILITAGS: 10 x 10)
    if (hipSetDevice(device) != hipSuccess) return A3_ERR_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    a3_synth_frame* d_frames = nullptr; a3_synth_marker* d_markers = nullptr;
    hipError_t e = hipMalloc(&d_frames, sizeof(a3_synth_frame) * n_frames);
    if (e == hipSuccess && n_markers) e = hipMalloc(&d_markers, sizeof(a3_synth_marker) * n_markers);
    if (e == hipSuccess) e = hipMemcpyAsync(d_frames, frames, sizeof(a3_synth_frame) * n_frames, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_markers) e = hipMemcpyAsync(d_markers, markers, sizeof(a3_synth_marker) * n_markers, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_synth_render(st, d_frames, n_frames, d_markers, width, height, paper, black, white, supersample,
                                                 reinterpret_cast<uint8_t*>(out_rgb_device), row_stride, frame_stride);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_frames) (void)hipFree(d_frames);
    if (d_markers) (void)hipFree(d_markers);
    return e == hipSuccess ? A3_OK : A3_ERR_HIP;
}

int a3_get_stats(const a3_ctx* ctx, a3_stats* stats) {
    if (!ctx || !stats) return A3_ERR_INVALID;
    *stats = ctx->stats;
    stats->stepping = ctx->stepping | (std::min(ctx->released_others, 255u) << 8) | (std::min(ctx->reruns, 255u) << 16);
    return A3_OK;
}

static int download_plane(a3_ctx* ctx, const DevBuf& buf, uint32_t frame, uint8_t* dst) {
    if (!ctx || !dst) return A3_ERR_INVALID;
    if (frame >= ctx->frames) return fail(ctx, A3_ERR_INVALID, "frame index outside the last batch");
    const size_t npx = (size_t)ctx->W * ctx->H;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(hipMemcpy(dst, buf.as<uint8_t>() + npx * frame, npx, hipMemcpyDeviceToHost));
    return A3_OK;
}

int a3_download_grey(a3_ctx* ctx, uint32_t frame, uint8_t* dst) {
    if (!ctx) return A3_ERR_INVALID;
    if (!ctx->grey_valid) return fail(ctx, A3_ERR_INVALID, "the grey plane is only kept when debug taps are enabled before the batch");
    return download_plane(ctx, ctx->grey, frame, dst);
}
int a3_download_thresholded(a3_ctx* ctx, uint32_t frame, uint8_t* dst) {
    if (!ctx || !dst) return A3_ERR_INVALID;
    if (frame >= ctx->frames) return fail(ctx, A3_ERR_INVALID, "frame index outside the last batch");
    const size_t npx = (size_t)ctx->W * ctx->H, wpf = (size_t)words_per_row(ctx->W) * ctx->H;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(ctx->tmp_a.ensure(npx));
    A3_HIP(launch_unpack_bits(ctx->stream, ctx->bin.as<uint64_t>() + wpf * frame, (int)ctx->W, (int)ctx->H, ctx->tmp_a.as<uint8_t>()));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    A3_HIP(hipMemcpy(dst, ctx->tmp_a.p, npx, hipMemcpyDeviceToHost));
    return A3_OK;
}

int a3_candidate_count(a3_ctx* ctx, uint32_t frame, uint32_t* n_pre, uint32_t* n_final) {
    if (!ctx) return A3_ERR_INVALID;
    if (frame >= ctx->frames) return fail(ctx, A3_ERR_INVALID, "frame index outside the last batch");
    if (ctx->counts_valid && frame < ctx->h_cand_fin.size()) {   // the tapped batch brought them along
        if (n_pre) *n_pre = ctx->h_cand_pre[frame];
        if (n_final) *n_final = ctx->h_cand_fin[frame];
        return A3_OK;
    }
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    uint32_t a = 0, b = 0;
    A3_HIP(hipMemcpy(&a, ctx->cand_count + frame, 4, hipMemcpyDeviceToHost));
    A3_HIP(hipMemcpy(&b, ctx->fin_count.as<uint32_t>() + frame, 4, hipMemcpyDeviceToHost));
    if (n_pre) *n_pre = std::min(a, ctx->max_cand);
    if (n_final) *n_final = b;
    return A3_OK;
}

int a3_download_candidates(a3_ctx* ctx, uint32_t frame, int before_discard, uint32_t* dst_xy, size_t cap_quads) {
    if (!ctx || !dst_xy) return A3_ERR_INVALID;
    uint32_t a = 0, b = 0;
    if (int rc = a3_candidate_count(ctx, frame, &a, &b)) return rc;
    const uint32_t cnt = before_discard ? a : b;
    if (cnt > cap_quads) return fail(ctx, A3_ERR_CAPACITY, "cap_quads too small");
    std::vector<uint16_t> tmp((size_t)cnt * 8);
    const DevBuf& src = before_discard ? ctx->pre_xy : ctx->fin_xy;
    if (cnt) A3_HIP(hipMemcpy(tmp.data(), src.as<uint16_t>() + (size_t)frame * ctx->max_cand * 8, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); i++) dst_xy[i] = tmp[i];
    return A3_OK;
}

struct DecodeOutHost { uint64_t code; uint64_t codes[4]; uint32_t id; uint8_t valid, rotation, hamming, hom_ok; int32_t decode_ok; uint32_t patch; };

int a3_download_homographies(a3_ctx* ctx, uint32_t frame, uint8_t* dst, uint8_t* ok, uint64_t* codes4, int32_t* decode_ok, size_t cap) {
    if (!ctx) return A3_ERR_INVALID;
    uint32_t b = 0;
    if (int rc = a3_candidate_count(ctx, frame, nullptr, &b)) return rc;
    if (b > cap) return fail(ctx, A3_ERR_CAPACITY, "cap too small");
    if (sizeof(DecodeOutHost) != decode_out_bytes()) return fail(ctx, A3_ERR_INTERNAL, "DecodeOut layout mismatch");
    if (dst && !ctx->debug_taps) return fail(ctx, A3_ERR_INVALID, "patches are only kept after a3_set_debug_taps(ctx, 1)");
    if (ctx->pending.active) return fail(ctx, A3_ERR_INVALID, "a submitted batch has not been collected");   // (its staging buffer is in use)
    if (b == 0) return A3_OK;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    const uint32_t S = ctx->cfg.homography_sample_size;
    const size_t S2 = (size_t)S * S, rec_bytes = (size_t)b * sizeof(DecodeOutHost);
    const uint8_t* d_outs = (const uint8_t*)ctx->outs.p + (size_t)frame * ctx->max_cand * sizeof(DecodeOutHost);
    // One trip: the frame's decode records, and its patches gathered by a kernel into one dense array, land in pinned memory
    // through two copies on the stream and one wait (it was a blocking copy per candidate).
    const size_t flag_off = (rec_bytes + 15) & ~(size_t)15, patch_off = flag_off + 16;
    const size_t total = patch_off + (dst ? (size_t)b * S2 : 0);
    A3_HIP(ctx->tmp_a.ensure(total));
    uint8_t* d_tmp = ctx->tmp_a.as<uint8_t>();
    unsigned int* d_missing = reinterpret_cast<unsigned int*>(d_tmp + flag_off);
    hipStream_t st = ctx->stream;
    if (dst) {
        A3_HIP(hipMemsetAsync(d_missing, 0, 4, st));
        A3_HIP(launch_gather_patches(st, d_outs, b, ctx->patches.as<uint8_t>(), ctx->patch_cap, (uint32_t)S2, d_tmp + patch_off, d_missing));
    }
    if (int rc = ensure_pinned(ctx, total + (1 << 16))) return rc;
    uint8_t* hp = (uint8_t*)ctx->pinned;
    A3_HIP(hipMemcpyAsync(hp, d_outs, rec_bytes, hipMemcpyDeviceToHost, st));
    if (dst) A3_HIP(hipMemcpyAsync(hp + flag_off, d_tmp + flag_off, 16 + (size_t)b * S2, hipMemcpyDeviceToHost, st));
    A3_HIP(wait_stream(st));
    const DecodeOutHost* o = reinterpret_cast<const DecodeOutHost*>(hp);
    for (uint32_t k = 0; k < b; k++) {
        if (ok) ok[k] = o[k].hom_ok;
        if (decode_ok) decode_ok[k] = o[k].decode_ok;
        if (codes4) for (int r = 0; r < 4; r++) codes4[4 * k + r] = o[k].codes[r];
    }
    if (dst) {
        memcpy(dst, hp + patch_off, (size_t)b * S2);
        if (*reinterpret_cast<const unsigned int*>(hp + flag_off))
            return fail(ctx, A3_ERR_CAPACITY, "the patch tap was full: a candidate's patch was not kept (split the batch: at most 1024 frames per tapped call)");
    }
    return A3_OK;
}

// ---- Detection gather records on the device (multi-GPU) ----
size_t a3_detection_record_bytes(uint32_t max_markers_per_frame, int with_poses) {
    return 8 + (size_t)max_markers_per_frame * (sizeof(a3_marker) + (with_poses ? 2 * sizeof(a3_pose) : 0));
}

int a3_pack_detections(a3_ctx* ctx, uint32_t first_frame_global, uint32_t max_markers_per_frame, int with_poses, void* dst_device, size_t dst_bytes) {
    if (!ctx || !dst_device) return A3_ERR_INVALID;
    if (!ctx->markers_valid) return fail(ctx, A3_ERR_INVALID, "a3_pack_detections: no finished batch on this context");
    if (with_poses && !ctx->poses_valid) return fail(ctx, A3_ERR_INVALID, "a3_pack_detections: the last batch was not an a3_detect_batch_pose call");
    if (max_markers_per_frame == 0) return fail(ctx, A3_ERR_INVALID, "a3_pack_detections: max_markers_per_frame must be > 0");
    const size_t need = a3_detection_record_bytes(max_markers_per_frame, with_poses) * ctx->last_n;
    if (dst_bytes < need) return fail(ctx, A3_ERR_CAPACITY, "a3_pack_detections: dst_bytes smaller than frames x record size");
    if (reinterpret_cast<uintptr_t>(dst_device) % 4 != 0) return fail(ctx, A3_ERR_INVALID, "a3_pack_detections: dst must be 4-byte aligned");
    // the host already knows the per-frame counts of that batch: refuse before anything is clipped
    if (ctx->last_max_per_frame > max_markers_per_frame) {
        char msg[160];
        snprintf(msg, sizeof msg, "a3_pack_detections: a frame holds %u markers, the record only %u", ctx->last_max_per_frame, max_markers_per_frame);
        return fail(ctx, A3_ERR_CAPACITY, msg);
    }
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    // scratch word 3 is the kernel's overflow flag (cannot fire after the host check; kept as the device-side guard)
    A3_HIP(launch_pack_detections(ctx->stream, ctx->markers_ptr, with_poses ? ctx->pose_buf.as<a3_pose>() : nullptr, ctx->per_frame, ctx->last_n,
                                  first_frame_global, max_markers_per_frame, dst_device, ctx->scratch_u32 + 3));
    return A3_OK;
}

// ---- find_contours output of the last batch (debug taps; src/aruco.rs:64) ----
namespace {
int fetch_contours(a3_ctx* ctx, uint32_t frame, std::vector<ContourRec>* recs) {
    if (!ctx->contours_valid) return fail(ctx, A3_ERR_INVALID, "contours are only kept for a single-chunk batch run after a3_set_debug_taps(ctx, 1)");
    if (frame >= ctx->frames) return fail(ctx, A3_ERR_INVALID, "frame index outside the last batch");
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    std::vector<ContourRec> all(ctx->tap_contours);
    if (!all.empty()) A3_HIP(hipMemcpy(all.data(), ctx->contours.p, all.size() * sizeof(ContourRec), hipMemcpyDeviceToHost));
    recs->clear();
    for (const ContourRec& r : all) if (r.frame == frame) recs->push_back(r);
    // the reference's discovery order = ascending start key (2 * raster index of the start pixel, +1 for a hole border)
    std::sort(recs->begin(), recs->end(), [](const ContourRec& a, const ContourRec& b) { return a.start_key < b.start_key; });
    return A3_OK;
}
}  // namespace

int a3_contour_count(a3_ctx* ctx, uint32_t frame, uint32_t* n_contours, uint64_t* n_points) {
    if (!ctx) return A3_ERR_INVALID;
    std::vector<ContourRec> recs;
    if (int rc = fetch_contours(ctx, frame, &recs)) return rc;
    uint64_t pts = 0;
    for (const ContourRec& r : recs) pts += r.n;
    if (n_contours) *n_contours = (uint32_t)recs.size();
    if (n_points) *n_points = pts;
    return A3_OK;
}

int a3_download_contours(a3_ctx* ctx, uint32_t frame, uint32_t* start_keys, uint32_t* lengths, uint32_t* points_xy, size_t cap_contours,
                         size_t cap_points) {
    if (!ctx || !start_keys || !lengths || !points_xy) return A3_ERR_INVALID;
    std::vector<ContourRec> recs;
    if (int rc = fetch_contours(ctx, frame, &recs)) return rc;
    uint64_t pts = 0;
    for (const ContourRec& r : recs) pts += r.n;
    if (recs.size() > cap_contours || pts > cap_points) return fail(ctx, A3_ERR_CAPACITY, "a3_download_contours: caps too small");
    std::vector<uint32_t> pool(ctx->tap_points);
    if (!pool.empty()) A3_HIP(hipMemcpy(pool.data(), ctx->points.p, pool.size() * 4, hipMemcpyDeviceToHost));
    size_t o = 0;
    for (size_t i = 0; i < recs.size(); i++) {
        start_keys[i] = recs[i].start_key;
        lengths[i] = recs[i].n;
        for (uint32_t k = 0; k < recs[i].n; k++) {
            const uint32_t p = pool[(size_t)recs[i].point_base + k];
            points_xy[2 * o] = p & 0xFFFFu; points_xy[2 * o + 1] = p >> 16;
            o++;
        }
    }
    return A3_OK;
}

// ---- internal (a3_internal.h): the small helpers of src/aruco.rs run on their own, for the reference's vectors ----
int a3_debug_clockwise(a3_ctx* ctx, const int32_t* quads_xy, size_t n, int32_t* out_xy) {
    if (!ctx || !quads_xy || !out_xy) return A3_ERR_INVALID;
    if (n == 0) return A3_OK;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(ctx->tmp_a.ensure(n * 32)); A3_HIP(ctx->tmp_c.ensure(n * 32));
    A3_HIP(hipMemcpyAsync(ctx->tmp_a.p, quads_xy, n * 32, hipMemcpyHostToDevice, ctx->stream));
    A3_HIP(launch_debug_clockwise(ctx->stream, ctx->tmp_a.as<int32_t>(), (uint32_t)n, ctx->tmp_c.as<int32_t>()));
    A3_HIP(hipMemcpyAsync(out_xy, ctx->tmp_c.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    return A3_OK;
}

int a3_debug_rotate_bits(a3_ctx* ctx, const uint8_t* bits, uint32_t n, uint32_t times, uint8_t* out) {
    if (!ctx || !bits || !out || n == 0 || n > 16) return A3_ERR_INVALID;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(ctx->tmp_a.ensure(256)); A3_HIP(ctx->tmp_c.ensure(256));
    A3_HIP(hipMemcpyAsync(ctx->tmp_a.p, bits, (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
    A3_HIP(launch_debug_rotate_bits(ctx->stream, ctx->tmp_a.as<uint8_t>(), n, times & 3u, ctx->tmp_c.as<uint8_t>()));
    A3_HIP(hipMemcpyAsync(out, ctx->tmp_c.p, (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    return A3_OK;
}

// quads in the order given (= the reference's candidate order) through k_frame_candidates' discard_too_near
int a3_debug_discard_too_near(a3_ctx* ctx, const uint32_t* quads_xy, size_t n, float min_distance, uint32_t* out_xy, size_t* n_out) {
    if (!ctx || !quads_xy || !out_xy || !n_out) return A3_ERR_INVALID;
    *n_out = 0;
    if (n == 0) return A3_OK;
    constexpr uint32_t kMaxCand = kMaxCandDefault;
    if (n > kMaxCand) return fail(ctx, A3_ERR_CAPACITY, "a3_debug_discard_too_near: at most 1024 quads");
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    std::vector<CandRec> h(n);
    for (size_t i = 0; i < n; i++) {
        h[i].start_key = (uint32_t)i;   // the given order
        for (int k = 0; k < 8; k++) {
            if (quads_xy[8 * i + k] > 65535u) return fail(ctx, A3_ERR_INVALID, "coordinates above 65535");
            h[i].xy[k] = (uint16_t)quads_xy[8 * i + k];
        }
    }
    A3_HIP(ctx->tmp_a.ensure(kMaxCand * sizeof(CandRec)));
    A3_HIP(ctx->tmp_b.ensure(kMaxCand * 16 * 2));          // pre_xy | fin_xy
    A3_HIP(ctx->tmp_c.ensure(kMaxCand * 4 + 64));          // work list | cand_count, fin_count, work_count
    uint32_t* small = ctx->tmp_c.as<uint32_t>() + kMaxCand;
    const uint32_t init[3] = {(uint32_t)n, 0u, 0u};
    A3_HIP(hipMemcpyAsync(ctx->tmp_a.p, h.data(), n * sizeof(CandRec), hipMemcpyHostToDevice, ctx->stream));
    A3_HIP(hipMemcpyAsync(small, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    uint16_t* pre = ctx->tmp_b.as<uint16_t>(); uint16_t* fin = pre + kMaxCand * 8;
    A3_HIP(launch_frame_candidates(ctx->stream, ctx->tmp_a.as<CandRec>(), small, 1, kMaxCand, min_distance, pre, fin, small + 1,
                                   ctx->tmp_c.as<uint32_t>(), small + 2, 0u, nullptr, nullptr));
    uint32_t cnt = 0;
    A3_HIP(hipMemcpyAsync(&cnt, small + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint16_t> o((size_t)cnt * 8);
    if (cnt) A3_HIP(hipMemcpy(o.data(), fin, o.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < o.size(); i++) out_xy[i] = o[i];
    *n_out = cnt;
    return A3_OK;
}

int a3_debug_inject_candidates(a3_ctx* ctx, const uint32_t* quads_xy, size_t n) {
    if (!ctx || (!quads_xy && n)) return A3_ERR_INVALID;
    if (ctx->pending.active || ctx->pending_trivial) return fail(ctx, A3_ERR_INVALID, "a3_debug_inject_candidates: a submitted batch has not been collected");
    if (n > kMaxCandDefault) return fail(ctx, A3_ERR_CAPACITY, "a3_debug_inject_candidates: at most 1024 quads");
    ctx->inject.resize(n);
    for (size_t i = 0; i < n; i++) {
        ctx->inject[i].start_key = (uint32_t)i;   // the given order
        for (int k = 0; k < 8; k++) {
            if (quads_xy[8 * i + k] > 65535u) return fail(ctx, A3_ERR_INVALID, "coordinates above 65535");
            ctx->inject[i].xy[k] = (uint16_t)quads_xy[8 * i + k];
        }
    }
    ctx->inject_armed = true;
    return A3_OK;
}

static int pose_common(a3_ctx* ctx, const uint32_t* corners, const float* norm, size_t n, int mode, float size_mm, const a3_intrinsics* intr,
                       uint32_t iw, uint32_t ih, a3_pose* out) {
    if (!ctx || !out || (!corners && !norm)) return A3_ERR_INVALID;
    if (n == 0) return A3_OK;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    const size_t in_bytes = n * 8 * 4;
    A3_HIP(ctx->tmp_a.ensure(in_bytes));
    A3_HIP(ctx->tmp_b.ensure(n * 2 * sizeof(a3_pose)));
    A3_HIP(hipMemcpyAsync(ctx->tmp_a.p, corners ? (const void*)corners : (const void*)norm, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    float fx = 1, fy = 1, cx = 0, cy = 0;
    if (intr) { fx = intr->focal_x; fy = intr->focal_y; cx = intr->principal_x; cy = intr->principal_y; }
    A3_HIP(launch_pose(ctx->stream, corners ? ctx->tmp_a.as<uint32_t>() : nullptr, 8u, norm ? ctx->tmp_a.as<float>() : nullptr, (uint32_t)n, nullptr, mode,
                       size_mm, (float)iw, (float)ih, fx, fy, cx, cy, ctx->tmp_b.as<a3_pose>()));
    A3_HIP(hipMemcpyAsync(out, ctx->tmp_b.p, n * 2 * sizeof(a3_pose), hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    return A3_OK;
}

int a3_estimate_pose(a3_ctx* ctx, const uint32_t* corners_xy, size_t n, float marker_size_mm, const a3_intrinsics* intr, uint32_t image_width,
                     uint32_t image_height, a3_pose* out) {
    if (!corners_xy) return A3_ERR_INVALID;
    return pose_common(ctx, corners_xy, nullptr, n, intr ? 1 : 0, marker_size_mm, intr, image_width, image_height, out);
}

int a3_estimate_pose_normalized(a3_ctx* ctx, const float* points_xy, size_t n, float marker_size_mm, a3_pose* out) {
    if (!points_xy) return A3_ERR_INVALID;
    return pose_common(ctx, nullptr, points_xy, n, 2, marker_size_mm, nullptr, 1, 1, out);
}

int a3_find_nearest(a3_ctx* ctx, const uint64_t* bits, size_t n, uint32_t* idx, uint8_t* dist) {
    if (!ctx || !bits || !idx || !dist) return A3_ERR_INVALID;
    if (n == 0) return A3_OK;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(ctx->tmp_a.ensure(n * 8));
    A3_HIP(ctx->tmp_c.ensure(n * 4));
    A3_HIP(ctx->tmp_d.ensure(n));
    A3_HIP(hipMemcpyAsync(ctx->tmp_a.p, bits, n * 8, hipMemcpyHostToDevice, ctx->stream));
    A3_HIP(launch_find_nearest(ctx->stream, ctx->dict.as<uint64_t>(), ctx->n_codes, ctx->tmp_a.as<uint64_t>(), (uint32_t)n, ctx->tmp_c.as<uint32_t>(),
                               ctx->tmp_d.as<uint8_t>()));
    A3_HIP(hipMemcpyAsync(idx, ctx->tmp_c.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipMemcpyAsync(dist, ctx->tmp_d.p, n, hipMemcpyDeviceToHost, ctx->stream));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    return A3_OK;
}

int a3_set_profiling(a3_ctx* ctx, int enabled) {
    if (!ctx) return A3_ERR_INVALID;
    ctx->profiling = enabled == A3_PROFILE_STAGES ? 2 : ((enabled == A3_PROFILE_THRESHOLD_ONLY || enabled == A3_PROFILE_THRESHOLD_SAMPLED) ? 1 : 0);
    ctx->profile_every = enabled == A3_PROFILE_THRESHOLD_SAMPLED ? 4 : 1;
    return A3_OK;
}

int a3_get_profile(a3_ctx* ctx, int stage, double* total_ms, uint64_t* launches, int reset) {
    if (!ctx || stage < 0 || stage >= A3_STAGE_COUNT) return A3_ERR_INVALID;
    if (total_ms) *total_ms = ctx->prof_ms[stage];
    if (launches) *launches = ctx->prof_n[stage];
    if (reset) { ctx->prof_ms[stage] = 0; ctx->prof_n[stage] = 0; }
    return A3_OK;
}

int a3_selftest_ieee(a3_ctx* ctx, const double* a, const double* b, size_t n, double* sqrt_a, double* a_div_b, float* sqrtf_a, float* a_divf_b) {
    if (!ctx || !a || !b || !sqrt_a || !a_div_b || !sqrtf_a || !a_divf_b) return A3_ERR_INVALID;
    if (n == 0) return A3_OK;
    A3_HIP(hipSetDevice(ctx->device));
    if (int rcs_ = need_stream(ctx)) return rcs_;
    A3_HIP(ctx->tmp_a.ensure(n * 8)); A3_HIP(ctx->tmp_b.ensure(n * 8)); A3_HIP(ctx->tmp_c.ensure(n * 16)); A3_HIP(ctx->tmp_d.ensure(n * 8));
    A3_HIP(hipMemcpy(ctx->tmp_a.p, a, n * 8, hipMemcpyHostToDevice));
    A3_HIP(hipMemcpy(ctx->tmp_b.p, b, n * 8, hipMemcpyHostToDevice));
    double* sq = ctx->tmp_c.as<double>(); double* dv = sq + n;
    float* sqf = ctx->tmp_d.as<float>(); float* dvf = sqf + n;
    A3_HIP(launch_selftest(ctx->stream, ctx->tmp_a.as<double>(), ctx->tmp_b.as<double>(), (uint32_t)n, sq, dv, sqf, dvf));
    A3_HIP(hipStreamSynchronize(ctx->stream));
    A3_HIP(hipMemcpy(sqrt_a, sq, n * 8, hipMemcpyDeviceToHost));
    A3_HIP(hipMemcpy(a_div_b, dv, n * 8, hipMemcpyDeviceToHost));
    A3_HIP(hipMemcpy(sqrtf_a, sqf, n * 4, hipMemcpyDeviceToHost));
    A3_HIP(hipMemcpy(a_divf_b, dvf, n * 4, hipMemcpyDeviceToHost));
    return A3_OK;
}

}  // extern "C"
