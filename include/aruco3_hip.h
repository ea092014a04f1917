/*
 * aruco3_hip.h -- C ABI of libaruco3_hip.so: the MI355X (gfx950) implementation of the
 * aruco3 detection hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference crate has no FFI seam today: this ABI is placed
 * INSIDE `Detector::detect` (src/aruco.rs:52-121) and the pose solvers
 * (src/pose.rs:52-81) so that their public Rust signatures stay unchanged; the Rust-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: one a3_ctx per (device, stream); a context is not re-entrant.
 * Errors: every entry point returns 0 on success and a negative A3_ERR_* otherwise;
 * a3_last_error() gives the message.  Nothing unwinds across the boundary.
 */
#ifndef ARUCO3_HIP_H
#define ARUCO3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3_ABI_VERSION 5

enum {
    A3_OK = 0,
    A3_ERR_INVALID = -1,    /* bad argument (null pointer, zero size, threshold_window == 0 ...) */
    A3_ERR_HIP = -2,        /* a HIP runtime call failed */
    A3_ERR_CAPACITY = -3,   /* caller's output array (or a fixed device pool) is too small */
    A3_ERR_INTERNAL = -4,   /* an invariant of the contour stage did not hold; results withheld */
    A3_ERR_NO_DEVICE = -5,
    A3_ERR_LIMIT = -6       /* a fixed limit of this implementation, which no larger buffer of the caller's cures: more than
                             * A3_MAX_CANDIDATES_PER_FRAME quad candidates in one frame (the reference's lists are unbounded Vecs,
                             * src/aruco.rs:128; here a3_marker.candidate_index is 16 bits), frames x candidate slots per frame
                             * beyond 2^32 in one call, or a frame whose contour graph needs more than 2^32 nodes.  Growing `out`
                             * and calling again -- the cure for A3_ERR_CAPACITY -- does not help. */
};
/* The per-frame candidate tables start at 1024 slots and grow with a re-run of the batch (2048, 4096, 6144: ordered and thinned --
 * discard_too_near -- in one workgroup's LDS; 12 288 ... 65 536: the same walk through memory, correct but slow -- 0.1 s for a frame
 * of 7 000 quads -- which only a frame tiled with thousands of small squares reaches). */
#define A3_MAX_CANDIDATES_PER_FRAME 65536

/* pixel layouts accepted where the reference takes an image::DynamicImage (src/aruco.rs:52,60) */
enum { A3_FMT_RGB8 = 0, A3_FMT_RGBA8 = 1, A3_FMT_L8 = 2,
       A3_FMT_BGRA8 = 3 /* webcam byte order (examples/webcam_kamera.rs:38-52 swizzles it on the CPU first) */ };
/* where `pixels` lives */
enum { A3_MEM_HOST = 0, A3_MEM_DEVICE = 1 };

/* DetectorConfig, src/aruco.rs:23-43 (same fields, same defaults via a3_default_config) */
typedef struct a3_config {
    uint32_t threshold_window;               /* 7.  The block radius of adaptive_threshold (src/aruco.rs:61): the window is 2r+1 wide.  Every value
                                                * >= 1 gives the reference's result; 1..31 run fused one-pass kernels (1..7 at about the speed of 7,
                                                * 8..26 at 1.1-1.35 x, 27..31 at 2.3-2.5 x), larger windows a separable three-pass path (8-9 x) */
    double   contour_simplification_epsilon; /* 0.05 */
    float    min_side_length_factor;         /* 0.2 */
    float    min_corner_separation_factor;   /* 0.1 */
    uint32_t homography_sample_size;         /* 49 */
    uint8_t  filter_high_bit_errors;         /* 1 */
} a3_config;

/* Marker, src/aruco.rs:8-13, plus the frame it belongs to in a batch */
typedef struct a3_marker {
    uint32_t frame;
    uint32_t id;               /* index into the dictionary */
    uint64_t code;             /* the code as read (uncorrected), src/aruco.rs:10 */
    uint32_t corners[8];       /* x0,y0 .. x3,y3 after corners.rotate_left(rotation) */
    uint8_t  hamming_distance;
    uint8_t  rotation;         /* 0..3, src/aruco.rs:89,103 */
    uint16_t candidate_index;  /* position in Detection.candidates of that frame */
} a3_marker;

/* MarkerPose, src/pose.rs:8-12; rotation row-major as Matrix3::new(...) is written */
typedef struct a3_pose {
    float error;
    float rotation[9];
    float translation[3];
} a3_pose;

/* CameraIntrinsics, src/pinhole.rs:11-18 */
typedef struct a3_intrinsics {
    uint32_t image_width, image_height;
    float focal_x, focal_y, principal_x, principal_y;
} a3_intrinsics;

/* per-batch stage counters (the reference prints the rejects in debug builds, src/aruco.rs:163-164) */
typedef struct a3_stats {
    uint64_t darts;                 /* contour-graph nodes built */
    uint64_t contours_traced;       /* borders the reference would have followed (excluding 1-pixel specks) */
    uint64_t contours_materialised; /* those that survived the parity-safe size pruning */
    uint64_t candidates_pre;        /* after contours_to_candidates */
    uint64_t candidates;            /* after discard_too_near */
    uint64_t markers;
    uint32_t resolve_iterations;    /* start-resolution passes over all darts (0 = none needed) */
    uint32_t jump_rounds;           /* pointer-doubling rounds that did work */
    uint32_t chunks;                /* sub-batches the frames were split into */
    uint32_t stepping;              /* bits 0-7: A3_STEP_* -- how the library scheduled the batch (see a3_order_after);
                                     * bits 8-15: chains of OTHER contexts this batch's submit released (A3_STEP_BURST_LAST);
                                     * bits 16-23: synchronous re-runs of the batch the device asked for (a pool or table that
                                     * had to grow, more passes, a host-side plan) -- each costs a whole batch */
} a3_stats;
/* a3_stats.stepping & 0xFF */
enum {
    A3_STEP_WHOLE = 0,                  /* the whole batch was enqueued by its own call (a3_detect_batch always; a submit outside a burst) */
    A3_STEP_DECODE_DEFERRED = 1,        /* contexts sharing one stream: decode stage released from inside the next batch's launch sequence */
    A3_STEP_HELD_RELEASED_BY_LAST = 2,  /* burst member: chain held at submit, enqueued by the submit of the burst's last member */
    A3_STEP_HELD_RELEASED_EARLY = 3,    /* burst member whose chain went out before a last member came: at collect, at a gate on it,
                                         * at a3_set_stream */
    A3_STEP_BURST_LAST = 4,             /* the burst's last member: released the others' chains behind its threshold kernel */
    A3_STEP_HELD = 5                    /* only ever seen BETWEEN submit and collect of a burst member: its chain is still held; the
                                         * collected batch reports 2 or 3 */
};

typedef struct a3_ctx a3_ctx;

int  a3_abi_version(void);
void a3_default_config(a3_config *cfg);

/* Detector { config, dictionary } (src/aruco.rs:46-49).  `codes` is ARDictionary.code_list,
 * tau == 0 means "compute the minimum pairwise distance" (src/dictionaries.rs:124). */
int  a3_create(int device, const a3_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
               a3_ctx **out);
void a3_destroy(a3_ctx *ctx);
const char *a3_last_error(const a3_ctx *ctx); /* ctx may be NULL: message of the last failed a3_create */
/* run on the caller's HIP stream (hipStream_t as void*); NULL = the context's own stream */
int  a3_set_stream(a3_ctx *ctx, void *hip_stream);
/* the stream the context enqueues on (its own unless a3_set_stream changed it): a caller that follows a batch or
 * a3_pack_detections with work of its own orders that work after this stream */
int  a3_get_stream(const a3_ctx *ctx, void **hip_stream);
/* total darts / contour points the device pools may hold (0 keeps the default); call before detect */
int  a3_set_pool_limits(a3_ctx *ctx, uint64_t max_darts, uint64_t max_points);
int  a3_get_tau(const a3_ctx *ctx, uint8_t *tau);
/* debug taps: keep the grey plane (Detection.grey -- not even computed otherwise, the decode stage samples the frames
 * themselves) and the warped 49x49 patches (Detection.homographies) of the next batches for a3_download_* */
int  a3_set_debug_taps(a3_ctx *ctx, int enabled);

/* Detector::detect over a batch of independent frames (src/aruco.rs:52-121).
 * pixels: n_frames images, `frame_stride` bytes apart, rows `row_stride` bytes apart.
 * out: markers of frame 0 first, in the reference's order; per_frame_count[n_frames] optional.
 * Limits: width, height <= 65535, width * height < 2^30, n_frames <= 65535 per call. */
int  a3_detect_batch(a3_ctx *ctx, const void *pixels, int memory, int fmt, uint32_t width, uint32_t height,
                     size_t row_stride, size_t frame_stride, uint32_t n_frames,
                     a3_marker *out, size_t out_cap, uint32_t *per_frame_count, size_t *out_n);
/* detect + pose in one call (BASELINE config 5; callers always solve the pose right after detect,
 * examples/webcam_kamera.rs:67-71): poses[2*i], poses[2*i+1] belong to out[i], lower error first.
 * intr == NULL: solve_with_undistorted_points with the frame size, else solve_with_intrinsics. */
int  a3_detect_batch_pose(a3_ctx *ctx, const void *pixels, int memory, int fmt, uint32_t width, uint32_t height,
                          size_t row_stride, size_t frame_stride, uint32_t n_frames, float marker_size_mm,
                          const a3_intrinsics *intr, a3_marker *out, a3_pose *poses, size_t out_cap,
                          uint32_t *per_frame_count, size_t *out_n);
/* The same call in two halves, for callers that keep the GPU fed: submit enqueues the batch and returns without
 * waiting; collect waits for it and hands out the results (re-running the batch synchronously in the rare cases
 * a3_detect_batch would).  One batch may be in flight per context; several contexts keep several batches in flight (see
 * a3_order_after).  out_cap of submit bounds the marker list; collect's must not be smaller than what was found.
 *
 * Frames: device-resident frames and pinned host frames must stay valid AND UNMODIFIED from submit until collect returns --
 * also by work the caller queues on its own stream behind the submit.  Part of a submitted batch may run on streams the library
 * owns (the device-wide decode and copy streams) and a chain held back for a burst is enqueued later (a3_order_after), ordered
 * against the caller's stream only through events recorded at submit; and the decode stage samples the FRAMES themselves (no grey plane is kept), so
 * a frame overwritten before collect changes what is read.  Pageable host frames have been read when submit returns.
 *
 * Hardware queues: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), handed
 * out in the order the streams are first used, and two streams that share a queue run in order whatever their events say -- a
 * stream that merely waits (a collective waiting for its peers) holds up the others of its queue.  A process that keeps four
 * contexts on their own streams plus the library's decode / copy streams in flight should export GPU_MAX_HW_QUEUES before the
 * runtime starts: 8 when nothing else of the process uses streams; 16 when it also runs collectives or copies on streams of its
 * own (bench.py sets 16: with 8 its collective's side stream landed on a context's queue and every collective stalled that
 * context).  With fewer queues results are the same and the batches merely overlap less. */
int  a3_detect_batch_submit(a3_ctx *ctx, const void *pixels, int memory, int fmt, uint32_t width, uint32_t height,
                            size_t row_stride, size_t frame_stride, uint32_t n_frames, size_t out_cap);
int  a3_detect_batch_collect(a3_ctx *ctx, a3_marker *out, size_t out_cap, uint32_t *per_frame_count, size_t *out_n);
/* a3_detect_batch_pose in the same two halves (BASELINE config 5 pipelined like config 2) */
int  a3_detect_batch_pose_submit(a3_ctx *ctx, const void *pixels, int memory, int fmt, uint32_t width, uint32_t height,
                                 size_t row_stride, size_t frame_stride, uint32_t n_frames, float marker_size_mm,
                                 const a3_intrinsics *intr, size_t out_cap);
int  a3_detect_batch_pose_collect(a3_ctx *ctx, a3_marker *out, a3_pose *poses, size_t out_cap, uint32_t *per_frame_count,
                                  size_t *out_n);

/* Several contexts in flight on one device.  A context per batch in flight, each on a stream of its own, used in rotation (batch j
 * on context j % N, batch j + N submitted as soon as batch j is collected), lets the contour / decode chains of consecutive batches
 * overlap one another (they are latency-, LDS- and issue-bound and leave most of the chip idle); the threshold kernel does not take
 * part in that -- one launch occupies every register of the chip.  Nothing more is needed: with N = 3 or 4 and a batch of its own
 * per context this FREE-RUNNING rotation is the fastest arrangement measured.
 *
 * BURSTS are the same rotation with phases: context k calls
 *     a3_order_after(ctx[k], ctx[m])   for every m in k+1 .. N-1
 * before each submit; the threshold kernels of one rotation then run back to back on an otherwise quiet GPU, and its chains
 * together -- within 2 % of the free-running rotation, and the arrangement to choose when something else (a collective, a copy)
 * should find the GPU in a known phase.  What the library does with these calls, always (there is no mode to select; since ABI 5):
 *   - the call itself: the next batch submitted on `ctx` starts on the device only after everything enqueued so far on `other`
 *     (its batch in flight included) has finished -- the batches of one rotation start once the previous rotation has drained;
 *   - a submit on a context that declared such gates since its previous submit enqueues the batch's THRESHOLD KERNEL ONLY and
 *     holds the rest of the batch (contour stage ... read-back: its "chain") back;
 *   - the first submit on the device WITHOUT gates is the burst's last member (context N-1 above): it enqueues its threshold
 *     kernel, then the held chains of the other members behind that kernel, then its own chain.  On the GPU a rotation is then N
 *     threshold kernels back to back followed by N chains side by side, with no host in between;
 *   - a held chain nobody released goes out when its batch is collected, when another context declares a gate on it, or when
 *     a3_set_stream moves its context.  a3_stats.stepping of the collected batch says which of these happened.
 * A context that runs on a CALLER's stream of its own (a3_set_stream; no other context on it) is stepped the same way: with gates
 * declared its chain is held and enqueued LATER on that stream -- by whichever thread releases it -- and so lands behind any work
 * the caller queued on the stream after the submit.
 * The hold does not apply -- the whole batch is enqueued at submit, gates still ordering it -- to the first batch of a shape
 * (frame count / size) on a context and to batches whose contour graph must be planned on the host (graphs that outgrow the
 * pools: uniform-noise frames), while a3_set_profiling(A3_PROFILE_STAGES) is on, and to contexts that share one stream: those
 * are in order already, a3_order_after between them is a no-op, and their decode stage is deferred behind the next batch's
 * contour stage instead (A3_STEP_DECODE_DEFERRED).  Scheduling only: results are
 * identical in every arrangement.
 *
 * Threading.  One context is never used from two threads at once.  DIFFERENT contexts may be driven from different threads, with
 * one addition for bursts: the contexts of a device whose chains are held form one scheduling domain -- the submit of the last
 * member (and a3_order_after / a3_detect_batch_collect / a3_set_stream / a3_destroy naming a holder) enqueues the held chains of
 * sibling contexts, under a process-wide lock, from the calling thread.  That touches the siblings' internal state only between
 * their submit and their collect, when the only calls a caller may make on them are collect, a3_set_stream and a3_destroy (all of
 * which take the same lock); nothing is allocated and the device is never waited for under that lock (held batches are planned
 * on the device and their buffers allocated by their own submit).  An error met while enqueueing a sibling's chain is reported
 * by that sibling's collect, not by the call that met it. */
int  a3_order_after(a3_ctx *ctx, a3_ctx *other);

/* Host frames (A3_MEM_HOST) cross the link on the device's copy stream, beside the kernels of whatever batch another
 * context has in flight.  From pageable memory the runtime stages the copy and the call returns once the caller's buffer has
 * been read.  From PINNED memory -- allocated with a3_host_alloc, or the caller's own ring pinned once with a3_host_register
 * (a webcam loop's frame buffers, examples/webcam_kamera.rs:36-71) -- the copy is asynchronous and runs at the link's rate:
 * a3_detect_batch_submit returns at once, and the buffer must then stay untouched until the batch is collected. */
int  a3_host_alloc(size_t bytes, void **out);
int  a3_host_free(void *p);
int  a3_host_register(void *p, size_t bytes);
int  a3_host_unregister(void *p);
int  a3_get_stats(const a3_ctx *ctx, a3_stats *stats);

/* Detection.grey / .candidates / .homographies of the last batch (src/aruco.rs:16-21,115-120),
 * plus the thresholded image; host destinations. */
int  a3_download_grey(a3_ctx *ctx, uint32_t frame, uint8_t *dst);
int  a3_download_thresholded(a3_ctx *ctx, uint32_t frame, uint8_t *dst);
int  a3_candidate_count(a3_ctx *ctx, uint32_t frame, uint32_t *n_pre, uint32_t *n_final);
int  a3_download_candidates(a3_ctx *ctx, uint32_t frame, int before_discard, uint32_t *dst_xy, size_t cap_quads);
int  a3_download_homographies(a3_ctx *ctx, uint32_t frame, uint8_t *dst, uint8_t *ok, uint64_t *codes4, int32_t *decode_ok,
                              size_t cap);

/* The result of find_contours (src/aruco.rs:64) for one frame of the last batch: available when that batch ran with debug
 * taps on and fitted one chunk.  With taps on, every border the reference follows is materialised (the parity-safe size
 * pruning is off) except 1-pixel components, which own no border pixel pair.  Contours come in the reference's discovery
 * order; start_keys[i] = 2 * (y * width + x) of the first point, + 1 for a hole border; points_xy holds x, y pairs. */
int  a3_contour_count(a3_ctx *ctx, uint32_t frame, uint32_t *n_contours, uint64_t *n_points);
int  a3_download_contours(a3_ctx *ctx, uint32_t frame, uint32_t *start_keys, uint32_t *lengths, uint32_t *points_xy,
                          size_t cap_contours, size_t cap_points);

/* Multi-GPU gather (frames are sharded by rank): the markers of the last finished batch as
 * fixed-capacity records in DEVICE memory, one per frame, ready for one all-gather:
 *   u32 count | u32 global frame index (first_frame_global + f) | max_markers_per_frame x a3_marker
 *   | with_poses: max_markers_per_frame x 2 x a3_pose (the poses of marker k at 2k, 2k+1; the last batch must then have
 *     been an a3_detect_batch_pose call -- BASELINE config 5 gathers poses with the detections)
 * with a3_marker.frame rewritten to the global index and unused slots zeroed.  Written by a kernel on the context's stream
 * (no host copy); the caller orders its collective after it.  A frame with more markers than the record holds is an
 * error (A3_ERR_CAPACITY), never a silent clip. */
size_t a3_detection_record_bytes(uint32_t max_markers_per_frame, int with_poses);
int  a3_pack_detections(a3_ctx *ctx, uint32_t first_frame_global, uint32_t max_markers_per_frame, int with_poses,
                        void *dst_device, size_t dst_bytes);

/* pose::solve_with_undistorted_points (intr == NULL, src/pose.rs:59-62) or
 * pose::solve_with_intrinsics (src/pose.rs:52-55) for n markers; out holds 2*n poses,
 * the lower-error one first (src/pose.rs:76-80). */
int  a3_estimate_pose(a3_ctx *ctx, const uint32_t *corners_xy, size_t n, float marker_size_mm, const a3_intrinsics *intr,
                      uint32_t image_width, uint32_t image_height, a3_pose *out);
/* pose::solve_with_normalized_points (src/pose.rs:64-81) */
int  a3_estimate_pose_normalized(a3_ctx *ctx, const float *points_xy, size_t n, float marker_size_mm, a3_pose *out);

/* ARDictionary::find_nearest for n codes (src/dictionaries.rs:160-196) and calculate_tau (:129-138) */
int  a3_find_nearest(a3_ctx *ctx, const uint64_t *bits, size_t n, uint32_t *idx, uint8_t *dist);
int  a3_calculate_tau(int device, const uint64_t *codes, size_t n_codes, uint8_t *tau);

/* measurement: device time spent in a stage, summed over the launches since the last reset */
enum { A3_STAGE_THRESHOLD = 0, A3_STAGE_CONTOUR = 1, A3_STAGE_DECODE = 2, A3_STAGE_COUNT = 3 };
/* mode: stage times are taken with HIP events on the context's stream; an event record between two kernels costs about
 * 6 us of device time, so the lighter mode times the threshold stage only (two records per batch instead of four) */
enum { A3_PROFILE_OFF = 0, A3_PROFILE_STAGES = 1, A3_PROFILE_THRESHOLD_ONLY = 2,
       A3_PROFILE_THRESHOLD_SAMPLED = 3 /* the threshold stage of every 4th batch: a quarter of the records' cost */ };
int  a3_set_profiling(a3_ctx *ctx, int mode);
int  a3_get_profile(a3_ctx *ctx, int stage, double *total_ms, uint64_t *launches, int reset);

/* Synthetic frames rendered on the device (the reference's counterparts are its test renderer
 * and ARDictionary::make_binary_image, src/dictionaries.rs:209-232).  The caller lays the frames out -- background
 * gradient, and per marker the inverse homography image -> cell coordinates, a bounding box and the n x n cell bitmap
 * (bit r*n+c = 1: white; n = code side + 2 <= 11: CHILITAGS is 10 x 10) -- and the kernel paints RGB8 frames into device memory `out`. */
typedef struct a3_synth_marker {
    float    hinv[9];            /* image (x, y, 1) -> marker cells (u, v, w), row-major */
    int32_t  x0, y0, x1, y1;     /* pixels the marker (with its one-cell quiet zone) can touch: [x0,x1) x [y0,y1) */
    uint64_t cells[2];           /* bit r*n+c of the 128 = cell (r, c); 1: white */
    uint32_t n, reserved;
} a3_synth_marker;
typedef struct a3_synth_frame {
    float    base, gx, gy;       /* background = base + gx * xs + gy * ys, xs, ys in [-1, 1] */
    float    noise_sigma;        /* additive Gaussian noise per channel, 0 = none */
    uint32_t first_marker, n_markers;
    uint64_t seed;               /* of the noise */
} a3_synth_frame;
int  a3_synth_render(int device, void *hip_stream, const a3_synth_frame *frames, uint32_t n_frames, const a3_synth_marker *markers,
                     uint32_t n_markers, uint32_t width, uint32_t height, int paper, float black, float white, int supersample,
                     void *out_rgb_device, size_t row_stride, size_t frame_stride);

#ifdef __cplusplus
}
#endif
#endif
