/*
 * a3_internal.h -- entry points of libaruco3_hip.so that are NOT part of the binding surface (include/aruco3_hip.h):
 * tuning probes and single-stage hooks used by this repository's tests and tools only.  A Rust/C binding of the detector
 * has no business calling them; they may change without an ABI version bump.
 */
#ifndef ARUCO3_HIP_INTERNAL_H
#define ARUCO3_HIP_INTERNAL_H

#include "../../include/aruco3_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel-level timing for tuning (tools/kernel_probe.py): re-runs one kernel (0 dart_count, 1 dart_assign, 2 local_contract,
 * 3 decode, 4 decode with the frames evicted from the caches before every run: bench.py's roofline_warp) on the buffers of the last single-chunk batch, optionally truncated (dbg), and returns the average device time.
 * The internal contour buffers hold garbage afterwards; results already returned are unaffected. */
int  a3_debug_kernel_time(a3_ctx *ctx, int kernel, int dbg, int reps, float *avg_ms);

/* Where the decode stage of a SUBMITTED batch is released when another context submits behind it (see a3_api.hip,
 * "deferred decode"): 0 = never deferred (both halves of a batch enqueued at once, as a3_detect_batch always does), 1 = behind
 * the next batch's threshold kernel, 2 = behind its k_local_contract (the default); | 0x100: contexts created from now on get a
 * decode stream of the lowest priority instead of the default one.  Process-wide; results are identical in every mode -- tools/ use it for A/B timing inside one process, since two boxes of the pool differ by more than the effect. */
int  a3_debug_set_overlap(int mode);

/* K1 waves per SIMD the threshold kernel's strip model sizes its launches for (2: the whole chip in one round; 1: one wave per
 * SIMD, twice as tall strips).  Process-wide; results are identical. */
int  a3_debug_set_k1_waves(int waves_per_simd);

/* A stand-in for a collective's channel kernels, for a box with one GPU: `workgroups` workgroups of `threads` threads stay
 * resident on `hip_stream` for `usec` microseconds (48 live registers per lane, a read and a short sleep per turn).
 * tools/spin_probe.py measures what such company does to the threshold kernel and to a step. */
int  a3_debug_spin(void *hip_stream, int workgroups, int threads, int usec);

/* Release point of a caller's side work (a collective's kernels), measured and NOT adopted (tools/spin_probe.py,
 * profiles/r04_spin_probe.txt): with a3_debug_set_mark_threshold(1) an event is recorded behind the threshold kernel of every
 * batch -- which alone costs ~2 % of a step -- and a3_debug_stream_wait_threshold makes work enqueued on `hip_stream` afterwards
 * wait for the threshold kernel of ctx's batch in flight.  Company released there costs a step as much as company released at once. */
int  a3_debug_set_mark_threshold(int on);
/* priority probe (measured, not adopted: profiles/r05_k1_priority.txt): 1 / 2 = the threshold kernel of every batch on one device-wide
 * stream of the lowest / highest priority, ordered against the context's stream by two events; 0 = on the context's stream (the
 * product).  Call before any context is used. */
int  a3_debug_set_k1_stream(int mode);
int  a3_debug_stream_wait_threshold(a3_ctx *ctx, void *hip_stream);

/* 0: contexts that declared burst gates (a3_order_after) enqueue their whole batch at submit, as round 3's library did; 1 (default):
 * they hold the chain behind their threshold kernel until the burst's last member has enqueued its own (a3_api.hip, submit_common).
 * Process-wide, for A/B timing; results are identical. */
int  a3_debug_set_hold(int on);

/* the threshold kernel alone, asynchronously, on the context's stream (tools/k1_concurrency.py: how several launches in flight
 * at once share the chip) */
int  a3_debug_launch_threshold(a3_ctx *ctx, const void *pixels_device, int fmt, uint32_t width, uint32_t height, uint32_t n_frames);

/* which build this is: bit 0 = -DA3_TUNING (the library reads tuning knobs from the environment), bit 1 = a non-default kernel
 * build option (A3_T_LPX, A3_T_WAVES ...).  0 for the product library; bench.py and the GPU tests report it. */
int  a3_debug_build_flags(void);

/* CU partition (measurement aid): the threshold kernel of every batch on a device-wide stream restricted to k1_cus compute units
 * (hipExtStreamCreateWithCUMask), every other stream the library creates restricted to the remaining ones.  pattern 0: the first
 * k1_cus units as the runtime numbers them, 1: the same share of every group of 16.  0 = off.  Call before any context is used. */
int  a3_debug_set_partition(int k1_cus, int pattern);

/* numerics self-check used by the GPU tests: evaluates the IEEE operations the kernels rely on (f64 sqrt/div, f32 sqrt/div)
 * for n inputs so that the host can compare them bit for bit */
int  a3_selftest_ieee(a3_ctx *ctx, const double *a, const double *b, size_t n, double *sqrt_a, double *a_div_b,
                      float *sqrtf_a, float *a_divf_b);

/* The reference's own vectors for its small helpers (src/aruco.rs:400-459), fed through the device code that implements
 * them inside the pipeline kernels:
 *   enforce_clockwise_corners (src/aruco.rs:168-185)  -> the winding fix of k_contour_quads, n quads of 4 (x, y) i32 pairs
 *   rotate_bit_matrix         (src/aruco.rs:315-326)  -> k_decode's rotation mapping applied `times` times to an n x n matrix
 *   discard_too_near          (src/aruco.rs:187-232)  -> k_frame_candidates on one frame whose candidates are `quads_xy` in order */
int  a3_debug_clockwise(a3_ctx *ctx, const int32_t *quads_xy, size_t n, int32_t *out_xy);
int  a3_debug_rotate_bits(a3_ctx *ctx, const uint8_t *bits, uint32_t n, uint32_t times, uint8_t *out);
/* Quirk Q4 (src/aruco.rs:255-257: a failed projection -> 1 x 1 black patch -> code 0 looked up): the quads given here REPLACE the
 * candidate list of frame 0 of the NEXT batch of this context (a synchronous a3_detect_batch of one frame), in the given order,
 * between the contour stage and k_frame_candidates -- the place of the list enforce_clockwise_corners leaves behind
 * (src/aruco.rs:68).  A convex hull never yields the collinear / repeated-corner quad that makes the 8 x 8 solve fail; this is
 * the only way to lead one through discard_too_near, the solve, k_decode's 1 x 1 stand-in and the accept test on the device.
 * One shot: the batch after that runs unchanged. */
int  a3_debug_inject_candidates(a3_ctx *ctx, const uint32_t *quads_xy, size_t n);
int  a3_debug_discard_too_near(a3_ctx *ctx, const uint32_t *quads_xy, size_t n, float min_distance, uint32_t *out_xy, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
