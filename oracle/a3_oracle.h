/*
 * a3_oracle.h -- CPU restatement of the aruco3 detection + pose path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / as the timed CPU baseline ("port").
 *
 * PARITY STATUS: "parity unpinned" for the image stages.  The reference is Rust
 * (no toolchain here) and the per-pixel arithmetic of stages C,D,E,G,H,K,L of
 * SURVEY.md section 8a lives in the crates `image` ^0.25 / `imageproc` ^0.25 /
 * `nalgebra` ^0.33 whose sources are not under /root/reference (no Cargo.lock, not
 * vendored).  Those stages restate the crates' published algorithms from their
 * call sites in src/aruco.rs.  What IS pinned by the reference's own tests:
 * hamming (src/lib.rs:28-40), find_nearest / try_find_nearest / tau
 * (src/dictionaries.rs:239-281), enforce_clockwise / rotate_bit_matrix /
 * discard_too_near (src/aruco.rs:400-459) and every pose KAT (src/pose.rs:379-598)
 * -- see tests/test_oracle_kat.py.
 */
#ifndef A3_ORACLE_H
#define A3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { A3O_FMT_RGB8 = 0, A3O_FMT_RGBA8 = 1, A3O_FMT_L8 = 2 };

/* mirrors DetectorConfig, src/aruco.rs:23-43 */
typedef struct {
    uint32_t threshold_window;
    double   contour_simplification_epsilon;
    float    min_side_length_factor;
    float    min_corner_separation_factor;
    uint32_t homography_sample_size;
    uint8_t  filter_high_bit_errors;
} a3o_config;

/* mirrors Marker, src/aruco.rs:8-13 (+ the rotation index that was applied) */
typedef struct {
    uint32_t id;
    uint32_t rotation;
    uint64_t code;
    uint32_t corners[8]; /* x0,y0,...,x3,y3 */
    uint32_t hamming_distance;
    uint32_t candidate_index; /* index into candidates[] this marker came from */
} a3o_marker;

/* mirrors MarkerPose, src/pose.rs:8-12; rotation row-major */
typedef struct {
    float error;
    float rotation[9];
    float translation[3];
} a3o_pose;

typedef struct {
    uint32_t *offsets; /* n_contours + 1 */
    uint32_t *points;  /* 2 * offsets[n_contours], x then y */
    uint8_t  *border_type; /* 0 outer, 1 hole */
    int32_t  *parent;      /* -1 none */
    uint32_t  n_contours;
} a3o_contours;

/* mirrors Detection, src/aruco.rs:15-21, plus every intermediate as a debug tap */
typedef struct {
    uint32_t width, height;
    uint8_t *grey;        /* w*h */
    uint8_t *thresholded; /* w*h, 0/255 */
    uint32_t n_contours;
    uint64_t n_contour_points;
    uint32_t stat_reject_point_count, stat_reject_convexity, stat_reject_edge_length;
    uint32_t n_candidates_pre;  /* after contours_to_candidates */
    uint32_t *candidates_pre;   /* n*8, after enforce_clockwise, before discard */
    uint32_t *candidates_pre_start; /* n: raster index (y*w+x) of the source contour's first point */
    uint32_t n_candidates;      /* after discard_too_near */
    uint32_t *candidates;       /* n*8 */
    uint8_t  *homographies;     /* n * S*S; a failed projection leaves zeros */
    uint8_t  *homography_ok;    /* n: 1 = warped, 0 = 1x1 black stand-in */
    uint32_t sample;            /* S */
    int32_t  *decode_ok;        /* n: 1 if the border test passed */
    uint64_t *codes;            /* n*4 */
    uint32_t n_markers;
    a3o_marker *markers;
} a3o_detection;

/* ---- L0: dictionaries ---- */
uint32_t a3o_hamming_distance(uint64_t a, uint64_t b);
uint8_t  a3o_calculate_tau(const uint64_t *codes, size_t n);
uint8_t  a3o_mark_size(uint8_t num_bits);
void     a3o_find_nearest(const uint64_t *codes, size_t n, uint64_t bits, size_t *idx, uint8_t *dist);
/* width and width*width cells (0/1), src/dictionaries.rs:212-232 */
uint8_t  a3o_make_binary_image(uint64_t code, uint8_t num_bits, uint8_t *cells);

/* ---- L1: image ops ---- */
void a3o_to_luma8(const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, uint8_t *grey);
void a3o_adaptive_threshold(const uint8_t *grey, uint32_t w, uint32_t h, uint32_t block_radius, uint8_t *out);
int  a3o_find_contours(const uint8_t *bin, uint32_t w, uint32_t h, a3o_contours *out);
void a3o_free_contours(a3o_contours *c);
/* returns number of output points written to out (x,y pairs); out must hold n points */
size_t a3o_approximate_polygon_dp(const uint32_t *pts, size_t n, double epsilon, int closed, uint32_t *out);
size_t a3o_convex_hull(const uint32_t *pts, size_t n, uint32_t *out);
int  a3o_from_control_points(const float from[8], const float to[8], float transform[9], float inverse[9]);
void a3o_warp_into(const uint8_t *grey, uint32_t w, uint32_t h, const float map[9], uint8_t *out, uint32_t ow, uint32_t oh);
uint8_t a3o_otsu_level(const uint8_t *img, uint32_t w, uint32_t h);
void a3o_resize_triangle(const uint8_t *img, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out);

/* ---- L2: aruco.rs helpers ---- */
void   a3o_enforce_clockwise_corners(uint32_t *quads, size_t n);
size_t a3o_discard_too_near(uint32_t *quads, size_t n, float min_distance, uint32_t *kept_index);
void   a3o_rotate_bit_matrix(const uint8_t *in, uint32_t rows, uint32_t cols, uint8_t *out);
int    a3o_homography_to_code_permutations(const uint8_t *patch, uint32_t pw, uint32_t ph, uint8_t mark_size, uint64_t codes[4]);

int  a3o_detect(const a3o_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
                const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, int keep_debug,
                a3o_detection *out);
/* a3o_detect with the candidate list (after enforce_clockwise_corners, before discard_too_near) handed in: test aid for quirk Q4 */
int  a3o_detect_quads(const a3o_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
                      const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, int keep_debug,
                      const uint32_t *quads, size_t n_quads, a3o_detection *out);
void a3o_free_detection(a3o_detection *d);

/* ---- L3: pose ---- */
void a3o_make_marker_square(float marker_size_mm, float out[12]);
void a3o_compute_homography_from_marker_square(float marker_size_mm, const float pts[8], float h[9]);
void a3o_solve_canonical_form(const float obj[12], const float pts[8], const float h[9], a3o_pose *p1, a3o_pose *p2);
void a3o_solve_with_normalized_points(const float pts[8], float marker_size_mm, a3o_pose *p1, a3o_pose *p2);
void a3o_solve_with_undistorted_points(const uint32_t corners[8], float marker_size_mm, uint32_t iw, uint32_t ih,
                                       a3o_pose *p1, a3o_pose *p2);
void a3o_solve_with_intrinsics(const uint32_t corners[8], float marker_size_mm, float fx, float fy, float cx, float cy,
                               a3o_pose *p1, a3o_pose *p2);
void a3o_apply_transform(const a3o_pose *p, const float *pts, size_t n, float *out);
void a3o_apply_inverse_transform(const a3o_pose *p, const float *pts, size_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif
