/*
 * caller.c -- a C11 program that uses libaruco3_hip.so through include/aruco3_hip.h and nothing else: the nearest thing to the
 * reference's own callers (src/lib.rs:6-9, src/aruco.rs:46-52: `Detector { config, dictionary }.detect(image)`) that this image
 * can compile.  Built by tests/test_c_abi.py with  gcc -std=c11 -pedantic -Wall -Wextra -Werror.
 *
 *   caller layout
 *       prints sizeof / offsetof of every field of every struct of the header, one `struct.field offset size` line each
 *       (compared with the #[repr(C)] declarations of integration/aruco3_hip.rs and with aruco3_amd/_lib.py's ctypes)
 *   caller detect <frame.raw> <width> <height> <fmt> <dictionaries.bin> <offset> <count> <num_bits> <tau>
 *       a3_create -> a3_detect_batch (one host frame) -> a3_get_stats -> a3_destroy; one `marker ...` line per marker
 */
#include <inttypes.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aruco3_hip.h"

#define FIELD(S, F) printf(#S "." #F " %zu %zu\n", offsetof(S, F), sizeof(((S *)0)->F))
#define WHOLE(S) printf(#S " - %zu\n", sizeof(S))

static int layout(void) {
    WHOLE(a3_config);
    FIELD(a3_config, threshold_window); FIELD(a3_config, contour_simplification_epsilon); FIELD(a3_config, min_side_length_factor);
    FIELD(a3_config, min_corner_separation_factor); FIELD(a3_config, homography_sample_size); FIELD(a3_config, filter_high_bit_errors);
    WHOLE(a3_marker);
    FIELD(a3_marker, frame); FIELD(a3_marker, id); FIELD(a3_marker, code); FIELD(a3_marker, corners); FIELD(a3_marker, hamming_distance);
    FIELD(a3_marker, rotation); FIELD(a3_marker, candidate_index);
    WHOLE(a3_pose);
    FIELD(a3_pose, error); FIELD(a3_pose, rotation); FIELD(a3_pose, translation);
    WHOLE(a3_intrinsics);
    FIELD(a3_intrinsics, image_width); FIELD(a3_intrinsics, image_height); FIELD(a3_intrinsics, focal_x); FIELD(a3_intrinsics, focal_y);
    FIELD(a3_intrinsics, principal_x); FIELD(a3_intrinsics, principal_y);
    WHOLE(a3_stats);
    FIELD(a3_stats, darts); FIELD(a3_stats, contours_traced); FIELD(a3_stats, contours_materialised); FIELD(a3_stats, candidates_pre);
    FIELD(a3_stats, candidates); FIELD(a3_stats, markers); FIELD(a3_stats, resolve_iterations); FIELD(a3_stats, jump_rounds);
    FIELD(a3_stats, chunks); FIELD(a3_stats, stepping);
    WHOLE(a3_synth_marker);
    FIELD(a3_synth_marker, hinv); FIELD(a3_synth_marker, x0); FIELD(a3_synth_marker, y0); FIELD(a3_synth_marker, x1); FIELD(a3_synth_marker, y1);
    FIELD(a3_synth_marker, cells); FIELD(a3_synth_marker, n); FIELD(a3_synth_marker, reserved);
    WHOLE(a3_synth_frame);
    FIELD(a3_synth_frame, base); FIELD(a3_synth_frame, gx); FIELD(a3_synth_frame, gy); FIELD(a3_synth_frame, noise_sigma);
    FIELD(a3_synth_frame, first_marker); FIELD(a3_synth_frame, n_markers); FIELD(a3_synth_frame, seed);
    printf("A3_ABI_VERSION - %d\n", A3_ABI_VERSION);
    return 0;
}

static void *slurp(const char *path, size_t offset, size_t bytes) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fseek(f, (long)offset, SEEK_SET) != 0 || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "%s: short read\n", path);
        free(p); fclose(f);
        return NULL;
    }
    fclose(f);
    return p;
}

static int detect(char **a) {
    const uint32_t w = (uint32_t)strtoul(a[1], NULL, 10), h = (uint32_t)strtoul(a[2], NULL, 10);
    const int fmt = atoi(a[3]);
    const size_t off = (size_t)strtoull(a[5], NULL, 10), count = (size_t)strtoull(a[6], NULL, 10);
    const uint8_t num_bits = (uint8_t)atoi(a[7]), tau = (uint8_t)atoi(a[8]);
    const size_t bpp = fmt == A3_FMT_L8 ? 1 : (fmt == A3_FMT_RGB8 ? 3 : 4);
    uint8_t *px = slurp(a[0], 0, (size_t)w * h * bpp);
    uint64_t *codes = slurp(a[4], off * sizeof(uint64_t), count * sizeof(uint64_t));
    if (!px || !codes) return 2;
    if (a3_abi_version() != A3_ABI_VERSION) { fprintf(stderr, "header is ABI %d, library %d\n", A3_ABI_VERSION, a3_abi_version()); return 3; }

    a3_config cfg;
    a3_default_config(&cfg);          /* DetectorConfig::default(), src/aruco.rs:32-43 */
    printf("config %" PRIu32 " %.17g %.9g %.9g %" PRIu32 " %u\n", cfg.threshold_window, cfg.contour_simplification_epsilon,
           (double)cfg.min_side_length_factor, (double)cfg.min_corner_separation_factor, cfg.homography_sample_size, (unsigned)cfg.filter_high_bit_errors);
    a3_ctx *ctx = NULL;
    int rc = a3_create(0, &cfg, codes, count, num_bits, tau, &ctx);
    if (rc != A3_OK) { fprintf(stderr, "a3_create: %d %s\n", rc, a3_last_error(NULL)); return 4; }

    enum { CAP = 64 };
    a3_marker out[CAP];
    uint32_t per_frame[1] = {0};
    size_t n = 0;
    rc = a3_detect_batch(ctx, px, A3_MEM_HOST, fmt, w, h, (size_t)w * bpp, (size_t)w * h * bpp, 1, out, CAP, per_frame, &n);
    if (rc != A3_OK) { fprintf(stderr, "a3_detect_batch: %d %s\n", rc, a3_last_error(ctx)); a3_destroy(ctx); return 5; }
    /* an output array that is too small is an error, never a clip (ownership) */
    size_t n_small = 0;
    const int rc_small = n > 1 ? a3_detect_batch(ctx, px, A3_MEM_HOST, fmt, w, h, (size_t)w * bpp, (size_t)w * h * bpp, 1, out, n - 1, per_frame, &n_small) : A3_ERR_CAPACITY;
    printf("too_small_rc %d\n", rc_small);
    rc = a3_detect_batch(ctx, px, A3_MEM_HOST, fmt, w, h, (size_t)w * bpp, (size_t)w * h * bpp, 1, out, CAP, per_frame, &n);
    if (rc != A3_OK) { fprintf(stderr, "a3_detect_batch (again): %d %s\n", rc, a3_last_error(ctx)); a3_destroy(ctx); return 5; }
    a3_stats st;
    memset(&st, 0, sizeof st);
    if (a3_get_stats(ctx, &st) != A3_OK) { a3_destroy(ctx); return 6; }
    printf("markers %zu per_frame %" PRIu32 " candidates_pre %" PRIu64 " candidates %" PRIu64 "\n", n, per_frame[0], st.candidates_pre, st.candidates);
    for (size_t i = 0; i < n; i++) {
        const a3_marker *m = &out[i];
        printf("marker %" PRIu32 " %" PRIu32 " %" PRIu64 " %u %u %u", m->frame, m->id, m->code, (unsigned)m->hamming_distance, (unsigned)m->rotation,
               (unsigned)m->candidate_index);
        for (int k = 0; k < 8; k++) printf(" %" PRIu32, m->corners[k]);
        printf("\n");
    }
    /* pose::solve_with_undistorted_points of the first marker (src/pose.rs:59-62), two solutions, lower error first */
    if (n) {
        a3_pose poses[2];
        rc = a3_estimate_pose(ctx, out[0].corners, 1, 40.0f, NULL, w, h, poses);
        if (rc != A3_OK) { fprintf(stderr, "a3_estimate_pose: %d %s\n", rc, a3_last_error(ctx)); a3_destroy(ctx); return 7; }
        printf("pose_errors %.9g %.9g\n", (double)poses[0].error, (double)poses[1].error);
    }
    /* bad arguments come back as codes with a message, nothing unwinds */
    printf("null_pixels_rc %d\n", a3_detect_batch(ctx, NULL, A3_MEM_HOST, fmt, w, h, (size_t)w * bpp, (size_t)w * h * bpp, 1, out, CAP, per_frame, &n_small));
    a3_destroy(ctx);
    free(px); free(codes);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 2 && strcmp(argv[1], "layout") == 0) return layout();
    if (argc == 11 && strcmp(argv[1], "detect") == 0) return detect(argv + 2);
    fprintf(stderr, "usage: caller layout | caller detect <frame.raw> <w> <h> <fmt> <dictionaries.bin> <offset> <count> <num_bits> <tau>\n");
    return 64;
}
