/*
 * a3_oracle.c -- CPU restatement of aruco3's Detector::detect() and pose solvers.
 *
 * TEST INFRASTRUCTURE ONLY (see a3_oracle.h).  "parity unpinned" for the stages that
 * live in third-party crates (image ^0.25, imageproc ^0.25, nalgebra ^0.33; not under
 * /root/reference): they restate those crates' published algorithms, stage by stage,
 * anchored on the call sites in src/aruco.rs.  First-party stages follow
 * src/aruco.rs, src/dictionaries.rs, src/lib.rs, src/pose.rs, src/pinhole.rs line by
 * line and are pinned by the reference's own known-answer tests.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (no FMA contraction: Rust
 * never contracts, and the HIP path is built the same way).
 */
#include "a3_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small growable arrays                                                     */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t *v; size_t n, cap; } u32vec;

static int u32vec_push(u32vec *a, uint32_t x) {
    if (a->n == a->cap) {
        size_t nc = a->cap ? a->cap * 2 : 1024;
        uint32_t *nv = (uint32_t *)realloc(a->v, nc * sizeof(uint32_t));
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* L0  dictionaries                                                          */
/* ------------------------------------------------------------------------- */

/* src/lib.rs:11-21 -- bit loop popcount of a^b */
uint32_t a3o_hamming_distance(uint64_t a, uint64_t b) {
    uint64_t flipped = a ^ b;
    uint32_t count = 0;
    while (flipped > 0) {
        if (flipped % 2 == 1) count += 1;
        flipped >>= 1;
    }
    return count;
}

/* src/dictionaries.rs:129-138 */
uint8_t a3o_calculate_tau(const uint64_t *codes, size_t n) {
    uint32_t tau = 255;
    for (size_t i = 0; i < n; i++)
        for (size_t j = i + 1; j < n; j++) {
            uint32_t d = a3o_hamming_distance(codes[i], codes[j]);
            if (d < tau) tau = d;
        }
    return (uint8_t)tau;
}

/* src/dictionaries.rs:154-156 */
uint8_t a3o_mark_size(uint8_t num_bits) {
    return (uint8_t)((uint8_t)ceilf(sqrtf((float)num_bits)) + 2);
}

/* src/dictionaries.rs:160-196 -- strict '<' so the lowest index wins ties */
void a3o_find_nearest(const uint64_t *codes, size_t n, uint64_t bits, size_t *idx, uint8_t *dist) {
    size_t min_index = 0;
    uint8_t min_distance = 0xFF;
    for (size_t i = 0; i < n; i++) {
        uint8_t d = (uint8_t)a3o_hamming_distance(codes[i], bits);
        if (d < min_distance) { min_distance = d; min_index = i; }
    }
    *idx = min_index;
    *dist = min_distance;
}

/* src/dictionaries.rs:212-232 -- LSB-first cell packing (quirk Q6) */
uint8_t a3o_make_binary_image(uint64_t code, uint8_t num_bits, uint8_t *cells) {
    uint8_t width = a3o_mark_size(num_bits);
    size_t len = 0;
    for (uint8_t i = 0; i < width; i++) cells[len++] = 0;
    for (uint8_t i = 0; i < num_bits; i++) {
        if ((uint8_t)len % width == 0) cells[len++] = 0;
        cells[len++] = (code & ((uint64_t)1 << i)) != 0;
        if ((uint8_t)len % width == width - 1) cells[len++] = 0;
    }
    for (uint8_t i = 0; i < width; i++) cells[len++] = 0;
    return width;
}

/* ------------------------------------------------------------------------- */
/* L1  image ops (third-party behaviour, restated)                           */
/* ------------------------------------------------------------------------- */

/* image ^0.25 DynamicImage::into_luma8 (call: src/aruco.rs:60).
 * Rgb8/Rgba8: L = (2126 R + 7152 G + 722 B) / 10000 in u32, truncating; alpha
 * ignored; Luma8 passes through. */
void a3o_to_luma8(const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, uint8_t *grey) {
    const uint32_t bpp = fmt == A3O_FMT_RGB8 ? 3 : fmt == A3O_FMT_RGBA8 ? 4 : 1;
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = px + (size_t)y * row_stride;
        uint8_t *g = grey + (size_t)y * w;
        if (fmt == A3O_FMT_L8) { memcpy(g, row, w); continue; }
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *p = row + (size_t)x * bpp;
            uint32_t l = 2126u * p[0] + 7152u * p[1] + 722u * p[2];
            g[x] = (uint8_t)(l / 10000u);
        }
    }
}

/* imageproc 0.25 contrast::adaptive_threshold(image, block_radius)
 * (call: src/aruco.rs:61).  u32 integral image; window clipped to the image; mean =
 * sum / clipped_area (truncating); white iff px >= mean. */
void a3o_adaptive_threshold(const uint8_t *grey, uint32_t w, uint32_t h, uint32_t r, uint8_t *out) {
    /* integral image with a zero row/column in front: I[(y+1)][(x+1)] = sum over [0..=y]x[0..=x] */
    size_t iw = (size_t)w + 1;
    uint32_t *I = (uint32_t *)calloc(iw * ((size_t)h + 1), sizeof(uint32_t));
    for (uint32_t y = 0; y < h; y++) {
        uint32_t rowsum = 0;
        for (uint32_t x = 0; x < w; x++) {
            rowsum += grey[(size_t)y * w + x];
            I[(size_t)(y + 1) * iw + (x + 1)] = I[(size_t)y * iw + (x + 1)] + rowsum;
        }
    }
    for (uint32_t y = 0; y < h; y++) {
        uint32_t y_low = y > r ? y - r : 0;
        uint32_t y_high = (y + r < h - 1) ? y + r : h - 1;
        for (uint32_t x = 0; x < w; x++) {
            uint32_t x_low = x > r ? x - r : 0;
            uint32_t x_high = (x + r < w - 1) ? x + r : w - 1;
            uint32_t area = (y_high - y_low + 1) * (x_high - x_low + 1);
            uint32_t sum = I[(size_t)(y_high + 1) * iw + (x_high + 1)] - I[(size_t)y_low * iw + (x_high + 1)]
                         - I[(size_t)(y_high + 1) * iw + x_low] + I[(size_t)y_low * iw + x_low];
            uint32_t mean = sum / area;
            out[(size_t)y * w + x] = ((uint32_t)grey[(size_t)y * w + x] >= mean) ? 255 : 0;
        }
    }
    free(I);
}

/* imageproc 0.25 contours::find_contours::<u32> == find_contours_with_threshold(img, 0)
 * (call: src/aruco.rs:64).  Suzuki-Abe border following, 8-connected foreground.
 * Neighbour ring, clockwise on screen: W NW N NE E SE S SW. */
static const int DX[8] = { -1, -1, 0, 1, 1, 1, 0, -1 };
static const int DY[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

static int dir_index(int dx, int dy) {
    for (int k = 0; k < 8; k++) if (DX[k] == dx && DY[k] == dy) return k;
    return -1;
}

int a3o_find_contours(const uint8_t *bin, uint32_t w, uint32_t h, a3o_contours *out) {
    const int W = (int)w, H = (int)h;
    int32_t *val = (int32_t *)malloc((size_t)w * h * sizeof(int32_t));
    if (!val) return -1;
    for (size_t i = 0; i < (size_t)w * h; i++) val[i] = bin[i] > 0 ? 1 : 0;
#define AT(x, y) val[(size_t)(y) * w + (x)]
#define NONZERO(x, y) ((x) >= 0 && (x) < W && (y) >= 0 && (y) < H && AT(x, y) != 0)

    u32vec offs = {0}, pts = {0}, btype = {0}, parents = {0};
    u32vec_push(&offs, 0);
    int32_t curr_border_num = 1;

    for (int y = 0; y < H; y++) {
        int32_t parent_border_num = 1;
        for (int x = 0; x < W; x++) {
            if (AT(x, y) == 0) continue;
            int adjx = 0, adjy = y, have = 0, is_hole = 0;
            if (AT(x, y) == 1 && x > 0 && AT(x - 1, y) == 0) { adjx = x - 1; have = 1; is_hole = 0; }
            else if (AT(x, y) > 0 && x + 1 < W && AT(x + 1, y) == 0) { adjx = x + 1; have = 1; is_hole = 1; }
            if (have) {
                curr_border_num += 1;
                int32_t parent = -1;
                if (parent_border_num > 1) {
                    int32_t pi = parent_border_num - 2;
                    int parent_is_outer = btype.v[pi] == 0;
                    if ((!is_hole) ^ parent_is_outer) parent = pi;
                    else parent = (int32_t)parents.v[pi];
                }
                /* clockwise search starting at adj for the first non-zero neighbour */
                int f = dir_index(adjx - x, adjy - y);
                int p1x = 0, p1y = 0, found = 0;
                for (int k = 0; k < 8; k++) {
                    int d = (f + k) & 7;
                    int nx = x + DX[d], ny = y + DY[d];
                    if (NONZERO(nx, ny)) { p1x = nx; p1y = ny; found = 1; break; }
                }
                if (found) {
                    int p2x = p1x, p2y = p1y, p3x = x, p3y = y;
                    for (;;) {
                        u32vec_push(&pts, (uint32_t)p3x);
                        u32vec_push(&pts, (uint32_t)p3y);
                        f = dir_index(p2x - p3x, p2y - p3y);
                        /* counter-clockwise: the rotated ring iterated in reverse */
                        int p4x = 0, p4y = 0, d4 = -1;
                        for (int k = 7; k >= 0; k--) {
                            int d = (f + k) & 7;
                            int nx = p3x + DX[d], ny = p3y + DY[d];
                            if (NONZERO(nx, ny)) { p4x = nx; p4y = ny; d4 = d; break; }
                        }
                        int is_right_edge = 0;
                        for (int k = 7; k >= 0; k--) {
                            int d = (f + k) & 7;
                            if (d == d4) break;
                            if (d == 4) { is_right_edge = 1; break; }
                        }
                        if (p3x + 1 == W || is_right_edge) AT(p3x, p3y) = -curr_border_num;
                        else if (AT(p3x, p3y) == 1) AT(p3x, p3y) = curr_border_num;
                        if (p4x == x && p4y == y && p3x == p1x && p3y == p1y) break;
                        p2x = p3x; p2y = p3y; p3x = p4x; p3y = p4y;
                    }
                } else {
                    u32vec_push(&pts, (uint32_t)x);
                    u32vec_push(&pts, (uint32_t)y);
                    AT(x, y) = -curr_border_num;
                }
                u32vec_push(&offs, (uint32_t)(pts.n / 2));
                u32vec_push(&btype, (uint32_t)is_hole);
                u32vec_push(&parents, (uint32_t)parent);
            }
            if (AT(x, y) != 1) parent_border_num = AT(x, y) < 0 ? -AT(x, y) : AT(x, y);
        }
    }
#undef AT
#undef NONZERO
    free(val);
    out->n_contours = (uint32_t)btype.n;
    out->offsets = offs.v;
    out->points = pts.v ? pts.v : (uint32_t *)calloc(2, sizeof(uint32_t));
    out->border_type = (uint8_t *)malloc(btype.n ? btype.n : 1);
    out->parent = (int32_t *)malloc((btype.n ? btype.n : 1) * sizeof(int32_t));
    for (size_t i = 0; i < btype.n; i++) { out->border_type[i] = (uint8_t)btype.v[i]; out->parent[i] = (int32_t)parents.v[i]; }
    free(btype.v); free(parents.v);
    return 0;
}

void a3o_free_contours(a3o_contours *c) {
    free(c->offsets); free(c->points); free(c->border_type); free(c->parent);
    memset(c, 0, sizeof(*c));
}

/* imageproc 0.25 geometry::approximate_polygon_dp (call: src/aruco.rs:133).
 * Distance to the INFINITE line through curve[first], curve[last] in f64; first strict
 * maximum; the same epsilon at every level; the recursion's concatenation keeps the
 * retained indices in increasing order, so a work stack that marks them is identical. */
size_t a3o_approximate_polygon_dp(const uint32_t *pts, size_t n, double epsilon, int closed, uint32_t *out) {
    if (n == 0) return 0;
    uint8_t *keep = (uint8_t *)calloc(n, 1);
    size_t *stack = (size_t *)malloc(2 * (n + 1) * sizeof(size_t));
    size_t sp = 0;
    int first_equals_last = 0;
    stack[sp++] = 0; stack[sp++] = n - 1;
    keep[0] = 1; keep[n - 1] = 1;
    if (n == 1) first_equals_last = 1; /* result is [c0, c0] before the closed pop */
    while (sp) {
        size_t b = stack[--sp], a = stack[--sp];
        double px = pts[2 * a], py = pts[2 * a + 1], qx = pts[2 * b], qy = pts[2 * b + 1];
        double la = py - qy, lb = qx - px, lc = px * qy - qx * py;
        double denom = sqrt(la * la + lb * lb);
        double dmax = 0.0; size_t index = a;
        for (size_t i = a + 1; i <= b; i++) {
            double d = fabs(la * (double)pts[2 * i] + lb * (double)pts[2 * i + 1] + lc) / denom;
            if (d > dmax) { index = i; dmax = d; }
        }
        if (dmax > epsilon) {
            keep[index] = 1;
            stack[sp++] = a; stack[sp++] = index;
            stack[sp++] = index; stack[sp++] = b;
        }
    }
    size_t m = 0;
    for (size_t i = 0; i < n; i++) if (keep[i]) { out[2 * m] = pts[2 * i]; out[2 * m + 1] = pts[2 * i + 1]; m++; }
    if (first_equals_last) { out[2 * m] = pts[0]; out[2 * m + 1] = pts[1]; m++; }
    if (closed) m--;
    free(keep); free(stack);
    return m;
}

/* imageproc 0.25 geometry::convex_hull (call: src/aruco.rs:143): Graham scan. */
static int orientation(int64_t px, int64_t py, int64_t qx, int64_t qy, int64_t rx, int64_t ry) {
    int64_t v = (qy - py) * (rx - qx) - (qx - px) * (ry - qy);
    return v == 0 ? 0 : (v > 0 ? 1 : -1); /* 0 collinear, 1 clockwise, -1 counter-clockwise */
}

size_t a3o_convex_hull(const uint32_t *pts_in, size_t n, uint32_t *out) {
    if (n == 0) return 0;
    int64_t *p = (int64_t *)malloc(2 * n * sizeof(int64_t));
    for (size_t i = 0; i < 2 * n; i++) p[i] = pts_in[i];
    size_t sp0 = 0;
    for (size_t i = 1; i < n; i++)
        if (p[2 * i + 1] < p[2 * sp0 + 1] || (p[2 * i + 1] == p[2 * sp0 + 1] && p[2 * i] < p[2 * sp0])) sp0 = i;
    int64_t sx = p[2 * sp0], sy = p[2 * sp0 + 1];
    /* points.swap(0, pos); points.remove(0) */
    p[2 * sp0] = p[0]; p[2 * sp0 + 1] = p[1];
    int64_t *q = p + 2; size_t m = n - 1;
    /* stable insertion sort with the reference comparator (never returns Equal) */
    for (size_t i = 1; i < m; i++) {
        int64_t ax = q[2 * i], ay = q[2 * i + 1];
        size_t j = i;
        while (j > 0) {
            int64_t bx = q[2 * (j - 1)], by = q[2 * (j - 1) + 1];
            int o = orientation(sx, sy, ax, ay, bx, by);
            int less;
            if (o == 0) {
                int64_t da = (ax - sx) * (ax - sx) + (ay - sy) * (ay - sy);
                int64_t db = (bx - sx) * (bx - sx) + (by - sy) * (by - sy);
                less = da < db;
            } else less = (o == -1);
            if (!less) break;
            q[2 * j] = bx; q[2 * j + 1] = by; j--;
        }
        q[2 * j] = ax; q[2 * j + 1] = ay;
    }
    int64_t *st = (int64_t *)malloc(2 * (n + 1) * sizeof(int64_t));
    size_t sn = 0;
    st[0] = sx; st[1] = sy; sn = 1;
    for (size_t i = 0; i < m; i++) {
        int64_t rx = q[2 * i], ry = q[2 * i + 1];
        while (sn > 1 && orientation(st[2 * (sn - 2)], st[2 * (sn - 2) + 1], st[2 * (sn - 1)], st[2 * (sn - 1) + 1], rx, ry) != -1) sn--;
        st[2 * sn] = rx; st[2 * sn + 1] = ry; sn++;
    }
    for (size_t i = 0; i < 2 * sn; i++) out[i] = (uint32_t)st[i];
    free(p); free(st);
    return sn;
}

/* imageproc 0.25 Projection::from_control_points (call: src/aruco.rs:244): 8x8 DLT
 * system in f64, LU with partial pivoting (nalgebra), h33 = 1, cast to f32, 3x3
 * adjugate inverse in f32 normalised by its last element. */
static int lu_solve8(double A[8][8], double b[8]) {
    int perm_i[8], perm_j[8], np = 0;
    for (int i = 0; i < 8; i++) {
        int piv = i; double best = fabs(A[i][i]);
        for (int r = i + 1; r < 8; r++) { double v = fabs(A[r][i]); if (v > best) { best = v; piv = r; } }
        double diag = A[piv][i];
        if (diag == 0.0) continue;
        if (piv != i) {
            perm_i[np] = i; perm_j[np] = piv; np++;
            for (int c = 0; c < 8; c++) { double t = A[i][c]; A[i][c] = A[piv][c]; A[piv][c] = t; }
        }
        double inv_diag = 1.0 / diag;
        for (int r = i + 1; r < 8; r++) A[r][i] *= inv_diag;
        for (int c = i + 1; c < 8; c++) {
            double pr = -A[i][c];
            for (int r = i + 1; r < 8; r++) A[r][c] = pr * A[r][i] + A[r][c];
        }
    }
    for (int k = 0; k < np; k++) { double t = b[perm_i[k]]; b[perm_i[k]] = b[perm_j[k]]; b[perm_j[k]] = t; }
    for (int i = 0; i < 7; i++) {           /* unit lower triangular, column oriented */
        double coeff = -b[i];
        for (int r = i + 1; r < 8; r++) b[r] = coeff * A[r][i] + b[r];
    }
    for (int i = 7; i >= 0; i--) {          /* upper triangular, column oriented */
        double diag = A[i][i];
        if (diag == 0.0) return 0;
        double coeff = b[i] / diag;
        b[i] = coeff;
        double nc = -coeff;
        for (int r = 0; r < i; r++) b[r] = nc * A[r][i] + b[r];
    }
    return 1;
}

static int try_inverse3(const float t[9], float inv[9]) {
    float t00 = t[0], t01 = t[1], t02 = t[2], t10 = t[3], t11 = t[4], t12 = t[5], t20 = t[6], t21 = t[7], t22 = t[8];
    float m00 = t11 * t22 - t12 * t21;
    float m01 = t10 * t22 - t12 * t20;
    float m02 = t10 * t21 - t11 * t20;
    float det = t00 * m00 - t01 * m01 + t02 * m02;
    if (fabsf(det) < 1e-10f) return 0;
    float m10 = t01 * t22 - t02 * t21;
    float m11 = t00 * t22 - t02 * t20;
    float m12 = t00 * t21 - t01 * t20;
    float m20 = t01 * t12 - t02 * t11;
    float m21 = t00 * t12 - t02 * t10;
    float m22 = t00 * t11 - t01 * t10;
    float r[9] = { m00 / det, -m10 / det, m20 / det, -m01 / det, m11 / det, -m21 / det, m02 / det, -m12 / det, m22 / det };
    for (int i = 0; i < 8; i++) inv[i] = r[i] / r[8];
    inv[8] = 1.0f;
    return 1;
}

int a3o_from_control_points(const float from[8], const float to[8], float transform[9], float inverse[9]) {
    double A[8][8], b[8];
    for (int i = 0; i < 4; i++) {
        double xf = from[2 * i], yf = from[2 * i + 1], x = to[2 * i], y = to[2 * i + 1];
        double r0[8] = { 0.0, 0.0, 0.0, -xf, -yf, -1.0, y * xf, y * yf };
        double r1[8] = { xf, yf, 1.0, 0.0, 0.0, 0.0, -x * xf, -x * yf };
        memcpy(A[2 * i], r0, sizeof r0);
        memcpy(A[2 * i + 1], r1, sizeof r1);
        b[2 * i] = -y; b[2 * i + 1] = x;
    }
    if (!lu_solve8(A, b)) return 0;
    for (int i = 0; i < 8; i++) transform[i] = (float)b[i];
    transform[8] = 1.0f;
    return try_inverse3(transform, inverse);
}

/* imageproc 0.25 warp_into + interpolate_bilinear (call: src/aruco.rs:253).
 * `map` is the matrix applied to OUTPUT pixel coordinates (Projection::invert() of the
 * control-point projection, i.e. its `inverse`); integer pixel coordinates, no centre
 * offset; the two horizontal lerps are truncated to u8 before the vertical lerp. */
static uint8_t clamp_u8(float x) {
    if (x < 255.0f) { if (x > 0.0f) return (uint8_t)x; return 0; }
    return 255; /* also NaN */
}

static uint32_t sat_u32(float x) { /* Rust `as u32` */
    if (!(x > 0.0f)) return 0; /* negative, zero, NaN */
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

static uint8_t sample_bilinear(const uint8_t *img, uint32_t w, uint32_t h, float x, float y) {
    float left = floorf(x), right = left + 1.0f, top = floorf(y), bottom = top + 1.0f;
    float rw = x - left, bw = y - top;
    if (left < 0.0f || right >= (float)w || top < 0.0f || bottom >= (float)h) return 0;
    uint32_t l = sat_u32(left), r = sat_u32(right), t = sat_u32(top), b = sat_u32(bottom);
    float tl = img[(size_t)t * w + l], tr = img[(size_t)t * w + r], bl = img[(size_t)b * w + l], br = img[(size_t)b * w + r];
    uint8_t tv = clamp_u8((1.0f - rw) * tl + rw * tr);
    uint8_t bv = clamp_u8((1.0f - rw) * bl + rw * br);
    return clamp_u8((1.0f - bw) * (float)tv + bw * (float)bv);
}

void a3o_warp_into(const uint8_t *grey, uint32_t w, uint32_t h, const float t[9], uint8_t *out, uint32_t ow, uint32_t oh) {
    for (uint32_t y = 0; y < oh; y++)
        for (uint32_t x = 0; x < ow; x++) {
            float fx = (float)x, fy = (float)y;
            float d = t[6] * fx + t[7] * fy + t[8];
            float px = (t[0] * fx + t[1] * fy + t[2]) / d;
            float py = (t[3] * fx + t[4] * fy + t[5]) / d;
            out[(size_t)y * ow + x] = sample_bilinear(grey, w, h, px, py);
        }
}

/* imageproc 0.25 contrast::otsu_level (call: src/aruco.rs:264) */
uint8_t a3o_otsu_level(const uint8_t *img, uint32_t w, uint32_t h) {
    uint32_t hist[256] = {0};
    for (size_t i = 0; i < (size_t)w * h; i++) hist[img[i]]++;
    uint32_t total_weight = w * h;
    double total_pixel_sum = 0.0;
    for (uint32_t t = 0; t < 256; t++) total_pixel_sum = total_pixel_sum + (double)(t * hist[t]);
    double background_pixel_sum = 0.0;
    uint32_t background_weight = 0, foreground_weight;
    double largest_variance = 0.0;
    uint8_t best_threshold = 0;
    for (uint32_t t = 0; t < 256; t++) {
        background_weight += hist[t];
        if (background_weight == 0) continue;
        foreground_weight = total_weight - background_weight;
        if (foreground_weight == 0) break;
        background_pixel_sum += (double)(t * hist[t]);
        double foreground_pixel_sum = total_pixel_sum - background_pixel_sum;
        double background_mean = background_pixel_sum / (double)background_weight;
        double foreground_mean = foreground_pixel_sum / (double)foreground_weight;
        double diff = background_mean - foreground_mean;
        double mean_diff_squared = diff * diff;
        double intra_class_variance = (double)background_weight * (double)foreground_weight * mean_diff_squared;
        if (intra_class_variance > largest_variance) { largest_variance = intra_class_variance; best_threshold = (uint8_t)t; }
    }
    return best_threshold;
}

/* image ^0.25 imageops::resize(.., FilterType::Triangle) (call: src/aruco.rs:273):
 * vertical pass into f32 first, then horizontal pass, weights normalised by their f32
 * sum, final value clamped and rounded to nearest. */
static float triangle_kernel(float x) { return fabsf(x) < 1.0f ? 1.0f - fabsf(x) : 0.0f; }

static int64_t clamp_i64(int64_t a, int64_t lo, int64_t hi) { return a < lo ? lo : (a > hi ? hi : a); }

static uint32_t resize_weights(uint32_t in_len, uint32_t out_len, uint32_t o, float *ws, uint32_t *count) {
    float ratio = (float)in_len / (float)out_len;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = 1.0f * sratio;
    float input = ((float)o + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(input - src_support);
    left = clamp_i64(left, 0, (int64_t)in_len - 1);
    int64_t right = (int64_t)ceilf(input + src_support);
    right = clamp_i64(right, left + 1, (int64_t)in_len);
    input = input - 0.5f;
    float sum = 0.0f; uint32_t n = 0;
    for (int64_t i = left; i < right; i++) {
        float wgt = triangle_kernel(((float)i - input) / sratio);
        ws[n++] = wgt; sum += wgt;
    }
    for (uint32_t i = 0; i < n; i++) ws[i] /= sum;
    *count = n;
    return (uint32_t)left;
}

void a3o_resize_triangle(const uint8_t *img, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t *out) {
    if (w == 0 || h == 0) { memset(out, 0, (size_t)nw * nh); return; }
    if (nw == w && nh == h) { memcpy(out, img, (size_t)w * h); return; }
    float *tmp = (float *)malloc((size_t)w * nh * sizeof(float));
    float *ws = (float *)malloc(((size_t)(w > h ? w : h) + 4) * sizeof(float));
    for (uint32_t oy = 0; oy < nh; oy++) {
        uint32_t cnt; uint32_t left = resize_weights(h, nh, oy, ws, &cnt);
        for (uint32_t x = 0; x < w; x++) {
            float t = 0.0f;
            for (uint32_t i = 0; i < cnt; i++) t += (float)img[(size_t)(left + i) * w + x] * ws[i];
            tmp[(size_t)oy * w + x] = t;
        }
    }
    for (uint32_t ox = 0; ox < nw; ox++) {
        uint32_t cnt; uint32_t left = resize_weights(w, nw, ox, ws, &cnt);
        for (uint32_t y = 0; y < nh; y++) {
            float t = 0.0f;
            for (uint32_t i = 0; i < cnt; i++) t += tmp[(size_t)y * w + left + i] * ws[i];
            float c = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
            out[(size_t)y * nw + ox] = (uint8_t)roundf(c);
        }
    }
    free(tmp); free(ws);
}

/* ------------------------------------------------------------------------- */
/* L2  src/aruco.rs helpers                                                  */
/* ------------------------------------------------------------------------- */

/* src/aruco.rs:168-185 */
void a3o_enforce_clockwise_corners(uint32_t *quads, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t *q = quads + 8 * i;
        int32_t dx1 = (int32_t)q[2] - (int32_t)q[0], dy1 = (int32_t)q[3] - (int32_t)q[1];
        int32_t dx2 = (int32_t)q[4] - (int32_t)q[0], dy2 = (int32_t)q[5] - (int32_t)q[1];
        if (dx1 * dy2 - dy1 * dx2 < 0) {
            uint32_t sx = q[2], sy = q[3];
            q[2] = q[6]; q[3] = q[7]; q[6] = sx; q[7] = sy;
        }
    }
}

/* src/aruco.rs:328-338 */
static float perimeter4(const uint32_t *q) {
    float p = 0.0f;
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) % 4;
        float dx = (float)q[2 * i] - (float)q[2 * j];
        float dy = (float)q[2 * i + 1] - (float)q[2 * j + 1];
        p += sqrtf((dx * dx) + (dy * dy));
    }
    return p;
}

/* src/aruco.rs:187-232; kept_index (optional) receives the surviving original indices */
size_t a3o_discard_too_near(uint32_t *quads, size_t n, float min_distance, uint32_t *kept_index) {
    if (n == 0) return 0;
    uint8_t *dead = (uint8_t *)calloc(n, 1);
    for (size_t i = 0; i + 1 < n; i++) {
        if (dead[i]) continue;
        float perimeter_i = perimeter4(quads + 8 * i);
        for (size_t j = i + 1; j < n; j++) {
            if (dead[j]) continue;
            float distance = 0.0f;
            for (int p = 0; p < 4; p++) {
                float dx = (float)quads[8 * i + 2 * p] - (float)quads[8 * j + 2 * p];
                float dy = (float)quads[8 * i + 2 * p + 1] - (float)quads[8 * j + 2 * p + 1];
                distance += sqrtf((dx * dx) + (dy * dy));
            }
            if ((distance / 4.0f) < min_distance) {
                float perimeter_j = perimeter4(quads + 8 * j);
                if (dead[i] || dead[j]) {
                    /* one of them is already going away */
                } else if (perimeter_i >= perimeter_j) dead[j] = 1;
                else dead[i] = 1;
            }
        }
    }
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (dead[i]) continue;
        if (m != i) memmove(quads + 8 * m, quads + 8 * i, 8 * sizeof(uint32_t));
        if (kept_index) kept_index[m] = (uint32_t)i;
        m++;
    }
    free(dead);
    return m;
}

/* src/aruco.rs:315-326: new[r][c] = old[c][cols-1-r] (90 degrees counter-clockwise) */
void a3o_rotate_bit_matrix(const uint8_t *in, uint32_t rows, uint32_t cols, uint8_t *out) {
    for (uint32_t r = 0; r < cols; r++)
        for (uint32_t c = 0; c < rows; c++)
            out[(size_t)r * rows + c] = in[(size_t)c * cols + (cols - 1 - r)];
}

/* src/aruco.rs:263-313 */
int a3o_homography_to_code_permutations(const uint8_t *patch, uint32_t pw, uint32_t ph, uint8_t mark_size, uint64_t codes[4]) {
    uint8_t otsu = a3o_otsu_level(patch, pw, ph);
    size_t np = (size_t)pw * ph;
    uint8_t *bin = (uint8_t *)malloc(np);
    for (size_t i = 0; i < np; i++) bin[i] = patch[i] > otsu ? 255 : 0;
    uint32_t n = mark_size;
    uint8_t *reduced = (uint8_t *)malloc((size_t)n * n);
    a3o_resize_triangle(bin, pw, ph, n, n, reduced);
    uint8_t *bits = (uint8_t *)malloc((size_t)n * n), *rot = (uint8_t *)malloc((size_t)n * n);
    for (size_t i = 0; i < (size_t)n * n; i++) bits[i] = reduced[i] > 127;
    int ok = 1;
    uint32_t end = n ? n - 1 : 0;
    for (uint32_t i = 0; i < n && ok; i++) {
        if (bits[i * n + 0] || bits[i * n + end]) ok = 0;
        else if (bits[0 * n + i] || bits[end * n + i]) ok = 0;
    }
    if (ok) {
        for (int r = 0; r < 4; r++) {
            uint64_t v = 0;
            for (uint32_t y = 1; y + 1 < n; y++)
                for (uint32_t x = 1; x + 1 < n; x++) {
                    if (bits[y * n + x]) v |= 1;
                    v = (v << 1) | (v >> 63);
                }
            v = (v >> 1) | (v << 63);
            codes[r] = v;
            a3o_rotate_bit_matrix(bits, n, n, rot);
            memcpy(bits, rot, (size_t)n * n);
        }
    }
    free(bin); free(reduced); free(bits); free(rot);
    return ok;
}

/* src/aruco.rs:52-121.  `inject` (test aid, NULL in a3o_detect): n_inject quads of 4 (x, y) pairs that REPLACE what
 * contours_to_candidates + enforce_clockwise_corners (src/aruco.rs:65-68) delivered, in the given order -- the only way to lead a
 * degenerate quad into extract_homographies' failure branch (src/aruco.rs:255-257, quirk Q4): a convex hull of four points never is. */
static int detect_impl(const a3o_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
                       const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, int keep_debug,
                       const uint32_t *inject, size_t n_inject, a3o_detection *out) {
    memset(out, 0, sizeof(*out));
    out->width = w; out->height = h;
    uint32_t minwh = w < h ? w : h;
    uint32_t min_edge_length = (uint32_t)((float)minwh * cfg->min_side_length_factor);
    float min_corner_separation = (float)minwh * cfg->min_corner_separation_factor;

    uint8_t *grey = (uint8_t *)malloc((size_t)w * h);
    uint8_t *thr = (uint8_t *)malloc((size_t)w * h);
    if (!grey || !thr) return -1;
    a3o_to_luma8(px, fmt, w, h, row_stride, grey);
    a3o_adaptive_threshold(grey, w, h, cfg->threshold_window, thr);
    a3o_contours cs;
    if (a3o_find_contours(thr, w, h, &cs)) return -1;
    out->n_contours = cs.n_contours;
    out->n_contour_points = cs.offsets[cs.n_contours];

    /* contours_to_candidates, src/aruco.rs:124-166 */
    u32vec cand = {0}, cand_start = {0};
    size_t maxlen = 1;
    for (uint32_t c = 0; c < cs.n_contours; c++) { size_t l = cs.offsets[c + 1] - cs.offsets[c]; if (l > maxlen) maxlen = l; }
    uint32_t *edges = (uint32_t *)malloc(2 * (maxlen + 2) * sizeof(uint32_t));
    for (uint32_t c = 0; c < cs.n_contours; c++) {
        const uint32_t *p = cs.points + 2 * (size_t)cs.offsets[c];
        size_t len = cs.offsets[c + 1] - cs.offsets[c];
        size_t ne = a3o_approximate_polygon_dp(p, len, (double)len * cfg->contour_simplification_epsilon, 1, edges);
        if (ne != 4) { out->stat_reject_point_count++; continue; }
        uint32_t hull[10];
        size_t nh = a3o_convex_hull(edges, 4, hull);
        if (nh != 4) { out->stat_reject_convexity++; continue; }
        uint32_t cmin = min_edge_length + 1;
        for (int i = 0; i < 4; i++) {
            int j = (i + 1) % 4;
            int32_t dx = (int32_t)hull[2 * i] - (int32_t)hull[2 * j];
            int32_t dy = (int32_t)hull[2 * i + 1] - (int32_t)hull[2 * j + 1];
            uint32_t d2 = (uint32_t)((dx * dx) + (dy * dy));
            if (d2 < cmin) cmin = d2;
        }
        if (cmin < min_edge_length) { out->stat_reject_edge_length++; continue; }
        for (int i = 0; i < 8; i++) u32vec_push(&cand, hull[i]);
        u32vec_push(&cand_start, p[1] * w + p[0]);
    }
    free(edges);
    size_t nc = cand.n / 8;
    a3o_enforce_clockwise_corners(cand.v, nc);
    if (inject) {
        cand.n = 0; cand_start.n = 0;
        for (size_t i = 0; i < n_inject; i++) {
            for (int k = 0; k < 8; k++) u32vec_push(&cand, inject[8 * i + k]);
            u32vec_push(&cand_start, (uint32_t)i);
        }
        nc = n_inject;
    }
    out->n_candidates_pre = (uint32_t)nc;
    if (keep_debug) {
        out->candidates_pre = (uint32_t *)malloc((nc ? nc : 1) * 8 * sizeof(uint32_t));
        if (nc) memcpy(out->candidates_pre, cand.v, nc * 8 * sizeof(uint32_t));
        out->candidates_pre_start = (uint32_t *)malloc((nc ? nc : 1) * sizeof(uint32_t));
        if (nc) memcpy(out->candidates_pre_start, cand_start.v, nc * sizeof(uint32_t));
    }
    nc = a3o_discard_too_near(cand.v, nc, min_corner_separation, NULL);
    out->n_candidates = (uint32_t)nc;
    out->candidates = (uint32_t *)malloc((nc ? nc : 1) * 8 * sizeof(uint32_t));
    if (nc) memcpy(out->candidates, cand.v, nc * 8 * sizeof(uint32_t));

    /* extract_homographies, src/aruco.rs:234-261 */
    uint32_t S = cfg->homography_sample_size;
    out->sample = S;
    out->homographies = (uint8_t *)calloc((nc ? nc : 1) * (size_t)S * S, 1);
    out->homography_ok = (uint8_t *)calloc(nc ? nc : 1, 1);
    out->decode_ok = (int32_t *)calloc(nc ? nc : 1, sizeof(int32_t));
    out->codes = (uint64_t *)calloc((nc ? nc : 1) * 4, sizeof(uint64_t));
    out->markers = (a3o_marker *)calloc(nc ? nc : 1, sizeof(a3o_marker));
    uint8_t mark_size = a3o_mark_size(num_bits);
    float hs = (float)S;
    for (size_t k = 0; k < nc; k++) {
        const uint32_t *q = out->candidates + 8 * k;
        float from[8], to[8] = { 0.0f, 0.0f, hs, 0.0f, hs, hs, 0.0f, hs };
        for (int i = 0; i < 8; i++) from[i] = (float)q[i];
        float transform[9], inverse[9];
        uint8_t *patch = out->homographies + k * (size_t)S * S;
        uint32_t pw = S, ph = S;
        if (a3o_from_control_points(from, to, transform, inverse)) {
            a3o_warp_into(grey, w, h, inverse, patch, S, S);
            out->homography_ok[k] = 1;
        } else { pw = 1; ph = 1; } /* GrayImage::new(1, 1): one zero pixel (quirk Q4) */

        /* marker accept, src/aruco.rs:75-113 */
        uint64_t perm[4] = {0, 0, 0, 0};
        int have = a3o_homography_to_code_permutations(patch, pw, ph, mark_size, perm);
        out->decode_ok[k] = have;
        int found_any = 0;
        uint32_t min_code_distance = 0x7FFFFFFF;
        uint64_t min_code = 0x7FFFFFFF;
        size_t min_code_id = 0x7FFFFFFF;
        uint32_t min_rotation = 0;
        if (have) {
            memcpy(out->codes + 4 * k, perm, sizeof perm);
            for (uint32_t r = 0; r < 4; r++) {
                size_t id; uint8_t dist;
                a3o_find_nearest(codes, n_codes, perm[r], &id, &dist);
                if ((uint32_t)dist < min_code_distance) {
                    min_code = perm[r]; min_code_distance = dist; min_code_id = id; min_rotation = r; found_any = 1;
                }
            }
        }
        if (found_any && (!cfg->filter_high_bit_errors || min_code_distance < (uint32_t)tau)) {
            a3o_marker *m = &out->markers[out->n_markers++];
            m->id = (uint32_t)min_code_id;
            m->code = min_code;
            m->rotation = min_rotation;
            m->hamming_distance = (uint8_t)min_code_distance;
            m->candidate_index = (uint32_t)k;
            for (int i = 0; i < 4; i++) { /* corners.rotate_left(min_rotation) */
                int s = (i + (int)min_rotation) % 4;
                m->corners[2 * i] = q[2 * s]; m->corners[2 * i + 1] = q[2 * s + 1];
            }
        }
    }
    free(cand.v); free(cand_start.v);
    a3o_free_contours(&cs);
    if (keep_debug) { out->grey = grey; out->thresholded = thr; }
    else { free(grey); free(thr); }
    return 0;
}

int a3o_detect(const a3o_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
               const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, int keep_debug,
               a3o_detection *out) {
    return detect_impl(cfg, codes, n_codes, num_bits, tau, px, fmt, w, h, row_stride, keep_debug, NULL, 0, out);
}

/* a3o_detect with the candidate list handed in (see detect_impl) */
int a3o_detect_quads(const a3o_config *cfg, const uint64_t *codes, size_t n_codes, uint8_t num_bits, uint8_t tau,
                     const uint8_t *px, int fmt, uint32_t w, uint32_t h, size_t row_stride, int keep_debug,
                     const uint32_t *quads, size_t n_quads, a3o_detection *out) {
    static const uint32_t none[8] = {0};
    return detect_impl(cfg, codes, n_codes, num_bits, tau, px, fmt, w, h, row_stride, keep_debug, quads ? quads : none, n_quads, out);
}

void a3o_free_detection(a3o_detection *d) {
    free(d->grey); free(d->thresholded); free(d->candidates_pre); free(d->candidates_pre_start);
    free(d->candidates); free(d->homographies); free(d->homography_ok); free(d->decode_ok); free(d->codes); free(d->markers);
    memset(d, 0, sizeof(*d));
}

/* ------------------------------------------------------------------------- */
/* L3  pose (src/pose.rs, src/pinhole.rs:88-93); matrices row-major m[r*3+c]  */
/* ------------------------------------------------------------------------- */

/* src/pose.rs:85-93 */
void a3o_make_marker_square(float s, float o[12]) {
    float hw = 0.5f * s;
    float v[12] = { -hw, hw, 0.0f, hw, hw, 0.0f, hw, -hw, 0.0f, -hw, -hw, 0.0f };
    memcpy(o, v, sizeof v);
}

/* src/pose.rs:96-123 */
void a3o_compute_homography_from_marker_square(float marker_size_mm, const float t[8], float h[9]) {
    float p1x = -t[0], p1y = -t[1], p2x = -t[2], p2y = -t[3], p3x = -t[4], p3y = -t[5], p4x = -t[6], p4y = -t[7];
    float half_width = marker_size_mm / 2.0f;
    float det_inv = -1.0f / (half_width * (p1x * p2y - p2x * p1y - p1x * p4y + p2x * p3y - p3x * p2y + p4x * p1y + p3x * p4y - p4x * p3y));
    h[0] = det_inv * (p1x * p3x * p2y - p2x * p3x * p1y - p1x * p4x * p2y + p2x * p4x * p1y - p1x * p3x * p4y + p1x * p4x * p3y + p2x * p3x * p4y - p2x * p4x * p3y);
    h[1] = det_inv * (p1x * p2x * p3y - p1x * p3x * p2y - p1x * p2x * p4y + p2x * p4x * p1y + p1x * p3x * p4y - p3x * p4x * p1y - p2x * p4x * p3y + p3x * p4x * p2y);
    h[2] = det_inv * half_width * (p1x * p2x * p3y - p2x * p3x * p1y - p1x * p2x * p4y + p1x * p4x * p2y - p1x * p4x * p3y + p3x * p4x * p1y + p2x * p3x * p4y - p3x * p4x * p2y);
    h[3] = det_inv * (p1x * p2y * p3y - p2x * p1y * p3y - p1x * p2y * p4y + p2x * p1y * p4y - p3x * p1y * p4y + p4x * p1y * p3y + p3x * p2y * p4y - p4x * p2y * p3y);
    h[4] = det_inv * (p2x * p1y * p3y - p3x * p1y * p2y - p1x * p2y * p4y + p4x * p1y * p2y + p1x * p3y * p4y - p4x * p1y * p3y - p2x * p3y * p4y + p3x * p2y * p4y);
    h[5] = det_inv * half_width * (p1x * p2y * p3y - p3x * p1y * p2y - p2x * p1y * p4y + p4x * p1y * p2y - p1x * p3y * p4y + p3x * p1y * p4y + p2x * p3y * p4y - p4x * p2y * p3y);
    h[6] = -det_inv * (p1x * p3y - p3x * p1y - p1x * p4y - p2x * p3y + p3x * p2y + p4x * p1y + p2x * p4y - p4x * p2y);
    h[7] = det_inv * (p1x * p2y - p2x * p1y - p1x * p3y + p3x * p1y + p2x * p4y - p4x * p2y - p3x * p4y + p4x * p3y);
    h[8] = 1.0f;
}

/* src/pose.rs:238-267 */
static void find_rotation_to_z(const float v[3], float rot[9]) {
    memset(rot, 0, 9 * sizeof(float));
    float a = v[0] * v[0], b = v[1] * v[1], c = v[2] * v[2];
    float n = sqrtf(a + b + c);
    float ax = v[0] / n, ay = v[1] / n, az = v[2] / n;
    if (fabsf(1.0f + az) < 1e-6f) {
        rot[0] = 1.0f; rot[4] = 1.0f; rot[8] = -1.0f;
    } else {
        float d = 1.0f / (1.0f + az);
        float ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
        rot[0] = -ax2 * d + 1.0f; rot[1] = -axay * d;       rot[2] = -ax;
        rot[3] = -axay * d;       rot[4] = -ay2 * d + 1.0f; rot[5] = -ay;
        rot[6] = ax;              rot[7] = ay;              rot[8] = 1.0f - (ax2 + ay2) * d;
    }
}

/* src/pose.rs:158-235 */
static void compute_rotations(const float j[4], float tx, float ty, float r1[9], float r2[9]) {
    float t[3] = { tx, ty, 1.0f };
    float rz[9], rv[9];
    find_rotation_to_z(t, rz);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rv[r * 3 + c] = rz[c * 3 + r];
#define RV(r, c) rv[((r) - 1) * 3 + ((c) - 1)]
    float b00 = RV(1, 1) - tx * RV(3, 1);
    float b01 = RV(1, 2) - tx * RV(3, 2);
    float b10 = RV(2, 1) - ty * RV(3, 1);
    float b11 = RV(2, 2) - ty * RV(3, 2);
    float inv_det = 1.0f / (b00 * b11 - b01 * b10);
    float binv00 = inv_det * b11, binv01 = -inv_det * b01, binv10 = -inv_det * b10, binv11 = inv_det * b00;
    float a00 = binv00 * j[0] + binv01 * j[2];
    float a01 = binv00 * j[1] + binv01 * j[3];
    float a10 = binv10 * j[0] + binv11 * j[2];
    float a11 = binv10 * j[1] + binv11 * j[3];
    float ata00 = a00 * a00 + a01 * a01;
    float ata01 = a00 * a10 + a01 * a11;
    float ata11 = a10 * a10 + a11 * a11;
    float gamma = sqrtf(0.5f * (ata00 + ata11 + sqrtf((ata00 - ata11) * (ata00 - ata11) + 4.0f * ata01 * ata01)));
    float rt00 = a00 / gamma, rt01 = a01 / gamma, rt10 = a10 / gamma, rt11 = a11 / gamma;
    float rt00_2 = rt00 * rt00, rt01_2 = rt01 * rt01, rt10_2 = rt10 * rt10, rt11_2 = rt11 * rt11;
    float b0 = sqrtf(-rt00_2 - rt10_2 + 1.0f);
    float b1 = sqrtf(-rt01_2 - rt11_2 + 1.0f);
    float sp = -rt00 * rt01 - rt10 * rt11;
    if (sp < 0.0f) b1 = -b1;
    for (int r = 1; r <= 3; r++) {
        r1[(r - 1) * 3 + 0] = (rt00) * RV(r, 1) + (rt10) * RV(r, 2) + (b0) * RV(r, 3);
        r1[(r - 1) * 3 + 1] = (rt01) * RV(r, 1) + (rt11) * RV(r, 2) + (b1) * RV(r, 3);
        r1[(r - 1) * 3 + 2] = (b1 * rt10 - b0 * rt11) * RV(r, 1) + (b0 * rt01 - b1 * rt00) * RV(r, 2) + (rt00 * rt11 - rt01 * rt10) * RV(r, 3);
        r2[(r - 1) * 3 + 0] = (rt00) * RV(r, 1) + (rt10) * RV(r, 2) + (-b0) * RV(r, 3);
        r2[(r - 1) * 3 + 1] = (rt01) * RV(r, 1) + (rt11) * RV(r, 2) + (-b1) * RV(r, 3);
        r2[(r - 1) * 3 + 2] = (b0 * rt11 - b1 * rt10) * RV(r, 1) + (b1 * rt00 - b0 * rt01) * RV(r, 2) + (rt00 * rt11 - rt01 * rt10) * RV(r, 3);
    }
#undef RV
}

/* src/pose.rs:269-335 */
static void compute_translation(const float obj[12], const float pts[8], const float rot[9], float t[3]) {
    float m11 = 4.0f, m13 = 0.0f, m22 = 4.0f, m23 = 0.0f, m31 = 0.0f, m32 = 0.0f, m33 = 0.0f;
    float atb0 = 0.0f, atb1 = 0.0f, atb2 = 0.0f;
    for (int i = 0; i < 4; i++) {
        float ox = obj[3 * i], oy = obj[3 * i + 1];
        float rx = rot[0] * ox + rot[1] * oy;
        float ry = rot[3] * ox + rot[4] * oy;
        float rz = rot[6] * ox + rot[7] * oy;
        float a2 = -pts[2 * i], b2 = -pts[2 * i + 1];
        m13 += a2; m23 += b2; m31 += a2; m32 += b2;
        m33 += a2 * a2 + b2 * b2;
        float bx = -a2 * rz - rx;
        float by = -b2 * rz - ry;
        atb0 += bx; atb1 += by;
        atb2 += a2 * bx + b2 * by;
    }
    float det_a_inv = 1.0f / (m11 * m22 * m33 - m11 * m23 * m32 - m13 * m22 * m31);
    float s11 = m22 * m33 - m23 * m32, s12 = m13 * m32, s13 = -m13 * m22;
    float s21 = m23 * m31, s22 = m11 * m33 - m13 * m31, s23 = -m11 * m23;
    float s31 = -m22 * m31, s32 = -m11 * m32, s33 = m11 * m22;
    t[0] = det_a_inv * (s11 * atb0 + s12 * atb1 + s13 * atb2);
    t[1] = det_a_inv * (s21 * atb0 + s22 * atb1 + s23 * atb2);
    t[2] = det_a_inv * (s31 * atb0 + s32 * atb1 + s33 * atb2);
}

/* src/pose.rs:24-28 (rotation * p + translation) */
void a3o_apply_transform(const a3o_pose *p, const float *pts, size_t n, float *out) {
    const float *r = p->rotation;
    for (size_t i = 0; i < n; i++) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        out[3 * i]     = (r[0] * x + r[1] * y + r[2] * z) + p->translation[0];
        out[3 * i + 1] = (r[3] * x + r[4] * y + r[5] * z) + p->translation[1];
        out[3 * i + 2] = (r[6] * x + r[7] * y + r[8] * z) + p->translation[2];
    }
}

/* src/pose.rs:35-39 (rotation^T * (p - translation)) */
void a3o_apply_inverse_transform(const a3o_pose *p, const float *pts, size_t n, float *out) {
    const float *r = p->rotation;
    for (size_t i = 0; i < n; i++) {
        float x = pts[3 * i] - p->translation[0], y = pts[3 * i + 1] - p->translation[1], z = pts[3 * i + 2] - p->translation[2];
        out[3 * i]     = r[0] * x + r[3] * y + r[6] * z;
        out[3 * i + 1] = r[1] * x + r[4] * y + r[7] * z;
        out[3 * i + 2] = r[2] * x + r[5] * y + r[8] * z;
    }
}

/* src/pose.rs:337-348 */
static float compute_reprojection_error(const a3o_pose *p, const float obj[12], const float pts[8]) {
    float proj[12];
    a3o_apply_transform(p, obj, 4, proj);
    float error = 0.0f;
    for (int i = 0; i < 4; i++) {
        float z = proj[3 * i + 2] > 1e-5f ? proj[3 * i + 2] : 1e-5f;
        float dx = (proj[3 * i] / z) - pts[2 * i];
        float dy = (proj[3 * i + 1] / z) - pts[2 * i + 1];
        error += sqrtf(dx * dx + dy * dy);
    }
    return error;
}

/* src/pose.rs:130-156 */
void a3o_solve_canonical_form(const float obj[12], const float pts[8], const float h[9], a3o_pose *p1, a3o_pose *p2) {
    float j[4] = { h[0] - h[6] * h[2], h[1] - h[7] * h[2], h[3] - h[6] * h[5], h[4] - h[7] * h[5] };
    compute_rotations(j, h[2], h[5], p1->rotation, p2->rotation);
    compute_translation(obj, pts, p1->rotation, p1->translation);
    compute_translation(obj, pts, p2->rotation, p2->translation);
    p1->error = compute_reprojection_error(p1, obj, pts);
    p2->error = compute_reprojection_error(p2, obj, pts);
}

/* src/pose.rs:64-81 */
void a3o_solve_with_normalized_points(const float pts[8], float marker_size_mm, a3o_pose *o1, a3o_pose *o2) {
    float obj[12], h[9];
    a3o_pose p1, p2;
    a3o_make_marker_square(marker_size_mm, obj);
    a3o_compute_homography_from_marker_square(marker_size_mm, pts, h);
    a3o_solve_canonical_form(obj, pts, h, &p1, &p2);
    if (p1.error < p2.error) { *o1 = p1; *o2 = p2; } else { *o1 = p2; *o2 = p1; }
}

/* src/pose.rs:59-62 */
void a3o_solve_with_undistorted_points(const uint32_t c[8], float marker_size_mm, uint32_t iw, uint32_t ih, a3o_pose *p1, a3o_pose *p2) {
    float pts[8];
    for (int i = 0; i < 4; i++) { pts[2 * i] = (float)c[2 * i] / (float)iw; pts[2 * i + 1] = (float)c[2 * i + 1] / (float)ih; }
    a3o_solve_with_normalized_points(pts, marker_size_mm, p1, p2);
}

/* src/pose.rs:52-55 + src/pinhole.rs:88-93 */
void a3o_solve_with_intrinsics(const uint32_t c[8], float marker_size_mm, float fx, float fy, float cx, float cy, a3o_pose *p1, a3o_pose *p2) {
    float pts[8];
    for (int i = 0; i < 4; i++) { pts[2 * i] = ((float)c[2 * i] - cx) / fx; pts[2 * i + 1] = ((float)c[2 * i + 1] - cy) / fy; }
    a3o_solve_with_normalized_points(pts, marker_size_mm, p1, p2);
}
