// k_decode.hip -- K5..K7: candidate ordering + discard_too_near, homography warp + Otsu +
// triangle resize + bit decode + dictionary lookup + marker accept, IPPE pose, dictionary utilities.
//
// Replaces src/aruco.rs:69-113 (discard_too_near, extract_homographies,
// homography_to_code_permutations, the find_nearest loop and the accept test),
// src/dictionaries.rs:129-138,160-196 and src/pose.rs:52-348.
//
// Floating point follows the reference operation by operation (f64 for the 8x8 solve and
// Otsu, f32 elsewhere); the library is built with -ffp-contract=off so that no a*b+c is
// fused, exactly like the Rust reference and the CPU oracle.
#include <cstdlib>

#include "a3_common.h"

namespace a3 {

// ---------------------------------------------------------------------------------------
// per frame: order candidates as the reference found them, then discard_too_near
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float perimeter4(const uint16_t* q) {  // src/aruco.rs:328-338
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = (i + 1) & 3;
        const float dx = (float)q[2 * i] - (float)q[2 * j];
        const float dy = (float)q[2 * i + 1] - (float)q[2 * j + 1];
        p += sqrtf((dx * dx) + (dy * dy));
    }
    return p;
}

struct __attribute__((aligned(8))) ProjRec { float inv[9]; int ok; };
__device__ bool solve_projection(const float* from, float S, float* inv_out);

// one wave per frame.  cands: unordered CandRec[max_cand] per frame (as k_contour_quads appended them).
// proj != nullptr: the wave also solves the projection of every surviving candidate (one lane each: the 8x8 system needs
// ~200 VGPRs, affordable in a one-wave workgroup) into proj[work index], which saves the separate k_projection launch.
// BIG (tables beyond kFrameCandLds slots per frame -- a frame tiled with thousands of small squares; no camera frame gets there): the
// sorted quads live where they end up anyway (pre_xy), keys and perimeters in a scratch plane of the caller's, and only the dead flags
// in LDS (a byte per slot: 64 KB at the format's limit of 65 536 -- a3_marker.candidate_index is 16 bits).  Same walk, same order,
// every LDS access of the small form a trip to the L2 instead: correct, not fast (0.1 s for 7 000 quads, all of them far apart).
constexpr uint32_t kFrameCandLds = 6144;   // 21 bytes of LDS per slot in the small form: 129 KB of the CU's 160
#ifndef A3_FC_WAVES
#define A3_FC_WAVES 1
#endif
template <bool BIG>
__global__ __launch_bounds__(64, A3_FC_WAVES) void k_frame_candidates(const CandRec* __restrict__ cands, const uint32_t* __restrict__ cand_count,
                                                         uint32_t max_cand, float min_distance, uint16_t* pre_xy,
                                                         uint16_t* __restrict__ fin_xy, uint32_t* __restrict__ fin_count,
                                                         uint32_t* __restrict__ work, unsigned int* __restrict__ work_count,
                                                         uint32_t S, ProjRec* __restrict__ proj, float* big_scratch /* BIG: frames x max_cand */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t* s_xy = BIG ? pre_xy + (size_t)blockIdx.x * max_cand * 8 : reinterpret_cast<uint16_t*>(smem);                      // max_cand * 8
    float* s_per = BIG ? big_scratch + (size_t)blockIdx.x * max_cand : reinterpret_cast<float*>(smem + (size_t)max_cand * 16);   // max_cand
    uint8_t* s_dead = BIG ? smem : smem + (size_t)max_cand * 20;                                                                // max_cand
    // (BIG: what one lane wrote to memory another reads after the barrier -- the fence makes the wave's stores visible and drops stale lines)
#define FC_SYNC() do { if (BIG) __threadfence(); __syncthreads(); } while (0)
    const uint32_t f = blockIdx.x;
    const int lane = threadIdx.x;
    // a frame that overflowed its table: k_contour_quads has flagged it, the host re-runs the batch with larger tables and nothing of
    // this pass is used -- no point in ordering the slots that did fit (seconds, at the large end of the through-memory form)
    const uint32_t c = cand_count[f] > max_cand ? 0u : cand_count[f];
    const CandRec* src = cands + (size_t)f * max_cand;
    // rank sort by start key (keys are unique: one border starts per pixel visit).  The keys go to LDS first: ranking
    // straight from global memory was a chain of c dependent loads per lane (s_per doubles as the key buffer until the
    // perimeters are written).
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_per);
    for (uint32_t i = lane; i < c; i += 64) s_key[i] = src[i].start_key;
    FC_SYNC();
    for (uint32_t i = lane; i < c; i += 64) {
        const uint32_t key = s_key[i];
        const CandRec r = src[i];   // in flight while the rank is counted
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; j++) rank += s_key[j] < key;
        for (int k = 0; k < 8; k++) s_xy[rank * 8 + k] = r.xy[k];
    }
    FC_SYNC();
    for (uint32_t i = lane; i < c; i += 64) {
        s_per[i] = perimeter4(&s_xy[i * 8]);
        s_dead[i] = 0;
        if (!BIG) for (int k = 0; k < 8; k++) pre_xy[((size_t)f * max_cand + i) * 8 + k] = s_xy[i * 8 + k];
    }
    FC_SYNC();
    // discard_too_near, src/aruco.rs:187-232: i ascending; for j > i ascending, a close pair kills the smaller
    // perimeter; once i itself is dead the rest of its row is a no-op.
    if (c >= 2 && c <= 64) {
        // The usual case (a frame holds a few dozen quads): the same walk with everything in registers -- lane j holds quad j,
        // row i's quad comes through v_readlane (i is uniform), the dead set is a 64-bit mask every lane carries -- so a row is
        // four square roots, a division and two ballots, without LDS traffic or a barrier.  (s_memtime stamps: the LDS version
        // below spent 850 cycles per live row, half of the kernel's 20 us on BASELINE config 2.)
        const uint32_t j = min((uint32_t)lane, c - 1u);
        float xj[8];
#pragma unroll
        for (int k = 0; k < 8; k++) xj[k] = (float)s_xy[j * 8 + k];
        const float per_j = s_per[j];
        unsigned long long dead = 0;
        for (uint32_t i = 0; i + 1 < c; i++) {
            if ((dead >> i) & 1ull) continue;   // uniform
            float xi[8];
#pragma unroll
            for (int k = 0; k < 8; k++) xi[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xj[k]), (int)i));
            const float per_i = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(per_j), (int)i));
            float distance = 0.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float dx = xi[2 * p] - xj[2 * p];
                const float dy = xi[2 * p + 1] - xj[2 * p + 1];
                distance += sqrtf((dx * dx) + (dy * dy));
            }
            const bool cand = (uint32_t)lane > i && (uint32_t)lane < c && !((dead >> lane) & 1ull);
            const bool close_alive = cand && (distance / 4.0f) < min_distance;
            const bool bigger = close_alive && !(per_i >= per_j);
            const unsigned long long m_close = __ballot(close_alive), m_big = __ballot(bigger);
            unsigned long long kill = m_close;
            if (m_big) {
                kill = m_close & ((1ull << (__ffsll((long long)m_big) - 1)) - 1ull);
                dead |= 1ull << i;
            }
            dead |= kill;
        }
        if ((uint32_t)lane < c) s_dead[lane] = (uint8_t)((dead >> lane) & 1ull);
    } else
    for (uint32_t i = 0; i + 1 < c; i++) {
        if (s_dead[i]) continue;  // uniform: LDS value
        const float per_i = s_per[i];
        bool i_dead = false;
        for (uint32_t j0 = i + 1; j0 < c && !i_dead; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool close_alive = false, bigger = false;
            if (j < c && !s_dead[j]) {
                float distance = 0.0f;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float dx = (float)s_xy[i * 8 + 2 * p] - (float)s_xy[j * 8 + 2 * p];
                    const float dy = (float)s_xy[i * 8 + 2 * p + 1] - (float)s_xy[j * 8 + 2 * p + 1];
                    distance += sqrtf((dx * dx) + (dy * dy));
                }
                close_alive = (distance / 4.0f) < min_distance;
                bigger = close_alive && !(per_i >= s_per[j]);
            }
            const unsigned long long m_close = __ballot(close_alive), m_big = __ballot(bigger);
            unsigned long long kill = m_close;
            if (m_big) {
                const int first = __ffsll((long long)m_big) - 1;
                kill = m_close & ((1ull << first) - 1ull);
                i_dead = true;
            }
            if ((kill >> lane) & 1ull) s_dead[j] = 1;
        }
        if (i_dead && lane == 0) s_dead[i] = 1;
        __syncthreads();   // (only s_dead changes inside this loop, and s_dead lives in LDS in both forms: no memory fence per row)
    }
    FC_SYNC();
    // survivors, order preserved: count them, take a range of the work list, then write quads, work items and projections
    uint32_t total = 0;
    for (uint32_t i0 = 0; i0 < c; i0 += 64) {
        const uint32_t i = i0 + lane;
        total += (uint32_t)__popcll(__ballot(i < c && !s_dead[i]));
    }
    uint32_t w0 = 0;
    if (lane == 0) {
        fin_count[f] = total;
        if (total) w0 = atomicAdd(work_count, total);
    }
    w0 = __shfl(w0, 0);
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < c; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool alive = i < c && !s_dead[i];
        const unsigned long long m = __ballot(alive);
        if (alive) {
            const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            float from[8];
            for (int k = 0; k < 8; k++) { const uint16_t v = s_xy[i * 8 + k]; fin_xy[((size_t)f * max_cand + pos) * 8 + k] = v; from[k] = (float)v; }
            work[w0 + pos] = f * max_cand + pos;
            if (proj) {
                ProjRec r;
                r.ok = solve_projection(from, (float)S, r.inv) ? 1 : 0;
                proj[w0 + pos] = r;
            }
        }
        base += __popcll(m);
    }
#undef FC_SYNC
}

// ---------------------------------------------------------------------------------------
// decode one candidate per workgroup
// ---------------------------------------------------------------------------------------
struct DecodeOut {      // one per final candidate slot
    uint64_t code;
    uint64_t codes[4];
    uint32_t id;
    uint8_t valid, rotation, hamming, hom_ok;
    int32_t decode_ok;
    uint32_t patch;   // debug taps: index of this candidate's warped patch in the patch buffer, kNone if it was not kept
};

// imageproc Projection::from_control_points: 8x8 system in f64, LU with partial pivoting (nalgebra order of
// operations), cast to f32, adjugate inverse normalised by its last element.  Single lane.
__device__ bool solve_projection(const float* from, float S, float* inv_out) {
    double A[8][8], b[8];
    const float to[8] = {0.0f, 0.0f, S, 0.0f, S, S, 0.0f, S};
    for (int i = 0; i < 4; i++) {
        const double xf = from[2 * i], yf = from[2 * i + 1], x = to[2 * i], y = to[2 * i + 1];
        A[2 * i][0] = 0.0; A[2 * i][1] = 0.0; A[2 * i][2] = 0.0; A[2 * i][3] = -xf; A[2 * i][4] = -yf; A[2 * i][5] = -1.0;
        A[2 * i][6] = y * xf; A[2 * i][7] = y * yf;
        A[2 * i + 1][0] = xf; A[2 * i + 1][1] = yf; A[2 * i + 1][2] = 1.0; A[2 * i + 1][3] = 0.0; A[2 * i + 1][4] = 0.0; A[2 * i + 1][5] = 0.0;
        A[2 * i + 1][6] = -x * xf; A[2 * i + 1][7] = -x * yf;
        b[2 * i] = -y; b[2 * i + 1] = x;
    }
    // Every index below is a compile-time constant after unrolling (the pivot row is exchanged by predicated moves over the
    // candidate rows, b follows its rows instead of replaying the permutation afterwards): A and b live in registers.  With
    // A[piv][c] indexed at run time they lived in scratch memory and this kernel took 24 us for 2.5 k candidates.
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int piv = i; double best = fabs(A[i][i]);
#pragma unroll
        for (int r = i + 1; r < 8; r++) { const double v = fabs(A[r][i]); if (v > best) { best = v; piv = r; } }
        double diag = A[i][i];
#pragma unroll
        for (int r = i + 1; r < 8; r++) diag = piv == r ? A[r][i] : diag;
        if (diag == 0.0) continue;
#pragma unroll
        for (int r = i + 1; r < 8; r++) {
            const bool sw = piv == r;
#pragma unroll
            for (int c = 0; c < 8; c++) { const double x = A[i][c], y = A[r][c]; A[i][c] = sw ? y : x; A[r][c] = sw ? x : y; }
            const double x = b[i], y = b[r]; b[i] = sw ? y : x; b[r] = sw ? x : y;
        }
        const double inv_diag = 1.0 / diag;
#pragma unroll
        for (int r = i + 1; r < 8; r++) A[r][i] *= inv_diag;
#pragma unroll
        for (int c = i + 1; c < 8; c++) {
            const double pr = -A[i][c];
#pragma unroll
            for (int r = i + 1; r < 8; r++) A[r][c] = pr * A[r][i] + A[r][c];
        }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double coeff = -b[i];
#pragma unroll
        for (int r = i + 1; r < 8; r++) b[r] = coeff * A[r][i] + b[r];
    }
    bool singular = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const double diag = A[i][i];
        if (diag == 0.0) singular = true;
        const double coeff = b[i] / diag;
        b[i] = coeff;
        const double nc = -coeff;
#pragma unroll
        for (int r = 0; r < i; r++) b[r] = nc * A[r][i] + b[r];
    }
    if (singular) return false;
    float t[9];
    for (int i = 0; i < 8; i++) t[i] = (float)b[i];
    t[8] = 1.0f;
    const float t00 = t[0], t01 = t[1], t02 = t[2], t10 = t[3], t11 = t[4], t12 = t[5], t20 = t[6], t21 = t[7], t22 = t[8];
    const float m00 = t11 * t22 - t12 * t21;
    const float m01 = t10 * t22 - t12 * t20;
    const float m02 = t10 * t21 - t11 * t20;
    const float det = t00 * m00 - t01 * m01 + t02 * m02;
    if (fabsf(det) < 1e-10f) return false;
    const float m10 = t01 * t22 - t02 * t21;
    const float m11 = t00 * t22 - t02 * t20;
    const float m12 = t00 * t21 - t01 * t20;
    const float m20 = t01 * t12 - t02 * t11;
    const float m21 = t00 * t12 - t02 * t10;
    const float m22 = t00 * t11 - t01 * t10;
    const float r[9] = {m00 / det, -m10 / det, m20 / det, -m01 / det, m11 / det, -m21 / det, m02 / det, -m12 / det, m22 / det};
    for (int i = 0; i < 8; i++) inv_out[i] = r[i] / r[8];
    inv_out[8] = 1.0f;
    return true;
}

// v_cvt_u32_f32 truncates, saturates (negative -> 0, >= 2^32 -> 0xFFFFFFFF) and turns NaN into 0: Rust's `as u32` in one
// instruction (written as asm because a C++ cast is undefined outside the range, i.e. free to be anything after optimisation).
__device__ __forceinline__ uint32_t sat_u32(float x) {
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// imageproc's Clamp for u8: x < 255 ? (x > 0 ? x as u8 : 0) : 255 -- so NaN gives 255.  Branch-free: the sampling loop runs
// this three times per sample, and the nested ifs became three exec-mask regions each.
__device__ __forceinline__ uint8_t clamp_u8(float x) {
    const uint32_t r = min(sat_u32(x), 255u);
    return (uint8_t)(x != x ? 255u : r);
}

// grey level of pixel x in a frame row: the plane K1 wrote, or into_luma8 of the caller's pixel (same integers)
__device__ __forceinline__ float grey_tap(const uint8_t* __restrict__ row, uint32_t x, int fmt) {
    if (fmt == A3_FMT_L8 || fmt == kFmtGreyPlane) return (float)row[x];
    if (fmt == A3_FMT_RGB8) { const uint8_t* p = row + 3u * (size_t)x; return (float)luma_of(p[0], p[1], p[2]); }
    const uint8_t* p = row + 4u * (size_t)x;
    return (float)(fmt == A3_FMT_BGRA8 ? luma_of(p[2], p[1], p[0]) : luma_of(p[0], p[1], p[2]));
}

// grey of one pixel in the low bytes of px (byte 0 = R, or B when bgr): byte-wise dot products with the split weights
// 2126 = 8*256+78, 7152 = 27*256+240, 722 = 2*256+210 and the exact /10000 (same integers as luma_of, as in K1)
__device__ __forceinline__ uint32_t luma_px(uint32_t px, bool bgr) {
    const uint32_t lo = __builtin_amdgcn_udot4(px, bgr ? 0x004EF0D2u : 0x00D2F04Eu, 0u, false);
    const uint32_t hi = __builtin_amdgcn_udot4(px, bgr ? 0x00081B02u : 0x00021B08u, 0u, false);
    const uint32_t l = lo + (hi << 8);   // <= 10000 * 255 < 2^22
    __builtin_assume(l < (1u << 22));     // both factors below 2^24: the full-rate v_mul_hi_u32_u24 instead of the quarter-rate v_mul_hi_u32
    return (uint32_t)(((uint64_t)l * 13743896ull) >> 37);
}

// imageproc interpolate_bilinear: default (0) outside, the two horizontal lerps truncated to u8 first.
// Split in two so that a lane can have the loads of several samples in flight before it converts any of them:
// issue() decides the case and starts the two 12-byte row reads, finish() turns them into the sample.
// The reads are UNCONDITIONAL -- a sample that falls outside the frame, or so close to its end that a 12-byte read would run
// past it, reads the frame's first bytes instead and ignores them -- and the case is data (`mode`), not control flow: with the
// loads behind an `if`, the compiler threads the case through to finish() and ends up with issue + wait + finish per sample,
// i.e. one round trip to memory per sample instead of one per kU samples.
struct TapLoad {
    uint32_t top[3], bot[3];
    float rw, bw;
    uint32_t sh;         // byte misalignment of the first tap, in bits (top row | bottom row << 8)
    int mode;            // 0 outside (sample = 0), 1 wide loads hold the taps, 2 near the end of the frame: byte loads in finish()
    uint32_t l, t;       // first tap (mode 2 only)
};

// The frame as 4-byte words in the GLOBAL address space: a pointer that went through uintptr_t arithmetic is a generic one to the
// compiler, and loads through it are flat_load (LDS / scratch aperture checks, and every LDS wait of the loop then waits for
// them as well); with the base uniform per workgroup and the offset 32 bits wide the tap loads are global_load ... s[base]
// with no 64-bit address arithmetic per sample.
typedef const uint32_t __attribute__((address_space(1)))* a3_gptr;

// OffT: uint32_t when the frame's bytes (+ 16) fit 32 bits and a row's stride 24, else uint64_t.  base4 = the frame's first byte rounded down to 4
// bytes (`mis` = what was rounded off), or any 16 readable bytes when the frame is smaller than one wide read (`tiny`: no
// sample then takes the wide path).  wide_limit = frame bytes - 12: the last offset at which a 12-byte read ends inside the frame.
template <typename OffT>
__device__ __forceinline__ void sample_issue(TapLoad& tl, a3_gptr base4, uint32_t mis, OffT row_stride, OffT wide_limit, bool tiny,
                                             uint32_t bpp, uint32_t w, uint32_t h, float x, float y, bool valid) {
    const float left = floorf(x), right = left + 1.0f, top = floorf(y), bottom = top + 1.0f;
    tl.rw = x - left; tl.bw = y - top;
    // (bitwise, not short-circuit: no branches)
    const bool inside = valid & !((left < 0.0f) | (right >= (float)w) | (top < 0.0f) | (bottom >= (float)h));
    // (converted unconditionally, selected afterwards: with the conversion inside the ternary each became an exec-mask region)
    const uint32_t lc = sat_u32(left), tc = sat_u32(top), bc = sat_u32(bottom);
    const uint32_t l = inside ? lc : 0u, t = inside ? tc : 0u, b = inside ? bc : 0u;
    tl.l = l; tl.t = t;
    OffT ot, ob;
    if constexpr (sizeof(OffT) == 4) {   // row index < 2^16, row_stride < 2^24 (the caller's condition for this path), products < 2^32: full-rate 24-bit multiplies
        const uint32_t xl = __umul24(bpp, l);
        ot = __umul24(t, (uint32_t)row_stride) + xl; ob = __umul24(b, (uint32_t)row_stride) + xl;
    } else {
        ot = (OffT)t * row_stride + (OffT)(bpp * l); ob = (OffT)b * row_stride + (OffT)(bpp * l);
    }
    // the wide read of the bottom row must end inside this frame (running on into the next row is fine)
    const bool wide = inside & !tiny & (ob <= wide_limit);
    tl.mode = inside ? (wide ? 1 : 2) : 0;
    const OffT at = (wide ? ot : (OffT)0) + (OffT)mis, ab = (wide ? ob : (OffT)0) + (OffT)mis;
    const a3_gptr qt = base4 + (at >> 2), qb = base4 + (ab >> 2);
    tl.top[0] = qt[0]; tl.top[1] = qt[1]; tl.top[2] = qt[2];
    tl.bot[0] = qb[0]; tl.bot[1] = qb[1]; tl.bot[2] = qb[2];
    // both rows start at the same misalignment only if row_stride % 4 == 0; keep one shift per row in sh's halves
    tl.sh = ((uint32_t)(at & 3u) * 8u) | (((uint32_t)(ab & 3u) * 8u) << 8);
}

__device__ __forceinline__ void pair_from(const uint32_t d[3], uint32_t sh, int fmt, uint32_t bpp, float* g0, float* g1) {
    const uint32_t w0 = __builtin_amdgcn_alignbit(d[1], d[0], sh), w1 = __builtin_amdgcn_alignbit(d[2], d[1], sh);   // bytes p .. p+7
    if (bpp == 1u) { *g0 = (float)(w0 & 0xFFu); *g1 = (float)((w0 >> 8) & 0xFFu); return; }
    const bool bgr = fmt == A3_FMT_BGRA8;
    const uint32_t px1 = bpp == 3u ? ((w0 >> 24) | (w1 << 8)) : w1;   // the dot products ignore byte 3
    *g0 = (float)luma_px(w0, bgr);
    *g1 = (float)luma_px(px1, bgr);
}

__device__ __forceinline__ uint8_t sample_finish(const TapLoad& tl, const uint8_t* __restrict__ img, size_t row_stride, int fmt, uint32_t bpp) {
    float a, b, c, d;
    pair_from(tl.top, tl.sh & 0xFFu, fmt, bpp, &a, &b);
    pair_from(tl.bot, tl.sh >> 8, fmt, bpp, &c, &d);
    if (tl.mode == 2) {   // the last few pixels of a frame: byte loads
        const uint8_t* rt = img + (size_t)tl.t * row_stride;
        const uint8_t* rb = rt + row_stride;
        a = grey_tap(rt, tl.l, fmt); b = grey_tap(rt, tl.l + 1u, fmt); c = grey_tap(rb, tl.l, fmt); d = grey_tap(rb, tl.l + 1u, fmt);
    }
    const uint8_t tv = clamp_u8((1.0f - tl.rw) * a + tl.rw * b);
    const uint8_t bv = clamp_u8((1.0f - tl.rw) * c + tl.rw * d);
    const uint8_t v = clamp_u8((1.0f - tl.bw) * (float)tv + tl.bw * (float)bv);
    return tl.mode == 0 ? (uint8_t)0 : v;
}

__device__ __forceinline__ float triangle_kernel(float x) { return fabsf(x) < 1.0f ? 1.0f - fabsf(x) : 0.0f; }

// image::imageops::resize weights for output index o (in_len -> out_len), normalised; returns left, sets count
__device__ uint32_t resize_weights(uint32_t in_len, uint32_t out_len, uint32_t o, float* ws, uint32_t* count) {
    const float ratio = (float)in_len / (float)out_len;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 1.0f * sratio;
    float input = ((float)o + 0.5f) * ratio;
    long long left = (long long)floorf(input - src_support);
    left = left < 0 ? 0 : (left > (long long)in_len - 1 ? (long long)in_len - 1 : left);
    long long right = (long long)ceilf(input + src_support);
    right = right < left + 1 ? left + 1 : (right > (long long)in_len ? (long long)in_len : right);
    input = input - 0.5f;
    float sum = 0.0f;
    uint32_t n = 0;
    for (long long i = left; i < right; i++) {
        const float w = triangle_kernel(((float)i - input) / sratio);
        ws[n++] = w;
        sum += w;
    }
    for (uint32_t i = 0; i < n; i++) ws[i] /= sum;
    *count = n;
    return (uint32_t)left;
}

// The 8x8 solve needs ~200 VGPRs for one lane's work; inside k_decode it would cap that kernel's occupancy, so it runs
// first, one lane per candidate (inside k_frame_candidates; k_projection is the stand-alone form the tuning probe uses), and
// leaves 9 floats + a flag per candidate.

// The triangle-resize weights of a full patch (S -> n) are the same for every candidate of a context: a table computed once
// (k_weight_table, at a3_create) replaces seven lanes of every decode workgroup computing them while the other 249 waited
// (~30 f32 divisions in a row: ~4 us on the critical path of each candidate).  Row o of the table: left, count, weights.
constexpr size_t kWeightTableBytes = 16384;   // n <= 10 rows of (2 + max_taps <= 202) floats
__global__ void k_weight_table(uint32_t S, uint32_t n, uint32_t max_taps, float* __restrict__ wtab) {
    if (threadIdx.x >= n) return;
    float* row = wtab + (size_t)threadIdx.x * (max_taps + 2);
    uint32_t cnt;
    const uint32_t left = resize_weights(S, n, threadIdx.x, row + 2, &cnt);
    row[0] = __uint_as_float(left); row[1] = __uint_as_float(cnt);
}
__global__ __launch_bounds__(64, A3_FC_WAVES) void k_projection(const uint16_t* __restrict__ fin_xy, const uint32_t* __restrict__ work,
                                                   const unsigned int* __restrict__ work_count, uint32_t S, ProjRec* __restrict__ proj) {
    const uint32_t n_work = *work_count;
    for (uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x; wi < n_work; wi += gridDim.x * blockDim.x) {
        const uint32_t slot = work[wi];
        const uint16_t* q = fin_xy + (size_t)slot * 8;
        float from[8];
        for (int i = 0; i < 8; i++) from[i] = (float)q[i];
        ProjRec r;
        r.ok = solve_projection(from, (float)S, r.inv) ? 1 : 0;
        proj[wi] = r;
    }
}

// rotate_bit_matrix (src/aruco.rs:315-326: new[a][b] = old[b][n-1-a], 90 degrees counter-clockwise) applied r times:
// cell (y, x) of the rotated matrix comes from the original at
//   r=0 (y,x)  r=1 (x,n-1-y)  r=2 (n-1-y,n-1-x)  r=3 (n-1-x,y)
__device__ __forceinline__ void rotated_source(uint32_t r, uint32_t n, uint32_t y, uint32_t x, uint32_t* sy, uint32_t* sx) {
    if (r == 0) { *sy = y; *sx = x; }
    else if (r == 1) { *sy = x; *sx = n - 1 - y; }
    else if (r == 2) { *sy = n - 1 - y; *sx = n - 1 - x; }
    else { *sy = n - 1 - x; *sx = y; }
}

// grid-stride over the work list; block = NT threads; dynamic LDS:
//   patch S*S | tmp n*S f32 | wtab n*max_taps f32 | wleft n u32 | wcnt n u32 | bits n*n
#ifndef A3_D_KU
#define A3_D_KU 2
#endif
#ifndef A3_D_WAVES
#define A3_D_WAVES 5
#endif
#ifndef A3_D_THREADS
#define A3_D_THREADS 256   // threads that sample one candidate (large batches; small ones: 256 throughout)
#endif
#ifndef A3_D_WAVES256
#define A3_D_WAVES256 5   // workgroups of 256 threads per CU-quarter (= waves per SIMD)
#endif
#ifndef A3_D_DB
#define A3_D_DB 8     // dictionary codes per lane and trip in the nearest-code scan
#endif
// Wave-wide scans and reductions by DPP row shifts (row_shr:1,2,4,8 inside rows of 16 lanes) and row broadcasts (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3): pure VALU, the result of a reduction ends up in lane 63.  The shuffle ladders
// they replace are six ds_bpermute round trips each.  `OLD` is what a lane without a source combines with: the identity.
#define A3_DPP_STEPS(STEP) STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
#define A3_S(CTRL, RM) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xF, false);
    A3_DPP_STEPS(A3_S)
#undef A3_S
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64_to63(unsigned long long v) {
#define A3_S(CTRL, RM)                                                                                                  \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, CTRL, RM, 0xF, false);           \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), CTRL, RM, 0xF, false);    \
        const unsigned long long o_ = ((unsigned long long)hi_ << 32) | lo_;                                             \
        v = o_ < v ? o_ : v;                                                                                             \
    }
    A3_DPP_STEPS(A3_S)
#undef A3_S
    return v;
}
// largest variance, lowest threshold among equals (the Otsu scan's first strict maximum); identity: (-1.0, INT_MAX)
__device__ __forceinline__ void wave_best_var_to63(double& var, int& best_t) {
#define A3_S(CTRL, RM)                                                                                                  \
    {                                                                                                                    \
        const unsigned long long vb_ = (unsigned long long)__double_as_longlong(var);                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)vb_, CTRL, RM, 0xF, false);           \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xBFF00000u, (int)(uint32_t)(vb_ >> 32), CTRL, RM, 0xF, false); \
        const int ot_ = __builtin_amdgcn_update_dpp(0x7FFFFFFF, best_t, CTRL, RM, 0xF, false);                             \
        const double ov_ = __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                     \
        if (ov_ > var || (ov_ == var && ot_ < best_t)) { var = ov_; best_t = ot_; }                                      \
    }
    A3_DPP_STEPS(A3_S)
#undef A3_S
}

// gather the accepted markers of frame f, candidate order preserved, by ONE WAVE (per_frame[] holds the number of accepted candidates
// of every frame: k_decode counted them; a frame's first output slot is the sum over the frames before it)
__device__ __forceinline__ void compact_frame_wave(uint32_t f, int lane, const DecodeOut* __restrict__ outs, const uint16_t* __restrict__ fin_xy,
                                                   const uint32_t* __restrict__ fin_count, uint32_t n_frames, uint32_t max_cand,
                                                   a3_marker* __restrict__ markers, uint32_t marker_cap, const uint32_t* __restrict__ per_frame,
                                                   unsigned int* __restrict__ marker_total, unsigned int* __restrict__ err_flags,
                                                   const uint32_t* __restrict__ cand_count, unsigned int* __restrict__ cand_pre_total) {
    uint32_t base = 0;
    for (uint32_t g = lane; g < f; g += 64) base += per_frame[g];
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    if (f + 1 == n_frames && lane == 0) *marker_total = base + per_frame[f];
    if (f + 1 == n_frames) {   // a3_stats.candidates_pre: quads after contours_to_candidates, summed over the batch
        uint32_t pre = 0, most = 0;   // cand_pre_total[3]: the largest number of quads any frame produced (beyond its table's slots: what the host grows the tables to)
        for (uint32_t g = lane; g < n_frames; g += 64) { pre += min(cand_count[g], max_cand); most = max(most, cand_count[g]); }
        for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); most = max(most, (uint32_t)__shfl_xor(most, o)); }
        if (lane == 0) { *cand_pre_total = pre; cand_pre_total[3] = most; }
    }
    const uint32_t c = fin_count[f];
    uint32_t pos = base;
    for (uint32_t k0 = 0; k0 < c; k0 += 64) {
        const uint32_t k = k0 + lane;
        DecodeOut o;
        o.valid = 0;
        if (k < c) o = outs[(size_t)f * max_cand + k];
        const unsigned long long m = __ballot(o.valid != 0);
        if (o.valid) {
            const uint32_t p = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (p < marker_cap) {
                a3_marker mk;
                mk.frame = f;
                mk.id = o.id;
                mk.code = o.code;
                const uint16_t* q = fin_xy + ((size_t)f * max_cand + k) * 8;
                for (int i = 0; i < 4; i++) {  // corners.rotate_left(min_rotation), src/aruco.rs:103
                    const int s2 = (i + o.rotation) & 3;
                    mk.corners[2 * i] = q[2 * s2];
                    mk.corners[2 * i + 1] = q[2 * s2 + 1];
                }
                mk.hamming_distance = o.hamming;
                mk.rotation = o.rotation;
                mk.candidate_index = (uint16_t)k;
                markers[p] = mk;
            } else atomicOr(err_flags, kErrMarkerCap);
        }
        pos += (uint32_t)__popcll(m);
    }
}

// NT threads sample one candidate, PT of them (64, or all) run the stages after the sampling.  History of the shape, on the
// 2.5 k candidates of BASELINE config 2 (tools/attic/tune_decode.sh): 256 threads throughout, 4 samples "in flight" per lane, row-major
// sample order (round 1): 116 us; 64 threads, 8 x 8 blocked order: 98 us; 256 threads sampling, the first wave doing the rest:
// 94 us -- and 7 us less than the 64-thread version inside the pipeline, where the frames are not in any cache.
template <int NT, int PT>
__global__ __launch_bounds__(NT, NT == 64 ? A3_D_WAVES : A3_D_WAVES256) void k_decode(PixelSrc src, int W, int H, uint32_t first_frame,
                                                const uint16_t* __restrict__ fin_xy, const uint32_t* __restrict__ work,
                                                const unsigned int* __restrict__ work_count, uint32_t max_cand, uint32_t S, uint32_t n,
                                                uint32_t max_taps, const uint64_t* __restrict__ dict, uint32_t n_codes, uint32_t tau,
                                                int filter, const ProjRec* __restrict__ proj, const float* __restrict__ wtab, DecodeOut* __restrict__ outs,
                                                uint8_t* __restrict__ patches /*nullable*/, uint32_t patch_cap, uint32_t* __restrict__ per_frame /*nullable*/, int dbg) {
    // dbg (a3_debug_kernel_time only, 0 in the product path): 1 = no sampling, 2 / 3 / 4 = stop after sampling / Otsu / bits
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_patch = smem;
    size_t o = ((size_t)S * S + 15) & ~(size_t)15;
    float* s_tmp = reinterpret_cast<float*>(smem + o); o += (size_t)n * S * 4;
    float* s_w = reinterpret_cast<float*>(smem + o); o += (size_t)n * max_taps * 4;
    uint32_t* s_left = reinterpret_cast<uint32_t*>(smem + o); o += (size_t)n * 4;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + o); o += (size_t)n * 4;
    uint8_t* s_bits = smem + o;
    __shared__ uint32_t s_hist[256];
    __shared__ float s_inv[9];
    __shared__ int s_ok, s_have;
    __shared__ uint32_t s_otsu;
    __shared__ uint64_t s_codes[4];
    __shared__ unsigned long long s_best[NT / 64][4];
    constexpr int NW = NT / 64;   // waves per candidate
    __shared__ uint32_t s_scan_w[NW], s_scan_s[NW];
    __shared__ double s_var[NW];
    __shared__ int s_vt[NW];

    const int tid = threadIdx.x;
    const uint32_t n_work = *work_count;
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint32_t slot = work[wi];
        const uint32_t fl = slot / max_cand;
        const uint8_t* img = src.base + (size_t)(first_frame + fl) * src.frame_stride;
        __syncthreads();
        if (tid < 9) s_inv[tid] = proj[wi].inv[tid];
        if (tid == 9) s_ok = proj[wi].ok;
        for (int i = tid; i < 256; i += NT) s_hist[i] = 0;
        __syncthreads();
        const bool ok = s_ok != 0;
        const uint32_t pw = ok ? S : 1u, ph = pw;
        // resize weights (same table for both passes: the patch and the grid are square): the launch-wide table for a full
        // patch, computed in place for the 1x1 stand-in of a failed projection (quirk Q4).  Fetched here, ahead of the
        // sampling, so that the trip to the table is not on the path between the Otsu level and the resize.
        if (ok) {
            for (uint32_t i = tid; i < n * (max_taps + 2); i += NT) {
                const uint32_t oi = i / (max_taps + 2), k = i - oi * (max_taps + 2);
                const float v = wtab[i];
                if (k == 0) s_left[oi] = __float_as_uint(v);
                else if (k == 1) s_cnt[oi] = __float_as_uint(v);
                else if (k - 2 < __float_as_uint(wtab[(size_t)oi * (max_taps + 2) + 1])) s_w[(size_t)oi * max_taps + (k - 2)] = v;
            }
        } else if (tid < (int)n) {
            const uint32_t oi = tid;
            uint32_t cnt;
            s_left[oi] = resize_weights(pw, n, oi, s_w + (size_t)oi * max_taps, &cnt);
            s_cnt[oi] = cnt;
        }
        // warp_into: integer output coordinates, no centre offset, mapping = the projection's inverse
        if (ok && dbg == 1) {
            for (uint32_t i = tid; i < S * S; i += NT) { s_patch[i] = (uint8_t)(i * 7u); atomicAdd(&s_hist[(i * 7u) & 255u], 1u); }
        } else if (ok) {
            const float t0 = s_inv[0], t1 = s_inv[1], t2 = s_inv[2], t3 = s_inv[3], t4 = s_inv[4], t5 = s_inv[5], t6 = s_inv[6],
                        t7 = s_inv[7], t8 = s_inv[8];
            // kU samples per lane per trip: all their row reads (one 12-byte load per row and sample: the two taps of a row
            // are adjacent) are in flight before the first is converted
            constexpr int kU = A3_D_KU;
            // Sample order: blocks of 8 x 8 output pixels, one block per wave instruction (lane = 8 * (y & 7) + (x & 7)).  Taps
            // of a block lie in a compact patch of the frame whatever the marker's rotation, so the 64 lanes of a load touch
            // a few dozen cache lines; a row-major order puts the 64 samples of an instruction on a slanted line that crosses
            // a new image row -- a new cache line -- at almost every sample of a rotated marker (A3_D_BLOCKED=0: row-major).
#ifndef A3_D_BLOCKED
#define A3_D_BLOCKED 1
#endif
            const uint32_t bpp = (src.fmt == A3_FMT_RGB8) ? 3u : ((src.fmt == A3_FMT_RGBA8 || src.fmt == A3_FMT_BGRA8) ? 4u : 1u);
            // (a frame smaller than one wide read cannot hold a candidate; the weight table is merely something of 16 KB to read)
            const unsigned long long frame_bytes = (unsigned long long)(H - 1) * src.row_stride + (unsigned long long)W * bpp;
            const bool tiny = frame_bytes < 16ull, off32 = frame_bytes < 0xFFFFFFE0ull && src.row_stride < (1ull << 24);
            const uintptr_t img_u = reinterpret_cast<uintptr_t>(img);
            const uint32_t mis = tiny ? 0u : (uint32_t)(img_u & 3u);
            const a3_gptr base4 = reinterpret_cast<a3_gptr>(tiny ? reinterpret_cast<uintptr_t>(wtab) : img_u - mis);
            const uint32_t nbx = (S + 7u) / 8u, n_slots = A3_D_BLOCKED ? nbx * nbx * 64u : S * S;
            // slot -> (x, y); i / S by a multiply-high in the row-major order (S is uniform but not a compile-time constant)
            // blk / nbx as (blk * Mb) >> 16, Mb = 2^16 / nbx + 1: exact while blk * nbx < 2^16 (blk < nbx^2 <= 625, nbx <= 25), and full rate
            const uint32_t M = S > 1 ? 0xFFFFFFFFu / S + 1u : 0u, Mb = 65536u / nbx + 1u;
            auto slot_xy = [&](uint32_t slot, uint32_t* x, uint32_t* y) -> bool {
                if (A3_D_BLOCKED) {
                    const uint32_t blk = slot >> 6, l = slot & 63u, by = __umul24(min(blk, 1023u), Mb) >> 16, bx = blk - __umul24(by, nbx);
                    *x = bx * 8u + (l & 7u); *y = by * 8u + (l >> 3);
                    return slot < n_slots && *x < S && *y < S;
                }
                const uint32_t row = S > 1 ? __umulhi(slot, M) : slot;
                *x = slot - row * S; *y = row;
                return slot < n_slots;
            };
            for (uint32_t i0 = tid; i0 < n_slots; i0 += NT * kU) {
                TapLoad tl[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    uint32_t x, y;
                    const bool valid = slot_xy(i0 + (uint32_t)NT * u, &x, &y);
                    const float fx = (float)x, fy = (float)y;
                    const float d = t6 * fx + t7 * fy + t8;
                    const float px = (t0 * fx + t1 * fy + t2) / d;
                    const float py = (t3 * fx + t4 * fy + t5) / d;
                    if (off32) sample_issue<uint32_t>(tl[u], base4, mis, (uint32_t)src.row_stride, (uint32_t)(frame_bytes - 12ull), tiny, bpp, (uint32_t)W, (uint32_t)H, px, py, valid);
                    else sample_issue<unsigned long long>(tl[u], base4, mis, src.row_stride, frame_bytes - 12ull, tiny, bpp, (uint32_t)W, (uint32_t)H, px, py, valid);
                }
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    uint32_t x, y;
                    if (slot_xy(i0 + (uint32_t)NT * u, &x, &y)) {
                        const uint8_t v = sample_finish(tl[u], img, (size_t)src.row_stride, src.fmt, bpp);
                        s_patch[__umul24(y, S) + x] = v;
                        atomicAdd(&s_hist[v], 1u);
                    }
                }
            }
        } else if (tid == 0) {  // GrayImage::new(1, 1): one black pixel (quirk Q4)
            s_patch[0] = 0;
            s_hist[0] = 1;
        }
        __syncthreads();
        if (dbg == 2) continue;
        // PT = 64: everything after the sampling is the work of ONE wave, whatever the number of waves that sampled: its stages are short
        // loops over 64 .. 400 items separated by synchronisation points, which for a single wave cost nothing (LDS operations
        // of a wave complete in order; POST_SYNC only keeps the compiler from moving accesses across it), while four waves pay
        // a barrier -- and the wait for every outstanding memory access that comes with it -- at each of them.  The other waves
        // go on to the top of the loop.
        static_assert(PT == 64 || PT == NT, "the stages after the sampling run on one wave or on all");
        constexpr int NWP = PT / 64;
#define POST_SYNC() { if constexpr (PT == NT) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } }
        if (PT == NT || tid < PT) {
            // debug taps: the patch buffer holds patch_cap patches, one per work item (NOT per candidate slot: slots are
            // frame * max_cand + k and would run past the buffer for frames beyond patch_cap / max_cand)
            const bool keep_patch = patches != nullptr && wi < patch_cap;
            if (keep_patch) {
                uint8_t* dst = patches + (size_t)wi * S * S;
                for (uint32_t i = tid; i < S * S; i += PT) dst[i] = ok ? s_patch[i] : 0;
            }
            // otsu_level (imageproc): the reference scans thresholds 0..255 keeping running integer sums and the first strict
            // maximum of w_b * w_f * (mean_b - mean_f)^2 in f64.  The running sums are exact integers, so every threshold can
            // be evaluated independently from prefix sums with the very same f64 operations; the first strict maximum is the
            // largest variance, lowest threshold among equals, and it must exceed the initial 0.0.
            {
                constexpr int B = 256 / PT;                 // thresholds per lane: t = tid * B + k
                uint32_t cw[B], cs[B];                      // inclusive prefix sums inside the lane's run of thresholds
                uint32_t rw = 0, rs = 0;
    #pragma unroll
                for (int k = 0; k < B; k++) {
                    const uint32_t t = (uint32_t)tid * B + k, hcnt = s_hist[t];
                    rw += hcnt; rs += t * hcnt;
                    cw[k] = rw; cs[k] = rs;
                }
                uint32_t bw = wave_incl_add(rw), bs = wave_incl_add(rs);   // inclusive prefix sums over lanes
                if ((tid & 63) == 63) { s_scan_w[tid >> 6] = bw; s_scan_s[tid >> 6] = bs; }
                POST_SYNC();
                uint32_t total_sum_u = 0;
                for (int w = 0; w < NWP; w++) {
                    if (w < (tid >> 6)) { bw += s_scan_w[w]; bs += s_scan_s[w]; }
                    total_sum_u += s_scan_s[w];
                }
                const uint32_t before_w = bw - rw, before_s = bs - rs;   // everything below this lane's first threshold
                const uint32_t total_weight = pw * ph;
                const double total_pixel_sum = (double)total_sum_u;
                double var = -1.0;   // "not a candidate"
                int best_t = tid * B;
    #pragma unroll
                for (int k = 0; k < B; k++) {
                    const uint32_t bwk = before_w + cw[k], bsk = before_s + cs[k];
                    const uint32_t fw = total_weight - bwk;
                    if (bwk != 0 && fw != 0) {
                        const double background_pixel_sum = (double)bsk;
                        const double foreground_pixel_sum = total_pixel_sum - background_pixel_sum;
                        const double background_mean = background_pixel_sum / (double)bwk;
                        const double foreground_mean = foreground_pixel_sum / (double)fw;
                        const double diff = background_mean - foreground_mean;
                        const double mean_diff_squared = diff * diff;
                        const double v = (double)bwk * (double)fw * mean_diff_squared;
                        if (v > var) { var = v; best_t = tid * B + k; }   // ascending k: the first strict maximum of the run
                    }
                }
                wave_best_var_to63(var, best_t);
                if ((tid & 63) == 63) { s_var[tid >> 6] = var; s_vt[tid >> 6] = best_t; }
                POST_SYNC();
                if (tid == 0) {
                    double bv = s_var[0]; int bt = s_vt[0];
                    for (int w = 1; w < NWP; w++) if (s_var[w] > bv || (s_var[w] == bv && s_vt[w] < bt)) { bv = s_var[w]; bt = s_vt[w]; }
                    s_otsu = bv > 0.0 ? (uint32_t)bt : 0u;
                }
            }
            if (dbg == 3) continue;
            // threshold(.., Binary) is applied where the patch is read: a pixel counts as 255 iff it is above the Otsu level
            POST_SYNC();   // s_otsu (and, long ago, the weights)
            const uint32_t otsu = s_otsu;
            if (pw == n) {  // resize() copies when the size already matches
                for (uint32_t i = tid; i < n * n; i += PT) s_bits[i] = s_patch[i] > otsu;   // (255 or 0) > 127
            } else {
                for (uint32_t i = tid; i < n * pw; i += PT) {  // vertical pass into f32
                    const uint32_t oy = i / pw, x = i - oy * pw;
                    const float* ws = s_w + (size_t)oy * max_taps;
                    const uint32_t left = s_left[oy], cnt = s_cnt[oy];
                    float t = 0.0f;
                    for (uint32_t k = 0; k < cnt; k++) t += (s_patch[(left + k) * pw + x] > otsu ? 255.0f : 0.0f) * ws[k];
                    s_tmp[oy * pw + x] = t;
                }
                POST_SYNC();
                for (uint32_t i = tid; i < n * n; i += PT) {  // horizontal pass, clamp, round to nearest
                    const uint32_t y = i / n, ox = i - y * n;
                    const float* ws = s_w + (size_t)ox * max_taps;
                    const uint32_t left = s_left[ox], cnt = s_cnt[ox];
                    float t = 0.0f;
                    for (uint32_t k = 0; k < cnt; k++) t += s_tmp[y * pw + left + k] * ws[k];
                    const float c = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
                    s_bits[y * n + ox] = (uint8_t)roundf(c) > 127;
                }
            }
            POST_SYNC();
            if (dbg == 4) continue;
            // border test + 4 rotated codes, src/aruco.rs:287-310.  One lane per (rotation, cell): the bit goes to its place in
            // the code with an LDS atomic (row-major, first cell = most significant bit).
            if (tid < 4) s_codes[tid] = 0;
            {   // the marker's border must be black all round: one lane per border cell (left, right, top, bottom x n)
                const uint32_t end = n ? n - 1 : 0;
                int lit = 0;
                for (uint32_t t = (uint32_t)tid; t < 4u * n; t += PT) {
                    const uint32_t side = t / n, i = t - side * n;
                    lit |= s_bits[side == 0 ? i * n : (side == 1 ? i * n + end : (side == 2 ? i : end * n + i))];
                }
                const int any_lit = (PT == NT ? __syncthreads_or(lit) : (int)(__ballot(lit) != 0ull));
                if (tid == 0) s_have = !any_lit;
            }
            POST_SYNC();
            {
                const uint32_t inner = n >= 2 ? n - 2 : 0, cells = inner * inner;   // <= 64 cells, 4 rotations
                if (s_have) {
                    for (uint32_t t = (uint32_t)tid; t < 4u * cells; t += PT) {
                        const uint32_t r = t / cells, idx = t - r * cells;
                        const uint32_t y = 1 + idx / inner, x = 1 + idx % inner;
                        // rotation r of the bit matrix, read row-major
                        uint32_t sy, sx;
                        rotated_source(r, n, y, x, &sy, &sx);
                        if (s_bits[sy * n + sx]) atomicOr(reinterpret_cast<unsigned long long*>(&s_codes[r]), 1ull << (cells - 1 - idx));
                    }
                }
            }
            POST_SYNC();
            const int have = s_have;
            // find_nearest for the 4 codes: strict '<' => lowest index among equal distances
            unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
            if (have) {
                const uint64_t c0 = s_codes[0], c1 = s_codes[1], c2 = s_codes[2], c3 = s_codes[3];
                // eight codes per lane and trip, their loads issued together (unconditional, clamped index): one code per trip is a
                // chain of n_codes / 64 round trips to the table -- 16 for a 1024-code dictionary
                constexpr int DB = A3_D_DB;
                for (uint32_t i0 = tid; i0 < n_codes; i0 += PT * DB) {
                    uint64_t cw[DB];
    #pragma unroll
                    for (int u = 0; u < DB; u++) cw[u] = dict[min(i0 + (uint32_t)(u * PT), n_codes - 1u)];
    #pragma unroll
                    for (int u = 0; u < DB; u++) {
                        const uint32_t i = i0 + (uint32_t)(u * PT);
                        const unsigned long long past = 0ull - (unsigned long long)(i >= n_codes);   // all ones: never the minimum
                        const uint64_t c = cw[u];
                        const unsigned long long k0 = (((unsigned long long)__popcll(c ^ c0) << 32) | i) | past;
                        const unsigned long long k1 = (((unsigned long long)__popcll(c ^ c1) << 32) | i) | past;
                        const unsigned long long k2 = (((unsigned long long)__popcll(c ^ c2) << 32) | i) | past;
                        const unsigned long long k3 = (((unsigned long long)__popcll(c ^ c3) << 32) | i) | past;
                        best[0] = k0 < best[0] ? k0 : best[0];
                        best[1] = k1 < best[1] ? k1 : best[1];
                        best[2] = k2 < best[2] ? k2 : best[2];
                        best[3] = k3 < best[3] ? k3 : best[3];
                    }
                }
    #pragma unroll
                for (int r = 0; r < 4; r++) best[r] = wave_min_u64_to63(best[r]);
                if ((tid & 63) == 63)
                    for (int r = 0; r < 4; r++) s_best[tid >> 6][r] = best[r];
            }
            POST_SYNC();
            if (tid == 0) {
                DecodeOut out;
                out.valid = 0; out.id = 0; out.code = 0; out.rotation = 0; out.hamming = 0; out.patch = keep_patch ? wi : kNone;
                out.hom_ok = ok; out.decode_ok = have;
                for (int r = 0; r < 4; r++) out.codes[r] = have ? s_codes[r] : 0;
                int found_any = 0;
                uint32_t min_code_distance = 0x7FFFFFFFu, min_rotation = 0, min_code_id = 0x7FFFFFFFu;
                uint64_t min_code = 0x7FFFFFFFull;
                if (have) {
                    for (uint32_t r = 0; r < 4; r++) {
                        unsigned long long b = s_best[0][r];
                        for (int w = 1; w < NWP; w++) b = s_best[w][r] < b ? s_best[w][r] : b;
                        // empty dictionary: find_nearest returns (0, 0xFF)
                        const uint32_t nearest_dist = n_codes ? (uint32_t)(b >> 32) : 0xFFu;
                        const uint32_t nearest_id = n_codes ? (uint32_t)b : 0u;
                        if (nearest_dist < min_code_distance) {
                            min_code = s_codes[r]; min_code_distance = nearest_dist; min_code_id = nearest_id; min_rotation = r; found_any = 1;
                        }
                    }
                }
                if (found_any && (!filter || min_code_distance < tau)) {
                    out.valid = 1; out.id = min_code_id; out.code = min_code; out.rotation = (uint8_t)min_rotation;
                    out.hamming = (uint8_t)min_code_distance;
                }
                outs[slot] = out;
                if (out.valid && per_frame) atomicAdd(&per_frame[first_frame + fl], 1u);   // one address per frame: no contention to speak of
            }
        }
#undef POST_SYNC
    }
}

// gather the accepted markers, frame by frame, candidate order preserved: one wave per frame.  per_frame[] already holds
// the number of accepted candidates of every frame (k_decode counted them), so a frame's first output slot is a sum over
// the frames before it, which every wave forms for itself (n_frames is a few hundred; the single-workgroup kernel below
// takes over beyond kCompactParMax frames).
constexpr uint32_t kCompactParMax = 4096;
__global__ __launch_bounds__(256) void k_compact_markers_par(const DecodeOut* __restrict__ outs, const uint16_t* __restrict__ fin_xy,
                                                             const uint32_t* __restrict__ fin_count, uint32_t n_frames, uint32_t max_cand,
                                                             a3_marker* __restrict__ markers, uint32_t marker_cap,
                                                             const uint32_t* __restrict__ per_frame, unsigned int* __restrict__ marker_total,
                                                             unsigned int* __restrict__ err_flags, const uint32_t* __restrict__ cand_count,
                                                             unsigned int* __restrict__ cand_pre_total) {
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    compact_frame_wave(f, threadIdx.x & 63, outs, fin_xy, fin_count, n_frames, max_cand, markers, marker_cap, per_frame, marker_total, err_flags, cand_count,
                       cand_pre_total);
}

// The same with one workgroup and no pre-computed counts (any number of frames).
__global__ __launch_bounds__(256) void k_compact_markers(const DecodeOut* __restrict__ outs, const uint16_t* __restrict__ fin_xy,
                                                         const uint32_t* __restrict__ fin_count, uint32_t n_frames, uint32_t first_frame,
                                                         uint32_t max_cand, a3_marker* __restrict__ markers, uint32_t marker_cap,
                                                         uint32_t* __restrict__ per_frame, unsigned int* __restrict__ marker_total,
                                                         unsigned int* __restrict__ err_flags, const uint32_t* __restrict__ cand_count,
                                                         unsigned int* __restrict__ cand_pre_total) {
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = *marker_total;
    {
        uint32_t pre = 0, most = 0;
        for (uint32_t g = tid; g < n_frames; g += 256) { pre += min(cand_count[g], max_cand); most = max(most, cand_count[g]); }
        if (pre) atomicAdd(cand_pre_total, pre);
        if (most) atomicMax(cand_pre_total + 3, most);
    }
    __syncthreads();
    for (uint32_t f0 = 0; f0 < n_frames; f0 += 256) {
        const uint32_t f = f0 + tid;
        uint32_t cnt = 0;
        if (f < n_frames) {
            const uint32_t c = fin_count[f];
            for (uint32_t k = 0; k < c; k++) cnt += outs[(size_t)f * max_cand + k].valid;
        }
        s_scan[tid] = cnt;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const uint32_t v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const uint32_t excl = s_scan[tid] - cnt, tile_total = s_scan[255];
        const uint32_t base = s_base;
        if (f < n_frames) {
            per_frame[first_frame + f] = cnt;
            uint32_t pos = base + excl;
            const uint32_t c = fin_count[f];
            for (uint32_t k = 0; k < c; k++) {
                const DecodeOut o = outs[(size_t)f * max_cand + k];
                if (!o.valid) continue;
                if (pos < marker_cap) {
                    a3_marker m;
                    m.frame = first_frame + f;
                    m.id = o.id;
                    m.code = o.code;
                    const uint16_t* q = fin_xy + ((size_t)f * max_cand + k) * 8;
                    for (int i = 0; i < 4; i++) {  // corners.rotate_left(min_rotation), src/aruco.rs:103
                        const int s = (i + o.rotation) & 3;
                        m.corners[2 * i] = q[2 * s];
                        m.corners[2 * i + 1] = q[2 * s + 1];
                    }
                    m.hamming_distance = o.hamming;
                    m.rotation = o.rotation;
                    m.candidate_index = (uint16_t)k;
                    markers[pos] = m;
                } else atomicOr(err_flags, kErrMarkerCap);
                pos++;
            }
        }
        __syncthreads();
        if (tid == 0) s_base = base + tile_total;
        __syncthreads();
    }
    if (tid == 0) *marker_total = s_base;
}

// Fixed-capacity detection records for the multi-GPU gather: one record per frame,
//   u32 count | u32 global frame index | maxm x a3_marker | (with poses: maxm x 2 x a3_pose, the pair of marker k at 2k, 2k+1)
// (marker.frame rewritten to the global index, unused slots zero), written straight from the device-resident marker (and pose)
// list of the last batch: one wave per frame.  A frame with more than maxm markers raises *overflow (the host turns it into
// A3_ERR_CAPACITY: records are never clipped silently).
__global__ __launch_bounds__(256) void k_pack_detections(const a3_marker* __restrict__ markers, const a3_pose* __restrict__ poses /*nullable*/,
                                                         const uint32_t* __restrict__ per_frame,
                                                         uint32_t n_frames, uint32_t first_frame_global, uint32_t maxm,
                                                         uint32_t* __restrict__ dst, unsigned int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    uint32_t base = 0;
    for (uint32_t g = lane; g < f; g += 64) base += per_frame[g];
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    const uint32_t cnt = per_frame[f];
    constexpr uint32_t kMarkerWords = sizeof(a3_marker) / 4, kPoseWords = 2 * sizeof(a3_pose) / 4;
    const uint32_t rec_words = 2u + maxm * (kMarkerWords + (poses ? kPoseWords : 0u));
    uint32_t* rec = dst + (size_t)f * rec_words;
    if (cnt > maxm) { if (lane == 0) atomicOr(overflow, 1u); }
    const uint32_t kept = min(cnt, maxm);
    if (lane == 0) { rec[0] = kept; rec[1] = first_frame_global + f; }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(markers + base);
    for (uint32_t w = lane; w < maxm * kMarkerWords; w += 64) {
        uint32_t v = 0u;
        if (w < kept * kMarkerWords) v = (w % kMarkerWords == 0u) ? first_frame_global + f : src[w];   // word 0 of a marker = .frame
        rec[2u + w] = v;
    }
    if (poses) {
        const uint32_t* ps = reinterpret_cast<const uint32_t*>(poses + 2 * (size_t)base);
        uint32_t* pd = rec + 2u + maxm * kMarkerWords;
        for (uint32_t w = lane; w < maxm * kPoseWords; w += 64) pd[w] = w < kept * kPoseWords ? ps[w] : 0u;
    }
}

// Detection.homographies of one frame in one piece (debug taps): the frame's candidates keep their patches wherever their work
// items fell in the tap; one workgroup per candidate copies its patch into a dense array that leaves in a single D2H copy
// (a copy per patch was a blocking hipMemcpy per candidate).  A candidate whose patch was not kept gets zeros and sets *missing.
__global__ __launch_bounds__(256) void k_gather_patches(const DecodeOut* __restrict__ outs, uint32_t n_cand, const uint8_t* __restrict__ patches,
                                                        uint32_t patch_cap, uint32_t S2, uint8_t* __restrict__ dst, unsigned int* __restrict__ missing) {
    const uint32_t k = blockIdx.x;
    if (k >= n_cand) return;
    const uint32_t slot = outs[k].patch;
    const bool have = slot < patch_cap;
    if (!have && threadIdx.x == 0) atomicOr(missing, 1u);
    const uint8_t* src = patches + (size_t)(have ? slot : 0u) * S2;
    for (uint32_t i = threadIdx.x; i < S2; i += blockDim.x) dst[(size_t)k * S2 + i] = have ? src[i] : (uint8_t)0;
}

hipError_t launch_gather_patches(hipStream_t st, const void* outs_frame, uint32_t n_cand, const uint8_t* patches, uint32_t patch_cap, uint32_t S2,
                                 uint8_t* dst, unsigned int* missing) {
    if (n_cand == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_patches, dim3(n_cand), dim3(256), 0, st, reinterpret_cast<const DecodeOut*>(outs_frame), n_cand, patches, patch_cap, S2,
                       dst, missing);
    return hipGetLastError();
}

// the bit-matrix rotation of the decode kernel on its own (reference vectors: src/aruco.rs:414-444)
__global__ void k_debug_rotate_bits(const uint8_t* __restrict__ in, uint32_t n, uint32_t r, uint8_t* __restrict__ out) {
    const uint32_t i = threadIdx.x;
    if (i >= n * n) return;
    uint32_t sy, sx;
    rotated_source(r, n, i / n, i % n, &sy, &sx);
    out[i] = in[sy * n + sx];
}

// ---------------------------------------------------------------------------------------
// IPPE pose, one lane per marker (src/pose.rs:52-348; matrices row-major)
// ---------------------------------------------------------------------------------------
__device__ void find_rotation_to_z(const float v[3], float rot[9]) {  // src/pose.rs:238-267
    for (int i = 0; i < 9; i++) rot[i] = 0.0f;
    const float a = v[0] * v[0], b = v[1] * v[1], c = v[2] * v[2];
    const float nrm = sqrtf(a + b + c);
    const float ax = v[0] / nrm, ay = v[1] / nrm, az = v[2] / nrm;
    if (fabsf(1.0f + az) < 1e-6f) {
        rot[0] = 1.0f; rot[4] = 1.0f; rot[8] = -1.0f;
    } else {
        const float d = 1.0f / (1.0f + az);
        const float ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
        rot[0] = -ax2 * d + 1.0f; rot[1] = -axay * d;       rot[2] = -ax;
        rot[3] = -axay * d;       rot[4] = -ay2 * d + 1.0f; rot[5] = -ay;
        rot[6] = ax;              rot[7] = ay;              rot[8] = 1.0f - (ax2 + ay2) * d;
    }
}

__device__ void compute_rotations(const float j[4], float tx, float ty, float r1[9], float r2[9]) {  // src/pose.rs:158-235
    const float t[3] = {tx, ty, 1.0f};
    float rz[9], rv[9];
    find_rotation_to_z(t, rz);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rv[r * 3 + c] = rz[c * 3 + r];
#define RV(r, c) rv[((r) - 1) * 3 + ((c) - 1)]
    const float b00 = RV(1, 1) - tx * RV(3, 1);
    const float b01 = RV(1, 2) - tx * RV(3, 2);
    const float b10 = RV(2, 1) - ty * RV(3, 1);
    const float b11 = RV(2, 2) - ty * RV(3, 2);
    const float inv_det = 1.0f / (b00 * b11 - b01 * b10);
    const float binv00 = inv_det * b11, binv01 = -inv_det * b01, binv10 = -inv_det * b10, binv11 = inv_det * b00;
    const float a00 = binv00 * j[0] + binv01 * j[2];
    const float a01 = binv00 * j[1] + binv01 * j[3];
    const float a10 = binv10 * j[0] + binv11 * j[2];
    const float a11 = binv10 * j[1] + binv11 * j[3];
    const float ata00 = a00 * a00 + a01 * a01;
    const float ata01 = a00 * a10 + a01 * a11;
    const float ata11 = a10 * a10 + a11 * a11;
    const float gamma = sqrtf(0.5f * (ata00 + ata11 + sqrtf((ata00 - ata11) * (ata00 - ata11) + 4.0f * ata01 * ata01)));
    const float rt00 = a00 / gamma, rt01 = a01 / gamma, rt10 = a10 / gamma, rt11 = a11 / gamma;
    const float rt00_2 = rt00 * rt00, rt01_2 = rt01 * rt01, rt10_2 = rt10 * rt10, rt11_2 = rt11 * rt11;
    const float b0 = sqrtf(-rt00_2 - rt10_2 + 1.0f);
    float b1 = sqrtf(-rt01_2 - rt11_2 + 1.0f);
    const float sp = -rt00 * rt01 - rt10 * rt11;
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

__device__ void compute_translation(const float obj[12], const float pts[8], const float rot[9], float t[3]) {  // src/pose.rs:269-335
    float m11 = 4.0f, m13 = 0.0f, m22 = 4.0f, m23 = 0.0f, m31 = 0.0f, m32 = 0.0f, m33 = 0.0f;
    float atb0 = 0.0f, atb1 = 0.0f, atb2 = 0.0f;
    for (int i = 0; i < 4; i++) {
        const float ox = obj[3 * i], oy = obj[3 * i + 1];
        const float rx = rot[0] * ox + rot[1] * oy;
        const float ry = rot[3] * ox + rot[4] * oy;
        const float rz = rot[6] * ox + rot[7] * oy;
        const float a2 = -pts[2 * i], b2 = -pts[2 * i + 1];
        m13 += a2; m23 += b2; m31 += a2; m32 += b2;
        m33 += a2 * a2 + b2 * b2;
        const float bx = -a2 * rz - rx;
        const float by = -b2 * rz - ry;
        atb0 += bx; atb1 += by;
        atb2 += a2 * bx + b2 * by;
    }
    const float det_a_inv = 1.0f / (m11 * m22 * m33 - m11 * m23 * m32 - m13 * m22 * m31);
    const float s11 = m22 * m33 - m23 * m32, s12 = m13 * m32, s13 = -m13 * m22;
    const float s21 = m23 * m31, s22 = m11 * m33 - m13 * m31, s23 = -m11 * m23;
    const float s31 = -m22 * m31, s32 = -m11 * m32, s33 = m11 * m22;
    t[0] = det_a_inv * (s11 * atb0 + s12 * atb1 + s13 * atb2);
    t[1] = det_a_inv * (s21 * atb0 + s22 * atb1 + s23 * atb2);
    t[2] = det_a_inv * (s31 * atb0 + s32 * atb1 + s33 * atb2);
}

__device__ float reprojection_error(const a3_pose& p, const float obj[12], const float pts[8]) {  // src/pose.rs:337-348
    float error = 0.0f;
    const float* r = p.rotation;
    for (int i = 0; i < 4; i++) {
        const float x = obj[3 * i], y = obj[3 * i + 1], z = obj[3 * i + 2];
        const float px = (r[0] * x + r[1] * y + r[2] * z) + p.translation[0];
        const float py = (r[3] * x + r[4] * y + r[5] * z) + p.translation[1];
        const float pz = (r[6] * x + r[7] * y + r[8] * z) + p.translation[2];
        const float zz = pz > 1e-5f ? pz : 1e-5f;
        const float dx = (px / zz) - pts[2 * i];
        const float dy = (py / zz) - pts[2 * i + 1];
        error += sqrtf(dx * dx + dy * dy);
    }
    return error;
}

__device__ void solve_normalized(const float pts[8], float marker_size_mm, a3_pose* o1, a3_pose* o2) {  // src/pose.rs:64-156
    const float hw = 0.5f * marker_size_mm;
    const float obj[12] = {-hw, hw, 0.0f, hw, hw, 0.0f, hw, -hw, 0.0f, -hw, -hw, 0.0f};
    const float p1x = -pts[0], p1y = -pts[1], p2x = -pts[2], p2y = -pts[3], p3x = -pts[4], p3y = -pts[5], p4x = -pts[6], p4y = -pts[7];
    const float half_width = marker_size_mm / 2.0f;
    const float det_inv = -1.0f / (half_width * (p1x * p2y - p2x * p1y - p1x * p4y + p2x * p3y - p3x * p2y + p4x * p1y + p3x * p4y - p4x * p3y));
    float h[9];
    h[0] = det_inv * (p1x * p3x * p2y - p2x * p3x * p1y - p1x * p4x * p2y + p2x * p4x * p1y - p1x * p3x * p4y + p1x * p4x * p3y + p2x * p3x * p4y - p2x * p4x * p3y);
    h[1] = det_inv * (p1x * p2x * p3y - p1x * p3x * p2y - p1x * p2x * p4y + p2x * p4x * p1y + p1x * p3x * p4y - p3x * p4x * p1y - p2x * p4x * p3y + p3x * p4x * p2y);
    h[2] = det_inv * half_width * (p1x * p2x * p3y - p2x * p3x * p1y - p1x * p2x * p4y + p1x * p4x * p2y - p1x * p4x * p3y + p3x * p4x * p1y + p2x * p3x * p4y - p3x * p4x * p2y);
    h[3] = det_inv * (p1x * p2y * p3y - p2x * p1y * p3y - p1x * p2y * p4y + p2x * p1y * p4y - p3x * p1y * p4y + p4x * p1y * p3y + p3x * p2y * p4y - p4x * p2y * p3y);
    h[4] = det_inv * (p2x * p1y * p3y - p3x * p1y * p2y - p1x * p2y * p4y + p4x * p1y * p2y + p1x * p3y * p4y - p4x * p1y * p3y - p2x * p3y * p4y + p3x * p2y * p4y);
    h[5] = det_inv * half_width * (p1x * p2y * p3y - p3x * p1y * p2y - p2x * p1y * p4y + p4x * p1y * p2y - p1x * p3y * p4y + p3x * p1y * p4y + p2x * p3y * p4y - p4x * p2y * p3y);
    h[6] = -det_inv * (p1x * p3y - p3x * p1y - p1x * p4y - p2x * p3y + p3x * p2y + p4x * p1y + p2x * p4y - p4x * p2y);
    h[7] = det_inv * (p1x * p2y - p2x * p1y - p1x * p3y + p3x * p1y + p2x * p4y - p4x * p2y - p3x * p4y + p4x * p3y);
    h[8] = 1.0f;
    const float j[4] = {h[0] - h[6] * h[2], h[1] - h[7] * h[2], h[3] - h[6] * h[5], h[4] - h[7] * h[5]};
    a3_pose a, b;
    compute_rotations(j, h[2], h[5], a.rotation, b.rotation);
    compute_translation(obj, pts, a.rotation, a.translation);
    compute_translation(obj, pts, b.rotation, b.translation);
    a.error = reprojection_error(a, obj, pts);
    b.error = reprojection_error(b, obj, pts);
    if (a.error < b.error) { *o1 = a; *o2 = b; } else { *o1 = b; *o2 = a; }
}

// mode 0: pts = corners / (w,h) (solve_with_undistorted_points); 1: unproject through intrinsics; 2: already normalised
// corner_stride: u32 words between the corner lists of consecutive markers (8 for a packed list, 14 inside a3_marker[]);
// n_dev (optional): marker count produced on the device by k_compact_markers.
__global__ void k_pose(const uint32_t* __restrict__ corners, uint32_t corner_stride, const float* __restrict__ norm_pts, uint32_t n,
                       const unsigned int* __restrict__ n_dev, int mode, float marker_size_mm,
                       float iw, float ih, float fx, float fy, float cx, float cy, a3_pose* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    float pts[8];
    for (int k = 0; k < 4; k++) {
        if (mode == 2) { pts[2 * k] = norm_pts[8 * i + 2 * k]; pts[2 * k + 1] = norm_pts[8 * i + 2 * k + 1]; continue; }
        const float x = (float)corners[(size_t)corner_stride * i + 2 * k], y = (float)corners[(size_t)corner_stride * i + 2 * k + 1];
        if (mode == 0) { pts[2 * k] = x / iw; pts[2 * k + 1] = y / ih; }            // src/pose.rs:60
        else { pts[2 * k] = (x - cx) / fx; pts[2 * k + 1] = (y - cy) / fy; }         // src/pinhole.rs:88-93
    }
    a3_pose p1, p2;
    solve_normalized(pts, marker_size_mm, &p1, &p2);
    out[2 * i] = p1;
    out[2 * i + 1] = p2;
}

// ---------------------------------------------------------------------------------------
// dictionary utilities
// ---------------------------------------------------------------------------------------
__global__ void k_find_nearest(const uint64_t* __restrict__ dict, uint32_t n_codes, const uint64_t* __restrict__ bits, uint32_t n,
                               uint32_t* __restrict__ idx, uint8_t* __restrict__ dist) {
    // one wave per query
    const int lane = threadIdx.x & 63;
    const uint32_t qi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (qi >= n) return;
    const uint64_t b = bits[qi];
    unsigned long long best = ~0ull;
    for (uint32_t i = lane; i < n_codes; i += 64) {
        const unsigned long long k = ((unsigned long long)__popcll(dict[i] ^ b) << 32) | i;
        best = k < best ? k : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if (lane == 0) {
        idx[qi] = n_codes ? (uint32_t)best : 0u;
        dist[qi] = n_codes ? (uint8_t)(best >> 32) : 0xFF;
    }
}

__global__ void k_calc_tau(const uint64_t* __restrict__ dict, uint32_t n_codes, unsigned int* __restrict__ tau) {
    unsigned int best = 255;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_codes; i += gridDim.x * blockDim.x) {
        const uint64_t c = dict[i];
        for (uint32_t j = i + 1; j < n_codes; j++) {
            const unsigned int d = __popcll(c ^ dict[j]);
            best = d < best ? d : best;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned int other = __shfl_xor(best, o); best = other < best ? other : best; }
    if ((threadIdx.x & 63) == 0) atomicMin(tau, best);
}

__global__ void k_selftest_ieee(const double* __restrict__ a, const double* __restrict__ b, uint32_t n, double* __restrict__ sq,
                                double* __restrict__ dv, float* __restrict__ sqf, float* __restrict__ dvf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sq[i] = sqrt(a[i]);
    dv[i] = a[i] / b[i];
    sqf[i] = sqrtf((float)a[i]);
    dvf[i] = (float)a[i] / (float)b[i];
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
size_t decode_lds_bytes(uint32_t S, uint32_t n, uint32_t max_taps) {
    size_t o = ((size_t)S * S + 15) & ~(size_t)15;
    o += (size_t)n * S * 4 + (size_t)n * max_taps * 4 + (size_t)n * 8 + (size_t)n * n;
    return (o + 15) & ~(size_t)15;
}

hipError_t launch_frame_candidates(hipStream_t st, const CandRec* cands, const uint32_t* cand_count, uint32_t n_frames, uint32_t max_cand,
                                   float min_distance, uint16_t* pre_xy, uint16_t* fin_xy, uint32_t* fin_count, uint32_t* work,
                                   unsigned int* work_count, uint32_t S, void* proj, float* big_scratch /* max_cand > frame_cand_lds_slots(): frames x max_cand floats */) {
    const bool big = max_cand > kFrameCandLds;
    if (big && !big_scratch) return hipErrorInvalidValue;
    const size_t lds = big ? (size_t)max_cand + 16 : (size_t)max_cand * 21 + 16;
    if (lds > 48 * 1024) {   // tables grown past the default
        const hipError_t e = hipFuncSetAttribute(big ? reinterpret_cast<const void*>(k_frame_candidates<true>) : reinterpret_cast<const void*>(k_frame_candidates<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (big)
        hipLaunchKernelGGL(k_frame_candidates<true>, dim3(n_frames), dim3(64), lds, st, cands, cand_count, max_cand, min_distance, pre_xy, fin_xy,
                           fin_count, work, work_count, S, reinterpret_cast<ProjRec*>(proj), big_scratch);
    else
        hipLaunchKernelGGL(k_frame_candidates<false>, dim3(n_frames), dim3(64), lds, st, cands, cand_count, max_cand, min_distance, pre_xy, fin_xy,
                           fin_count, work, work_count, S, reinterpret_cast<ProjRec*>(proj), nullptr);
    return hipGetLastError();
}
uint32_t frame_cand_lds_slots() { return kFrameCandLds; }

size_t proj_rec_bytes() { return sizeof(ProjRec); }
size_t weight_table_bytes() { return kWeightTableBytes; }
hipError_t launch_weight_table(hipStream_t st, uint32_t S, uint32_t n, uint32_t max_taps, float* wtab) {
    if ((size_t)n * (max_taps + 2) * 4 > kWeightTableBytes || n > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_weight_table, dim3(1), dim3(64), 0, st, S, n, max_taps, wtab);
    return hipGetLastError();
}

hipError_t launch_decode(hipStream_t st, PixelSrc src, int W, int H, uint32_t first_frame, const uint16_t* fin_xy, const uint32_t* work,
                         const unsigned int* work_count, uint32_t max_cand, uint32_t S, uint32_t n, uint32_t max_taps, const uint64_t* dict,
                         uint32_t n_codes, uint32_t tau, int filter, void* proj, const float* wtab, void* outs, uint8_t* patches, uint32_t patch_cap, uint32_t* per_frame, int grid_blocks, int dbg, int few) {
    ProjRec* recs = reinterpret_cast<ProjRec*>(proj);
    if (dbg > 0 || dbg == -1000) hipLaunchKernelGGL(k_projection, dim3(256), dim3(64), 0, st, fin_xy, work, work_count, S, recs);
    // Four waves sample a candidate: a candidate is a chain of round trips to the frame (19 of them for one wave: 65 us for a lone
    // candidate however idle the chip is, 33 us with four waves).  What follows the sampling runs on all four waves when the
    // batch is small and every microsecond of that chain shows (`few`: at most 64 frames; 36 us for the 337 candidates of 32
    // frames against 39), and on one wave -- no workgroup barriers -- when thousands of candidates are in flight and only the
    // throughput counts (BASELINE config 2: decode stage 0.123 ms against 0.130 with one wave per candidate throughout).
    const int d = dbg == -1000 ? 0 : (dbg < 0 ? -dbg : dbg);
    few = tuning_knob("A3_DECODE_WIDE", few);   // (-DA3_TUNING builds only)
    if (few)
        hipLaunchKernelGGL((k_decode<256, 256>), dim3(grid_blocks), dim3(256), decode_lds_bytes(S, n, max_taps), st, src, W, H, first_frame, fin_xy, work, work_count,
                           max_cand, S, n, max_taps, dict, n_codes, tau, filter, recs, wtab, reinterpret_cast<DecodeOut*>(outs), patches, patch_cap, per_frame, d);
    else
        hipLaunchKernelGGL((k_decode<A3_D_THREADS, 64>), dim3(grid_blocks), dim3(A3_D_THREADS), decode_lds_bytes(S, n, max_taps), st, src, W, H, first_frame, fin_xy, work,
                           work_count, max_cand, S, n, max_taps, dict, n_codes, tau, filter, recs, wtab, reinterpret_cast<DecodeOut*>(outs), patches, patch_cap,
                           per_frame, d);
    return hipGetLastError();
}

size_t decode_out_bytes() { return sizeof(DecodeOut); }

hipError_t launch_compact_markers(hipStream_t st, const void* outs, const uint16_t* fin_xy, const uint32_t* fin_count, uint32_t n_frames,
                                  uint32_t first_frame, uint32_t max_cand, a3_marker* markers, uint32_t marker_cap, uint32_t* per_frame,
                                  unsigned int* marker_total, unsigned int* err_flags, const uint32_t* cand_count, unsigned int* cand_pre_total) {
    if (first_frame == 0 && n_frames <= kCompactParMax)
        hipLaunchKernelGGL(k_compact_markers_par, dim3((n_frames + 3) / 4), dim3(256), 0, st, reinterpret_cast<const DecodeOut*>(outs), fin_xy, fin_count,
                           n_frames, max_cand, markers, marker_cap, per_frame, marker_total, err_flags, cand_count, cand_pre_total);
    else
        hipLaunchKernelGGL(k_compact_markers, dim3(1), dim3(256), 0, st, reinterpret_cast<const DecodeOut*>(outs), fin_xy, fin_count, n_frames,
                           first_frame, max_cand, markers, marker_cap, per_frame, marker_total, err_flags, cand_count, cand_pre_total);
    return hipGetLastError();
}

hipError_t launch_pack_detections(hipStream_t st, const a3_marker* markers, const a3_pose* poses, const uint32_t* per_frame, uint32_t n_frames,
                                  uint32_t first_frame_global, uint32_t maxm, void* dst, unsigned int* overflow) {
    hipLaunchKernelGGL(k_pack_detections, dim3((n_frames + 3) / 4), dim3(256), 0, st, markers, poses, per_frame, n_frames, first_frame_global, maxm,
                       reinterpret_cast<uint32_t*>(dst), overflow);
    return hipGetLastError();
}

hipError_t launch_debug_rotate_bits(hipStream_t st, const uint8_t* in, uint32_t n, uint32_t r, uint8_t* out) {
    hipLaunchKernelGGL(k_debug_rotate_bits, dim3(1), dim3(256), 0, st, in, n, r, out);
    return hipGetLastError();
}

hipError_t launch_pose(hipStream_t st, const uint32_t* corners, uint32_t corner_stride, const float* norm_pts, uint32_t n,
                       const unsigned int* n_dev, int mode, float marker_size_mm, float iw, float ih, float fx, float fy, float cx, float cy,
                       a3_pose* out) {
    hipLaunchKernelGGL(k_pose, dim3((n + 63) / 64), dim3(64), 0, st, corners, corner_stride, norm_pts, n, n_dev, mode, marker_size_mm, iw, ih,
                       fx, fy, cx, cy, out);
    return hipGetLastError();
}

hipError_t launch_find_nearest(hipStream_t st, const uint64_t* dict, uint32_t n_codes, const uint64_t* bits, uint32_t n, uint32_t* idx,
                               uint8_t* dist) {
    hipLaunchKernelGGL(k_find_nearest, dim3((n * 64 + 255) / 256), dim3(256), 0, st, dict, n_codes, bits, n, idx, dist);
    return hipGetLastError();
}

hipError_t launch_calc_tau(hipStream_t st, const uint64_t* dict, uint32_t n_codes, unsigned int* tau) {
    hipLaunchKernelGGL(k_calc_tau, dim3(64), dim3(256), 0, st, dict, n_codes, tau);
    return hipGetLastError();
}

hipError_t launch_selftest(hipStream_t st, const double* a, const double* b, uint32_t n, double* sq, double* dv, float* sqf, float* dvf) {
    hipLaunchKernelGGL(k_selftest_ieee, dim3((n + 255) / 256), dim3(256), 0, st, a, b, n, sq, dv, sqf, dvf);
    return hipGetLastError();
}

}  // namespace a3
