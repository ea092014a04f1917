// k_threshold.hip -- K1: pixels -> grey (into_luma8) -> 15x15 adaptive threshold, fused.
//
// Replaces `image.into_luma8()` + `imageproc::contrast::adaptive_threshold(&grey, 7)`
// (src/aruco.rs:60-61).  Pure integer, so every output byte must equal the oracle's.
//
//   grey  L = (2126 R + 7152 G + 722 B) / 10000          (u32, truncating)
//   white iff L >= floor(sum / area) over the window clipped to the image
//         <=> sum < (L + 1) * area                        (no division)
//
// HBM-bound: 3 B read + 1 B grey + 1/8 B binary per pixel (RGB8): the thresholded image leaves the kernel
// bit-packed (a3_common.h: words_per_row), which is all the contour stage reads.  One 256-thread workgroup
// produces a 240 x 64 tile: 78 rows x 256 columns of pixels are loaded once with 12-byte
// (4-pixel) lane loads, one wave-instruction per image row segment, converted to grey
// bytes in LDS; a separable box sum follows -- horizontal 15-tap sums as u16 (24 bytes in,
// 8 sums out per lane), then vertical sliding sums on packed u16 pairs (15*15*255 < 2^16,
// so two sums share a dword and plain 32-bit adds never carry across).  Out-of-image
// pixels are stored as 0, which makes the unclipped sum equal the clipped one.
#include <cstdlib>
#include <utility>

#include "a3_common.h"

namespace a3 {

constexpr int T_R = 7;               // fast path radius (threshold_window = 7)
constexpr int T_LPX = 16;            // pixels per lane and row
constexpr int T_OUT = 62 * T_LPX;    // 992 output columns per wave (lanes 0 and 63 only feed their neighbours)
#ifndef A3_T_PF
#define A3_T_PF 3
#endif
#ifndef A3_T_WAVES
#define A3_T_WAVES 2
#endif
constexpr int T_PF = A3_T_PF;        // rows of loads kept in flight per lane


// grey of one RGB(A) pixel held in the low 3 bytes of `px` (4th byte ignored): two byte-wise dot products with
// the split weights 2126 = 8*256+78, 7152 = 27*256+240, 722 = 2*256+210, then the exact /10000.
template <bool BGR = false>
__device__ __forceinline__ uint32_t luma_dot(uint32_t px) {
    // byte 0 of px is R (RGB/RGBA) or B (BGRA): the weight bytes swap ends
    const uint32_t lo = __builtin_amdgcn_udot4(px, BGR ? 0x004EF0D2u : 0x00D2F04Eu, 0u, false);
    const uint32_t hi = __builtin_amdgcn_udot4(px, BGR ? 0x00081B02u : 0x00021B08u, 0u, false);
    // l <= 10000*255 < 2^22; floor(l / 10000) == (l * 13743896) >> 37 for every such l (checked exhaustively;
    // 429497 >> 32 is NOT exact), i.e. one full-rate v_mul_hi_u32_u24 plus a shift instead of a quarter-rate
    // 32-bit multiply-high
    const uint32_t l = lo + (hi << 8);
    __builtin_assume(l < (1u << 22));
    return (uint32_t)(((uint64_t)l * 13743896ull) >> 37);
}

template <int FMT> struct RawRow { static constexpr int NDW = FMT == A3_FMT_RGB8 ? 12 : (FMT == A3_FMT_L8 ? 4 : 16); uint32_t d[NDW]; };

// 16 consecutive pixels of row y starting at x0 (a multiple of 16, may be negative or past the image): raw bytes,
// zero where the image is not.  Fully-inside lanes use 16-byte vector loads.
template <int FMT, bool FAST>
__device__ __forceinline__ void load_raw16(const uint8_t* __restrict__ frame, size_t row_stride, int x0, int y, int W, int H, bool aligned,
                                           RawRow<FMT>& r) {
    constexpr int NDW = RawRow<FMT>::NDW;
    constexpr int BPP = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_L8 ? 1 : 4);
#pragma unroll
    for (int i = 0; i < NDW; i++) r.d[i] = 0u;
    if (y < 0 || y >= H || x0 + T_LPX <= 0 || x0 >= W) return;
    const uint8_t* row = frame + (size_t)y * row_stride;
    if (FAST || (aligned && x0 >= 0 && x0 + T_LPX <= W)) {
        const uint4* p = reinterpret_cast<const uint4*>(row + (size_t)x0 * BPP);
#pragma unroll
        for (int i = 0; i < NDW / 4; i++) { const uint4 v = p[i]; r.d[4 * i] = v.x; r.d[4 * i + 1] = v.y; r.d[4 * i + 2] = v.z; r.d[4 * i + 3] = v.w; }
        return;
    }
    for (int i = 0; i < T_LPX; i++) {
        const int x = x0 + i;
        if (x < 0 || x >= W) continue;
        for (int c = 0; c < BPP; c++) {
            const int byte = i * BPP + c;
            r.d[byte >> 2] |= (uint32_t)row[(size_t)x * BPP + c] << (8 * (byte & 3));
        }
    }
}

// raw row -> 16 grey bytes in 4 dwords
template <int FMT>
__device__ __forceinline__ void grey16(const RawRow<FMT>& r, uint32_t g[4]) {
    if constexpr (FMT == A3_FMT_L8) {
#pragma unroll
        for (int i = 0; i < 4; i++) g[i] = r.d[i];
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t l[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = 4 * q + j;
                uint32_t px;
                if constexpr (FMT == A3_FMT_RGBA8 || FMT == A3_FMT_BGRA8) px = r.d[i];
                else {
                    const int byte = 3 * i, k = byte >> 2, sh = 8 * (byte & 3);   // compile-time after unrolling
                    px = sh == 0 ? r.d[k] : __builtin_amdgcn_alignbit(k + 1 < 12 ? r.d[k + 1] : 0u, r.d[k], sh);
                }
                l[j] = luma_dot<FMT == A3_FMT_BGRA8>(px);
            }
            g[q] = l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24);
        }
    }
}

// full-rate 24-bit multiply (the compiler prefers the quarter-rate v_mul_lo_u32 when one operand is scalar)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// s +/- one 16-bit half of a packed pair, operand-selected (SDWA): no separate unpack instruction
template <int HALF> __device__ __forceinline__ uint32_t add_half(uint32_t s, uint32_t p) {
    uint32_t r;
    if constexpr (HALF == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(s), "v"(p));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(s), "v"(p));
    return r;
}
template <int HALF> __device__ __forceinline__ uint32_t sub_half(uint32_t s, uint32_t p) {
    uint32_t r;
    if constexpr (HALF == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(s), "v"(p));
    else asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(s), "v"(p));
    return r;
}
// bits = (bits << 1) | (a < b): compare into vcc, then add-with-carry
__device__ __forceinline__ uint32_t shift_in_lt(uint32_t bits, uint32_t a, uint32_t b) {
    asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");
    return bits;
}

__device__ __forceinline__ uint32_t wave_from_left(uint32_t v) {   // lane i <- lane i-1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_from_right(uint32_t v) {  // lane i <- lane i+1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}

// One wave walks down a strip: lane l owns columns xs - 16 + 16 l .. + 15 and keeps, in registers, the last 15 grey rows
// of those 16 columns plus their running vertical sums (u16 pairs: 15*255 < 2^16).  Per output row it needs 7 column
// sums from each neighbouring lane (wave shifts, no LDS), slides a 15-wide window over 30 column sums and compares
// sum < (L+1)*area.  No LDS, no barriers; T_PF rows of loads stay in flight per lane.
// grid: 8 * ceil(frames * strips_x / 8) * strips_y workgroups of one wave.
template <int FMT, bool FAST>
__global__ __launch_bounds__(64, A3_T_WAVES) void k_grey_threshold7(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride,
                                                        int W, int H, int rows_per_wave, int strips_y, int n_pairs,
                                                        uint8_t* __restrict__ grey,
                                                        uint8_t* __restrict__ bits, int aligned_in, int aligned_out) {
    const int lane = threadIdx.x;
    // XCD-aware block -> strip mapping.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), each
    // with its own L2.  Vertically adjacent strips of one column share 14 rows of input, so all strips of a
    // (frame, column) pair are given to ONE XCD, in top-to-bottom order: the shared rows are then L2 hits instead of a
    // second trip over the fabric.  (Placement only affects speed; any mapping is correct.)
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int pair = (k / strips_y) * 8 + xcd, sy = k % strips_y;
    if (pair >= n_pairs) return;
    const int strips_x = (W + T_OUT - 1) / T_OUT;
    const int sx = pair % strips_x;
    const uint32_t f = pair / strips_x;
    const uint8_t* frame = pixels + (size_t)f * frame_stride;
    uint8_t* gout = grey + (size_t)f * W * H;
    const bool write_grey = grey != nullptr;
    const size_t bpr = (size_t)words_per_row((uint32_t)W) * 8;
    uint8_t* bout = bits + (size_t)f * bpr * H;

    const int x0 = sx * T_OUT - T_LPX + T_LPX * lane;
    const int y_begin = sy * rows_per_wave, y_end = min(H, y_begin + rows_per_wave);
    const bool owner = lane >= 1 && lane <= 62 && x0 < W;   // lanes that write results

    // clipped window widths of the lane's 16 columns, 4 bits each (0 past the right edge: the comparison then fails)
    uint32_t axp[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < T_LPX; i++) {
        const int x = x0 + i;
        int a = 0;
        if (x >= 0 && x < W) a = min(x + T_R, W - 1) - max(x - T_R, 0) + 1;
        axp[i >> 3] |= (uint32_t)a << (4 * (i & 7));
    }

    uint32_t area[T_LPX];   // clipped window area of each column for the current row's window height
    uint32_t ay_cur = 0;
    uint32_t ring[15][4];   // the last 15 grey rows; row `it` lives in slot it % 15 (static: the row loop is unrolled 15x)
    uint32_t VE[4] = {0, 0, 0, 0}, VO[4] = {0, 0, 0, 0};   // column sums, VE[i] = v(4i) | v(4i+2)<<16, VO[i] = v(4i+1) | v(4i+3)<<16
#pragma unroll
    for (int k = 0; k < 15; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) ring[k][i] = 0u;

    // Odd strips walk upwards.  Strip k (going down) and strip k+1 (going up) then both reach their common boundary --
    // the 14 rows each must also read from the other's territory -- at the END of their runs, and strips k+1 and k+2
    // both START at theirs: neighbours touch the shared rows at about the same time, so the second one finds them in
    // the XCD's L2 instead of fetching them again from HBM.  The box filter is symmetric, so direction only changes
    // the order rows enter and leave the window.
    const int dir = (sy & 1) ? -1 : 1;
    const int r_first = dir > 0 ? y_begin - T_R : y_end - 1 + T_R, n_rows = (y_end - y_begin) + 2 * T_R;
    RawRow<FMT> q[T_PF];
    // FAST: every load is unconditional (row and column clamped into the image) so that the loop body has no branch
    // around a load and the compiler can keep T_PF rows in flight with counted waits; what the clamped address
    // fetched for an outside lane/row is discarded by zeroing the grey below.
    constexpr int BPPK = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_L8 ? 1 : 4);
    const bool lane_in = x0 >= 0 && x0 + T_LPX <= W;
    const uint8_t* lane_ptr = frame + (size_t)(lane_in ? x0 : 0) * BPPK;
    auto issue = [&](int r, RawRow<FMT>& dst) {
        if constexpr (FAST) {
            const int rc = min(max(r, 0), H - 1);
            const uint4* p = reinterpret_cast<const uint4*>(lane_ptr + (size_t)(uint32_t)rc * row_stride);
#pragma unroll
            for (int i = 0; i < RawRow<FMT>::NDW / 4; i++) { const uint4 v = p[i]; dst.d[4 * i] = v.x; dst.d[4 * i + 1] = v.y; dst.d[4 * i + 2] = v.z; dst.d[4 * i + 3] = v.w; }
        } else {
            load_raw16<FMT, false>(frame, row_stride, x0, r, W, H, aligned_in != 0, dst);
        }
    };
#pragma unroll
    for (int k = 0; k < T_PF; k++) issue(r_first + dir * k, q[k]);

    static_assert(15 % T_PF == 0, "the load queue index must be static inside the 15x unrolled body");
    // FAST: whole blocks of 15 rows and no exit test inside the unrolled body (rows past the strip are clamped loads whose
    // results are never stored), so the body is straight-line code apart from the store predicates
    const int n_iter = FAST ? ((n_rows + 14) / 15) * 15 : n_rows;
    for (int base = 0; base < n_iter; base += 15) {
#pragma unroll
        for (int k15 = 0; k15 < 15; k15++) {
            const int k = k15 % T_PF;
            const int it = base + k15;
            if (!FAST && it >= n_iter) break;
            const int r = r_first + dir * it;
            uint32_t g[4];
            grey16<FMT>(q[k], g);                                   // consumes the row loaded T_PF iterations ago ...
            if (FAST || it + T_PF < n_rows) issue(r + dir * T_PF, q[k]);  // ... and its registers take the next load at once
            if constexpr (FAST) {
                if (!(lane_in && r >= 0 && r < H)) { g[0] = 0u; g[1] = 0u; g[2] = 0u; g[3] = 0u; }
            }
            // Detection.grey of the rows this wave owns -- only when somebody reads the plane (debug taps); the decode stage
            // otherwise recomputes the few grey levels it samples from the frame itself
            if (write_grey && owner && r >= y_begin && r < y_end) {
                uint8_t* dst = gout + (size_t)r * W + x0;
                if (FAST || (aligned_out && x0 + T_LPX <= W)) *reinterpret_cast<uint4*>(dst) = make_uint4(g[0], g[1], g[2], g[3]);
                else {
#pragma unroll
                    for (int i = 0; i < T_LPX; i++) if (x0 + i < W) dst[i] = (uint8_t)(g[i >> 2] >> (8 * (i & 3)));
                }
            }
            // vertical sliding sums: + newest row, - the row that leaves the 15-row window
            // VE = (v0 | v2<<16), VO = (v1 | v3<<16): add the new row's bytes, subtract the leaving row's (the one that
            // entered 15 iterations ago); each 16-bit half is updated in place by one byte-selecting SDWA instruction, the
            // eight independent registers advance in lock step so that no dependent pair is back to back
#define A3_SDWA(OP, DST, SRC, W, B) asm(OP " %0, %0, %1 dst_sel:WORD_" #W " dst_unused:UNUSED_PRESERVE src0_sel:WORD_" #W " src1_sel:BYTE_" #B : "+v"(DST) : "v"(SRC))
#pragma unroll
            for (int i = 0; i < 4; i++) { A3_SDWA("v_add_u32_sdwa", VE[i], g[i], 0, 0); A3_SDWA("v_add_u32_sdwa", VO[i], g[i], 0, 1); }
#pragma unroll
            for (int i = 0; i < 4; i++) { A3_SDWA("v_add_u32_sdwa", VE[i], g[i], 1, 2); A3_SDWA("v_add_u32_sdwa", VO[i], g[i], 1, 3); }
#pragma unroll
            for (int i = 0; i < 4; i++) { A3_SDWA("v_sub_u32_sdwa", VE[i], ring[k15][i], 0, 0); A3_SDWA("v_sub_u32_sdwa", VO[i], ring[k15][i], 0, 1); }
#pragma unroll
            for (int i = 0; i < 4; i++) { A3_SDWA("v_sub_u32_sdwa", VE[i], ring[k15][i], 1, 2); A3_SDWA("v_sub_u32_sdwa", VO[i], ring[k15][i], 1, 3); }
#undef A3_SDWA
#pragma unroll
            for (int i = 0; i < 4; i++) ring[k15][i] = g[i];
            const uint32_t* centre = ring[(k15 + 8) % 15];   // the row 7 iterations old: the one being thresholded

            const int y = r - dir * T_R;   // the row whose window is now complete
            if (y < y_begin || y >= y_end) continue;   // wave-uniform
            // 30 column sums: 7 from the left lane, own 16, 7 from the right lane
            const uint32_t le2 = wave_from_left(VE[2]), lo2 = wave_from_left(VO[2]), le3 = wave_from_left(VE[3]), lo3 = wave_from_left(VO[3]);
            const uint32_t re0 = wave_from_right(VE[0]), ro0 = wave_from_right(VO[0]), re1 = wave_from_right(VE[1]), ro1 = wave_from_right(VO[1]);
            uint32_t e[30];   // the 30 column sums: 7 from the left lane, own 16, 7 from the right lane
            e[0] = lo2 & 0xFFFFu; e[1] = le2 >> 16; e[2] = lo2 >> 16;
            e[3] = le3 & 0xFFFFu; e[4] = lo3 & 0xFFFFu; e[5] = le3 >> 16; e[6] = lo3 >> 16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                e[7 + 4 * i] = VE[i] & 0xFFFFu; e[8 + 4 * i] = VO[i] & 0xFFFFu; e[9 + 4 * i] = VE[i] >> 16; e[10 + 4 * i] = VO[i] >> 16;
            }
            e[23] = re0 & 0xFFFFu; e[24] = ro0 & 0xFFFFu; e[25] = re0 >> 16; e[26] = ro0 >> 16;
            e[27] = re1 & 0xFFFFu; e[28] = ro1 & 0xFFFFu; e[29] = re1 >> 16;
            const uint32_t ay = (uint32_t)(min(y + T_R, H - 1) - max(y - T_R, 0) + 1);
            if (ay != ay_cur) {   // wave-uniform; only the first and last 7 image rows differ from 15
                ay_cur = ay;
#pragma unroll
                for (int i = 0; i < T_LPX; i++) area[i] = mul24((axp[i >> 3] >> (4 * (i & 7))) & 15u, ay);
            }
            // window of pixel 15 first, then slide left: the comparison bits are shifted in from the top
            uint32_t S = 0;
#pragma unroll
            for (int j = 15; j < 30; j++) S += e[j];
            uint32_t outb = 0;
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                if (i < 15) S += e[i] - e[i + 15];
                const uint32_t gv = (centre[i >> 2] >> (8 * (i & 3))) & 255u;
                outb = shift_in_lt(outb, S, mad24(gv, area[i], area[i]));   // S < (L+1)*area
            }
            if (owner) *reinterpret_cast<uint16_t*>(bout + (size_t)y * bpr + (x0 >> 3)) = (uint16_t)outb;
        }
    }
}

// ---- generic radius: plain two-kernel path (correct for any threshold_window, not tuned) ----
template <int FMT>
__global__ void k_grey_generic(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride, int W, int H,
                               uint8_t* __restrict__ grey) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint32_t f = blockIdx.z;
    constexpr int BPP = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_L8 ? 1 : 4);
    const uint8_t* p = pixels + (size_t)f * frame_stride + (size_t)y * row_stride + (size_t)x * BPP;
    grey[(size_t)f * W * H + (size_t)y * W + x] =
        BPP == 1 ? p[0] : (uint8_t)(FMT == A3_FMT_BGRA8 ? luma_of(p[2], p[1], p[0]) : luma_of(p[0], p[1], p[2]));
}

// one wave per packed word: 64 consecutive pixels, result gathered with a ballot
__global__ __launch_bounds__(64) void k_threshold_generic(const uint8_t* __restrict__ grey, int W, int H, int radius, uint64_t* __restrict__ bits) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    const uint8_t* g = grey + (size_t)blockIdx.z * W * H;
    bool white = false;
    if (x < W) {
        const int ylo = y > radius ? y - radius : 0, yhi = (long long)y + radius < H - 1 ? y + radius : H - 1;
        const int xlo = x > radius ? x - radius : 0, xhi = (long long)x + radius < W - 1 ? x + radius : W - 1;
        unsigned long long sum = 0;
        for (int yy = ylo; yy <= yhi; yy++)
            for (int xx = xlo; xx <= xhi; xx++) sum += g[(size_t)yy * W + xx];
        const unsigned long long area = (unsigned long long)(yhi - ylo + 1) * (unsigned long long)(xhi - xlo + 1);
        const unsigned long long gv = g[(size_t)y * W + x];
        white = sum < (gv + 1) * area;
    }
    const unsigned long long m = __ballot(white);
    const size_t wpr = words_per_row((uint32_t)W);
    if (threadIdx.x == 0) bits[(size_t)blockIdx.z * wpr * H + (size_t)y * wpr + blockIdx.x] = m;
}

// ---- host launcher -------------------------------------------------------------------------
hipError_t launch_grey_threshold(hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H,
                                 uint32_t n, uint32_t radius, uint8_t* grey, uint64_t* bits) {
    if (radius == (uint32_t)T_R) {
        const int aligned_in = ((uintptr_t)pixels % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
        const int aligned_out = (W % 16 == 0) && ((uintptr_t)grey % 16 == 0);
        uint8_t* bin = reinterpret_cast<uint8_t*>(bits);
        if (W % 64 != 0) {  // packed rows end in padding bits that no tile writes
            hipError_t e = hipMemsetAsync(bits, 0, (size_t)words_per_row((uint32_t)W) * 8 * H * n, st);
            if (e != hipSuccess) return e;
        }
        // rows per wave: enough waves to fill the chip several times over, few enough that the 14 extra rows each
        // wave reads above/below its strip stay a small fraction
        const int strips_x = (W + T_OUT - 1) / T_OUT;
        int rows_per_wave = 106;   // 106 + 14 halo rows = 8 blocks of 15
        if (const char* ev = getenv("A3_ROWS_PER_WAVE")) rows_per_wave = atoi(ev) > 0 ? atoi(ev) : rows_per_wave;  // tuning knob
        while (rows_per_wave > 31 && (long long)strips_x * ((H + rows_per_wave - 1) / rows_per_wave) * n < 3 * 256 * 8) rows_per_wave -= 15;
        const int strips_y = (H + rows_per_wave - 1) / rows_per_wave;
        const int n_pairs = (int)n * strips_x;
        dim3 grid(8 * ((n_pairs + 7) / 8) * strips_y), block(64);
        const bool fast = aligned_in && aligned_out;   // W % 16 == 0: a lane's 16 pixels are all inside or all outside
#define A3_LAUNCH_K1(F, B) hipLaunchKernelGGL((k_grey_threshold7<F, B>), grid, block, 0, st, pixels, row_stride, frame_stride, W, H, \
                                              rows_per_wave, strips_y, n_pairs, grey, bin, aligned_in, aligned_out)
        if (fmt == A3_FMT_RGB8) { if (fast) A3_LAUNCH_K1(A3_FMT_RGB8, true); else A3_LAUNCH_K1(A3_FMT_RGB8, false); }
        else if (fmt == A3_FMT_RGBA8) { if (fast) A3_LAUNCH_K1(A3_FMT_RGBA8, true); else A3_LAUNCH_K1(A3_FMT_RGBA8, false); }
        else if (fmt == A3_FMT_BGRA8) { if (fast) A3_LAUNCH_K1(A3_FMT_BGRA8, true); else A3_LAUNCH_K1(A3_FMT_BGRA8, false); }
        else { if (fast) A3_LAUNCH_K1(A3_FMT_L8, true); else A3_LAUNCH_K1(A3_FMT_L8, false); }
#undef A3_LAUNCH_K1
        return hipGetLastError();
    }
    dim3 block(64), gridg((W + 63) / 64, H, n), grid1(words_per_row((uint32_t)W), H, n);
    if (fmt == A3_FMT_RGB8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGB8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else if (fmt == A3_FMT_RGBA8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGBA8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else if (fmt == A3_FMT_BGRA8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_BGRA8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else hipLaunchKernelGGL(k_grey_generic<A3_FMT_L8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    hipLaunchKernelGGL(k_threshold_generic, grid1, block, 0, st, grey, W, H, (int)radius, bits);
    return hipGetLastError();
}

}  // namespace a3
