// k_threshold_big.hip -- threshold windows 8..31 (adaptive_threshold(&grey, R), src/aruco.rs:61 with DetectorConfig.threshold_window
// above the default 7), fused like K1: pixels -> grey -> (2R+1)^2 box sums -> compare -> packed bits in one pass over the frames.
//
// k_grey_threshold7 (k_threshold_k1.h) keeps a ring of the last 2R+1 rows of horizontal sums in registers and adds packed u16 pairs;
// from R = 8 on neither holds: (2R+1)^2 * 256 needs 18 bits, and 31 rows of anything per lane do not fit a register file.  Here
//   * the ring holds GREY rows (16 bytes per lane and row), in two stages: the R rows behind the newest one in registers, as a
//     shift register (4 R full-rate moves per row; the oldest of them is the window's centre row), the R + 1 rows behind those in
//     LDS (1 KB per row and wave, slot = row mod (R + 1), indexed at run time) -- so the row loop is unrolled only over the load
//     queue (3 rows), the LDS a wave needs stays below 16 KB + its parked result bits, and a CU holds eight waves for every radius
//     (with the whole ring in LDS it held 7 .. 4, unevenly spread over the SIMDs: 1.5 .. 2.3 x the time of window 7);
//   * vertical first: V[x] = sum of the window's 2R+1 rows of column x (at most 31 * 255: packed u16 pairs, updated with the new
//     row's bytes minus those of the row that leaves, which the ring hands back);
//   * horizontal on V, in 32 bits: v_dot2_u32_u16 with weights (1, 1) adds both halves of a dword of V to a sum, weights (1, 0) /
//     (0, 1) one half, v_dot2_i32_i16 with -1 subtracts: two chains (pixels 0 and 8) of R + 1 instructions to start and two per
//     slide step; the 2R columns beside the lane's 16 come from the neighbouring lanes (wave shifts, as in K1);
//   * compare per pixel in 32 bits: d = S - (L + 1) * area, the result bit is d's sign, shifted in with one v_alignbit.
// Radii 16..31 (round 5) are the same kernel with TWO lanes of apron on either side of a wave (a window reaches 31 columns: 60 lanes
// own output instead of 62), E[] 80 columns wide, and one wave per SIMD -- 31 grey rows in registers (the compiler parks part of them in
// AGPRs) + 32 KB of ring per wave in LDS leave room for four waves per CU.
// Lanes, strips (2048 waves of 270 rows for 256 frames of 1920x1080), directions, the load queue and the XCD mapping are K1's, and so
// are its two ways of reading a row: FAST (16-byte aligned pointers and strides, W % 16 == 0: unconditional vector loads from clamped
// addresses) and the per-pixel loads of any other layout (load_raw).
#include "k_threshold_k1.h"

namespace a3 {

static_assert(T_LPX == 16, "the ring kernel is written for 16 pixels per lane");

typedef unsigned short a3_us2 __attribute__((ext_vector_type(2)));
typedef short a3_s2 __attribute__((ext_vector_type(2)));
// acc + E.lo * wlo + E.hi * whi with weights in {0, 1} (unsigned) resp. {-1, 0, 1} (signed); E's halves are below 2^15
__device__ __forceinline__ uint32_t dot_add(uint32_t e, uint32_t w, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(a3_us2, e), __builtin_bit_cast(a3_us2, w), acc, false);
}
__device__ __forceinline__ uint32_t dot_signed(uint32_t e, uint32_t w, uint32_t acc) {
    return (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(a3_s2, e), __builtin_bit_cast(a3_s2, w), (int)acc, false);
}

// E[k] = columns 2k - EB (low half) and 2k - EB + 1 (high half): the neighbouring lanes' 16 columns each (EB = 16: one lane on either
// side, 24 dwords; EB = 32: two, 40 dwords) and the lane's own.
// sum of the columns lo .. hi (inclusive, -EB <= lo <= hi <= 15 + EB) added to acc; every index is a compile-time constant
template <int EB, int LO, int HI>
__device__ __forceinline__ uint32_t sum_cols(const uint32_t* E, uint32_t acc) {
#pragma unroll
    for (int k = (LO + EB) >> 1; k <= (HI + EB) >> 1; k++) {
        const bool lo_in = 2 * k - EB >= LO, hi_in = 2 * k - EB + 1 <= HI;
        acc = dot_add(E[k], (lo_in ? 1u : 0u) | (hi_in ? 0x10000u : 0u), acc);
    }
    return acc;
}
// one column of E: value of column c
template <int EB, int C>
__device__ __forceinline__ uint32_t plus_col(const uint32_t* E, uint32_t acc) { return dot_add(E[(C + EB) >> 1], ((C + EB) & 1) ? 0x10000u : 1u, acc); }
template <int EB, int C>
__device__ __forceinline__ uint32_t minus_col(const uint32_t* E, uint32_t acc) { return dot_signed(E[(C + EB) >> 1], ((C + EB) & 1) ? 0xFFFF0000u : 0xFFFFu, acc); }

template <int EB, int R, int X>
__device__ __forceinline__ void slide_chain(const uint32_t* E, uint32_t* S) {   // S[X + 1 .. X + 7] from S[X]
    if constexpr (X % 8 != 7) {
        S[X + 1] = minus_col<EB, X - R>(E, plus_col<EB, X + R + 1>(E, S[X]));
        slide_chain<EB, R, X + 1>(E, S);
    }
}

// The ring's split.  RA rows (the youngest) in registers, NB = 2R + 1 - RA in LDS.  Two waves per SIMD -- eight per CU -- are worth a
// factor 1.8 (measured: 0.41 against 0.72 ms for radius 16), and eight waves fit a CU's LDS with at most 19 KB of ring each: up to radius
// 18 the split is R | R + 1; beyond, as long as the registers hold the rest without scratch (115 + 4 RA VGPRs for RGB with vector loads,
// 12 more for four-byte pixels, 25 fewer for L8, ~40 more with per-pixel loads), the LDS part stays at 19 rows and the register part
// grows (1.26-1.45 x of window 7); the largest radii fall back to R | R + 1 and one wave per SIMD (2.4-2.6 x).
constexpr int kRingLdsRows8 = 19;
constexpr int ring_reg_rows(int R, bool fast, int fmt) {   // (the limits: the largest radius whose instantiation needs no scratch)
    const int last8 = !fast ? 19 : (fmt == A3_FMT_L8 ? 28 : (fmt == A3_FMT_RGB8 ? 26 : 24));
    return R + 1 <= kRingLdsRows8 ? R : (R <= last8 ? 2 * R + 1 - kRingLdsRows8 : R);
}
// apron lanes on either side of a wave and the output columns left to it
constexpr int ring_halo_lanes(int R) { return R > 15 ? 2 : 1; }
constexpr int ring_out_cols(int R) { return (64 - 2 * ring_halo_lanes(R)) * T_LPX; }

// grid: 8 * ceil(frames / 8) * strips_x * strips_y workgroups of one wave; dynamic LDS: (R + 1) KB of ring + (flush_rows + T_PF) * 128 B
template <int FMT, int R, bool FAST>
__global__ __launch_bounds__(64, 2 * R + 1 - ring_reg_rows(R, FAST, FMT) <= kRingLdsRows8 ? 2 : 1) void k_grey_threshold_ring(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride, int W, int H,
                                                                int rows_per_wave, int strips_y, int n_frames, uint8_t* __restrict__ grey,
                                                                uint8_t* __restrict__ bits, int flush_rows, int aligned_in) {
    static_assert(R >= 8 && R <= 31, "radii 8..31: one (..15) or two neighbouring lanes' 16 columns cover the window's reach");
    constexpr int RA = ring_reg_rows(R, FAST, FMT), NB = 2 * R + 1 - RA, PF = A3_T_PF;   // RA / NB: rows of the ring's first (registers) and second stage (LDS)
    static_assert(RA >= R && NB >= 1, "the window's centre row (R iterations old) is still in registers");
    constexpr int HL = ring_halo_lanes(R), EB = 16 * HL, OUT = ring_out_cols(R);
    // LDS admits eight waves per CU; they must sit two on every SIMD.  Below 169 VGPRs a SIMD takes three, the eight are then
    // spread unevenly (3 + 3 + 2 + 0 at worst), and a kernel bound by its instruction issue runs at the pace of the fullest SIMD (radii 8..13
    // compile to 140 .. 166 VGPRs and took 1.4 x window 7 that way, 14 and 15 -- 170 and 175 -- 1.17 x): the clobber lifts the
    // allocation above the line for every radius.
    if constexpr (R <= 15) asm volatile("" ::: "v176");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    uint4* s_ring = reinterpret_cast<uint4*>(s_raw);                                   // [NB][64]
    uint16_t* s_out = reinterpret_cast<uint16_t*>(s_raw + (size_t)NB * 1024);          // [flush_rows + PF][64]
    const int lane = threadIdx.x;
    // every strip of a frame on one XCD, top to bottom (k_grey_threshold7's map_by_frame)
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int strips_x = (W + OUT - 1) / OUT, per_frame = strips_x * strips_y, idx = kb % per_frame;
    const int f = (kb / per_frame) * 8 + xcd;
    if (f >= n_frames) return;
    const int sx = idx / strips_y, sy = idx % strips_y;
    const uint8_t* frame = pixels + (size_t)f * frame_stride;
    uint8_t* gout = grey + (size_t)f * W * H;
    const bool write_grey = grey != nullptr;
    const size_t bpr = (size_t)words_per_row((uint32_t)W) * 8;
    uint8_t* bout = bits + (size_t)f * bpr * H;

    const int x0 = sx * OUT - HL * T_LPX + T_LPX * lane;
    const int y_begin = sy * rows_per_wave, y_end = min(H, y_begin + rows_per_wave);
    const bool owner = lane >= HL && lane <= 63 - HL && x0 < W;
    const bool lane_in = x0 >= 0 && x0 + T_LPX <= W;   // (FAST, W % 16 == 0: a lane's columns are all inside or all outside)

    // clipped window widths of the lane's columns (0 outside the image: the comparison then fails), 8 bits each
    uint32_t axp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < T_LPX; i++) {
        const int x = x0 + i;
        int a = 0;
        if (x >= 0 && x < W) a = min(x + R, W - 1) - max(x - R, 0) + 1;
        axp[i >> 2] |= (uint32_t)a << (8 * (i & 3));
    }
    uint32_t area[T_LPX];
    uint32_t ay_cur = 0;
    uint32_t V[8];             // column sums over the window's rows: columns 2k | 2k + 1 << 16
#pragma unroll
    for (int i = 0; i < 8; i++) V[i] = 0u;
    for (int s = 0; s < NB; s++) s_ring[s * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t A[RA][T_NG];      // the ring's first stage: A[i] = the grey row i + 1 iterations old; A[R - 1] is the window's centre row
#pragma unroll
    for (int i = 0; i < RA; i++) {
#pragma unroll
        for (int j = 0; j < T_NG; j++) A[i][j] = 0u;
    }

    const int dir = (sy & 1) ? -1 : 1;
    const int r_first = dir > 0 ? y_begin - R : y_end - 1 + R, n_rows = (y_end - y_begin) + 2 * R;
    RawRow<FMT> q[PF];
    const uint8_t* lane_ptr = frame + (size_t)(lane_in ? x0 : 0) * RawRow<FMT>::BPP;
    auto issue = [&](int r, RawRow<FMT>& dst) {
        if constexpr (FAST) load_vec<FMT>(lane_ptr + (size_t)(uint32_t)min(max(r, 0), H - 1) * row_stride, dst);
        else load_raw<FMT>(frame, row_stride, x0, r, W, H, aligned_in != 0, dst);   // (zeros where the image is not)
    };
#pragma unroll
    for (int k = 0; k < PF; k++) issue(r_first + dir * k, q[k]);

    int slot = 0;   // LDS slot of the row that leaves the window (2R + 1 iterations old); the centre row takes its place
    int n_buf = 0, y_buf0 = 0;
    const int n_iter = ((n_rows + PF - 1) / PF) * PF;
    for (int base = 0; base < n_iter; base += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int it = base + k;
            const int r = r_first + dir * it;
            uint32_t g[T_NG];
            grey_row<FMT>(q[k], g);
            issue(r + dir * PF, q[k]);
            if constexpr (FAST) {   // what the clamped address fetched for a lane or row outside the image is discarded
                if (!(lane_in && r >= 0 && r < H)) {
#pragma unroll
                    for (int i = 0; i < T_NG; i++) g[i] = 0u;
                }
            }
            if (write_grey && owner && r >= y_begin && r < y_end) {
                uint8_t* dst = gout + (size_t)r * W + x0;
                if constexpr (FAST) *reinterpret_cast<uint4*>(dst) = make_uint4(g[0], g[1], g[2], g[3]);
                else {
#pragma unroll
                    for (int i = 0; i < T_LPX; i++) if (x0 + i < W) dst[i] = (uint8_t)(g[i >> 2] >> (8 * (i & 3)));
                }
            }
            // the ring: the oldest register row (RA iterations old; the centre row when RA == R) leaves the registers for the LDS slot of
            // the row that leaves the window (2R + 1 iterations old; zeros at first), the others move up by one
            const uint4 old = s_ring[slot * 64 + lane];
            const uint32_t gc[4] = {A[R - 1][0], A[R - 1][1], A[R - 1][2], A[R - 1][3]};
            s_ring[slot * 64 + lane] = make_uint4(A[RA - 1][0], A[RA - 1][1], A[RA - 1][2], A[RA - 1][3]);
            slot = slot + 1 == NB ? 0 : slot + 1;
#pragma unroll
            for (int i = RA - 1; i > 0; i--) {
#pragma unroll
                for (int j = 0; j < T_NG; j++) A[i][j] = A[i - 1][j];
            }
#pragma unroll
            for (int j = 0; j < T_NG; j++) A[0][j] = g[j];
            // V += new - old, column pairs: bytes (0, 1) and (2, 3) of every grey dword as u16 pairs.  The add comes first and every
            // half is a true column sum in 0 .. 31 * 255, so no half borrows from or carries into the other.
            const uint32_t go[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                V[2 * i] = V[2 * i] + __builtin_amdgcn_perm(0u, g[i], 0x0C010C00u) - __builtin_amdgcn_perm(0u, go[i], 0x0C010C00u);
                V[2 * i + 1] = V[2 * i + 1] + __builtin_amdgcn_perm(0u, g[i], 0x0C030C02u) - __builtin_amdgcn_perm(0u, go[i], 0x0C030C02u);
            }
            const int y = r - dir * R;   // the row whose window is now complete; its grey bytes: gc
            if (y < y_begin || y >= y_end) continue;   // wave-uniform
            const uint32_t ay = (uint32_t)(min(y + R, H - 1) - max(y - R, 0) + 1);
            if (ay != ay_cur) {   // wave-uniform; only the first and last R image rows differ from 2R+1
                ay_cur = ay;
#pragma unroll
                for (int j = 0; j < T_LPX; j++) area[j] = mul24((axp[j >> 2] >> (8 * (j & 3))) & 255u, ay);
            }
            // E[]: the column sums of columns -EB .. 15 + EB (only the dwords a window of this radius reaches are fetched)
            uint32_t E[8 + 16 * HL];
            constexpr int KLO = (EB - R) >> 1, KHI = (EB + 15 + R) >> 1;   // first and last dword of E any window touches
#pragma unroll
            for (int i = 0; i < 8; i++) {
                E[8 * HL + i] = V[i];
                const uint32_t l1 = 8 * (HL - 1) + i >= KLO ? wave_from_left(V[i]) : 0u;
                const uint32_t r1 = 8 * (HL + 1) + i <= KHI ? wave_from_right(V[i]) : 0u;
                E[8 * (HL - 1) + i] = l1;
                E[8 * (HL + 1) + i] = r1;
                if constexpr (HL == 2) {
                    E[i] = i >= KLO ? wave_from_left(l1) : 0u;
                    E[32 + i] = 32 + i <= KHI ? wave_from_right(r1) : 0u;
                }
            }
            uint32_t S[T_LPX];
            S[0] = sum_cols<EB, -R, R>(E, 0u);
            S[8] = sum_cols<EB, 8 - R, 8 + R>(E, 0u);
            slide_chain<EB, R, 0>(E, S);
            slide_chain<EB, R, 8>(E, S);
            // white iff S < (L + 1) * area: the sign of S - (L * area + area), shifted in from column 15 down to column 0
            uint32_t acc = 0u;
#pragma unroll
            for (int j = T_LPX - 1; j >= 0; j--) {
                const uint32_t L = (gc[j >> 2] >> (8 * (j & 3))) & 255u;
                const uint32_t d = S[j] - (__umul24(L, area[j]) + area[j]);        // < 2^21 either way (63 * 63 * 256): the sign bit is the borrow
                acc = __builtin_amdgcn_alignbit(acc, d, 31);              // acc * 2 + (d >> 31)
            }
            if (flush_rows <= 0) {
                if (owner) *reinterpret_cast<uint16_t*>(bout + (size_t)y * bpr + (x0 >> 3)) = (uint16_t)acc;
            } else {
                if (n_buf == 0) y_buf0 = y;
                s_out[n_buf * 64 + lane] = (uint16_t)acc;
                n_buf++;
            }
        }
        if (flush_rows > 0 && n_buf >= flush_rows) {   // wave-uniform
            if (owner)
                for (int b = 0; b < n_buf; b++) *reinterpret_cast<uint16_t*>(bout + (size_t)(y_buf0 + dir * b) * bpr + (x0 >> 3)) = s_out[b * 64 + lane];
            n_buf = 0;
        }
    }
    if (flush_rows > 0 && owner)
        for (int b = 0; b < n_buf; b++) *reinterpret_cast<uint16_t*>(bout + (size_t)(y_buf0 + dir * b) * bpr + (x0 >> 3)) = s_out[b * 64 + lane];
}

template <int R>
hipError_t launch_ring(hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n, uint8_t* grey,
                       uint64_t* bits) {
    constexpr int OUT = ring_out_cols(R);
    uint8_t* bin = reinterpret_cast<uint8_t*>(bits);
    if (W % 64 != 0) {  // packed rows end in padding bits that no strip writes
        hipError_t e = hipMemsetAsync(bits, 0, (size_t)words_per_row((uint32_t)W) * 8 * H * n, st);
        if (e != hipSuccess) return e;
    }
    const int aligned_in = ((uintptr_t)pixels % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
    const bool fast = aligned_in && W % 16 == 0 && (uintptr_t)grey % 16 == 0;   // (as in launch_k1)
    // waves per CU: two per SIMD while the ring's LDS part is at most 19 KB per wave (then at least 4 parked rows each), else one
    const int NB = 2 * R + 1 - ring_reg_rows(R, fast, fmt), WPC = NB <= kRingLdsRows8 ? 8 : 4;
    // strips: K1's model (time ~ rounds x (rows per strip + 2R))
    const int strips_x = (W + OUT - 1) / OUT;
    const long long slots = (long long)g_k1_cus * WPC, cols = (long long)strips_x * n;
    int best_sy = 1; double best_cost = 1e300;
    for (int sy = 1; sy <= std::max(1, H / 16); sy++) {
        const int rows = (H + sy - 1) / sy;
        const long long waves = cols * ((H + rows - 1) / rows);
        const double cost = (double)((waves + slots - 1) / slots) * (rows + 2 * R);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_sy = sy; }
    }
    const int rows_per_wave = (H + best_sy - 1) / best_sy;
    const int strips_y = (H + rows_per_wave - 1) / rows_per_wave;
    const int per_wave = (160 * 1024) / WPC - 64;   // (the allocation granule)
    const int flush_rows = std::max(1, std::min({128, rows_per_wave, (per_wave - NB * 1024) / 128 - A3_T_PF}));
    const size_t lds_bytes = (size_t)NB * 1024 + (size_t)(flush_rows + A3_T_PF) * 128;
    if ((per_wave - NB * 1024) / 128 - A3_T_PF < 1) return hipErrorInvalidValue;   // (cannot happen: 19 KB + 4 rows fit an eighth, 32 KB + 60 rows a quarter)
    dim3 grid(8 * (((int)n + 7) / 8) * strips_x * strips_y), block(64);
#define A3_LAUNCH_RING(F)                                                                                                              \
    {                                                                                                                                  \
        if (fast) hipLaunchKernelGGL((k_grey_threshold_ring<F, R, true>), grid, block, lds_bytes, st, pixels, row_stride, frame_stride, W, H, rows_per_wave, strips_y, \
                                     (int)n, grey, bin, flush_rows, aligned_in);                                                       \
        else hipLaunchKernelGGL((k_grey_threshold_ring<F, R, false>), grid, block, lds_bytes, st, pixels, row_stride, frame_stride, W, H, rows_per_wave, strips_y,    \
                                (int)n, grey, bin, flush_rows, aligned_in);                                                            \
    }
    if (fmt == A3_FMT_RGB8) A3_LAUNCH_RING(A3_FMT_RGB8)
    else if (fmt == A3_FMT_RGBA8) A3_LAUNCH_RING(A3_FMT_RGBA8)
    else if (fmt == A3_FMT_BGRA8) A3_LAUNCH_RING(A3_FMT_BGRA8)
    else A3_LAUNCH_RING(A3_FMT_L8)
#undef A3_LAUNCH_RING
    return hipGetLastError();
}

// the radii this file covers (any frame layout)
bool ring_kernel_applies(uint32_t radius, const uint8_t*, size_t, size_t, int) { return radius >= 8 && radius <= 31; }

hipError_t launch_ring_threshold(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H,
                                 uint32_t n, uint8_t* grey, uint64_t* bits) {
    switch (radius) {
        case 8: return launch_ring<8>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 9: return launch_ring<9>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 10: return launch_ring<10>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 11: return launch_ring<11>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 12: return launch_ring<12>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 13: return launch_ring<13>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 14: return launch_ring<14>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 15: return launch_ring<15>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 16: return launch_ring<16>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 17: return launch_ring<17>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 18: return launch_ring<18>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 19: return launch_ring<19>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 20: return launch_ring<20>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 21: return launch_ring<21>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 22: return launch_ring<22>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 23: return launch_ring<23>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 24: return launch_ring<24>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 25: return launch_ring<25>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 26: return launch_ring<26>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 27: return launch_ring<27>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 28: return launch_ring<28>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 29: return launch_ring<29>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 30: return launch_ring<30>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 31: return launch_ring<31>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace a3
