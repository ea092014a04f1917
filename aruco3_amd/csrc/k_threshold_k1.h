// k_threshold_k1.h -- the register-resident threshold kernel (template on format, radius, load-queue depth) and its launcher
// template: shared by k_threshold.hip (radius 7, the default) and k_threshold_r1.hip / _r2.hip (radii 1..3 / 4..6), which exist
// only so that the instantiations compile in parallel.  Everything about the kernel is described in k_threshold.hip's header.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "a3_common.h"

namespace a3 {

constexpr int T_R = 7;               // fast path radius (threshold_window = 7)
#ifdef A3_TUNING
static __device__ unsigned long long* g_k1_stamps = nullptr;   // per translation unit; k_threshold.hip's copy (radius 7) is the one a3_debug_set_k1_stamps sets
#endif
#ifndef A3_T_LPX
#define A3_T_LPX 16
#endif
#ifndef A3_T_PF
#define A3_T_PF 3
#endif
#ifndef A3_T_RECOMPUTE
#define A3_T_RECOMPUTE 0
#endif
#ifndef A3_T_WAVES
#define A3_T_WAVES ((A3_T_LPX == 8 || A3_T_RECOMPUTE) ? 3 : 2)
#endif
// Pixels per lane and row.  16 (the default): 256 VGPRs (the ring of row sums alone is 120), two waves per SIMD.  8: every
// per-lane array halves, 144 VGPRs, three waves per SIMD -- built to see whether occupancy was what kept the kernel (stores off)
// 0.025 ms above the bare reads of tools/micro/readbench.hip.  It was not: 0.293 ms against 0.286 with stores off, 0.343 against
// 0.324 with them (tools/attic/tune_k1.sh); the difference to the microbenchmark is the 5 % of halo rows and the feeder lanes.
constexpr int T_LPX = A3_T_LPX;
static_assert(T_LPX == 8 || T_LPX == 16, "a lane owns 8 or 16 consecutive pixels");
constexpr int T_NG = T_LPX / 4;      // grey dwords per lane and row
constexpr int T_NP = T_LPX / 2;      // packed pairs per lane and row: pixel j with pixel j + T_NP
constexpr int T_OUT = 62 * T_LPX;    // output columns per wave (lanes 0 and 63 only feed their neighbours)
typedef typename std::conditional<T_LPX == 16, uint16_t, uint8_t>::type out_bits_t;   // a lane's result bits of one row


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// packed u16 pairs in one dword: v_pk_add_u16 / v_pk_sub_u16 / v_pk_mad_u16 / v_pk_sub_u16 clamp / v_pk_min_u16
// Sums of pairs: both halves at once with the plain 32-bit add / subtract.  No half ever carries into or borrows from the other --
// every partial result is a true window sum (0 .. 15 * 15 * 255 + 15 * 255 = 61 200 < 2^16), and the add comes before the subtract
// -- and v_add_u32 / v_sub_u32 issue at the SIMD's full rate (2 cycles per wave64) where v_pk_add_u16 / v_pk_sub_u16 take 4
// (tools/micro/valubench.hip).  -DA3_T_ADD32=0 restores the packed forms.
#ifndef A3_T_ADD32
#define A3_T_ADD32 1
#endif
// -DA3_T_WIDE_FLUSH=1 (round 5, measured): the parked result bits leave in 16-byte stores, eight rows per store
// instruction (lanes 8j .. 8j+7 write row j's 124 bytes), instead of one 2-byte store per lane and row: 18 store instructions per
// burst of 128 rows instead of 143.
#ifndef A3_T_WIDE_FLUSH
#define A3_T_WIDE_FLUSH 0
#endif
#ifndef A3_T_LUMA_GROUPS
#define A3_T_LUMA_GROUPS 1
#endif
#ifndef A3_T_COMPARE4
#define A3_T_COMPARE4 1
#endif
#ifndef A3_T_CMP_NOP
#define A3_T_CMP_NOP "s_nop 0\n\t"
#endif
#if A3_T_ADD32
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return a - b; }
#else
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u32(as_pk(a) + as_pk(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u32(as_pk(a) - as_pk(b)); }
#endif
// The compare stage is written with these three as inline assembly: given the vector expressions the optimiser rewrites
// min(sat(T - S), 1) into two scalar compares, two selects and a re-pack per pair (5x the instructions).
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {   // a * b + c per half
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_shift_in(uint32_t acc, uint32_t bit) {      // acc * 2 + bit per half
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(acc), "v"(bit));
    return r;
}
__device__ __forceinline__ uint32_t pk_nonzero_diff(uint32_t t, uint32_t s) {      // min(saturating t - s, 1) per half: 1 iff s < t
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(t), "v"(s));
    return d;
}

// (l * 13743896) >> 32 for l < 2^22: the upper half of floor(l / 10000) == (l * 13743896) >> 37, exact for every such l
// (checked exhaustively; 429497 >> 32 is NOT exact): one full-rate v_mul_hi_u32_u24, the remaining >> 5 is done by the
// instruction that puts the byte into place
template <bool BGR, int BYTE_OFF>
__device__ __forceinline__ uint32_t luma_hi(uint32_t px) {
    // pixel in bytes BYTE_OFF .. BYTE_OFF+2 of px (the other byte has weight 0).  Byte-wise dot products with the split
    // weights 2126 = 8*256+78, 7152 = 27*256+240, 722 = 2*256+210; the first byte is R (RGB/RGBA) or B (BGRA)
    constexpr uint32_t wlo = (BGR ? 0x004EF0D2u : 0x00D2F04Eu) << (8 * BYTE_OFF), whi = (BGR ? 0x00081B02u : 0x00021B08u) << (8 * BYTE_OFF);
    const uint32_t lo = __builtin_amdgcn_udot4(px, wlo, 0u, false);
    const uint32_t hi = __builtin_amdgcn_udot4(px, whi, 0u, false);
    const uint32_t l = lo + (hi << 8);   // <= 10000 * 255 < 2^22
    __builtin_assume(l < (1u << 22));
    return (uint32_t)(((uint64_t)l * 13743896ull) >> 32);   // both factors < 2^24: one v_mul_hi_u32_u24
}

template <int FMT> struct RawRow {
    static constexpr int BPP = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_L8 ? 1 : 4);
    static constexpr int NDW = T_LPX * BPP / 4;
    uint32_t d[NDW];
};
// the lane's NDW dwords of one row with the widest vector loads its alignment allows (8 bytes always; 16 when NDW % 4 == 0)
template <int FMT>
__device__ __forceinline__ void load_vec(const uint8_t* __restrict__ p, RawRow<FMT>& r) {
    constexpr int NDW = RawRow<FMT>::NDW;
    if constexpr (NDW % 4 == 0) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int i = 0; i < NDW / 4; i++) { const uint4 v = q[i]; r.d[4 * i] = v.x; r.d[4 * i + 1] = v.y; r.d[4 * i + 2] = v.z; r.d[4 * i + 3] = v.w; }
    } else {
        const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int i = 0; i < NDW / 2; i++) { const uint2 v = q[i]; r.d[2 * i] = v.x; r.d[2 * i + 1] = v.y; }
    }
}

// T_LPX consecutive pixels of row y starting at x0 (a multiple of T_LPX, may be negative or past the image): raw bytes,
// zero where the image is not.  Fully-inside lanes use vector loads.
template <int FMT>
__device__ __forceinline__ void load_raw(const uint8_t* __restrict__ frame, size_t row_stride, int x0, int y, int W, int H, bool aligned,
                                         RawRow<FMT>& r) {
    constexpr int NDW = RawRow<FMT>::NDW, BPP = RawRow<FMT>::BPP;
#pragma unroll
    for (int i = 0; i < NDW; i++) r.d[i] = 0u;
    if (y < 0 || y >= H || x0 + T_LPX <= 0 || x0 >= W) return;
    const uint8_t* row = frame + (size_t)y * row_stride;
    if (aligned && x0 >= 0 && x0 + T_LPX <= W) { load_vec<FMT>(row + (size_t)x0 * BPP, r); return; }
    for (int i = 0; i < T_LPX; i++) {
        const int x = x0 + i;
        if (x < 0 || x >= W) continue;
        for (int c = 0; c < BPP; c++) {
            const int byte = i * BPP + c;
            r.d[byte >> 2] |= (uint32_t)row[(size_t)x * BPP + c] << (8 * (byte & 3));
        }
    }
}

// raw row -> T_LPX grey bytes in T_NG dwords (byte i & 3 of g[i >> 2] = pixel i)
template <int FMT>
__device__ __forceinline__ void grey_row(const RawRow<FMT>& r, uint32_t g[T_NG]) {
    if constexpr (FMT == A3_FMT_L8) {
#pragma unroll
        for (int i = 0; i < T_NG; i++) g[i] = r.d[i];
    } else {
        uint32_t m[T_LPX];   // (l * 13743896) >> 32; grey = m >> 5
#if A3_T_LUMA_GROUPS
        // Four pixels at a time, stage by stage: the eight dot products, then the four sums, then the four multiplies.  A v_dot4's
        // result is not forwarded to the next two instructions (the compiler pads a closer consumer with s_nop, which costs an issue
        // slot like any instruction: 24 of them per row when every pixel's chain dot -> add -> multiply was emitted on its own);
        // with the stages of four pixels interleaved every consumer is at least three instructions behind its producer.
#pragma unroll
        for (int i0 = 0; i0 < T_LPX; i0 += 4) {
            uint32_t px[4], lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j;
                if constexpr (FMT == A3_FMT_RGBA8 || FMT == A3_FMT_BGRA8) px[j] = r.d[i];
                else {
                    const int byte = 3 * i, k = byte >> 2, off = byte & 3;   // compile-time after unrolling
                    px[j] = off <= 1 ? r.d[k] : __builtin_amdgcn_alignbit(r.d[k + 1], r.d[k], 8 * off);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                constexpr bool BGR = FMT == A3_FMT_BGRA8;
                // RGB8: pixel i starts at byte 3 i; pixels at byte offset 1 of their dword keep the data and move the weights
                const int off = (FMT == A3_FMT_RGB8) ? ((3 * (i0 + j)) & 3) : 0;
                const uint32_t sh = off == 1 ? 8u : 0u;
                const uint32_t wlo = (BGR ? 0x004EF0D2u : 0x00D2F04Eu) << sh, whi = (BGR ? 0x00081B02u : 0x00021B08u) << sh;
                lo[j] = __builtin_amdgcn_udot4(px[j], wlo, 0u, false);
                hi[j] = __builtin_amdgcn_udot4(px[j], whi, 0u, false);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) { lo[j] = lo[j] + (hi[j] << 8); __builtin_assume(lo[j] < (1u << 22)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) m[i0 + j] = (uint32_t)(((uint64_t)lo[j] * 13743896ull) >> 32);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int i = 0; i < T_LPX; i++) {
            if constexpr (FMT == A3_FMT_RGBA8 || FMT == A3_FMT_BGRA8) m[i] = luma_hi<FMT == A3_FMT_BGRA8, 0>(r.d[i]);
            else {
                // RGB8: pixel i starts at byte 3 i.  Pixels 0 and 3 of every group of four lie inside one dword (byte offsets
                // 0 and 1: the dot weights move instead of the data); pixels 1 and 2 straddle two dwords (one v_alignbit)
                const int byte = 3 * i, k = byte >> 2, off = byte & 3;   // compile-time after unrolling
                if (off == 0) m[i] = luma_hi<false, 0>(r.d[k]);
                else if (off == 1) m[i] = luma_hi<false, 1>(r.d[k]);
                else m[i] = luma_hi<false, 0>(__builtin_amdgcn_alignbit(r.d[k + 1], r.d[k], 8 * off));
            }
        }
#endif
        // g[q].byte[j] = m[4 q + j] >> 5, the shift writing its byte in place (SDWA dst_sel).  One asm block so that the order
        // is fixed: gfx950 needs one instruction between a dst_sel write of a VGPR and the next read of it (the partial
        // write is not forwarded); consecutive writes of one g[q] are T_NG instructions apart here, and the s_nop covers
        // whatever the compiler schedules right behind the block.
#define A3_SH(DST, SRC, B, U) "v_lshrrev_b32_sdwa " DST ", %" A3_STR(A3_SHAMT) ", " SRC " dst_sel:BYTE_" #B " dst_unused:UNUSED_" U " src0_sel:DWORD src1_sel:DWORD\n\t"
        if constexpr (T_LPX == 16) {
#define A3_SHAMT 4
#define A3_STR_(x) #x
#define A3_STR(x) A3_STR_(x)
            asm(A3_SH("%0", "%5", 0, "PAD") A3_SH("%1", "%9", 0, "PAD") A3_SH("%2", "%13", 0, "PAD") A3_SH("%3", "%17", 0, "PAD")
                A3_SH("%0", "%6", 1, "PRESERVE") A3_SH("%1", "%10", 1, "PRESERVE") A3_SH("%2", "%14", 1, "PRESERVE") A3_SH("%3", "%18", 1, "PRESERVE")
                A3_SH("%0", "%7", 2, "PRESERVE") A3_SH("%1", "%11", 2, "PRESERVE") A3_SH("%2", "%15", 2, "PRESERVE") A3_SH("%3", "%19", 2, "PRESERVE")
                A3_SH("%0", "%8", 3, "PRESERVE") A3_SH("%1", "%12", 3, "PRESERVE") A3_SH("%2", "%16", 3, "PRESERVE") A3_SH("%3", "%20", 3, "PRESERVE")
                "s_nop 0"
                : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[T_NG - 2]), "=&v"(g[T_NG - 1])
                : "v"(5u), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[T_LPX - 8]), "v"(m[T_LPX - 7]),
                  "v"(m[T_LPX - 6]), "v"(m[T_LPX - 5]), "v"(m[T_LPX - 4]), "v"(m[T_LPX - 3]), "v"(m[T_LPX - 2]), "v"(m[T_LPX - 1]));
#undef A3_SHAMT
        } else {
#define A3_SHAMT 2
            asm(A3_SH("%0", "%3", 0, "PAD") A3_SH("%1", "%7", 0, "PAD")
                A3_SH("%0", "%4", 1, "PRESERVE") A3_SH("%1", "%8", 1, "PRESERVE")
                A3_SH("%0", "%5", 2, "PRESERVE") A3_SH("%1", "%9", 2, "PRESERVE")
                A3_SH("%0", "%6", 3, "PRESERVE") A3_SH("%1", "%10", 3, "PRESERVE")
                "s_nop 0"
                : "=&v"(g[0]), "=&v"(g[1])
                : "v"(5u), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]));
#undef A3_SHAMT
        }
#undef A3_SH
    }
}

// full-rate 24-bit multiply (the compiler prefers the quarter-rate v_mul_lo_u32 when one operand is scalar)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // volatile: stays inside its (rarely taken) branch
    return r;
}

__device__ __forceinline__ uint32_t wave_from_left(uint32_t v) {   // lane i <- lane i-1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_from_right(uint32_t v) {  // lane i <- lane i+1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}

// bytes k & 3 of `lo` and of `hi` as a pair of u16 (lo's in the low half): the grey levels of pixels p and p + T_NP
__device__ __forceinline__ uint32_t byte_pair(int k, uint32_t hi, uint32_t lo) {
    return __builtin_amdgcn_perm(hi, lo, 0x0C000C00u | (uint32_t)(k & 3) | ((uint32_t)(4 + (k & 3)) << 16));
}

// (2R+1)-wide horizontal sums of one grey row: Hp[j] = (sum over pixels j-R .. j+R) | (sum over pixels j+T_NP-R .. j+T_NP+R) << 16,
// j = 0 .. T_NP-1, for the lane's T_LPX pixels; up to 8 grey bytes come from each neighbouring lane (wave shifts, no LDS).
// R <= 7: a window never reaches past the neighbouring lane's nearest 8 pixels, and sums of 2R+1 bytes fit 16 bits with room.
// sum of the bytes lo .. hi (inclusive) of the byte string D[] (byte k = byte k & 3 of D[k >> 2]) added to acc: whole dwords
// cost one v_sad_u8, partial ones an AND more; every index is a compile-time constant
template <int LO, int HI>
__device__ __forceinline__ uint32_t sum_bytes(const uint32_t* D, uint32_t acc) {
#pragma unroll
    for (int i = LO >> 2; i <= HI >> 2; i++) {
        const int b0 = i * 4 < LO ? LO - i * 4 : 0, b1 = i * 4 + 3 > HI ? HI - i * 4 : 3;   // bytes b0 .. b1 of dword i are inside
        const uint32_t mask = (b1 == 3 ? 0xFFFFFFFFu : ((1u << (8 * (b1 + 1))) - 1u)) & ~((1u << (8 * b0)) - 1u);
        acc = __builtin_amdgcn_sad_u8(mask == 0xFFFFFFFFu ? D[i] : (D[i] & mask), 0u, acc);
    }
    return acc;
}
template <int R = T_R>
__device__ __forceinline__ void row_sums(const uint32_t g[T_NG], uint32_t Hp[T_NP]) {
    static_assert(R >= 1 && R <= 7, "the packed 16-bit sums and the one-lane halo hold for radii 1..7");
    // D[] = the dwords of pixels -8 .. T_LPX+7; B[k] = byte k & 3 of D[k >> 2] = pixel k - 8
    uint32_t D[T_NG + 4];
    D[0] = wave_from_left(g[T_NG - 2]); D[1] = wave_from_left(g[T_NG - 1]);
#pragma unroll
    for (int i = 0; i < T_NG; i++) D[2 + i] = g[i];
    D[T_NG + 2] = wave_from_right(g[0]); D[T_NG + 3] = wave_from_right(g[1]);
    uint32_t ha, hb;
    if constexpr (R == 7) {
        // the two chains start at pixels 0 and T_NP: bytes p-7 .. p+7 = bytes 1..3 of D[(p + 1) >> 2] and the three dwords behind it
        if constexpr (T_LPX == 16) {
            const uint32_t mid = __builtin_amdgcn_sad_u8(D[3], 0u, 0u);                               // pixels 4..7, in both
            ha = __builtin_amdgcn_sad_u8(D[2], 0u, __builtin_amdgcn_sad_u8(D[1], 0u, __builtin_amdgcn_sad_u8(D[0] & 0xFFFFFF00u, 0u, mid)));
            hb = __builtin_amdgcn_sad_u8(D[5], 0u, __builtin_amdgcn_sad_u8(D[4], 0u, __builtin_amdgcn_sad_u8(D[2] & 0xFFFFFF00u, 0u, mid)));
        } else {
            const uint32_t mid = __builtin_amdgcn_sad_u8(D[3], 0u, __builtin_amdgcn_sad_u8(D[2], 0u, 0u));   // pixels 0..7, in both
            ha = __builtin_amdgcn_sad_u8(D[1], 0u, __builtin_amdgcn_sad_u8(D[0] & 0xFFFFFF00u, 0u, mid));
            hb = __builtin_amdgcn_sad_u8(D[4], 0u, __builtin_amdgcn_sad_u8(D[1] & 0xFFFFFF00u, 0u, mid));
        }
    } else {   // pixel p is byte p + 8: chain a sums bytes 8-R .. 8+R, chain b bytes 8+T_NP-R .. 8+T_NP+R
        ha = sum_bytes<8 - R, 8 + R>(D, 0u);
        hb = sum_bytes<8 + T_NP - R, 8 + T_NP + R>(D, 0u);
    }
    Hp[0] = ha | (hb << 16);
    // slide both chains one pixel: + (B[J+9+R], B[J+9+R+T_NP]) - (B[J+8-R], B[J+8-R+T_NP])
#pragma unroll
    for (int J = 0; J + 1 < T_NP; J++)
        Hp[J + 1] = pk_sub(pk_add(Hp[J], byte_pair(J + 9 + R, D[(J + 9 + R + T_NP) >> 2], D[(J + 9 + R) >> 2])),
                           byte_pair(J + 8 - R, D[(J + 8 - R + T_NP) >> 2], D[(J + 8 - R) >> 2]));
}

// One wave walks down a strip: lane l owns columns xs - T_LPX + T_LPX l .. + T_LPX - 1.  Per image row it converts its pixels
// to grey, forms their 15-wide horizontal sums (row_sums) and slides a 15-row vertical window over those row sums, all as
// packed u16 pairs (pixel j with pixel j + T_NP); the row 7 iterations old is then thresholded: sum < (L+1)*area.
// No barriers; T_PF rows of loads stay in flight per lane; LDS only parks the result bits between bursts of stores.
// grid: 8 * ceil(frames / 8) * strips_x * strips_y workgroups of one wave.
// R: the window's radius (threshold_window), 1..7; the window is NR = 2R+1 rows tall.  The rings have NRING slots, NR rounded up to
// a multiple of the load queue's depth, and the row loop is unrolled NRING times: the ring slot (row % NRING) and the queue slot
// (row % T_PF) of every row are then compile-time constants, and the row that leaves the window -- NR rows old -- sits in slot
// (row - NR) % NRING (for the default radius NR = NRING = 15: the very slot the new row overwrites).
template <int FMT, bool FAST, int T_PF = A3_T_PF, int R = T_R>
__global__ __launch_bounds__(64, T_PF == A3_T_PF ? A3_T_WAVES : 1) void k_grey_threshold7(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride,
                                                        int W, int H, int rows_per_wave, int strips_y, int n_pairs,
                                                        uint8_t* __restrict__ grey,
                                                        uint8_t* __restrict__ bits, int aligned_in, int aligned_out, int map_by_frame,
                                                        int flush_rows) {
    // flush_rows > 0: the result bits per lane and row are parked in LDS (flush_rows + 15 rows of 64 lanes) and leave in bursts
    // of flush_rows rows.  66 MB of small stores dribbling into a saturating read stream cost K1 ~0.07 ms (HBM bus turnarounds:
    // tools/micro/readbench.hip); the same bytes in a few large bursts per wave cost about half of that.
    constexpr int NR = 2 * R + 1, NRING = ((NR + T_PF - 1) / T_PF) * T_PF, UNROLL = NRING;
    extern __shared__ uint8_t s_out_raw[];
    out_bits_t* s_out = reinterpret_cast<out_bits_t*>(s_out_raw);
    int n_buf = 0, y_buf0 = 0;
    const int lane = threadIdx.x;
    // XCD-aware block -> strip mapping.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), each
    // with its own L2.  Vertically adjacent strips of one column share 14 rows of input, so all strips of a
    // (frame, column) pair are given to ONE XCD, in top-to-bottom order: the shared rows are then L2 hits instead of a
    // second trip over the fabric.  (Placement only affects speed; any mapping is correct.)
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int strips_x = (W + T_OUT - 1) / T_OUT;
    int pair, sy;
    if (map_by_frame) {   // every strip of a frame on one XCD: the column strips share the lines they both touch
        const int per_frame = strips_x * strips_y, idx = k % per_frame;
        pair = ((k / per_frame) * 8 + xcd) * strips_x + idx / strips_y;
        sy = idx % strips_y;
    } else { pair = (k / strips_y) * 8 + xcd; sy = k % strips_y; }
    if (pair >= n_pairs) return;
#ifdef A3_TUNING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();   // (tools/attic/k1_wave_times.py: when a launch's waves start and end)
#endif
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

    // clipped window widths of the lane's columns, 4 bits each (0 past the right edge: the comparison then fails)
    uint32_t axp[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < T_LPX; i++) {
        const int x = x0 + i;
        int a = 0;
        if (x >= 0 && x < W) a = min(x + R, W - 1) - max(x - R, 0) + 1;
        axp[i >> 3] |= (uint32_t)a << (4 * (i & 7));
    }

    uint32_t area[T_NP];       // clipped window areas of columns j | j + T_NP << 16 for the current row's window height
    uint32_t ay_cur = 0;
    uint32_t gring[NRING][T_NG];  // grey rows; row `it` lives in slot it % NRING (static: the row loop is unrolled); a row is read
                               // again R iterations later, so only R + 1 of the slots are live at any time
#if !A3_T_RECOMPUTE
    uint32_t hring[NRING][T_NP];  // the last NRING rows of horizontal sums (pairs); the window's oldest is NR rows back
#endif
    uint32_t S[T_NP];          // NR x NR window sums of the row R iterations old (pairs)
#pragma unroll
    for (int i = 0; i < T_NP; i++) S[i] = 0u;
#pragma unroll
    for (int q = 0; q < NRING; q++) {
#pragma unroll
        for (int i = 0; i < T_NG; i++) gring[q][i] = 0u;
#if !A3_T_RECOMPUTE
#pragma unroll
        for (int i = 0; i < T_NP; i++) hring[q][i] = 0u;
#endif
    }

    // Odd strips walk upwards.  Strip k (going down) and strip k+1 (going up) then both reach their common boundary --
    // the 14 rows each must also read from the other's territory -- at the END of their runs, and strips k+1 and k+2
    // both START at theirs: neighbours touch the shared rows at about the same time, so the second one finds them in
    // the XCD's L2 instead of fetching them again from HBM.  The box filter is symmetric, so direction only changes
    // the order rows enter and leave the window.
    const int dir = (sy & 1) ? -1 : 1;
    // the parked rows [0, nb) (row q is image row yb0 + dir * q) leave for the packed image
    auto flush_parked = [&](int nb, int yb0) {
#if A3_T_WIDE_FLUSH
        if constexpr (sizeof(out_bits_t) == 2) {
            // bytes of a row this strip owns: 2 per owner lane (lanes 1 .. 62 whose columns start inside the image)
            const int first_x = sx * T_OUT, n_own = min(62, (W - first_x + T_LPX - 1) / T_LPX);
            const int row_bytes = 2 * n_own, c = lane & 7, qo = lane >> 3;
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            uint8_t* const dst0 = bout + (first_x >> 3) + 16 * c;
            const uint8_t* const src0 = reinterpret_cast<const uint8_t*>(s_out) + 16 * c;
            for (int q0 = 0; q0 < nb; q0 += 8) {
                const int q = q0 + qo;
                if (q >= nb || 16 * c >= row_bytes) continue;
                uint8_t* dst = dst0 + (size_t)(yb0 + dir * q) * bpr;
                const uint8_t* src = src0 + (size_t)q * 128;
                if (16 * c + 16 <= row_bytes) *reinterpret_cast<u32x4_a4*>(dst) = *reinterpret_cast<const u32x4_a4*>(src);
                else for (int b = 0; 16 * c + b < row_bytes; b += 2) *reinterpret_cast<uint16_t*>(dst + b) = *reinterpret_cast<const uint16_t*>(src + b);
            }
            return;
        }
#endif
        if (owner)
            for (int q = 0; q < nb; q++)
                *reinterpret_cast<out_bits_t*>(bout + (size_t)(yb0 + dir * q) * bpr + (x0 >> 3)) = s_out[q * 64 + lane];
    };
    const int r_first = dir > 0 ? y_begin - R : y_end - 1 + R, n_rows = (y_end - y_begin) + 2 * R;
    RawRow<FMT> q[T_PF];
    // FAST: every load is unconditional (row and column clamped into the image) so that the loop body has no branch
    // around a load and the compiler can keep T_PF rows in flight with counted waits; what the clamped address
    // fetched for an outside lane/row is discarded by zeroing the grey below.
    const bool lane_in = x0 >= 0 && x0 + T_LPX <= W;
    const uint8_t* lane_ptr = frame + (size_t)(lane_in ? x0 : 0) * RawRow<FMT>::BPP;
    auto issue = [&](int r, RawRow<FMT>& dst) {
        if constexpr (FAST) load_vec<FMT>(lane_ptr + (size_t)(uint32_t)min(max(r, 0), H - 1) * row_stride, dst);
        else load_raw<FMT>(frame, row_stride, x0, r, W, H, aligned_in != 0, dst);
    };
#pragma unroll
    for (int k = 0; k < T_PF; k++) issue(r_first + dir * k, q[k]);

    // FAST: whole blocks of UNROLL rows and no exit test inside the unrolled body (rows past the strip are clamped loads whose
    // results are never stored), so the body is straight-line code apart from the store predicates
    const int n_iter = FAST ? ((n_rows + UNROLL - 1) / UNROLL) * UNROLL : n_rows;
    for (int base = 0; base < n_iter; base += UNROLL) {
#pragma unroll
        for (int ku = 0; ku < UNROLL; ku++) {
            const int k15 = ku;            // ring slot of this row (UNROLL == NRING)
            const int kold = (ku + NRING - NR) % NRING;   // slot of the row that leaves the window
            const int k = ku % T_PF;       // load-queue slot of this row
            const int it = base + ku;
            if (!FAST && it >= n_iter) break;
            const int r = r_first + dir * it;
            uint32_t g[T_NG];
            grey_row<FMT>(q[k], g);                                 // consumes the row loaded T_PF iterations ago ...
            if (FAST || it + T_PF < n_rows) issue(r + dir * T_PF, q[k]);  // ... and its registers take the next load at once
            if constexpr (FAST) {
                if (!(lane_in && r >= 0 && r < H)) {
#pragma unroll
                    for (int i = 0; i < T_NG; i++) g[i] = 0u;
                }
            }
            // Detection.grey of the rows this wave owns -- only when somebody reads the plane (debug taps); the decode stage
            // otherwise recomputes the few grey levels it samples from the frame itself
            if (write_grey && owner && r >= y_begin && r < y_end) {
                uint8_t* dst = gout + (size_t)r * W + x0;
                if (FAST || (aligned_out && x0 + T_LPX <= W)) {
                    if constexpr (T_LPX == 16) *reinterpret_cast<uint4*>(dst) = make_uint4(g[0], g[1], g[T_NG - 2], g[T_NG - 1]);
                    else *reinterpret_cast<uint2*>(dst) = make_uint2(g[0], g[1]);
                } else {
#pragma unroll
                    for (int i = 0; i < T_LPX; i++) if (x0 + i < W) dst[i] = (uint8_t)(g[i >> 2] >> (8 * (i & 3)));
                }
            }
            // horizontal sums of the new row, then the vertical window: + the new row's sums, - those of the row that
            // entered 15 iterations ago (they never underflow: the add comes first and the true sum is >= 0)
            uint32_t Hn[T_NP];
            row_sums<R>(g, Hn);
#if A3_T_RECOMPUTE
            {   // the sums of the row that leaves the window are formed again from its grey bytes (slot k15 still holds that row)
                uint32_t Ho[T_NP];
                row_sums<R>(gring[kold], Ho);
#pragma unroll
                for (int j = 0; j < T_NP; j++) S[j] = pk_sub(pk_add(S[j], Hn[j]), Ho[j]);
            }
#else
#pragma unroll
            for (int j = 0; j < T_NP; j++) { S[j] = pk_sub(pk_add(S[j], Hn[j]), hring[kold][j]); hring[k15][j] = Hn[j]; }
#endif
#pragma unroll
            for (int i = 0; i < T_NG; i++) gring[k15][i] = g[i];
            const uint32_t* centre = gring[(k15 + NRING - R) % NRING];   // the row R iterations old: the one being thresholded

            const int y = r - dir * R;   // the row whose window is now complete
            if (y < y_begin || y >= y_end) continue;   // wave-uniform
            const uint32_t ay = (uint32_t)(min(y + R, H - 1) - max(y - R, 0) + 1);
            if (ay != ay_cur) {   // wave-uniform; only the first and last R image rows differ from NR
                ay_cur = ay;
#pragma unroll
                for (int j = 0; j < T_NP; j++)
                    area[j] = mul24((axp[j >> 3] >> (4 * (j & 7))) & 15u, ay) | (mul24((axp[(j + T_NP) >> 3] >> (4 * ((j + T_NP) & 7))) & 15u, ay) << 16);
            }
            // white iff S < (L + 1) * area, two pixels per instruction: T = L * area + area, d = saturating T - S (non-zero
            // iff S < T), bit = min(d, 1), shifted in from the last pair down to the first: acc = acc * 2 + bit
            uint32_t acc = 0u;
#if defined(A3_TUNING) && defined(A3_K1_PROBE_SKIP_COMPARE)
            // timing probe (wrong results): the compare stage's 40 instructions per row replaced by 8 that keep S and the centre row alive
#pragma unroll
            for (int j = 0; j < T_NP; j++) acc ^= S[j] + centre[j & (T_NG - 1)];
#else
#if A3_T_COMPARE4
            // four pairs per asm block, stage by stage (4 x T = L * area + area, 4 x saturating T - S, 4 x min(.., 1), then the four
            // bits of each half combined by a tree: b1 * 2 + b0, b3 * 2 + b2, then * 4 +): no instruction directly behind the one
            // whose result it reads, and two asm blocks per row instead of twenty-four (the compiler pads an asm statement it cannot
            // see into with s_nop, which costs an issue slot each)
            {
                uint32_t part[T_NP / 4];
#pragma unroll
                for (int q = 0; q < T_NP / 4; q++) {
                    uint32_t Lp[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) Lp[u] = byte_pair(4 * q + u, centre[(4 * q + u + T_NP) >> 2], centre[(4 * q + u) >> 2]);
                    uint32_t t0, t1, t2, t3;
                    asm("v_pk_mad_u16 %0, %5, %9, %9\n\t"
                        "v_pk_mad_u16 %1, %6, %10, %10\n\t"
                        "v_pk_mad_u16 %2, %7, %11, %11\n\t"
                        "v_pk_mad_u16 %3, %8, %12, %12\n\t"
                        "v_pk_sub_u16 %0, %0, %13 clamp\n\t"
                        "v_pk_sub_u16 %1, %1, %14 clamp\n\t"
                        "v_pk_sub_u16 %2, %2, %15 clamp\n\t"
                        "v_pk_sub_u16 %3, %3, %16 clamp\n\t"
                        "v_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\t"
                        "v_pk_min_u16 %1, %1, 1 op_sel_hi:[1,0]\n\t"
                        "v_pk_min_u16 %2, %2, 1 op_sel_hi:[1,0]\n\t"
                        "v_pk_min_u16 %3, %3, 1 op_sel_hi:[1,0]\n\t"
                        "v_pk_mad_u16 %1, %1, 2, %0 op_sel_hi:[1,0,1]\n\t"
                        "v_pk_mad_u16 %3, %3, 2, %2 op_sel_hi:[1,0,1]\n\t"
                        A3_T_CMP_NOP
                        "v_pk_mad_u16 %4, %3, 4, %1 op_sel_hi:[1,0,1]"
                        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(part[q])
                        : "v"(Lp[0]), "v"(Lp[1]), "v"(Lp[2]), "v"(Lp[3]), "v"(area[4 * q]), "v"(area[4 * q + 1]), "v"(area[4 * q + 2]), "v"(area[4 * q + 3]),
                          "v"(S[4 * q]), "v"(S[4 * q + 1]), "v"(S[4 * q + 2]), "v"(S[4 * q + 3]));
                }
                acc = part[0];   // bits of pairs 0..3; every further group of four sits 4 bits higher (no carries: disjoint bits)
#pragma unroll
                for (int q = 1; q < T_NP / 4; q++) acc |= part[q] << (4 * q);
            }
#else
#pragma unroll
            for (int j = T_NP - 1; j >= 0; j--) {
                const uint32_t Lp = byte_pair(j, centre[(j + T_NP) >> 2], centre[j >> 2]);
                const uint32_t T = pk_mad(Lp, area[j], area[j]);
                acc = pk_shift_in(acc, pk_nonzero_diff(T, S[j]));
            }
#endif
#endif
            // the bits of pixels 0 .. T_NP-1 sit in the low half of acc, those of pixels T_NP .. in the high half
            const uint32_t outb = T_LPX == 16 ? __builtin_amdgcn_perm(0u, acc, 0x0C0C0200u) : ((acc | (acc >> 12)) & 0xFFu);
            if (flush_rows <= 0) {
#ifdef A3_TUNING
                if (owner && (flush_rows == 0 || outb == 0x12345u))   // (-1: timing probe, no stores)
#else
                if (owner)
#endif
                    *reinterpret_cast<out_bits_t*>(bout + (size_t)y * bpr + (x0 >> 3)) = (out_bits_t)outb;
            } else {
                if (n_buf == 0) y_buf0 = y;
#if A3_T_WIDE_FLUSH
                s_out[n_buf * 64 + (sizeof(out_bits_t) == 2 ? ((lane + 63) & 63) : lane)] = (out_bits_t)outb;   // owners 1..62 at slots 0..61: a row's bytes as they lie in memory
#else
                s_out[n_buf * 64 + lane] = (out_bits_t)outb;
#endif
                n_buf++;
            }
        }
        // (checked once per block of UNROLL rows, outside the unrolled body: the buffer holds flush_rows + UNROLL rows)
        if (flush_rows > 0 && n_buf >= flush_rows) {   // wave-uniform
            flush_parked(n_buf, y_buf0);
            n_buf = 0;
        }
    }
    if (flush_rows > 0) flush_parked(n_buf, y_buf0);
#ifdef A3_TUNING
    if (g_k1_stamps && lane == 0) {   // 100 MHz timestamps of this wave's life + where it ran (HW_ID, XCC_ID)
        unsigned long long* o = g_k1_stamps + (size_t)blockIdx.x * 4;
        o[0] = t_begin; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
        o[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
#endif
}

// launch geometry switches (k_threshold.hip; a3_internal.h: a3_debug_set_k1_waves, a3_debug_set_partition)
extern int g_k1_waves, g_k1_cus;

// the register-resident kernel for one radius R in 1..7 (the default, 7, is the one every figure of DESIGN.md is about)
template <int R>
hipError_t launch_k1(hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n,
                            uint8_t* grey, uint64_t* bits) {
    constexpr int NR = 2 * R + 1, UNROLL = ((NR + A3_T_PF - 1) / A3_T_PF) * A3_T_PF;   // (as in the kernel)
    const int aligned_in = ((uintptr_t)pixels % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
    const int aligned_out = (W % 16 == 0) && ((uintptr_t)grey % 16 == 0);
    uint8_t* bin = reinterpret_cast<uint8_t*>(bits);
    if (W % 64 != 0) {  // packed rows end in padding bits that no tile writes
        hipError_t e = hipMemsetAsync(bits, 0, (size_t)words_per_row((uint32_t)W) * 8 * H * n, st);
        if (e != hipSuccess) return e;
    }
    // Rows per wave.  Every wave also reads and converts 2R rows outside its strip, so strips should be tall; but the
    // chip holds 256 CUs x 4 SIMDs x A3_T_WAVES waves at once and a launch runs in whole rounds of that many, so the
    // number of strips should fill the last round.  Model: time ~ rounds x (rows per strip + 2R); take the best
    // strip count (at least 16 rows per strip).  256 frames of 1920x1080, R = 7: 2 column strips x 4 strips of 270 rows = 2048 waves
    // = exactly one round of two waves per SIMD.
    const int strips_x = (W + T_OUT - 1) / T_OUT;
    const long long slots = (long long)g_k1_cus * 4 * g_k1_waves, cols = (long long)strips_x * n;
    // (Strips of ONE row for a single small frame -- 480 waves of one 15-row block instead of 30 waves of two -- were tried in round 5:
    // 23.0 us against 16.8 for one 640x480 frame; a wave's fixed costs outweigh the block saved.  The model stays at >= 16 rows.)
    int best_sy = 1; double best_cost = 1e300;
    for (int sy = 1; sy <= std::max(1, H / 16); sy++) {
        const int rows = (H + sy - 1) / sy;
        const long long waves = cols * ((H + rows - 1) / rows);
        const double cost = (double)((waves + slots - 1) / slots) * (rows + 2 * R);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_sy = sy; }
    }
    int rows_per_wave = (H + best_sy - 1) / best_sy;
    if (const int rv = tuning_knob("A3_ROWS_PER_WAVE", 0); rv > 0) rows_per_wave = rv;   // (-DA3_TUNING builds only)
    const int strips_y = (H + rows_per_wave - 1) / rows_per_wave;
    const int n_pairs = (int)n * strips_x;
    // every strip of a frame on one XCD (1) or every (frame, column strip) pair on its own XCD (0).  By frame is ~3 % faster:
    // the two column strips of a frame overlap by 32 columns and write the same lines of the packed image
    const int map_by_frame = tuning_knob("A3_K1_MAP", 1);
    // rows of results a wave parks in LDS before it writes them out (0: store row by row): 128 (+ UNROLL) rows x 64 lanes x 1 or 2
    // bytes = 9 or 18 KB per wave; twelve resp. eight waves per CU fit the 160 KB
    // (-1 = "no stores at all" is a timing probe that leaves the binary image stale: it exists in -DA3_TUNING builds only)
    // (every resident wave's parking area must fit the CU's 160 KB: 4 x A3_T_WAVES waves)
    constexpr int flush_cap = (160 * 1024 / (4 * A3_T_WAVES)) / (64 * (int)sizeof(out_bits_t)) - UNROLL;
    const int fv = tuning_knob("A3_K1_FLUSH", flush_cap < 128 ? flush_cap : 128);
#ifdef A3_TUNING
    const int flush_rows = fv < 0 ? -1 : std::min(fv, rows_per_wave);
#else
    const int flush_rows = std::min(fv < 0 ? 128 : fv, rows_per_wave);
#endif
    const size_t lds_bytes = flush_rows > 0 ? (size_t)(flush_rows + UNROLL) * 64 * sizeof(out_bits_t) : 0;
    dim3 grid(map_by_frame ? 8 * (((int)n + 7) / 8) * strips_x * strips_y : 8 * ((n_pairs + 7) / 8) * strips_y), block(64);
    const bool fast = aligned_in && aligned_out;   // W % 16 == 0: a lane's 16 pixels are all inside or all outside
#define A3_LAUNCH_K1(F, B, PF) hipLaunchKernelGGL((k_grey_threshold7<F, B, PF, R>), grid, block, lds_bytes, st, pixels, row_stride, frame_stride, W, H, \
                                              rows_per_wave, strips_y, n_pairs, grey, bin, aligned_in, aligned_out, map_by_frame, flush_rows)
    if (fmt == A3_FMT_RGB8) {
        if constexpr (R == T_R && A3_T_WAVES != 1) { if (fast && g_k1_waves == 1) { A3_LAUNCH_K1(A3_FMT_RGB8, true, 5); return hipGetLastError(); } }
        if (fast) A3_LAUNCH_K1(A3_FMT_RGB8, true, A3_T_PF); else A3_LAUNCH_K1(A3_FMT_RGB8, false, A3_T_PF);
    }
    else if (fmt == A3_FMT_RGBA8) { if (fast) A3_LAUNCH_K1(A3_FMT_RGBA8, true, A3_T_PF); else A3_LAUNCH_K1(A3_FMT_RGBA8, false, A3_T_PF); }
    else if (fmt == A3_FMT_BGRA8) { if (fast) A3_LAUNCH_K1(A3_FMT_BGRA8, true, A3_T_PF); else A3_LAUNCH_K1(A3_FMT_BGRA8, false, A3_T_PF); }
    else { if (fast) A3_LAUNCH_K1(A3_FMT_L8, true, A3_T_PF); else A3_LAUNCH_K1(A3_FMT_L8, false, A3_T_PF); }
#undef A3_LAUNCH_K1
    return hipGetLastError();
}


}  // namespace a3
