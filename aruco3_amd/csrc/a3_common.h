// a3_common.h -- shared device/host declarations for libaruco3_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#ifdef A3_TUNING
#include <cstdlib>
#endif

#include "../../include/aruco3_hip.h"

namespace a3 {

constexpr int WAVE = 64;

// Tuning knobs exist only in -DA3_TUNING builds (`make tuning` -> build/tuning/libaruco3_hip.so, used by tools/ sweeps):
// the product library never reads the environment, and no knob that changes results can be compiled into it.
#ifdef A3_TUNING
inline int tuning_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int tuning_knob(const char*, int dflt) { return dflt; }
#endif

// ---- contour graph ("darts") ------------------------------------------------------------
// A dart is (pixel, direction of a foreground 8-neighbour).  Directions are numbered
// clockwise on screen starting at west, the ring order of the border follower the
// reference calls (imageproc find_contours, src/aruco.rs:64): W NW N NE E SE S SW.
__device__ __constant__ const int8_t kDX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
__device__ __constant__ const int8_t kDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

constexpr uint32_t kNoKey = 0xFFFFFFFFu;  // "this dart is not a start event"
constexpr uint32_t kNone = 0xFFFFFFFFu;

// dart info byte: bits 0-2 = incoming direction, bit 3 = W-event dart, bit 4 = E-event dart,
// bit 5 = successor missing (chain end; only face cycles of the 8-neighbour graph do this)
constexpr uint8_t kInfoW = 1u << 3;
constexpr uint8_t kInfoE = 1u << 4;
constexpr uint8_t kInfoBroken = 1u << 5;

// per-dart record, one u64: bits 0-15 x, 16-31 y, 32-39 foreground-neighbour mask F of the pixel, 40-47 info byte,
// 48-63 frame index inside the chunk (chunks hold at most 65536 frames)
constexpr uint32_t kMaxChunkFrames = 65536u;
__host__ __device__ inline uint64_t dart_rec(uint32_t xy, uint32_t F, uint32_t info, uint32_t frame) {
    return (uint64_t)xy | ((uint64_t)(F & 0xFFu) << 32) | ((uint64_t)(info & 0xFFu) << 40) | ((uint64_t)(frame & 0xFFFFu) << 48);
}
__host__ __device__ inline uint32_t rec_frame(uint64_t r) { return (uint32_t)(r >> 48); }
__host__ __device__ inline uint32_t rec_xy(uint64_t r) { return (uint32_t)r; }
__host__ __device__ inline uint32_t rec_F(uint64_t r) { return (uint32_t)(r >> 32) & 0xFFu; }
__host__ __device__ inline uint32_t rec_info(uint64_t r) { return (uint32_t)(r >> 40) & 0xFFu; }

// state carried by the pointer-doubling rounds (16 bytes per dart = one dwordx4, ping-pong)
struct __attribute__((aligned(16))) JumpState {
    uint64_t key;   // (start-event key << 32) | dart index : minimum over the window
    uint32_t ptr;   // succ^(2^round)
    uint32_t off;   // hops from this dart to the first dart holding `key`
};
static_assert(sizeof(JumpState) == 16, "JumpState layout");

// The thresholded image is kept bit-packed: row y of a frame is `words_per_row(W)` little-endian u64 words,
// bit i of word j is pixel x = 64 j + i (1 = white / foreground); bits past the image width are 0.
__host__ __device__ inline uint32_t words_per_row(uint32_t W) { return (W + 63u) / 64u; }

// Directions k whose dart (p,k) can lie on a border the reference follows, from the foreground-neighbour mask F
// of p: neighbour k foreground, neighbour k-1 background (the counter-clockwise sweep is not empty), and for a
// 4-neighbour direction (k even) neighbour k-2 background as well; every other dart only belongs to a
// triangular face cycle of the 8-neighbour graph (tests/dart_model.py, rule "pdart").
__host__ __device__ inline uint32_t pdart_mask(uint32_t F) {
    const uint32_t rot1 = ((F << 1) | (F >> 7)) & 0xFFu, rot2 = ((F << 2) | (F >> 6)) & 0xFFu;
    return F & ~rot1 & (0xAAu | ~rot2) & 0xFFu;
}

// Where the decode stage reads grey levels from: the caller's frames (grey is recomputed per tap, so K1 need not write a
// grey plane at all) or, with debug taps / other threshold windows, the grey plane K1 wrote (fmt = kFmtGreyPlane).
constexpr int kFmtGreyPlane = 4;
struct PixelSrc {
    const uint8_t* base;
    unsigned long long row_stride, frame_stride;
    int fmt;   // A3_FMT_* or kFmtGreyPlane
};

// image::DynamicImage::into_luma8 for one pixel: (2126 R + 7152 G + 722 B) / 10000, truncating
__host__ __device__ inline uint32_t luma_of(uint32_t r, uint32_t g, uint32_t b) { return (2126u * r + 7152u * g + 722u * b) / 10000u; }

// one border that survived the size pruning and gets its points written out
struct ContourRec {
    uint32_t frame;
    uint32_t start_key;   // 2*raster(start pixel) (+1 when started as a hole border): the reference's contour order
    uint32_t point_base;  // into the point pool
    uint32_t n;           // number of points
};

// a quad candidate (contours_to_candidates + enforce_clockwise, src/aruco.rs:124-185)
struct CandRec {
    uint32_t start_key;
    uint16_t xy[8];
};

struct DeviceCounters {
    unsigned long long darts;       // per chunk: next free dart
    unsigned long long points;      // next free point
    unsigned int contours;          // next free ContourRec
    unsigned int traced;            // cycles with a start event (stat)
    unsigned int err_flags;         // bit 0: event dart on a broken chain, bit 1: point pool overflow,
                                    // bit 2: contour table overflow, bit 3: candidate table overflow
    unsigned int resolve_needed;    // set by k_resolve_fast: some border's first start event does not fire -> run the fixpoint passes
    unsigned int jump_changed[32];     // per doubling round: darts whose key changed
    unsigned int resolve_changed[16];  // per start-resolution pass: cycles whose start moved
    unsigned int entry_overflow;    // k_entry_frame: a frame has more entries than fit in LDS -> re-run with the global doubling rounds
    unsigned int pad[1];
};

constexpr unsigned kErrBrokenEvent = 1u, kErrPointPool = 2u, kErrContourTable = 4u, kErrCandTable = 8u, kErrResolve = 16u, kErrMarkerCap = 32u;

}  // namespace a3
