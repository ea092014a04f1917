// k_contours.hip -- K2..K4: bit-packed binary image -> quad candidates, fully data-parallel.
//
// Replaces `imageproc::contours::find_contours` + `contours_to_candidates` +
// `enforce_clockwise_corners` (src/aruco.rs:64-68, 124-185).
//
// The reference follows borders one at a time in raster order with a label image
// (Suzuki-Abe).  Here the same borders come out of a graph formulation with no
// sequential scan (executable model, checked against the oracle: tests/dart_model.py):
//
//   dart       (pixel p, direction k of a foreground 8-neighbour) with a non-empty
//              counter-clockwise sweep (pdart_mask in a3_common.h): the only darts a border
//              can pass through, about 1.2 per border pixel.
//   succ       (p,k) -> (p + dir(k'), opposite(k')), k' = next foreground neighbour
//              counter-clockwise after k.  Darts fall into cycles (plus open chains that
//              belong to no border), and every border the reference traces is one cycle,
//              rotated to its start dart.
//   doubling   pointer jumping gives every dart its cycle's leader (the dart holding the smallest
//              start-event key) and its hop distance to the leader (=> cycle length and rank
//              along the border): 11 rounds inside 2048-dart tiles in LDS, then only the darts
//              whose predecessor lies in another tile ("entries") are doubled globally.
//   events     a W-event at pixel q (x>0, west neighbour background) can start a border as
//              "outer", an E-event (x+1<W, east neighbour background) as "hole"; which event
//              starts a cycle is the fixpoint of the rule in k_resolve_eval (the reference's
//              label tests `== 1` / `> 0`, restated on cycles); k_resolve_fast confirms the
//              usual answer (every border starts at its smallest event) from the leaders alone.
//              Start keys give the reference's contour order.
//   pruning    only parity-safe (see k_cycle_select): a border too short to hold one
//              candidate edge, or so long that epsilon = 0.05*len exceeds the image
//              diagonal, can never yield a 4-point candidate (src/aruco.rs:133-158).
//   DP         one wave per surviving border: Douglas-Peucker with a wave arg-max, split
//              count capped at 3 (exactly 4 points are needed), hull + winding + edge test.
//
// Border pixels are found 64 at a time with word-wide bit operations on the packed image; dart
// ranges come from prefix sums over per-tile counts (k_dart_count -> k_tile_scan -> k_plan), so
// numbering is deterministic and no hot path bumps one address from many workgroups: a returning
// atomic on one address costs ~11 ns and they serialise (12 k tiles = 140 us).  Counters that
// remain are sharded 16 ways (entry and leader lists) or bumped once per workgroup.
#include <algorithm>
#include <cstdlib>

#include "a3_common.h"

// Darts per lane and trip in the per-dart sweeps (build knobs, swept on BASELINE config 2).  k_jump_finalize: 2 / 4 / 6 / 8 darts
// -> 53 / 48 / 44.5 / 45 us with its three dependent loads of rounds 2-5; with two (round 6: the entry's slot comes with the state)
// 4 / 6 / 8 / 10 -> 44.7 / 41.9 / 39.0 / 165 us (10 spills).  k_scatter_points: 1 / 2 / 3 / 4 -> 34.5 / 37 / 37.5 / 41 us
// (on 8192 workgroups): its loads of one dart already come two and three at a time, and more threads beat more darts per thread.
#ifndef A3_FIN_B
#define A3_FIN_B 8
#endif
#ifndef A3_SCAT_B
#define A3_SCAT_B 1
#endif

namespace a3 {

// ---------------------------------------------------------------------------------------
// 64 pixels per lane: neighbour occupancy words in ring order W NW N NE E SE S SW
// ---------------------------------------------------------------------------------------
struct Nb8 { uint64_t c; uint64_t n[8]; };

__device__ __forceinline__ uint64_t ldw(const uint64_t* __restrict__ img, int wpr, int H, int j, int y) {
    return (y >= 0 && y < H && j >= 0 && j < wpr) ? img[(size_t)y * wpr + j] : 0ull;
}

// per direction: the pixels of this word that own a dart in that direction (pdart_mask, 64 pixels at once)
__device__ __forceinline__ void pdart_words(const Nb8& nb, uint64_t p[8]) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t m = nb.c & nb.n[k] & ~nb.n[(k + 7) & 7];
        if ((k & 1) == 0) m &= ~nb.n[(k + 6) & 7];
        p[k] = m;
    }
}

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // inclusive scan over the wave with DPP row shifts and row broadcasts (VALU only; a __shfl_up ladder is six ds_bpermute
    // round trips through the LDS crossbar)
    uint32_t inc = v;
#define A3_DPP_ADD(CTRL, ROWMASK) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, CTRL, ROWMASK, 0xF, false);
    A3_DPP_ADD(0x111, 0xF)   // row_shr:1, 2, 4, 8: inclusive sums inside each row of 16 lanes
    A3_DPP_ADD(0x112, 0xF)
    A3_DPP_ADD(0x114, 0xF)
    A3_DPP_ADD(0x118, 0xF)
    A3_DPP_ADD(0x142, 0xA)   // row_bcast:15: rows 1 and 3 add the totals of rows 0 and 2
    A3_DPP_ADD(0x143, 0xC)   // row_bcast:31: rows 2 and 3 add the total of rows 0-1
#undef A3_DPP_ADD
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// One packed word (64 pixels) per lane; a workgroup covers a compact tile of 4 words x 64 rows (256 x 64 pixels) so
// that the darts it creates -- which get consecutive indices -- are neighbours in the image: successor pointers then
// mostly stay inside a small index range, which k_local_contract exploits.
constexpr int kTileWords = 4, kTileRows = 64;
__host__ __device__ inline uint32_t dart_tiles_x(uint32_t W) { return (words_per_row(W) + kTileWords - 1) / kTileWords; }
__host__ __device__ inline uint32_t dart_tiles(uint32_t W, uint32_t H) { return dart_tiles_x(W) * ((H + kTileRows - 1) / kTileRows); }

// The tile's words plus a one-word / one-row apron are staged in LDS with row-contiguous loads (6 words per row,
// 66 rows) instead of nine strided 8-byte loads per lane.
__device__ __forceinline__ void tile_stage(const uint64_t* __restrict__ img, int W, int H, uint32_t tile, uint64_t (*s_t)[kTileWords + 2]) {
    const int wpr = (int)words_per_row((uint32_t)W);
    const int tx = tile % dart_tiles_x((uint32_t)W), ty = tile / dart_tiles_x((uint32_t)W);
    const int j0 = tx * kTileWords - 1, y0 = ty * kTileRows - 1;
    // both words of a lane are requested before either is stored (unconditional loads from clamped addresses, masked
    // afterwards: a load behind a bounds test is issued and waited for on its own)
    constexpr int kStage = (kTileRows + 2) * (kTileWords + 2);
    static_assert(kStage <= 512, "two words per lane");
    uint64_t v[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = min((int)threadIdx.x + 256 * k, kStage - 1);
        const int r = i / (kTileWords + 2), c = i - r * (kTileWords + 2);
        const int jj = j0 + c, yy = y0 + r;
        const uint64_t w = img[(size_t)min(max(yy, 0), H - 1) * wpr + min(max(jj, 0), wpr - 1)];
        v[k] = w & (0ull - (uint64_t)(yy >= 0 && yy < H && jj >= 0 && jj < wpr));
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = (int)threadIdx.x + 256 * k;
        if (i < kStage) (&s_t[0][0])[i] = v[k];
    }
}
// the word (rl, jl) of the staged tile and its eight neighbour words' bits, aligned to it
__device__ __forceinline__ Nb8 tile_nb8(const uint64_t (*s_t)[kTileWords + 2], int rl, int jl) {
    Nb8 r;
    const uint64_t al = s_t[rl][jl], a = s_t[rl][jl + 1], ar = s_t[rl][jl + 2];
    const uint64_t cl = s_t[rl + 1][jl], c = s_t[rl + 1][jl + 1], cr = s_t[rl + 1][jl + 2];
    const uint64_t bl = s_t[rl + 2][jl], b = s_t[rl + 2][jl + 1], br = s_t[rl + 2][jl + 2];
    r.c = c;
    r.n[0] = (c << 1) | (cl >> 63);
    r.n[1] = (a << 1) | (al >> 63);
    r.n[2] = a;
    r.n[3] = (a >> 1) | (ar << 63);
    r.n[4] = (c >> 1) | (cr << 63);
    r.n[5] = (b >> 1) | (br << 63);
    r.n[6] = b;
    r.n[7] = (b << 1) | (bl >> 63);
    return r;
}

// Dart counts per 256 x 64-pixel tile, without LDS staging: one wave walks down one tile row (64 image rows plus the row
// above and below), lane l holding packed word l - 1 of the row (lanes 0 and 61 only feed their neighbours; images wider
// than 60 words take several waves side by side).  Of the horizontal neighbours only one bit is needed (bit 63 of the word
// to the left, bit 0 of the word to the right): two wave shifts per row; the rows above and below stay in registers.
// grid: (chunks_x * tiles_y, frames) workgroups of one wave.  (The LDS-tiled version took 44 us for 256 frames of
// 1920x1080: it staged every tile with its apron before looking at it; most words of a frame are all white and are now
// dismissed after three compares.)
constexpr int kCountLanes = 60;   // words per wave and row: 15 whole tiles (lanes 1..60; lanes 0 and 61 feed their neighbours)
// Images of at most 30 words per row (1920 pixels) put TWO tile rows into one wave: lanes 0..31 walk one, lanes 32..63 the
// next (words 0..29 in lanes 1..30 of each half; lanes 0 and 31 hold zeros, which is what lies outside the image, so the wave
// shifts across the middle deliver the right bits).
// A tile row is walked by kCountHalves waves, 64 / kCountHalves image rows each (blockIdx.z; each writes its own count array,
// k_tile_scan adds them): with one wave per tile row pair the launch was 2304 waves on 1024 SIMDs, i.e. three on some SIMDs
// and two on the others, and the kernel is bound by its bit arithmetic (VALU busy 70 %), so it took as long as three waves on
// one SIMD.  1 / 2 / 4 parts: 36 / 31 / 27.6 us (each part re-reads two rows of its neighbours).
#ifndef A3_COUNT_PARTS
#define A3_COUNT_PARTS 4
#endif
constexpr int kCountHalves = A3_COUNT_PARTS;
template <int G>
__global__ __launch_bounds__(64) void k_dart_count(const uint64_t* __restrict__ bits, int W, int H, uint32_t first_frame,
                                                   unsigned long long* __restrict__ frame_darts, uint32_t* __restrict__ tile_darts,
                                                   uint32_t* __restrict__ tile_darts_h1, unsigned long long* __restrict__ tile_mask) {
    // tile_mask[(frame * tiles + tile) * kCountHalves + part]: bit jl * 16 + r = word jl of row r of this part owns darts (column by
    // column: k_dart_assign numbers a tile's words, and with them its darts, in that order).  A part
    // is 16 rows x 4 words = the 64 words that one wave of k_dart_assign's phase 1 would look at: the mask lets that kernel
    // hand only the words that own darts to its lanes.
    static_assert(kCountLanes % kTileWords == 0, "a wave counts whole tiles");
    static_assert((kTileRows / kCountHalves) * kTileWords == 64, "one 64-bit mask per tile and part");
    constexpr int kGroups = 64 / G, kOwners = G == 64 ? kCountLanes : 30, kTilesPerGroup = (kOwners + kTileWords - 1) / kTileWords;
    __shared__ uint32_t s_tile[kGroups][kTilesPerGroup];
    __shared__ unsigned long long s_mask[kGroups][kTilesPerGroup];
    // Words that can own darts (not empty, not in the middle of a white area: ~5 % of a clean frame's words) are QUEUED -- the
    // word, the words above and below, six neighbour bits, its lane and row -- and worked on 64 at a time: the bit arithmetic
    // that finds their darts (75 instructions) then runs with every lane busy, once per 64 such words, instead of once per image
    // row for the two or three lanes that hold one (A3_COUNT_QUEUE=0: in place).
    constexpr uint32_t kQ = 128;                   // ring of two flushes: a row adds at most 64 entries to fewer than 64 waiting
    __shared__ uint64_t q_c[kQ], q_up[kQ], q_dn[kQ];
    __shared__ uint32_t q_meta[kQ];                // bits 0-5: l/r bits of the row, the row above, the row below; 8-13: lane; 16-19: row in the part
    const int wpr = (int)words_per_row((uint32_t)W);
    const uint32_t f = blockIdx.y;
    const uint32_t tiles_x = dart_tiles_x((uint32_t)W), tiles_y = ((uint32_t)H + kTileRows - 1) / kTileRows;
    const int chunks_x = G == 64 ? (wpr + kCountLanes - 1) / kCountLanes : 1;
    const int lane = threadIdx.x, gl = lane & (G - 1), grp = lane / G;
    const int cx = blockIdx.x % chunks_x, ty = (blockIdx.x / chunks_x) * kGroups + grp;
    const int j = cx * kOwners + gl - 1;                       // this lane's word column (may lie outside the image: zeros)
    const bool owner = gl >= 1 && gl <= kOwners && j < wpr && ty < (int)tiles_y;
    const uint64_t* img = bits + (size_t)(first_frame + f) * wpr * H;
    if (lane < kGroups * kTilesPerGroup) { (&s_tile[0][0])[lane] = 0; (&s_mask[0][0])[lane] = 0ull; }
    constexpr int kRows = kTileRows / kCountHalves;
    const int y0 = ty * kTileRows + (int)blockIdx.z * kRows;
    // (unconditional load from a clamped address, then a select: a load behind a branch cannot be batched with its neighbours)
    const int jc = min(max(j, 0), wpr - 1);
    auto word = [&](int y) -> uint64_t {
        const uint64_t v = img[(size_t)min(max(y, 0), H - 1) * wpr + jc];
        return v & (0ull - (uint64_t)(y >= 0 && y < H && j >= 0 && j < wpr));   // (an AND, not a select: a select lets the compiler sink the load into a branch again)
    };
    auto edge_bits = [](uint64_t c, uint32_t* lr) {   // bit 0: bit 63 of the word to the left; bit 1: bit 0 of the word to the right
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(c >> 32), 0x138 /* wave_shr:1 */, 0xF, 0xF, true) >> 31;
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)c, 0x130 /* wave_shl:1 */, 0xF, 0xF, true) & 1u;
        *lr = l | (r << 1);
    };
    struct Row3 { uint64_t c; uint32_t lr; };
    // rows are fetched kAhead at a time, and the next kAhead rows are requested before the current ones are looked at: the
    // launch is only a couple of waves per SIMD, so a tile row's time is its chain of round trips to memory -- five of them
    // now, overlapped with the bit work, instead of ten
    constexpr int kAhead = 16;
    static_assert(kRows % kAhead == 0, "whole batches of rows");
    Row3 up, cur;
    uint64_t nxt[kAhead], ahead[kAhead];
    up.c = word(y0 - 1);
    cur.c = word(y0);
#pragma unroll
    for (int u = 0; u < kAhead; u++) nxt[u] = word(y0 + u + 1);
    edge_bits(up.c, &up.lr);
    edge_bits(cur.c, &cur.lr);
    uint32_t qhead = 0, qtail = 0;                 // wave-uniform
    const uint32_t lane_row0 = (uint32_t)lane << 8;
    // darts of one queued word -> its tile's count and its bit of the tile's mask (LDS atomics: a few per flush)
    auto work = [&](uint32_t slot) {
        const uint64_t c = q_c[slot], uc = q_up[slot], dc = q_dn[slot];
        const uint32_t m = q_meta[slot];
        Nb8 nb;
        nb.c = c;
        nb.n[0] = (c << 1) | (m & 1u);                  nb.n[4] = (c >> 1) | ((uint64_t)((m >> 1) & 1u) << 63);
        nb.n[1] = (uc << 1) | ((m >> 2) & 1u);          nb.n[3] = (uc >> 1) | ((uint64_t)((m >> 3) & 1u) << 63);
        nb.n[2] = uc;                                   nb.n[6] = dc;
        nb.n[7] = (dc << 1) | ((m >> 4) & 1u);          nb.n[5] = (dc >> 1) | ((uint64_t)((m >> 5) & 1u) << 63);
        uint64_t p[8];
        pdart_words(nb, p);
        uint32_t ndw = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) ndw += __popcll(p[q]);
        if (ndw) {
            const uint32_t src = (m >> 8) & 63u, r = (m >> 16) & 15u, sgl = src & (G - 1), sgrp = src / G;
            const uint32_t tile = (sgl - 1u) / kTileWords, jl = (sgl - 1u) & (kTileWords - 1);
            atomicAdd(&s_tile[sgrp][tile], ndw);
            atomicOr(&s_mask[sgrp][tile], 1ull << (16u * jl + r));
        }
    };
    auto wave_sync = []() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
#pragma unroll
    for (int r0 = 0; r0 < kRows; r0 += kAhead) {
        if (r0 + kAhead < kRows) {
#pragma unroll
            for (int u = 0; u < kAhead; u++) ahead[u] = word(y0 + r0 + kAhead + u + 1);
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            Row3 dn;
            dn.c = nxt[u]; edge_bits(dn.c, &dn.lr);
            const uint64_t c = cur.c;
            // no foreground pixel, or a word in the middle of a white area (the word, the words above and below and the
            // six neighbour bits all ones: most words of a frame on white paper): no dart
            const bool white = (c & up.c & dn.c) == ~0ull && (cur.lr & up.lr & dn.lr) == 3u;
            const bool hot = owner && c != 0ull && !white;
            const unsigned long long bal = __ballot(hot);
            if (bal) {   // uniform
                if (hot) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    const uint32_t slot = (qtail + below) & (kQ - 1);
                    q_c[slot] = c; q_up[slot] = up.c; q_dn[slot] = dn.c;
                    q_meta[slot] = cur.lr | (up.lr << 2) | (dn.lr << 4) | lane_row0 | ((uint32_t)(r0 + u) << 16);
                }
                qtail += (uint32_t)__popcll(bal);
                if (qtail - qhead >= 64u) {   // uniform
                    wave_sync();
                    work((qhead + (uint32_t)lane) & (kQ - 1));
                    qhead += 64u;
                    wave_sync();
                }
            }
            up = cur; cur = dn;
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) nxt[u] = ahead[u];
    }
    wave_sync();
    if ((uint32_t)lane < qtail - qhead) work((qhead + (uint32_t)lane) & (kQ - 1));
    // a tile is four word columns: sum over its lanes through LDS (one wave per workgroup: no barrier needed)
    __builtin_amdgcn_s_waitcnt(0);   // LDS atomics of this wave have landed before it reads the totals back
    __builtin_amdgcn_wave_barrier();
    uint32_t total = 0;
    if (gl < kTilesPerGroup && ty < (int)tiles_y) {
        const uint32_t tx = (uint32_t)(cx * kTilesPerGroup + gl);
        if (tx < tiles_x) {
            total = s_tile[grp][gl];
            // part 0 goes to tile_darts itself, part z > 0 to the z-th array behind tile_off; k_tile_scan adds them up
            (blockIdx.z ? tile_darts_h1 + (size_t)(blockIdx.z - 1) * gridDim.y * (tiles_x * tiles_y) : tile_darts)[(size_t)(first_frame + f) * (tiles_x * tiles_y) + ty * tiles_x + tx] = total;
        }
    }
    if (gl >= 1 && gl <= kOwners && ((gl - 1) & (kTileWords - 1)) == 0 && ty < (int)tiles_y) {
        const uint32_t tx = (uint32_t)(cx * kTilesPerGroup + (gl - 1) / kTileWords);
        if (tx < tiles_x) tile_mask[((size_t)(first_frame + f) * (tiles_x * tiles_y) + ty * tiles_x + tx) * kCountHalves + blockIdx.z] = s_mask[grp][(gl - 1) / kTileWords];
    }
    // the wave's total: DPP row shifts and broadcasts leave it in lane 63
#define A3_DPP_ADD(CTRL, ROWMASK) total += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)total, CTRL, ROWMASK, 0xF, false);
    A3_DPP_ADD(0x111, 0xF) A3_DPP_ADD(0x112, 0xF) A3_DPP_ADD(0x114, 0xF) A3_DPP_ADD(0x118, 0xF) A3_DPP_ADD(0x142, 0xA) A3_DPP_ADD(0x143, 0xC)
#undef A3_DPP_ADD
    if (lane == 63 && total) atomicAdd(&frame_darts[f], (unsigned long long)total);
}

// Device-side plan for a batch whose contour graph is expected to fit one chunk (the host learnt its size from the previous
// batch): exclusive prefix sums of the per-frame dart counts -> frame_base[0..n], total -> plan[0] (0 and plan[1] = 1 when
// the total exceeds `cap`, which makes every later kernel a no-op; the host then re-plans with a read-back), largest
// frame -> plan[2].  One workgroup of 256 threads (the last one of k_tile_scan's launch); n_frames is small.
__device__ void plan_frames(unsigned long long* __restrict__ frame_darts, uint32_t n_frames, unsigned long long cap,
                            uint32_t* __restrict__ frame_base, uint32_t* __restrict__ plan) {
    __shared__ unsigned long long s_pw[4];
    __shared__ unsigned long long s_run;
    __shared__ unsigned int s_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_run = 0; s_max = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n_frames; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_frames ? frame_darts[i] : 0ull;
        if (i < n_frames) frame_darts[i] = 0ull;   // handed back zeroed: the next batch's k_dart_count adds to it again
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_pw[wave] = inc;
        atomicMax(&s_max, (unsigned int)min(v, 0xFFFFFFFFull));
        __syncthreads();
        unsigned long long before = s_run, tot = 0;
        for (int w = 0; w < 4; w++) { const unsigned long long t = s_pw[w]; if (w < wave) before += t; tot += t; }
        if (i < n_frames) frame_base[i] = (uint32_t)min(before + inc - v, 0xFFFFFFFFull);
        __syncthreads();
        if (threadIdx.x == 0) s_run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned long long total = s_run;
        const bool fits = total <= cap;
        frame_base[n_frames] = fits ? (uint32_t)total : 0u;
        plan[0] = fits ? (uint32_t)total : 0u;
        plan[1] = fits ? 0u : 1u;
        plan[2] = s_max;
        plan[3] = (uint32_t)min(total, 0xFFFFFFFFull);
    }
}

// Per frame: exclusive prefix sums of its tiles' dart counts, so that every tile knows its dart range without an atomic
// (and dart numbering no longer depends on scheduling).  grid: frames; tiles per frame is a few hundred.
// (Workgroup n_frames of the launch, when there is one, does the device-side plan: it needs the per-frame totals k_dart_count
// has just added up and nothing of this kernel, and a launch of its own costs 6 us for 2 us of work.)
__global__ __launch_bounds__(256) void k_tile_scan(uint32_t* __restrict__ tile_darts, const uint32_t* __restrict__ tile_darts_h1, uint32_t tiles, uint32_t first_frame,
                                                   uint32_t* __restrict__ tile_off, uint32_t n_frames,
                                                   unsigned long long* __restrict__ frame_darts, unsigned long long cap,
                                                   uint32_t* __restrict__ frame_base, uint32_t* __restrict__ plan,
                                                   uint4* __restrict__ zero_p, uint32_t zero_n16) {
    if (blockIdx.x == n_frames) {
        // the batch's block of counters (a3_api's zero block, which also holds `plan`): zeroed here, ahead of its first use
        for (uint32_t i = threadIdx.x; i < zero_n16; i += 256) zero_p[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        plan_frames(frame_darts, n_frames, cap, frame_base, plan);
        return;
    }
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_run;
    const size_t base = (size_t)(first_frame + blockIdx.x) * tiles;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < tiles; t0 += 256) {
        const uint32_t t = t0 + threadIdx.x;
        uint32_t v = 0;
        if (t < tiles) {   // the parts of k_dart_count
            v = tile_darts[base + t];
#pragma unroll
            for (int z = 1; z < kCountHalves; z++) v += tile_darts_h1[(size_t)(z - 1) * n_frames * tiles + base + t];
        }
        if (t < tiles) tile_darts[base + t] = v;   // lets k_dart_assign skip empty tiles
        uint32_t total;
        const uint32_t excl = block_excl_scan_256(v, s_wave, &total);
        if (t < tiles) tile_off[base + t] = s_run + excl;
        __syncthreads();
        if (threadIdx.x == 0) s_run += total;
        __syncthreads();
    }
}

// Same traversal; hands every border pixel a contiguous dart range inside its frame's range and writes the per-dart
// records.  Tiles without darts (most of a clean frame) leave at once.
// Phase 1 works a word (64 pixels) per lane; phase 2 re-distributes the tile's border pixels evenly over the 256 lanes
// (a word on a horizontal edge holds up to 64 of them, most words none), each lane locating its pixel by a binary
// search over the per-word prefix sums and a rank-select in the word's node mask.
// neighbour mask of pixel i of tile word (rl, jl), straight from the staged tile (ring order W NW N NE E SE S SW)
__device__ __forceinline__ uint32_t row3(uint64_t l, uint64_t c, uint64_t r, int i) {
    // pixels i-1, i, i+1 of the word as bits 0..2; pixel -1 is bit 63 of the left word, pixel 64 bit 0 of the right one
    const uint64_t v = i == 0 ? ((c << 1) | (l >> 63)) : (c >> (i - 1));
    return ((uint32_t)v & 7u) | (i == 63 ? ((uint32_t)r & 1u) << 2 : 0u);
}
// The staged tile as a bit string: a row is 6 words = 12 dwords; pixel i of tile word jl is bit (jl + 1) * 64 + i of its row,
// so the pixels i-1, i, i+1 are three consecutive bits of a 64-bit window that one two-dword LDS read delivers (no word-edge
// cases; this function runs once per border pixel and once per dart, and k_dart_assign is VALU-bound).
__device__ __forceinline__ uint32_t tile_F(const uint64_t (*s_t)[kTileWords + 2], int rl, int jl, int i) {
    const uint32_t* base = reinterpret_cast<const uint32_t*>(&s_t[0][0]);
    const int p1 = (jl + 1) * 64 + i - 1, d = p1 >> 5, sh = p1 & 31;
    constexpr int kRowDwords = 2 * (kTileWords + 2);
    uint32_t rows[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t* r = base + (rl + k) * kRowDwords + d;
        rows[k] = __builtin_amdgcn_alignbit(r[1], r[0], (uint32_t)sh) & 7u;
    }
    const uint32_t top = rows[0], mid = rows[1], bot = rows[2];
    return (mid & 1u) | ((top & 7u) << 1) | ((mid & 4u) << 2) | ((bot & 4u) << 3) | ((bot & 2u) << 5) | ((bot & 1u) << 7);
}
// position of the r-th (0-based) set bit of m; m has more than r bits set
__device__ __forceinline__ int select_bit(uint64_t m, uint32_t r) {
    uint32_t w = (uint32_t)m, c = (uint32_t)__popc(w), pos = 0;
    if (r >= c) { w = (uint32_t)(m >> 32); r -= c; pos = 32; }
    c = (uint32_t)__popc(w & 0xFFFFu); if (r >= c) { w >>= 16; r -= c; pos += 16; }
    c = (uint32_t)__popc(w & 0xFFu);   if (r >= c) { w >>= 8;  r -= c; pos += 8; }
    c = (uint32_t)__popc(w & 0xFu);    if (r >= c) { w >>= 4;  r -= c; pos += 4; }
    c = (uint32_t)__popc(w & 3u);      if (r >= c) { w >>= 2;  r -= c; pos += 2; }
    return (int)(pos + (r >= (w & 1u) ? 1u : 0u));
}
// Per neighbour mask F of a border pixel (ring order W NW N NE E SE S SW): the darts it owns (pdart_mask, bits 0-7), the direction of
// its W-event dart (bits 8-11: first foreground neighbour clockwise from W when the west side is background; 15 = none) and of its
// E-event dart (bits 12-15).  256 entries computed at compile time; k_dart_assign keeps a copy in LDS: two loads instead of ~ 35
// instructions per border pixel, one instead of 8 per successor.
struct PixLut { uint16_t v[256]; };
constexpr PixLut make_pix_lut() {
    PixLut t{};
    for (uint32_t F = 0; F < 256; F++) {
        const uint32_t rot1 = ((F << 1) | (F >> 7)) & 0xFFu, rot2 = ((F << 2) | (F >> 6)) & 0xFFu;
        const uint32_t P = F & ~rot1 & (0xAAu | ~rot2) & 0xFFu;                    // pdart_mask (a3_common.h)
        uint32_t kW = 15, kE = 15;
        if (!(F & 1u)) { kW = 0; for (uint32_t b = 0; b < 7; b++) if ((F >> 1) & (1u << b)) { kW = b + 1; break; } }
        if (!(F & 16u)) {
            const uint32_t rr = ((F >> 5) | (F << 3)) & 0xFFu;                      // bit t <-> direction (5 + t) & 7
            uint32_t first = 0;                                                     // __ffs(rr): 0 when rr == 0 (then F == 0: no dart anyway)
            for (uint32_t b = 0; b < 8; b++) if (rr & (1u << b)) { first = b + 1; break; }
            kE = (5 + first - 1) & 7;
        }
        t.v[F] = (uint16_t)(P | (kW << 8) | (kE << 12));
    }
    return t;
}
__device__ __constant__ const PixLut kPixLut = make_pix_lut();

// kDX / kDY without a table load
__device__ __forceinline__ int dir_dx(int k) { return (int)((0x1A90u >> (2 * k)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int k) { return (int)((0xA901u >> (2 * k)) & 3u) - 1; }

// (Staging four stacked tiles per workgroup was tried here: 138 us instead of 125 us -- the tiles of a
// stack are then worked through one after the other and the barriers in between cost more than the shared staging saves.)
__global__ __launch_bounds__(256) void k_dart_assign(const uint64_t* __restrict__ bits, int W, int H, uint32_t first_frame,
                                                     const uint32_t* __restrict__ frame_base, const uint32_t* __restrict__ tile_off,
                                                     uint32_t* __restrict__ pix_base, const uint32_t* __restrict__ tile_darts,
                                                     uint64_t* __restrict__ d_rec, uint32_t* __restrict__ d_succ,
                                                     const uint32_t* __restrict__ n_live, int dbg,
                                                     const unsigned long long* __restrict__ tile_mask, uint32_t tiles, uint32_t n_frames) {
    // dbg (a3_debug_kernel_time only; 0 in the product path): stop after 1 = the empty-tile test, 2 = phase 1 and its scans,
    // 3 = the range allocation; 4 = run phase 2 without its global stores
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_t[kTileRows + 2][kTileWords + 2];
    __shared__ uint64_t s_nodes[256];
    __shared__ uint64_t s_c0[256], s_c1[256], s_c2[256];   // darts per pixel (0..4) as three bit planes
    __shared__ uint32_t s_dbase[256];
    // Locating the n-th border pixel of the tile (phase 2) without a search.  Phase 1 numbers the words that own darts (rank k: its
    // lane index); behind the scan every such word w leaves its first pixel's index under its rank (s_nbk), a marker bit at that
    // index - 1 (s_mark: one bit per border pixel of the tile, 64 per segment), and -- if it holds the first pixel of a 64-pixel
    // segment -- its rank as the segment's first (s_fr).  A wave of phase 2 works on exactly one segment, lane l on its pixel l: the
    // rank of that pixel's word is s_fr + the markers below bit l (v_mbcnt), where the 8-step binary search over the 256 prefix
    // sums used to be 64 of the ~ 230 instructions a border pixel costs before its darts.
    __shared__ uint32_t s_mark[512];
    __shared__ uint16_t s_nbk[256];
    __shared__ uint8_t s_wofrank[256], s_rank[256], s_fr[256];
    __shared__ uint32_t s_multi;   // some pixel of the tile owns more than one dart
    __shared__ uint16_t s_lut[256];
    const int wpr = (int)words_per_row((uint32_t)W);
    // XCD-aware workgroup -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, and its L2).  A tile
    // row of the packed image is 32 bytes of a 128-byte line, and with (tile, frame) = (blockIdx.x, blockIdx.y) on a 1920-pixel frame
    // -- 8 tiles side by side -- XCD k got tile column k of EVERY frame: each XCD fetched every line of every frame, 251 MB from the
    // fabric per 256 frames for an image of 66 MB.  Here XCD k works through one contiguous eighth of the (frame, tile) sequence in
    // order: a line's tiles meet in one L2, and so do the apron rows that vertical neighbours share.
    const uint32_t n_wg = tiles * n_frames, chunk = (n_wg + 7u) >> 3, g = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (g >= n_wg) return;
    const uint32_t f = g / tiles, tile = g - f * tiles;
    if (tile_darts[(size_t)(first_frame + f) * tiles + tile] == 0u) return;   // uniform for the workgroup
    if (n_live && *n_live == 0u) return;   // device-side plan: the graph does not fit the pool, the host re-plans
    const uint32_t dart0 = frame_base[f] + tile_off[(size_t)(first_frame + f) * tiles + tile];
    if (dbg == 1) return;
    // Phase 1 works on the words that own darts only (k_dart_count's masks say which: typically 60 of a tile's 256, so one wave
    // does what four did -- the kernel is bound by its instruction count), lane k taking the k-th of them; every word's entry
    // in the per-word arrays starts at zero.
    __shared__ uint32_t s_cnt[256];   // per word: darts (low 17 bits) | border pixels << 17
    const unsigned long long* tm = tile_mask + ((size_t)(first_frame + f) * tiles + tile) * kCountHalves;
    static_assert(kCountHalves == 4 && kTileWords == 4 && kTileRows == 64, "four 64-bit word masks per tile, four word columns of 64 rows");
    // Words are NUMBERED -- and with them the tile's darts -- column by column: w = jl * 64 + rl (word column jl = 64 pixels wide, row
    // rl).  Consecutive darts then fill blocks about as tall as wide instead of strips 256 pixels wide and a few rows high, and
    // k_local_contract's tiles of consecutive darts have less than half the rim: fewer windows that freeze, fewer entries for the
    // global rounds, more short borders that close (and are finished with) inside a tile.  Matters on dense graphs only (noise-like
    // frames: 16 k darts per tile); a clean frame's tile holds a few hundred darts.  k_dart_count's masks come column by column (bit
    // jl * 16 + r of part p = word jl of row 16 p + r): a column's 64 rows are 16 bits of each of the four parts.
    unsigned long long cm[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        cm[c] = ((tm[0] >> (16 * c)) & 0xFFFFull) | (((tm[1] >> (16 * c)) & 0xFFFFull) << 16) | (((tm[2] >> (16 * c)) & 0xFFFFull) << 32) | (((tm[3] >> (16 * c)) & 0xFFFFull) << 48);
    const unsigned long long m0 = cm[0], m1 = cm[1], m2 = cm[2], m3 = cm[3];
    tile_stage(bits + (size_t)(first_frame + f) * wpr * H, W, H, tile, s_t);
    // (s_nodes / s_c0..2 are read for words that own darts only, and phase 1 writes those: nothing to zero there)
    s_cnt[threadIdx.x] = 0;
    s_mark[threadIdx.x] = 0; s_mark[256 + threadIdx.x] = 0;
    s_lut[threadIdx.x] = kPixLut.v[threadIdx.x];
    if (threadIdx.x == 0) s_multi = 0;
    __syncthreads();
    const uint32_t a0 = (uint32_t)__popcll(m0), a1 = a0 + (uint32_t)__popcll(m1), a2 = a1 + (uint32_t)__popcll(m2), n_act = a2 + (uint32_t)__popcll(m3);
    for (uint32_t k = threadIdx.x; k < n_act; k += 256) {
        const uint32_t part = (k >= a0) + (k >= a1) + (k >= a2);
        const unsigned long long mm = part == 0 ? m0 : (part == 1 ? m1 : (part == 2 ? m2 : m3));
        const uint32_t w = part * 64u + (uint32_t)select_bit(mm, k - (part == 0 ? 0u : (part == 1 ? a0 : (part == 2 ? a1 : a2))));   // (part = word column)
        const Nb8 nb = tile_nb8(s_t, (int)(w & 63u), (int)(w >> 6));
        uint64_t p[8];
        pdart_words(nb, p);
        uint64_t nodes = 0;
        uint32_t nd = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { nd += __popcll(p[q]); nodes |= p[q]; }
        // bit-sliced sum of the eight direction planes (a pixel owns at most 4 darts)
        const uint64_t s1 = p[0] ^ p[1] ^ p[2], k1 = (p[0] & p[1]) | (p[2] & (p[0] ^ p[1]));
        const uint64_t s2 = p[3] ^ p[4] ^ p[5], k2 = (p[3] & p[4]) | (p[5] & (p[3] ^ p[4]));
        const uint64_t s3 = p[6] ^ p[7], k3 = p[6] & p[7];
        const uint64_t k4 = (s1 & s2) | (s3 & (s1 ^ s2));
        const uint64_t t = k1 ^ k2 ^ k3, q1 = (k1 & k2) | (k3 & (k1 ^ k2));
        s_nodes[w] = nodes;
        const uint64_t c1 = t ^ k4, c2 = q1 | (t & k4);
        s_c0[w] = s1 ^ s2 ^ s3; s_c1[w] = c1; s_c2[w] = c2;
        if ((c1 | c2) != 0ull) s_multi = 1u;   // (a plain store of the same value from whoever sees it)
        s_cnt[w] = nd | ((uint32_t)__popcll(nodes) << 17);
        s_wofrank[k] = (uint8_t)w; s_rank[w] = (uint8_t)k;
    }
    __syncthreads();
    // one block scan for both counts: darts (<= 65536 per tile) in the low 17 bits, border pixels (<= 16384) above
    uint32_t total_p;
    const uint32_t excl_p = block_excl_scan_256(s_cnt[threadIdx.x], s_wave, &total_p);
    const uint32_t excl_d = excl_p & 0x1FFFFu, excl_n = excl_p >> 17, total_n = total_p >> 17;
    s_dbase[threadIdx.x] = excl_d;
    {   // (thread = word) a word with border pixels: first index under its rank, marker, first-of-segment
        const uint32_t cw = s_cnt[threadIdx.x] >> 17;
        if (cw) {
            const uint32_t k = s_rank[threadIdx.x];
            s_nbk[k] = (uint16_t)excl_n;
            if (excl_n) atomicOr(&s_mark[(excl_n - 1u) >> 5], 1u << ((excl_n - 1u) & 31u));
            const uint32_t seg = (excl_n + 63u) >> 6;               // the first segment boundary at or behind the word's first pixel
            if ((seg << 6) < excl_n + cw) s_fr[seg] = (uint8_t)k;   // (a word holds at most 64 border pixels: at most one boundary)
        }
    }
    if (dbg == 2) return;
    __syncthreads();
    if (dbg == 3) return;
    const int tx = tile % dart_tiles_x((uint32_t)W), ty = tile / dart_tiles_x((uint32_t)W);
    uint32_t* pbf = pix_base + (size_t)f * W * H;
    // No pixel of the tile owns more than one dart -- on clean frames 999 border pixels in a thousand own exactly one, so most tiles
    // qualify: a pixel's darts before it in its word are then the border pixels before it (one popcount of the node mask that is in
    // hand anyway) instead of the sum over three bit planes, for the pixel itself and for its successor's pixel.
    const bool single = __builtin_amdgcn_readfirstlane((int)s_multi) == 0;
    for (uint32_t n = threadIdx.x; n < total_n; n += 256) {
        // word holding the n-th border pixel of the tile: the segment's first word + the words that begin inside the segment at or
        // before this pixel (n = threadIdx.x + 256 t: a wave's lanes are the 64 pixels of segment n >> 6, lane = n & 63)
        const uint32_t seg = n >> 6;
        const uint32_t marks_below = __builtin_amdgcn_mbcnt_hi(s_mark[2u * seg + 1u], __builtin_amdgcn_mbcnt_lo(s_mark[2u * seg], 0u));
        const uint32_t kr = (uint32_t)s_fr[seg] + marks_below;
        const uint32_t w = s_wofrank[kr], r = n - (uint32_t)s_nbk[kr];
        const uint64_t m = s_nodes[w];
        const int i = select_bit(m, r);
        const int jl = w >> 6, rl = w & 63;
        const int wj = tx * kTileWords + jl, wy = ty * kTileRows + rl;
        const uint32_t F = tile_F(s_t, rl, jl, i);
        const uint32_t lut = s_lut[F];
        uint32_t P = lut & 0xFFu;
        const uint64_t below = (1ull << i) - 1ull;
        // darts of the word's earlier border pixels
        const uint32_t off = single ? (uint32_t)__popcll(m & below)
                                    : (uint32_t)__popcll(s_c0[w] & below) + 2u * (uint32_t)__popcll(s_c1[w] & below) + 4u * (uint32_t)__popcll(s_c2[w] & below);
        const int x = 64 * wj + i;
        uint32_t cur = dart0 + s_dbase[w] + off;
        // only pixels on the rim of the tile can be the target of a successor pointer from another tile
        if ((i == 0 || i == 63 || rl == 0 || rl == kTileRows - 1) && dbg != 4) pbf[(size_t)wy * W + x] = cur;
        // event darts: first foreground neighbour clockwise from W (resp. E) when that side is background (kPixLut); no W-event in
        // column 0, no E-event in the last column (the reference's x > 0 / x + 1 < width guards)
        const int kW = x > 0 ? (int)((lut >> 8) & 15u) : 15, kE = x + 1 < W ? (int)(lut >> 12) : 15;
        const uint32_t xy = (uint32_t)x | ((uint32_t)wy << 16);
        while (P) {
            const int k = __ffs(P) - 1;
            P &= P - 1;
            const uint32_t info = (uint32_t)k | (k == kW ? kInfoW : 0u) | (k == kE ? kInfoE : 0u);
            if (dbg != 4) d_rec[cur] = dart_rec(xy, F, info, f);   // one 8-byte store per dart
            // successor: next foreground neighbour counter-clockwise after k; resolved here when the target pixel lies in
            // this tile (its dart indices follow from the tile's prefix sums), otherwise left to k_dart_link
            const uint32_t rr = ((F >> k) | (F << (8 - k))) & 0xFFu;   // bit t <-> direction (k + t) & 7
            const int ko = (k + (31 - __clz(rr))) & 7, kin = (ko + 4) & 7;
            const int lx = jl * 64 + i + dir_dx(ko), ly = rl + dir_dy(ko);
            uint32_t succ;
            if (lx >= 0 && lx < kTileWords * 64 && ly >= 0 && ly < kTileRows) {
                const int j2 = lx >> 6, i2 = lx & 63, w2 = j2 * kTileRows + ly;
                const uint32_t P2 = (uint32_t)s_lut[tile_F(s_t, ly, j2, i2)] & 0xFFu;
                succ = cur;   // chain end unless the target dart exists
                if ((P2 >> kin) & 1u) {
                    const uint64_t below2 = (1ull << i2) - 1ull;
                    const uint32_t off2 = single ? (uint32_t)__popcll(s_nodes[w2] & below2)
                                                 : (uint32_t)__popcll(s_c0[w2] & below2) + 2u * (uint32_t)__popcll(s_c1[w2] & below2) +
                                                   4u * (uint32_t)__popcll(s_c2[w2] & below2);
                    succ = dart0 + s_dbase[w2] + off2 + (uint32_t)__popc(P2 & ((1u << kin) - 1u));
                }
            } else {
                succ = kNone;   // only rim pixels get here; k_dart_link fills these in
            }
            if (dbg != 4 || succ == 0x12345678u) d_succ[cur] = succ;
            cur++;
        }
    }
}

// neighbour mask of one pixel straight from the packed image (ring order W NW N NE E SE S SW): one word per row, plus the
// neighbouring word only when the pixel sits on a word edge (columns >= W and everything outside the image read as zeros)
__device__ __forceinline__ uint32_t pixel_F(const uint64_t* __restrict__ img, int wpr, int H, int x, int y) {
    const int j = x >> 6, i = x & 63;
    uint32_t rows[3];
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
        const uint64_t c = ldw(img, wpr, H, j, y + dy);
        const uint64_t l = i == 0 ? ldw(img, wpr, H, j - 1, y + dy) : 0ull;
        const uint64_t r = i == 63 ? ldw(img, wpr, H, j + 1, y + dy) : 0ull;
        rows[dy + 1] = row3(l, c, r, i);
    }
    const uint32_t top = rows[0], mid = rows[1], bot = rows[2];
    return (mid & 1u) | ((top & 7u) << 1) | ((mid & 4u) << 2) | ((bot & 4u) << 3) | ((bot & 2u) << 5) | ((bot & 1u) << 7);
}

// A successor that leaves its tile (k_dart_assign left kNone): looked up through the target pixel's first-dart index
// (pix_base, written for tile-rim pixels that own darts); whether it owns the wanted dart follows from its neighbour mask.
// A compact list of these darts would need one same-address atomic per tile (~11 ns each, serialised); sweeping d_succ
// is cheaper (32 us).  Resolving them inside k_local_contract's load phase was tried: the dependent loads of the few
// cross-tile darts stall every tile's first barrier (+47 us).
__device__ __forceinline__ uint32_t cross_tile_succ(uint32_t d, uint64_t rec, int W, int H, int wpr, uint32_t first_frame,
                                                    const uint32_t* __restrict__ pix_base, const uint64_t* __restrict__ bits) {
    const uint32_t f = rec_frame(rec);
    const uint64_t* img = bits + (size_t)(first_frame + f) * wpr * H;
    const uint32_t* pb = pix_base + (size_t)f * W * H;
    const uint32_t xy = rec_xy(rec);
    const int x = xy & 0xFFFF, y = xy >> 16;
    const uint32_t F = rec_F(rec);
    const int k = rec_info(rec) & 7;
    const uint32_t r = ((F >> k) | (F << (8 - k))) & 0xFFu;  // bit t <-> direction (k + t) & 7
    const int ko = (k + (31 - __clz(r))) & 7;
    const int nx = x + dir_dx(ko), ny = y + dir_dy(ko);
    const uint32_t tP = pdart_mask(pixel_F(img, wpr, H, nx, ny));   // the target is a foreground pixel by construction
    const int kin = (ko + 4) & 7;
    return ((tP >> kin) & 1u) ? pb[(size_t)ny * W + nx] + __popc(tP & ((1u << kin) - 1u)) : d;  // d = chain end
}

__global__ __launch_bounds__(256) void k_dart_link(int W, int H, uint32_t first_frame, const uint32_t* __restrict__ pix_base,
                                                   const uint64_t* __restrict__ bits, const uint64_t* __restrict__ d_rec,
                                                   uint32_t* __restrict__ d_succ, uint32_t n_darts, const uint32_t* __restrict__ n_live) {
    if (n_live) n_darts = min(n_darts, *n_live);
    const int wpr = (int)words_per_row((uint32_t)W);
    // A wave sweeps 8 x 64 successors per step (eight loads in flight per lane); the few that are open (~4 %) are then dealt
    // out over the lanes -- lane j takes the j-th open dart of the step, found from the eight ballot masks -- so that the
    // chain of dependent loads behind an open successor (record -> image words -> pix_base) is walked once per step by ~20
    // busy lanes, not once per load by one or two.
    constexpr int B = 8;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t w0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); w0 < n_darts; w0 += B * stride) {   // wave-uniform
        uint32_t sv[B];
#pragma unroll
        for (int u = 0; u < B; u++) sv[u] = d_succ[min(w0 + lane + (uint32_t)u * stride, n_darts - 1u)];
        uint64_t m[B];
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < B; u++) {
            m[u] = __ballot(w0 + lane + (uint32_t)u * stride < n_darts && sv[u] == kNone);
            total += (uint32_t)__popcll(m[u]);
        }
        for (uint32_t j = lane; j < total; j += 64u) {
            uint32_t acc = 0, r = 0, su = 0;
            uint64_t mm = 0;
#pragma unroll
            for (int u = 0; u < B; u++) {
                const uint32_t c = (uint32_t)__popcll(m[u]);
                if (j >= acc && j < acc + c) { mm = m[u]; r = j - acc; su = (uint32_t)u; }
                acc += c;
            }
            const uint32_t d = w0 + (uint32_t)select_bit(mm, r) + su * stride;
            d_succ[d] = cross_tile_succ(d, d_rec[d], W, H, wpr, first_frame, pix_base, bits);
        }
    }
}

// ---------------------------------------------------------------------------------------
// cycle leaders and ranks: pointer doubling, first inside LDS tiles, then over tile-crossing darts only
// ---------------------------------------------------------------------------------------
// A window is a path [d, ptr): `dist` hops long, `key` = smallest (event key, dart) on it, `off` = hops from d to that
// dart.  Joining window(d) with window(ptr) doubles the path; once a window wraps a whole cycle its key is the cycle's
// leader and off the hop distance to it.
// Darts per local tile (consecutive indices = neighbouring pixels): k_local_contract<LT>.  2048 with the global doubling rounds
// (noise-like frames: every tile boundary a chain crosses makes an entry, and there the entries are the expensive part: 1024
// costs the reference's noise recipe 19 %), 1024 with the per-frame entry resolution of clean frames, where the entries are
// cheap and the smaller tile saves a doubling round and half of the LDS per workgroup (72 -> 60 us on BASELINE config 2;
// 512: 68 us, 4096: 103 us).
#ifndef A3_KLT
#define A3_KLT 2048
#endif
constexpr int kLT = A3_KLT;                   // the larger of the two: sizes the entry slot space of the global rounds
constexpr int kLTFrame = 1024;
// Entry slots are handed out from 16 counters (one same-address atomic costs ~11 ns and they serialise): tile t uses
// shard t & 15, whose slots are [shard * cap, shard * cap + count[shard]); a tile holds at most kLT entries, so
// cap = ceil(tiles / 16) * kLT can never overflow.
constexpr uint32_t kEntryShards = 16;
__host__ __device__ inline uint32_t entry_shard_cap(uint32_t n_darts) {
    const uint32_t tiles = (n_darts + kLT - 1) / kLT;
    return ((tiles + kEntryShards - 1) / kEntryShards) * kLT;
}
// the sparse slot space as a dense loop index: shard-major, every shard padded to the largest count
struct EntrySpace {
    uint32_t cnt[kEntryShards], span, total;
    __device__ explicit EntrySpace(const unsigned int* __restrict__ entry_count) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t sh = 0; sh < kEntryShards; sh++) { cnt[sh] = entry_count[sh]; m = max(m, cnt[sh]); }
        span = max((m + 63u) & ~63u, 64u);
        total = span * kEntryShards;
    }
    // -> slot, or kNone for padding
    __device__ uint32_t slot(uint32_t i0, uint32_t cap) const {
        const uint32_t sh = i0 / span, i = i0 - sh * span;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kEntryShards; k++) c = (k == sh) ? cnt[k] : c;
        return i < c ? sh * cap + i : kNone;
    }
};
constexpr uint32_t kFrozen = 0x80000000u;     // the window reached a dart outside the tile
// k_local_contract's result per dart is one JumpState whose `off` word carries three things: hops to the window's minimum
// (bits 0-11, <= 2047), window length in hops (bits 12-24, <= 2048) and kFrozen
__device__ __forceinline__ uint32_t loc_pack(uint32_t off, uint32_t dist, bool frozen) { return off | (dist << 12) | (frozen ? kFrozen : 0u); }
__device__ __forceinline__ uint32_t loc_off(uint32_t w) { return w & 0xFFFu; }
// After k_jump_finalize the `off` word of a dart is either still the packed local one (the local window was final: its low 12
// bits are the hop count) or kFinal | hops to the leader (< 2^30)
constexpr uint32_t kFinal = 0x40000000u;
// k_local_contract: the dart lies on a DEAD cycle -- one that closed inside its tile, carries a start event, is too short ever to be
// materialised (k_cycle_select's parity-safe length test) and whose smallest event fires whatever the other borders do (see
// static_fire).  Nothing downstream lists, evaluates or scatters it; its key stays in place: the start resolution of OTHER
// borders through shared pixels reads it (a traced border labels its pixels however short it is).  Only in states whose window
// did not freeze (kFinal and kDead never meet: k_jump_finalize rewrites frozen windows only).
constexpr uint32_t kDead = 0x20000000u;
// (readers test both bits: in a kFinal word bit 29 belongs to the hop count -- a dart 2^29 or more hops from its leader is not a dead one)
__device__ __forceinline__ bool is_dead(uint32_t off_word) { return (off_word & (kFinal | kDead)) == kDead; }
__device__ __forceinline__ uint32_t fin_off(uint32_t w) { return (w & kFinal) ? (w & 0x3FFFFFFFu) : (w & 0xFFFu); }
__device__ __forceinline__ uint32_t loc_dist(uint32_t w) { return (w >> 12) & 0x1FFFu; }
// What every dart knows once k_jump_finalize has run, in 8 bytes: its cycle's leader and its hop distance to it, plus whether the
// cycle carries a start event and whether it is a dead one (k_local_contract).  Round 6: the sweeps that need the final answer per dart
// (k_scatter_points, the fixpoint passes) read this array, and k_jump_finalize writes it front to back instead of patching a third
// of the 16-byte local states in place (93 MB of scattered line write-backs per 256 frames, and 98 MB for k_scatter_points to read).
// The 16-byte LOCAL states stay as k_local_contract left them; a LEADER's local key is already final (its own key is the smallest
// of its cycle, hence of every window that holds it).  k_cycle_select parks a listed leader's border slot in the hops field of the
// leader's OWN FinState (a leader is 0 hops from itself; kFinHops = no slot): k_scatter_points then finds "does my leader lead" and
// "where do its points go" in one 8-byte load -- and the test is made on FINAL states: in a run that has not converged (too few
// global rounds: the batch is re-run) a dart may name a leader that does not hold its own key, and nothing of such a dart is used.
struct __attribute__((aligned(8))) FinState { uint32_t leader; uint32_t w; };
constexpr uint32_t kFinEvent = 0x80000000u, kFinDead = 0x40000000u, kFinHops = 0x3FFFFFFFu;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load, store and returning
// atomic the wave has in flight (s_waitcnt vmcnt(0)); where the barrier only hands LDS data between waves that wait would
// put the round trip of an unrelated global access on the critical path.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Does the start event of this dart fire under EVERY assignment of starts to the other borders?  (k_resolve_eval's rule: a W-event
// fires iff every cycle through its pixel starts at or after it -- certain when the pixel owns this one dart only; an E-event
// fires iff not (the pixel has a W side and that W-event fires) -- certain when it has no W side.)  A border whose SMALLEST event
// passes this test starts there, whatever else happens: it needs no look at its neighbours' states.
__device__ __forceinline__ bool static_fire(uint64_t rec) {
    const uint32_t info = rec_info(rec), F = rec_F(rec);
    if (info & kInfoW) return __popc(pdart_mask(F)) == 1;
    if (info & kInfoE) return (rec_xy(rec) & 0xFFFFu) == 0u || (F & 1u) != 0u;   // no W side: column 0, or the west neighbour is foreground
    return false;
}

// Phase 1: 11 doubling rounds inside one tile, entirely in LDS.  A window stops growing ("freezes") when its end leaves
// the tile; cycles that close inside the tile finish here.  Darts that some frozen window ends on become "entries".
template <int LT, bool DEAD /* the early finish of short borders (dense graphs) is compiled in: its LDS and registers cost the clean-frame instantiation a wave per SIMD */>
__global__ __launch_bounds__(256, (LT >= 2048 || DEAD) ? 4 : 8) void k_local_contract(uint32_t n_darts, int W, const uint64_t* __restrict__ d_rec,
                                                        const uint32_t* __restrict__ d_succ,
                                                        JumpState* __restrict__ loc,
                                                        uint32_t* __restrict__ entry_list,
                                                        unsigned int* __restrict__ entry_count, uint32_t ecap,
                                                        const uint32_t* __restrict__ frame_base, uint32_t* __restrict__ frame_entries,
                                                        const uint32_t* __restrict__ n_live, int dbg,
                                                        uint32_t min_edge_length, unsigned int* __restrict__ dead_count /*[16]; nullptr: mark nothing dead*/,
                                                        int trust_natural /* the launch sequence ends in k_cycle_select's inline check (no fixpoint passes): see the epilogue */) {
    // frame_entries != nullptr: entries get slots grouped by frame (slot = frame_base[f] + running count of the frame: a
    // frame has at most as many entries as darts), for k_entry_frame; else the 16-shard allocation of the global rounds
    // dbg (a3_debug_kernel_time only; 0 in the product path): n > 0 runs n doubling rounds instead of 11; -1 = none
    // one 16-byte record per dart (a single ds_read_b128 fetches the next window): off <= 2047 and dist <= 2048 share a word
    struct __attribute__((aligned(16))) Win { uint64_t key; uint32_t ptr; uint32_t offdist; };
    __shared__ Win s_win[LT];
    constexpr uint32_t kFrameWin = 64;               // frames a tile may span with block-aggregated counting (beyond: direct atomics)
    __shared__ uint32_t s_fcnt[kFrameWin], s_fbase[kFrameWin];
    __shared__ uint32_t s_new_count, s_new_base, s_f0, s_dead_n;
    __shared__ __attribute__((aligned(4))) uint8_t s_dead[DEAD ? LT : 4];   // per dart of the tile: bit 0 = it LEADS a dead cycle, bit 1 = its own start event passes static_fire
    constexpr uint32_t kQCap = DEAD ? 1024 : 1;      // leaders whose border may be finished here, queued for the epilogue (a tile of noise holds ~200)
    __shared__ uint32_t s_q[kQCap];
    __shared__ uint32_t s_qn;
    static_assert(LT <= 2048, "a queue entry holds two 11-bit tile indices");
    if constexpr (!DEAD) dead_count = nullptr;
    if (n_live) n_darts = min(n_darts, *n_live);
    const uint32_t lo = blockIdx.x * LT;
    if (lo >= n_darts) return;   // the grid covers the pool's capacity, the graph may be smaller
    const uint32_t cnt = min((uint32_t)LT, n_darts - lo);
    // every lane owns 8 darts (i = lane + 256 u) and keeps their window state in registers across the rounds; LDS holds the
    // copy the other lanes read (per dart and round: one 16-byte read of the next window, one 16-byte write)
    constexpr int PER = LT / 256;
    uint64_t nk[PER]; uint32_t np[PER], no[PER], nd[PER], succ0[PER], frm[PER];
    if (threadIdx.x == 0) { s_new_count = 0; s_dead_n = 0; s_qn = 0; }
    if (threadIdx.x < kFrameWin) s_fcnt[threadIdx.x] = 0;
    if (dead_count) { for (uint32_t i4 = threadIdx.x; i4 < LT / 4; i4 += 256) reinterpret_cast<uint32_t*>(s_dead)[i4] = 0u; }
    uint32_t sfire = 0;                              // bit u: the start event of my dart u fires unconditionally (static_fire)
    uint32_t ecand = 0, epix = 0;                    // bit u: an event that is not static (a W-event on a pixel with several darts, an E-event on a pixel with a W side: the epilogue looks at the pixel's other darts); epix: 4 bits per dart = my rank among the pixel's darts | (their number - 1) << 2
    // all 16 loads of a lane are issued before the first is used: unconditional, from a clamped index (behind an `if (i < cnt)`
    // the compiler issues them one at a time, each with its own wait: 16 round trips to memory instead of one)
    uint64_t recs[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t d = lo + min(threadIdx.x + u * 256u, cnt - 1u);
        recs[u] = d_rec[d];
        succ0[u] = d_succ[d];
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t i = threadIdx.x + u * 256;
        const uint64_t rec = recs[u];
        const uint32_t xy = rec_xy(rec), info = rec_info(rec);
        const uint32_t q = (xy >> 16) * (uint32_t)W + (xy & 0xFFFF);
        const uint32_t ek = (info & kInfoW) ? 2u * q : ((info & kInfoE) ? 2u * q + 1u : kNoKey);
        frm[u] = rec_frame(rec);   // kept, like succ0, for the entry registration (no second read)
        // A successor outside the tile freezes the window at once; its pointer then names the dart INSIDE the tile that the window
        // ends on -- lo + LT + i: outside [lo, lo + cnt) whatever the arithmetic wraps to -- because that dart's lane knows the slot
        // the successor is registered under as an entry (below), which is what everybody downstream wants from a frozen window.
        nk[u] = ((uint64_t)ek << 32) | (lo + i); np[u] = (succ0[u] - lo) < cnt ? succ0[u] : lo + (uint32_t)LT + i; no[u] = 0; nd[u] = 1;
        if (i < cnt) s_win[i] = Win{nk[u], np[u], 1u << 16};
        if (dead_count && ek != kNoKey) {
            if (static_fire(rec)) sfire |= 1u << u;
            else {                      // (an E-event: the pixel has a W side, and with it a W-event on another of its darts; a W-event: other darts)
                const uint32_t P = pdart_mask(rec_F(rec));
                ecand |= 1u << u;
                epix |= ((uint32_t)__popc(P & ((1u << (info & 7u)) - 1u)) | ((uint32_t)(__popc(P) - 1) << 2)) << (4 * u);
            }
        }
    }
    if (threadIdx.x == 0) s_f0 = frm[0];   // darts are frame-major: the tile's frames are f0, f0+1, ...
    lds_barrier();
    if (dead_count && sfire) {             // (after the barrier: s_dead was zeroed word-wise by other lanes; read only in the epilogue)
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (sfire & (1u << u)) s_dead[threadIdx.x + u * 256] = 2;
    }
    // Entries.  The entries of the reduced list are exactly the successors that lie outside their predecessor's tile; succ is
    // injective, so each is registered once, by that predecessor, with no global dedupe.  Slots are counted per workgroup
    // (one global atomic per tile and frame) -- and counted HERE, before the doubling rounds, which need nothing of it: the
    // returning atomics are then in flight while the rounds run, instead of being two more round trips to memory at the end
    // of every workgroup's life.
    const uint32_t f0 = frame_entries ? s_f0 : 0u;
    uint32_t my_slot[PER];
    uint32_t direct = 0;   // bit u: my_slot[u] is the slot itself, not a rank inside (tile, frame)
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t i = threadIdx.x + u * 256;
        my_slot[u] = kNone;
        if (i >= cnt) continue;
        const uint32_t s0 = succ0[u];
        if ((s0 - lo) >= cnt) {
            if (frame_entries) {   // a border never leaves its frame: the successor's frame is this dart's
                const uint32_t f = frm[u];
                if (f - f0 < kFrameWin) my_slot[u] = atomicAdd(&s_fcnt[f - f0], 1u);          // rank inside (tile, frame)
                else {                                                                           // a tile over > 64 tiny frames
                    my_slot[u] = frame_base[f] + atomicAdd(&frame_entries[f], 1u);
                    direct |= 1u << u;
                }
            } else my_slot[u] = atomicAdd(&s_new_count, 1u);
        }
    }
    lds_barrier();
    // (threads < kFrameWin, resp. thread 0): first slot of the tile's entries in that frame / shard = slot_a + slot_b, added
    // only after the rounds so that nothing waits for the atomic's return before
    uint32_t slot_a = 0, slot_b = 0;
    if (frame_entries) {
        if (threadIdx.x < kFrameWin && s_fcnt[threadIdx.x]) {
            slot_a = frame_base[f0 + threadIdx.x];
            slot_b = atomicAdd(&frame_entries[f0 + threadIdx.x], s_fcnt[threadIdx.x]);
        }
    } else if (threadIdx.x == 0) {
        slot_a = (blockIdx.x & (kEntryShards - 1)) * ecap;
        if (s_new_count) slot_b = atomicAdd(&entry_count[blockIdx.x & (kEntryShards - 1)], s_new_count);
    }
    // (Skipping windows that are already final -- frozen, or wrapped, visible as "the next window has the same minimum" --
    // was tried: most windows of a clean frame only become final in the last rounds, and the extra flags made it 20 % slower.)
    constexpr int kLocalRounds = LT == 4096 ? 12 : (LT == 2048 ? 11 : (LT == 1024 ? 10 : 9));   // 2^rounds >= LT
    const int n_rounds = dbg == 0 ? kLocalRounds : (dbg < 0 ? 0 : dbg);
    // Dense graphs (the global-rounds path: noise-like frames): most cycles are a handful of darts long and complete after three or
    // four rounds.  A window that meets its own minimum again in the window it is joined with has wrapped its cycle (keys are
    // unique: the same dart at two positions of the walk) -- its key and offset are final, it takes no further part (`done`), and
    // a window that later joins it becomes complete in turn.  On clean frames most windows only complete in the last rounds and
    // the test buys nothing (round 3: -20 %), so it is made on the dense path only.
    const bool early = frame_entries == nullptr;
    uint32_t done = 0;
    for (int round = 0; round < n_rounds; round++) {
        uint32_t upd = 0;   // bit u: window u was joined with its successor window in this round (a frozen one is not: its
                            // LDS copy stays as it is and need not be written again)
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t i = threadIdx.x + u * 256;
            if (i < cnt && !((done >> u) & 1u)) {
                const uint32_t t = np[u] - lo;          // unsigned: also catches ptr < lo
                if (t < cnt) {
                    const Win w = s_win[t];
                    if (early && w.key == nk[u]) done |= 1u << u;
                    if (w.key < nk[u]) { nk[u] = w.key; no[u] = nd[u] + (w.offdist & 0xFFFFu); }
                    nd[u] += w.offdist >> 16;
                    np[u] = w.ptr;
                    upd |= 1u << u;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t i = threadIdx.x + u * 256;
            if (upd & (1u << u)) s_win[i] = Win{nk[u], np[u], no[u] | (nd[u] << 16)};
        }
        if (round == n_rounds - 1) {   // the entry slots' bases ride on the last round's barrier (the atomics have long returned)
            if (frame_entries) { if (threadIdx.x < kFrameWin) s_fbase[threadIdx.x] = slot_a + slot_b; }
            else if (threadIdx.x == 0) s_new_base = slot_a + slot_b;
        }
        lds_barrier();
    }
    // (Leaving the loop as soon as no window of the tile grew any more -- all frozen, wrapped round their cycle, or run into a
    // dead end -- was tried as well: most tiles of a clean frame hold a chain that needs all eleven rounds.)
    // results
    if (n_rounds <= 0) {   // (the probe's "no rounds" form)
        if (frame_entries) { if (threadIdx.x < kFrameWin) s_fbase[threadIdx.x] = slot_a + slot_b; }
        else if (threadIdx.x == 0) s_new_base = slot_a + slot_b;
        lds_barrier();
    }
    // Dead cycles (kDead): a leader whose window wrapped inside the tile knows its border's length from its successor's window
    // (hops back to the leader + 1), here in LDS; too short for k_cycle_select's test and starting unconditionally, the border is
    // finished with: counted as traced, listed nowhere.  Noise-like frames: most of the millions of borders.
    if (dead_count) {
        // The leaders in question are a few per wave and dart slot: they are QUEUED (tile index, successor's index, pixel info) and
        // worked on with every lane busy -- evaluated in place, each of a lane's eight slots cost the wave a chain of LDS round trips
        // for the two or three lanes that had a candidate there (+ 107 us on the reference's bench input with the witness test).
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t i = threadIdx.x + u * 256;
            const uint32_t sl = succ0[u] - lo;
            const bool push = i < cnt && ((sfire | ecand) & (1u << u)) && (uint32_t)nk[u] == lo + i && (np[u] - lo) < cnt   // the leader, its window in the tile
                              && sl < cnt && sl != i;                                                                          // (a chain end: the normal path reports it)
            const unsigned long long m = __ballot(push);
            if (m == 0ull) continue;   // wave-uniform
            uint32_t base = 0;
            if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&s_qn, (uint32_t)__popcll(m));   // one LDS atomic per wave and slot
            base = __shfl(base, __ffsll((long long)m) - 1);
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
            if (push && slot < kQCap) s_q[slot] = i | (sl << 11) | (((epix >> (4 * u)) & 15u) << 22) | (((sfire >> u) & 1u) << 26);
        }
        lds_barrier();
        const uint32_t nq = min(s_qn, kQCap);
        uint32_t my_dead = 0;
        for (uint32_t k = threadIdx.x; k < nq; k += 256) {
            const uint32_t e = s_q[k], i = e & 2047u, sl = (e >> 11) & 2047u;
            const Win wi = s_win[i], ws = s_win[sl];
            if ((uint32_t)ws.key != lo + i) continue;                                  // the successor's window did not wrap to this leader
            const uint32_t n = (ws.offdist & 0xFFFFu) + 1u;                            // border length in points
            if (n >= 5u && (uint64_t)n * n >= 8ull * min_edge_length) continue;        // k_cycle_select's parity-safe test (the diagonal bound cannot bind in a tile)
            if (!((e >> 26) & 1u)) {
                // Not static: the pixel's other darts -- the leader's neighbours in the dart order -- decide (k_resolve_eval: the W-event
                // of pixel q fires iff every border through q starts at or after it; its E-event fires iff not (q has a W side and the
                // W-event fires)).  A window that did not freeze holds its border's smallest event and its leader.
                //  * CERTAIN, whatever the other borders' starts turn out to be: an E-event with a WITNESS -- a border through q whose
                //    smallest event lies before 2q and passes static_fire (bit 1 of s_dead: the border starts there under every
                //    assignment): the W-event cannot fire, the E-event does.  (tests/dart_model.py: 64 % of a noise frame's borders start
                //    certainly by static_fire alone, 91 % with witnesses.)
                //  * UNDER THE NATURAL ASSIGNMENT (every border starts at its smallest event), when the launch sequence relies on it
                //    (trust_natural: k_cycle_select checks the listed borders the same way, and if ANY border's natural start does not
                //    fire the batch is re-run with the fixpoint passes and without this rule): all of the pixel's darts in this tile with
                //    closed windows, and the event fires given their smallest events.  If every border passes -- here or there -- the
                //    natural assignment is the fixpoint (k_resolve_fast's argument) and a short border that was finished with here was
                //    rightly counted as traced.  One that does not pass stays listed, and k_cycle_select raises the re-run.
                const uint32_t pi = (e >> 22) & 15u, rank = pi & 3u, others = pi >> 2, first = i - rank;
                if (rank > i || first + others >= cnt) continue;                       // the pixel's darts straddle the tile
                const uint32_t ek = (uint32_t)(wi.key >> 32), w2q = ek & ~1u;          // the leader's key: 2q (W-event) or 2q + 1 (E-event)
                Win wj[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) wj[j] = s_win[first + min(j, others)];
                bool witness = false, closed = true, wfires = true;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t lj = (uint32_t)wj[j].key - lo;
                    const bool in_tile = (wj[j].ptr - lo) < cnt, before = (uint32_t)(wj[j].key >> 32) < w2q;
                    if (in_tile && before && lj < cnt && (s_dead[lj] & 2u)) witness = true;
                    closed = closed && in_tile;
                    wfires = wfires && !before;
                }
                const bool is_e = (ek & 1u) != 0u;
                if (!((is_e && witness) || (trust_natural && closed && (is_e ? !wfires : wfires)))) continue;
            }
            s_dead[i] |= 1; my_dead++;
        }
        if (my_dead) atomicAdd(&s_dead_n, my_dead);
        lds_barrier();
        if (threadIdx.x == 0 && s_dead_n) atomicAdd(&dead_count[blockIdx.x & 15u], s_dead_n);
    }
    // Entries get their slots; the slot of dart i's successor is left in LDS under i (the windows are not needed any more), where
    // the windows that froze on i find it: a frozen window's state carries the SLOT of the entry it ends on.  (Until round 5 it
    // carried the entry's dart index and every reader went through an entry_pos[] array: a scattered 4-byte read per frozen dart in
    // k_jump_finalize -- three dependent loads where two do -- and an array of 4 bytes per dart of the pool.)
    // (behind the last round's barrier -- or the epilogue's -- s_fbase / s_new_base are in place and nobody reads s_win any more)
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(s_win);
#pragma unroll
    for (int u = 0; u < PER; u++)
        if (my_slot[u] != kNone) {
            const uint32_t slot = (direct & (1u << u)) ? my_slot[u] : (frame_entries ? s_fbase[frm[u] - f0] : s_new_base) + my_slot[u];
            s_slot[threadIdx.x + u * 256] = slot;
            entry_list[slot] = succ0[u];
        }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t i = threadIdx.x + u * 256;
        if (i >= cnt) continue;
        const uint32_t e = np[u];
        const bool frozen = (e - lo) >= cnt;
        const uint32_t li = (uint32_t)nk[u] - lo;                                      // my window's minimum: in the tile when it did not freeze
        const bool dead = dead_count != nullptr && !frozen && li < cnt && (s_dead[li] & 1u) != 0;
        JumpState r;
        r.key = nk[u]; r.ptr = frozen ? s_slot[(e - lo - (uint32_t)LT) & (uint32_t)(LT - 1)] : e; r.off = loc_pack(no[u], nd[u], frozen) | (dead ? kDead : 0u);
        loc[lo + i] = r;
    }
}

struct __attribute__((aligned(8))) EntryState { uint64_t key; uint32_t ptr; uint32_t off; uint32_t dist; uint32_t pad; };

// Phase 2, one workgroup per frame, all rounds in LDS: the entries of a clean frame number a few hundred, and a border never
// leaves its frame, so the reduced list of a frame closes on itself.  Replaces k_entry_init + ~8 k_entry_jump launches (each
// is mostly launch latency).  A frame with more than kEntryLdsCap entries raises ctr->entry_overflow: the batch is re-run
// with the global rounds below (noise-like frames).
constexpr uint32_t kEntryLdsCap = 2048;
__global__ __launch_bounds__(256) void k_entry_frame(const uint32_t* __restrict__ entry_list, const uint32_t* __restrict__ frame_entries,
                                                     const uint32_t* __restrict__ frame_base, const JumpState* __restrict__ loc,
                                                     EntryState* __restrict__ es, DeviceCounters* __restrict__ ctr) {
    // one 16-byte record per entry (a single ds_read_b128 fetches the window an entry is joined with) + its length
    struct __attribute__((aligned(16))) Rec { uint64_t key; uint32_t ptr; uint32_t off; };
    __shared__ Rec s_rec[kEntryLdsCap];
    __shared__ uint32_t s_dist[kEntryLdsCap];
    const uint32_t f = blockIdx.x;
    const uint32_t cnt = frame_entries[f], base = frame_base[f];
    if (cnt == 0) return;
    if (cnt > kEntryLdsCap) { if (threadIdx.x == 0) ctr->entry_overflow = 1u; return; }
    // every lane owns up to 8 entries (i = lane + 256 u) and keeps their windows in registers across the rounds; LDS holds the
    // copy the other lanes read
    constexpr int PER = kEntryLdsCap / 256;
    uint64_t nk[PER]; uint32_t np[PER], no[PER], nd[PER];
    // the two dependent loads (entry -> its local state, which names the slot of the entry it froze at) for four entries per lane at
    // a time: a clean frame has several hundred entries, i.e. one trip of this loop
    constexpr int EB = 4;
    static_assert(PER % EB == 0, "whole batches");
#pragma unroll
    for (int u0 = 0; u0 < PER; u0 += EB) {
        if ((uint32_t)u0 * 256u >= cnt) break;   // uniform
        uint32_t e[EB], pos[EB];
        JumpState l[EB];
#pragma unroll
        for (int u = 0; u < EB; u++) e[u] = entry_list[base + min(threadIdx.x + 256u * (uint32_t)(u0 + u), cnt - 1u)];
#pragma unroll
        for (int u = 0; u < EB; u++) l[u] = loc[e[u]];
#pragma unroll
        for (int u = 0; u < EB; u++) pos[u] = l[u].ptr;   // (some dart, and ignored, when the window did not freeze)
        // (the values are "used" here so that the compiler cannot sink the later entries' loads into the `i < cnt` test below,
        // which would turn overlapped chains of round trips into chains in a row)
#pragma unroll
        for (int u = 0; u < EB; u++) asm volatile("" : "+v"(pos[u]), "+v"(l[u].key), "+v"(l[u].off));
#pragma unroll
        for (int u = 0; u < EB; u++) {
            const uint32_t i = threadIdx.x + 256u * (uint32_t)(u0 + u);
            nk[u0 + u] = l[u].key; no[u0 + u] = loc_off(l[u].off); nd[u0 + u] = loc_dist(l[u].off);
            // an entry's local window always freezes (its predecessor lies in another tile) unless its chain dead-ends in the tile
            np[u0 + u] = (l[u].off & kFrozen) ? pos[u] - base : i;
            if (i < cnt) { s_rec[i] = Rec{nk[u0 + u], np[u0 + u], no[u0 + u]}; s_dist[i] = nd[u0 + u]; }
        }
    }
    __syncthreads();
    for (int round = 0; round < 12; round++) {   // 2^11 = kEntryLdsCap hops, + the round that sees nothing move
        int changed = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t i = threadIdx.x + u * 256;
            if (i < cnt) {
                const uint32_t t = np[u] < cnt ? np[u] : i;   // (a corrupt pointer cannot leave the frame's slots)
                const Rec w = s_rec[t];
                const uint32_t wd = s_dist[t];
                if (w.key < nk[u]) { nk[u] = w.key; no[u] = nd[u] + w.off; changed = 1; }
                nd[u] += wd;
                np[u] = w.ptr;
            }
        }
        const int any = __syncthreads_or(changed);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t i = threadIdx.x + u * 256;
            if (i < cnt) { s_rec[i] = Rec{nk[u], np[u], no[u]}; s_dist[i] = nd[u]; }
        }
        __syncthreads();
        if (!any) break;   // no key moved: every window wraps its cycle
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t i = threadIdx.x + u * 256;
        if (i < cnt) {
            EntryState o;
            o.key = nk[u]; o.ptr = base + np[u]; o.off = no[u]; o.dist = nd[u]; o.pad = 0;
            es[base + i] = o;
        }
    }
}

// Phase 2 set-up: the reduced list over entries.  An entry's local window always freezes (its predecessor lies in another
// tile, so it cannot sit on a tile-local cycle) unless its chain dead-ends inside the tile; then it points at itself.
__global__ void k_entry_init(const uint32_t* __restrict__ entry_list, const unsigned int* __restrict__ entry_count,
                             const JumpState* __restrict__ loc, EntryState* __restrict__ es, uint32_t cap) {
    const EntrySpace sp(entry_count);
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < sp.total; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = sp.slot(i0, cap);
        if (i == kNone) continue;
        const uint32_t e = entry_list[i];
        const JumpState l = loc[e];
        EntryState s;
        s.key = l.key; s.off = loc_off(l.off); s.dist = loc_dist(l.off); s.pad = 0;
        s.ptr = (l.off & kFrozen) ? l.ptr : i;
        es[i] = s;
    }
}

// Phase 2: doubling over entries (hop counts double per round; `dist` carries the real path length)
__global__ __launch_bounds__(256) void k_entry_jump(const EntryState* __restrict__ in, EntryState* __restrict__ out,
                                                    const unsigned int* __restrict__ entry_count, uint32_t cap, int round,
                                                    DeviceCounters* __restrict__ ctr) {
    // no key moved in the previous round => every window already wraps its cycle; both buffers hold final key/off
    if (round > 0 && ctr->jump_changed[round - 1] == 0) return;
    const EntrySpace sp(entry_count);
    uint32_t changed = 0;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < sp.total; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = sp.slot(i0, cap);
        if (i == kNone) continue;
        EntryState s = in[i];
        // An entry whose window met its own minimum again in the window it was joined with has wrapped its cycle (keys are unique): key
        // and offset are final.  It is written once more (pad 1 -> 2: both ping-pong buffers then hold the final state) and takes
        // no further part -- no scattered read of its partner, no write.  Noise-like frames: most cycles close within a few rounds,
        // and the scattered 24-byte partner reads (a 128-byte line each) are what a round costs.
        if (s.pad >= 2u) continue;
        if (s.pad == 1u) { s.pad = 2u; out[i] = s; const_cast<EntryState*>(in)[i].pad = 2u; continue; }   // (nobody reads a partner's pad)
        const EntryState t = in[s.ptr];
        if (t.key == s.key) s.pad = 1u;
        if (t.key < s.key) { s.key = t.key; s.off = s.dist + t.off; changed++; }
        s.dist += t.dist;
        s.ptr = t.ptr;
        out[i] = s;
    }
    // a flag, not a count: same-address atomics serialise (~11 ns each), plain stores of the same value do not
    if (__ballot(changed != 0) && (threadIdx.x & 63) == 0) ctr->jump_changed[round] = 1u;
}

// Phase 3: every dart learns its cycle's leader and its hop distance to it
constexpr uint32_t kLeaderShards = 16;
// darts handled by the blocks of one shard: n/16 plus at most one 256-dart slice per block of the shard and iteration
__host__ __device__ inline uint32_t leader_shard_cap(uint32_t n_darts) { return n_darts / kLeaderShards + n_darts / 64u + 262144u; }

// ... and the leaders of cycles that carry at least one start event are collected (one atomic per wave) for the
// per-border kernels that follow.
__global__ __launch_bounds__(256) void k_jump_finalize(uint32_t n_darts, const JumpState* __restrict__ loc,
                                                       const EntryState* __restrict__ es,
                                                       FinState* __restrict__ fin,
                                                       uint32_t* __restrict__ leader_list,
                                                       unsigned int* __restrict__ leader_count /*[kLeaderShards]*/, uint32_t shard_cap,
                                                       const uint32_t* __restrict__ n_live, const DeviceCounters* __restrict__ ctr) {
    // a frame's entries did not fit k_entry_frame's LDS: their states were never written and the batch is re-run; with no
    // leaders listed and no points scattered everything downstream is a no-op
    if (ctr->entry_overflow) return;
    if (n_live) n_darts = min(n_darts, *n_live);
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint32_t shard = blockIdx.x & (kLeaderShards - 1);   // spread the slot counter over 16 addresses
    const uint32_t stride = gridDim.x * blockDim.x;            // the launcher keeps ceil(n_darts / stride) <= 32
    uint32_t mask = 0;                                          // bit i: my i-th dart leads a cycle that has a start event
    // B darts per lane at a time, each of the two dependent loads (local state, which names the slot of the entry the window froze
    // at -> that entry's state) issued for all of them before the first is used: the kernel is a chain of round trips to memory.
    // Loads are unconditional from clamped indices (slot 0 for windows that did not freeze: a cached line); behind an `if` the
    // compiler would issue them one at a time again.  (Round 6: the slot comes with the state; rounds 2-5 fetched it from an
    // entry_pos[] array in between, a third dependent, scattered load.)
    constexpr int B = A3_FIN_B;
    int it = 0;
    for (uint32_t d0 = blockIdx.x * blockDim.x + threadIdx.x; d0 < n_darts; d0 += B * stride, it += B) {
        JumpState s[B];
        uint32_t od[B], pos[B];
        EntryState g[B];
#pragma unroll
        for (int u = 0; u < B; u++) s[u] = loc[min(d0 + (uint32_t)u * stride, n_darts - 1u)];
#pragma unroll
        for (int u = 0; u < B; u++) { od[u] = s[u].off; pos[u] = (od[u] & kFrozen) ? s[u].ptr : 0u; }
#pragma unroll
        for (int u = 0; u < B; u++) g[u] = es[pos[u]];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const uint32_t d = d0 + (uint32_t)u * stride;
            if (d >= n_darts) break;
            // a window that was final inside its tile -- wrapped, or frozen at an entry whose own cycle minimum is not smaller --
            // keeps its local answer; the others take the entry's
            const bool better = (od[u] & kFrozen) && g[u].key < s[u].key;
            if (better) { s[u].key = g[u].key; s[u].off = (loc_dist(od[u]) + g[u].off) | kFinal; }
            const bool event = (uint32_t)(s[u].key >> 32) != kNoKey, dead = is_dead(s[u].off);
            fin[d] = FinState{(uint32_t)s[u].key, fin_off(s[u].off) | (event ? kFinEvent : 0u) | (dead ? kFinDead : 0u)};
            if ((uint32_t)s[u].key == d && event && !dead) mask |= 1u << (it + u);
        }
    }
    // one global atomic per workgroup: leaders are counted in a block scan first
    uint32_t total;
    const uint32_t excl = block_excl_scan_256((uint32_t)__popc(mask), s_wave, &total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&leader_count[shard], total) : 0u;
    __syncthreads();
    uint32_t slot = s_base + excl;
    while (mask) {
        const int i = __ffs(mask) - 1;
        mask &= mask - 1;
        leader_list[(size_t)shard * shard_cap + slot++] = blockIdx.x * blockDim.x + threadIdx.x + (uint32_t)i * stride;
    }
}

// ---------------------------------------------------------------------------------------
// start resolution
// ---------------------------------------------------------------------------------------
constexpr uint64_t kInf64 = ~0ull;

// Fast path, one lane per border that has a start event: under the natural assignment T0 (every border starts at its
// smallest event, which is its leader dart) evaluate only that smallest event.  If it fires for every border, the Jacobi
// step of k_resolve_eval maps T0 to itself (T'(c) = min firing event >= min event = T0(c)), so T0 is the fixpoint and the
// passes over all darts below are skipped; this is the case unless a component's first pixel lies in column 0.
// T0 is written for the listed leaders only -- the only slots k_cycle_select reads.
// does the smallest event of the border led by dart d (key0 = st[d].key) fire under the natural assignment?
__device__ __forceinline__ bool natural_start_fires(uint32_t d, uint64_t key0, const JumpState* __restrict__ st, const FinState* __restrict__ fin,
                                                    const uint64_t* __restrict__ d_rec, int W) {
    const uint64_t rec = d_rec[d];
    if (static_fire(rec)) return true;   // (the leader's key is its own event's: no look at the neighbours' states needed)
    const uint32_t info = rec_info(rec);
    const uint32_t xy = rec_xy(rec);
    const uint32_t x = xy & 0xFFFF, y = xy >> 16;
    const uint32_t q = y * (uint32_t)W + x;
    const uint32_t F = rec_F(rec), P = pdart_mask(F);
    const int k = info & 7;
    const uint32_t base = d - __popc(P & ((1u << k) - 1u));
    const int cnt = __popc(P);
    bool wfires = true;
    // a pixel owns at most four darts: their leaders, then the leaders' own keys (a leader's local state is final), each as one batch
    // of loads (clamped indices) rather than a chain of up to eight round trips with an early exit
    uint32_t lj[4];
    uint64_t lk[4];
#pragma unroll
    for (int j = 0; j < 4; j++) lj[j] = fin[base + (uint32_t)min(j, cnt - 1)].leader;
#pragma unroll
    for (int j = 0; j < 4; j++) lk[j] = st[lj[j]].key;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // T0 of the border through this dart: its leader's event key, provided the dart sits on an intact cycle with an event
        uint32_t t = (uint32_t)(lk[j] >> 32);
        if (lj[j] != d && (uint32_t)lk[j] != lj[j]) t = kNoKey;
        if (j < cnt && t < 2u * q) wfires = false;
    }
    const bool has_w = x > 0 && !(F & 1u);
    uint32_t key = kNoKey;
    if ((info & kInfoW) && wfires) key = 2u * q;
    else if ((info & kInfoE) && !(has_w && wfires)) key = 2u * q + 1u;
    return key == (uint32_t)(key0 >> 32);
}

// (Launched only when the full passes below are in the launch sequence; otherwise k_cycle_select does this check itself.)
__global__ __launch_bounds__(256) void k_resolve_fast(const JumpState* __restrict__ st, const FinState* __restrict__ fin, const uint32_t* __restrict__ leader_list,
                                                      const unsigned int* __restrict__ leader_count, uint32_t shard_cap, int W,
                                                      const uint64_t* __restrict__ d_rec, uint64_t* __restrict__ t_cur,
                                                      DeviceCounters* __restrict__ ctr) {
    uint32_t n_max = 0;
    for (uint32_t sh = 0; sh < kLeaderShards; sh++) n_max = max(n_max, leader_count[sh]);
    const uint32_t span = (n_max + 63u) & ~63u;
    const uint32_t n_leaders = span * kLeaderShards;
    bool moved = false;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n_leaders; i0 += gridDim.x * blockDim.x) {
        const uint32_t sh = i0 / span, i = i0 - sh * span;
        if (i >= leader_count[sh]) continue;
        const uint32_t d = leader_list[(size_t)sh * shard_cap + i];
        const uint64_t key0 = st[d].key;            // (smallest event key << 32) | d
        t_cur[d] = key0;
        if (!natural_start_fires(d, key0, st, fin, d_rec, W)) moved = true;
    }
    if (moved) ctr->resolve_needed = 1u;
}

// leaders get their natural start (their own key: the smallest event on the cycle); every other slot is "never"
__global__ void k_resolve_init(const JumpState* __restrict__ st, const FinState* __restrict__ fin, uint32_t n_darts, uint64_t* __restrict__ t_cur,
                               uint64_t* __restrict__ t_next, const DeviceCounters* __restrict__ ctr, const uint32_t* __restrict__ n_live) {
    if (!ctr->resolve_needed) return;
    if (n_live) n_darts = min(n_darts, *n_live);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_darts; d += gridDim.x * blockDim.x) {
        const FinState fs = fin[d];
        const bool natural = fs.leader == d && (fs.w & kFinEvent) != 0u;
        t_cur[d] = natural ? st[d].key : kInf64;   // (a leader's local key is its own: final)
        t_next[d] = kInf64;
    }
}

// Evaluate every start event under the current assignment T (Jacobi step):
//   Wfires(q)  = every cycle through q has T >= 2q          (reference: label(q) == 1 when the scan arrives)
//   W-event(q) fires iff Wfires(q); E-event(q) fires iff not (hasW(q) and Wfires(q))
//                                                           (reference: the `else if`, label(q) > 0)
// and propose T'(cycle) = min key of its firing events.
__global__ __launch_bounds__(256) void k_resolve_eval(const FinState* __restrict__ fin, uint32_t n_darts, int W,
                                                      const uint64_t* __restrict__ d_rec, const uint64_t* __restrict__ t_cur,
                                                      uint64_t* __restrict__ t_next, int iter, DeviceCounters* __restrict__ ctr,
                                                      const uint32_t* __restrict__ n_live) {
    if (!ctr->resolve_needed || (iter > 0 && ctr->resolve_changed[iter - 1] == 0)) return;
    if (n_live) n_darts = min(n_darts, *n_live);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_darts; d += gridDim.x * blockDim.x) {
        const uint64_t rec = d_rec[d];
        const uint32_t info = rec_info(rec);
        if (!(info & (kInfoW | kInfoE))) continue;
        // an event dart must sit on an intact cycle, whose leader is its own leader (open chains never carry events)
        const uint32_t my_leader = fin[d].leader;
        if (fin[my_leader].leader != my_leader) { atomicOr(&ctr->err_flags, kErrBrokenEvent); continue; }
        const uint32_t xy = rec_xy(rec);
        const uint32_t x = xy & 0xFFFF, y = xy >> 16;
        const uint32_t q = y * (uint32_t)W + x;
        const uint32_t F = rec_F(rec), P = pdart_mask(F);
        const int k = info & 7;
        const uint32_t base = d - __popc(P & ((1u << k) - 1u));
        const int cnt = __popc(P);
        bool wfires = true;
        for (int i = 0; i < cnt; i++) {
            const uint32_t leader = fin[base + i].leader;
            const uint32_t t = (uint32_t)(t_cur[leader] >> 32);
            if (t < 2u * q) { wfires = false; break; }
        }
        const bool has_w = x > 0 && !(F & 1u);
        uint32_t key = kNoKey;
        if ((info & kInfoW) && wfires) key = 2u * q;
        else if ((info & kInfoE) && !(has_w && wfires)) key = 2u * q + 1u;
        if (key != kNoKey) atomicMin(reinterpret_cast<unsigned long long*>(&t_next[my_leader]),
                                     (unsigned long long)(((uint64_t)key << 32) | d));
    }
}

// adopt T' as T, count the cycles whose start moved (into this pass's slot), clear T' for the next pass
__global__ void k_resolve_commit(const FinState* __restrict__ fin, uint32_t n_darts, uint64_t* __restrict__ t_cur,
                                 uint64_t* __restrict__ t_next, int iter, int last, DeviceCounters* __restrict__ ctr,
                                 const uint32_t* __restrict__ n_live) {
    if (!ctr->resolve_needed || (iter > 0 && ctr->resolve_changed[iter - 1] == 0)) return;
    if (n_live) n_darts = min(n_darts, *n_live);
    uint32_t changed = 0;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_darts; d += gridDim.x * blockDim.x) {
        if (fin[d].leader != d) continue;
        const uint64_t a = t_cur[d], b = t_next[d];
        if (a != b) { changed++; t_cur[d] = b; }
        t_next[d] = kInf64;
    }
    for (int o = 32; o > 0; o >>= 1) changed += __shfl_down(changed, o);
    if ((threadIdx.x & 63) == 0 && changed) {
        atomicAdd(&ctr->resolve_changed[iter], changed);
        if (last) atomicOr(&ctr->err_flags, kErrResolve);  // still moving after the last pass we are willing to run
    }
}

// ---------------------------------------------------------------------------------------
// select the borders worth materialising, then write their points in traversal order
// ---------------------------------------------------------------------------------------
// The border slot of a listed leader (kFinHops: not materialised) is written into the hops field of the leader's OWN FinState -- a
// leader is 0 hops from itself, and every reader of that field special-cases the leader -- so that k_scatter_points finds "does my
// leader lead itself" and "where do its points go" in one 8-byte load per dart instead of two scattered ones (on noise-like frames,
// 50 M darts, the scattered loads are the kernel).
__global__ __launch_bounds__(256) void k_cycle_select(const JumpState* __restrict__ st, FinState* fin, const uint32_t* __restrict__ leader_list,
                                                      const unsigned int* __restrict__ leader_count, const uint32_t* __restrict__ d_succ,
                                                      const uint64_t* __restrict__ t_cur, const uint32_t* __restrict__ frame_base,
                                                      uint32_t n_frames, uint32_t first_frame, uint32_t min_edge_length, double eps_factor,
                                                      double image_diag, ContourRec* __restrict__ contours,
                                                      uint32_t* __restrict__ cyc_start_off, uint32_t max_contours, uint64_t max_points,
                                                      DeviceCounters* __restrict__ ctr, uint32_t shard_cap, const uint64_t* __restrict__ d_rec,
                                                      int W /* > 0: no resolve kernel ran; borders start naturally, checked here */,
                                                      uint32_t* __restrict__ keep_tmp /* one word per leader-list slot */,
                                                      int keep_all /* debug taps: materialise every traced border (a3_download_contours) */) {
    __shared__ uint32_t s_wave[4], s_wave_t[4];
    __shared__ unsigned long long s_wave_p[4];
    __shared__ uint32_t s_cbase;
    __shared__ unsigned long long s_pbase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the leader list is 16 segments of shard_cap slots; segment s holds leader_count[s] entries
    uint32_t n_max = 0;
    for (uint32_t sh = 0; sh < kLeaderShards; sh++) n_max = max(n_max, leader_count[sh]);
    const uint32_t span = (n_max + 63u) & ~63u;           // per-shard index space
    const uint32_t n_leaders = span * kLeaderShards;
    const uint32_t stride = gridDim.x * blockDim.x;

    // one leader: is its border traced, how long is it, is it worth materialising
    struct Eval { bool valid, traced, keep, broken; uint32_t n, d; uint64_t t; };
    auto eval = [&](uint32_t i0) -> Eval {
        Eval e{false, false, false, false, 0u, 0u, kInf64};
        const uint32_t sh = i0 / span, i = i0 - sh * span;
        e.valid = i < leader_count[sh];
        if (!e.valid) return e;
        e.d = leader_list[(size_t)sh * shard_cap + i];
        e.t = W > 0 ? st[e.d].key : t_cur[e.d];   // listed leaders carry a start event: their natural start is their key
        e.traced = e.t != kInf64;
        if (!e.traced) return e;
        // the successor's window must have wrapped around to this leader, else this is a chain, not a cycle
        const uint32_t sl = d_succ[e.d];
        const FinState fsl = fin[sl];
        if (sl == e.d || fsl.leader != e.d) { e.broken = true; return e; }
        e.n = (fsl.w & kFinHops) + 1u;
        // Parity-safe pruning (src/aruco.rs:133-158):
        //  (1) a candidate keeps 4 border points in convex position whose hull edges are all >= sqrt(min_edge_length) long
        //      (src/aruco.rs:149-159).  The closed border visits the four in some order; between two of them it needs at least
        //      their Chebyshev distance >= Euclidean distance / sqrt(2) steps (consecutive border points are 8-neighbours), and no
        //      closed tour through four points in convex position is shorter than their hull's perimeter.  So
        //      n >= 4 sqrt(min_edge_length) / sqrt(2), i.e. n^2 >= 8 min_edge_length is necessary (round 4 used the bound of ONE
        //      edge, n^2 >= 2 min_edge_length: on noise-like frames the tighter one materialises a third as many borders);
        //  (2) Douglas-Peucker splits only when a point is further than eps = eps_factor*n from a chord,
        //      and no two pixels are further apart than the image diagonal (+1 slack for rounding).
        const double eps = (double)e.n * eps_factor;
        e.keep = keep_all || (e.n >= 5u && (uint64_t)e.n * e.n >= 8ull * min_edge_length && eps < image_diag + 1.0);
        return e;
    };

    // Pass 1: what this workgroup will allocate.  Pass 2 re-evaluates (the loads hit the cache) and fills the slots.  The
    // two global counters are bumped once per workgroup: per-wave bumps of one address serialise at ~11 ns each, which on
    // noise frames (140 k borders per frame) was 1.6 ms of a 1.75 ms kernel.
    uint32_t my_keep = 0, my_traced = 0;
    unsigned long long my_points = 0;
    bool broken = false, moved = false;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n_leaders; i0 += stride) {
        const Eval e = eval(i0);
        my_traced += e.traced; my_keep += e.keep; my_points += e.keep ? e.n : 0u;
        broken |= e.broken;
        // pass 2 only re-evaluates the borders that are kept (on noise frames one in a hundred); for the others it needs to
        // know just that, from a coalesced read instead of five scattered ones
        if (e.valid) { const uint32_t sh = i0 / span; keep_tmp[(size_t)sh * shard_cap + (i0 - sh * span)] = e.keep ? 1u : 0u; }
        // the k_resolve_fast test, folded in: if some border's smallest event does not fire the batch is re-run with the
        // fixpoint passes (what is selected below is then discarded)
        if (W > 0 && e.valid && !natural_start_fires(e.d, e.t, st, fin, d_rec, W)) moved = true;
    }
    if (moved) ctr->resolve_needed = 1u;
    if (broken) atomicOr(&ctr->err_flags, kErrBrokenEvent);
    // exclusive scans of (keep count, point count) over the workgroup
    uint32_t inc_k = my_keep, inc_t = my_traced;
    unsigned long long inc_p = my_points;
    // inclusive scans over the wave by DPP row shifts / broadcasts (for the 64-bit one both halves of the source lane's value
    // are moved, then added as one number)
#define A3_DPP_ADD64(V, CTRL, RM)                                                                                       \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(V), CTRL, RM, 0xF, false);           \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((V) >> 32), CTRL, RM, 0xF, false);    \
        (V) += ((unsigned long long)hi_ << 32) | lo_;                                                                    \
    }
#define A3_DPP_ADD32(V, CTRL, RM) (V) += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(V), CTRL, RM, 0xF, false);
#define A3_STEP(CTRL, RM) A3_DPP_ADD32(inc_k, CTRL, RM) A3_DPP_ADD32(inc_t, CTRL, RM) A3_DPP_ADD64(inc_p, CTRL, RM)
    A3_STEP(0x111, 0xF) A3_STEP(0x112, 0xF) A3_STEP(0x114, 0xF) A3_STEP(0x118, 0xF) A3_STEP(0x142, 0xA) A3_STEP(0x143, 0xC)
#undef A3_STEP
#undef A3_DPP_ADD32
#undef A3_DPP_ADD64
    if (lane == 63) { s_wave[wave] = inc_k; s_wave_p[wave] = inc_p; s_wave_t[wave] = inc_t; }
    __syncthreads();
    uint32_t base_k = 0, tot_k = 0;
    unsigned long long base_p = 0, tot_p = 0;
    for (int w = 0; w < 4; w++) {
        if (w < wave) { base_k += s_wave[w]; base_p += s_wave_p[w]; }
        tot_k += s_wave[w]; tot_p += s_wave_p[w];
    }
    if (threadIdx.x == 0) {
        const uint32_t tot_t = s_wave_t[0] + s_wave_t[1] + s_wave_t[2] + s_wave_t[3];
        if (tot_t) atomicAdd(&ctr->traced, tot_t);
        s_cbase = tot_k ? atomicAdd(&ctr->contours, tot_k) : 0u;
        s_pbase = tot_k ? atomicAdd(&ctr->points, tot_p) : 0ull;
    }
    __syncthreads();
    uint32_t c = s_cbase + base_k + (inc_k - my_keep);
    unsigned long long pb = s_pbase + base_p + (inc_p - my_points);

    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n_leaders; i0 += stride) {
        {
            const uint32_t sh = i0 / span, i = i0 - sh * span;
            if (i >= leader_count[sh]) continue;
            if (!keep_tmp[(size_t)sh * shard_cap + i]) { fin[leader_list[(size_t)sh * shard_cap + i]].w = kFinEvent | kFinHops; continue; }
        }
        const Eval e = eval(i0);
        if (!e.valid) continue;
        uint32_t slot = kNone;
        if (e.keep) {
            if (c >= max_contours || c >= kFinHops) atomicOr(&ctr->err_flags, kErrContourTable);   // (a slot must fit the 30-bit field)
            else if (pb + e.n > max_points) atomicOr(&ctr->err_flags, kErrPointPool);
            else {
                uint32_t lo = 0, hi = n_frames;  // frame of this dart: binary search in frame_base
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (frame_base[mid] <= e.d) lo = mid; else hi = mid; }
                ContourRec r;
                r.frame = first_frame + lo;
                r.start_key = (uint32_t)(e.t >> 32);
                r.point_base = (uint32_t)pb;
                r.n = e.n;
                contours[c] = r;
                cyc_start_off[c] = (uint32_t)e.t == e.d ? 0u : fin[(uint32_t)e.t].w & kFinHops;   // (the leader's own hops field is about to hold its slot)
                slot = c;
            }
            c++; pb += e.n;
        }
        fin[e.d].w = kFinEvent | (slot == kNone ? kFinHops : slot);   // (listed leaders carry an event and are not dead)
    }
}

__global__ __launch_bounds__(256) void k_scatter_points(const FinState* __restrict__ fin, uint32_t n_darts, const uint64_t* __restrict__ d_rec,
                                                        const ContourRec* __restrict__ contours,
                                                        const uint32_t* __restrict__ cyc_start_off, uint32_t* __restrict__ points,
                                                        const uint32_t* __restrict__ n_live, const DeviceCounters* __restrict__ ctr) {
    if (ctr->entry_overflow) return;
    if (n_live) n_darts = min(n_darts, *n_live);
    // B darts per lane at a time, three rounds of loads instead of five per dart: {state, record} -> {leader's key, border
    // slot of the leader} -> {border record, start offset}.  Unconditional loads from clamped indices, see k_jump_finalize.
    constexpr int B = A3_SCAT_B;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t d0 = blockIdx.x * blockDim.x + threadIdx.x; d0 < n_darts; d0 += B * stride) {
        FinState s[B];
        uint64_t rec[B];
        uint32_t c[B], so[B];
        FinState ls[B];
        ContourRec r[B];
        bool live[B];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const uint32_t d = min(d0 + (uint32_t)u * stride, n_darts - 1u);
            s[u] = fin[d];
        }
#pragma unroll
        for (int u = 0; u < B; u++) {
            live[u] = d0 + (uint32_t)u * stride < n_darts && (s[u].w & kFinEvent) != 0u &&   // else: no start event on this cycle
                      (s[u].w & kFinDead) == 0u;                                              // ... or a dead one (k_local_contract)
            const uint32_t leader = live[u] ? s[u].leader : 0u;
            ls[u] = fin[leader];   // the leader's own final state -- leader: does it lead itself; hops field: its border slot (k_cycle_select)
            c[u] = ls[u].w & kFinHops;
            // the dart's record travels with the second round of loads, and only for darts that may be written out (on noise-like
            // frames nine darts in ten are not: a third of the kernel's bytes)
            rec[u] = live[u] ? d_rec[min(d0 + (uint32_t)u * stride, n_darts - 1u)] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < B; u++) {
            // a leader that does not hold its own key: an open chain, or states of a run that has not converged (the batch is
            // then re-run) -- its slot was never written this batch
            live[u] = live[u] && ls[u].leader == s[u].leader && c[u] != kFinHops;
            r[u] = contours[live[u] ? c[u] : 0u];
            so[u] = cyc_start_off[live[u] ? c[u] : 0u];
        }
#pragma unroll
        for (int u = 0; u < B; u++) {
            if (!live[u]) continue;
            // off = hops forward to the leader; position along the border counted from the start dart
            const uint32_t off = s[u].leader == d0 + (uint32_t)u * stride ? 0u : s[u].w & kFinHops;   // (the leader's own hops field holds its slot)
            const uint32_t rank = so[u] >= off ? so[u] - off : so[u] + r[u].n - off;
            points[r[u].point_base + rank] = rec_xy(rec[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// K4: Douglas-Peucker + hull + winding + edge test, one wave per border
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int orient(int px, int py, int qx, int qy, int rx, int ry) {
    const long long v = (long long)(qy - py) * (rx - qx) - (long long)(qx - px) * (ry - qy);
    return v == 0 ? 0 : (v > 0 ? 1 : -1);  // 0 collinear, 1 clockwise, -1 counter-clockwise (imageproc naming)
}

// imageproc::geometry::convex_hull restricted to what the caller needs: the 4 input points in hull order
// if all four are hull vertices, otherwise false.  Mirrors the Graham scan (start = top-most then left-most,
// angular insertion sort, pop while the turn is not counter-clockwise).
__device__ bool hull4(const int* __restrict__ in /*8*/, int* __restrict__ out /*8*/) {
    int sp = 0;
    for (int i = 1; i < 4; i++)
        if (in[2 * i + 1] < in[2 * sp + 1] || (in[2 * i + 1] == in[2 * sp + 1] && in[2 * i] < in[2 * sp])) sp = i;
    const int sx = in[2 * sp], sy = in[2 * sp + 1];
    int qx[3], qy[3];
    {   // swap(0, sp); remove(0)
        int tx[4], ty[4];
        for (int i = 0; i < 4; i++) { tx[i] = in[2 * i]; ty[i] = in[2 * i + 1]; }
        tx[sp] = tx[0]; ty[sp] = ty[0];
        for (int i = 0; i < 3; i++) { qx[i] = tx[i + 1]; qy[i] = ty[i + 1]; }
    }
    for (int i = 1; i < 3; i++) {
        const int ax = qx[i], ay = qy[i];
        int j = i;
        while (j > 0) {
            const int bx = qx[j - 1], by = qy[j - 1];
            const int o = orient(sx, sy, ax, ay, bx, by);
            bool less;
            if (o == 0) {
                const long long da = (long long)(ax - sx) * (ax - sx) + (long long)(ay - sy) * (ay - sy);
                const long long db = (long long)(bx - sx) * (bx - sx) + (long long)(by - sy) * (by - sy);
                less = da < db;
            } else less = (o == -1);
            if (!less) break;
            qx[j] = bx; qy[j] = by; j--;
        }
        qx[j] = ax; qy[j] = ay;
    }
    int hx[5], hy[5], sn = 1;
    hx[0] = sx; hy[0] = sy;
    for (int i = 0; i < 3; i++) {
        while (sn > 1 && orient(hx[sn - 2], hy[sn - 2], hx[sn - 1], hy[sn - 1], qx[i], qy[i]) != -1) sn--;
        hx[sn] = qx[i]; hy[sn] = qy[i]; sn++;
    }
    if (sn != 4) return false;
    for (int i = 0; i < 4; i++) { out[2 * i] = hx[i]; out[2 * i + 1] = hy[i]; }
    return true;
}

// enforce_clockwise_corners for one quad, src/aruco.rs:168-185: (p1-p0) x (p2-p0) < 0 -> swap p1, p3
__device__ __forceinline__ void enforce_clockwise(int* __restrict__ hq /*8*/) {
    const int dx1 = hq[2] - hq[0], dy1 = hq[3] - hq[1], dx2 = hq[4] - hq[0], dy2 = hq[5] - hq[1];
    if (dx1 * dy2 - dy1 * dx2 < 0) { const int tx = hq[2], ty = hq[3]; hq[2] = hq[6]; hq[3] = hq[7]; hq[6] = tx; hq[7] = ty; }
}

// the winding fix of k_contour_quads on its own (reference vectors: src/aruco.rs:400-412)
__global__ void k_debug_clockwise(const int32_t* __restrict__ in, uint32_t n, int32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int q[8];
    for (int k = 0; k < 8; k++) q[k] = in[8 * i + k];
    enforce_clockwise(q);
    for (int k = 0; k < 8; k++) out[8 * i + k] = q[k];
}

// Largest value over a group of G = 64 or 16 lanes, returned in every lane of the group.  DPP row shifts (and, for the wave,
// row broadcasts) run in the VALU; a butterfly of __shfl_xor is two ds_bpermute round trips through the LDS crossbar per
// step for a 64-bit value, and Douglas-Peucker waits for one such reduction per chord.
#define A3_DPP_MAX64(V, CTRL, ROWMASK)                                                                                   \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(V), CTRL, ROWMASK, 0xF, false);        \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((V) >> 32), CTRL, ROWMASK, 0xF, false); \
        const unsigned long long o_ = ((unsigned long long)hi_ << 32) | lo_;                                             \
        (V) = o_ > (V) ? o_ : (V);                                                                                       \
    }
template <int G>
__device__ __forceinline__ unsigned long long group_max_u64(unsigned long long v) {
    A3_DPP_MAX64(v, 0x111, 0xF)   // row_shr:1 .. 8: lane 15 of every row of 16 holds the row's maximum (lanes without a source keep 0)
    A3_DPP_MAX64(v, 0x112, 0xF)
    A3_DPP_MAX64(v, 0x114, 0xF)
    A3_DPP_MAX64(v, 0x118, 0xF)
    if constexpr (G == 16) return __shfl(v, 15, 16);
    A3_DPP_MAX64(v, 0x142, 0xA)   // row_bcast:15 into rows 1 and 3
    A3_DPP_MAX64(v, 0x143, 0xC)   // row_bcast:31 into rows 2 and 3: lane 63 holds the maximum
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
#undef A3_DPP_MAX64

// G lanes work on one border: 64 for long ones, 16 (four borders per wave) for the short ones that noisy frames produce by
// the hundred thousand -- a full wave per 20-point border is latency with 60 idle lanes.  Control flow is uniform inside a
// group and the shuffles stay inside it.
constexpr uint32_t kSmallBorder = 64;
template <int G>
__device__ __forceinline__ void contour_quads_body(uint32_t block, uint32_t n_blocks, const ContourRec* __restrict__ contours, const DeviceCounters* __restrict__ ctr,
                                                       uint32_t max_contours, const uint32_t* __restrict__ points, double eps_factor,
                                                       uint32_t min_edge_length, uint32_t first_frame, uint32_t max_cand,
                                                       CandRec* __restrict__ cands, uint32_t* __restrict__ cand_count,
                                                       unsigned int* __restrict__ err_flags, bool coords14) {
    // coords14: every coordinate is below 2^14 (the image is at most 16384 x 16384): the distance numerators fit 32 bits
    const uint32_t n_contours = min(ctr->contours, max_contours);
    const int lane = threadIdx.x & (G - 1);
    const uint32_t wave_global = (block * blockDim.x + threadIdx.x) / G, n_waves = (n_blocks * blockDim.x) / G;
    for (uint32_t c = wave_global; c < n_contours; c += n_waves) {
        const ContourRec r = contours[c];
        if ((r.n <= kSmallBorder) != (G == 16)) continue;   // the other instantiation's share
        const uint32_t* P = points + r.point_base;
        const uint32_t n = r.n;
        const double eps = (double)n * eps_factor;  // c.points.len() as f64 * epsilon, src/aruco.rs:133
        // work list of chords (a,b); every split adds one kept point, exactly 3 splits are needed
        uint32_t seg_a[8], seg_b[8];
        int nseg = 1, splits = 0;
        uint32_t kept[3];
        seg_a[0] = 0; seg_b[0] = n - 1;
        bool reject = false;
        while (nseg > 0 && !reject) {
            nseg--;
            const uint32_t a = seg_a[nseg], b = seg_b[nseg];
            const uint32_t pa = P[a], pb = P[b];
            const int ax = pa & 0xFFFF, ay = pa >> 16, bx = pb & 0xFFFF, by = pb >> 16;
            // coordinates < 2^16: the line coefficients fit 17 bits + sign, |num| < 2^34, a border has < 2^30 points
            const int la = ay - by, lb = bx - ax;
            const long long lc = (long long)ax * by - (long long)bx * ay;
            unsigned long long best = 0;  // (|num| << 30) | ~index (30 bits): max picks largest num, then smallest index
            // four strided points per trip, their loads issued together: a long border is a chain of dependent round trips to
            // the point pool otherwise (the arg-max keeps the smallest index among equal distances whatever the visiting order)
            for (uint32_t i0 = a + 1 + lane; i0 <= b; i0 += 4 * G) {
                uint32_t p[4];
#pragma unroll
                for (int u = 0; u < 4; u++) p[u] = P[min(i0 + (uint32_t)u * G, b)];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + (uint32_t)u * G;
                    if (i > b) break;
                    unsigned long long anum;
                    if (coords14) {   // |la x|, |lb y|, |lc| < 2^28: two 24-bit multiply-adds instead of two 32 x 32 -> 64 ones
                        const int v = __mul24(la, (int)(p[u] & 0xFFFF)) + __mul24(lb, (int)(p[u] >> 16)) + (int)lc;
                        anum = (unsigned long long)(uint32_t)(v < 0 ? -v : v);
                    } else {
                        long long num = (long long)la * (int)(p[u] & 0xFFFF) + (long long)lb * (int)(p[u] >> 16) + lc;
                        anum = (unsigned long long)(num < 0 ? -num : num);
                    }
                    const unsigned long long cand = (anum << 30) | (unsigned long long)(~i & 0x3FFFFFFFu);
                    if (cand > best) best = cand;
                }
            }
            best = group_max_u64<G>(best);
            const unsigned long long num = best >> 30;
            if (num == 0) continue;
            const uint32_t index = ~(uint32_t)best & 0x3FFFFFFFu;
            // d = |a x + b y + c| / sqrt(a^2 + b^2) in f64, compared with `>` (imageproc approximate_polygon_dp).  The quotient
            // carries a relative error of a few 2^-53, so unless num^2 and eps^2 (a^2 + b^2) agree to nine digits the squared
            // comparison -- no square root, no division: those two are ~60 instructions for every chord, in every lane --
            // gives the same answer; in the band in between the reference's expression is evaluated as it stands.
            const double dn = (double)num, den = (double)((long long)la * la + (long long)lb * lb);
            const double lhs = dn * dn, rhs = eps * eps * den;
            bool far;
            if (lhs > rhs * (1.0 + 1e-9)) far = true;
            else if (lhs < rhs * (1.0 - 1e-9)) far = false;
            else far = dn / sqrt(den) > eps;
            if (far) {
                if (splits == 3) { reject = true; break; }
                kept[splits++] = index;
                seg_a[nseg] = a; seg_b[nseg] = index; nseg++;
                seg_a[nseg] = index; seg_b[nseg] = b; nseg++;
            }
        }
        if (reject || splits != 3) continue;  // edges.len() != 4 (after the closed pop)
        if (lane != 0) continue;
        // kept indices in increasing order, preceded by point 0 (the last point is popped: closed = true)
        uint32_t k0 = kept[0], k1 = kept[1], k2 = kept[2], t;
        if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
        if (k1 > k2) { t = k1; k1 = k2; k2 = t; }
        if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
        const uint32_t idx[4] = {0u, k0, k1, k2};
        int q[8], hq[8];
        for (int i = 0; i < 4; i++) { const uint32_t p = P[idx[i]]; q[2 * i] = p & 0xFFFF; q[2 * i + 1] = p >> 16; }
        if (!hull4(q, hq)) continue;  // convexity, src/aruco.rs:143-147
        uint32_t cmin = min_edge_length + 1u;  // src/aruco.rs:149-159 (squared length vs unsquared threshold, quirk Q1)
        for (int i = 0; i < 4; i++) {
            const int j = (i + 1) & 3;
            const int dx = hq[2 * i] - hq[2 * j], dy = hq[2 * i + 1] - hq[2 * j + 1];
            const uint32_t d2 = (uint32_t)(dx * dx + dy * dy);
            cmin = d2 < cmin ? d2 : cmin;
        }
        if (cmin < min_edge_length) continue;
        enforce_clockwise(hq);
        const uint32_t fl = r.frame - first_frame;
        const uint32_t slot = atomicAdd(&cand_count[fl], 1u);
        if (slot >= max_cand) { atomicOr(err_flags, kErrCandTable); continue; }
        CandRec cr;
        cr.start_key = r.start_key;
        for (int i = 0; i < 8; i++) cr.xy[i] = (uint16_t)hq[i];
        cands[(size_t)fl * max_cand + slot] = cr;
    }
}

// one launch for both group widths: the first `blocks64` workgroups take the long borders, the rest the short ones
__global__ __launch_bounds__(256) void k_contour_quads(uint32_t blocks64, const ContourRec* __restrict__ contours, const DeviceCounters* __restrict__ ctr,
                                                       uint32_t max_contours, const uint32_t* __restrict__ points, double eps_factor,
                                                       uint32_t min_edge_length, uint32_t first_frame, uint32_t max_cand,
                                                       CandRec* __restrict__ cands, uint32_t* __restrict__ cand_count,
                                                       unsigned int* __restrict__ err_flags, int coords14) {
    if (blockIdx.x < blocks64)
        contour_quads_body<64>(blockIdx.x, blocks64, contours, ctr, max_contours, points, eps_factor, min_edge_length, first_frame, max_cand, cands,
                               cand_count, err_flags, coords14 != 0);
    else
        contour_quads_body<16>(blockIdx.x - blocks64, gridDim.x - blocks64, contours, ctr, max_contours, points, eps_factor, min_edge_length,
                               first_frame, max_cand, cands, cand_count, err_flags, coords14 != 0);
}

// expand the packed thresholded image to 0/255 bytes (debug tap a3_download_thresholded)
__global__ void k_unpack_bits(const uint64_t* __restrict__ bits, int W, int H, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int wpr = (int)words_per_row((uint32_t)W);
    out[(size_t)y * W + x] = ((bits[(size_t)y * wpr + (x >> 6)] >> (x & 63)) & 1ull) ? 255 : 0;
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
// grid caps of the per-dart sweeps, overridable for tuning (tools/attic/sweep_grids.sh)
static inline int env_cap(const char* name, int dflt) { const int v = tuning_knob(name, dflt); return v > 0 ? v : dflt; }   // grid caps: constants unless -DA3_TUNING
static inline int blocks_for(uint64_t n, int per_block, int cap) {
    uint64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (uint64_t)cap) b = cap;
    return (int)b;
}

// zeroing as a kernel: a hipMemsetAsync between two kernels costs its own ~4 us plus a ~6 us switch of packet type
__global__ __launch_bounds__(256) void k_zero(uint4* __restrict__ p, uint32_t n16) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
hipError_t launch_zero(hipStream_t st, void* p, size_t bytes /* multiple of 16, p 16-byte aligned */) {
    const uint32_t n16 = (uint32_t)(bytes / 16);
    hipLaunchKernelGGL(k_zero, dim3(blocks_for(n16, 256, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n16);
    return hipGetLastError();
}

// tile_darts[frames * tiles] followed by tile_off[frames * tiles]
// tile_darts[frames * tiles] | tile_off[frames * tiles] | second-half counts of k_dart_count[frames * tiles]
size_t tile_mask_offset_bytes(uint32_t W, uint32_t H, uint32_t n_frames);
size_t tile_darts_bytes(uint32_t W, uint32_t H, uint32_t n_frames) { return tile_mask_offset_bytes(W, H, n_frames) + (size_t)dart_tiles(W, H) * n_frames * kCountHalves * 8; }
size_t tile_off_offset(uint32_t W, uint32_t H, uint32_t n_frames) { return (size_t)dart_tiles(W, H) * n_frames; }
// ... followed (8-byte aligned) by the word masks, kCountHalves 64-bit words per tile
size_t tile_mask_offset_bytes(uint32_t W, uint32_t H, uint32_t n_frames) { return ((size_t)dart_tiles(W, H) * n_frames * 4 * (1 + kCountHalves) + 7) & ~(size_t)7; }

// plan != nullptr: the launch of k_tile_scan also plans the batch on the device (frame_base[0..n_frames], plan[0..3])
hipError_t launch_dart_count(hipStream_t st, const uint64_t* bits, int W, int H, uint32_t first_frame, uint32_t n_frames,
                             unsigned long long* frame_darts, uint32_t* tile_darts, uint64_t plan_cap, uint32_t* frame_base, uint32_t* plan,
                             void* zero_p, size_t zero_bytes /* with plan: the counter block to zero, a multiple of 16 bytes */) {
    const uint32_t tiles_x = dart_tiles_x((uint32_t)W), tiles_y = ((uint32_t)H + kTileRows - 1) / kTileRows;
    const uint32_t wpr = words_per_row((uint32_t)W);
    uint32_t* h1 = tile_darts + (size_t)tiles_x * tiles_y * n_frames * 2;
    unsigned long long* tmask = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(tile_darts) + tile_mask_offset_bytes((uint32_t)W, (uint32_t)H, n_frames));
    if (wpr <= 30)   // two tile rows per wave
        hipLaunchKernelGGL(k_dart_count<32>, dim3((tiles_y + 1) / 2, n_frames, kCountHalves), dim3(64), 0, st, bits, W, H, first_frame, frame_darts, tile_darts, h1, tmask);
    else
        hipLaunchKernelGGL(k_dart_count<64>, dim3(((wpr + kCountLanes - 1) / kCountLanes) * tiles_y, n_frames, kCountHalves), dim3(64), 0, st, bits, W, H, first_frame,
                           frame_darts, tile_darts, h1, tmask);
    hipLaunchKernelGGL(k_tile_scan, dim3(n_frames + (plan ? 1u : 0u)), dim3(256), 0, st, tile_darts, h1, tiles_x * tiles_y, first_frame,
                       tile_darts + (size_t)tiles_x * tiles_y * n_frames, n_frames, frame_darts, (unsigned long long)plan_cap, frame_base, plan,
                       reinterpret_cast<uint4*>(zero_p), (uint32_t)(zero_bytes / 16));
    return hipGetLastError();
}

hipError_t launch_dart_build(hipStream_t st, const uint64_t* bits, int W, int H, uint32_t first_frame, uint32_t n_frames,
                             const uint32_t* frame_base, const uint32_t* tile_off, uint32_t* pix_base, const uint32_t* tile_darts, uint64_t* d_rec,
                             uint32_t* d_succ, uint32_t n_darts, const uint32_t* n_live, int dbg, const unsigned long long* tile_mask) {
    const uint32_t tiles = dart_tiles((uint32_t)W, (uint32_t)H);
    hipLaunchKernelGGL(k_dart_assign, dim3((tiles * n_frames + 7u) / 8u * 8u), dim3(256), 0, st, bits, W, H, first_frame, frame_base, tile_off,
                       pix_base, tile_darts, d_rec, d_succ, n_live, dbg, tile_mask, tiles, n_frames);
    if (dbg && dbg != 5) return hipGetLastError();   // 5 = everything (the probe's reference point), others leave d_succ alone
    hipLaunchKernelGGL(k_dart_link, dim3(blocks_for(n_darts, 256, env_cap("A3_LINK_BLOCKS", 4096))), dim3(256), 0, st, W, H, first_frame, pix_base, bits, d_rec, d_succ, n_darts, n_live);
    return hipGetLastError();
}

size_t entry_state_bytes() { return sizeof(EntryState); }
size_t fin_state_bytes() { return sizeof(FinState); }
size_t entry_slots(uint32_t n_darts) { return (size_t)entry_shard_cap(n_darts) * kEntryShards; }
size_t leader_list_bytes(uint32_t n_darts) { return (size_t)leader_shard_cap(n_darts) * kLeaderShards * 4; }

// leaders + ranks for every dart of the chunk.  loc/fin: JumpState[n_darts]; es_a/es_b: EntryState[n_darts] (upper bound);
// entry_count[16] and leader_count[16] arrive zeroed.
hipError_t launch_rank_cycles(hipStream_t st, uint32_t n_darts, int W, const uint64_t* d_rec, const uint32_t* d_succ,
                              JumpState* loc, uint32_t* entry_list,
                              unsigned int* entry_count, void* es_a, void* es_b, void* fin /* FinState[n_darts] */, uint32_t* leader_list,
                              unsigned int* leader_count, int max_rounds, DeviceCounters* ctr, const uint32_t* n_live, int dbg,
                              const uint32_t* frame_base, uint32_t* frame_entries /*nullptr: global rounds*/, uint32_t n_frames,
                              int phase /* 0: everything, 1: k_local_contract only, 2: what follows it */,
                              uint32_t min_edge_length, unsigned int* dead_count /* [16], nullptr: no border is dropped early (debug taps) */,
                              int trust_natural /* no fixpoint passes follow: k_cycle_select checks the natural starts inline */) {
    // entry_count[16] and leader_count[16] arrive zeroed (the caller's per-batch / per-chunk memset)
    const uint32_t ecap = entry_shard_cap(n_darts);
    if (phase != 2) {
        if (frame_entries && !dead_count)
            hipLaunchKernelGGL((k_local_contract<kLTFrame, false>), dim3((n_darts + kLTFrame - 1) / kLTFrame), dim3(256), 0, st, n_darts, W, d_rec, d_succ, loc,
                               entry_list, entry_count, ecap, frame_base, frame_entries, n_live, dbg, min_edge_length, dead_count, trust_natural);
        else if (frame_entries)   // (a dense graph's first batch, before its entries overflow k_entry_frame and the global rounds take over)
            hipLaunchKernelGGL((k_local_contract<kLTFrame, true>), dim3((n_darts + kLTFrame - 1) / kLTFrame), dim3(256), 0, st, n_darts, W, d_rec, d_succ, loc,
                               entry_list, entry_count, ecap, frame_base, frame_entries, n_live, dbg, min_edge_length, dead_count, trust_natural);
        else
            hipLaunchKernelGGL((k_local_contract<kLT, true>), dim3((n_darts + kLT - 1) / kLT), dim3(256), 0, st, n_darts, W, d_rec, d_succ, loc,
                               entry_list, entry_count, ecap, frame_base, frame_entries, n_live, dbg, min_edge_length, dead_count, trust_natural);
    }
    if (dbg || phase == 1) return hipGetLastError();
    EntryState* a = reinterpret_cast<EntryState*>(es_a);
    EntryState* b = reinterpret_cast<EntryState*>(es_b);
    if (frame_entries) {   // clean frames: every frame's entry list fits LDS, one launch instead of ~9
        hipLaunchKernelGGL(k_entry_frame, dim3(n_frames), dim3(256), 0, st, entry_list, frame_entries, frame_base, loc, a, ctr);
        const int fin_blocks = std::max(blocks_for(n_darts, 256, env_cap("A3_FIN_BLOCKS", 1536)), (int)(((uint64_t)n_darts + 256ull * 32 - 1) / (256ull * 32)));
        hipLaunchKernelGGL(k_jump_finalize, dim3(fin_blocks), dim3(256), 0, st, n_darts, loc, a, reinterpret_cast<FinState*>(fin),
                           leader_list, leader_count, leader_shard_cap(n_darts), n_live, ctr);
        return hipGetLastError();
    }
    const dim3 grid(blocks_for(n_darts / 16 + 1, 256, 1024)), block(256);   // entries are a few % of the darts on clean frames
    hipLaunchKernelGGL(k_entry_init, grid, block, 0, st, entry_list, entry_count, loc, a, ecap);
    for (int r = 0; r < max_rounds; r++) {
        hipLaunchKernelGGL(k_entry_jump, grid, block, 0, st, a, b, entry_count, ecap, r, ctr);
        EntryState* t = a; a = b; b = t;
    }
    const int fin_blocks = std::max(blocks_for(n_darts, 256, env_cap("A3_FIN_BLOCKS", 1536)), (int)(((uint64_t)n_darts + 256ull * 32 - 1) / (256ull * 32)));
    hipLaunchKernelGGL(k_jump_finalize, dim3(fin_blocks), block, 0, st, n_darts, loc, a, reinterpret_cast<FinState*>(fin),
                       leader_list, leader_count, leader_shard_cap(n_darts), n_live, ctr);
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, const JumpState* loc, const void* fin8, uint32_t n_darts, int W, const uint64_t* d_rec, const uint32_t* leader_list,
                          const unsigned int* leader_count, uint64_t* t_cur, uint64_t* t_next, DeviceCounters* ctr, int max_iters,
                          const uint32_t* n_live) {
    const FinState* fin = reinterpret_cast<const FinState*>(fin8);
    if (max_iters <= 0) return hipSuccess;   // k_cycle_select checks the natural starts itself; the caller re-runs the batch if they do not hold
    const dim3 grid(blocks_for(n_darts, 256, 4096)), block(256);
    hipLaunchKernelGGL(k_resolve_fast, dim3(blocks_for(n_darts / 16 + 1, 256, 1024)), block, 0, st, loc, fin, leader_list, leader_count,
                       leader_shard_cap(n_darts), W, d_rec, t_cur, ctr);
    hipLaunchKernelGGL(k_resolve_init, grid, block, 0, st, loc, fin, n_darts, t_cur, t_next, ctr, n_live);
    for (int it = 0; it < max_iters; it++) {
        hipLaunchKernelGGL(k_resolve_eval, grid, block, 0, st, fin, n_darts, W, d_rec, t_cur, t_next, it, ctr, n_live);
        hipLaunchKernelGGL(k_resolve_commit, grid, block, 0, st, fin, n_darts, t_cur, t_next, it, it == max_iters - 1 ? 1 : 0, ctr, n_live);
    }
    return hipGetLastError();
}

hipError_t launch_select_scatter(hipStream_t st, const JumpState* loc, const void* fin8, uint32_t n_darts, const uint32_t* leader_list,
                                 const unsigned int* leader_count, const uint32_t* d_succ, const uint64_t* t_cur,
                                 const uint32_t* frame_base, uint32_t n_frames, uint32_t first_frame, uint32_t min_edge_length,
                                 double eps_factor, double image_diag, ContourRec* contours, uint32_t* cyc_start_off,
                                 uint32_t max_contours, uint64_t max_points, DeviceCounters* ctr, const uint64_t* d_rec, uint32_t* points,
                                 const uint32_t* n_live, int inline_resolve_W, uint32_t* keep_tmp, int keep_all) {
    // 8192 workgroups for the graphs of clean frames (6-8 M darts), more for the tens of millions of darts of noise-like ones
    const dim3 grid(blocks_for(n_darts, 256, env_cap("A3_SCATTER_BLOCKS", (int)std::min<uint32_t>(65536u, std::max<uint32_t>(8192u, n_darts / 1024u))))), block(256);
    hipLaunchKernelGGL(k_cycle_select, dim3(blocks_for(n_darts / 64 + 1, 256, env_cap("A3_SELECT_BLOCKS", 1024))), block, 0, st, loc, reinterpret_cast<FinState*>(const_cast<void*>(fin8)), leader_list, leader_count, d_succ, t_cur,
                       frame_base, n_frames, first_frame, min_edge_length,
                       eps_factor, image_diag, contours, cyc_start_off, max_contours, max_points, ctr, leader_shard_cap(n_darts), d_rec,
                       inline_resolve_W, keep_tmp, keep_all);
    hipLaunchKernelGGL(k_scatter_points, grid, block, 0, st, reinterpret_cast<const FinState*>(fin8), n_darts, d_rec, contours, cyc_start_off, points, n_live, ctr);
    return hipGetLastError();
}

hipError_t launch_contour_quads(hipStream_t st, const ContourRec* contours, const DeviceCounters* ctr, uint32_t max_contours,
                                const uint32_t* points, double eps_factor, uint32_t min_edge_length, uint32_t first_frame, uint32_t max_cand,
                                CandRec* cands, uint32_t* cand_count, unsigned int* err_flags, int coords14, uint32_t n_darts) {
    // 2560 + 4096 workgroups for the millions of darts of a batch (or of one noise frame); a graph of a few ten thousand darts -- one
    // clean frame per call -- gets a grid in proportion: dispatching 6656 workgroups that find nothing to do took 8 of that call's 13 us
    const uint32_t b64 = std::min<uint32_t>((uint32_t)env_cap("A3_QUAD_BLOCKS64", 2560), std::max<uint32_t>(40u, n_darts / 2048u)),
                   b16 = std::min<uint32_t>((uint32_t)env_cap("A3_QUAD_BLOCKS16", 4096), std::max<uint32_t>(64u, n_darts / 1024u));
    hipLaunchKernelGGL(k_contour_quads, dim3(b64 + b16), dim3(256), 0, st, b64, contours, ctr, max_contours, points, eps_factor, min_edge_length,
                       first_frame, max_cand, cands, cand_count, err_flags, coords14);
    return hipGetLastError();
}

hipError_t launch_debug_clockwise(hipStream_t st, const int32_t* in, uint32_t n, int32_t* out) {
    hipLaunchKernelGGL(k_debug_clockwise, dim3((n + 63) / 64), dim3(64), 0, st, in, n, out);
    return hipGetLastError();
}

hipError_t launch_unpack_bits(hipStream_t st, const uint64_t* bits, int W, int H, uint8_t* out) {
    hipLaunchKernelGGL(k_unpack_bits, dim3((W + 255) / 256, H), dim3(256), 0, st, bits, W, H, out);
    return hipGetLastError();
}

}  // namespace a3
