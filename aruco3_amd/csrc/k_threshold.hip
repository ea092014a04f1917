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
#include "a3_common.h"

namespace a3 {

constexpr int T_TW = 240;           // output tile width  (30 groups of 8)
constexpr int T_TH = 64;            // output tile height (8 segments of 8 rows)
constexpr int T_R = 7;              // fast path radius
constexpr int T_LW = 256;           // loaded columns: x0-8 .. x0+247
constexpr int T_LH = T_TH + 2 * T_R;  // 78 loaded rows: y0-7 .. y0+70
constexpr int T_G = T_TW / 8;       // 30 column groups

__device__ __forceinline__ uint32_t luma_of(uint32_t r, uint32_t g, uint32_t b) {
    return (2126u * r + 7152u * g + 722u * b) / 10000u;
}

// 4 consecutive pixels starting at (x, y) -> 4 grey bytes packed little-endian; 0 outside the image
template <int FMT>
__device__ __forceinline__ uint32_t load_grey4(const uint8_t* __restrict__ frame, size_t row_stride, int x, int y, int W, int H,
                                               bool aligned) {
    if (y < 0 || y >= H || x + 3 < 0 || x >= W) return 0u;
    const uint8_t* row = frame + (size_t)y * row_stride;
    constexpr int BPP = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_RGBA8 ? 4 : 1);
    if (aligned && x >= 0 && x + 3 < W) {
        if constexpr (FMT == A3_FMT_RGB8) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row + (size_t)x * 3);
            uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            uint32_t g0 = luma_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u);
            uint32_t g1 = luma_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u);
            uint32_t g2 = luma_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u);
            uint32_t g3 = luma_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24);
            return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else if constexpr (FMT == A3_FMT_RGBA8) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + (size_t)x * 4);
            uint32_t g0 = luma_of(q.x & 255u, (q.x >> 8) & 255u, (q.x >> 16) & 255u);
            uint32_t g1 = luma_of(q.y & 255u, (q.y >> 8) & 255u, (q.y >> 16) & 255u);
            uint32_t g2 = luma_of(q.z & 255u, (q.z >> 8) & 255u, (q.z >> 16) & 255u);
            uint32_t g3 = luma_of(q.w & 255u, (q.w >> 8) & 255u, (q.w >> 16) & 255u);
            return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else {
            return *reinterpret_cast<const uint32_t*>(row + x);
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int xi = x + i;
        if (xi < 0 || xi >= W) continue;
        const uint8_t* p = row + (size_t)xi * BPP;
        uint32_t g = BPP == 1 ? (uint32_t)p[0] : luma_of(p[0], p[1], p[2]);
        out |= g << (8 * i);
    }
    return out;
}

template <int FMT>
__global__ __launch_bounds__(256) void k_grey_threshold7(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride,
                                                         int W, int H, uint8_t* __restrict__ grey, uint8_t* __restrict__ bits,
                                                         int aligned_in, int aligned_out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_g[T_LH][T_LW];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[T_LH][T_G][8];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * T_TW, y0 = blockIdx.y * T_TH;
    const uint32_t f = blockIdx.z;
    const uint8_t* frame = pixels + (size_t)f * frame_stride;
    uint8_t* gout = grey + (size_t)f * W * H;
    const size_t bpr = (size_t)words_per_row((uint32_t)W) * 8;  // bytes per packed row
    uint8_t* bout = bits + (size_t)f * bpr * H;

    // ---- phase A: load + convert; lane l of a wave owns columns 4l..4l+3 of one loaded row ----
    const int lane_c = (tid & 63) * 4;
    const int wave = tid >> 6;
#pragma unroll 5
    for (int r = wave; r < T_LH; r += 4) {
        const int x = x0 - 8 + lane_c, y = y0 - T_R + r;
        uint32_t g4 = load_grey4<FMT>(frame, row_stride, x, y, W, H, aligned_in != 0);
        *reinterpret_cast<uint32_t*>(&s_g[r][lane_c]) = g4;
        // the tile's own pixels also go out as Detection.grey
        if (r >= T_R && r < T_R + T_TH && lane_c >= 8 && lane_c < 8 + T_TW && y < H && x < W) {
            uint8_t* dst = gout + (size_t)y * W + x;
            if (aligned_out && x + 3 < W) *reinterpret_cast<uint32_t*>(dst) = g4;
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) if (x + i < W) dst[i] = (uint8_t)(g4 >> (8 * i));
            }
        }
    }
    __syncthreads();

    // ---- phase B: horizontal 15-tap sums, 8 outputs per task from 24 grey bytes ----
    for (int t = tid; t < T_LH * T_G; t += 256) {
        const int r = t / T_G, j = t - r * T_G;
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&s_g[r][8 * j]);
        const uint64_t q0 = src[0], q1 = src[1], q2 = src[2];
        uint32_t b[24];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            b[i] = (uint32_t)(q0 >> (8 * i)) & 255u;
            b[8 + i] = (uint32_t)(q1 >> (8 * i)) & 255u;
            b[16 + i] = (uint32_t)(q2 >> (8 * i)) & 255u;
        }
        uint32_t s = 0;
#pragma unroll
        for (int k = 1; k <= 15; k++) s += b[k];
        uint32_t o[8];
        o[0] = s;
#pragma unroll
        for (int i = 1; i < 8; i++) { s += b[15 + i] - b[i]; o[i] = s; }
        uint4 v;
        v.x = o[0] | (o[1] << 16); v.y = o[2] | (o[3] << 16); v.z = o[4] | (o[5] << 16); v.w = o[6] | (o[7] << 16);
        *reinterpret_cast<uint4*>(&s_h[r][j][0]) = v;
    }
    __syncthreads();

    // ---- phase C: vertical sliding sums on packed u16 pairs, compare, store ----
    if (tid < T_G * 8) {
        const int j = tid % T_G, seg = tid / T_G;
        const int xb = x0 + 8 * j;
        if (xb < W) {
            uint32_t ax[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int x = xb + i;
                int hi = x + T_R < W - 1 ? x + T_R : W - 1, lo = x > T_R ? x - T_R : 0;
                ax[i] = x < W ? (uint32_t)(hi - lo + 1) : 0u;
            }
            uint4 acc = make_uint4(0, 0, 0, 0);
            const int rr0 = seg * 8;
#pragma unroll
            for (int k = 0; k < 15; k++) {
                uint4 v = *reinterpret_cast<const uint4*>(&s_h[rr0 + k][j][0]);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int rr = rr0 + i, y = y0 + rr;
                if (i > 0) {
                    uint4 a = *reinterpret_cast<const uint4*>(&s_h[rr + 14][j][0]);
                    uint4 d = *reinterpret_cast<const uint4*>(&s_h[rr - 1][j][0]);
                    acc.x += a.x - d.x; acc.y += a.y - d.y; acc.z += a.z - d.z; acc.w += a.w - d.w;
                }
                if (y >= H) break;
                const int yhi = y + T_R < H - 1 ? y + T_R : H - 1, ylo = y > T_R ? y - T_R : 0;
                const uint32_t ay = (uint32_t)(yhi - ylo + 1);
                const uint64_t gq = *reinterpret_cast<const uint64_t*>(&s_g[rr + T_R][8 + 8 * j]);
                const uint32_t sums[8] = {acc.x & 0xFFFFu, acc.x >> 16, acc.y & 0xFFFFu, acc.y >> 16,
                                          acc.z & 0xFFFFu, acc.z >> 16, acc.w & 0xFFFFu, acc.w >> 16};
                uint32_t outb = 0;
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    uint32_t gv = (uint32_t)(gq >> (8 * p)) & 255u;
                    if (sums[p] < (gv + 1u) * (ax[p] * ay)) outb |= 1u << p;   // ax == 0 past the right edge: stays 0
                }
                bout[(size_t)y * bpr + (xb >> 3)] = (uint8_t)outb;
            }
        }
    }
}

// ---- generic radius: plain two-kernel path (correct for any threshold_window, not tuned) ----
template <int FMT>
__global__ void k_grey_generic(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride, int W, int H,
                               uint8_t* __restrict__ grey) {
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= W) return;
    const uint32_t f = blockIdx.z;
    uint32_t g4 = load_grey4<FMT>(pixels + (size_t)f * frame_stride, row_stride, x4, y, W, H, false);
    uint8_t* dst = grey + (size_t)f * W * H + (size_t)y * W + x4;
    for (int i = 0; i < 4; i++) if (x4 + i < W) dst[i] = (uint8_t)(g4 >> (8 * i));
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
        const int bpp = fmt == A3_FMT_RGB8 ? 3 : (fmt == A3_FMT_RGBA8 ? 4 : 1);
        const size_t need = bpp == 3 ? 4 : (bpp == 4 ? 16 : 4);
        const int aligned_in = ((uintptr_t)pixels % need == 0) && (row_stride % need == 0) && (frame_stride % need == 0);
        const int aligned_out = (W % 4 == 0) && ((uintptr_t)grey % 4 == 0);
        uint8_t* bin = reinterpret_cast<uint8_t*>(bits);
        if (W % 64 != 0) {  // packed rows end in padding bits that no tile writes
            hipError_t e = hipMemsetAsync(bits, 0, (size_t)words_per_row((uint32_t)W) * 8 * H * n, st);
            if (e != hipSuccess) return e;
        }
        dim3 grid((W + T_TW - 1) / T_TW, (H + T_TH - 1) / T_TH, n), block(256);
        if (fmt == A3_FMT_RGB8)
            hipLaunchKernelGGL(k_grey_threshold7<A3_FMT_RGB8>, grid, block, 0, st, pixels, row_stride, frame_stride, W, H, grey, bin, aligned_in, aligned_out);
        else if (fmt == A3_FMT_RGBA8)
            hipLaunchKernelGGL(k_grey_threshold7<A3_FMT_RGBA8>, grid, block, 0, st, pixels, row_stride, frame_stride, W, H, grey, bin, aligned_in, aligned_out);
        else
            hipLaunchKernelGGL(k_grey_threshold7<A3_FMT_L8>, grid, block, 0, st, pixels, row_stride, frame_stride, W, H, grey, bin, aligned_in, aligned_out);
        return hipGetLastError();
    }
    dim3 block(64), grid4(((W + 3) / 4 + 63) / 64, H, n), grid1(words_per_row((uint32_t)W), H, n);
    if (fmt == A3_FMT_RGB8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGB8>, grid4, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else if (fmt == A3_FMT_RGBA8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGBA8>, grid4, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else hipLaunchKernelGGL(k_grey_generic<A3_FMT_L8>, grid4, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    hipLaunchKernelGGL(k_threshold_generic, grid1, block, 0, st, grey, W, H, (int)radius, bits);
    return hipGetLastError();
}

}  // namespace a3
