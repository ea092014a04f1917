// k_threshold.hip -- K1: pixels -> grey (into_luma8) -> 15x15 adaptive threshold, fused.
//
// Replaces `image.into_luma8()` + `imageproc::contrast::adaptive_threshold(&grey, 7)`
// (src/aruco.rs:60-61).  Pure integer, so every output byte must equal the oracle's.
//
//   grey  L = (2126 R + 7152 G + 722 B) / 10000          (u32, truncating)
//   white iff L >= floor(sum / area) over the window clipped to the image
//         <=> sum < (L + 1) * area                        (no division)
//
// What must move per pixel: 3 B read (RGB8) + 1/8 B written -- the thresholded image leaves the kernel bit-packed
// (a3_common.h: words_per_row), which is all the contour stage reads, and no grey plane is written unless somebody asks
// for Detection.grey (debug taps).  No LDS, no barriers: one wave walks down a strip of 1024 columns, lane l owns 16
// consecutive columns, and everything between the 48-byte row slice it loads and the 16 bits it stores stays in
// registers (k_grey_threshold7 below).  The arithmetic is laid out for instruction count -- the first register-resident
// version spent 21 VALU lane-instructions per pixel and was issue-bound at 0.56 of the HBM roofline:
//   grey        two v_dot4_u32_u8 + one v_mul_hi_u32_u24 per pixel, the >> 5 writes the byte into place (SDWA dst_sel)
//   horizontal  15-wide sums of the NEW grey row first, on bytes: four v_sad_u8 start two chains (pixels 0 and 8), each
//               slide step advances both chains with two packed-u16 instructions (byte pairs come from v_perm_b32)
//   vertical    15-row sliding sums of those row sums as packed u16 pairs: v_pk_add_u16 / v_pk_sub_u16 against a ring of
//               the last 15 rows of row sums (15 * 255 * 15 < 2^16)
//   compare     sum < (L + 1) * area on pairs: v_pk_mad_u16, saturating v_pk_sub_u16, v_pk_min_u16, bits shifted in by
//               another v_pk_mad_u16
// Out-of-image pixels count as 0, which makes the unclipped sum equal the clipped one; the area is the clipped window.
#include "k_threshold_k1.h"

namespace a3 {

// ---- generic radius: plain kernels (correct for any threshold_window and any alignment, not tuned): what is left for them are
// windows above 15 and radius 0 (k_threshold_big.hip: ring_kernel_applies) ----
// four consecutive pixels per thread: dword loads and one dword store where the row is 4-byte aligned, bytes otherwise
template <int FMT>
__global__ void k_grey_generic(const uint8_t* __restrict__ pixels, size_t row_stride, size_t frame_stride, int W, int H,
                               uint8_t* __restrict__ grey) {
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
    if (x >= W) return;
    const uint32_t f = blockIdx.z;
    constexpr int BPP = FMT == A3_FMT_RGB8 ? 3 : (FMT == A3_FMT_L8 ? 1 : 4);
    const uint8_t* p = pixels + (size_t)f * frame_stride + (size_t)y * row_stride + (size_t)x * BPP;
    uint8_t* out = grey + (size_t)f * W * H + (size_t)y * W + x;
    auto luma = [](uint32_t b0, uint32_t b1, uint32_t b2) -> uint32_t { return FMT == A3_FMT_BGRA8 ? luma_of(b2, b1, b0) : luma_of(b0, b1, b2); };
    if (x + 3 < W && (reinterpret_cast<uintptr_t>(p) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        uint32_t d[BPP];
#pragma unroll
        for (int i = 0; i < BPP; i++) d[i] = reinterpret_cast<const uint32_t*>(p)[i];
        uint32_t g = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t v;
            if (BPP == 1) v = (d[0] >> (8 * i)) & 0xFFu;
            else {
                const int b = i * BPP;   // byte offset of pixel i inside d[]
                auto byte = [&](int k) -> uint32_t { return (d[k >> 2] >> (8 * (k & 3))) & 0xFFu; };
                v = luma(byte(b), byte(b + 1), byte(b + 2));
            }
            g |= v << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(out) = g;
        return;
    }
    for (int i = 0; i < 4 && x + i < W; i++) {
        const uint8_t* q = p + (size_t)i * BPP;
        out[i] = BPP == 1 ? q[0] : (uint8_t)luma(q[0], q[1], q[2]);
    }
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

// Windows above 7 on this path, separable: horizontal sums of the grey plane into a u16 plane (2R+1 <= 257 values of at most 255), then a
// vertical running sum per column with the compare.  Plain kernels (no register ring: (2R+1)^2 * 256 no longer fits 16 bits), but
// the work per pixel no longer grows with the window's AREA as in k_threshold_generic.
// k_hsum_generic: a workgroup covers 1024 columns of one row, four consecutive outputs per thread (a sliding sum: 2R + 4 byte reads for
// four results); the row segment is staged in LDS with dword loads.
constexpr int kHsumCols = 1024;
__global__ __launch_bounds__(256) void k_hsum_generic(const uint8_t* __restrict__ grey, int W, int H, int radius, uint16_t* __restrict__ hsum) {
    extern __shared__ uint8_t s_row[];   // grey bytes of columns x0 - R4 .. of this row, R4 = R rounded up to 4 (zeros outside the image)
    const int x0 = blockIdx.x * kHsumCols, y = blockIdx.y;
    const size_t base = ((size_t)blockIdx.z * H + y) * W;
    const int r4 = (radius + 3) & ~3, span = kHsumCols + 2 * r4;
    const bool row_aligned = ((base & 3) == 0) && ((W & 3) == 0);
    for (int i = threadIdx.x * 4; i < span; i += 1024) {
        const int x = x0 - r4 + i;
        uint32_t v = 0;
        if (row_aligned && x >= 0 && x + 3 < W) v = *reinterpret_cast<const uint32_t*>(grey + base + x);
        else {
#pragma unroll
            for (int b = 0; b < 4; b++) if (x + b >= 0 && x + b < W) v |= (uint32_t)grey[base + x + b] << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(s_row + i) = v;
    }
    __syncthreads();
    const int x = x0 + 4 * (int)threadIdx.x;
    if (x >= W) return;
    const uint8_t* p = s_row + r4 + 4 * threadIdx.x - radius;   // p[k] = grey(x - R + k)
    uint32_t sum = 0;
    for (int k = 0; k <= 2 * radius; k++) sum += p[k];
    uint32_t out[4];
    out[0] = sum;
#pragma unroll
    for (int j = 1; j < 4; j++) { sum += p[2 * radius + j]; sum -= p[j - 1]; out[j] = sum; }
#pragma unroll
    for (int j = 0; j < 4; j++) if (x + j < W) hsum[base + x + j] = (uint16_t)out[j];
}

// one wave per 64 columns and strip of rows: lane = column, V = sum of hsum over the rows y - R .. y + R inside the image; the loads of
// eight rows are issued together (they do not depend on V), then the eight rows are finished in order
constexpr int kVsumRows = 128;
__global__ __launch_bounds__(64) void k_vsum_threshold_generic(const uint8_t* __restrict__ grey, const uint16_t* __restrict__ hsum, int W, int H, int radius,
                                                               uint64_t* __restrict__ bits) {
    const int x = blockIdx.x * 64 + threadIdx.x, y0 = blockIdx.y * kVsumRows, y1 = min(H, y0 + kVsumRows);
    const size_t plane = (size_t)blockIdx.z * W * H;
    const bool in = x < W;
    const int xc = in ? x : W - 1;
    const uint16_t* hs = hsum + plane + xc;
    const uint8_t* g = grey + plane + xc;
    const uint32_t ax = in ? (uint32_t)(min(x + radius, W - 1) - max(x - radius, 0) + 1) : 0u;
    uint32_t V = 0;
    for (int yy = max(y0 - radius, 0); yy <= min(y0 + radius, H - 1); yy++) V += hs[(size_t)yy * W];
    const size_t wpr = words_per_row((uint32_t)W);
    uint64_t* brow = bits + (size_t)blockIdx.z * wpr * H + blockIdx.x;
    constexpr int B = 8;
    for (int yb = y0; yb < y1; yb += B) {
        uint32_t gv[B], add[B], sub[B];
#pragma unroll
        for (int u = 0; u < B; u++) {   // unconditional loads from clamped rows, masked afterwards
            const int y = yb + u, ya = y + radius + 1, ys = y - radius;
            gv[u] = g[(size_t)min(y, H - 1) * W];
            add[u] = hs[(size_t)min(ya, H - 1) * W] & (0u - (uint32_t)(ya < H));
            sub[u] = hs[(size_t)max(ys, 0) * W] & (0u - (uint32_t)(ys >= 0));
        }
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int y = yb + u;
            if (y >= y1) break;
            const uint32_t ay = (uint32_t)(min(y + radius, H - 1) - max(y - radius, 0) + 1);
            const bool white = in && V < (gv[u] + 1u) * (ax * ay);     // sum < (L + 1) * area  <=>  L >= floor(sum / area)
            const unsigned long long m = __ballot(white);
            if (threadIdx.x == 0) brow[(size_t)y * wpr] = m;
            V += add[u]; V -= sub[u];
        }
    }
}

// ---- host launcher -------------------------------------------------------------------------
// K1 waves per SIMD the strip model sizes a launch for (a3_internal.h: a3_debug_set_k1_waves).  2 = the whole chip in one
// round (every register of every SIMD); 1 = one wave per SIMD, strips twice as tall: half of every SIMD's registers and wave
// slots stay free for the kernels of another batch's contour / decode stage.
int g_k1_waves = A3_T_WAVES;
void set_k1_waves(int w) { g_k1_waves = w < 1 ? 1 : (w > A3_T_WAVES ? A3_T_WAVES : w); }
bool k1_build_is_default() { return A3_T_LPX == 16 && A3_T_PF == 3 && A3_T_WAVES == 2 && A3_T_RECOMPUTE == 0 && A3_T_ADD32 == 1 && A3_T_LUMA_GROUPS == 1 && A3_T_COMPARE4 == 1; }
int g_k1_cus = 256;   // compute units the kernel's stream may use (a3_debug_set_partition)
#ifdef A3_TUNING
// (tuning builds only) where the radius-7 kernel writes one record per wave: {begin, end (100 MHz), HW_ID, XCC_ID}; nullptr = off
extern "C" __attribute__((visibility("default"))) int a3_debug_set_k1_stamps(void* device_buffer) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(device_buffer);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k1_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -5;
}
#endif
void set_k1_cus(int c) { g_k1_cus = c < 8 ? 8 : (c > 256 ? 256 : c); }

// radii 1..3 and 4..6: k_threshold_r1.hip, k_threshold_r2.hip
hipError_t launch_k1_r1(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n,
                        uint8_t* grey, uint64_t* bits);
hipError_t launch_k1_r2(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n,
                        uint8_t* grey, uint64_t* bits);

// threshold windows the register-resident kernel covers: radii 1..kFusedMaxRadius.  Beyond 7 the packed 16-bit arithmetic ends --
// (2R+1)^2 * 256 no longer fits a u16 from R = 8 on (289 * 256 = 73 984), so window sums and the compare would need 32-bit lanes and a ring
// of 2R+1 rows no register file holds.
// Radii 8..31 have a fused kernel of their own (k_threshold_big.hip: the grey ring in registers + LDS, 32-bit sums); what is left for
// the separable path -- which needs a grey plane and a plane of row sums -- is radius 0 and radii above 31.
constexpr uint32_t kFusedMaxRadius = 7;
bool ring_kernel_applies(uint32_t radius, const uint8_t* pixels, size_t row_stride, size_t frame_stride, int W);
hipError_t launch_ring_threshold(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H,
                                 uint32_t n, uint8_t* grey, uint64_t* bits);
bool threshold_writes_grey_plane(uint32_t radius, const uint8_t* pixels, size_t row_stride, size_t frame_stride, int W) {
    return (radius == 0 || radius > kFusedMaxRadius) && !ring_kernel_applies(radius, pixels, row_stride, frame_stride, W);
}

hipError_t launch_grey_threshold(hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H,
                                 uint32_t n, uint32_t radius, uint8_t* grey, uint64_t* bits, uint16_t* hsum_tmp) {
    switch (radius) {
        case 1: case 2: case 3: return launch_k1_r1(radius, st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 4: case 5: case 6: return launch_k1_r2(radius, st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 7: return launch_k1<7>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        default: break;
    }
    if (ring_kernel_applies(radius, pixels, row_stride, frame_stride, W)) return launch_ring_threshold(radius, st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
    dim3 block(64), gridg((W + 255) / 256, H, n), grid1(words_per_row((uint32_t)W), H, n);
    if (fmt == A3_FMT_RGB8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGB8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else if (fmt == A3_FMT_RGBA8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_RGBA8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else if (fmt == A3_FMT_BGRA8) hipLaunchKernelGGL(k_grey_generic<A3_FMT_BGRA8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    else hipLaunchKernelGGL(k_grey_generic<A3_FMT_L8>, gridg, block, 0, st, pixels, row_stride, frame_stride, W, H, grey);
    if (hsum_tmp && radius <= 128u) {   // separable: row sums (<= 257 * 255 fit a u16), then a running column sum with the compare
        hipLaunchKernelGGL(k_hsum_generic, dim3((W + kHsumCols - 1) / kHsumCols, H, n), dim3(256), kHsumCols + 2 * ((radius + 3) & ~3u) + 16, st, grey, W, H, (int)radius, hsum_tmp);
        hipLaunchKernelGGL(k_vsum_threshold_generic, dim3(words_per_row((uint32_t)W), (H + kVsumRows - 1) / kVsumRows, n), dim3(64), 0, st, grey, hsum_tmp,
                           W, H, (int)radius, bits);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_threshold_generic, grid1, block, 0, st, grey, W, H, (int)radius, bits);
    return hipGetLastError();
}

}  // namespace a3
