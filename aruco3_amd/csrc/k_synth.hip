// k_synth.hip -- device-side synthetic frame generator: the frames of aruco3_amd/synth.py
// rendered straight into HBM, so a benchmark or a soak test needs neither host rendering nor the 6 MB/frame H2D copy.
// Not part of Detector::detect; it plays the role of the reference's test renderer (tests/common) and of
// ARDictionary::make_binary_image (src/dictionaries.rs:209-232: the cell order is the caller's choice, carried in `cells`).
// The host decides the layout (which markers, where: aruco3_amd/synth.py::frame_layout, same seeds as the host renderer);
// one thread per pixel paints the background gradient, then every marker whose bounding box covers the pixel through the
// inverse homography with ss x ss supersampling, then the channel tint and optional Gaussian noise.
#include "a3_common.h"

namespace a3 {

__device__ __forceinline__ uint32_t hash32(uint32_t x) {   // lowbias32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void k_synth_render(const a3_synth_frame* __restrict__ frames, const a3_synth_marker* __restrict__ markers,
                                                      uint32_t W, uint32_t H, int paper, float black, float white, int ss,
                                                      uint8_t* __restrict__ out, size_t row_stride, size_t frame_stride) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const a3_synth_frame fr = frames[f];
    const float xs = W > 1 ? -1.0f + 2.0f * (float)x / (float)(W - 1) : -1.0f;
    const float ys = H > 1 ? -1.0f + 2.0f * (float)y / (float)(H - 1) : -1.0f;
    float gray = fr.base + fr.gx * xs + fr.gy * ys;
    const float k = (float)(ss * ss);
    for (uint32_t mi = 0; mi < fr.n_markers; mi++) {
        const a3_synth_marker& m = markers[fr.first_marker + mi];
        if ((int)x < m.x0 || (int)x >= m.x1 || (int)y < m.y0 || (int)y >= m.y1) continue;
        const int n = (int)m.n;
        float acc = 0.0f, cov = 0.0f;
        for (int sy = 0; sy < ss; sy++)
            for (int sx = 0; sx < ss; sx++) {
                const float px = (float)x + ((float)sx + 0.5f) / (float)ss - 0.5f, py = (float)y + ((float)sy + 0.5f) / (float)ss - 0.5f;
                const float d = m.hinv[6] * px + m.hinv[7] * py + m.hinv[8];
                const float u = (m.hinv[0] * px + m.hinv[1] * py + m.hinv[2]) / d;
                const float v = (m.hinv[3] * px + m.hinv[4] * py + m.hinv[5]) / d;
                const bool inside = u >= -1.0f && u < (float)(n + 1) && v >= -1.0f && v < (float)(n + 1);
                if (!inside) continue;
                const int ui = (int)floorf(u), vi = (int)floorf(v);   // cell coordinates; -1 and n are the white quiet zone
                const bool in_cells = ui >= 0 && ui < n && vi >= 0 && vi < n;
                const int cell = vi * n + ui;   // < 121: two words
                const float val = in_cells ? (float)((m.cells[cell >> 6] >> (cell & 63)) & 1ull) : 1.0f;
                if (paper) { if (val == 0.0f) { acc += black; cov += 1.0f; } }
                else { acc += black + (white - black) * val; cov += 1.0f; }
            }
        gray = (gray * (k - cov) + acc) / k;
    }
    float rgb[3] = {gray * 1.00f, gray * 0.98f, gray * 0.94f};   // slight, fixed channel tint so that the luma weights matter
    if (fr.noise_sigma > 0.0f) {
        const uint32_t pix = (y * W + x) * 3u;
        for (int c = 0; c < 3; c++) {   // Box-Muller on two hashed uniforms per channel
            const uint32_t h1 = hash32(pix + (uint32_t)c + (uint32_t)fr.seed * 0x9E3779B9u), h2 = hash32(h1 ^ (uint32_t)(fr.seed >> 32) ^ 0x85EBCA6Bu);
            const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
            rgb[c] += fr.noise_sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
        }
    }
    uint8_t* dst = out + (size_t)f * frame_stride + (size_t)y * row_stride + (size_t)x * 3u;
    for (int c = 0; c < 3; c++) dst[c] = (uint8_t)rintf(fminf(fmaxf(rgb[c], 0.0f), 255.0f));
}

hipError_t launch_synth_render(hipStream_t st, const a3_synth_frame* frames, uint32_t n_frames, const a3_synth_marker* markers, uint32_t W,
                               uint32_t H, int paper, float black, float white, int ss, uint8_t* out, size_t row_stride, size_t frame_stride) {
    hipLaunchKernelGGL(k_synth_render, dim3((W + 255) / 256, H, n_frames), dim3(256), 0, st, frames, markers, W, H, paper, black, white, ss, out,
                       row_stride, frame_stride);
    return hipGetLastError();
}

// A stand-in for a collective's channel kernels (a3_internal.h: a3_debug_spin): `workgroups` workgroups that stay resident
// for `usec` microseconds, each lane holding kSpinRegs registers' worth of state alive and touching a little memory per turn.  The
// exit condition is the wall clock (constant 100 MHz), which every wave reaches.
constexpr int kSpinRegs = 48;
__global__ __launch_bounds__(512) void k_spin(unsigned long long ticks, uint32_t* __restrict__ sink) {
    const unsigned long long t0 = wall_clock64();
    uint32_t r[kSpinRegs];
#pragma unroll
    for (int i = 0; i < kSpinRegs; i++) r[i] = threadIdx.x * 31u + (uint32_t)i;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < kSpinRegs; i++) r[i] = r[i] * 1664525u + 1013904223u + r[(i + 7) % kSpinRegs];
        r[0] += sink[(blockIdx.x * 64u + (threadIdx.x & 63u)) & 4095u];   // a read per turn, as a channel kernel polls its flags
        __builtin_amdgcn_s_sleep(8);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < kSpinRegs; i++) acc ^= r[i];
    if (acc == 0x12345678u) sink[threadIdx.x & 4095u] = acc;
}

hipError_t launch_spin(hipStream_t st, int workgroups, int threads, int usec, uint32_t* sink) {
    hipLaunchKernelGGL(k_spin, dim3(workgroups), dim3(threads), 0, st, (unsigned long long)usec * 100ull, sink);
    return hipGetLastError();
}

}  // namespace a3
static_assert(sizeof(a3_synth_marker) == 80 && sizeof(a3_synth_frame) == 32, "record layouts mirrored in aruco3_amd/synth.py");
