// scatterbench.hip -- what the decode stage's sampling pattern can reach on this GPU with everything but the loads taken away
// (tuning aid, not part of the library).  k_decode reads, per candidate, 49 x 49 bilinear samples from the caller's RGB frame:
// per sample two 12-byte reads (the two taps of a row are adjacent) from the 4-byte-aligned address below the first tap, one
// per image row of the sample -- 4802 scattered 12-byte reads per candidate, arranged 8 x 8 output pixels per wave
// instruction, 256 threads per candidate, KU samples in flight per lane.  This program issues exactly those reads for
// `n_cand` quads of BASELINE config 2's shape (sides 140..280 px, any rotation, spread over 256 frames of 1920x1080 RGB) and
// does nothing with them but an XOR, at the occupancy the real kernel runs at (5 workgroups of 256 threads per CU, set with
// LDS).  Its time is the ceiling the memory system puts on the sampling loop; tools/kernel_probe.py times the real loop
// (k_decode with dbg = 2: stop after sampling) on the same kind of data.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Quad { float ox, oy, ux, uy, vx, vy; uint32_t frame, pad; };   // sample (x, y) -> origin + x * u + y * v (pixels)

template <int KU, int MODE>   // MODE 0: 2 x 12 B per sample (the kernel's pattern); 1: 2 x 4 B (request-rate probe: same addresses, a third of the bytes)
__global__ __launch_bounds__(256, 5) void k_scatter(const uint8_t* __restrict__ frames, size_t row_stride, size_t frame_stride, const Quad* __restrict__ quads,
                                                    uint32_t n_cand, uint32_t S, uint32_t* __restrict__ out) {
    extern __shared__ uint8_t lds[];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    const uint32_t nbx = (S + 7u) / 8u, n_slots = nbx * nbx * 64u;
    for (uint32_t c = blockIdx.x; c < n_cand; c += gridDim.x) {
        const Quad q = quads[c];
        const uint8_t* img = frames + (size_t)q.frame * frame_stride;
        for (uint32_t i0 = tid; i0 < n_slots; i0 += 256 * KU) {
            uint32_t t[KU][3], b[KU][3];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const uint32_t slot = i0 + 256u * u, blk = slot >> 6, l = slot & 63u;
                const uint32_t by = blk / nbx, bx = blk - by * nbx;
                const uint32_t x = min(bx * 8u + (l & 7u), S - 1u), y = min(by * 8u + (l >> 3), S - 1u);
                const float px = q.ox + (float)x * q.ux + (float)y * q.vx, py = q.oy + (float)x * q.uy + (float)y * q.vy;
                const uint32_t ix = (uint32_t)px, iy = (uint32_t)py;
                const uintptr_t pt = reinterpret_cast<uintptr_t>(img + (size_t)iy * row_stride + 3u * (size_t)ix);
                const uint32_t* qt = reinterpret_cast<const uint32_t*>(pt & ~(uintptr_t)3);
                const uint32_t* qb = reinterpret_cast<const uint32_t*>((pt + row_stride) & ~(uintptr_t)3);
                if (MODE == 0) { t[u][0] = qt[0]; t[u][1] = qt[1]; t[u][2] = qt[2]; b[u][0] = qb[0]; b[u][1] = qb[1]; b[u][2] = qb[2]; }
                else { t[u][0] = qt[0]; t[u][1] = 0; t[u][2] = 0; b[u][0] = qb[0]; b[u][1] = 0; b[u][2] = 0; }
            }
#pragma unroll
            for (int u = 0; u < KU; u++) acc ^= t[u][0] ^ t[u][1] ^ t[u][2] ^ b[u][0] ^ b[u][1] ^ b[u][2];
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + tid] = acc + lds[tid];
}

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}
// something that evicts the frames from the caches between launches (inside the pipeline the decode stage finds them cold:
// 1.6 GB of frames went through K1 three kernels earlier, the contour stage's buffers since)
__global__ void k_sweep(const uint32_t* p, size_t n, uint32_t* out) {
    uint32_t a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i];
    if (a == 0x12345678u) out[0] = a;
}

template <int KU, int MODE>
float run(const uint8_t* d, size_t rs, size_t fs, const Quad* dq, uint32_t n_cand, uint32_t* out, const uint32_t* junk, size_t junk_n, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float total = 0;
    for (int i = 0; i < reps + 1; i++) {
        hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, 0, junk, junk_n, out);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_scatter<KU, MODE>), dim3(4096), dim3(256), 9000, 0, d, rs, fs, dq, n_cand, 49u, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i) total += ms;
    }
    return total / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int frames = 256, W = 1920, H = 1080;
    const uint32_t n_cand = argc > 1 ? (uint32_t)atoi(argv[1]) : 2571u;
    const size_t rs = (size_t)W * 3, fs = rs * H, bytes = fs * frames;
    uint8_t* d; uint32_t* out; uint32_t* junk; Quad* dq;
    const size_t junk_n = (size_t)512 << 18;   // 512 MB
    CK(hipMalloc(&d, bytes + 65536)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&junk, junk_n * 4)); CK(hipMalloc(&dq, sizeof(Quad) * n_cand));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(d), (bytes + 65536) / 4);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, junk, junk_n);
    std::vector<Quad> q(n_cand);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (uint32_t i = 0; i < n_cand; i++) {
        const double side = 140.0 + 140.0 * rnd(), ang = 6.283185307 * rnd(), step = side / 49.0;
        const double cx = 250.0 + (W - 500.0) * rnd(), cy = 250.0 + (H - 500.0) * rnd();
        Quad& k = q[i];
        k.ux = (float)(step * std::cos(ang)); k.uy = (float)(step * std::sin(ang)); k.vx = -k.uy; k.vy = k.ux;
        k.ox = (float)(cx - 24.5 * (k.ux + k.vx)); k.oy = (float)(cy - 24.5 * (k.uy + k.vy));
        k.frame = (uint32_t)((uint64_t)i * frames / n_cand); k.pad = 0;   // ~10 candidates per frame, as in config 2
    }
    CK(hipMemcpy(dq, q.data(), sizeof(Quad) * n_cand, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    printf("%u candidates x 2401 samples x 2 rows; us per launch (frames cold: 512 MB swept in between)\n", n_cand);
    printf("12-byte reads, KU 1 / 2 / 4 samples in flight per lane: %.1f / %.1f / %.1f us\n",
           run<1, 0>(d, rs, fs, dq, n_cand, out, junk, junk_n, 10), run<2, 0>(d, rs, fs, dq, n_cand, out, junk, junk_n, 10), run<4, 0>(d, rs, fs, dq, n_cand, out, junk, junk_n, 10));
    printf(" 4-byte reads at the same addresses, KU 2 / 4: %.1f / %.1f us\n",
           run<2, 1>(d, rs, fs, dq, n_cand, out, junk, junk_n, 10), run<4, 1>(d, rs, fs, dq, n_cand, out, junk, junk_n, 10));
    return 0;
}
