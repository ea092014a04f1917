// readbench.hip -- what a streaming read of K1's shape can reach on this GPU (tuning aid, not part of the library).
//   pattern 0: K1's: lane l reads 48 contiguous bytes per row (3 x dwordx4 at l*48 + {0,16,32})
//   pattern 1: each instruction contiguous: 3 x dwordx4 at i*1024 + l*16
//   pattern 2: dwordx3 per lane, 4 instructions, each contiguous (768 B per instruction)
// Every wave walks `rows` rows of a strip of 3072 bytes, PF rows of loads in flight; occupancy is capped with LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// HALO > 0: like K1, every wave also reads HALO rows above and below its strip (clamped to the frame): 14 more rows per 270
template <int PAT, int PF, int WORK, int STORE, int HALO = 0>
__global__ __launch_bounds__(64) void k_read(const uint8_t* __restrict__ base, size_t row_stride, int rows, int strips_x, int strips_y, uint32_t* __restrict__ out, uint16_t* __restrict__ bits) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int pair = (k / strips_y) * 8 + xcd, sy = k % strips_y;
    const int sx = pair % strips_x, f = pair / strips_x;
    const uint8_t* p = base + (size_t)f * row_stride * (size_t)(rows * strips_y) + (size_t)sy * rows * row_stride + (size_t)sx * 2976;
    if (HALO) {   // the halo rows: 2 * HALO clamped row reads before the strip's own (no stores for them)
        const int H = rows * strips_y;
        uint32_t hacc = 0;
        for (int h = 0; h < 2 * HALO; h++) {
            const int y = sy * rows + (h < HALO ? -HALO + h : rows + h - HALO);
            const uint8_t* rp = base + (size_t)f * row_stride * (size_t)H + (size_t)min(max(y, 0), H - 1) * row_stride + (size_t)sx * 2976;
            const uint4* v = reinterpret_cast<const uint4*>(rp + lane * 48);
            const uint4 a = v[0], b = v[1], c = v[2];
            hacc += a.x ^ a.w ^ b.y ^ c.z;
        }
        if (hacc == 0x12345678u) out[blockIdx.x * 64 + lane] = hacc;
    }
    uint32_t acc = 0;
    uint4 q[PF][3];
    auto issue = [&](int r, uint4* d) {
        const uint8_t* rp = p + (size_t)r * row_stride;
        if (PAT == 0) { const uint4* v = reinterpret_cast<const uint4*>(rp + lane * 48); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
        else if (PAT == 3) {   // K1's pattern with non-temporal loads (streaming: the frames should not push the dirty result lines out of L2)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* v = reinterpret_cast<const u32x4*>(rp + lane * 48);
            const u32x4 a = __builtin_nontemporal_load(v), b = __builtin_nontemporal_load(v + 1), c = __builtin_nontemporal_load(v + 2);
            d[0] = make_uint4(a.x, a.y, a.z, a.w); d[1] = make_uint4(b.x, b.y, b.z, b.w); d[2] = make_uint4(c.x, c.y, c.z, c.w);
        }
        else if (PAT == 1) { const uint4* v = reinterpret_cast<const uint4*>(rp + lane * 16); d[0] = v[0]; d[1] = v[64]; d[2] = v[128]; }
        else {
            const uint32_t* v = reinterpret_cast<const uint32_t*>(rp + lane * 12);
            d[0] = make_uint4(v[0], v[1], v[2], v[192]); d[1] = make_uint4(v[193], v[194], v[384], v[385]); d[2] = make_uint4(v[386], v[576], v[577], v[578]);
        }
    };
#pragma unroll
    for (int i = 0; i < PF; i++) issue(i, q[i]);
    for (int r = 0; r < rows; r += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            uint32_t t = q[i][0].x ^ q[i][0].w ^ q[i][1].y ^ q[i][2].z ^ q[i][2].w ^ q[i][1].x ^ q[i][0].y ^ q[i][0].z ^ q[i][1].z ^ q[i][1].w ^ q[i][2].x ^ q[i][2].y;
            uint32_t u = t * 3u, v = t + 7u, w = t ^ 0x55u;
#pragma unroll
            for (int k = 0; k < WORK / 4; k++) { t = t * 5u + u; u = u * 3u + v; v = (v >> 3) + w; w = w * 7u + t; }   // four independent-ish chains
            acc += t ^ u ^ v ^ w;
            if (STORE == 1) bits[((size_t)blockIdx.x * rows + r + i) * 64 + lane] = (uint16_t)acc;
            if (STORE == 5 && ((r + i) & 7) == 7) reinterpret_cast<uint4*>(bits + ((size_t)blockIdx.x * rows + r + i - 7) * 64)[lane] = make_uint4(acc, t, u, v);
            if (STORE == 7 && ((r + i) & 63) == 63) reinterpret_cast<uint4*>(bits + ((size_t)blockIdx.x * rows + r + i - 63) * 64)[lane] = make_uint4(acc, t, u, v);
            if (STORE == 8 && ((r + i) & 7) == 7) __builtin_nontemporal_store(make_uint4(acc, t, u, v).x, reinterpret_cast<uint32_t*>(bits + ((size_t)blockIdx.x * rows + r + i - 7) * 64) + lane);
            if (STORE == 9 && ((r + i) & 7) == 7) {
                uint4* dst = reinterpret_cast<uint4*>(bits + ((size_t)blockIdx.x * rows + r + i - 7) * 64) + lane;
                __builtin_nontemporal_store(acc, &dst->x); __builtin_nontemporal_store(t, &dst->y); __builtin_nontemporal_store(u, &dst->z); __builtin_nontemporal_store(v, &dst->w);
            }
            if (STORE == 10 && ((r + i) & 7) == 7) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(bits + ((size_t)blockIdx.x * rows + r + i - 7) * 64) + lane * 4;
                __hip_atomic_store(dst, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(dst + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 2, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(dst + 3, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (STORE == 12) {   // sparse output: an 8-byte flag word per row from one lane, the 128 data bytes for one row in ten
                unsigned long long* fl = reinterpret_cast<unsigned long long*>(bits + (size_t)gridDim.x * rows * 64) + (size_t)blockIdx.x * rows + r + i;
                if (lane == 0) *fl = acc;
                if (((r + i) % 10) == 0) bits[((size_t)blockIdx.x * rows + r + i) * 64 + lane] = (uint16_t)acc;
            }
            if (STORE == 13) {   // the same with the flag words of 8 rows in one 64-byte store
                if (((r + i) & 7) == 7 && lane < 8) reinterpret_cast<unsigned long long*>(bits + (size_t)gridDim.x * rows * 64)[(size_t)blockIdx.x * rows + r + i - 7 + lane] = acc;
                if (((r + i) % 10) == 0) bits[((size_t)blockIdx.x * rows + r + i) * 64 + lane] = (uint16_t)acc;
            }
            if (STORE >= 600) {   // parking only: the result bits go to LDS and are read back in bursts, but never stored
                constexpr int PER = STORE - 600;
                const int rr = r + i;
                reinterpret_cast<uint16_t*>(lds)[(rr % PER) * 64 + lane] = (uint16_t)acc;
                if (rr % PER == PER - 1 || rr == rows - 1) {
                    const int first = rr - rr % PER, cnt = rr - first + 1;
                    for (int q = 0; q < cnt; q += 8) {
                        uint4 v = reinterpret_cast<const uint4*>(lds + (size_t)q * 128)[lane];
                        acc ^= v.x ^ v.y ^ v.z ^ v.w;
                    }
                }
            } else
            if (STORE >= 400) {   // the same bursts, destination wrapped into a 4 MB window: the stores hit in the L2s and nothing is written back
                constexpr int PER = STORE - 400;
                const int rr = r + i;
                reinterpret_cast<uint16_t*>(lds)[(rr % PER) * 64 + lane] = (uint16_t)acc;
                if (rr % PER == PER - 1 || rr == rows - 1) {
                    const int first = rr - rr % PER, cnt = rr - first + 1;
                    for (int q = 0; q < cnt; q += 8) {
                        uint4 v = reinterpret_cast<const uint4*>(lds + (size_t)q * 128)[lane];
                        reinterpret_cast<uint4*>(bits + ((((size_t)blockIdx.x * rows + first + q) * 64) & ((2u << 20) - 1u)))[lane] = v;
                    }
                }
            } else
            if (STORE >= 100) {
                constexpr int PER = STORE - 100;
                const int rr = r + i;
                reinterpret_cast<uint16_t*>(lds)[(rr % PER) * 64 + lane] = (uint16_t)acc;
                if (rr % PER == PER - 1 || rr == rows - 1) {
                    const int first = rr - rr % PER, cnt = rr - first + 1;
                    for (int q = 0; q < cnt; q += 8) {
                        uint4 v = reinterpret_cast<const uint4*>(lds + (size_t)q * 128)[lane];
                        reinterpret_cast<uint4*>(bits + ((size_t)blockIdx.x * rows + first + q) * 64)[lane] = v;
                    }
                }
            }
            if (STORE == 11) reinterpret_cast<uint16_t*>(lds)[(r + i) * 64 + lane] = (uint16_t)acc;
            if (STORE == 6 && ((r + i) & 3) == 3) reinterpret_cast<uint2*>(bits + ((size_t)blockIdx.x * rows + r + i - 3) * 64)[lane] = make_uint2(acc, t);
            if (STORE >= 2 && STORE <= 4) {   // gather 2 / 4 / 8 lanes' 16 bits into one dword / dwordx2 / dwordx4 store
                uint16_t* rowp = bits + ((size_t)blockIdx.x * rows + r + i) * 64;
                const uint32_t x = acc & 0xFFFFu;
                const uint32_t t = x | (__shfl_down(x, 1) << 16);
                if (STORE == 2) { if ((lane & 1) == 0) *reinterpret_cast<uint32_t*>(rowp + lane) = t; }
                else {
                    const uint32_t t2 = __shfl_down(t, 2);
                    if (STORE == 3) { if ((lane & 3) == 0) *reinterpret_cast<uint2*>(rowp + lane) = make_uint2(t, t2); }
                    else {
                        const uint32_t t4 = __shfl_down(t, 4), t6 = __shfl_down(t2, 4);
                        if ((lane & 7) == 0) *reinterpret_cast<uint4*>(rowp + lane) = make_uint4(t, t2, t4, t6);
                    }
                }
            }
            const int nr = r + i + PF;
            issue(nr < rows ? nr : rows - 1, q[i]);
        }
    }
    if (STORE == 11) {   // everything the wave produced leaves in one burst at the end of its life
        for (int r = 0; r < rows; r += 8) {
            uint4 v = reinterpret_cast<const uint4*>(lds + (size_t)r * 128)[lane];
            reinterpret_cast<uint4*>(bits + ((size_t)blockIdx.x * rows + r) * 64)[lane] = v;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * 64 + lane] = acc + lds[lane];
}

template <int PAT, int PF, int WORK, int STORE, int HALO = 0>
float run(const uint8_t* d, size_t row_stride, int frames, int H, int strips_y, size_t lds, uint32_t* out, int reps) {
    uint16_t* bits = reinterpret_cast<uint16_t*>(out + (8 << 20));
    const int strips_x = 2, rows = H / strips_y;
    const int n_pairs = frames * strips_x;
    dim3 grid(8 * ((n_pairs + 7) / 8) * strips_y), block(64);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_read<PAT, PF, WORK, STORE, HALO>), grid, block, lds, 0, d, row_stride, rows, strips_x, strips_y, out, bits);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read<PAT, PF, WORK, STORE, HALO>), grid, block, lds, 0, d, row_stride, rows, strips_x, strips_y, out, bits);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x;
    }
}

int main(int argc, char** argv) {
    const bool quick = argc > 1;   // any argument: skip the store-width sweep, print the K1-shape lines only
    const int frames = 256, W = 1920, H = 1080;
    const size_t row_stride = (size_t)W * 3, bytes = row_stride * H * frames;
    uint8_t* d; uint32_t* out;
    CK(hipMalloc(&d, bytes + 65536)); CK(hipMalloc(&out, 256 << 20));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(d), (bytes + 65536) / 4);
    CK(hipDeviceSynchronize());
    const double gb = (double)frames * H * 2 * 3072 / 1e9;
    printf("random data, %.2f GB requested per launch; ms per launch\n", gb);
    for (int occ : {2}) {
        if (quick) break;
        const size_t lds = occ == 8 ? 0 : (occ == 4 ? 9000 : 19000);
        for (int sy : {4, 8}) {
            printf("occ %d strips_y %d: read-only %.3f | short x64 %.3f | dword x32 %.3f | dwordx2 x16 %.3f | dwordx4 x8 %.3f | 1 KB every 8 rows %.3f | 512 B every 4 rows %.3f | 1 KB every 64 rows %.3f | nt dword every 8 rows %.3f | nt 1 KB / 8 rows %.3f | sc1 1 KB / 8 rows %.3f | LDS-buffered, burst at wave end %.3f | sparse: flags/row + 10%% rows %.3f | sparse, flags per 8 rows %.3f | LDS flush every 32 rows %.3f | 64 rows %.3f | 96 rows %.3f | 136 rows %.3f\n", occ, sy,
                   run<0, 3, 0, 0>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 1>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 2>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 3>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 4>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 5>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 6>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 7>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 8>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 9>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 10>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 11>(d, row_stride, frames, H, sy, (size_t)(H / sy + 8) * 128 > lds ? (size_t)(H / sy + 8) * 128 : lds, out, 10),
                   run<0, 3, 0, 12>(d, row_stride, frames, H, sy, lds, out, 10), run<0, 3, 0, 13>(d, row_stride, frames, H, sy, lds, out, 10),
                   run<0, 3, 0, 132>(d, row_stride, frames, H, sy, 19000, out, 10), run<0, 3, 0, 164>(d, row_stride, frames, H, sy, 19000, out, 10),
                   run<0, 3, 0, 196>(d, row_stride, frames, H, sy, 19000, out, 10), run<0, 3, 0, 236>(d, row_stride, frames, H, sy, 19000, out, 10));
            fflush(stdout);
        }
    }
    // K1's own shape, arithmetic stripped: 4 strips of 270 rows per frame column, 14 halo rows per strip, result bits parked in LDS
    // and written in bursts of 128 rows; beside it the same without the halo and without the stores
    printf("K1 shape (occ 2, strips_y 4): reads only %.3f | + 14 halo rows %.3f | + halo + stores in bursts of 128 rows %.3f | no halo, bursts of 128 rows %.3f\n",
           run<0, 3, 0, 0>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 3, 0, 0, 7>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<0, 3, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 3, 0, 228>(d, row_stride, frames, H, 4, 19000, out, 20));
    // where does the price of the stores arise?  the same bursts into a 4 MB window (they stay in the L2s: no write-back traffic) against the real destination
    printf("K1 shape, halo + bursts of 128 rows: to the 66 MB image %.3f | into a 4 MB window (L2-resident) %.3f | parked in LDS and read back, never stored %.3f | reads + halo only %.3f\n",
           run<0, 3, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 3, 0, 528, 7>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<0, 3, 0, 728, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 3, 0, 0, 7>(d, row_stride, frames, H, 4, 19000, out, 20));
    printf("the same with the LDS allocation but no stores at all (occupancy and LDS carve-out as K1's): %.3f ; with 64 KB of LDS per workgroup (one wave per SIMD... two per CU): %.3f\n",
           run<0, 3, 0, 0, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 3, 0, 0, 7>(d, row_stride, frames, H, 4, 64000, out, 20));
    // more bytes in flight per SIMD: deeper load queues at two waves per SIMD, three waves per SIMD (LDS cap 13000: bursts of 64 rows)
    printf("K1 shape, halo + bursts of 128 rows, load queue depth 3 / 5 / 6 / 9 rows: %.3f | %.3f | %.3f | %.3f ; reads only, depth 5 / 9: %.3f | %.3f\n",
           run<0, 3, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 5, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<0, 6, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 9, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<0, 5, 0, 0>(d, row_stride, frames, H, 4, 19000, out, 20), run<0, 9, 0, 0>(d, row_stride, frames, H, 4, 19000, out, 20));
    printf("three waves per SIMD (6 strips of 180 rows), halo + bursts of 64 rows, depth 3 / 5: %.3f | %.3f ; reads only: %.3f\n",
           run<0, 3, 0, 164, 7>(d, row_stride, frames, H, 6, 13000, out, 20), run<0, 5, 0, 164, 7>(d, row_stride, frames, H, 6, 13000, out, 20),
           run<0, 3, 0, 0>(d, row_stride, frames, H, 6, 13000, out, 20));
    printf("the same with NON-TEMPORAL loads: reads only %.3f | + halo %.3f | + halo + stores in bursts of 128 rows %.3f | no halo, bursts of 128 rows %.3f | row-by-row stores %.3f\n",
           run<3, 3, 0, 0>(d, row_stride, frames, H, 4, 19000, out, 20), run<3, 3, 0, 0, 7>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<3, 3, 0, 228, 7>(d, row_stride, frames, H, 4, 19000, out, 20), run<3, 3, 0, 228>(d, row_stride, frames, H, 4, 19000, out, 20),
           run<3, 3, 0, 1, 7>(d, row_stride, frames, H, 4, 19000, out, 20));
    return 0;
}
