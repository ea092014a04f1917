// valubench.hip -- what one SIMD of gfx950 issues per cycle for the integer vector instructions the pipeline kernels are made
// of, by waves per SIMD (tuning aid, not part of the library).  Every wave runs ITER rounds of 8 independent chains of one
// instruction; the grid places W waves on every SIMD (256 CUs x 4 SIMDs x W one-wave workgroups, occupancy capped by LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k_valu(uint32_t* out, int iters, uint32_t seed) {
    extern __shared__ uint8_t lds[];
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 77u;
    const uint32_t b = seed ^ 0x01020304u, c = 0x05010400u;
    uint32_t sg = 0; uint64_t d[4] = {seed * 3ull, seed * 5ull, seed * 7ull, seed * 11ull}; const uint64_t mask64 = 0x5555AAAA3333CCCCull ^ seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 3) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 5) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 6) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 9) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 10) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 12) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 13) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 14) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 15) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 16) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 18) asm volatile("v_cndmask_b32 %0, %0, %1, %3" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 19) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 20) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 21) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 22) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 23) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 25) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 26) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 27) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 28) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 29) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 30) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 31) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 33) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 34) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 35) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 36) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 37) asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 38) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c), "s"(mask64));
                if (OP == 39) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i & 3]));
                if (OP == 40) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i & 3]));
                if (OP == 41) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[i & 3]));
                if (OP == 42) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(d[i & 3]));
                if (OP == 43) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i]));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a[i];
    if ((s ^ sg ^ (uint32_t)(d[0] ^ d[1] ^ d[2] ^ d[3])) == 0x12345678u) out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int OP>
double run(int waves_per_simd, uint32_t* out, int iters) {
    // occupancy cap through LDS: 160 KB per CU / (4 * W) workgroups
    const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / (4 * waves_per_simd)) - 512;
    dim3 grid(256 * 4 * waves_per_simd), block(64);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_valu<OP>, grid, block, lds, 0, out, iters, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_valu<OP>, grid, block, lds, 0, out, iters, 12345u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double insts_per_simd = (double)waves_per_simd * iters * 64;
    return ms * 1e-3 / insts_per_simd * 1e9;   // ns per wave-instruction per SIMD
}

int main() {
    uint32_t* out; CK(hipMalloc(&out, 64 << 20));
    const char* names[] = {"v_xor_b32", "v_add_u32", "v_perm_b32", "v_pk_add_u16", "v_dot4_u32_u8", "v_sad_u8", "v_mul_hi_u32_u24", "v_mul_lo_u32", "v_fma_f32", "v_mov_b32_dpp", "v_bfe_u32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_min_u32", "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_u32 (vcc)", "v_bcnt_u32_b32", "v_bfi_b32", "v_add3_u32", "v_or3_b32", "v_and_or_b32", "v_lshl_or_b32", "v_lshl_add_u32", "v_alignbit_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_f32", "v_add_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_rcp_f32", "v_med3_u32", "v_add_u32_dpp", "v_lshrrev_b32_sdwa", "v_pk_mad_u16", "v_add_f64", "v_fma_f64", "v_lshlrev_b64", "v_lshl_add_u64", "v_readlane_b32"};
    printf("ns per wave64 instruction per SIMD (every SIMD of the chip busy; at 2.4 GHz 1 cycle = 0.417 ns)\n%-18s", "waves per SIMD");
    for (int w : {1, 2, 4, 8}) printf("%8d", w);
    printf("\n");
#define ROW(OP) { printf("%-18s", names[OP]); for (int w : {1, 2, 4, 8}) printf("%8.3f", run<OP>(w, out, 2000)); printf("\n"); fflush(stdout); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31) ROW(32) ROW(33) ROW(34) ROW(35) ROW(36) ROW(37) ROW(38) ROW(39) ROW(40) ROW(41) ROW(42) ROW(43)
    return 0;
}
