// Issue rate of the instruction classes the tracking kernels are made of (gfx950): f64 arithmetic and conversions,
// cross-lane moves, byte extraction.  Same harness as valu_rate.hip: long unrolled streams of independent
// instructions at 8 waves per SIMD; prints SIMD cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/f64_rate.hip -o tools/probe/f64_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 128

// d: four f64 accumulators, two f64 sources
#define D_KERNEL(name, asmline)                                                                       \
    __global__ __launch_bounds__(256) void name(float *out, float seed) {                            \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = seed * 0.5, b1 = seed * 0.25; \
        for (int it = 0; it < ITER; ++it) {                                                           \
            REP16(asm volatile(asmline : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) \
        }                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                             \
    }
// m: f64 sources -> 32-bit results (or the reverse): four 32-bit registers r, four f64 registers a
#define M_KERNEL(name, asmline)                                                                       \
    __global__ __launch_bounds__(256) void name(float *out, float seed) {                            \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                        \
        int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;                                  \
        for (int it = 0; it < ITER; ++it) {                                                           \
            REP16(asm volatile(asmline : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));) \
        }                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3) + r0 + r1 + r2 + r3;        \
    }

D_KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5")
D_KERNEL(k_add_f64, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4")
D_KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4")
D_KERNEL(k_ceil_f64, "v_ceil_f64 %0, %0\n v_ceil_f64 %1, %1\n v_ceil_f64 %2, %2\n v_ceil_f64 %3, %3")
D_KERNEL(k_floor_f64, "v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3")
D_KERNEL(k_fract_f64, "v_fract_f64 %0, %0\n v_fract_f64 %1, %1\n v_fract_f64 %2, %2\n v_fract_f64 %3, %3")
D_KERNEL(k_rcp_f64, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
D_KERNEL(k_max_f64, "v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4")
D_KERNEL(k_cmp_gt_f64, "v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4")
D_KERNEL(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1\n v_ldexp_f64 %1, %1, 1\n v_ldexp_f64 %2, %2, 1\n v_ldexp_f64 %3, %3, 1")
D_KERNEL(k_mov_b64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4")
M_KERNEL(k_cvt_i32_f64, "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3")
M_KERNEL(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7")
M_KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3")
M_KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7")
M_KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7")
M_KERNEL(k_cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte1 %6, %6\n v_cvt_f32_ubyte1 %7, %7")
M_KERNEL(k_bfe_i32, "v_bfe_i32 %4, %4, 8, 8\n v_bfe_i32 %5, %5, 8, 8\n v_bfe_i32 %6, %6, 8, 8\n v_bfe_i32 %7, %7, 8, 8")
M_KERNEL(k_alignbyte, "v_alignbyte_b32 %4, %4, %5, 1\n v_alignbyte_b32 %5, %5, %6, 1\n v_alignbyte_b32 %6, %6, %7, 1\n v_alignbyte_b32 %7, %7, %4, 1")
M_KERNEL(k_sin_f32, "v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7")
M_KERNEL(k_mov_dpp_shr, "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 row_shr:1 row_mask:0xf bank_mask:0xf")
M_KERNEL(k_bpermute, "ds_bpermute_b32 %4, %5, %4\n ds_bpermute_b32 %5, %6, %5\n ds_bpermute_b32 %6, %7, %6\n ds_bpermute_b32 %7, %4, %7\n s_waitcnt lgkmcnt(0)")
M_KERNEL(k_readlane, "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3")

template <class K>
static void run(const char *name, K kern, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int grid = 256 * 8 * 4;
    kern<<<grid, 256>>>(out, 1.0f);
    float best = 1e9;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        kern<<<grid, 256>>>(out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double winst = (double)grid * 4 * 64.0 * ITER;
    const double per_simd_per_s = winst / (best * 1e-3) / 1024.0;
    printf("%-18s %8.3f ms  -> %5.2f cyc/inst/SIMD @2.4GHz (%5.2f @2.0GHz)\n", name, best, 2.4e9 / per_simd_per_s, 2.0e9 / per_simd_per_s);
}

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 4 * 256);
#define R(k) run(#k, k, out)
    R(k_fma_f64); R(k_add_f64); R(k_mul_f64); R(k_ceil_f64); R(k_floor_f64); R(k_fract_f64); R(k_rcp_f64); R(k_max_f64);
    R(k_cmp_gt_f64); R(k_ldexp_f64); R(k_mov_b64); R(k_cvt_i32_f64); R(k_cvt_f64_i32); R(k_cvt_f32_f64); R(k_cvt_f64_f32);
    R(k_cvt_f32_i32); R(k_cvt_f32_ubyte1); R(k_bfe_i32); R(k_alignbyte); R(k_sin_f32); R(k_mov_dpp_shr); R(k_bpermute); R(k_readlane);
    return 0;
}
