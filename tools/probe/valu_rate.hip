// Issue rate of the instruction classes the search kernels are made of (gfx950), one class per kernel:
// every wave runs a long unrolled stream of independent instructions; full occupancy (8 waves/SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Prints cycles per wave-instruction per SIMD at the clock derived from s_memtime-free wall time
// (assumes the clock reported by hipDeviceProp; also prints G wave-instr/s so the ratio between
// classes is clock-independent).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 256  // outer loop trips; body = 64 instructions

#define VALU_KERNEL(name, asmline)                                                              \
    __global__ __launch_bounds__(256) void name(float *out, float seed) {                      \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                  \
        float b0 = seed * 0.5f, b1 = seed * 0.25f;                                              \
        float c0[2] = {a0, a1}, c1[2] = {a2, a3};                                               \
        (void)c0; (void)c1;                                                                     \
        for (int it = 0; it < ITER; ++it) {                                                     \
            REP16(asm volatile(asmline : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;                                \
    }

// each asm line = 4 independent instructions
VALU_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
VALU_KERNEL(k_add_f32, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
VALU_KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4")
VALU_KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5")
VALU_KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4")
VALU_KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_mul_f16 %3, %3, %4")
VALU_KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3")
VALU_KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %4\n v_cvt_pkrtz_f16_f32 %1, %1, %4\n v_cvt_pkrtz_f16_f32 %2, %2, %4\n v_cvt_pkrtz_f16_f32 %3, %3, %4")
VALU_KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3")
VALU_KERNEL(k_mov_b32, "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4")
VALU_KERNEL(k_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
VALU_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc")
VALU_KERNEL(k_max_f32, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4")
VALU_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
VALU_KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3")

VALU_KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]")
VALU_KERNEL(k_cmp_gt_f32, "v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4")
VALU_KERNEL(k_cmp_cnd_pair, "v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc")
VALU_KERNEL(k_max3_f32, "v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5")
VALU_KERNEL(k_fma_mix_f32, "v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[1,1,0]")
VALU_KERNEL(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f16_f32 %2, %2\n v_cvt_f16_f32 %3, %3")
VALU_KERNEL(k_pack_b32_f16, "v_pack_b32_f16 %0, %0, %4\n v_pack_b32_f16 %1, %1, %4\n v_pack_b32_f16 %2, %2, %4\n v_pack_b32_f16 %3, %3, %4")
VALU_KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5")
VALU_KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4")
VALU_KERNEL(k_and_b32, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4")
VALU_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
VALU_KERNEL(k_max_f32_e64, "v_max_f32_e64 %0, %0, %4\n v_max_f32_e64 %1, %1, %4\n v_max_f32_e64 %2, %2, %4\n v_max_f32_e64 %3, %3, %4")
VALU_KERNEL(k_fma_f32_neg, "v_fma_f32 %0, -%0, %4, %5\n v_fma_f32 %1, -%1, %4, %5\n v_fma_f32 %2, -%2, %4, %5\n v_fma_f32 %3, -%3, %4, %5")
VALU_KERNEL(k_sub_f32, "v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4")
VALU_KERNEL(k_fmac_f32, "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5")
VALU_KERNEL(k_rsq_f32, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3")

// packed f32: 64-bit register pairs
__global__ __launch_bounds__(256) void k_pk_fma_f32(float *out, float seed) {
    double a0, a1, a2, a3, b0, b1;  // just 64-bit containers
    float2 t = make_float2(seed + threadIdx.x, seed);
    memcpy(&a0, &t, 8), memcpy(&a1, &t, 8), memcpy(&a2, &t, 8), memcpy(&a3, &t, 8), memcpy(&b0, &t, 8), memcpy(&b1, &t, 8);
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(256) void k_pk_add_f32(float *out, float seed) {
    double a0, a1, a2, a3, b0;
    float2 t = make_float2(seed + threadIdx.x, seed);
    memcpy(&a0, &t, 8), memcpy(&a1, &t, 8), memcpy(&a2, &t, 8), memcpy(&a3, &t, 8), memcpy(&b0, &t, 8);
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(256) void k_pk_mul_f32(float *out, float seed) {
    double a0, a1, a2, a3, b0;
    float2 t = make_float2(seed + threadIdx.x, seed);
    memcpy(&a0, &t, 8), memcpy(&a1, &t, 8), memcpy(&a2, &t, 8), memcpy(&a3, &t, 8), memcpy(&b0, &t, 8);
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}

// ---- LDS: conflict-free unit-stride accesses ---------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f4;
#define LDS_PROLOGUE(bytes)                                                                     \
    __shared__ __attribute__((aligned(16))) char lds[256 * bytes * 4];                          \
    unsigned addr = threadIdx.x * bytes;                                                        \
    for (int i = threadIdx.x; i < 256 * bytes; i += 256) ((float *)lds)[i] = seed;             \
    __syncthreads();
#define LDS_EPILOGUE                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
    out[blockIdx.x * 256 + threadIdx.x] = ((float *)lds)[threadIdx.x];
__global__ __launch_bounds__(256) void k_ds_read_b32(float *out, float seed) {
    LDS_PROLOGUE(4)
    unsigned r0, r1, r2, r3;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072"
                           : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_read_b64(float *out, float seed) {
    LDS_PROLOGUE(8)
    double r0, r1, r2, r3;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144"
                           : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_read_b128(float *out, float seed) {
    LDS_PROLOGUE(16)
    f4 r0, r1, r2, r3;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288"
                           : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_write_b32(float *out, float seed) {
    LDS_PROLOGUE(4)
    unsigned r0 = threadIdx.x;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %1 offset:3072"
                           : : "v"(addr), "v"(r0) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_write2_b32(float *out, float seed) {
    LDS_PROLOGUE(8)
    unsigned r0 = threadIdx.x;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_write2_b32 %0, %1, %1 offset0:0 offset1:1\n ds_write2_b32 %0, %1, %1 offset0:128 offset1:129\n ds_write2_b32 %0, %1, %1 offset0:64 offset1:65\n ds_write2_b32 %0, %1, %1 offset0:192 offset1:193"
                           : : "v"(addr), "v"(r0) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_write_b64(float *out, float seed) {
    LDS_PROLOGUE(8)
    double r0 = threadIdx.x;
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:6144"
                           : : "v"(addr), "v"(r0) : "memory");)
    }
    LDS_EPILOGUE
}
__global__ __launch_bounds__(256) void k_ds_write_b128(float *out, float seed) {
    LDS_PROLOGUE(16)
    f4 r0 = {seed, seed, seed, seed};
    for (int it = 0; it < ITER / 4; ++it) {
        REP16(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288"
                           : : "v"(addr), "v"(r0) : "memory");)
    }
    LDS_EPILOGUE
}

template <class K>
static void run(const char *name, K kern, float *out, double inst_per_thread, int bytes_per_inst) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int grid = 256 * 8 * 4;  // 8 blocks of 256 threads per CU resident (8 waves/SIMD), 4 rounds
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
    const double waves = (double)grid * 4;
    const double winst = waves * inst_per_thread;
    const double per_simd_per_s = winst / (best * 1e-3) / 1024.0;
    printf("%-16s %8.3f ms  %7.2f G wave-inst/s/chip  -> %5.2f cyc/inst/SIMD @2.4GHz (%5.2f @2.0GHz)", name, best,
           winst / (best * 1e-3) / 1e9, 2.4e9 / per_simd_per_s, 2.0e9 / per_simd_per_s);
    if (bytes_per_inst) printf("   %6.1f TB/s chip, %5.1f B/clk/CU @2.4GHz", winst * 64 * bytes_per_inst / (best * 1e-3) / 1e12,
                               winst * 64 * bytes_per_inst / (best * 1e-3) / 256 / 2.4e9);
    printf("\n");
}

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 4 * 256);
    const double n = 64.0 * ITER;
#define R(k) run(#k, k, out, n, 0)
    R(k_fma_f32);
    R(k_add_f32);
    R(k_mul_f32);
    R(k_pk_fma_f32);
    R(k_pk_add_f32);
    R(k_pk_mul_f32);
    R(k_pk_fma_f16);
    R(k_pk_add_f16);
    R(k_pk_mul_f16);
    R(k_dot2_f32_f16);
    R(k_cvt_f32_f16);
    R(k_cvt_pkrtz);
    R(k_sqrt_f32);
    R(k_mov_b32);
    R(k_add_u32);
    R(k_mad_u32_u24);
    R(k_cndmask);
    R(k_max_f32);
    R(k_cndmask_sgpr);
    R(k_cmp_gt_f32);
    R(k_cmp_cnd_pair);
    R(k_max3_f32);
    R(k_max_f32_e64);
    R(k_fma_mix_f32);
    R(k_cvt_f16_f32);
    R(k_pack_b32_f16);
    R(k_perm_b32);
    R(k_lshl_add_u32);
    R(k_and_b32);
    R(k_mul_lo_u32);
    R(k_fma_f32_neg);
    R(k_sub_f32);
    R(k_fmac_f32);
    R(k_rsq_f32);
    const double nl = 64.0 * ITER / 4;
    run("ds_read_b32", k_ds_read_b32, out, nl, 4);
    run("ds_read_b64", k_ds_read_b64, out, nl, 8);
    run("ds_read_b128", k_ds_read_b128, out, nl, 16);
    run("ds_write_b32", k_ds_write_b32, out, nl, 4);
    run("ds_write2_b32", k_ds_write2_b32, out, nl, 8);
    run("ds_write_b64", k_ds_write_b64, out, nl, 8);
    run("ds_write_b128", k_ds_write_b128, out, nl, 16);
    return 0;
}
