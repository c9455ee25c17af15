// VALU issue rate of one SIMD as a function of the waves resident on it (gfx950): grid = 256 CUs x k workgroups of 256
// threads (one wave per SIMD each), every thread runs the same number of independent or dependent v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/issue_rate.hip -o tools/probe/issue_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 512
__global__ __launch_bounds__(256) void k_indep(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = seed * 0.5f, b1 = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_dep(float *out, float seed) {
    float a0 = seed + threadIdx.x, b0 = seed * 0.5f, b1 = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0;
}
__global__ __launch_bounds__(256) void k_dep2(float *out, float seed) {  // two interleaved dependent chains
    float a0 = seed + threadIdx.x, a1 = a0 + 1, b0 = seed * 0.5f, b1 = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1;
}
__global__ __launch_bounds__(256) void k_pk(float *out, float seed) {
    double a0, a1, a2, a3, b0, b1;
    float2 t = make_float2(seed + threadIdx.x, seed);
    __builtin_memcpy(&a0, &t, 8), __builtin_memcpy(&a1, &t, 8), __builtin_memcpy(&a2, &t, 8), __builtin_memcpy(&a3, &t, 8), __builtin_memcpy(&b0, &t, 8), __builtin_memcpy(&b1, &t, 8);
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(256) void k_mix(float *out, float seed) {  // 3 fast ops + 1 slow (v_max_f32) per group
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = seed * 0.5f, b1 = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_lds(float *out, float seed) {  // 3 fma + 1 ds_read_b64 per group
    __shared__ double lds[512];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed;
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, b0 = seed * 0.5f, b1 = seed * 0.25f;
    double d; unsigned addr = threadIdx.x * 8;
    for (int it = 0; it < ITER; ++it) {
        REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n ds_read_b64 %3, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "=v"(d) : "v"(b0), "v"(b1), "v"(addr) : "memory");)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + (float)d;
}
template <class K> void run(const char *name, K kern, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k : {1, 2, 3, 4, 6, 8}) {
        const int grid = 256 * k;
        kern<<<grid, 256>>>(out, 1.0f);
        float best = 1e9;
        for (int i = 0; i < 5; ++i) {
            hipEventRecord(e0); kern<<<grid, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double inst = 64.0 * ITER;  // per wave
        // every SIMD holds k waves for the whole run: cycles per instruction per SIMD = time * clk / (k * inst)
        printf("%-8s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.1GHz (per wave: one instruction every %.2f cycles)\n",
               name, k, best, best * 1e-3 * 2.1e9 / (k * inst), best * 1e-3 * 2.1e9 / inst);
    }
}
int main() {
    float *out; hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    run("indep4", k_indep, out); run("pk_fma", k_pk, out); run("mix3+max", k_mix, out); run("3fma+ds", k_lds, out);
    return 0;
}
