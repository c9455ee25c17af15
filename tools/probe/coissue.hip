// Co-issue of LDS traffic and vector arithmetic on one SIMD (gfx950): what does a ds_write_b64 / ds_read_b64 cost a wave
// whose other instructions are v_fma_f32 -- spread between them, or as one burst that is waited for -- and what does a
// wave that only streams LDS writes cost a co-resident wave that only computes?  Replaces the additive model
// (VALU cycles + LDS cycles) of DESIGN.md 1.6 by measurements.  Times are shader-clock cycles (s_memtime) per wave, averaged.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/coissue.hip -o gpurun_out/coissue && gpurun_out/coissue
//
// A "group" is 16 v_fma_f32 (independent, 8 accumulators) + the LDS instructions of the variant; ITER groups per wave.
// Reported per variant and waves/SIMD k: cycles per group per wave, and per SIMD (= per wave / k: the throughput number).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITER = 2048;

#define FMA8 "v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n" \
             "v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n v_fma_f32 %7, %7, %9, %10\n"
#define ACC "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(rdv)
#define INS "v"(b0), "v"(b1), "v"(addr), "v"(val)
// %11 = LDS byte address, %12 = 64-bit value; reads land in %8
#define WR "ds_write_b64 %11, %12\n"
#define RD "ds_read_b64 %8, %11\n"
#define WR32 "ds_write_b32 %11, %9\n"
#define X8(x) x x x x x x x x

struct Res {
    unsigned long long cyc;
};

#define KERNEL(name, BODY, TAIL)                                                                              \
    __global__ __launch_bounds__(256) void name(unsigned long long *out, float seed, int iters) {              \
        extern __shared__ char lds[];                                                                        \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7, b0 = seed * 0.5f, b1 = seed * 0.25f;                                               \
        const unsigned addr = threadIdx.x * 8u;                                                              \
        const double val = seed;                                                                             \
        double rdv;                                                                                          \
        ((double *)lds)[threadIdx.x] = val;                                                                  \
        __syncthreads();                                                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < iters / 8; ++it) { /* 8 groups per trip: the loop branch costs ~30 cycles */   \
            asm volatile(X8(BODY) : ACC : INS : "memory");                                   \
        }                                                                                                    \
        asm volatile(TAIL ::: "memory");                                                                     \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 0;                                    \
    }

// 16 fma only
KERNEL(k_valu, FMA8 FMA8, "s_nop 0")
// spread: 8 fma, write, 8 fma, read   (the ratio of the column pass: 16 VALU per write + read)
KERNEL(k_spread_wr, FMA8 WR FMA8 RD, "s_waitcnt lgkmcnt(0)")
// spread: write only / read only
KERNEL(k_spread_w, FMA8 FMA8 WR, "s_waitcnt lgkmcnt(0)")
KERNEL(k_spread_r, FMA8 FMA8 RD, "s_waitcnt lgkmcnt(0)")
// two b32 writes instead of one b64
KERNEL(k_spread_w32x2, FMA8 WR32 FMA8 WR32, "s_waitcnt lgkmcnt(0)")
// LDS only
KERNEL(k_only_w, WR, "s_waitcnt lgkmcnt(0)")
KERNEL(k_only_r, RD, "s_waitcnt lgkmcnt(0)")
KERNEL(k_only_wr, WR RD, "s_waitcnt lgkmcnt(0)")

// bursts as the search kernels have them today: 16 groups of VALU, then 16 writes + 16 reads back to back, then wait
#define FMA16 FMA8 FMA8
#define X16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(256) void k_burst(unsigned long long *out, float seed, int iters) {
    extern __shared__ char lds[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7,
          b0 = seed * 0.5f, b1 = seed * 0.25f;
    const unsigned addr = threadIdx.x * 8u;
    const double val = seed;
    double rdv;
    ((double *)lds)[threadIdx.x] = val;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters / 16; ++it) {
        asm volatile(X16(FMA16) : ACC : INS : "memory");
        asm volatile(X16(WR) X16(RD) "s_waitcnt lgkmcnt(0)" : ACC : INS : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 0;
}
// the same burst, not waited for until the next burst's reads are needed (one group of VALU later)
__global__ __launch_bounds__(256) void k_burst_late(unsigned long long *out, float seed, int iters) {
    extern __shared__ char lds[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7,
          b0 = seed * 0.5f, b1 = seed * 0.25f;
    const unsigned addr = threadIdx.x * 8u;
    const double val = seed;
    double rdv;
    ((double *)lds)[threadIdx.x] = val;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters / 16; ++it) {
        asm volatile(X16(WR) X16(RD) : ACC : INS : "memory");
        asm volatile(X16(FMA16) "s_waitcnt lgkmcnt(0)" : ACC : INS : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 0;
}

// the same flops as "16fma" issued as 8 v_pk_fma_f32 (two FMAs per lane and instruction; 4 cycles each on the vector pipe)
#define PK8 "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
template <int MODE>
__global__ __launch_bounds__(256) void k_pk(unsigned long long *out, float seed, int iters) {
    extern __shared__ char lds[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b0 = {seed * 0.5f, seed * 0.5f}, b1 = {seed * 0.25f, seed * 0.25f};
    const unsigned addr = threadIdx.x * 8u;
    const double val = seed;
    double rdv;
    ((double *)lds)[threadIdx.x] = val;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters / 8; ++it) {
        if (MODE == 0)
            asm volatile(X8(PK8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "memory");
        else  // 4 pk, write, 4 pk, read
            asm volatile(X8("v_pk_fma_f32 %0, %0, %9, %10\n v_pk_fma_f32 %1, %1, %9, %10\n v_pk_fma_f32 %2, %2, %9, %10\n v_pk_fma_f32 %3, %3, %9, %10\n"
                            "ds_write_b64 %11, %12\n"
                            "v_pk_fma_f32 %4, %4, %9, %10\n v_pk_fma_f32 %5, %5, %9, %10\n v_pk_fma_f32 %6, %6, %9, %10\n v_pk_fma_f32 %7, %7, %9, %10\n"
                            "ds_read_b64 %8, %11\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(rdv)
                         : "v"(b0), "v"(b1), "v"(addr), "v"(val)
                         : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    const f2 sm = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (sm.x + sm.y == 1.2345f) out[0] = 0;
}

// role split: even workgroups compute only (16 fma per group), odd workgroups stream LDS only (one write + one read per
// group, or writes only); with 2 workgroups per CU every SIMD holds one wave of each kind
template <int MODE>
__global__ __launch_bounds__(256) void k_roles(unsigned long long *out, float seed, int iters, int rolebit) {
    extern __shared__ char lds[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7,
          b0 = seed * 0.5f, b1 = seed * 0.25f;
    const unsigned addr = threadIdx.x * 8u;
    const double val = seed;
    double rdv;
    ((double *)lds)[threadIdx.x] = val;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    // (blockIdx & 7 is the XCD; which bit separates the residents of one CU depends on the dispatcher: both are tried)
    if ((blockIdx.x >> rolebit) & 1) {
        for (int it = 0; it < iters / 8; ++it) {
            if (MODE == 0) asm volatile(X8(WR RD) : ACC : INS : "memory");
            if (MODE == 1) asm volatile(X8(WR) : ACC : INS : "memory");
            if (MODE == 2) asm volatile(X8(RD) : ACC : INS : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        for (int it = 0; it < iters / 8; ++it) asm volatile(X8(FMA16) : ACC : INS : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 0;
}

template <class K>
void run_roles(const char *name, K kern, unsigned long long *d_out) {
    std::vector<unsigned long long> h(256 * 8 * 4);
    for (int rolebit : {3, 8, 9})
        for (int k : {2, 4}) {
            const int grid = 256 * k;
            kern<<<grid, 256, 2048>>>(d_out, 1.0f, ITER, rolebit);
            kern<<<grid, 256, 2048>>>(d_out, 1.0f, ITER, rolebit);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
            double sc = 0, sl = 0;
            int nc = 0, nl = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < 4; ++w) {
                    if ((b >> rolebit) & 1) sl += (double)h[b * 4 + w], ++nl;
                    else sc += (double)h[b * 4 + w], ++nc;
                }
            printf("%-16s role bit %d waves/SIMD %d: compute waves %7.2f cycles per 16-fma group, LDS waves %7.2f per group\n", name,
                   rolebit, k, nc ? sc / nc / ITER : 0., nl ? sl / nl / ITER : 0.);
        }
}

template <class K>
void run(const char *name, K kern, unsigned long long *d_out) {
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int k : {1, 2, 3, 4, 6, 8}) {
        const int grid = 256 * k;
        kern<<<grid, 256, 2048>>>(d_out, 1.0f, ITER);
        hipEventRecord(e0);
        kern<<<grid, 256, 2048>>>(d_out, 1.0f, ITER);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
        s /= grid * 4.0 * ITER;
        printf("%-26s waves/SIMD %d: %7.2f cycles per group per wave, %7.2f per SIMD;  wall %7.2f ns per group per SIMD\n", name, k, s,
               s / k, ms * 1e6 / ((double)k * ITER));
    }
}

int main() {
    unsigned long long *d_out;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 8 * 4);
    run("16fma", k_valu, d_out);
    run("8pk_fma (= 16 fma)", k_pk<0>, d_out);
    run("8fma,w,8fma,r", k_spread_wr, d_out);
    run("4pk,w,4pk,r", k_pk<1>, d_out);
    run("16fma,w", k_spread_w, d_out);
    run("16fma,r", k_spread_r, d_out);
    run("8fma,w32,8fma,w32", k_spread_w32x2, d_out);
    run("w only", k_only_w, d_out);
    run("r only", k_only_r, d_out);
    run("w,r only", k_only_wr, d_out);
    run("burst 16x(16fma|w|r) wait", k_burst, d_out);
    run("burst, waited late", k_burst_late, d_out);
    run_roles("roles fma | w,r", k_roles<0>, d_out);
    run_roles("roles fma | w", k_roles<1>, d_out);
    run_roles("roles fma | r", k_roles<2>, d_out);
    return 0;
}
