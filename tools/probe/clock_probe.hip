// Engine clock under load: a grid of busy workgroups times itself with the shader clock (s_memtime) against the constant
// reference clock (s_memrealtime, hipDeviceAttributeWallClockRate).  Prints GHz for a light and a heavy (fp32 FMA on all SIMDs) load.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void k_busy(unsigned long long *acc, float *sink, int iters) {
    const long long c0 = clock64(), r0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a = fmaf(a, b, c);
            c = fmaf(c, b, d);
            d = fmaf(d, b, a);
        }
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    if (a + c + d == 123.f) sink[0] = a;
    if (threadIdx.x == 0) {
        atomicAdd(acc, (unsigned long long)(c1 - c0));
        atomicAdd(acc + 1, (unsigned long long)(r1 - r0));
    }
}

int main() {
    unsigned long long *d_acc, h[2];
    float *d_sink;
    (void)hipMalloc(&d_acc, 16);
    (void)hipMalloc(&d_sink, 4);
    int wall_khz = 0, clk_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, reported engine clock %d kHz\n", wall_khz, clk_khz);
    for (int blocks : {8, 256, 2048, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(d_acc, 0, 16);
            hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, 0, d_acc, d_sink, 200000);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, d_acc, 16, hipMemcpyDeviceToHost);
            printf("%5d workgroups: shader clocks / wall clocks = %.4f -> %.3f GHz\n", blocks, (double)h[0] / (double)h[1],
                   (double)h[0] / (double)h[1] * wall_khz * 1e-6);
        }
    }
    return 0;
}
