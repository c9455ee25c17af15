// Write -> read-back bandwidth around the 256 MB Infinity Cache (MALL) of MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mall_probe.hip -o gpurun_out/mall_probe && gpurun_out/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_write(uint4 *p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}
__global__ __launch_bounds__(256) void k_read(const uint4 *p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t maxb = 4ull << 30;
    uint4 *buf;
    unsigned *out;
    hipMalloc(&buf, maxb);
    hipMalloc(&out, 4);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreate(&e2);
    const int grid = 256 * 16;
    for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t n = mb * (1ull << 20) / 16;
        float bw = 1e9, br = 1e9;
        for (int it = 0; it < 10; ++it) {
            hipEventRecord(e0);
            k_write<<<grid, 256>>>(buf, n, it);
            hipEventRecord(e1);
            k_read<<<grid, 256>>>(buf, n, out);
            hipEventRecord(e2);
            hipEventSynchronize(e2);
            float a, b;
            hipEventElapsedTime(&a, e0, e1);
            hipEventElapsedTime(&b, e1, e2);
            if (a < bw) bw = a;
            if (b < br) br = b;
        }
        printf("%5zu MB  write %6.2f TB/s (%.1f us)   read-after-write %6.2f TB/s (%.1f us)\n", mb, mb / 1048.576 / bw, bw * 1e3, mb / 1048.576 / br, br * 1e3);
    }
    // interleaved ring: write chunk k, read chunk k-1 concurrently is not measured here
    return 0;
}
