// What does a large hipMalloc cost on this box?  (round 5: the multi-PRN launch pairs want an inter-pass buffer of 40 - 160 GB)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/alloc_probe.hip -o tools/probe/alloc_probe.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(char *p, size_t n, size_t stride) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i < n) p[i] = 1;
}

int main(int argc, char **argv) {
    (void)hipFree(nullptr);
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    printf("free %.1f GiB of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    // sizes in GiB from the command line (default: an ascending series, every allocation freed before the next)
    double dflt[] = {1, 5, 20, 40, 80, 160, 160, 20};
    int ns = argc > 1 ? argc - 1 : (int)(sizeof(dflt) / sizeof(dflt[0]));
    for (int k = 0; k < ns; ++k) {
        const double gb = argc > 1 ? atof(argv[k + 1]) : dflt[k];
        const size_t n = (size_t)(gb * 1073741824.0);
        char *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc((void **)&p, n);
        double t1 = now();
        if (e != hipSuccess) {
            printf("%6.0f GiB: hipMalloc failed: %s\n", gb, hipGetErrorString(e));
            continue;
        }
        // first touch of every 2 MiB page
        const size_t stride = 2u << 20, cnt = n / stride;
        hipLaunchKernelGGL(touch, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, p, n, stride);
        (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipMemsetAsync(p, 0, n, 0);
        (void)hipDeviceSynchronize();
        double t3 = now();
        (void)hipFree(p);
        double t4 = now();
        printf("%6.0f GiB: hipMalloc %8.1f ms, first touch %8.1f ms, memset %8.1f ms, hipFree %8.1f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
    return 0;
}
