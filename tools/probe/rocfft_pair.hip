// The north star's literal design for the cfg3 search loop, timed on this chip: per PRN and Doppler row
//   product  X_b .* conj(C_p)  (data + pilot)  ->  batched rocFFT inverse c2c (fp32)  ->  w_d|y_d| + w_p|y_p|, maximum per cell
// for one launch group of 201 cells x 2 components = 402 transforms (B1C/acquisition.m:198-220, GPU_acquisition.m:184-227),
// at the padded 5-smooth length L = 3 145 728 this library uses and at the reference's own length N = 1 987 500 (= 2^2 3 5^5 53:
// rocFFT takes it through Bluestein / a large prime-factor kernel).  Spectra are fp32 complex as in GPU_acquisition.m (single).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/rocfft_pair.hip -lrocfft -o rocfft_pair && ./rocfft_pair
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)
#define RK(x)                                                  \
    do {                                                       \
        rocfft_status s_ = (x);                                \
        if (s_ != rocfft_status_success) {                     \
            fprintf(stderr, "%s: rocfft status %d\n", #x, s_); \
            exit(1);                                           \
        }                                                      \
    } while (0)

// out[(2 g + c) L + k] = X[g L + k] * conj(C[c L + k])
__global__ void k_product(const float2 *__restrict__ X, const float2 *__restrict__ C, float2 *__restrict__ out, long L, int G) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L) return;
    const float2 c0 = C[k], c1 = C[L + k];
    for (int g = blockIdx.y; g < G; g += gridDim.y) {
        const float2 x = X[(long)g * L + k];
        out[(long)(2 * g) * L + k] = make_float2(x.x * c0.x + x.y * c0.y, x.y * c0.x - x.x * c0.y);
        out[(long)(2 * g + 1) * L + k] = make_float2(x.x * c1.x + x.y * c1.y, x.y * c1.x - x.x * c1.y);
    }
}

// per cell: max over lags < n_lags of w0 |y0| + w1 |y1|  (value and first index through a packed 64-bit atomic max)
__global__ void k_magmax(const float2 *__restrict__ y, long L, long n_lags, float w0, float w1, unsigned long long *__restrict__ best) {
    const int g = blockIdx.y;
    const float2 *y0 = y + (long)(2 * g) * L, *y1 = y0 + L;
    float bv = -1.f;
    long bi = 0;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n_lags; k += (long)gridDim.x * blockDim.x) {
        const float2 a = y0[k], b = y1[k];
        const float v = w0 * sqrtf(a.x * a.x + a.y * a.y) + w1 * sqrtf(b.x * b.x + b.y * b.y);
        if (v > bv) bv = v, bi = k;
    }
    unsigned long long pk = ((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)~(unsigned)bi;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long q = __shfl_xor(pk, o);
        pk = q > pk ? q : pk;
    }
    if ((threadIdx.x & 63) == 0 && bv >= 0.f) atomicMax(&best[g], pk);
}

static void run(long L, long n_lags, int G, const char *label) {
    const int B = 2 * G;
    float2 *X, *C, *Y;
    unsigned long long *best;
    CK(hipMalloc(&X, sizeof(float2) * (size_t)G * L));
    CK(hipMalloc(&C, sizeof(float2) * (size_t)2 * L));
    CK(hipMalloc(&Y, sizeof(float2) * (size_t)B * L));
    CK(hipMalloc(&best, sizeof(unsigned long long) * G));
    {
        std::vector<float2> h((size_t)L);
        unsigned s = 12345u;
        for (long i = 0; i < L; ++i) {
            s = s * 1664525u + 1013904223u;
            h[(size_t)i] = make_float2((float)((s >> 8) & 0xffff) / 65536.f - 0.5f, (float)((s >> 16) & 0xffff) / 65536.f - 0.5f);
        }
        for (int g = 0; g < G; ++g) CK(hipMemcpy(X + (size_t)g * L, h.data(), sizeof(float2) * L, hipMemcpyHostToDevice));
        for (int c = 0; c < 2; ++c) CK(hipMemcpy(C + (size_t)c * L, h.data(), sizeof(float2) * L, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    rocfft_plan plan = nullptr;
    size_t len[1] = {(size_t)L};
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, st));
    RK(rocfft_plan_create(&plan, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single, 1, len, (size_t)B, nullptr));
    size_t wsize = 0;
    RK(rocfft_plan_get_work_buffer_size(plan, &wsize));
    void *wbuf = nullptr;
    if (wsize) CK(hipMalloc(&wbuf, wsize));
    rocfft_execution_info info = nullptr;
    RK(rocfft_execution_info_create(&info));
    if (wsize) RK(rocfft_execution_info_set_work_buffer(info, wbuf, wsize));
    RK(rocfft_execution_info_set_stream(info, st));
    CK(hipEventRecord(t1, st));
    CK(hipEventSynchronize(t1));
    float plan_ms = 0;
    CK(hipEventElapsedTime(&plan_ms, t0, t1));
    hipEvent_t e[4];
    for (auto &x : e) CK(hipEventCreate(&x));
    const dim3 gp((unsigned)((L + 255) / 256), 3), gm(1024, (unsigned)G);
    double acc[3] = {0, 0, 0};
    const int reps = 5;
    for (int r = -1; r < reps; ++r) {  // r = -1: warm-up
        CK(hipMemsetAsync(best, 0, sizeof(unsigned long long) * G, st));
        CK(hipEventRecord(e[0], st));
        hipLaunchKernelGGL(k_product, gp, dim3(256), 0, st, X, C, Y, L, G);
        CK(hipEventRecord(e[1], st));
        void *bufs[1] = {Y};
        RK(rocfft_execute(plan, bufs, nullptr, info));
        CK(hipEventRecord(e[2], st));
        hipLaunchKernelGGL(k_magmax, gm, dim3(256), 0, st, Y, L, n_lags, 0.5244f, 0.8515f, best);
        CK(hipEventRecord(e[3], st));
        CK(hipEventSynchronize(e[3]));
        if (r < 0) continue;
        for (int i = 0; i < 3; ++i) {
            float ms;
            CK(hipEventElapsedTime(&ms, e[i], e[i + 1]));
            acc[i] += ms;
        }
    }
    const double p = acc[0] / reps, f = acc[1] / reps, m = acc[2] / reps;
    printf("%-34s L = %8ld  batch %d  work buffer %.2f GB  plan %.0f ms | product %.3f ms  rocFFT inverse %.3f ms  |.|+max %.3f ms  total %.3f ms per %d cells"
           "  (%.1f us per cell; inverse alone %.1f GB/s of its 2 x 8 B per point)\n",
           label, L, B, wsize / 1e9, plan_ms, p, f, m, p + f + m, G, (p + f + m) * 1e3 / G, 16.0 * B * L / (f * 1e-3) / 1e9);
    RK(rocfft_execution_info_destroy(info));
    RK(rocfft_plan_destroy(plan));
    if (wbuf) CK(hipFree(wbuf));
    CK(hipFree(X));
    CK(hipFree(C));
    CK(hipFree(Y));
    CK(hipFree(best));
    CK(hipStreamDestroy(st));
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 201;
    RK(rocfft_setup());
    run(3145728, 1987500, G, "padded 2^20 x 3 (this library's L)");
    run(1987500, 1987500, G, "reference length N = 2^2 3 5^5 53");
    RK(rocfft_cleanup());
    return 0;
}
