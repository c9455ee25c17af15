// Does v_dot2_f32_f16 honour the VOP3P source modifiers (neg_hi, op_sel / op_sel_hi) on gfx950?  The row pass's spectrum product
// X conj(C) needs (xr, xi).(cr, -ci) and (xr, xi).(ci, cr): with the modifiers both come from the stored words, without a prepared
// (xr, -xi) copy of every signal word.  (op_sel on a dot instruction is rejected by the assembler: the (xi, xr) copy stays.)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
__global__ void k(const uint32_t *x, const uint32_t *c, float *re, float *im, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r, m;
    asm volatile("v_dot2_f32_f16 %0, %2, %3, 0 neg_hi:[0,1,0]\n"
                 "v_dot2_f32_f16 %1, %2, %3, 0 neg_lo:[0,1,0]\n s_nop 2"
                 : "=&v"(r), "=&v"(m)
                 : "v"(x[i]), "v"(c[i]));
    re[i] = r, im[i] = m;
}
int main() {
    const int n = 4096;
    uint32_t *hx = new uint32_t[n], *hc = new uint32_t[n];
    float *xr = new float[n], *xi = new float[n], *cr = new float[n], *ci = new float[n];
    for (int i = 0; i < n; ++i) {
        xr[i] = __half2float(__float2half(drand48() * 4 - 2)), xi[i] = __half2float(__float2half(drand48() * 4 - 2));
        cr[i] = __half2float(__float2half(drand48() * 4 - 2)), ci[i] = __half2float(__float2half(drand48() * 4 - 2));
        const uint16_t a = __half_as_ushort(__float2half(xr[i])), b = __half_as_ushort(__float2half(xi[i]));
        const uint16_t d = __half_as_ushort(__float2half(cr[i])), e = __half_as_ushort(__float2half(ci[i]));
        hx[i] = a | ((uint32_t)b << 16), hc[i] = d | ((uint32_t)e << 16);
    }
    uint32_t *dx, *dc;
    float *dre, *dim;
    hipMalloc(&dx, n * 4), hipMalloc(&dc, n * 4), hipMalloc(&dre, n * 4), hipMalloc(&dim, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice), hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dc, dre, dim, n);
    float *re = new float[n], *im = new float[n];
    hipMemcpy(re, dre, n * 4, hipMemcpyDeviceToHost), hipMemcpy(im, dim, n * 4, hipMemcpyDeviceToHost);
    double wr = 0, wi = 0;
    for (int i = 0; i < n; ++i) {
        wr = fmax(wr, fabs(re[i] - (xr[i] * cr[i] - xi[i] * ci[i])));
        wi = fmax(wi, fabs(im[i] - (-xr[i] * cr[i] + xi[i] * ci[i])));
    }
    printf("v_dot2_f32_f16 with neg_hi on the code word: worst |re - (xr cr - xi ci)| = %.3g; with neg_lo: worst |v - (-xr cr + xi ci)| = %.3g\n", wr, wi);
    return 0;
}
