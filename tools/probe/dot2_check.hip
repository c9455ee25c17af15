// numeric check of the packed-fp16 complex product and conversions used by bds_acq_f32.h
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bds_acq_f32.h"
using namespace bds;
__global__ void k(const uint32_t *x, const uint32_t *c, float2 *out, uint32_t *packed, float2 *unpacked) {
    const int i = threadIdx.x;
    const uint32_t xs = __builtin_amdgcn_alignbit(x[i], x[i], 16);
    out[i] = cmul_h(x[i], xs, c[i]);
    packed[i] = f2_to_h2(out[i]);
    unpacked[i] = h2_to_f2(x[i]);
}
int main() {
    const int n = 8;
    _Float16 hx[n][2] = {{1, 2}, {3, -4}, {0.5, 0.25}, {-7, 9}, {100, -200}, {0.001f, 3}, {300, 5}, {30000, 20000}};
    _Float16 hc[n][2] = {{5, 6}, {-1, 2}, {8, -16}, {0.5, 0.5}, {30, 40}, {1000, -2}, {300, 2}, {20000, -30000}};
    uint32_t *dx, *dc, *dp; float2 *dout, *dun;
    hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&dout, n * 8); hipMalloc(&dun, n * 8);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice);
    k<<<1, n>>>(dx, dc, dout, dp, dun);
    float2 out[n], un[n]; _Float16 pk[n][2];
    hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost); hipMemcpy(pk, dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(un, dun, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float xr = (float)hx[i][0], xi = (float)hx[i][1], cr = (float)hc[i][0], ci = (float)hc[i][1];
        const float re = xr * cr - xi * ci, im = xr * ci + xi * cr;
        const bool ok = out[i].x == re && out[i].y == im && un[i].x == xr && un[i].y == xi && (float)pk[i][0] == (float)(_Float16)re && (float)pk[i][1] == (float)(_Float16)im;
        printf("%d: got (%g, %g) want (%g, %g) packed (%g, %g) unpacked (%g, %g) %s\n", i, out[i].x, out[i].y, re, im, (float)pk[i][0], (float)pk[i][1], un[i].x, un[i].y, ok ? "ok" : "BAD");
        bad += !ok;
    }
    printf("%s\n", bad ? "FAILED" : "all ok");
    return bad;
}
