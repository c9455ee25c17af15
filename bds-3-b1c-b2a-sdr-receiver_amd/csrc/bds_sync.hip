// Frame-synchronisation correlators on the prompt outputs of tracking:
//   B1C  B1C/include/BCNAV1decoding.m:66-91   bits = sign(Pilot_I_P | Pilot_Q_P),
//        Secondary = generate2ndCode(PRN) (1800 chips), XcorrResult = xcorr(bits, Secondary),
//        second half, index = find(abs(.) >= 1799.5)
//   B2a  B2a/include/BCNAV2decoding.m:69-97   bits = sign(I_P), preamble_ms = kron(preamble_bits,
//        secondCode) (24 x 5 = 120 taps), index = find(abs(xcorr second half) > 115)
// The correlation is integer (+-1 against +-1), so the device result equals the reference's exactly.
// One thread per lag; the pattern sits in LDS, the thresholded bits are int8 in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "bds_internal.h"

namespace bds {

static constexpr int kMaxPattern = 1800;

// bits(bits > 0) = 1; bits(bits <= 0) = -1   (NaN > 0 is false -> -1, as in MATLAB)
__global__ __launch_bounds__(256) void k_sync_bits(const double *__restrict__ prompt, long n, int8_t *__restrict__ bits) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        bits[i] = prompt[i] > 0 ? 1 : -1;
}

// out[ch][lag] = sum_k bits[ch][lag + k] * pattern[ch][k],  lag = 0 .. M-1, M = max(n, m): the second
// half of xcorr(bits, pattern), which zero-pads the shorter input to the longer one's length.
__global__ __launch_bounds__(256) void k_sync_xcorr(const int8_t *__restrict__ bits, int n,
                                                    const int8_t *__restrict__ pattern, int m, int M,
                                                    int32_t *__restrict__ out) {
    __shared__ int8_t s_pat[kMaxPattern];
    const int ch = blockIdx.y;
    for (int i = threadIdx.x; i < m; i += blockDim.x) s_pat[i] = pattern[(long)ch * m + i];
    __syncthreads();
    const int8_t *b = bits + (long)ch * n;
    for (int lag = blockIdx.x * blockDim.x + threadIdx.x; lag < M; lag += gridDim.x * blockDim.x) {
        int acc = 0;
        const int kmax = lag < n ? min(m, n - lag) : 0;
        for (int k = 0; k < kmax; ++k) acc += (int)b[lag + k] * (int)s_pat[k];
        out[(long)ch * M + lag] = acc;
    }
}

// B2a/include/unpack_cplx.m:16-63: byte -> (I1, Q1, I2, Q2) int8; low nibble first; per nibble bit 0 / 1
// = sign of I / Q, bit 2 / 3 = magnitude 1 or 3 of I / Q.  One 32-bit store per input byte.
__device__ __forceinline__ uint32_t unpack_nibble(uint32_t v) {
    const int i = ((v & 4) ? 3 : 1) * ((v & 1) ? -1 : 1);
    const int q = ((v & 8) ? 3 : 1) * ((v & 2) ? -1 : 1);
    return (uint32_t)(uint8_t)(int8_t)i | ((uint32_t)(uint8_t)(int8_t)q << 8);
}
__global__ __launch_bounds__(256) void k_unpack_cplx(const uint8_t *__restrict__ in, size_t n, uint32_t *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = in[i];
        out[i] = unpack_nibble(b & 15u) | (unpack_nibble(b >> 4) << 16);
    }
}

static hipStream_t st(bds_ctx *ctx) { return (hipStream_t)ctx->stream; }

}  // namespace bds

using namespace bds;

extern "C" int bds_sync_pattern(int signal, int prn, int8_t *out, int cap) {
    if (!out) return BDS_ERR_ARG;
    if (signal == BDS_SIGNAL_B1C) {
        if (cap < 1800) return BDS_ERR_ARG;
        return gen_secondary(prn, out);  // BCNAV1decoding.m:80
    }
    if (signal == BDS_SIGNAL_B2A) {
        if (cap < 120) return BDS_ERR_ARG;
        static const int8_t second[5] = {1, 1, 1, -1, 1};  // BCNAV2decoding.m:69
        static const int8_t preamble[24] = {-1, -1, -1, 1, 1, 1, -1, 1, 1, -1, 1, 1,
                                            -1, -1, 1, -1, -1, -1, -1, 1, -1, 1, 1, 1};  // :74
        for (int i = 0; i < 24; ++i)
            for (int j = 0; j < 5; ++j) out[i * 5 + j] = (int8_t)(preamble[i] * second[j]);  // kron, :78
        return 120;
    }
    return BDS_ERR_ARG;
}

extern "C" int bds_frame_sync(bds_ctx *ctx, int signal, int n_ch, const int32_t *prn, const double *prompt, int n,
                              int32_t *xcorr, int32_t *index, int32_t *n_index, int cap) {
    if (!ctx || !prn || !prompt || n_ch < 1 || n < 1) return BDS_ERR_ARG;
    if (signal != BDS_SIGNAL_B1C && signal != BDS_SIGNAL_B2A) return fail(ctx, BDS_ERR_ARG, "bds_frame_sync: signal invalid");
    const int m = signal == BDS_SIGNAL_B1C ? 1800 : 120;
    const int M = std::max(n, m);
    std::vector<int8_t> pat((size_t)n_ch * m);
    for (int c = 0; c < n_ch; ++c) {
        const int p = signal == BDS_SIGNAL_B1C ? prn[c] : 1;
        if (bds_sync_pattern(signal, p, &pat[(size_t)c * m], m) < 0)
            return fail(ctx, BDS_ERR_ARG, "bds_frame_sync: channel %d PRN %d out of range", c + 1, prn[c]);
    }
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    double *d_p = nullptr;
    int8_t *d_bits = nullptr, *d_pat = nullptr;
    int32_t *d_out = nullptr;
    const size_t np = (size_t)n_ch * n;
    auto release = [&]() {
        for (void *q : {(void *)d_p, (void *)d_bits, (void *)d_pat, (void *)d_out})
            if (q) (void)hipFree(q);
    };
    hipError_t e = hipMalloc((void **)&d_p, sizeof(double) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&d_bits, np);
    if (e == hipSuccess) e = hipMalloc((void **)&d_pat, pat.size());
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(int32_t) * (size_t)n_ch * M);
    if (e != hipSuccess) {
        release();
        return fail(ctx, BDS_ERR_NOMEM, "bds_frame_sync: %s", hipGetErrorString(e));
    }
    (void)hipMemcpyAsync(d_p, prompt, sizeof(double) * np, hipMemcpyHostToDevice, st(ctx));
    (void)hipMemcpyAsync(d_pat, pat.data(), pat.size(), hipMemcpyHostToDevice, st(ctx));
    hipLaunchKernelGGL(k_sync_bits, dim3((unsigned)std::min<size_t>(1024, (np + 255) / 256)), dim3(256), 0, st(ctx),
                       (const double *)d_p, (long)np, d_bits);
    hipLaunchKernelGGL(k_sync_xcorr, dim3((unsigned)((M + 255) / 256), (unsigned)n_ch), dim3(256), 0, st(ctx),
                       (const int8_t *)d_bits, n, (const int8_t *)d_pat, m, M, d_out);
    std::vector<int32_t> h((size_t)n_ch * M);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, st(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(st(ctx));
    release();
    if (e != hipSuccess) return fail(ctx, BDS_ERR_HIP, "bds_frame_sync: %s", hipGetErrorString(e));
    if (xcorr) std::copy(h.begin(), h.end(), xcorr);
    int total = 0;
    for (int c = 0; c < n_ch; ++c) {
        int cnt = 0;
        for (int lag = 0; lag < M; ++lag) {
            const int v = std::abs(h[(size_t)c * M + lag]);
            // abs(XcorrResult) >= 1799.5 (BCNAV1decoding.m:91); abs(tlmXcorrResult) > 115 (BCNAV2decoding.m:97)
            const bool hit = signal == BDS_SIGNAL_B1C ? (double)v >= 1799.5 : v > 115;
            if (!hit) continue;
            if (index && cnt < cap) index[(size_t)c * cap + cnt] = lag + 1;  // 1-based like find()
            ++cnt;
        }
        if (n_index) n_index[c] = cnt;
        total += cnt;
    }
    return total;
}

extern "C" int bds_unpack_cplx(bds_ctx *ctx, const uint8_t *in, size_t n_bytes, int8_t *out) {
    if (!ctx || (!in && n_bytes) || (!out && n_bytes)) return BDS_ERR_ARG;
    if (n_bytes == 0) return BDS_OK;
    BDS_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d_in = nullptr;
    uint32_t *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, n_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, 4 * n_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, n_bytes, hipMemcpyHostToDevice, st(ctx));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_unpack_cplx, dim3((unsigned)std::min<size_t>(8192, (n_bytes + 255) / 256)), dim3(256), 0, st(ctx),
                           (const uint8_t *)d_in, n_bytes, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 4 * n_bytes, hipMemcpyDeviceToHost, st(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(st(ctx));
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, BDS_ERR_HIP, "bds_unpack_cplx: %s", hipGetErrorString(e));
    return BDS_OK;
}

// unpack_cplx(filename_in, filename_out): 4e6-byte pieces in the reference (:11), 64 MiB here
extern "C" int bds_unpack_cplx_file(bds_ctx *ctx, const char *path_in, const char *path_out) {
    if (!ctx || !path_in || !path_out) return BDS_ERR_ARG;
    FILE *fi = fopen(path_in, "rb");
    if (!fi) return fail(ctx, BDS_ERR_IO, "Unable to read file %s", path_in);
    FILE *fo = fopen(path_out, "wb");
    if (!fo) {
        fclose(fi);
        return fail(ctx, BDS_ERR_IO, "Unable to write file %s", path_out);
    }
    const size_t piece = 64u << 20;
    std::vector<uint8_t> a(piece);
    std::vector<int8_t> b(4 * piece);
    int rc = BDS_OK;
    for (;;) {
        const size_t n = fread(a.data(), 1, piece, fi);
        if (n == 0) break;
        if ((rc = bds_unpack_cplx(ctx, a.data(), n, b.data()))) break;
        if (fwrite(b.data(), 1, 4 * n, fo) != 4 * n) {
            rc = fail(ctx, BDS_ERR_IO, "short write on %s", path_out);
            break;
        }
    }
    fclose(fi);
    fclose(fo);
    return rc;
}
