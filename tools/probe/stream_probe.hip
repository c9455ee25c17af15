// What this box's HBM delivers to a streaming kernel, by access form (VERDICT r4 "what's weak" 6: the round-3/4 "memory floor"
// rested on a plain grid-stride uint4 read, tools/probe/mall_probe.hip, that tops out at 5.1 TB/s).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/stream_probe.hip -o gpurun_out/stream_probe && gpurun_out/stream_probe
// Forms: 16-byte loads per lane, U of them in flight per lane before the first use (plain / nontemporal), the same through
// buffer loads with the cache-policy bits of the instruction (aux 0 / 2 = nt), global -> LDS DMA (aux 0 / 2), float4 copy
// (plain and nontemporal stores), write-only (plain / nt).  4 GiB per pass (16x the 256 MB Infinity Cache), best of 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n, unsigned *out) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + i + u * stride)) : *reinterpret_cast<const v4u *>(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// contiguous per workgroup: a workgroup walks its own 1/grid slice (DRAM page locality) instead of the grid-stride interleave
template <int U, int AUX>
__global__ __launch_bounds__(256) void k_read_buf(const uint4 *__restrict__ p, size_t n, unsigned *out) {
    const size_t per = n / gridDim.x;  // elements of this workgroup's slice (multiple of 256 U by construction)
    const uint4 *base = p + per * blockIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((i + u * 256) * 16), 0, AUX);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// global -> LDS DMA: every wave streams 1 KiB pieces into its own LDS ring of R slots and never reads them back with the
// vector unit beyond one dword per slot (the probe measures the fetch path, not a consumer)
template <int R, int AUX>
__global__ __launch_bounds__(256) void k_read_lds(const uint4 *__restrict__ p, size_t n, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[4][R][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = n / gridDim.x;
    const uint4 *base = p + per * blockIdx.x;
    unsigned acc = 0;
    for (size_t i = (size_t)wave * 64; i + 192 + (R - 1) * 256 < per; i += (size_t)R * 256) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            __builtin_amdgcn_global_load_lds(base + i + r * 256 + lane, (__attribute__((address_space(3))) void *)&ring[wave][r][0], 16, 0, AUX);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        acc += *reinterpret_cast<volatile unsigned *>(&ring[wave][0][lane * 4]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const v4f v = *reinterpret_cast<const v4f *>(s + i);
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(d + i));
        else *reinterpret_cast<v4f *>(d + i) = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ d, size_t n, unsigned x) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const v4u v = {x, x + 1, x + 2, (unsigned)i};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(d + i));
        else *reinterpret_cast<v4u *>(d + i) = v;
    }
}

template <class F>
static float best_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    hipEventDestroy(e0), hipEventDestroy(e1);
    return best;
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    uint4 *a, *b;
    unsigned *out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(a, 1, bytes), hipMemset(b, 2, bytes);
    const double gb = bytes / 1e9;
    auto row = [&](const char *name, double moved_gb, float ms) { printf("%-58s %7.3f ms  %6.2f TB/s\n", name, ms, moved_gb / ms); };
    for (int grid : {256 * 8, 256 * 16, 256 * 32}) {
        printf("-- grid %d workgroups of 256\n", grid);
        row("read 16 B/lane, grid-stride, 1 in flight (mall_probe form)", gb, best_ms([&] { k_read<1, false><<<grid, 256>>>(a, n, out); }));
        row("read grid-stride, 4 in flight", gb, best_ms([&] { k_read<4, false><<<grid, 256>>>(a, n, out); }));
        row("read grid-stride, 8 in flight", gb, best_ms([&] { k_read<8, false><<<grid, 256>>>(a, n, out); }));
        row("read grid-stride, 8 in flight, nontemporal", gb, best_ms([&] { k_read<8, true><<<grid, 256>>>(a, n, out); }));
        row("read buffer_load, slice per workgroup, 4 in flight, aux 0", gb, best_ms([&] { k_read_buf<4, 0><<<grid, 256>>>(a, n, out); }));
        row("read buffer_load, slice per workgroup, 8 in flight, aux 0", gb, best_ms([&] { k_read_buf<8, 0><<<grid, 256>>>(a, n, out); }));
        row("read buffer_load, slice per workgroup, 8 in flight, aux 2 (nt)", gb, best_ms([&] { k_read_buf<8, 2><<<grid, 256>>>(a, n, out); }));
        row("read global->LDS DMA, 4 x 1 KiB per wave in flight, aux 0", gb, best_ms([&] { k_read_lds<4, 0><<<grid, 256>>>(a, n, out); }));
        row("read global->LDS DMA, 8 x 1 KiB per wave in flight, aux 0", gb, best_ms([&] { k_read_lds<8, 0><<<grid, 256>>>(a, n, out); }));
        row("read global->LDS DMA, 8 x 1 KiB per wave in flight, aux 2 (nt)", gb, best_ms([&] { k_read_lds<8, 2><<<grid, 256>>>(a, n, out); }));
        row("copy float4 (read + write counted)", 2 * gb, best_ms([&] { k_copy<false><<<grid, 256>>>((const float4 *)a, (float4 *)b, n); }));
        row("copy float4, nontemporal stores", 2 * gb, best_ms([&] { k_copy<true><<<grid, 256>>>((const float4 *)a, (float4 *)b, n); }));
        row("write 16 B/lane", gb, best_ms([&] { k_write<false><<<grid, 256>>>(b, n, 3); }));
        row("write 16 B/lane, nontemporal", gb, best_ms([&] { k_write<true><<<grid, 256>>>(b, n, 3); }));
    }
    return 0;
}
