// LDS b64 access rate for the lane -> address patterns of the wave-private column pass (bds_acq_wcols.h) and some
// alternatives: which lanes the hardware serves together decides what "conflict free" means, and the model in
// tools/proto_cols_wave.py assumes lanes 0-31 / 32-63.  Prints cycles per wave-instruction for each pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/lds_pattern.hip -o gpurun_out/lds_pattern && gpurun_out/lds_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
constexpr int ITER = 512;

__global__ __launch_bounds__(256) void k_write(const int *offs, float *out, int which) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = (unsigned)offs[which * 64 + lane] + wave * 16384u;
    float2 v = make_float2((float)lane, 1.f);
    for (int it = 0; it < ITER; ++it) {
        REP8(asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");)
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = ((float *)lds)[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_read(const int *offs, float *out, int which) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = (unsigned)offs[which * 64 + lane] + wave * 16384u;
    float2 acc = make_float2(0.f, 0.f), v;
    for (int it = 0; it < ITER; ++it) {
        REP8(asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc.x += v.x;)
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}
// reads without the per-instruction wait (throughput, 8 in flight)
__global__ __launch_bounds__(256) void k_read_tp(const int *offs, float *out, int which) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = (unsigned)offs[which * 64 + lane] + wave * 16384u;
    float2 v0, v1, v2, v3, v4, v5, v6, v7;
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n"
                     "ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        acc += v0.x + v7.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef float f4_t __attribute__((ext_vector_type(4)));
// b128 accesses (two elements per lane): byte address = 2 x the pattern's
__global__ __launch_bounds__(256) void k_write128(const int *offs, float *out, int which) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = 2u * (unsigned)offs[which * 64 + lane] + wave * 16384u;
    f4_t v = {(float)lane, 1.f, 2.f, 3.f};
    for (int it = 0; it < ITER; ++it) {
        REP8(asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");)
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = ((float *)lds)[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_read128_tp(const int *offs, float *out, int which) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = 2u * (unsigned)offs[which * 64 + lane] + wave * 16384u;
    f4_t v0, v1, v2, v3, v4, v5, v6, v7;
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n"
                     "ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        acc += v0.x + v7.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    struct Pat { const char *name; int off[64]; };
    std::vector<Pat> pats;
    auto add = [&](const char *name, auto f) { Pat p; p.name = name; for (int l = 0; l < 64; ++l) p.off[l] = 8 * f(l); pats.push_back(p); };
    const int MS = 68, RS = 24 * 68 + 8;
    add("linear lane", [](int l) { return l; });
    add("phase A write  cp*RS + bq (RS = 1640: 16 dwords mod 64)", [=](int l) { return (l & 3) * RS + (l >> 2); });
    add("phase A alt    cp*(RS+8) + bq (32 dwords mod 64)", [=](int l) { return (l & 3) * (RS + 8) + (l >> 2); });
    add("phase A alt    cp*16 + bq (one region, 16 elements apart)", [=](int l) { return (l & 3) * 16 + (l >> 2); });
    add("phase A alt    lanes (bq = l & 15, cp = l >> 4): cp*RS + bq", [=](int l) { return (l >> 4) * RS + (l & 15); });
    add("stage 2 r/w    ml*68 + bl", [=](int l) { return (l & 7) * MS + (l >> 3); });
    add("stage 2 alt    ml*72 + bl", [=](int l) { return (l & 7) * 72 + (l >> 3); });
    add("stage 2 alt    lanes (bl = l & 7, ml = l >> 3): ml*72 + bl", [=](int l) { return (l >> 3) * 72 + (l & 7); });
    add("stage 3 read   ml*68 + 8 u + ((j + u) & 7), j = 0", [=](int l) { return (l & 7) * MS + 8 * (l >> 3) + ((l >> 3) & 7); });
    add("stage 3 read   j = 3", [=](int l) { return (l & 7) * MS + 8 * (l >> 3) + ((3 + (l >> 3)) & 7); });
    add("stage 3 unrot  ml*68 + 8 u", [=](int l) { return (l & 7) * MS + 8 * (l >> 3); });
    // wave-private row pass (bds_acq_wrows.h): lanes (ql = l & 3, bl = l >> 2)
    add("rows 1a write  64 u + lane", [](int l) { return l; });
    add("rows 1b read   64 bl + ql + 4 ((j + bl) & 15), j = 5", [](int l) { return 64 * (l >> 2) + (l & 3) + 4 * ((5 + (l >> 2)) & 15); });
    add("rows exch write 19 bl + q' (q' = ql)", [](int l) { return 19 * (l >> 2) + (l & 3); });
    add("rows exch write 17 bl + q'", [](int l) { return 17 * (l >> 2) + (l & 3); });
    add("rows exch write 21 bl + q'", [](int l) { return 21 * (l >> 2) + (l & 3); });
    add("rows exch write 16 bl + ((q' + 4 (bl >> 1)) & 15)  (swizzled)", [](int l) { return 16 * (l >> 2) + (((l & 3) + 4 * ((l >> 2) >> 1)) & 15); });
    add("rows phase 2 read 19 e'' + q'", [](int l) { return 19 * l + 3; });
    add("rows phase 2 read 17 e'' + q'", [](int l) { return 17 * l + 3; });
    add("rows phase 2 read 16 e'' + ((j + (e'' >> 1)) & 15)  (swizzled), j = 3", [](int l) { return 16 * l + ((3 + (l >> 1)) & 15); });
    add("stride 2 elements (2-way by any model)", [](int l) { return 2 * l; });
    add("stride 32 elements (all one bank pair)", [](int l) { return 32 * (l & 31) + (l >> 5); });
    std::vector<int> h;
    for (auto &p : pats) h.insert(h.end(), p.off, p.off + 64);
    int *d_off; float *d_out;
    hipMalloc(&d_off, h.size() * 4); hipMemcpy(d_off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * 2;
    hipMalloc(&d_out, blocks * 256 * 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t i = 0; i < pats.size(); ++i) {
        double res[3];
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 65536, 0, d_off, d_out, (int)i);
                else if (mode == 1) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 65536, 0, d_off, d_out, (int)i);
                else hipLaunchKernelGGL(k_read_tp, dim3(blocks), dim3(256), 65536, 0, d_off, d_out, (int)i);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per CU: 2 workgroups x 4 waves = 8 waves (2 per SIMD) each ITER*8 instructions; cycles per wave-instruction per CU
            res[mode] = ms * 1e-3 * clk / (2.0 * 4 * ITER * 8);
        }
        printf("%-62s write %6.1f  read(lat) %6.1f  read(tp) %6.1f  CU-cycles per wave-instruction\n", pats[i].name, res[0], res[1], res[2]);
    }
    // b128: the same lane -> element patterns, two elements (16 bytes) per lane
    for (size_t i : {(size_t)0, (size_t)1, (size_t)5}) {
        double res[2];
        for (int mode = 0; mode < 2; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k_write128, dim3(blocks), dim3(256), 65536, 0, d_off, d_out, (int)i);
                else hipLaunchKernelGGL(k_read128_tp, dim3(blocks), dim3(256), 65536, 0, d_off, d_out, (int)i);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            res[mode] = ms * 1e-3 * clk / (2.0 * 4 * ITER * 8);
        }
        printf("b128 %-57s write %6.1f  read(tp) %6.1f  CU-cycles per wave-instruction (1024 bytes)\n", pats[i].name, res[0], res[1]);
    }
    return 0;
}
