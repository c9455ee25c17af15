// Debug build (BDS_DEBUG=1 ./build.sh -> libbds_mi355x_debug.so): device-side bounds checks on every
// code-table, IF-window and candidate-list index of the kernels.  A failed check prints file:line and the condition (the first
// few of a launch) and counts; bds_debug_failures() returns and clears the count (tests/conftest.py asserts it is zero at the
// end of a run on the debug library).  No trap: __builtin_trap() inside the tracking kernels makes this compiler emit an
// "Illegal instruction: operand has incorrect register class" error next to their DPP reductions.  Compiled out of the
// product build.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#ifdef BDS_DEBUG
static __device__ unsigned int g_bds_dassert_failures;  // per translation unit; summed by bds_debug_failures() (bds_api.hip)
#define BDS_DASSERT(cond)                                                                                          \
    do {                                                                                                           \
        if (!(cond)) {                                                                                             \
            if (atomicAdd(&g_bds_dassert_failures, 1u) < 8u)                                                       \
                printf("BDS_DASSERT failed %s:%d: %s (block %d thread %d)\n", __FILE__, __LINE__, #cond, (int)blockIdx.x, (int)threadIdx.x); \
        }                                                                                                          \
    } while (0)
// read and clear this translation unit's counter (host)
#define BDS_DEBUG_TU_READER(name)                                                                       \
    extern "C" unsigned int name() {                                                                    \
        unsigned int h = 0, z = 0;                                                                      \
        (void)hipDeviceSynchronize();                                                                   \
        if (hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_bds_dassert_failures), sizeof(h)) != hipSuccess) return 0xffffffffu; \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bds_dassert_failures), &z, sizeof(z));                     \
        return h;                                                                                       \
    }
#else
#define BDS_DASSERT(cond) ((void)0)
#define BDS_DEBUG_TU_READER(name)
#endif
