// One instantiation of the wave-private column pass on its own: fast turn-around for register / ISA checks
//   tools/wcols_res.sh [-DBDS_WCOLS_OCC=2 ...]
#include "bds_acq_wcols.h"
#ifndef WC_S
#define WC_S 768
#endif
#ifndef WC_NV
#define WC_NV 6
#endif
template __global__ void bds::k_cols_wave_f<WC_S, 2, false, __half2, WC_NV>(bds::WColsArgs);
template __global__ void bds::k_cols_wave_f<WC_S, 2, false, __half2, WC_NV, true>(bds::WColsArgs);
template __global__ void bds::k_cols_wave_f<WC_S, 2, false, __half2, WC_NV, true, true>(bds::WColsArgs);
