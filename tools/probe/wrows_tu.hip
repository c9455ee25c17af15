// One instantiation of the wave-private row pass on its own (register / ISA checks): tools/wcols_res.sh WROWS=1
#include "bds_acq_wrows.h"
template __global__ void bds::k_rows_wave_f<2, true, true>(bds::RowsFArgs);
template __global__ void bds::k_rows_wave_f<2, true, false>(bds::RowsFArgs);
