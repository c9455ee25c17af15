#!/usr/bin/env python3
"""profiles/traffic_<workload>.json from a PMC summary (tools/pmc_run.sh -> gpurun_out/pmc_summary.txt): HBM bytes of one
launch pair (row pass + column pass of the default fp32-arithmetic search kernels) = FETCH_SIZE x 2 (gfx950: the
counter tallies 128-B requests as 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, averaged per dispatch.
    python tools/make_traffic.py gpurun_out/pmc_summary.txt b1c 201 profiles/r03_b1c_pmc.txt [column-kernel prefix]
"""
import json
import re
import shutil
import sys

src, workload, cells, keep = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
txt = open(src).read()
blocks = {b.split("\n")[0]: b for b in re.split(r"^== ", txt, flags=re.M) if b.strip()}


def counters(prefix):
    if prefix == "k_rows_":  # the row pass of the search: wave-private kernel when the plan has one, k_rows_inv_f otherwise
        prefix = "k_pfa_rows" if any(n.startswith("k_pfa_rows") for n in blocks) else "k_rows_wave_f" if any(n.startswith("k_rows_wave_f") for n in blocks) else "k_rows_inv_f"
    for name, b in blocks.items():
        if name.startswith(prefix):
            d = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([\d.]+)", b, re.M)}
            return name, d
    raise SystemExit(f"no kernel {prefix} in {src}")


rn, r = counters("k_rows_")
cn, c = counters(sys.argv[5] if len(sys.argv) > 5 else "k_cols_wave_f")
# FETCH_SIZE correction per kernel (MI355X_MICROARCH.md, HBM: "x 2 for wide coalesced streaming reads ... other access widths are
# uncalibrated: calibrate on a known byte count in your own access pattern").  argv[7] = factor of the column pass.  The N-point
# pair's column pass (k_pfa_cols) reads every byte of the inter-pass buffer exactly once -- 16.26 MB per cell by construction: that is
# the calibration.  With the buffer in tiles (a workgroup's item one contiguous 83 KB block) its raw FETCH_SIZE is 0.53 x that: the
# guide's factor 2 (1.06 x the known bytes).  With the first layout of round 6 (64-byte pieces 50 KB apart) the raw counter was
# 1.10 x the known bytes and the factor 1 (doubled it would have been 8.3 TB/s with the matrix instructions compiled out).
cf = float(sys.argv[7]) if len(sys.argv) > 7 else 2.0
total = 1024.0 * (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"] + cf * c["FETCH_SIZE"] + c["WRITE_SIZE"])
out = {"workload": workload, "cells_per_pair": cells, "round": (sys.argv[6] if len(sys.argv) > 6 else "round 4"), "bytes_per_pair": total,
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_run.sh), avg per dispatch of {rn} + {cn} "
                 f"(one dispatch pair = {cells} (PRN, bin) cells); FETCH_SIZE x 2 (row pass) / x {cf:g} (column pass) per the gfx950 correction and its "
                 f"calibration rule (MI355X_MICROARCH.md, HBM; tools/make_traffic.py); see {keep}",
       "rows": {"FETCH_SIZE_KiB": r["FETCH_SIZE"], "WRITE_SIZE_KiB": r["WRITE_SIZE"], "duration_us": r["~duration_ns"] / 1e3},
       "cols": {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"], "duration_us": c["~duration_ns"] / 1e3},
       "cols_fetch_factor": cf, "per_cell_MB": total / cells / 1e6}
json.dump(out, open(f"profiles/traffic_{workload}.json", "w"), indent=1)

shutil.copy(src, keep)
print(json.dumps(out, indent=1))
