#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 (rocpd sqlite) results:
    python tools/pmc_summary.py gpurun_out/pmc/pass*_results.db
"""
import sqlite3
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) "
         "from counters_collection group by kernel_name, counter_name")
    for k, cn, v, n, dur in db.execute(q):
        k = k.split("(")[0].replace("void bds::", "").replace("bds::", "").replace("pfa::", "")
        acc[k][cn] = [v / max(n, 1), n]
        acc[k]["~duration_ns"] = [dur, n]
for k in sorted(acc):
    print(f"== {k}")
    for cn in sorted(acc[k]):
        v, n = acc[k][cn]
        print(f"   {cn:32s} {v:18.1f}  (avg per dispatch over {n})")
