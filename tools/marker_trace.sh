#!/bin/bash
# Per-stage timeline from the library's ROCTx ranges (csrc/bds_internal.h RoctxRange): rocprofv3 --marker-trace of a short bench run
# and of a short tracking run (GPU box).  -> gpurun_out/markers.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_mark
timeout 600 rocprofv3 --marker-trace --kernel-trace -d gpurun_out/prof_mark -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-b2a --no-cold > gpurun_out/prof_mark.log 2>&1
echo "rc=$?"
db=$(find gpurun_out/prof_mark -name "*_results.db" | head -1)
python - "$db" > gpurun_out/markers.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
import json
print("# rocprofv3 --marker-trace: ROCTx ranges of libbds_mi355x (csrc/bds_internal.h RoctxRange), python bench.py --steps 2 --warmup 1")
print("# (host-side ranges: 'acq.search' is the ENQUEUE of the launch pairs, the device time of a call ends inside 'acq.refine', which waits for it)")
agg = collections.OrderedDict()
for row in db.execute("select * from regions order by start"):
    msg = None
    for v in row:
        if isinstance(v, str) and '"message"' in v:
            msg = json.loads(v).get("message")
    if msg is None:
        continue
    start, end = [v for v in row if isinstance(v, int) and v > 10**12][:2]
    a = agg.setdefault(msg, [0, 0.0])
    a[0] += 1
    a[1] += (end - start) / 1e6
print(f"{'range':40s} {'count':>6s} {'total ms':>12s} {'avg ms':>10s}")
for k, (n, ms) in agg.items():
    print(f"{str(k)[:40]:40s} {n:6d} {ms:12.3f} {ms / n:10.3f}")
PY
find gpurun_out/prof_mark -name "*.db" -size +20M -delete
cat gpurun_out/markers.txt
