#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 (rocpd sqlite) result: start offset, duration, gap to the previous
kernel's end, grid -- what a stage of a call spends between its kernels.
    python tools/rocprof_timeline.py gpurun_out/prof_b2a/b2a_results.db [last_n] [anchor kernel substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
anchor = sys.argv[3] if len(sys.argv) > 3 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
if anchor:
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    i0 = idx[-1] if idx else max(0, len(rows) - n)
    rows = rows[max(0, i0 - 2):i0 - 2 + n]
else:
    rows = rows[-n:]
t0, prev = rows[0][1], rows[0][1]
print(f"{'t_us':>10} {'dur_us':>9} {'gap_us':>8} {'grid':>14}  kernel")
for name, s, e, gx, gy, wx in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {(s - prev) / 1e3:8.1f} {str(gx // max(wx, 1)) + 'x' + str(gy):>14}  {name[:110]}")
    prev = e
