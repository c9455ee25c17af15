#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result as text:
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"{'calls':>8} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:8d} {total:14.1f} {avg:12.3f} {pct:7.2f}  {name[:150]}")
