"""Rough VGPR pressure profile of one kernel's ISA (linear scan, loop-carried registers counted as live
everywhere): where the peak is and which phase it belongs to.  usage: isa_pressure.py file.s [kernel-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "k_cols_wave_f"
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and pat in l and l.rstrip().endswith(":") or (pat in l and "; @" in l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
first, last = {}, {}
defs = {}
for i, l in enumerate(body):
    l = l.split(";")[0]
    if not l.strip() or l.strip().startswith("."):
        continue
    ops = l.strip().split(None, 1)
    if len(ops) < 2:
        continue
    args = ops[1].split(",")
    for k, a in enumerate(args):
        for m in reg.finditer(a):
            rs = [int(m.group(1))] if m.group(1) else range(int(m.group(2)), int(m.group(3)) + 1)
            for r in rs:
                first.setdefault(r, i)
                last[r] = i
                if k == 0 and not ops[0].startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "v_cmp", "s_")):
                    defs.setdefault(r, []).append(i)
# a register redefined many times is a temporary: split its range at each def (live from def to last use before next def)
events = []
for r in first:
    ds = sorted(set(defs.get(r, [first[r]])))
    if ds[0] > first[r]:
        ds = [first[r]] + ds
    uses = []
    for i, l in enumerate(body[first[r]:last[r] + 1], first[r]):
        if re.search(r"\bv%d\b" % r, l) or any(int(m.group(1)) <= r <= int(m.group(2)) for m in re.finditer(r"v\[(\d+):(\d+)\]", l)):
            uses.append(i)
    for j, d in enumerate(ds):
        nxt = ds[j + 1] if j + 1 < len(ds) else last[r] + 1
        u = [x for x in uses if d <= x <= nxt]
        if u:
            events.append((d, max(u)))
n = len(body)
live = [0] * (n + 1)
for a, b in events:
    live[a] += 1
    live[b + 1 if b + 1 <= n else n] -= 1
cur, prof = 0, []
for i in range(n):
    cur += live[i]
    prof.append(cur)
marks = [i for i, l in enumerate(body) if "s_barrier" in l]
print("lines", n, "barriers at", marks)
step = max(1, n // 60)
for i in range(0, n, step):
    seg = prof[i:i + step]
    print(f"{i:5d} max {max(seg):4d}  {'#' * (max(seg) // 4)}")
pk = max(range(n), key=lambda i: prof[i])
print("peak", prof[pk], "at line", pk, body[pk].strip())
