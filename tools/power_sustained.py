#!/usr/bin/env python3
"""Sustained power / clock measurement of the cfg3 search (VERDICT r4 "What's weak" 7): bds_acq_run in a loop for >= 20 s while a
sampler thread reads the GPU's power and engine clock at ~20 Hz -- from sysfs (hwmon power1_average / power1_input, freq1_input or
pp_dpm_sclk) when the container exposes it, else from `rocm-smi --showpower --showclocks --json` as fast as it answers.

    python tools/power_sustained.py [--seconds 25] [--prns 63] [--label default]     -> one JSON line

and the A/B the power-cap claim needs: the same loop under a lower clock ceiling (`--sclk-max MHz` tries
`rocm-smi --setextremum max sclk MHz`, then `--setperfdeterminism MHz`; if the container may not, it says so) -- when
the chip is power-limited, time per call does not follow the clock ceiling until the ceiling drops below the clock the
power cap allows."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sysfs_sources():
    """(power file, clock file) of every card the container shows -- a pool box exposes the sensors of all eight GPUs of its
    host while the process sees one of them: all are sampled, the card that draws the most power under the load is ours."""
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        pw = next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))), None)
        fq = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
        if pw:
            out.append((pw, fq))
    return out


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        j = json.loads(out[out.index("{"):])
        c = next(iter(j.values()))
        p = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k), None)
        s = next((v for k, v in c.items() if k.startswith("sclk clock speed")), None)
        return p, (float(s.strip("()Mhz")) if s else None)
    except Exception:  # noqa: BLE001
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=25.0)
    ap.add_argument("--prns", type=int, default=63)
    ap.add_argument("--label", default="default")
    ap.add_argument("--sclk-max", type=int, default=0)
    ap.add_argument("--env", action="append", default=[], help="K=V set before the context is created (switches of the test-hooks build)")
    args = ap.parse_args()
    notes = []
    if args.sclk_max:
        ok = False
        for cmd in (["rocm-smi", "--setextremum", "max", "sclk", str(args.sclk_max)], ["rocm-smi", "--setperfdeterminism", str(args.sclk_max)]):
            try:  # (rocm-smi asks for confirmation before it changes clocks: answered, never waited for)
                r = subprocess.run(cmd + ["--autorespond", "y"], capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=30)
            except subprocess.TimeoutExpired:
                notes.append("%s -> no answer within 30 s" % " ".join(cmd))
                continue
            notes.append("%s -> rc %d %s" % (" ".join(cmd), r.returncode, (r.stdout + r.stderr).strip().replace("\n", " | ")[-300:]))
            if r.returncode == 0 and "rror" not in r.stdout + r.stderr and "not" not in (r.stdout + r.stderr).lower():
                ok = True
                break
        notes.append("clock ceiling %s" % ("set" if ok else "NOT set (the container may not change it)"))
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
        os.environ.setdefault("BDS_LIB_PATH", os.path.join(ROOT, "bds-3-b1c-b2a-sdr-receiver_amd", "libbds_mi355x_hooks.so"))
    import bench
    import bds_amd

    s, x, _, _ = bench.build_workload("b1c")
    if args.prns != 63:
        s.acqSatelliteList = list(range(1, args.prns + 1))
    ctx = bds_amd.native.Context(0)
    ctx.acq_load(s, x)
    ctx.acq_prepare(s)
    ctx.acq_run(s)
    cards = sysfs_sources()
    samples, stop = [], threading.Event()

    def rd(path, scale):
        try:
            return float(open(path).read()) * scale
        except Exception:  # noqa: BLE001
            return None

    def sampler():
        while not stop.is_set():
            t = time.perf_counter()
            if cards:
                samples.append((t, [(rd(pw, 1e-6), rd(fq, 1e-6) if fq else None) for pw, fq in cards]))
                time.sleep(0.05)
            else:
                samples.append((t, [smi_sample()]))

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    calls, pair, clk = 0, [], []
    while time.perf_counter() - t0 < args.seconds:
        ctx.acq_run(s)
        tm = ctx.timing()
        pair.append(tm["cell_pair_ms"])
        calls += 1
    t1 = time.perf_counter()
    stop.set()
    th.join(timeout=10)
    ctx.close()
    # samples of the loaded interval only, the first second dropped (ramp); the card with the highest mean power is the one under load
    rows = [v for t, v in samples if t0 + 1.0 <= t <= t1]
    ncard = len(rows[0]) if rows else 0
    mean_p = [sum((r[c][0] or 0.0) for r in rows) / max(1, len(rows)) for c in range(ncard)]
    mine = max(range(ncard), key=lambda c: mean_p[c]) if ncard else 0
    ld = [r[mine] for r in rows if r[mine][0] is not None]
    ps = sorted(p for p, _ in ld)
    fs = sorted(f for _, f in ld if f is not None)
    pw = cards[mine][0] if cards else None
    med = lambda v: v[len(v) // 2] if v else None  # noqa: E731
    print(json.dumps({"label": args.label, "seconds": t1 - t0, "calls": calls, "ms_per_call": (t1 - t0) / calls * 1e3,
                      "pair_ms_mean": sum(pair) / len(pair), "pair_ms_first_last": [pair[0], pair[-1]],
                      "samples_under_load": len(ld), "sample_rate_Hz": len(ld) / max(1e-9, t1 - t0 - 1.0),
                      "power_W": {"min": ps[0] if ps else None, "median": med(ps), "max": ps[-1] if ps else None},
                      "sclk_MHz": {"min": fs[0] if fs else None, "median": med(fs), "max": fs[-1] if fs else None},
                      "source": ("sysfs %s (the hottest of %d cards visible; mean W per card: %s)" % (pw, ncard, [round(v) for v in mean_p])) if pw else "rocm-smi --json", "env": args.env, "notes": notes}))


if __name__ == "__main__":
    main()
