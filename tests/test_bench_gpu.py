"""bench.py end to end: the one JSON line of the contract at N = 1, and the N > 1 path (self-spawn under
torch.distributed.run, (signal, PRN) job shards, one exchange per signal, max-over-ranks timing) with two ranks
sharing the one GPU of the test box over gloo (BDS_BENCH_TEST_ONE_DEVICE: RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

FAST = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-tracking", "--no-strict-f32"]


def run_bench(args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.pop("BDS_LIB_PATH", None)  # bench.py runs on the RELEASE library (the suite itself on the test-hooks build)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_line(d, n_gpus, prns=63):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "Msamples/s" and d["value"] > 0 and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["config"]["satellites_detected"] == [p for p in d["config"]["satellites_injected"] if p <= prns]


def test_bench_line_single():
    d = run_bench(["--workload", "b2a"] + FAST)
    check_line(d, 1)
    assert d["config"]["jobs_rank0"] == {"b2a": 63}
    cold = d["cold"]["b2a"]  # SURVEY 8d "also report cold": fresh context, load + prepare + first run
    assert cold["cold_total_ms"] > cold["warm_run_ms"] > 0 and cold["load_ms"] > 0 and cold["prepare_ms"] > 0
    assert abs(cold["cold_total_ms"] - (cold["load_ms"] + cold["prepare_ms"] + cold["first_run_ms"])) < 1e-6


def test_bench_two_ranks_self_spawn():
    one = run_bench(["--workload", "b2a"] + FAST)
    two = run_bench(["--workload", "b2a", "--gpus", "2"] + FAST, {"BDS_BENCH_TEST_ONE_DEVICE": "1"})
    check_line(two, 2)
    assert 0 < two["config"]["jobs_rank0"]["b2a"] < 63 and two["config"]["jobs"] == 63
    assert two["config"]["satellites_detected"] == one["config"]["satellites_detected"]
    assert two["cpu_baseline"] is None  # rank 0 at N = 1 only
    assert two["config"]["collective"] == {"backend": "gloo", "ranks": 2} and one["config"]["collective"] is None


def test_bench_joint_two_ranks():
    d = run_bench(["--workload", "joint", "--gpus", "2", "--prns", "6"] + FAST, {"BDS_BENCH_TEST_ONE_DEVICE": "1"})
    check_line(d, 2, prns=6)
    assert d["config"]["jobs"] == 12
    per = d["config"]["satellites_detected_per_signal"]
    assert set(per) == {"b1c", "b2a"}


def test_joint_workload_at_full_size_is_bit_equal_to_the_single_signal_runs():
    """BASELINE.json configs[4]: B1C + B2a jointly, 63 PRNs each = 126 (signal, PRN) jobs, on the HIP path at full size:
    one rank, and two ranks (LPT shards, one exchange per signal) sharing the one GPU of the box over gloo.  acqResults of
    each signal must be the bits of that signal's own single-device run (zero-filled shards summed: x + 0)."""
    b1c = run_bench(["--workload", "b1c"] + FAST)
    # (the default B1C run also times cfg2 -- extra key `b2a` -- and labels its kernels from the plan)
    assert b1c["b2a"]["ms_per_step"] > 0 and b1c["b2a"]["satellites_detected"] == b1c["b2a"]["satellites_injected"]
    assert "k_cols_small_f<80>" in b1c["b2a"]["kernel"] and "k_pfa_cols<636>" in b1c["roofline"]["kernel"]
    for k in ("frac_strict_f32", "frac_at_stored_bytes", "traffic_source"):
        assert k in b1c["roofline"], k
    b2a = run_bench(["--workload", "b2a"] + FAST)
    want = {"b1c": b1c["config"]["results_sha256"]["b1c"], "b2a": b2a["config"]["results_sha256"]["b2a"]}
    one = run_bench(["--workload", "joint"] + FAST)
    check_line(one, 1)
    assert one["config"]["jobs"] == 126 and one["config"]["jobs_rank0"] == {"b1c": 63, "b2a": 63}
    assert one["config"]["results_sha256"] == want
    two = run_bench(["--workload", "joint", "--gpus", "2"] + FAST, {"BDS_BENCH_TEST_ONE_DEVICE": "1"}, timeout=1500)
    check_line(two, 2)
    assert two["config"]["jobs"] == 126 and 0 < two["config"]["jobs_rank0"]["b1c"] < 63
    assert two["config"]["results_sha256"] == want
    assert sorted(two["config"]["satellites_detected_per_signal"]["b1c"]) == b1c["config"]["satellites_detected"]
