#!/usr/bin/env python3
"""Acquisition throughput benchmark (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload b1c|b2a|joint]

A *step* is one complete acquisition pass of the hot path over one synthetic IF
block: forward transforms of every Doppler bin, the PRN x Doppler parallel
code-phase search, f64 refinement and the fine-Doppler search, with the int8 IF
block already resident in HBM and the code spectra cached (bds_acq_load /
bds_acq_prepare are outside the timed region).

Workload at N=1 (default): BASELINE.json configs[2] -- BDS-3 B1C full acquisition,
63 PRNs x 201 Doppler bins, 10 ms coherent data+pilot, fs = 99.375 MS/s,
IF = 14.58 MHz -- the configuration the north star quotes its roofline target on.
``--workload b2a`` runs configs[1] (B2a, 63 PRNs x 26 bins).

``--workload joint`` runs configs[4]: the B1C and the B2a block together, 126 (signal, PRN) jobs.

N > 1 (one rank per GPU under torch.distributed.run; ``--gpus N`` without a launcher
re-executes itself under it on 127.0.0.1): the (signal, PRN) jobs are spread over the ranks by
cost (bds_shard_jobs, longest-processing-time rule; round-robin for one signal); every
step ends with one RCCL all-reduce(SUM) per signal of the three per-PRN result vectors
(3 x 63 f64), the only exchange the path has.  Total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_workload(name):
    import bds_amd
    from bds_amd import synth

    prns_present = [1, 4, 9, 14, 19, 20, 27, 35, 46, 58]
    rng = np.random.default_rng(3550)
    if name == "b1c":
        s = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)),
                                      acqCohT=10, pilotACQflag=1)
        spc = 993750
        n_samples = 20 * spc  # B1C/postProcessing.m:94
        label = "BDS-3 B1C full acquisition: 63 PRNs x 201 Doppler bins, 10 ms coherent data+pilot, fs=99.375 MS/s"
    else:
        s = bds_amd.init_settings_b2a(acqSatelliteList=list(range(1, 64)))
        spc = 99375
        n_samples = 17 * spc  # (fineNoncoh+2)*spc, B2a/postProcessing.m:89-90
        label = "BDS-3 B2a full acquisition: 63 PRNs x 26 Doppler bins (+-5 kHz / 400 Hz), 1 ms code, fs=99.375 MS/s"
    sats = synth.random_sats(rng, prns_present, spc, cn0_dbhz=45.0)
    # only the first N + spc samples can be touched by acquisition;
    # the tail is noise-only to keep generation fast
    head = min(n_samples, 4 * spc if name == "b1c" else n_samples)
    x = np.empty(n_samples, dtype=np.int8)
    synth.make_if(s, sats, head, seed=3550, out=x[:head])
    if head < n_samples:
        x[head:] = np.clip(np.rint(rng.normal(0, 20.0, n_samples - head)), -127, 127).astype(np.int8)
    return s, x, sats, label


def cpu_baseline(s, x, name, budget_s=20.0):
    """The oracle (NumPy/SciPy float64 restatement = 'port') timed on the host cores on a
    bounded sample of the same workload: whole Doppler rows of one PRN until the budget is used."""
    cores = os.cpu_count() or 1
    os.environ["BDS_ORACLE_FFT_WORKERS"] = str(cores)
    from oracle import acquisition as oacq

    gen = oacq.b1c_coarse_rows if name == "b1c" else oacq.b2a_coarse_rows
    xf = x.astype(np.float64)
    t0 = time.perf_counter()
    cells = 0
    n = None
    for prn in range(1, 64):
        for b, row in gen(xf, s, prn):
            n = row.size
            cells += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": cells * n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cells} (PRN, Doppler-bin) cells of the same block (code-spectrum FFTs included), "
                      f"{dt:.1f} s, scipy.fft workers={cores}; float64 NumPy restatement of acquisition.m, not MATLAB"}


def tracking_leg(name, local_rank, base):
    """Second half of the hot path, reported beside the headline metric (not part of `value`):
    closed-loop tracking of 12 channels at 99.375 MS/s on a synthetic int8 record resident in HBM
    (BASELINE.json configs[3] shape, shortened): B1C wide-band, 10-ms epochs / B2a, 1-ms epochs.
    The record carries the 12 satellites, so the loops LOCK: one block of whole code periods (Dopplers on the
    fs / block grid: whole carrier cycles per block) is generated and repeated to the record's length."""
    from types import SimpleNamespace

    import bds_amd
    from bds_amd import synth

    # same front end as the acquisition workload (fs = 99.375 MS/s)
    if name == "b1c":
        epochs, mode, periods = 60, "WB", 2
        s = base.copy(msToProcess=epochs * 10, numberOfChannels=12, pilotTRKflag=2)
        dopplers = [-1500, -1000, -750, -500, -250, -100, 100, 250, 500, 750, 1000, 1500]  # 50-Hz grid (20-ms block)
        macs = 2 + 2 * 9
    else:
        epochs, mode, periods = 600, "B2A", 20
        s = base.copy(msToProcess=epochs, numberOfChannels=12)
        # B2a/tracking.m has no code-rate aiding: its DLL only holds small Dopplers (tests/golden/make_long_tracking.py)
        dopplers = [-100, -100, -50, -50, -50, 0, 0, 50, 50, 50, 100, 100]  # 50-Hz grid (20-ms block)
        macs = 2 + 2 * 6
    spc = int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))
    rng = np.random.default_rng(1)
    prns = list(range(1, 13))
    sats = [synth.Sat(p, float(d), float(rng.uniform(0.2, 0.8)), float(rng.uniform(0, 2 * np.pi)), 47.0) for p, d in zip(prns, dopplers)]
    block = synth.make_if(s, sats, periods * spc, seed=7)
    shift = 12345
    n = (epochs + 3) * spc + shift
    x = np.tile(np.roll(block, shift), -(-n // block.size))[:n]
    ch = []
    for sat in sats:
        cf = s.IF + round(sat.doppler / 25) * 25
        code_freq = s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis if name == "b1c" else s.codeFreqBasis
        ch.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(shift + int(np.ceil(sat.delay)) + 1),
                                  codeFreq=float(code_freq), status="T"))
    ctx = bds_amd.get_context(local_rank)
    bds_amd.tracking(x, ch, s, mode=mode)  # warm-up (H2D, code tables)
    res, _ = bds_amd.tracking(x, ch, s, mode=mode)
    dev_ms = ctx.timing()["total_ms"]
    samples = float(sum(np.diff(r.absoluteSample).sum() + spc for r in res))
    epoch_s = 0.010 if name == "b1c" else 0.001
    assert all(r.completed == epochs for r in res), [r.completed for r in res]
    half = epochs // 2
    locked = sum(bool(np.abs(r.I_P[half:]).mean() > 3 * np.abs(r.Q_P[half:]).mean()) for r in res)
    return {"mode": mode, "channels": 12, "epochs": epochs, "fs_MHz": s.samplingFreq / 1e6, "ms_per_epoch": dev_ms / epochs,
            "channel_Msamples_per_s": samples / dev_ms / 1e3, "x_realtime_12ch": epoch_s * epochs / (dev_ms * 1e-3),
            "int8_read_GBps": samples / (dev_ms * 1e-3) / 1e9,  # one byte per sample per channel (algorithmic)
            "correlator_GMACs": samples * macs / (dev_ms * 1e-3) / 1e9, "macs_per_sample": macs,
            "channels_locked": locked,
            "note": "device time of the epoch loop (one launch per epoch: correlate + the previous epoch's loop update; record window in HBM); locked synthetic record "
                    "(12 satellites at 47 dB-Hz, one block of whole code periods repeated)"}


def fast_path_leg(local_rank, s, x, n_cells_samples):
    """Extra key, never `value`: the same call with the opt-in packed-fp16 search arithmetic (BDS_ACQ_HMATH=1; the f64
    refinement still decides every result).  Narrower arithmetic than the reference's, so it earns no headline."""
    import bds_amd

    os.environ["BDS_ACQ_HMATH"] = "1"
    try:
        c = bds_amd.native.Context(local_rank)  # the knobs are read once, at context creation
    finally:
        del os.environ["BDS_ACQ_HMATH"]
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        c.acq_run(s)
        t0 = time.perf_counter()
        c.acq_run(s)
        dt = time.perf_counter() - t0
        tm = c.timing()
    finally:
        c.close()
    return {"dtype": "f16", "half_storage": int(tm["half_storage"]), "ms_per_step": dt * 1e3, "value": n_cells_samples / dt / 1e6,
            "unit": "Msamples/s", "pair_ms": tm["cell_pair_ms"],
            "note": "packed v_pk_*_f16 search arithmetic + f64 refinement; NOT the headline (narrower than the reference's single/double)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="b1c", choices=["b1c", "b2a", "joint"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracking", action="store_true", help="skip the (untimed) tracking leg")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the extra (never the headline) packed-fp16 sieve timing")
    ap.add_argument("--prns", type=int, default=63, help="tuning only: search PRNs 1..N instead of all 63")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become N ranks, one per GPU, over RCCL (rendezvous on 127.0.0.1)
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    xdev = "cuda"  # where the exchanged tensors live
    if world > 1:
        import torch.distributed as dist

        if os.environ.get("BDS_BENCH_TEST_ONE_DEVICE"):
            # test hook (tests/test_bench_gpu.py): every rank on device 0 of a one-GPU box, exchange over gloo --
            # exercises the launcher, the job shards and the exchange; RCCL refuses two ranks on one device
            local_rank, xdev = 0, "cpu"
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import bds_amd

    names = ["b1c", "b2a"] if args.workload == "joint" else [args.workload]
    sigs = []  # per signal: settings, block, injected satellites, label, this rank's PRN shard, its context
    for nm in names:
        s, x, sats, label = build_workload(nm)
        if args.prns != 63:
            s.acqSatelliteList = list(range(1, args.prns + 1))
            label += f" [TUNING RUN: PRNs 1..{args.prns} only]"
        sigs.append(dict(name=nm, s=s, x=x, sats=sats, label=label))
    shards = bds_amd.shard_joint([g["s"] for g in sigs], rank, world)  # (signal, PRN) jobs of this rank, by cost
    for k, g in enumerate(sigs):
        g["shard"] = shards[k]
        # one context per signal: each keeps its IF block and code spectra resident in HBM across steps
        g["ctx"] = bds_amd.get_context(local_rank) if k == 0 else bds_amd.native.Context(local_rank)
        if g["shard"]:
            g["ctx"].acq_load(g["s"], g["x"])
            g["ctx"].acq_prepare(g["s"])
    s, x, sats, label = sigs[0]["s"], sigs[0]["x"], sigs[0]["sats"], " + ".join(g["label"] for g in sigs)
    all_prns = [int(p) for p in s.acqSatelliteList]
    ctx = sigs[0]["ctx"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        out = []
        for g in sigs:
            max_prn = max(int(p) for p in g["s"].acqSatelliteList)
            if g["shard"]:
                carr, cph, pm, det = g["ctx"].acq_run(g["s"], g["shard"])
                g["tim"] = g["ctx"].timing()
            else:  # no job of this signal on this rank: zeros into the exchange
                carr = cph = pm = np.zeros(max_prn)
            if dist is not None:
                buf = torch.from_numpy(np.stack([carr, cph, pm])).to(xdev)
                dist.all_reduce(buf)  # RCCL all-reduce(SUM) of 3 x max_prn f64 per signal: x + 0 is exact
                carr, cph, pm = buf.cpu().numpy()
            out.append((carr, cph, pm))
        return out

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    tim = []
    for _ in range(args.steps):
        res_all = step()
        res = res_all[0]
        tim.append(sigs[0].get("tim") or ctx.timing())
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    tm = tim[-1]
    n_circ, n_bins, ncomp = tm["n_circ"], tm["n_bins"], tm["n_comp"]
    p_total = len(all_prns)
    ms_per_step = dt / args.steps * 1e3

    def sizes(st):  # N (circular correlation length), Doppler bins, components of one signal's search
        spc = int(np.floor(st.samplingFreq / (st.codeFreqBasis / st.codeLength) + 0.5))
        if st.signal.upper() == "B1C":
            n = int(np.floor(spc / 10 * (10 + st.acqCohT) + 0.5))
            nc = 2 if st.pilotACQflag == 1 else 1
        else:
            n, nc = 2 * spc, 2
        return n, int(np.floor(st.acqSearchBand * 2 / st.acqStep + 0.5)) + 1, nc

    cell_samples = 0.0  # IF samples through (PRN, bin) cells, all signals
    b_alg = 0.0         # algorithmic bytes of the whole job
    for g in sigs:
        n_g, d_g, nc_g = sizes(g["s"])
        p_g = len(set(int(p) for p in g["s"].acqSatelliteList))
        cell_samples += float(n_g) * p_g * d_g
        b_alg += 9.0 * n_g * d_g + 8.0 * (1 + nc_g) * n_g * p_g * d_g
    assert (n_circ, n_bins, ncomp) == sizes(s)
    cell_msps = cell_samples / (dt / args.steps) / 1e6
    # algorithmic bytes: per cell 8*(1+ncomp)*N (signal + code spectra, fp32 complex);
    # per bin 9*N (int8 in, spectrum out).  One launch pair (rows+cols kernels) = cells_per_pair cells.
    pair_ms = float(np.mean([t["cell_pair_ms"] for t in tim]))
    cells_per_pair = tm["cells_per_pair"]
    bytes_per_pair = cells_per_pair * 8 * (1 + ncomp) * n_circ
    achieved = bytes_per_pair / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0

    # HBM traffic per launch pair: measured separately with rocprofv3 --pmc (FETCH_SIZE x2 gfx950
    # correction + WRITE_SIZE, tools/pmc_run.sh) and committed under profiles/; null if no
    # measurement exists for this workload / cells-per-pair.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{names[0]}.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if int(tj.get("cells_per_pair", -1)) == int(cells_per_pair):
            traffic = tj["bytes_per_pair"] / 1e9  # GB per launch pair

    detected = sorted(int(p) for p in np.nonzero(res[0])[0] + 1)
    out = {
        "metric": "IF Msamples/s through acquisition (all PRNs x Doppler bins)",
        "value": cell_msps,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": {0: "f32", 1: "f32", 2: "f16"}[int(tm.get("half_storage", 0))],
        "dtype_detail": {0: "f32 search + f64 refinement", 1: "f32 search on f16-stored spectra + f64 refinement",
                         2: "f16 search (packed v_pk_*_f16, f32 magnitudes) + f64 refinement of every candidate"}[int(tm.get("half_storage", 0))],
        "data": "synthetic",
        "config": {"workload": label, "prns": p_total, "doppler_bins": n_bins, "n_circ": n_circ,
                   "fft_len": tm["fft_len"], "components": ncomp,
                   "parallelism": f"(signal, PRN) job shard x{world}, LPT by cost; one all-reduce(SUM) of 3 x 63 f64 per signal",
                   "jobs": sum(len(set(int(p) for p in g["s"].acqSatelliteList)) for g in sigs),
                   "jobs_rank0": {g["name"]: len(g["shard"]) for g in sigs},
                   "collective": ({"backend": dist.get_backend(), "ranks": dist.get_world_size()} if dist is not None else None),
                   "satellites_injected": sorted(sat.prn for sat in sats), "satellites_detected": detected},
        "block_msps": n_circ / (dt / args.steps) / 1e6,
        "whole_job_algorithmic_GBps": b_alg / (dt / args.steps) / 1e9,
        "whole_job_frac_of_hbm_peak": b_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        "stage_ms": {k: float(np.mean([t[k] for t in tim])) for k in ("total_ms", "forward_ms", "search_ms", "refine_ms")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_unit": "GB per launch pair (PMC)", "algorithmic_GB_per_pair": bytes_per_pair / 1e9,
                     "kernel": "row-pass + column-pass launch pair (k_rows_inv_* + k_cols_inv_max_*; one pair = %d (PRN, bin) cells)" % int(cells_per_pair),
                     "pair_ms": pair_ms, "rows_ms": tm.get("rows_ms"), "cols_ms": tm.get("cols_ms"), "n_extra": tm.get("n_extra"),
                     "storage": "fp16 complex" if tm.get("half_storage") else "fp32 complex",
                     # `achieved` prices every spectrum element at the 8 bytes of SURVEY.md 8d's model (fp32 complex); the
                     # same count at the element size the kernels really store, for comparison with `traffic`
                     "achieved_at_stored_element_size": achieved * (0.5 if tm.get("half_storage") else 1.0)},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s, x, names[0])
        else:
            out["cpu_baseline"] = None
        out["tracking"] = tracking_leg(names[0], local_rank, s) if world == 1 and not args.no_tracking else None
        out["fast_path"] = (fast_path_leg(local_rank, s, x, float(n_circ) * p_total * n_bins)
                            if world == 1 and len(sigs) == 1 and not args.no_fast_path else None)
        if len(sigs) > 1:
            out["config"]["satellites_detected_per_signal"] = {g["name"]: sorted(int(p) for p in np.nonzero(r[0])[0] + 1)
                                                                for g, r in zip(sigs, res_all)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
