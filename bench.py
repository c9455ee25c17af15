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


_CPU = {}


def _cpu_init(block_path, settings_dict, name):
    """Worker process of the CPU baseline (spawned: no GPU state, no GIL or allocator shared with its siblings)."""
    from types import SimpleNamespace

    from oracle import acquisition as oacq

    _CPU["x"] = np.load(block_path).astype(np.float64)
    _CPU["s"] = SimpleNamespace(**settings_dict)
    _CPU["gen"] = oacq.b1c_coarse_rows if name == "b1c" else oacq.b2a_coarse_rows


def _cpu_rows(job):
    prn, b0, b1 = job
    if b1 <= b0:
        return 0
    return sum(1 for _ in _CPU["gen"](_CPU["x"], _CPU["s"], prn, bins=range(b0, b1)))


def cpu_baseline(s, x, name, budget_s=15.0):
    """The oracle (NumPy/SciPy float64 restatement = 'port', not MATLAB) timed on the host cores on a bounded sample of
    the same workload: whole (PRN, Doppler-bin) cells of the same block, code-spectrum FFTs included.
    Two figures: ONE core (a few cells, in this process) and ALL cores -- one spawned worker process per core, each running
    the oracle's whole row pipeline (carrier, forward FFT, two spectrum products, two inverse FFTs, magnitudes) on its share
    of a PRN's Doppler rows.  (scipy.fft's `workers` alone does not do it -- a single 1-D transform does not parallelise,
    which is why the round-2 figure was a one-core number under a 256-core label -- and a thread per row scaled 6x on 201
    threads: the interpreter lock and the allocator serialise the NumPy temporaries.)  `cores` = worker processes used = the
    CPUs this process may use (affinity and cgroup quota), not the cores the host shows."""
    import multiprocessing as mp
    import tempfile

    from oracle import acquisition as oacq

    host_cores = os.cpu_count() or 1
    gen = oacq.b1c_coarse_rows if name == "b1c" else oacq.b2a_coarse_rows
    n_bins = len(oacq.freq_bins(s))
    xf = x.astype(np.float64)
    t0 = time.perf_counter()
    cells1, n = 0, None
    for b, row in gen(xf, s, 1):
        n = row.size
        cells1 += 1
        if cells1 >= 4 or time.perf_counter() - t0 > 3.0:
            break
    dt1 = time.perf_counter() - t0
    one_core = cells1 * n / dt1 / 1e6
    del xf
    # CPUs this process may really use: the scheduler affinity and the cgroup CPU quota (a container that shows 256 cores
    # under a 16-CPU quota gets 16 cores' worth of time; more workers than that only throttle each other -- measured on the
    # GPU box of this build's pool: 118 transforms/s with 32 workers, 41 with 256)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    if quota:
        usable = min(usable, max(1, int(np.ceil(quota))))
    procs = max(1, min(usable, 256))
    fd, path = tempfile.mkstemp(suffix=".npy")
    os.close(fd)
    cells, dt = 0, 1.0
    try:
        np.save(path, np.ascontiguousarray(x[:2 * n + 16]))  # (only the head of the block is ever read by the coarse search)
        sd = {k: v for k, v in vars(s).items() if not k.startswith("_")}
        with mp.get_context("spawn").Pool(procs, initializer=_cpu_init, initargs=(path, sd, name)) as pool:
            pool.map(_cpu_rows, [(1, 0, 0)] * procs)  # every worker is up (imports done, block loaded)
            per = max(1, -(-n_bins // procs))
            prns_per_round = max(1, procs // max(1, -(-n_bins // per)))  # several PRNs per round when bins < workers
            t0 = time.perf_counter()
            prn = 1
            while prn <= 63 and time.perf_counter() - t0 < budget_s:
                jobs = [(p, b0, min(n_bins, b0 + per)) for p in range(prn, min(64, prn + prns_per_round)) for b0 in range(0, n_bins, per)]
                cells += sum(pool.map(_cpu_rows, jobs, chunksize=1))
                prn += prns_per_round
            dt = time.perf_counter() - t0
    finally:
        os.remove(path)
    all_cores = cells * n / dt / 1e6
    out = {"value": all_cores, "unit": "Msamples/s", "cores": procs, "host_cores": host_cores, "cgroup_cpu_quota": quota, "kind": "port",
           "impl": "NumPy / SciPy (oracle/acquisition.py)",
           "one_core_value": one_core, "all_core_speedup": all_cores / one_core,
           "sample": f"{cells} (PRN, Doppler-bin) cells of the same block in {dt:.1f} s on {procs} worker processes (each a share of a PRN's "
                     f"Doppler rows, code-spectrum FFTs included) + {cells1} cells in {dt1:.1f} s on one core; "
                     "float64 NumPy restatement of acquisition.m, not MATLAB"}
    # SURVEY 8d's second CPU figure: the COMPILED float64 restatement (oracle/c/acq_oracle.c: own mixed-radix transform, OpenMP over
    # the Doppler bins) on the same block -- the whole Doppler row of one PRN when that fits the budget.  The faster of the two
    # figures is `value`; both are kept.
    try:
        from oracle import cfast

        if cfast.available():
            xf = x[:2 * n + 16].astype(np.float64)
            t0 = time.perf_counter()
            cfast.coarse_rows(xf, s, 1, bins=range(0, 1), threads=1)
            t_one = time.perf_counter() - t0  # plan + code spectra + one row
            est = t_one * n_bins / procs
            nb = n_bins if est < 1.5 * budget_s else max(procs, int(n_bins * 1.5 * budget_s / est) // procs * procs)
            t0 = time.perf_counter()
            cfast.coarse_rows(xf, s, 1, bins=range(0, nb), threads=procs)
            dtc = time.perf_counter() - t0
            c_val = nb * n / dtc / 1e6
            c_leg = {"value": c_val, "unit": "Msamples/s", "cores": procs, "kind": "port", "impl": "C, OpenMP (oracle/c/acq_oracle.c)",
                     "sample": f"{nb} of the {n_bins} Doppler rows of PRN 1 on the same block in {dtc:.1f} s on {procs} OpenMP threads (plan and "
                               "code-spectrum transforms of the PRN included); compiled float64 restatement of acquisition.m with its own "
                               "mixed-radix transform, not MATLAB"}
            out["c_port"] = c_leg
            if c_val > out["value"]:
                numpy_leg = {k: out[k] for k in ("value", "cores", "impl", "sample", "one_core_value", "all_core_speedup")}
                out.update({k: c_leg[k] for k in ("value", "cores", "impl", "sample")})
                out.pop("one_core_value"), out.pop("all_core_speedup")  # (NumPy figures: they stay in numpy_port)
                out["numpy_port"] = numpy_leg
    except Exception as e:  # noqa: BLE001  (the C leg is an extra; the NumPy figure stands without it)
        out["c_port"] = {"error": repr(e)}
    return out


def track_record(name, base, epochs=None):
    """Synthetic int8 record + channels for 12-channel closed-loop tracking at the workload's sampling rate: one block of whole code
    periods carrying the 12 satellites (Dopplers on the fs / block grid: whole carrier cycles per block), repeated to the record's
    length -- the loops LOCK.  Returns (settings, record, channels as preRun would hand them over, mode, MACs per sample)."""
    from types import SimpleNamespace

    from bds_amd import synth

    if name == "b1c":
        epochs, mode, periods = epochs or 60, "WB", 2
        s = base.copy(msToProcess=epochs * 10, numberOfChannels=12, pilotTRKflag=2)
        dopplers = [-1500, -1000, -750, -500, -250, -100, 100, 250, 500, 750, 1000, 1500]  # 50-Hz grid (20-ms block)
        macs = 2 + 2 * 9
    else:
        epochs, mode, periods = epochs or 600, "B2A", 20
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
    return s, x, ch, mode, macs, epochs, spc


def tracking_leg(name, local_rank, base):
    """Second half of the hot path, reported beside the headline metric (not part of `value`):
    closed-loop tracking of 12 channels at 99.375 MS/s on a synthetic int8 record resident in HBM
    (BASELINE.json configs[3] shape, shortened): B1C wide-band, 10-ms epochs / B2a, 1-ms epochs (track_record)."""
    import bds_amd

    s, x, ch, mode, macs, epochs, spc = track_record(name, base)
    ctx = bds_amd.get_context(local_rank)
    # the fp32 carrier / fp32 prefix sums of rounds 2-4 (BDS_TRK_PREC=0: SURVEY 8d only up to its first ceil() flip,
    # tests/test_track_long_gpu.py), timed beside the default for comparison
    os.environ["BDS_TRK_PREC"] = "0"
    try:
        ctx.reload_tuning()
        bds_amd.tracking(x, ch, s, mode=mode)
        bds_amd.tracking(x, ch, s, mode=mode)
        fast_ms = ctx.timing()["total_ms"]
    finally:
        del os.environ["BDS_TRK_PREC"]
        ctx.reload_tuning()
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
            "numerics": "strict (default, BDS_TRK_PREC=4): sin / cos of the reference's own trigarg(k) per sample in f64, f64 prefix sums -- 1e-13 of |P| from the "
                        "float64 oracle; SURVEY 8d until a channel's first ceil() flip (cfg4 at full rate: 11 of 12 channels over all 3 600 epochs, "
                        "profiles/r05_cfg4_full_vs_c_oracle.txt)",
            "fp32_carrier_ms_per_epoch": fast_ms / epochs,
            "fp32_carrier_note": "BDS_TRK_PREC=0: fp32 carrier recurrence + fp32 prefix sums; 8d tolerances hold until its first ceil() flip (epoch 142 / 1 588), a bounded floor after it",
            "note": "device time of the epoch loop (one launch per epoch: correlate + the previous epoch's loop update; record window in HBM); locked synthetic record "
                    "(12 satellites at 47 dB-Hz, one block of whole code periods repeated)"}


def cfg4_record(base, epochs=3600, realisations=32):
    """BASELINE.json configs[3]: B1C wide-band tracking, 12 channels x 36 000 ms at 99.375 MS/s.  The record is built from
    20-ms blocks: the 12 satellites' signal (47 dB-Hz, Dopplers on the 50-Hz grid and the code at its nominal rate: whole carrier
    cycles and code periods per block, so it continues seamlessly from block to block and the loops lock and stay locked) plus one of `realisations`
    independent noise realisations (sigma = 20 LSB), the blocks following each other in a seeded random order -- a single
    repeated block would make the noise periodic and the variance-based C/N0 estimator meaningless.
    Returns settings, channels (as preRun would hand them over), the int8 blocks [realisations][2 spc], their order in the
    record, the shift of the record against the block grid, and the record length in samples."""
    from types import SimpleNamespace

    from bds_amd import synth

    s = base.copy(msToProcess=epochs * 10, numberOfChannels=12, pilotTRKflag=2)
    dopplers = [-1500, -1000, -750, -500, -250, -100, 100, 250, 500, 750, 1000, 1500]
    spc = int(np.floor(s.samplingFreq / (s.codeFreqBasis / s.codeLength) + 0.5))
    rng = np.random.default_rng(1)
    sats = [synth.Sat(p, float(d), float(rng.uniform(0.2, 0.8)), float(rng.uniform(0, 2 * np.pi)), 47.0)
            for p, d in zip(range(1, 13), dopplers)]
    clean = synth.make_if(s, sats, 2 * spc, seed=7, clean=True, code_doppler=False, pilot61_secondary=True)  # code at the nominal rate: seamless repetition
    blocks = np.empty((realisations, 2 * spc), dtype=np.int8)
    for k in range(realisations):
        blocks[k] = np.clip(np.rint(clean + rng.normal(0.0, 20.0, clean.size)), -127, 127).astype(np.int8)
    shift = 12345
    n = (epochs + 3) * spc + shift
    order = rng.integers(0, realisations, size=n // (2 * spc) + 2)
    ch = []
    for sat in sats:
        cf = s.IF + round(sat.doppler / 25) * 25
        ch.append(SimpleNamespace(PRN=sat.prn, acquiredFreq=float(cf), codePhase=float(shift + int(np.ceil(sat.delay)) + 1),
                                  codeFreq=float(s.codeFreqBasis - (cf - s.IF) / s.carrFreqBasis * s.codeFreqBasis), status="T"))
    return s, ch, blocks, order, shift, n, spc


def record_bytes(blocks, order, shift, n):
    """The record in memory (tests at reduced length): block order[0] starts `shift` samples into the file (the file begins with
    the tail of a block), then order[1], ..."""
    blen = blocks.shape[1]
    stream = np.concatenate([blocks[order[-1]][blen - shift:]] + [blocks[k] for k in order[:(n - shift) // blen + 1]])
    return stream[:n]


def write_record(path, blocks, order, shift, n):
    """The same record as a raw int8 file (what the reference's fid points at)."""
    blen = blocks.shape[1]
    left = n
    with open(path, "wb") as f:
        head = blocks[order[-1]][blen - shift:]
        f.write(head.tobytes())
        left -= head.size
        for k in order:
            if left <= 0:
                break
            m = min(left, blen)
            f.write(blocks[k][:m].tobytes())
            left -= m
    assert left == 0


def tracking_full_leg(local_rank, base, epochs=3600, path=None):
    """Extra key: BASELINE.json configs[3] as one piece -- 12 channels x 3 600 ten-millisecond epochs x 99.375 MS/s from a raw
    int8 FILE, WALL time of the call: reading the window of the record the channels can touch (3.6 GB) into HBM, the
    3 600-launch epoch loop, C/N0 on the device and the result arrays back on the host."""
    import tempfile

    import bds_amd

    s, ch, blocks, order, shift, n, spc = cfg4_record(base, epochs)
    own = path is None
    if own:
        fd, path = tempfile.mkstemp(prefix="bds_cfg4_", suffix=".bin", dir=os.environ.get("BDS_BENCH_TMP", tempfile.gettempdir()))
        os.close(fd)
    try:
        t0 = time.perf_counter()
        write_record(path, blocks, order, shift, n)
        t_write = time.perf_counter() - t0
        ctx = bds_amd.get_context(local_rank)
        t0 = time.perf_counter()
        res, _ = bds_amd.tracking(path, ch, s, mode="WB")
        wall = time.perf_counter() - t0
        dev_ms = ctx.timing()["total_ms"]
        # CPU baseline of this leg: the oracle -- its sample loops in C, one thread per channel
        # (oracle/c/trk_oracle.c; the loop filters in oracle/tracking.py) -- on the head of the same record, all 12 channels
        cpu = None
        try:
            from oracle import cfast

            if cfast.available():
                n_cpu = 40
                data = np.memmap(path, dtype=np.int8, mode="r")
                cfast.tracking_parallel(data, ch[:1], s.copy(msToProcess=10), mode="WB")  # (code generation, library load)
                t0 = time.perf_counter()
                cfast.tracking_parallel(data, ch, s.copy(msToProcess=n_cpu * 10), mode="WB")
                dt_cpu = time.perf_counter() - t0
                del data
                cpu = {"ms_per_epoch_12ch": dt_cpu / n_cpu * 1e3, "x_realtime_12ch": n_cpu * 0.010 / dt_cpu, "threads": 12, "kind": "port",
                       "impl": "oracle/tracking.py with its sample loops in C (oracle/c/trk_oracle.c), one thread per channel",
                       "sample": f"the first {n_cpu} epochs of the same record, all 12 channels, in {dt_cpu:.1f} s (code generation in Python included); "
                                 "float64 restatement of WB_tracking.m, not MATLAB"}
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}
    finally:
        if own and os.path.exists(path):
            os.remove(path)
    half = epochs // 2
    locked = sum(bool(np.abs(r.Pilot_I_P[half:]).mean() > 3 * np.abs(r.Pilot_Q_P[half:]).mean()) or
                 bool(np.abs(r.I_P[half:]).mean() > 3 * np.abs(r.Q_P[half:]).mean()) for r in res)
    cno = [float(np.mean(r.B1C_CNo[len(r.B1C_CNo) // 2:])) for r in res]
    return {"mode": "WB", "channels": 12, "epochs": epochs, "completed": [int(r.completed) for r in res], "fs_MHz": s.samplingFreq / 1e6,
            "record_GB": n / 1e9, "wall_s": wall, "device_loop_ms": dev_ms, "x_realtime_12ch_wall": epochs * 0.010 / wall,
            "channels_locked": locked, "cno_dBHz_mean": float(np.mean(cno)), "cno_dBHz_min_max": [min(cno), max(cno)],
            "record_write_s": t_write, "cpu_baseline": cpu,
            "note": "wall time of one WB_tracking call on a raw int8 file: file -> HBM window, 3 600 one-launch epochs, C/N0, results; "
                    "the file itself (synthetic: the 12 satellites at 47 dB-Hz in 20-ms blocks with 32 noise realisations in random order) is written beforehand and not timed"}


def strict_f32_leg(local_rank, s, x, n_cells_samples, ncomp, n_circ):
    """Extra key, never `value`: the same call with fp32 STORAGE of the spectra and of the inter-pass buffer as well
    (BDS_ACQ_FP16=0): nothing between the int8 block and the f64 refinement is narrower than the reference's
    GPU_acquisition.m (single).  Twice the bytes of the default on every pass."""
    import bds_amd

    os.environ["BDS_ACQ_FP16"] = "0"
    try:
        c = bds_amd.native.Context(local_rank)  # the knobs are read once, at context creation
    finally:
        del os.environ["BDS_ACQ_FP16"]
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
    bytes_per_pair = tm["cells_per_pair"] * 8 * (1 + ncomp) * n_circ
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_b1c_f32.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if int(tj.get("cells_per_pair", -1)) == int(tm["cells_per_pair"]):
            traffic = tj["bytes_per_pair"] / 1e9
    return {"dtype": "f32", "half_storage": int(tm["half_storage"]), "storage": "fp32 complex", "ms_per_step": dt * 1e3,
            "value": n_cells_samples / dt / 1e6, "unit": "Msamples/s", "pair_ms": tm["cell_pair_ms"], "rows_ms": tm.get("rows_ms"),
            "cols_ms": tm.get("cols_ms"), "frac": bytes_per_pair / (tm["cell_pair_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if tm["cell_pair_ms"] else None,
            "traffic": traffic, "traffic_unit": "GB per launch pair (PMC)",
            "note": "fp32 storage AND arithmetic end to end (one extra call); the headline stores spectra and the inter-pass buffer as fp16 complex"}


ROWS_KERNEL = {0: "k_rows_inv (run-time plan)", 1: "k_rows_inv_f", 2: "k_rows_wave_f", 3: "k_pfa_rows"}
COLS_KERNEL = {0: "k_cols_inv_max (run-time plan)", 1: "k_cols_inv_max_f (tile)", 2: "k_cols_wave_f", 3: "k_cols_small_f", 4: "k_pfa_cols"}


def kernel_label(tm):
    """Which row / column kernels the library launched for this plan (bds_timing: plan_l1 x plan_l2, rows_kernel, cols_kernel)."""
    fl = int(tm.get("kernel_flags", 0))
    extra = [n for b, n in ((1, "components interleaved in the inter-pass buffer"), (2, "packed-fp32 butterflies")) if fl & b]
    return "%s<%d> + %s<%d> (plan %d x %d%s)" % (ROWS_KERNEL.get(tm.get("rows_kernel"), "?"), tm.get("plan_l2", 0),
                                                  COLS_KERNEL.get(tm.get("cols_kernel"), "?"), tm.get("plan_l1", 0),
                                                  tm.get("plan_l1", 0), tm.get("plan_l2", 0), "; " + ", ".join(extra) if extra else "")


def b2a_leg(local_rank, steps=5):
    """Extra key `b2a` (BASELINE.json configs[1], never `value`): the B2a full acquisition -- 63 PRNs x 26 Doppler bins, 1 ms code,
    99.375 MS/s -- timed like the headline (block resident in HBM, code spectra cached, `steps` complete bds_acq_run calls)."""
    import bds_amd

    s, x, sats, label = build_workload("b2a")
    c = bds_amd.native.Context(local_rank)
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        res = c.acq_run(s)
        tims = []
        t0 = time.perf_counter()
        for _ in range(steps):
            res = c.acq_run(s)
            tims.append(c.timing())
        dt = (time.perf_counter() - t0) / steps
        tm = tims[-1]
    finally:
        c.close()
    n, d, p, nc = tm["n_circ"], tm["n_bins"], tm["n_prn"], tm["n_comp"]
    b_alg = 9.0 * n * d + 8.0 * (1 + nc) * n * p * d
    return {"workload": label, "ms_per_step": dt * 1e3, "steps": steps, "value": float(n) * p * d / dt / 1e6, "unit": "Msamples/s",
            "stage_ms": {k: float(np.mean([t[k] for t in tims])) for k in ("total_ms", "forward_ms", "search_ms", "refine_ms")},
            "whole_job_frac_of_hbm_peak": b_alg / dt / 1e9 / HBM_PEAK_GBS, "kernel": kernel_label(tm),
            "pair_ms": tm["cell_pair_ms"], "cells_per_pair": tm["cells_per_pair"], "n_pairs": tm["n_pairs"],
            "satellites_detected": sorted(int(q) for q in np.nonzero(res[0])[0] + 1),
            "satellites_injected": sorted(sat.prn for sat in sats)}


def other_mode_leg(local_rank, s, x, budget, steps=3):
    """Extra keys `serving` / `default` / `minimal` (never `value`): the same workload in the search mode the headline does NOT use, timed like the
    headline (block resident, code spectra cached, `steps` whole bds_acq_run calls after one untimed call that allocates the buffers)."""
    import bds_amd

    c = bds_amd.native.Context(local_rank)
    try:
        if budget is not None:
            c.acq_set_pair_budget(budget)
        c.acq_load(s, x)
        c.acq_prepare(s)
        t0 = time.perf_counter()
        res = c.acq_run(s)
        first = time.perf_counter() - t0
        tims = []
        t0 = time.perf_counter()
        for _ in range(steps):
            res = c.acq_run(s)
            tims.append(c.timing())
        dt = (time.perf_counter() - t0) / steps
        tm = tims[-1]
    finally:
        c.close()
    n, d, p, nc = tm["n_circ"], tm["n_bins"], tm["n_prn"], tm["n_comp"]
    pair_ms = float(np.mean([t["cell_pair_ms"] for t in tims]))
    bpp = tm["cells_per_pair"] * 8 * (1 + nc) * n
    return {"mode": {None: "library default (inter-pass buffer budget 40 GiB)", 0: "minimal (bds_acq_set_pair_budget_gb(0): one PRN's Doppler row per launch pair)"}.get(
                budget, "serving (bds_acq_set_pair_budget_gb(auto))"),
            "ms_per_step": dt * 1e3, "steps": steps, "value": float(n) * p * d / dt / 1e6, "unit": "Msamples/s",
            "stage_ms": {k: float(np.mean([t[k] for t in tims])) for k in ("total_ms", "forward_ms", "search_ms", "refine_ms")},
            "frac": bpp / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pair_ms > 0 else None, "pair_ms": pair_ms, "rows_ms": tm.get("rows_ms"),
            "cols_ms": tm.get("cols_ms"), "cells_per_pair": tm["cells_per_pair"], "n_pairs": tm["n_pairs"],
            "inter_pass_buffer_GB": (-(-p // int(tm["n_pairs"])) * d if tm["n_pairs"] < p else tm["cells_per_pair"]) * nc * tm["fft_len"]
                                    * (4 if tm.get("half_storage") else 8) / 1e9,
            "first_run_ms": first * 1e3, "satellites_detected": sorted(int(q) for q in np.nonzero(res[0])[0] + 1)}


def cold_leg(local_rank, s, x, budget=None):
    """`cold` measured in a FRESH PROCESS (python bench.py --cold-child ...): the HIP runtime keeps freed device memory of moderate
    size in the process, so a fresh context inside this process would get its buffers back for nothing -- a first call from a new
    MATLAB session does not.  Falls back to the in-process measurement if the child fails."""
    import subprocess

    name = "b1c" if str(s.signal).upper() == "B1C" else "b2a"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-child", name, "--cold-budget", str(budget), "--gpus", "1",
                            "--cold-device", str(local_rank)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            d["process"] = "fresh process (python bench.py --cold-child)"
            return d
    except Exception:  # noqa: BLE001
        pass
    d = cold_leg_here(local_rank, s, x, budget)
    d["process"] = "this process (the child failed): buffers freed by the earlier legs may have been handed back without a fresh allocation"
    return d


def cold_leg_here(local_rank, s, x, budget=None):
    """Extra key `cold` (SURVEY.md 8d "also report cold"; the reference times the whole call, postProcessing.m:104-112): a
    FRESH context, wall time of each step with a device-wide synchronisation after it -- the IF block to HBM (bds_acq_load:
    H2D + block statistics), the per-PRN code generation and code-spectrum transforms (bds_acq_prepare), the first
    bds_acq_run (plan constants, buffers, first launches) and a second, warm one for comparison."""
    import torch

    import bds_amd

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    t0 = tick()
    c = bds_amd.native.Context(local_rank)
    try:
        if budget is not None:
            c.acq_set_pair_budget(budget)
        t1 = tick()
        c.acq_load(s, x)
        t2 = tick()
        c.acq_prepare(s)
        t3 = tick()
        c.acq_run(s)
        t4 = tick()
        c.acq_run(s)
        t5 = tick()
    finally:
        c.close()
    return {"create_ms": (t1 - t0) * 1e3, "load_ms": (t2 - t1) * 1e3, "prepare_ms": (t3 - t2) * 1e3, "first_run_ms": (t4 - t3) * 1e3,
            "cold_total_ms": (t4 - t1) * 1e3, "warm_run_ms": (t5 - t4) * 1e3, "block_MB": x.nbytes / 1e6,
            "mode": "library default" if budget is None else "minimal (budget 0)" if budget == 0 else
                    "serving (bds_acq_set_pair_budget_gb(auto): first_run includes the allocation of the inter-pass buffer)",
            "note": "fresh context; load = H2D of the int8 block + its statistics, prepare = 63 x code generation + code-spectrum "
                    "transforms (cached across calls afterwards), first_run = first bds_acq_run (buffers, plan constants); "
                    "cold_total = load + prepare + first_run; wall clock with a device synchronisation after each step"}


def clock_leg(local_rank, s, x):
    """Engine clock the search kernels actually run at (extra call, never `value`): with BDS_ACQ_CLOCKPROBE=1 sampled workgroups of
    the row and column pass time their own life with the shader clock against the constant reference clock
    (bds_timing::shader_clock_GHz).  The issue bound of roofline.valu is quoted at the 2.4 GHz peak clock; under this load the
    chip runs slower (power), and the bound scales with it."""
    import bds_amd

    os.environ["BDS_ACQ_CLOCKPROBE"] = "1"
    try:
        c = bds_amd.native.Context(local_rank)
    finally:
        del os.environ["BDS_ACQ_CLOCKPROBE"]
    try:
        c.acq_load(s, x)
        c.acq_prepare(s)
        c.acq_run(s)
        c.acq_run(s)
        tm = c.timing()
    finally:
        c.close()
    return {"shader_clock_GHz": tm.get("shader_clock_GHz"), "pair_ms": tm["cell_pair_ms"], "cells_per_pair": tm["cells_per_pair"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="b1c", choices=["b1c", "b2a", "joint"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracking", action="store_true", help="skip the (untimed) tracking leg")
    ap.add_argument("--no-strict-f32", dest="no_fast_path", action="store_true",
                    help="skip the extra (never the headline) call with fp32 storage end to end")
    ap.add_argument("--no-tracking-full", action="store_true", help="skip the cfg4 leg (12 channels x 3 600 epochs from a 3.6 GB file)")
    ap.add_argument("--no-b2a", action="store_true", help="skip the extra cfg2 leg (B2a full acquisition, key `b2a`)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start leg (key `cold`)")
    ap.add_argument("--prns", type=int, default=63, help="tuning only: search PRNs 1..N instead of all 63")
    ap.add_argument("--cold-child", default=None, choices=["b1c", "b2a"], help=argparse.SUPPRESS)  # internal: the cold leg in a fresh process
    ap.add_argument("--cold-budget", default="0", help=argparse.SUPPRESS)
    ap.add_argument("--cold-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--mode", default="default", choices=["serving", "default", "minimal"],
                    help="inter-pass buffer budget of the search the headline runs with (include/bds_mi355x.h, bds_acq_set_pair_budget_gb): "
                         "serving = 'auto', as many PRNs' Doppler rows per launch pair as 60 %% of the free HBM hold (allocated before the timed "
                         "region); default = the library's own default (40 GiB: 8 PRNs per pair at cfg3); minimal = 0 (one PRN per pair, 5 GB). "
                         "The other two are timed beside the headline (keys `serving` / `default` / `minimal`)")
    ap.add_argument("--lean", action="store_true", help="same as --mode minimal")
    args = ap.parse_args()
    if args.lean:
        args.mode = "minimal"

    if args.cold_child:  # the cold leg of another bench.py, in this fresh process
        s_, x_, _, _ = build_workload(args.cold_child)
        b = args.cold_budget
        print(json.dumps(cold_leg_here(args.cold_device, s_, x_, None if b == "None" else "auto" if b.lower().startswith("a") else float(b))))
        return

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
        # The headline is what a plain bds_acquire caller gets: the LIBRARY DEFAULT (round 6; rounds 1-5 put the opt-in serving mode
        # here).  The serving mode (include/bds_mi355x.h, bds_acq_set_pair_budget_gb / BDS_ACQ_PAIR_GB=auto -- as many PRNs' Doppler rows
        # per launch pair as 60 % of the free HBM hold) and the minimal footprint (one PRN per pair) are timed beside it (keys `serving`,
        # `minimal`); a first call in a fresh process is the `cold` key.
        if args.mode != "default":
            g["ctx"].acq_set_pair_budget({"serving": "auto", "minimal": 0}[args.mode])
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
    traffic = traffic_source = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{names[0]}.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("cells_per_pair", 0) > 0 and cells_per_pair > 0:
            # (a launch pair carries as many PRNs' Doppler rows as the device memory allows: the PMC run -- 2 PRNs -- and this run
            #  differ in cells per pair; traffic and instruction counts are proportional to the cells)
            traffic = tj["bytes_per_pair"] * (cells_per_pair / tj["cells_per_pair"]) / 1e9  # GB per launch pair
            traffic_source = ("replayed from profiles/traffic_%s.json (the builder's rocprofv3 --pmc run of %s: %.4f GB per %d-cell pair, "
                              "scaled to the %g cells of this run's pairs), NOT measured in this run"
                              % (names[0], tj.get("round", "an earlier round"), tj["bytes_per_pair"] / 1e9, int(tj["cells_per_pair"]), cells_per_pair))

    # What binds the launch pair: SIMD issue, not HBM.  profiles/valu_<workload>.json (tools/make_valu.py) holds, per kernel,
    # the vector instructions per dispatch counted by the hardware (PMC SQ_INSTS_VALU) and the issue cycles per instruction
    # class from the compiler's ISA (tools/isa_mix.py: 2-cycle fp32 ops, 4-cycle packed / conversions / compares, 8-cycle
    # square roots; LDS instructions occupy their SIMD too: 8 cycles per ds_read_b64, 24 per ds_write_b64,
    # tools/probe/valu_rate.hip).  bound_ms = those cycles per SIMD at the 2.4 GHz peak clock.
    valu = None
    vpath = os.path.join(ROOT, "profiles", f"valu_{names[0]}.json")
    if os.path.exists(vpath):
        vj = json.load(open(vpath))
        if vj.get("cells_per_pair", 0) > 0 and cells_per_pair > 0:
            k = cells_per_pair / vj["cells_per_pair"]
            per_pair = ("insts_per_pair", "valu_pipe_cycles_per_simd", "lds_marginal_issue_cycles_per_simd", "lds_unit_cycles_per_cu", "bound_ms",
                        "additive_r3_bound_ms", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "measured_ms")

            def scaled(d):
                return {kk: (scaled(v) if isinstance(v, dict) and kk != "power" else v * k if kk in per_pair and isinstance(v, (int, float)) else v)
                        for kk, v in d.items()}

            valu = scaled(vj)
            valu["cells_per_pair"] = cells_per_pair
            valu["measured_cells_per_pair"] = vj["cells_per_pair"]
            valu["source"] = ("replayed from profiles/valu_%s.json (hardware instruction counts of the builder's rocprofv3 --pmc run, %s, at %d cells "
                              "per pair, scaled to this run's %g; static instruction classes from the compiler's ISA), NOT measured in this run"
                              % (names[0], vj.get("round", "an earlier round"), int(vj["cells_per_pair"]), cells_per_pair))
            valu["frac_of_issue_bound"] = valu["bound_ms"] / pair_ms if pair_ms > 0 else None

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
        "dtype": "f32",
        "dtype_detail": {0: "f32 search + f64 refinement", 1: "f32 search on f16-stored spectra + f64 refinement"}[int(tm.get("half_storage", 0))],
        "data": "synthetic",
        "config": {"workload": label, "prns": p_total, "doppler_bins": n_bins, "n_circ": n_circ,
                   "search_mode": {"minimal": "minimal (bds_acq_set_pair_budget_gb(0)): one PRN's Doppler row per launch pair",
                                   "default": "library default (inter-pass buffer budget 40 GiB): %d launch pairs of %.0f (PRN, bin) cells on average" % (int(tm["n_pairs"]), tm["cells_per_pair"]),
                                   "serving": "serving (bds_acq_set_pair_budget_gb(auto) = BDS_ACQ_PAIR_GB=auto): %d launch pairs of %.0f (PRN, bin) cells on average, "
                                              "inter-pass buffer allocated in the warm-up call (a first call in a fresh process: cold.b1c_serving); the library default "
                                              "(40 GiB) and the minimal footprint are timed beside it (keys `default`, `minimal`)" % (int(tm["n_pairs"]), tm["cells_per_pair"])}[args.mode],
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
                     "frac": achieved / HBM_PEAK_GBS,
                     # the same pair at the bytes it really stores (fp16 complex spectra: 4 B per element instead of the model's 8)
                     "frac_at_stored_bytes": achieved * (0.5 if tm.get("half_storage") else 1.0) / HBM_PEAK_GBS,
                     "frac_strict_f32": None,  # filled from the strict_f32 leg below: fp32 storage AND arithmetic end to end
                     "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_unit": "GB per launch pair (PMC)", "algorithmic_GB_per_pair": bytes_per_pair / 1e9,
                     "kernel": "launch pair %s; one pair = %d (PRN, bin) cells" % (kernel_label(tm), int(cells_per_pair)),
                     "valu": valu,
                     "pair_ms": pair_ms, "rows_ms": tm.get("rows_ms"), "cols_ms": tm.get("cols_ms"), "n_extra": tm.get("n_extra"),
                     "storage": "fp16 complex" if tm.get("half_storage") else "fp32 complex",
                     # `achieved` prices every spectrum element at the 8 bytes of SURVEY.md 8d's model (fp32 complex); the
                     # same count at the element size the kernels really store, for comparison with `traffic`
                     "achieved_at_stored_element_size": achieved * (0.5 if tm.get("half_storage") else 1.0)},
    }
    # The timed contexts hold the search's inter-pass buffer (up to 60 % of the free device memory): give it back before the extra
    # legs build contexts of their own -- the cold leg is to find the device as a fresh process does.
    for k, g in enumerate(sigs):
        if k > 0:
            g["ctx"].close()
    bds_amd.release_context(local_rank)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s, x, names[0])
        else:
            out["cpu_baseline"] = None
        out["tracking"] = tracking_leg(names[0], local_rank, s) if world == 1 and not args.no_tracking else None
        out["tracking_full"] = (tracking_full_leg(local_rank, s) if world == 1 and names[0] == "b1c" and not args.no_tracking
                                and not args.no_tracking_full else None)
        import hashlib

        # acqResults of the last step, per signal, as a digest: sharded / joint runs must reproduce the single-device bits
        out["config"]["results_sha256"] = {g["name"]: hashlib.sha256(np.ascontiguousarray(np.stack(r), dtype=np.float64).tobytes()).hexdigest()
                                           for g, r in zip(sigs, res_all)}
        out["strict_f32"] = (strict_f32_leg(local_rank, s, x, float(n_circ) * p_total * n_bins, ncomp, n_circ)
                             if world == 1 and len(sigs) == 1 and not args.no_fast_path else None)
        if out["strict_f32"]:
            out["roofline"]["frac_strict_f32"] = out["strict_f32"]["frac"]
        if valu is not None and world == 1 and len(sigs) == 1 and not args.no_fast_path:
            ck = clock_leg(local_rank, s, x)
            if ck["shader_clock_GHz"]:
                valu["shader_clock_GHz"] = ck["shader_clock_GHz"]
                valu["shader_clock_source"] = "measured in this run (one extra call, BDS_ACQ_CLOCKPROBE=1: sampled workgroups, s_memtime against s_memrealtime)"
                valu["bound_ms_at_shader_clock"] = valu["bound_ms"] * valu["clock_GHz"] / ck["shader_clock_GHz"]
                # (the clock probe runs a context of its own, at the library's default budget -- compare at the same cells per pair)
                valu["frac_of_issue_bound_at_shader_clock"] = (valu["bound_ms_at_shader_clock"] * ck["cells_per_pair"] / cells_per_pair / ck["pair_ms"]
                                                               if ck["pair_ms"] and cells_per_pair else None)
                valu["shader_clock_probe_pair"] = {"pair_ms": ck["pair_ms"], "cells_per_pair": ck["cells_per_pair"], "mode": "library default"}
        if world == 1 and len(sigs) == 1 and args.prns == 63 and not args.no_cold:
            out["cold"] = {names[0]: cold_leg(local_rank, s, x)}
            if names == ["b1c"]:
                out["cold"]["b1c_serving"] = cold_leg(local_rank, s, x, budget="auto")
            if names == ["b1c"] and not args.no_b2a:
                s2, x2, _, _ = build_workload("b2a")
                out["cold"]["b2a"] = cold_leg(local_rank, s2, x2)
        if world == 1 and names == ["b1c"] and args.prns == 63 and not args.no_fast_path:
            for m, b in (("serving", "auto"), ("default", None), ("minimal", 0)):
                if m != args.mode:
                    out[m] = other_mode_leg(local_rank, s, x, b)
        out["b2a"] = b2a_leg(local_rank) if world == 1 and names == ["b1c"] and not args.no_b2a and args.prns == 63 else None
        if len(sigs) > 1:
            out["config"]["satellites_detected_per_signal"] = {g["name"]: sorted(int(p) for p in np.nonzero(r[0])[0] + 1)
                                                                for g, r in zip(sigs, res_all)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
