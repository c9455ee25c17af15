"""Multi-GPU acquisition with one process per GPU (torch.distributed; the nccl backend is RCCL over xGMI).

The (PRN, Doppler-bin) cells are independent and every per-PRN decision needs all bins of that PRN, so the
search shards by (signal, PRN) job.  Jobs are spread over the ranks by cost with the
library's longest-processing-time rule (bds_shard_jobs; for a single signal with uniform cost that is
round-robin), every rank searches its shard on its own GPU with the whole IF block, its result vectors are zero
outside the shard, and ONE ``all_reduce(SUM)`` of 3 x max_prn float64 per signal leaves the complete,
bit-identical acqResults on every rank (x + 0 is exact).  Tracking runs per GPU and needs no collective.

The same partition and exchange live behind the C ABI for a single host process driving all GPUs
(bds_acquire_multi, csrc/bds_multi.hip); this module is the launcher-per-GPU form bench.py uses.
"""
from __future__ import annotations

import numpy as np

from . import native


def job_list(settings_list):
    """[(signal index, PRN)] and per-job cost for a list of settings (one per signal); a repeated PRN is one job."""
    jobs, cost = [], []
    for i, s in enumerate(settings_list):
        c = native.acq_job_cost(s)
        seen = []
        for p in np.atleast_1d(s.acqSatelliteList):
            p = int(p)
            if p not in seen:
                seen.append(p)
                jobs.append((i, p))
                cost.append(c)
    return jobs, cost


def shard_joint(settings_list, rank: int, world: int):
    """PRN shard of `rank` for every signal: list (one entry per signal) of PRN lists."""
    jobs, cost = job_list(settings_list)
    owner = native.shard_jobs(cost, world)
    return [[p for (i, p), r in zip(jobs, owner) if i == k and r == rank] for k in range(len(settings_list))]


def shard_prns(prns, rank: int, world: int):
    """PRN shard of a rank for ONE signal (uniform cost: the LPT rule deals the list round-robin)."""
    prns = [int(p) for p in prns]
    uniq = []
    for p in prns:
        if p not in uniq:
            uniq.append(p)
    owner = native.shard_jobs([1.0] * len(uniq), world)
    return [p for p, r in zip(uniq, owner) if r == rank]


def sharded_acquisition_joint(blocks, acquire=None, device=None, verbose=False):
    """acquisition() of several signals across all ranks of the default process group.

    blocks: list of (long_signal, settings), one per signal (BASELINE.json configs[4]: the B1C and the B2a block).
    ``acquire(long_signal, settings, prn_list=...)`` defaults to the GPU path on this rank's device; tests inject a
    stand-in to exercise partition + collective on the gloo backend without a GPU.
    Returns a list of AcqResults, one per signal, identical on every rank.
    """
    import torch
    import torch.distributed as dist

    from .acquisition import AcqResults, acquisition

    rank, world = dist.get_rank(), dist.get_world_size()
    shards = shard_joint([s for _, s in blocks], rank, world)
    if acquire is None:
        dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)

        def acquire(x, s, prn_list):
            return acquisition(x, s, device=dev, prn_list=prn_list, verbose=verbose)

    out = []
    for (x, s), shard in zip(blocks, shards):
        max_prn = max(int(p) for p in np.atleast_1d(s.acqSatelliteList))
        if shard:
            part = acquire(x, s, prn_list=shard)
            buf = np.stack([np.asarray(part.carrFreq, dtype=np.float64), np.asarray(part.codePhase, dtype=np.float64),
                            np.asarray(part.peakMetric, dtype=np.float64)])
        else:  # this rank holds no job of the signal: contribute zeros
            buf = np.zeros((3, max_prn))
        t = torch.from_numpy(buf)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # one exchange per signal
        carr, cph, pm = t.cpu().numpy()
        out.append(AcqResults(carrFreq=carr, codePhase=cph, peakMetric=pm))
    return out


def sharded_acquisition(long_signal, settings, acquire=None, device=None, verbose=False):
    """acquisition() of one signal across all ranks (the single-signal case of sharded_acquisition_joint)."""
    return sharded_acquisition_joint([(long_signal, settings)], acquire=acquire, device=device, verbose=verbose)[0]
