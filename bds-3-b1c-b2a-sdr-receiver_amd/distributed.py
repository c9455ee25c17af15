"""PRN-sharded multi-GPU acquisition: one process per GPU, one all-reduce.

The (PRN, Doppler-bin) cells are independent and every per-PRN decision needs all bins of
that PRN, so the search shards by PRN.  Rank r searches
``acqSatelliteList[r::world]`` on its own GPU with the whole IF block; its result vectors are
zero outside the shard, so ONE ``all_reduce(SUM)`` of 3 x max_prn float64 (RCCL over xGMI
with the nccl backend) leaves the complete, bit-identical acqResults on every rank
(x + 0 is exact).  Tracking runs per GPU and needs no collective.
"""
from __future__ import annotations

import numpy as np


def shard_prns(prns, rank: int, world: int):
    """Round-robin PRN shard of a rank (cost per PRN is uniform within one signal)."""
    return [int(p) for p in list(prns)[rank::world]]


def sharded_acquisition(long_signal, settings, acquire=None, device=None, verbose=False):
    """acquisition() across all ranks of the default torch.distributed process group.

    ``acquire(long_signal, settings, prn_list=...)`` defaults to the GPU path
    (bds_amd.acquisition on this rank's device); tests inject a stand-in to exercise the
    sharding + collective on the gloo backend without a GPU.
    """
    import torch
    import torch.distributed as dist

    from .acquisition import AcqResults, acquisition

    rank, world = dist.get_rank(), dist.get_world_size()
    shard = shard_prns(np.atleast_1d(settings.acqSatelliteList), rank, world)
    max_prn = max(int(p) for p in np.atleast_1d(settings.acqSatelliteList))
    if acquire is None:
        dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)

        def acquire(x, s, prn_list):
            return acquisition(x, s, device=dev, prn_list=prn_list, verbose=verbose)

    if shard:
        part = acquire(long_signal, settings, prn_list=shard)
        buf = np.stack([np.asarray(part.carrFreq, dtype=np.float64), np.asarray(part.codePhase, dtype=np.float64),
                        np.asarray(part.peakMetric, dtype=np.float64)])
    else:  # more ranks than PRNs: contribute zeros
        buf = np.zeros((3, max_prn))
    t = torch.from_numpy(buf)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    carr, cph, pm = t.cpu().numpy()
    return AcqResults(carrFreq=carr, codePhase=cph, peakMetric=pm)
