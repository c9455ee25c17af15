"""N > 1 path on CPU: world_size-2 gloo process group, PRN sharding + all-reduce(SUM) reassembly.
The per-rank compute is injected (the oracle stands in for the GPU search; there is no CPU
product path), so what is tested is exactly the host logic bench.py / sharded_acquisition run."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bds_amd
    from bds_amd import synth
    from oracle import acquisition as oacq
    from helpers import spc_of

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9, 14], acqSearchBand=800, fineNoncoh=3)
    x = synth.make_if(s, [synth.Sat(9, -330.0, 12345.6, 2.0, 48.0)], 6 * spc_of(s), seed=2)

    def stand_in(sig, st, prn_list):
        return oacq.acquisition_b2a(sig.astype(np.float64), st.copy(acqSatelliteList=list(prn_list)))

    def pad(r, n=14):  # oracle sizes results by max(list); the ABI by max(acqSatelliteList)
        out = {}
        for f in ("carrFreq", "codePhase", "peakMetric"):
            v = np.zeros(n)
            v[: len(getattr(r, f))] = getattr(r, f)
            out[f] = v
        return type(r)(**out)

    res = bds_amd.sharded_acquisition(x, s, acquire=lambda sig, st, prn_list: pad(stand_in(sig, st, prn_list)))
    full = pad(stand_in(x, s, [5, 9, 14]))
    ok = all(np.array_equal(getattr(res, f), getattr(full, f)) for f in ("carrFreq", "codePhase", "peakMetric"))
    q.put((rank, ok, bds_amd.shard_prns([5, 9, 14], rank, world), float(res.carrFreq[8])))
    dist.destroy_process_group()


def test_two_rank_prn_shard_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert [o[1] for o in out] == [True, True]
    assert out[0][2] == [5, 14] and out[1][2] == [9]
    assert out[0][3] == out[1][3] != 0


def test_shard_covers_every_prn_once():
    import bds_amd

    prns = list(range(1, 64))
    for world in (1, 2, 4, 8):
        parts = [bds_amd.shard_prns(prns, r, world) for r in range(world)]
        assert sorted(p for part in parts for p in part) == prns
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
