"""N > 1 path on CPU: world_size-2 gloo process group, (signal, PRN) job sharding + all-reduce(SUM) reassembly.
The per-rank compute is injected (the oracle stands in for the GPU search; there is no CPU product path), so what
is tested is exactly the host logic bench.py / sharded_acquisition_joint run.  The partition itself
(bds_shard_jobs, host C code of the library) is checked for world sizes up to 8 on the BASELINE.json configs[4]
job list."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    import bds_amd
    from bds_amd import synth
    from helpers import spc_of

    s2 = bds_amd.init_settings_b2a(samplingFreq=25e6, IF=6.5e6, acqSatelliteList=[5, 9, 14], acqSearchBand=800, fineNoncoh=3)
    x2 = synth.make_if(s2, [synth.Sat(9, -330.0, 12345.6, 2.0, 48.0)], 6 * spc_of(s2), seed=2)
    s1 = bds_amd.init_settings_b1c(samplingFreq=5e6, IF=1.2e6, acqSatelliteList=[3, 7], acqSearchBand=100, acqStep=50)
    x1 = synth.make_if(s1, [synth.Sat(7, 40.0, 2345.6, 1.0, 47.0)], 3 * spc_of(s1), seed=3)
    return [(x1, s1), (x2, s2)]


def _stand_in(sig, st, prn_list):
    """The oracle on a PRN shard, padded to max(acqSatelliteList) like the C ABI's outputs."""
    from oracle import acquisition as oacq

    fn = oacq.acquisition_b1c if st.signal.upper() == "B1C" else oacq.acquisition_b2a
    r = fn(sig.astype(np.float64), st.copy(acqSatelliteList=list(prn_list)))
    n = max(int(p) for p in st.acqSatelliteList)
    out = {}
    for f in ("carrFreq", "codePhase", "peakMetric"):
        v = np.zeros(n)
        v[: len(getattr(r, f))] = getattr(r, f)
        out[f] = v
    return type(r)(**out)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bds_amd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = _blocks()
    calls = []

    def acquire(sig, st, prn_list):
        calls.append((st.signal.upper(), list(prn_list)))
        return _stand_in(sig, st, prn_list)

    res = bds_amd.sharded_acquisition_joint(blocks, acquire=acquire)
    ok = True
    for (x, s), r in zip(blocks, res):
        full = _stand_in(x, s, [int(p) for p in s.acqSatelliteList])
        ok = ok and all(np.array_equal(getattr(r, f), getattr(full, f)) for f in ("carrFreq", "codePhase", "peakMetric"))
    # the single-signal entry is the same machinery
    one = bds_amd.sharded_acquisition(blocks[1][0], blocks[1][1], acquire=_stand_in)
    ok = ok and np.array_equal(one.carrFreq, res[1].carrFreq)
    q.put((rank, ok, calls, float(res[1].carrFreq[8]), float(res[0].carrFreq[6])))
    dist.destroy_process_group()


def test_two_rank_joint_job_shard_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert [o[1] for o in out] == [True, True]
    # every (signal, PRN) job ran on exactly one rank
    ran = sorted((sig, p) for o in out for sig, prns in o[2] for p in prns)
    assert ran == sorted([("B1C", 3), ("B1C", 7), ("B2A", 5), ("B2A", 9), ("B2A", 14)])
    # both ranks hold the complete, identical results (detections non-zero)
    assert out[0][3] == out[1][3] != 0 and out[0][4] == out[1][4] != 0


def test_shard_covers_every_prn_once():
    import bds_amd

    prns = list(range(1, 64))
    for world in (1, 2, 4, 8):
        parts = [bds_amd.shard_prns(prns, r, world) for r in range(world)]
        assert sorted(p for part in parts for p in part) == prns
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert bds_amd.shard_prns([20, 19, 19, 5], 0, 2) == [20, 5] and bds_amd.shard_prns([20, 19, 19, 5], 1, 2) == [19]


def test_joint_partition_cfg5_properties():
    """BASELINE.json configs[4]: 63 B1C + 63 B2a PRNs at 99.375 MS/s over up to 8 ranks.  Each job on exactly one
    rank; one B1C job costs ~77 B2a jobs; the LPT rule keeps the heaviest rank within one B1C job of the mean."""
    import bds_amd
    from bds_amd import native

    s1 = bds_amd.init_settings_b1c(samplingFreq=99.375e6, IF=14.58e6, acqSatelliteList=list(range(1, 64)))
    s2 = bds_amd.init_settings_b2a(acqSatelliteList=list(range(1, 64)))
    c1, c2 = native.acq_job_cost(s1), native.acq_job_cost(s2)
    assert 70 < c1 / c2 < 85
    for world in (1, 2, 3, 4, 8):
        shards = [bds_amd.shard_joint([s1, s2], r, world) for r in range(world)]
        for k in (0, 1):
            assert sorted(p for sh in shards for p in sh[k]) == list(range(1, 64))
        load = [len(sh[0]) * c1 + len(sh[1]) * c2 for sh in shards]
        assert max(load) - sum(load) / world <= c1
        # ideal strong scaling of the joint job: total / max load
        assert sum(load) / max(load) >= {1: 1, 2: 1.99, 3: 2.9, 4: 3.9, 8: 7.8}[world]


def test_shard_jobs_argument_checks():
    from bds_amd import native

    assert list(native.shard_jobs([], 4)) == []
    assert list(native.shard_jobs([3.0, 1.0, 2.0], 1)) == [0, 0, 0]
    r = native.shard_jobs([5.0, 4.0, 3.0, 3.0, 1.0], 2)  # LPT: 5|4, 3->1 (4+3), 3->0 (5+3), 1->1 ... loads 8 | 8
    assert np.bincount(r, weights=[5.0, 4.0, 3.0, 3.0, 1.0]).tolist() == [8.0, 8.0]
