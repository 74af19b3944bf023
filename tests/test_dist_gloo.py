"""The N>1 path on CPU: world_size-2 gloo processes exercise the sample sharding and the metric
all-reduce that bench.py uses with RCCL on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from sparsebev_amd.parallel import SampleShard, init_distributed, reduce_mean
    r, w, dev = init_distributed(world, backend='gloo')
    shard = SampleShard(r, w)
    mine = shard.indices(7)
    # each rank "processes" its samples: result = f(global index)
    local = [{'sample': i, 'value': float(i * i)} for i in mine]
    shard.barrier()
    elapsed, total, chk, fastest = shard.reduce_metrics(1.0 + r, len(mine), sum(x['value'] for x in local), per_rank=True)
    assert fastest == 1.0                                      # MIN over ranks rides in the same MAX all-reduce
    gathered = shard.gather_results(local, 7)
    npos = torch.tensor([3.0 + 4.0 * r])                       # num_total_pos of this rank
    mean = reduce_mean(npos)
    assert float(npos[0]) == 3.0 + 4.0 * r                     # input untouched
    q.put((r, mine, elapsed, total, chk, [g['sample'] for g in gathered], float(mean[0])))
    shard.shutdown()


def test_two_rank_gloo_shard_and_metric_allreduce():
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, mine0, e0, t0, c0, g0, m0), (r1, mine1, e1, t1, c1, g1, m1) = res
    assert m0 == m1 == 5.0                                          # reduce_mean: (3 + 7) / 2 on both ranks
    assert mine0 == [0, 2, 4, 6] and mine1 == [1, 3, 5]            # DistributedSampler(shuffle=False) partition
    assert e0 == e1 == 2.0                                          # MAX over ranks
    assert t0 == t1 == 7.0                                          # SUM of samples
    assert c0 == c1 == float(sum(i * i for i in range(7)))          # SUM of checksums
    assert g0 == g1 == list(range(7))                               # gathered back in global order


def test_single_rank_is_collective_free():
    from sparsebev_amd.parallel import SampleShard
    s = SampleShard(0, 1)
    assert s.indices(3) == [0, 1, 2] and s.owns(5)
    assert s.reduce_metrics(0.5, 10, 3.0) == (0.5, 10.0, 3.0)
    assert not dist.is_initialized()
    from sparsebev_amd.parallel import reduce_mean
    t = torch.tensor([2.5])
    assert reduce_mean(t) is t                                      # no process group: identity
