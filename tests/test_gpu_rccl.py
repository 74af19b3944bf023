"""The RCCL branch of sparsebev_amd.parallel on the one GPU a test box has: a ONE-rank `nccl` process group
(torch.distributed backend "nccl" = RCCL on ROCm) bound to cuda:0 -- communicator creation with ``device_id``, the
device-tensor MAX / SUM all-reduces of the metric reduction (bench.py's only collective; the reference's counterpart is
the end-of-eval gather, val.py:94,132), ``reduce_mean`` (train.py:92 / sparsebev_head.py:247) and the object gather.
RCCL refuses two ranks on one device, so N > 1 itself is covered by the world-size-2 gloo tests + the driver's 8-GPU run."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from sparsebev_amd.parallel import SampleShard, init_distributed, reduce_mean
    try:
        rank, world, dev = init_distributed(1, force_group=True)
        assert dist.is_initialized() and dist.get_backend() == 'nccl' and dev.type == 'cuda'
        shard = SampleShard(rank, world)
        assert shard._device().type == 'cuda'                       # the metric tensors live on the GPU for RCCL
        shard.barrier()
        e_max, n, chk, e_min = shard.reduce_metrics(1.25, 40, 3.5, per_rank=True)
        npos = torch.tensor([7.0, 2.0], device=dev)
        mean = reduce_mean(npos)
        gathered = shard.gather_results([{'sample': 0}, {'sample': 1}], 2)
        x = torch.arange(8, device=dev, dtype=torch.float32)
        dist.all_reduce(x)                                          # a plain device all-reduce through the same communicator
        torch.cuda.synchronize()
        q.put(('ok', e_max, n, chk, e_min, mean.tolist(), mean.device.type, mean.data_ptr() != npos.data_ptr(),
               [g['sample'] for g in gathered], x.tolist()))
        shard.shutdown()
        assert not dist.is_initialized()
    except Exception as e:      # noqa: BLE001
        q.put(('error', repr(e)))
        raise


def test_one_rank_rccl_group_runs_the_metric_collectives_on_device():
    port = 29700 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert res[0] == 'ok', res
    _, e_max, n, chk, e_min, mean, mean_dev, fresh, order, x = res
    assert (e_max, n, chk, e_min) == (1.25, 40.0, 3.5, 1.25)
    assert mean == [7.0, 2.0] and mean_dev == 'cuda' and fresh
    assert order == [0, 1] and x == [float(i) for i in range(8)]
    assert p.exitcode == 0
