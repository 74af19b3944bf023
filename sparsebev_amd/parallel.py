"""Multi-GPU layer: one process per GPU, samples sharded across ranks, one collective.

The decoder path has no cross-sample term, so rank r of W simply owns samples
{r, r+W, ...} -- the ``DistributedSampler(shuffle=False)`` partition the reference's loader uses
(loaders/builder.py:26-27) -- with replicated weights and NO data-path collective.  The only exchange is the
end-of-run metric all-reduce, the counterpart of the reference's end-of-eval result gather (val.py:132):
``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm); ``gloo`` for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def gpu_numa_cpus(device_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node -> node<N>/cpulist), or None when
    the topology is not exposed (containers, single-node hosts report -1)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus or None
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def pin_to_gpu_numa_node(device_index, local_rank=0, ranks_on_node=1):
    """Pin this process's host threads next to its GPU: the CPUs of the GPU's NUMA node, and within the node an equal share
    per rank that shares it (N host processes on one box otherwise migrate across sockets and each spins up a full-size
    thread pool).  Returns the CPU set applied, or None when nothing was changed."""
    cpus = gpu_numa_cpus(device_index)
    if not cpus or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        allowed = sorted(cpus & os.sched_getaffinity(0))
        if not allowed:
            return None
        if ranks_on_node > 1:                       # ranks whose GPUs share this node split it
            share = max(1, len(allowed) // ranks_on_node)
            mine = allowed[(local_rank % ranks_on_node) * share:(local_rank % ranks_on_node + 1) * share] or allowed
        else:
            mine = allowed
        os.sched_setaffinity(0, set(mine))
        return set(mine)
    except OSError:
        return None


def init_distributed(n_gpus_requested=1, backend=None, force_group=False):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run sets them),
    bind this process to its GPU and create the process group when WORLD_SIZE > 1 (or when ``force_group`` asks for a
    one-rank group: the RCCL communicator, its device binding and the collectives below then run exactly as with N ranks
    -- how the 1-GPU test box executes this branch).  Returns (rank, world_size, device)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(n_gpus_requested, 1) and world > 1:
        raise RuntimeError('--gpus %d but WORLD_SIZE=%d' % (n_gpus_requested, world))
    use_gpu = torch.cuda.is_available()
    if os.environ.get('SBEV_SHARE_GPU') == '1':      # test hook: every rank on GPU 0, metric reduction over gloo (RCCL refuses
        local, backend = 0, 'gloo'                    # two ranks on one device) -- exercises the N > 1 bench path on a 1-GPU box
    device = torch.device('cuda', local) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(device)
        if world > 1 and os.environ.get('SBEV_NO_NUMA_PIN') != '1':
            # ranks on GPUs of the same NUMA node share its CPUs: count them (same sysfs answer on every rank)
            mine = gpu_numa_cpus(local)
            if mine:
                same = [i for i in range(min(world, torch.cuda.device_count())) if gpu_numa_cpus(i) == mine]
                pin_to_gpu_numa_node(local, same.index(local) if local in same else 0, max(1, len(same)))
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this driver
        backend = backend or ('nccl' if use_gpu else 'gloo')
        kw = {'device_id': device} if (use_gpu and backend == 'nccl') else {}
        dist.init_process_group(backend, init_method='env://', rank=rank, world_size=world, **kw)
    return rank, world, device


def reduce_mean(tensor):
    """Average a tensor over all ranks -- mmdet's ``reduce_mean`` as the head's losses use it for ``num_total_pos`` /
    ``cls_avg_factor`` (models/sparsebev_head.py:247,374-384).  One small all-reduce (SUM) over RCCL;
    the input is left untouched; a single process returns it as is."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    out = tensor.clone()
    dist.all_reduce(out.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return out


class SampleShard:
    """Which samples this rank owns, and the metric reduction at the end."""

    def __init__(self, rank, world):
        assert 0 <= rank < world
        self.rank, self.world = rank, world

    def owns(self, sample_index):
        return sample_index % self.world == self.rank

    def indices(self, n_samples):
        """Global sample indices of this rank: r, r+W, r+2W, ... < n_samples."""
        return list(range(self.rank, n_samples, self.world))

    def _device(self):
        if dist.is_initialized() and dist.get_backend() == 'nccl':
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    def _grouped(self):
        """Collectives run whenever a process group exists (also a one-rank group); without one the rank is alone."""
        return dist.is_available() and dist.is_initialized()

    def barrier(self):
        if self._grouped():
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()

    def reduce_metrics(self, elapsed_s, samples_done, checksum, per_rank=False):
        """(max elapsed over ranks, total samples, sum of per-rank output checksums[, min elapsed over ranks]).  Two tiny
        all-reduces (MAX and SUM) of fp64 scalars: latency-bound, link bandwidth irrelevant.  MAX over (elapsed, -elapsed)
        yields the slowest and the fastest rank in the same collective."""
        if not self._grouped():
            out = (float(elapsed_s), float(samples_done), float(checksum))
            return out + (float(elapsed_s),) if per_rank else out
        dev = self._device()
        t = torch.tensor([elapsed_s, -elapsed_s], dtype=torch.float64, device=dev)
        s = torch.tensor([samples_done, checksum], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        out = (float(t[0]), float(s[0]), float(s[1]))
        return out + (-float(t[1]),) if per_rank else out

    def gather_results(self, local_results, n_samples):
        """All ranks' per-sample results in global sample order (mirrors multi_gpu_test(gpu_collect=True),
        val.py:132): local_results[i] belongs to global sample rank + i*world."""
        if not self._grouped():
            return list(local_results)[:n_samples]
        bucket = [None] * self.world
        dist.all_gather_object(bucket, list(local_results))
        out = [None] * n_samples
        for r, items in enumerate(bucket):
            for i, item in enumerate(items):
                gi = r + i * self.world
                if gi < n_samples:
                    out[gi] = item
        return out

    def shutdown(self):
        if self._grouped():
            dist.destroy_process_group()
