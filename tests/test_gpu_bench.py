"""bench.py end to end on the GPU box: the single-process line, and the N > 1 launch exactly as the driver issues it
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), here with both ranks sharing GPU 0 and the
metric reduction over gloo (SBEV_SHARE_GPU=1: a one-GPU box cannot give RCCL two devices)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_single_process_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-alt'], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['value'] > 0
    assert d['roofline']['bound'] == 'hbm' and d['roofline']['achieved'] > 0 and 0 < d['roofline']['frac']
    assert abs(d['value'] - 1e3 / d['ms_per_step']) < 0.01 * d['value']          # bs 1: samples/s = 1 / step time
    # roofline.traffic is measured in the run (bench.py re-runs itself under rocprofv3 --pmc) when rocprofv3 is on the box
    import shutil
    if shutil.which('rocprofv3'):
        assert d['roofline']['traffic_source'].startswith('measured in this run'), d['roofline']['traffic_source']
        assert 1.5e8 < d['roofline']['traffic'] < 3.5e8 and d['roofline']['frac'] <= 1.0


def test_two_ranks_as_the_driver_launches_them():
    env = dict(os.environ, SBEV_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '2']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 0.01 * d['value']
    assert 'cpu_baseline' not in d or d['cpu_baseline'] is None      # the CPU leg runs at N = 1 only
