import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


def load_golden(name):
    """npz fixture -> dict of torch tensors (numeric arrays) / numpy (everything else)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in 'fiub' and a.ndim > 0 else a
    return out


def feats_of(g):
    n = len([k for k in g if k.startswith('feat') and k[4:].isdigit()])
    return [g['feat%d' % i] for i in range(n)]


def assert_same_with_nonfinite(got, ref, tol, what='', kinds=True):
    """Outputs that legitimately hold Inf / NaN (fixture G12): the SAME elements must be non-finite -- with kinds=True also of the same kind
    (NaN, +Inf, -Inf) -- and the finite ones agree to `tol`."""
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), fin), '%s: non-finite positions differ (%d vs %d)' % (what, int((~torch.isfinite(got)).sum()), int((~fin).sum()))
    if kinds:
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), '%s: NaN positions differ (%d vs %d NaNs)' % (what, int(torch.isnan(got).sum()), int(torch.isnan(ref).sum()))
        assert torch.equal(torch.isposinf(got), torch.isposinf(ref)) and torch.equal(torch.isneginf(got), torch.isneginf(ref)), what + ': Inf positions differ'
    assert fin.any() and (~fin).any(), what + ': the fixture must hold both finite and non-finite outputs'
    assert (got[fin] - ref[fin]).abs().max() < tol, what


@pytest.fixture(scope='session')
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


import contextlib


@contextlib.contextmanager
def op_by_op_runtime():
    """sbev_decoder_forward with one launch per op (row chains off): exactly the launches of the layer-by-layer Python
    path, so the two agree bit for bit; the row-chain kernels (csrc/row_chain.hip) sum in another order and agree to
    fp32 round-off (tests/test_gpu_chain.py)."""
    from sparsebev_amd import runtime
    runtime.row_chain(False)
    try:
        yield
    finally:
        runtime.row_chain(True)


def runtime_op_by_op(model, *args, exact_gemm=False, **kw):
    """the C++ runtime with one launch per op.  exact_gemm: in the 'f32' mode of the two mixing GEMMs -- the kernels the layer-by-layer
    Python path (layerwise=True, a debugging path) launches, so that the two can be compared bit for bit; the default mode (fp16 hi + lo
    split) is compared with the oracle / the recordings instead."""
    dec = getattr(model, 'decoder', model)
    mode = dec.gemm_mode
    if exact_gemm:
        dec.gemm_mode = 'f32'
    try:
        with op_by_op_runtime():
            return model(*args, **kw)
    finally:
        dec.gemm_mode = mode


@pytest.fixture(autouse=True)
def _inference_mode_by_default():
    """The parity tests exercise the inference path (the reference's val.py / timing.py run under no_grad); with grad
    enabled the drop-in module takes its differentiable path, exactly like the reference's nn.Module.  Training tests
    switch grad back on with ``torch.enable_grad()``."""
    with torch.no_grad():
        yield
