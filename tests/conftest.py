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


def free_running_bound(tag):
    """Per-layer bound for a FREE-RUNNING 6-layer comparison against G7's recording, from fixture G13 (tests/golden/make_golden.py::
    main_yardstick): 2 x the divergence the REFERENCE shows against itself on the same inputs -- the envelope of (its two samplers:
    native grid_sample path vs MSMV_CUDA path on the CUDA kernel's semantics) and (query_feat nudged by one fp32 ulp).  An
    implementation whose free-running drift stays inside it is as close to the reference as the reference is to itself.
    Returns {'cls' | 'bbox' | 'feat': float64 array [layers]}, the yardstick itself under 'yard_*'."""
    g = load_golden('g13_yardstick_' + tag)
    out = {}
    for what in ('cls', 'bbox', 'feat'):
        yard = np.maximum(g['div_%s_kernel' % what].numpy(), g['div_%s_ulp' % what].numpy())
        out['yard_' + what] = yard
        out[what] = 2.0 * yard
    return out


def report_free_running(label, tag, got, ref, bound):
    """max-abs divergence per layer of (cls, bbox[, feat]) against the reference recording, printed beside the G13 yardstick, asserted
    against 2 x it.  got / ref: tuples of [layers, ...] tensors in the order cls, bbox[, feat]."""
    names = ('cls', 'bbox', 'feat')[:len(got)]
    for what, a, b in zip(names, got, ref):
        d = np.array([(a[i].detach().cpu().double() - b[i].double()).abs().max().item() for i in range(b.shape[0])])
        print('free-running %-12s %-8s %-4s divergence per layer %s | reference-vs-itself (G13) %s'
              % (label, tag, what, ' '.join('%.1e' % v for v in d), ' '.join('%.1e' % v for v in bound['yard_' + what])))
        bad = [(i, d[i], bound[what][i]) for i in range(len(d)) if not d[i] <= bound[what][i]]
        assert not bad, '%s %s %s: layer divergence above 2 x the reference\'s own (layer, got, bound): %s' % (label, tag, what, bad)
