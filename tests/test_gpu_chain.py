"""The row-chain kernels (csrc/row_chain.hip: position encoder, attention projections, norms, ffn, branches, refine_bbox of a
layer as three launches with the rows in LDS) against the op-by-op launches they replace -- which the other decoder tests
pin to the reference recordings / the oracle -- and the lane-ordered weight image against a numpy restatement of its
layout.  The chains add the k-halves of a Linear in another order than the small-tile GEMM: agreement is fp32 round-off
(a few 1e-6 on O(1) outputs after one layer), not bit for bit; free-running layers amplify it ~7x per layer like any
rounding difference, so deeper layers are bounded loosely here and tightly layer by layer."""
import copy

import numpy as np
import pytest
import torch

from conftest import op_by_op_runtime
from sparsebev_amd import runtime, synthetic as S
from sparsebev_amd.transformer import SparseBEVTransformer, FeaturePyramid, DecoderContext

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def build(T, L, seed, num_layers):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=num_layers, num_levels=L, num_classes=10, code_size=10,
                             pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    return m.to(DEV).eval(), params


def inputs(B, Q, T, pyr, seed):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seed)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=seed + 1)]
    return feats, bbox, feat, S.make_img_metas(B, T, ih, iw), len(sizes)


def both(model, bbox, feat, feats, metas, mask=None):
    fl = (lambda: feats) if hasattr(feats, 'levels') else (lambda: list(feats))
    a = model(bbox, feat, fl(), mask, copy.deepcopy(metas))
    with op_by_op_runtime():
        b = model(bbox, feat, fl(), mask, copy.deepcopy(metas))
    return a, b


def packed_reference(W):
    """[N, K] -> [ceil(N/64)][K/16][4][64 lanes][4]: lane = column of the group, (step, m, e) -> k = 16 step + 4 m + e."""
    N, K = W.shape
    out = np.zeros(((N + 63) // 64, K // 16, 4, 64, 4), np.float32)
    for cg in range(out.shape[0]):
        rows = W[cg * 64:(cg + 1) * 64]
        out[cg, :, :, :rows.shape[0], :] = rows.reshape(rows.shape[0], K // 16, 4, 4).transpose(1, 2, 0, 3)
    return out.reshape(-1)


def test_weight_image_layout():
    model, _ = build(8, 4, 3, 1)
    feats, bbox, feat, metas, _ = inputs(1, 16, 8, 'tiny', 5)
    model(bbox, feat, feats, None, metas)                         # binds the runtime
    keep = model.decoder._runtime._keep
    image = keep['chain_pack'].cpu().numpy()
    order = ['ffn0_w', 'ffn1_w', 'cls0_w', 'reg0_w', 'cls3_w', 'reg2_w', 'cls6_w', 'reg4_w', 'pe3_w', 'attn_in_w', 'attn_out_w', 'samp_w']
    off = 0
    for name in order:
        want = packed_reference(keep[name].cpu().numpy())
        assert np.array_equal(image[off:off + want.size], want), name
        off += want.size
    # the small vectors behind the matrices: [tail | front | attention] blocks, zero-padded slots
    vec = image[off:]
    assert np.array_equal(vec[:256], keep['op_b'].cpu().numpy())
    assert np.array_equal(vec[768:1280], keep['ffn0_b'].cpu().numpy())
    assert np.array_equal(vec[3584:3594], keep['cls6_b'].cpu().numpy()) and not vec[3594:3648].any()
    assert np.array_equal(vec[4224:4224 + 768], keep['pe0_w'].cpu().numpy().reshape(-1))


@pytest.mark.parametrize('B,Q,T,pyr,layers', [(1, 49, 2, 'tiny', 1), (2, 49, 2, 'tiny', 3), (1, 900, 8, 'tiny', 2), (1, 100, 1, 'tiny5', 2),
                                              (1, 1024, 4, 'tiny', 2), (1, 9, 4, 'tiny', 6),
                                              # 8 rows per workgroup (1025 .. 2048 rows): partial tile, the 1600-query shape, the limit
                                              (1, 1089, 2, 'tiny', 2), (1, 1600, 2, 'tiny5', 3), (2, 1024, 2, 'tiny', 2),
                                              # 16 rows per workgroup (2049 .. 4096): partial tile, the batch configs' 3200 / 3600 rows
                                              (1, 2116, 2, 'tiny', 2), (8, 400, 2, 'tiny', 3), (4, 900, 2, 'tiny5', 2), (1, 4096, 1, 'tiny', 2)])
def test_row_chains_equal_op_by_op_to_roundoff(B, Q, T, pyr, layers):
    """98 rows (a partial 4-row tile), 1024 rows (the largest launch with 4 rows per workgroup), T = 1 (no velocity division),
    5 levels, 1 .. 6 layers (front-only, attention, tail, tail + next front launches); 8 and 16 rows per workgroup (round 3)."""
    feats, bbox, feat, metas, L = inputs(B, Q, T, pyr, 11)
    model, _ = build(T, L, 12, layers)
    a, b = both(model, bbox, feat, feats, metas)
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    # Free-running, the two paths differ by fp32 summation order only; a random-init layer amplifies any input difference (rounding
    # noise included) by up to ~8x (2e-6 -> 2.6e-2 over 6 layers between two CPU implementations).  So: one
    # layer deep 2e-5, two deep 3e-4, and from there every layer at most 10x the previous layer's deviation -- a defect of the
    # deeper launches (tail + next front) would break the growth law, which a flat bound could not see.  The one-layer-deep
    # comparison of EVERY layer is test_row_chains_every_layer_from_the_same_inputs.
    prev = 0.0
    for l in range(layers):
        dev = max((a[0][l] - b[0][l]).abs().max().item(), (a[1][l] - b[1][l]).abs().max().item())
        tol = 2e-5 if l == 0 else 3e-4 if l == 1 else 10.0 * max(prev, 3e-5)
        assert dev < tol, (l, dev, prev)
        prev = dev
    # run-to-run bit determinism, and rows do not depend on which workgroup / tile position computes them
    a2 = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])


@pytest.mark.parametrize('B,Q,T,pyr,layers', [(1, 1, 2, 'tiny', 2), (1, 4, 2, 'tiny', 1), (1, 9, 2, 'tiny', 3), (1, 900, 8, 'tiny', 6),
                                              (2, 441, 2, 'tiny', 2), (1, 961, 2, 'tiny5', 3), (1, 1024, 2, 'tiny', 2), (1, 1089, 2, 'tiny', 2)])
def test_tail_on_pairs_of_workgroups_equals_the_single_workgroup_tail(B, Q, T, pyr, layers):
    """Round 4: two workgroups per row block in the tail (ffn split by columns / k with an exchange of partial sums, one branch
    each, the x rows handed over).  1 .. 9 rows (one pair, partial), 900 (config 2), 961 with 5 levels, 1024 (256 blocks of
    8-row pairs: the whole device), 1089 (too many pairs for one round of 256 CUs: the single-workgroup tail, bit-identical
    either way).  Same growth law as above against the single-workgroup chains, no poll ran out, bit-stable run to run."""
    feats, bbox, feat, metas, L = inputs(B, Q, T, pyr, 71)
    model, _ = build(T, L, 72, layers)
    t0 = runtime.chain_pair_timeouts()
    a = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    prev_setting = runtime.chain_pair(False)
    try:
        b = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    finally:
        runtime.chain_pair(prev_setting)
    assert prev_setting is True
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    if B * Q > 1024:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    prev = 0.0
    for l in range(layers):
        dev = max((a[0][l] - b[0][l]).abs().max().item(), (a[1][l] - b[1][l]).abs().max().item())
        tol = 2e-5 if l == 0 else 3e-4 if l == 1 else 10.0 * max(prev, 3e-5)
        assert dev < tol, (l, dev, prev)
        prev = dev
    for _ in range(3):
        a2 = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
        assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])
    assert runtime.chain_pair_timeouts() == t0


def test_tail_pairs_under_uneven_load_and_graph_replay():
    """The hand-offs under load: 200 replays of the captured 6-layer step at config 2's row count while a second stream keeps part
    of the device busy with long-running copies, every output word compared with the first replay (a stale exchange row, a lost
    arrival or a counter that was not re-zeroed shows as a difference or a timeout)."""
    B, Q, T = 1, 900, 8
    feats, bbox, feat, metas, L = inputs(B, Q, T, 'tiny', 81)
    model, _ = build(T, L, 82, 6)
    t0 = runtime.chain_pair_timeouts()
    ref = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=DEV)
    for i in range(200):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk2 = junk * 1.0001 + 1.0           # ~0.5 GB of traffic beside the step
        out = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), i
    torch.cuda.synchronize()
    del junk2
    assert runtime.chain_pair_timeouts() == t0


def test_a_lost_pair_member_times_out_instead_of_hanging():
    """The poll bound: with the test hook one member of pair 0 leaves at once; its partner must give up after the bound (~1 s per
    hand-off), count a timeout and finish -- rows 8 .. of a ONE-layer run are untouched (rows are independent within the tail; the
    next layer's attention would mix the broken rows in).  Round 5: the fault is OBSERVABLE on the normal path -- a sticky
    word in pinned host memory; ``runtime.check_pair_faults()`` after the caller's own synchronisation raises for the step just
    finished, and without that the NEXT decoder call refuses (SBEV_EFAULT -> PairFaultError) instead of computing on top of 8 wrong
    rows; either way pair mode is off afterwards and the repeated step is right."""
    import time
    from sparsebev_amd import _lib
    feats, bbox, feat, metas, L = inputs(1, 100, 2, 'tiny', 91)
    model, _ = build(2, L, 92, 1)
    model.decoder.static_graph = False          # (a captured step would keep the hook's setting of the call it was recorded in)
    ref = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    t0 = runtime.chain_pair_timeouts()
    lib = _lib.load()
    assert lib.sbev_decoder_chain_pair_faults() == 0
    try:
        for how in ('check after sync', 'next call refuses'):
            assert runtime.chain_pair(True) in (True, False)
            assert lib.sbev_debug_chain_pair_drop(1) == 0
            try:
                tic = time.time()
                out = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
                torch.cuda.synchronize()
                took = time.time() - tic
            finally:
                assert lib.sbev_debug_chain_pair_drop(0) == 1
            assert took < 60.0
            assert runtime.chain_pair_timeouts() > t0
            assert lib.sbev_decoder_chain_pair_faults() > 0          # visible WITHOUT a device synchronisation of its own
            assert torch.equal(out[0][:, :, 8:], ref[0][:, :, 8:]) and torch.equal(out[1][:, :, 8:], ref[1][:, :, 8:])
            with pytest.raises(_lib.PairFaultError):
                if how == 'check after sync':
                    runtime.check_pair_faults()
                else:
                    model(bbox, feat, list(feats), None, copy.deepcopy(metas))
            assert lib.sbev_decoder_chain_pair_faults() == 0         # acknowledged ...
            assert runtime.chain_pair(False) is False                # ... and pair mode went off with the fault
            again = model(bbox, feat, list(feats), None, copy.deepcopy(metas))      # the repeated step: single-workgroup tail, right again
            assert (again[0] - ref[0]).abs().max().item() < 2e-5 and (again[1] - ref[1]).abs().max().item() < 2e-5
            t0 = runtime.chain_pair_timeouts()
    finally:
        runtime.chain_pair(True)
    exact = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert torch.equal(exact[0], ref[0]) and torch.equal(exact[1], ref[1])


@pytest.mark.parametrize('B,Q,T,pyr,layers', [(1, 4, 2, 'tiny', 1), (1, 100, 2, 'tiny', 2), (1, 900, 8, 'tiny', 6), (2, 441, 2, 'tiny', 2),
                                              (1, 1024, 2, 'tiny', 2), (1, 1089, 2, 'tiny', 2)])
def test_out_projection_fold_inside_the_launch_is_bit_identical(B, Q, T, pyr, layers):
    """Round 6: the out-projection's chunk-workgroups fold their split-K slabs inside the launch (gemm_bf16s.hip, out4 kernel: write-through
    slabs, one counter per row tile, ceil(rows / S) rows per chunk-workgroup summed in slab order) and the tail reads one row block.  Same
    summation order as the tail's own: outputs equal bit for bit with the fold off, eager and replayed, run to run; no poll ran out.
    1089 rows: more workgroups than CUs -- the fold is not taken (bit-identical trivially)."""
    feats, bbox, feat, metas, L = inputs(B, Q, T, pyr, 171)
    model, _ = build(T, L, 172, layers)
    t0 = runtime.chain_pair_timeouts()
    b = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    prev = runtime.out_fold(True)              # (an A/B switch, off by default: measured slower at config 2 -- DESIGN.md section 4.4)
    try:
        a = [[t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))] for _ in range(4)]      # eager, capture, replays
    finally:
        runtime.out_fold(prev)
    for x in a:
        assert torch.equal(x[0], b[0]) and torch.equal(x[1], b[1])
    assert runtime.chain_pair_timeouts() == t0


def test_fold_under_load_and_graph_replay():
    """200 replays of the captured 6-layer step at config 2's row count beside a second stream's traffic: every output word equal to the first
    replay (a stale slab, a lost arrival or a counter that was not re-zeroed would show as a difference or a timeout)."""
    feats, bbox, feat, metas, L = inputs(1, 900, 8, 'tiny', 181)
    model, _ = build(8, L, 182, 6)
    t0 = runtime.chain_pair_timeouts()
    ref = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    prev = runtime.out_fold(True)
    try:
        side = torch.cuda.Stream()
        junk = torch.randn(64 << 20, device=DEV)
        for i in range(200):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    junk2 = junk * 1.0001 + 1.0
            out = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), i
        torch.cuda.synchronize()
        del junk2
    finally:
        runtime.out_fold(prev)
    assert model.decoder._runtime.step_graphs.replays >= 150
    assert runtime.chain_pair_timeouts() == t0


def test_a_lost_fold_member_times_out_and_raises_the_fault_word():
    """The fold's poll bound: with the test hook chunk 1 of row tile 0 never arrives; the other chunk-workgroups of that tile give up after
    the bound, count a timeout and raise the host-mapped fault word (rows of tile 0 are wrong: the step is invalid); the next call
    refuses (PairFaultError), the in-launch hand-offs go off, the repeated step is right."""
    import time
    from sparsebev_amd import _lib
    feats, bbox, feat, metas, L = inputs(1, 289, 2, 'tiny', 191)
    model, _ = build(2, L, 192, 1)
    model.decoder.static_graph = False
    ref = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    lib = _lib.load()
    t0 = runtime.chain_pair_timeouts()
    assert lib.sbev_decoder_chain_pair_faults() == 0
    prev_fold = runtime.out_fold(True)
    try:
        assert lib.sbev_debug_out_fold_drop(1) == 0
        try:
            tic = time.time()
            out = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
            torch.cuda.synchronize()
            took = time.time() - tic
        finally:
            assert lib.sbev_debug_out_fold_drop(0) == 1
        assert took < 120.0
        assert runtime.chain_pair_timeouts() > t0
        assert lib.sbev_decoder_chain_pair_faults() > 0
        with pytest.raises(_lib.PairFaultError):
            model(bbox, feat, list(feats), None, copy.deepcopy(metas))
        assert lib.sbev_decoder_chain_pair_faults() == 0
        again = model(bbox, feat, list(feats), None, copy.deepcopy(metas))          # hand-offs off: slabs summed by the single-workgroup tail
        assert (again[0] - ref[0]).abs().max().item() < 2e-5 and (again[1] - ref[1]).abs().max().item() < 2e-5
    finally:
        lib.sbev_debug_out_fold_drop(0)
        lib.sbev_decoder_chain_pair_faults_ack()
        runtime.chain_pair(True)
        runtime.out_fold(prev_fold)
    exact = model(bbox, feat, list(feats), None, copy.deepcopy(metas))
    assert torch.equal(exact[0], ref[0]) and torch.equal(exact[1], ref[1])


def test_pair_fault_word_through_the_c_abi():
    """sbev_decoder_forward itself refuses while a fault stands (pure-C callers have no Python runtime around them): SBEV_EFAULT and
    a message, pair mode switched off by the refusing call, calls accepted again after sbev_decoder_chain_pair_faults_ack()."""
    import ctypes
    from sparsebev_amd import _lib
    feats, bbox, feat, metas, L = inputs(1, 64, 2, 'tiny', 93)
    model, _ = build(2, L, 94, 1)
    model.decoder.static_graph = False
    lib = _lib.load()
    ref = [t.clone() for t in model(bbox, feat, list(feats), None, copy.deepcopy(metas))]
    rt = model.decoder._runtime
    ctx = DecoderContext(copy.deepcopy(metas), 1, bbox.device)
    pyr = FeaturePyramid(list(feats))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        assert lib.sbev_debug_chain_pair_drop(1) == 0
        try:
            args, keep, cls, box = rt._prepare(bbox, feat, pyr, ctx, None)
            assert lib.sbev_decoder_forward(*args, stream) == 0          # the faulting step: accepted (nothing is known yet)
            torch.cuda.synchronize()
        finally:
            lib.sbev_debug_chain_pair_drop(0)
        n = lib.sbev_decoder_chain_pair_faults()
        assert n > 0
        args, keep, cls, box = rt._prepare(bbox, feat, pyr, ctx, None)
        assert lib.sbev_decoder_forward(*args, stream) == _lib.EFAULT      # sticky: every call refuses ...
        assert b'timed out' in lib.sbev_last_error()
        assert lib.sbev_decoder_forward(*args, stream) == _lib.EFAULT
        assert lib.sbev_decoder_chain_pair(1) == 0                        # ... and the first refusal switched pair mode off
        assert lib.sbev_decoder_chain_pair_faults_ack() == n
        assert lib.sbev_decoder_chain_pair_faults() == 0
        assert lib.sbev_decoder_forward(*args, stream) == 0               # acknowledged: accepted again (pair mode back on by the line above)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(cls), ref[0]) and torch.equal(torch.nan_to_num(box), ref[1])
        assert lib.sbev_decoder_chain_pair_faults() == 0
    finally:
        lib.sbev_decoder_chain_pair_faults_ack()
        runtime.chain_pair(True)


def test_row_chains_every_layer_from_the_same_inputs():
    """Layer by layer at the config-2 query count: every layer's chain launches from the op-by-op path's own inputs, so
    each comparison is one layer deep (2e-5), including the tail + next-front launch whose output (x, qkvt) only the NEXT
    layer shows: a 2-layer run whose layer 0 input is layer l's op-by-op state."""
    B, Q, T = 1, 900, 8
    feats, bbox, feat, metas, L = inputs(B, Q, T, 'tiny', 21)
    model6, _ = build(T, L, 22, 6)
    model2, _ = build(T, L, 22, 2)
    layer = model6.decoder.decoder_layer
    pyr, ctx = FeaturePyramid(feats), DecoderContext(metas, B, torch.device(DEV))
    qb, qf = bbox, feat
    for l in range(5):
        a, b = both(model2, qb, qf, pyr, metas)
        assert (a[0][0] - b[0][0]).abs().max() < 2e-5 and (a[1][0] - b[1][0]).abs().max() < 2e-5, l
        assert (a[0][1] - b[0][1]).abs().max() < 3e-4 and (a[1][1] - b[1][1]).abs().max() < 3e-4, l
        qf, _, qb = layer(qb, qf, pyr, None, ctx)               # the op-by-op layer's outputs feed the next comparison


def test_row_chains_with_denoising_mask_and_bf16x3_gemms():
    B, Q, T = 1, 64, 2
    feats, bbox, feat, metas, L = inputs(B, Q, T, 'tiny', 31)
    model, _ = build(T, L, 32, 2)
    mask = torch.zeros(Q, Q, dtype=torch.bool)
    mask[:48, 48:] = True
    mask[48:, :48] = True
    a, b = both(model, bbox, feat, feats, metas, mask.to(DEV))
    assert (a[0][0] - b[0][0]).abs().max() < 2e-5 and (a[0][1] - b[0][1]).abs().max() < 3e-4
    model.decoder.gemm_mode = runtime.GEMM_BF16X3
    a, b = both(model, bbox, feat, feats, metas)
    assert (a[0][0] - b[0][0]).abs().max() < 2e-5 and (a[1][0] - b[1][0]).abs().max() < 2e-5


def test_weight_image_follows_parameter_updates():
    """the image is re-packed when a parameter changes in place (optimizer step, load_state_dict)"""
    feats, bbox, feat, metas, L = inputs(1, 49, 2, 'tiny', 41)
    model, _ = build(2, L, 42, 1)
    a0, _ = both(model, bbox, feat, feats, metas)
    with torch.no_grad():
        model.decoder.decoder_layer.ffn.layers[1].bias.add_(0.5)
        model.decoder.decoder_layer.cls_branch[6].weight.mul_(2.0)
    a1, b1 = both(model, bbox, feat, feats, metas)
    assert (a1[0] - a0[0]).abs().max() > 1e-2
    assert (a1[0] - b1[0]).abs().max() < 2e-5 and (a1[1] - b1[1]).abs().max() < 2e-5


def test_large_batches_keep_the_op_by_op_launches():
    """above 4096 rows (more than one round of 16-row workgroups) the op-by-op launches stay (csrc/row_chain.hip: row_chain_pays):
    same launches either way"""
    feats, bbox, feat, metas, L = inputs(11, 400, 2, 'tiny', 51)
    model, _ = build(2, L, 52, 2)
    a, b = both(model, bbox, feat, feats, metas)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize('B,Q', [(1, 36), (1, 1089), (3, 729)])
def test_two_layer_chain_against_the_oracle(B, Q):
    """4, 8 and 16 rows per workgroup (36, 1089, 2187 rows: each with a partial last workgroup)"""
    from oracle import sparsebev_oracle as O
    T = 2
    ih, iw, sizes = S.PYRAMIDS['tiny']
    feats = S.make_features(B, T, sizes, seed=61)
    bbox, feat = S.make_queries(B, Q, seed=62)
    metas = S.make_img_metas(B, T, ih, iw)
    model, params = build(T, len(sizes), 63, 2)
    cls, box = model(bbox.to(DEV), feat.to(DEV), [f.to(DEV) for f in feats], None, copy.deepcopy(metas))
    ref_cls, ref_box, _ = O.decoder(params, bbox, feat, feats, metas, S.PC_RANGE, num_layers=2, sampler=O.msmv_sampling_kernel_semantics)
    # per query: a sample point within rounding of an image border may flip its in-view flag against the CPU oracle --
    # at most 3 such queries per 1000; every other query 1e-4 (layer 1) / 2e-3 (free-running layer 2)
    for l, tol in ((0, 1e-4), (1, 2e-3)):
        err = torch.maximum((cls[l].cpu() - ref_cls[l]).abs().amax(-1), (box[l].cpu() - ref_box[l]).abs().amax(-1)).reshape(-1)
        bad = int((err >= tol).sum())
        assert bad <= (3 * B * Q) // 1000, (l, bad, err.max().item())


@pytest.mark.parametrize('tag', ['c1', 'c2small', 'L5'])
def test_row_chains_against_the_reference_recordings(tag):
    """Fixture G7 (the reference decoder's own per-layer inputs and outputs): every layer through the chain kernels from the
    REFERENCE's inputs of that layer -- 1-layer runs (front, attention chain, tail) and 2-layer runs (tail + the next layer's
    front in one launch: its x / qkvt only show in layer 2) -- against the reference's recorded outputs at 1e-4."""
    from conftest import load_golden
    g = load_golden('g7_decoder_' + tag)
    B, Q, T, L = [int(v) for v in g['cfg']]
    seeds = [int(v) for v in g['seeds']]
    ih, iw, sizes = S.PYRAMIDS[str(g['pyramid'])]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seeds[2])]
    metas = S.make_img_metas(B, T, ih, iw)
    for b, m in enumerate(metas):
        m['img_timestamp'] = [float(v) for v in g['timestamps'][b]]
    model1, _ = build(T, L, seeds[0], 1)
    model2, _ = build(T, L, seeds[0], 2)
    assert model1.decoder._runtime is None
    n = g['out_cls'].shape[0]
    ins = [(g['query_bbox'], g['query_feat'])] + [(g['out_bbox'][i - 1], g['out_feat'][i - 1]) for i in range(1, n)]
    for i, (qb, qf) in enumerate(ins):
        c1, b1 = model1(qb.to(DEV), qf.to(DEV), list(feats), None, copy.deepcopy(metas))
        assert (c1[0].cpu() - g['out_cls'][i]).abs().max() < 1e-4 and (b1[0].cpu() - g['out_bbox'][i]).abs().max() < 1e-4, i
        if i + 1 < n:
            c2, b2 = model2(qb.to(DEV), qf.to(DEV), list(feats), None, copy.deepcopy(metas))
            assert (c2[1].cpu() - g['out_cls'][i + 1]).abs().max() < 1e-4 and (b2[1].cpu() - g['out_bbox'][i + 1]).abs().max() < 1e-4, i
    assert 'chain_pack' in model1.decoder._runtime._keep                       # the chains did run
