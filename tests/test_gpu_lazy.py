"""On-demand ("lazy") relayout (csrc/layout.hip, sample_point.hpp::touch_units, sbev_decoder_forward_lazy): the step moves only the
64-pixel x 64-channel feature units its sample points read.  The reference regroups everything (models/sparsebev_transformer.py:73-85)
and reads under half of it (sparsebev_sampling.py:88-109, msmv_sampling_forward.cu:41-66).  Pinned here:
  * the move kernels move exactly the marked units, once per step, and leave everything else alone;
  * the marks are exactly the corners the sampler's own rule reads (numpy restatement of msmv_chunk.inc phase 1);
  * decoder outputs are BIT-IDENTICAL to dense relayout + sbev_decoder_forward -- eager and replayed, row chains and op by op, fp32 and
    2-byte storage, destination buffers poisoned with NaN (a tap on an unmoved unit would surface as NaN) -- also at the full c2 / c3 / c4
    shapes, and when every unit is touched."""
import copy
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from conftest import has_gpu, op_by_op_runtime

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]

from sparsebev_amd import _lib, runtime, synthetic as S  # noqa: E402
from sparsebev_amd import transformer as TR  # noqa: E402
from sparsebev_amd.transformer import SparseBEVTransformer  # noqa: E402

DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _vp_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _lazy_move(src, out, hw, need, done, first, last, dtype_code):
    lib = _lib.load()
    n = len(src)
    c_hw = (ctypes.c_int32 * n)(*hw)
    st = lib.sbev_nchw_to_nhwc_lazy(None, None, _vp_array(src), _vp_array(out), n, c_hw, src[0].shape[0], src[0].shape[1], dtype_code,
                                    ctypes.c_void_p(need.data_ptr()), ctypes.c_void_p(done.data_ptr()), int(first), int(last), _stream())
    _lib.check(st, 'sbev_nchw_to_nhwc_lazy')


def _unit_mask(flags_u8, hw, n_img, base):
    """flags [tiles, 4] (bool) -> per level bool mask [n_img, S_l, 256] of the elements of marked units"""
    masks = []
    for l, s in enumerate(hw):
        tiles = (s + 63) // 64
        f = flags_u8[base[l]:base[l] + n_img * tiles].reshape(n_img, tiles, 4)
        m = f[:, :, None, :, None].expand(n_img, tiles, 64, 4, 64).reshape(n_img, tiles * 64, 256)[:, :s]
        masks.append(m)
    return masks


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_move_kernels_move_exactly_the_marked_units_once(dtype):
    lib = _lib.load()
    n_img, R = 5, 256
    hw = [16 * 44, 176, 250, 12, 3]                # two vector levels, odd plane sizes (scalar reads), a level smaller than a tile
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    g = torch.Generator(device=DEV).manual_seed(1)
    src = [torch.randn(n_img, R, s, device=DEV, generator=g).to(dtype) for s in hw]
    total = int(lib.sbev_lazy_relayout_tiles(len(hw), (ctypes.c_int32 * len(hw))(*hw), n_img, R))
    tiles = [(s + 63) // 64 for s in hw]
    base = np.concatenate([[0], np.cumsum([n_img * t for t in tiles])])
    assert total == base[-1]
    SENT = 7.0
    out = [torch.full((n_img, s, R), SENT, device=DEV, dtype=dtype) for s in hw]
    dense = [f.permute(0, 2, 1).contiguous() for f in src]
    # step 1, first launch: a third of the units marked (bytes of any non-zero value), done holds garbage
    need_b = (torch.rand(total, 4, device=DEV, generator=g) < 0.33)
    need = (need_b.to(torch.uint8) * torch.randint(1, 255, (total, 4), device=DEV, generator=g, dtype=torch.uint8)).contiguous()
    done = torch.randint(0, 255, (total, 4), device=DEV, generator=g, dtype=torch.uint8)
    _lazy_move(src, out, hw, need.view(torch.int32), done.view(torch.int32), True, False, code)
    m1 = _unit_mask(need_b, hw, n_img, base)
    for l in range(len(hw)):
        assert torch.equal(out[l][m1[l]], dense[l][m1[l]]), 'level %d: marked units' % l
        assert bool((out[l][~m1[l]] == SENT).all()), 'level %d: unmarked units were written' % l
    assert torch.equal(done, need_b.to(torch.uint8)), 'done != (need != 0) after the first launch'
    assert torch.equal(need != 0, need_b), 'the first launch must not clear need unless it is also the last'
    # same step, later launch: more marks; the SOURCE changes in between -- units moved before must not be moved again
    src2 = [f + 1 for f in src]
    dense2 = [f.permute(0, 2, 1).contiguous() for f in src2]
    more = (torch.rand(total, 4, device=DEV, generator=g) < 0.2)
    need2_b = need_b | more
    need.copy_(need2_b.to(torch.uint8))
    _lazy_move(src2, out, hw, need.view(torch.int32), done.view(torch.int32), False, False, code)
    new = _unit_mask(more & ~need_b, hw, n_img, base)
    both = _unit_mask(need2_b, hw, n_img, base)
    for l in range(len(hw)):
        assert torch.equal(out[l][new[l]], dense2[l][new[l]]), 'level %d: newly marked units' % l
        assert torch.equal(out[l][m1[l]], dense[l][m1[l]]), 'level %d: a unit moved earlier in the step was moved again' % l
        assert bool((out[l][~both[l]] == SENT).all())
    assert torch.equal(done, need2_b.to(torch.uint8))
    # the step's last launch clears need (nothing new: nothing moves)
    before = [o.clone() for o in out]
    _lazy_move(src2, out, hw, need.view(torch.int32), done.view(torch.int32), False, True, code)
    assert int(need.sum()) == 0 and all(torch.equal(a, b) for a, b in zip(before, out))
    # every unit marked == the dense relayout (first and last in one launch: a one-layer step)
    need.fill_(1)
    _lazy_move(src, out, hw, need.view(torch.int32), done.view(torch.int32), True, True, code)
    assert all(torch.equal(a, b) for a, b in zip(out, dense)) and int(need.sum()) == 0 and bool((done == 1).all())


def test_marks_are_exactly_the_corners_the_sampler_reads():
    """sbev_sample_and_project_touch against the numpy restatement of the sampler's corner rule (tools/relayout_footprint.py), fed with
    the kernel's OWN loc output: flags equal, unit for unit; and loc / weights unchanged by the marking."""
    import relayout_footprint as RF
    lib = _lib.load()
    B, Q, T, N, G, P = 2, 100, 2, 6, 4, 4
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    L = len(sizes)
    bbox, _ = S.make_queries(B, Q, seed=4)
    bbox = bbox.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    so = torch.randn(B * Q, G * P * (3 + L), device=DEV, generator=g)
    so[:, :G * P * 3] *= 2.0                       # offsets of up to a few box sizes: points in, around and outside the images
    ctx = TR.DecoderContext(S.make_img_metas(B, T, ih, iw), B, DEV)
    pc = (ctypes.c_double * 6)(*S.PC_RANGE)
    hw_px = [h * w for h, w in sizes]
    total = int(lib.sbev_lazy_relayout_tiles(L, (ctypes.c_int32 * L)(*hw_px), B * T * N, 256))
    need = torch.zeros(total, 4, device=DEV, dtype=torch.uint8)
    outs = []
    for touch in (False, True):
        loc = torch.empty(B * T * G, Q, P, 3, device=DEV)
        wbp = torch.empty(B * G * T, Q, P, L, device=DEV)
        a = (_p(bbox), _p(so), so.shape[1], ctypes.c_void_p(so.data_ptr() + 4 * G * P * 3), so.shape[1], _p(ctx.time_diff), _p(ctx.lidar2img), pc,
             B, Q, T, N, G, P, L, float(ih), float(iw), 1e-5, _p(loc), _p(wbp))
        if touch:
            c_hw = (ctypes.c_int32 * (2 * L))(*[v for hw in sizes for v in hw])
            _lib.check(lib.sbev_sample_and_project_touch(*a, c_hw, _p(need), _stream()), 'sbev_sample_and_project_touch')
        else:
            _lib.check(lib.sbev_sample_and_project(*a, _stream()), 'sbev_sample_and_project')
        outs.append((loc, wbp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    loc = outs[1][0].cpu()
    tiles = [(s + 63) // 64 for s in hw_px]
    base = np.concatenate([[0], np.cumsum([B * T * N * t for t in tiles])])
    got = need.cpu().numpy().astype(bool)
    for b in range(B):
        ref = RF.touched_units(loc.view(B, T * G, Q, P, 3)[b], sizes, T, Q, P)          # per level [T*N, tiles, G]
        for l in range(L):
            gl = got[base[l]:base[l + 1]].reshape(B, T * N, tiles[l], 4)[b]
            assert np.array_equal(gl, ref[l]), 'sample %d level %d: %d marked, %d expected' % (b, l, gl.sum(), ref[l].sum())
    assert 0.05 < got.mean() < 0.95


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def build(T, L, seed, num_layers=3, graph=True, P=4):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=P, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=P, num_layers=num_layers, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    m.decoder.static_graph = graph
    return m


def _runtime_of(model, bbox, feat, feats, metas):
    model(bbox, feat, list(feats), None, metas)      # binds the runtime (eager: graphs off or first sighting)
    return model.decoder._runtime


def _poisoned_like(feats):
    pyr = TR.FeaturePyramid.empty_like_nchw(feats)
    for lv in pyr.levels:
        lv.fill_(float('nan'))
    return pyr


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('chains,pyr_name', [(True, 'r50_704x256'), (False, 'r50_704x256'), (True, 'eva02_1600x640')])
def test_eager_lazy_step_equals_dense_relayout_bit_for_bit(chains, pyr_name, dtype):
    """row chains (the scans of layers 1 .. 2 ride in the generator GEMM's prologue) and op by op (every move a launch of its own); 4 levels and
    5 levels with an odd plane size (10 x 25: scalar reads)"""
    B, Q, T = (2, 100, 2) if pyr_name == 'r50_704x256' else (1, 64, 2)
    ih, iw, sizes = S.PYRAMIDS[pyr_name]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=3, dtype=dtype)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=5)]
    metas = S.make_img_metas(B, T, ih, iw)
    m = build(T, len(sizes), 21, graph=False)
    rt = _runtime_of(m, bbox, feat, feats, metas)
    ctx = TR.DecoderContext(metas, B, DEV)

    lib = _lib.load()

    def run():
        want = rt.forward(bbox, feat, TR.FeaturePyramid(feats), ctx)
        # the scans of layers 1 .. as launches of their own (the row-chain path otherwise carries them in the generator GEMM's prologue)
        prev_scan = lib.sbev_decoder_lazy_scan_launch(1)
        try:
            cls_s, box_s, _ = rt.forward_lazy(bbox, feat, feats, ctx, buffers=_poisoned_like(feats))
        finally:
            lib.sbev_decoder_lazy_scan_launch(prev_scan)
        assert torch.equal(cls_s, want[0]) and torch.equal(box_s, want[1]), 'scan as its own launch'
        pyr = _poisoned_like(feats)
        cls, box, pyr = rt.forward_lazy(bbox, feat, feats, ctx, buffers=pyr)
        assert torch.isfinite(cls).all() and torch.isfinite(box).all(), 'a tap read a unit that was not moved (NaN poison)'
        assert torch.equal(cls, want[0]) and torch.equal(box, want[1])
        moved = sum(int((~torch.isnan(lv.float())).sum()) for lv in pyr.levels) / sum(lv.numel() for lv in pyr.levels)
        # a second step on the SAME buffers with other queries: stale units of the first step must not be taken for moved ones
        bbox2, feat2 = [t.to(DEV) for t in S.make_queries(B, Q, seed=6)]
        feats2 = [f * 0.5 for f in feats]
        want2 = rt.forward(bbox2, feat2, TR.FeaturePyramid(feats2), ctx)
        cls2, box2, _ = rt.forward_lazy(bbox2, feat2, feats2, ctx, buffers=pyr)
        assert torch.equal(cls2, want2[0]) and torch.equal(box2, want2[1])
        return moved
    if chains:
        moved = run()
    else:
        with op_by_op_runtime():
            moved = run()
    print('lazy relayout moved %.1f %% of the pyramid (B=%d, Q=%d, T=%d, %s)' % (100 * moved, B, Q, T, dtype))
    assert 0.02 < moved < 0.9


def test_replayed_lazy_step_follows_fresh_tensors_and_equals_the_dense_step():
    """runtime.StepGraphs: NCHW lists of new tensors every call replay ONE graph whose relayout is on demand; equal to the dense-relayout
    graph and to the eager step, for changing queries / features / camera constants; SBEV_NO_SPARSE_RELAYOUT / lazy_relayout(False) keys
    another graph."""
    B, Q, T = 1, 144, 2
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    metas = S.make_img_metas(B, T, ih, iw)
    g, e = build(T, len(sizes), 31), build(T, len(sizes), 31, graph=False)
    prev = runtime.lazy_relayout(True)
    try:
        for step in range(5):
            feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=10 + step)]
            bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=20 + step)]
            m2 = copy.deepcopy(metas)
            for m in m2:
                m['lidar2img'] = [x * (1.0 + 0.01 * step) for x in m['lidar2img']]
            got = g(bbox, feat, list(feats), None, m2)
            runtime.lazy_relayout(False)            # (the eager step takes the on-demand path too: the reference here is the DENSE eager step)
            want = e(bbox, feat, list(feats), None, m2)
            runtime.lazy_relayout(True)
            eager_lazy = e(bbox, feat, list(feats), None, m2)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), 'step %d' % step
            assert torch.equal(eager_lazy[0], want[0]) and torch.equal(eager_lazy[1], want[1]), 'eager lazy step %d' % step
        sg = g.decoder._runtime.step_graphs
        assert sg.captures == 1 and sg.replays == 4
        nodes_lazy = next(v for v in sg.entries.values() if isinstance(v, dict))['graph'].num_nodes
        runtime.lazy_relayout(False)
        for step in range(3):
            got = g(bbox, feat, list(feats), None, m2)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert sg.captures == 2
        nodes_dense = [v for v in sg.entries.values() if isinstance(v, dict)][-1]['graph'].num_nodes
        print('graph nodes: lazy %d, dense %d' % (nodes_lazy, nodes_dense))
        # one dense relayout launch less, one move launch more for layer 0; the scans of layers 1 .. 2 ride in the generator GEMM's prologue
        # (fp16 GEMM modes, <= 1024 rows: csrc/decoder.hip `scan_in_gen`)
        assert nodes_lazy == nodes_dense - 1 + 1
    finally:
        runtime.lazy_relayout(prev)


def test_every_unit_touched_equals_dense():
    """queries spread so that (nearly) every unit is read is not constructible from boxes alone -- the adversarial case is exercised at the
    kernel level (all flags set, test above) and here through the decoder with a stale all-ones `need`: a first step on a fresh
    workspace whose flag words are garbage must still be exact (stale marks only move more)."""
    B, Q, T = 1, 100, 1
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=8)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=9)]
    metas = S.make_img_metas(B, T, ih, iw)
    m = build(T, len(sizes), 41, graph=False)
    rt = _runtime_of(m, bbox, feat, feats, metas)
    ctx = TR.DecoderContext(metas, B, DEV)
    want = rt.forward(bbox, feat, TR.FeaturePyramid(feats), ctx)
    rt._ws.fill_(0xff)                              # every flag byte set (and everything else of the workspace is scratch)
    pyr = _poisoned_like(feats)
    cls, box, pyr = rt.forward_lazy(bbox, feat, feats, ctx, buffers=pyr)
    assert torch.equal(cls, want[0]) and torch.equal(box, want[1])
    assert all(bool(torch.isfinite(lv).all()) for lv in pyr.levels), 'all-ones need: the whole pyramid is moved'
    for lv, f in zip(pyr.levels, feats):
        assert torch.equal(lv, f.permute(0, 1, 3, 4, 2).reshape(lv.shape))


@pytest.mark.parametrize('config', ['c2', 'c3', 'c4'])
def test_full_shape_lazy_equals_dense(config):
    """the BASELINE configs at full size, 6 layers, default GEMM mode: lazy == dense bit for bit; prints the moved fraction (NaN poison)"""
    pyr_name, Q, T, B = {'c2': ('r50_704x256', 900, 8, 1), 'c3': ('r50_704x256', 400, 8, 8), 'c4': ('r101_1408x512', 900, 8, 4)}[config]
    ih, iw, sizes = S.PYRAMIDS[pyr_name]
    feats = S.make_features(B, T, sizes, seed=0, device=DEV)
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=0)]
    metas = S.make_img_metas(B, T, ih, iw)
    torch.manual_seed(0)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=6, num_levels=len(sizes), num_classes=10, code_size=10, pc_range=S.PC_RANGE)
    m.init_weights()
    S.randomize_zero_init(m, std=0.02, seed=0)
    m = m.to(DEV).eval()
    m.decoder.static_graph = False
    rt = _runtime_of(m, bbox, feat, feats, metas)
    ctx = TR.DecoderContext(metas, B, DEV)
    want = rt.forward(bbox, feat, TR.FeaturePyramid(feats), ctx)
    pyr = _poisoned_like(feats)
    cls, box, pyr = rt.forward_lazy(bbox, feat, feats, ctx, buffers=pyr)
    assert torch.equal(cls, want[0]) and torch.equal(box, want[1])
    moved = sum(int((~torch.isnan(lv)).sum()) for lv in pyr.levels) / sum(lv.numel() for lv in pyr.levels)
    print('%s: lazy relayout moved %.1f %% of the pyramid over 6 layers' % (config, 100 * moved))
    assert 0.2 < moved < 0.7
