"""GPU parity of the detection-head pre / post-processing kernels (csrc/head.hip, sparsebev_amd/head.py) against the
golden vectors recorded from the reference's NMSFreeCoder (G9) and against the CPU oracle."""
import copy

import pytest
import torch

from conftest import load_golden
from oracle import sparsebev_oracle as O
from sparsebev_amd import synthetic as S
from sparsebev_amd import head as H

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
POST = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]


@pytest.mark.parametrize('tag', ['c2', 'small', 'few'])
def test_g9_nms_free_coder_vs_reference_recording(tag):
    g = load_golden('g9_nms_free_' + tag)
    B, Q, NC, max_num = [int(v) for v in g['cfg']]
    thr = float(g['thr'])
    thr = None if thr < 0 else thr
    coder = H.NMSFreeCoder(S.PC_RANGE, post_center_range=[float(v) for v in g['post']], max_num=max_num, score_threshold=thr, num_classes=NC)
    dec = coder.decode({'all_cls_scores': g['cls'].to(DEV), 'all_bbox_preds': g['box'].to(DEV)})
    assert len(dec) == B
    for i, d in enumerate(dec):
        assert d['labels'].dtype == torch.long
        assert torch.equal(d['labels'].cpu(), g['labels%d' % i]), i           # same boxes, same order: integer work is exact
        assert (d['scores'].cpu() - g['scores%d' % i]).abs().max() < 1e-6      # sigmoid: device expf vs torch CPU
        ref = g['bboxes%d' % i]
        got = d['bboxes'].cpu()
        assert got.shape == ref.shape
        assert torch.equal(got[:, [0, 1, 2, 7, 8]], ref[:, [0, 1, 2, 7, 8]])   # copied columns are bit-identical
        assert ((got[:, 3:6] - ref[:, 3:6]).abs() / ref[:, 3:6]).max() < 1e-6   # exp
        assert (got[:, 6] - ref[:, 6]).abs().max() < 1e-6                       # atan2
    # decode_single: the per-sample entry point of the reference class
    one = coder.decode_single(g['cls'][-1, 0].to(DEV), g['box'][-1, 0].to(DEV))
    assert torch.equal(one['labels'].cpu(), g['labels0'])


def test_decode_ties_padding_and_errors():
    """Equal scores resolve by flat index (a total order; torch.topk's is unspecified), NaN logits rank first like
    torch.topk, rows past the count are zero, and the reference's error cases raise."""
    Q, NC = 8, 4
    cls = torch.full((1, Q, NC), -1.0)
    cls[0, 5, 2] = 3.0
    cls[0, 1, 1] = 3.0
    cls[0, 6, 0] = 2.0
    box = torch.zeros(1, Q, 10)
    box[0, :, 0] = torch.arange(Q).float()             # cx = query index
    boxes, scores, labels, count = H.nms_free_decode(cls.to(DEV), box.to(DEV), NC, 5, 0.5, POST)
    assert int(count[0]) == 3
    assert boxes[0, :3, 0].tolist() == [1.0, 5.0, 6.0] and labels[0, :3].tolist() == [1, 2, 0]
    assert float(boxes[0, 3:].abs().max()) == 0.0 and float(scores[0, 3:].abs().max()) == 0.0
    # no threshold (None / 0.0 are both "off", nms_free_coder.py:72): all max_num entries inside the range survive
    _, _, _, count = H.nms_free_decode(cls.to(DEV), box.to(DEV), NC, 5, None, POST)
    assert int(count[0]) == 5
    cls[0, 7, 3] = float('nan')
    boxes, scores, labels, count = H.nms_free_decode(cls.to(DEV), box.to(DEV), NC, 5, None, POST)
    assert boxes[0, 0, 0].item() == 7.0 and labels[0, 0].item() == 3
    with pytest.raises(RuntimeError, match='out of range'):           # torch.topk: selected index k out of range
        H.nms_free_decode(cls.to(DEV), box.to(DEV), NC, Q * NC + 1, None, POST)
    with pytest.raises(NotImplementedError):
        H.nms_free_decode(cls.to(DEV), box.to(DEV), NC, 5, None, None)
    with pytest.raises(RuntimeError):
        H.nms_free_decode(cls, box, NC, 5, None, POST)                  # CPU tensors: no fallback


@pytest.mark.parametrize('Q,NC', [(900, 10), (1600, 10), (204, 10), (100, 1)])
def test_decode_all_sort_sizes_vs_oracle(Q, NC):
    g = torch.Generator().manual_seed(Q + NC)
    B, max_num = 3, min(300, Q * NC)
    cls = torch.randn(B, Q, NC, generator=g) * 2 - 1
    box = torch.randn(B, Q, 10, generator=g)
    box[..., 0:2] *= 45
    ref = O.get_bboxes(O.nms_free_decode(cls[None], box[None], NC, max_num, 0.1, POST))
    coder = H.NMSFreeCoder(S.PC_RANGE, post_center_range=POST, max_num=max_num, score_threshold=0.1, num_classes=NC)
    dec = coder._decode({'all_cls_scores': cls[None].to(DEV), 'all_bbox_preds': box[None].to(DEV)}, True)
    for (rb, rs, rl), d in zip(ref, dec):
        assert torch.equal(d['labels'].cpu(), rl)
        assert (d['scores'].cpu() - rs).abs().max() < 1e-6
        assert (d['bboxes'].cpu() - rb).abs().max() < 2e-5            # bottom-centre z = cz - h/2 with h = exp(.)


def test_head_prepare_and_denorm_bit_exact():
    g = torch.Generator().manual_seed(3)
    init, lab = torch.rand(900, 10, generator=g), torch.randn(11, 255, generator=g)
    qb, qf = H.head_prepare(init.to(DEV), lab.to(DEV), 10, 2)
    rb, rf = O.head_prepare(init, lab, 10, 2)
    assert torch.equal(qb.cpu(), rb) and torch.equal(qf.cpu(), rf)
    box = torch.rand(6, 2, 900, 10, generator=g)
    out = H.head_postprocess(box.to(DEV), S.PC_RANGE)
    assert torch.equal(out.cpu(), O.head_postprocess(box, S.PC_RANGE))       # two roundings, like the reference


def test_head_module_end_to_end_vs_oracle():
    """SparseBEVHead.forward + get_bboxes (queries from the embeddings, decoder, re-format, decode) against the oracle
    chain on the same random-init weights; reference state-dict names load strictly."""
    T, L, Q, B = 2, 4, 36, 2
    ih, iw, sizes = S.PYRAMIDS['tiny']
    head = H.SparseBEVHead(num_classes=10, in_channels=256, num_query=Q, code_size=10,
                           transformer=dict(type='SparseBEVTransformer', embed_dims=256, num_frames=T, num_points=4, num_layers=2,
                                            num_levels=L, num_classes=10, code_size=10, pc_range=S.PC_RANGE),
                           bbox_coder=dict(type='NMSFreeCoder', post_center_range=POST, max_num=30, score_threshold=None,
                                           num_classes=10, pc_range=S.PC_RANGE))
    keys = set(head.state_dict())
    assert {'init_query_bbox.weight', 'label_enc.weight', 'code_weights'} <= keys
    assert len([k for k in keys if k.startswith('transformer.decoder.decoder_layer.')]) == 48 and len(keys) == 51
    params = S.make_params(61, embed_dims=256, num_frames=T, num_points=4, num_levels=L)
    head.transformer.load_state_dict({'decoder.decoder_layer.' + k: v for k, v in params.items()}, strict=True)
    with torch.no_grad():
        head.init_query_bbox.weight[:, 2] = 0.5            # lift the grid to ~ -1 m so that the cameras see it
        head.init_query_bbox.weight[:, 5] = 0.5
    head = head.to(DEV).eval()
    feats = S.make_features(B, T, sizes, seed=62)
    metas = S.make_img_metas(B, T, ih, iw)
    outs = head([f.to(DEV) for f in feats], copy.deepcopy(metas))
    assert outs['all_cls_scores'].shape == (2, B, Q, 10) and outs['all_bbox_preds'].shape == (2, B, Q, 10)
    qb, qf = O.head_prepare(head.init_query_bbox.weight.cpu(), head.label_enc.weight.cpu(), 10, B)
    cls, box, _ = O.decoder(params, qb, qf, feats, metas, S.PC_RANGE, num_layers=2)
    box = O.head_postprocess(box, S.PC_RANGE)
    assert (outs['all_cls_scores'][0].cpu() - cls[0]).abs().max() < 1e-4
    assert (outs['all_bbox_preds'][0].cpu() - box[0]).abs().max() < 1e-3        # metres: 1e-4 normalised x 102.4 m span... < 1e-3
    res = head.get_bboxes(outs, metas)
    assert len(res) == B and all(r[0].shape[1] == 9 and r[0].shape[0] == r[1].shape[0] == r[2].shape[0] for r in res)
    # the decode step alone, from the device outputs, equals the oracle's on the same tensors
    ref = O.get_bboxes(O.nms_free_decode(outs['all_cls_scores'].cpu(), outs['all_bbox_preds'].cpu(), 10, 30, None, POST))
    assert sum(r[0].shape[0] for r in ref) > 0
    for (rb, rs, rl), (bb, ss, ll) in zip(ref, res):
        assert torch.equal(ll.cpu(), rl)
        if rl.numel():
            assert (ss.cpu() - rs).abs().max() < 1e-6 and (bb.cpu() - rb).abs().max() < 2e-5
    with pytest.raises(NotImplementedError):
        head.train()([f.to(DEV) for f in feats], copy.deepcopy(metas))
