"""Python operator layer over the C ABI -- mirrors the reference's operator interface
(models/csrc/wrapper.py:87-93 ``msmv_sampling``; models/sparsebev_sampling.py ``make_sample_points`` /
``sampling_4d``) with the same names, argument meaning and error behaviour, but every byte of work is
done by hand-written gfx950 kernels in libsbev_hip.so.  Tensors must live on a HIP device; there is no
CPU path (calling these with CPU tensors raises).
"""
import ctypes

import torch

from . import _lib

N_VIEWS = 6           # models/sparsebev_sampling.py:45
OUT_REF, OUT_MIX = 0, 1
_F32, _BF16, _F16 = 0, 1, 2


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sparsebev_amd ops need device tensors (no CPU fallback); got a %s tensor' % t.device)


def _check_sampling_args(feats, sampling_locations, scale_weights, Bp, what):
    """The argument checks of msmv_sampling.cpp:106-125 shared by every sampler entry point: the kernel reinterprets
    loc / weights as fp32 rows of 3 / L floats, so anything else must be refused here."""
    if not 1 <= len(feats) <= 5:
        raise RuntimeError('%s supports 1..5 feature levels, got %d' % (what, len(feats)))
    if sampling_locations.dtype != torch.float32 or scale_weights.dtype != torch.float32:
        raise RuntimeError('sampling_loc / attn_weight must be float32')
    if sampling_locations.dim() != 4 or sampling_locations.shape[-1] != 3 or sampling_locations.shape[0] != Bp:
        raise RuntimeError('sampling_loc must be [B, Q, P, 3]')
    _, Q, P, _ = sampling_locations.shape
    if tuple(scale_weights.shape) != (Bp, Q, P, len(feats)):
        raise RuntimeError('attn_weight must be [B, Q, P, %d]' % len(feats))
    if P > 32:
        raise RuntimeError('num_point exceed limits')
    return Q, P


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError('sparsebev_amd: this entry point is forward-only; the differentiable forms are '
                                  'ops.msmv_sampling (reference layout) and the decoder module in train() mode')


def _msmv_launch(feats, hw, feat_dtype, Bp, N, C, Q, P, gdiv, stride_bo, stride_g, stride_v, stride_px,
                 loc, weights, out, out_layout, T, G):
    L = len(feats)
    lib = _lib.load()
    c_feats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    c_hw = (ctypes.c_int32 * (2 * L))(*[v for pair in hw for v in pair])
    c_sbo = (ctypes.c_int64 * L)(*stride_bo)
    c_sv = (ctypes.c_int64 * L)(*stride_v)
    st = lib.sbev_msmv_fwd(c_feats, c_hw, L, feat_dtype, Bp, N, C, Q, P, gdiv, c_sbo, stride_g, c_sv, stride_px,
                           _ptr(loc), _ptr(weights), _ptr(out), out_layout, T, G, _stream())
    _lib.check(st, 'sbev_msmv_fwd')


def _feat_dtype(feats):
    dt = feats[0].dtype
    if any(f.dtype != dt for f in feats):
        raise RuntimeError('all feature levels must share one dtype')
    if dt == torch.float32:
        return _F32
    if dt == torch.bfloat16:
        return _BF16
    if dt == torch.float16:
        return _F16
    raise RuntimeError('feature dtype must be float32, bfloat16 or float16, got %s' % dt)


def _msmv_forward(feats, sampling_locations, scale_weights, out_layout, T, G):
    Bp, N, _, _, C = feats[0].shape
    _, Q, P, _ = sampling_locations.shape
    hw = [(f.shape[2], f.shape[3]) for f in feats]
    sbo = [N * h * w * C for h, w in hw]
    sv = [h * w * C for h, w in hw]
    if out_layout == OUT_REF:
        out = torch.empty(Bp, Q, C, P, device=feats[0].device, dtype=torch.float32)
    else:
        out = torch.empty(Bp // (T * G), Q, G, T * P, C, device=feats[0].device, dtype=torch.float32)
    _msmv_launch(feats, hw, _feat_dtype(feats), Bp, N, C, Q, P, 1, sbo, 0, sv, C,
                 sampling_locations, scale_weights, out, out_layout, T, G)
    return out


class MSMVSampling(torch.autograd.Function):
    """Autograd wrapper, the counterpart of MSMVSamplingC2345 / C23456 (models/csrc/wrapper.py:41-84): forward and
    backward are both HIP kernels (sbev_msmv_fwd / sbev_msmv_bwd); fp32 features, reference layout."""

    @staticmethod
    def forward(ctx, sampling_locations, scale_weights, *feats):
        ctx.save_for_backward(sampling_locations, scale_weights, *feats)
        return _msmv_forward(list(feats), sampling_locations, scale_weights, OUT_REF, 1, 1)

    @staticmethod
    def backward(ctx, grad_output):
        loc, weights, *feats = ctx.saved_tensors
        if feats[0].dtype != torch.float32:
            raise NotImplementedError('msmv_sampling backward needs fp32 features')
        grad_output = grad_output.contiguous().float()
        L = len(feats)
        Bp, N, _, _, C = feats[0].shape
        _, Q, P, _ = loc.shape
        gfeats = [torch.zeros_like(f) for f in feats]              # the op accumulates with atomics
        gloc = torch.empty_like(loc)
        gw = torch.empty_like(weights)
        hw = [(f.shape[2], f.shape[3]) for f in feats]
        lib = _lib.load()
        c_feats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
        c_gfeats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in gfeats])
        c_hw = (ctypes.c_int32 * (2 * L))(*[v for pair in hw for v in pair])
        c_sbo = (ctypes.c_int64 * L)(*[N * h * w * C for h, w in hw])
        c_sv = (ctypes.c_int64 * L)(*[h * w * C for h, w in hw])
        st = lib.sbev_msmv_bwd(c_feats, c_gfeats, c_hw, L, Bp, N, C, Q, P, 1, c_sbo, 0, c_sv, C,
                               _ptr(loc), _ptr(weights), _ptr(grad_output), _ptr(gloc), _ptr(gw), _stream())
        _lib.check(st, 'sbev_msmv_bwd')
        return (gloc, gw, *gfeats)


def msmv_sampling(mlvl_feats, sampling_locations, scale_weights, out_layout=OUT_REF, T=1, G=1):
    """Drop-in for the reference operator ``msmv_sampling`` (models/csrc/wrapper.py:87-93).

    mlvl_feats: list (1..5 levels) of contiguous channel-last ``[B', N, H_l, W_l, C]`` device tensors
    (fp32, or bf16 storage with fp32 accumulation); sampling_locations ``[B', Q, P, 3]``;
    scale_weights ``[B', Q, P, L]``.  Returns ``[B', Q, C, P]`` fp32 (or, with ``out_layout=OUT_MIX``,
    ``[B'/(T*G), Q, G, T*P, C]``).  Same preconditions as msmv_sampling.cpp:106-125 (contiguity, device,
    P <= 32); violations raise RuntimeError.  Differentiable (reference layout, fp32 features) like the
    reference's autograd Functions."""
    feats = list(mlvl_feats)
    _need_device(sampling_locations, scale_weights, *feats)
    for f in feats:
        if not f.is_contiguous():
            raise RuntimeError('value tensor has to be contiguous')
        if f.dim() != 5:
            raise RuntimeError('value tensor must be [B, N, H, W, C]')
    if not sampling_locations.is_contiguous():
        raise RuntimeError('sampling_loc tensor has to be contiguous')
    if not scale_weights.is_contiguous():
        raise RuntimeError('attn_weight tensor has to be contiguous')
    Bp = feats[0].shape[0] if feats else 0
    _check_sampling_args(feats, sampling_locations, scale_weights, Bp, 'msmv_sampling')
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (sampling_locations, scale_weights, *feats))
    if needs_grad:
        if out_layout != OUT_REF:
            raise NotImplementedError('autograd is wired for the reference output layout only')
        return MSMVSampling.apply(sampling_locations, scale_weights, *feats)
    return _msmv_forward(feats, sampling_locations, scale_weights, out_layout, T, G)


def msmv_sampling_nhwc(feats_nhwc, B, T, G, sampling_locations, scale_weights, out_layout=OUT_MIX):
    """Zero-copy variant: feats_nhwc is a list of ``[B*T*N, H_l, W_l, G*C]``
    channels-last pyramids straight from an NHWC neck; group g of sample batch b' = (b*T+t)*G+g is the
    channel slice [g*C, (g+1)*C) -- the reference's regroup copy (models/sparsebev_transformer.py:73-85,
    2x the feature bytes per call) never happens."""
    feats = list(feats_nhwc)
    _need_device(sampling_locations, scale_weights, *feats)
    _no_grad_only(sampling_locations, scale_weights, *feats)
    N = N_VIEWS
    Bp = B * T * G
    Q, P = _check_sampling_args(feats, sampling_locations, scale_weights, Bp, 'msmv_sampling_nhwc')
    GC = feats[0].shape[-1]
    if GC % G != 0 or (GC // G) % 4 != 0:
        raise RuntimeError('nhwc feature channels %d must split into G=%d groups of a multiple of 4 channels' % (GC, G))
    C = GC // G
    for f in feats:
        if not f.is_contiguous() or f.dim() != 4 or f.shape[0] != B * T * N or f.shape[-1] != GC:
            raise RuntimeError('nhwc feature level must be contiguous [B*T*6, H, W, G*C]')
    hw = [(f.shape[1], f.shape[2]) for f in feats]
    sbo = [N * h * w * GC for h, w in hw]
    sv = [h * w * GC for h, w in hw]
    if out_layout == OUT_REF:
        out = torch.empty(Bp, Q, C, P, device=feats[0].device, dtype=torch.float32)
    else:
        out = torch.empty(B, Q, G, T * P, C, device=feats[0].device, dtype=torch.float32)
    _msmv_launch(feats, hw, _feat_dtype(feats), Bp, N, C, Q, P, G, sbo, C, sv, GC,
                 sampling_locations.contiguous(), scale_weights.contiguous(), out, out_layout, T, G)
    return out


def msmv_sampling_nhwc_backward(feats_nhwc, B, T, G, sampling_locations, scale_weights, grad_out, grad_feats=None,
                                grad_layout=OUT_MIX):
    """Backward of msmv_sampling_nhwc (sbev_msmv_bwd_ex): grad_out in the forward's output layout ->
    (grad_loc [B',Q,P,3], grad_weights [B',Q,P,L]); grad wrt the features is ACCUMULATED (atomics) into ``grad_feats``
    (list of fp32 buffers shaped like feats_nhwc) or skipped entirely when it is None (frozen features)."""
    feats = list(feats_nhwc)
    _need_device(sampling_locations, scale_weights, grad_out, *feats)
    if _feat_dtype(feats) != _F32:
        raise NotImplementedError('the sampling backward needs fp32 feature maps (bf16 storage is an inference format)')
    N = N_VIEWS
    Bp = B * T * G
    Q, P = _check_sampling_args(feats, sampling_locations, scale_weights, Bp, 'msmv_sampling_nhwc_backward')
    GC = feats[0].shape[-1]
    C = GC // G
    L = len(feats)
    hw = [(f.shape[1], f.shape[2]) for f in feats]
    gloc = torch.empty_like(sampling_locations)
    gw = torch.empty_like(scale_weights)
    lib = _lib.load()
    c_feats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    c_gfeats = (ctypes.c_void_p * L)(*[g.data_ptr() for g in grad_feats]) if grad_feats is not None else None
    c_hw = (ctypes.c_int32 * (2 * L))(*[v for pair in hw for v in pair])
    c_sbo = (ctypes.c_int64 * L)(*[N * h * w * GC for h, w in hw])
    c_sv = (ctypes.c_int64 * L)(*[h * w * GC for h, w in hw])
    st = lib.sbev_msmv_bwd_ex(c_feats, c_gfeats, c_hw, L, Bp, N, C, Q, P, G, c_sbo, C, c_sv, GC,
                              _ptr(sampling_locations.contiguous()), _ptr(scale_weights.contiguous()), _ptr(grad_out.contiguous()),
                              grad_layout, T, G, _ptr(gloc), _ptr(gw), _stream())
    _lib.check(st, 'sbev_msmv_bwd_ex')
    return gloc, gw


def msmv_sampling_ring(levels, B, T, G, frame_slots, n_slots, sampling_locations, scale_weights, out_layout=OUT_MIX):
    """Sampler over the online frame ring (cache.FrameFeatureCache): levels[l] = [B*n_slots*6, H, W, G*C]; logical frame
    t of a sample is read from physical slot frame_slots[t] (sbev_msmv_fwd_ring)."""
    feats = list(levels)
    _need_device(sampling_locations, scale_weights, *feats)
    _no_grad_only(sampling_locations, scale_weights, *feats)
    N = N_VIEWS
    Bp = B * T * G
    Q, P = _check_sampling_args(feats, sampling_locations, scale_weights, Bp, 'msmv_sampling_ring')
    GC = feats[0].shape[-1]
    if GC % G != 0 or (GC // G) % 4 != 0:
        raise RuntimeError('ring feature channels %d must split into G=%d groups of a multiple of 4 channels' % (GC, G))
    C = GC // G
    L = len(feats)
    for f in feats:
        if not f.is_contiguous() or f.dim() != 4 or f.shape[0] != B * n_slots * N or f.shape[-1] != GC:
            raise RuntimeError('ring feature level must be contiguous [B*n_slots*6, H, W, G*C]')
    if len(frame_slots) != T:
        raise RuntimeError('frame_slots must name one slot per frame (T=%d)' % T)
    hw = [(f.shape[1], f.shape[2]) for f in feats]
    sslot = [N * h * w * GC for h, w in hw]
    sv = [h * w * GC for h, w in hw]
    if out_layout == OUT_REF:
        out = torch.empty(Bp, Q, C, P, device=feats[0].device, dtype=torch.float32)
    else:
        out = torch.empty(B, Q, G, T * P, C, device=feats[0].device, dtype=torch.float32)
    lib = _lib.load()
    c_feats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    c_hw = (ctypes.c_int32 * (2 * L))(*[v for pair in hw for v in pair])
    c_ss = (ctypes.c_int64 * L)(*sslot)
    c_sv = (ctypes.c_int64 * L)(*sv)
    c_slots = (ctypes.c_int32 * T)(*[int(s) for s in frame_slots])
    st = lib.sbev_msmv_fwd_ring(c_feats, c_hw, L, _feat_dtype(feats), Bp, N, C, Q, P, G, c_ss, C, c_sv, GC,
                                _ptr(sampling_locations.contiguous()), _ptr(scale_weights.contiguous()), _ptr(out),
                                out_layout, T, G, c_slots, n_slots, _stream())
    _lib.check(st, 'sbev_msmv_fwd_ring')
    return out


def sample_mix_supported(L, C, P, T, G):
    return bool(_lib.load().sbev_sample_mix_supported(L, C, P, T, G, G))


def query_order(query_bbox, pc_range):
    """Launch order of the fused gather + mixing items (sbev_query_order): int32 [B*Q], sample b's rows b*Q + q sorted by the
    direction of the box centre around the ego origin.  query_bbox [B, Q, >= 2] fp32 (columns 0, 1 = normalised centre)."""
    _need_device(query_bbox)
    if query_bbox.dim() != 3 or query_bbox.dtype != torch.float32 or query_bbox.shape[-1] < 2:
        raise RuntimeError('query_order: query_bbox must be fp32 [B, Q, >= 2]')
    qb = query_bbox.contiguous()
    B, Q = qb.shape[:2]
    order = torch.empty(B * Q, device=qb.device, dtype=torch.int32)
    pcr = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    _lib.check(_lib.load().sbev_query_order(_ptr(qb), qb.shape[-1], pcr, B, Q, _ptr(order), _stream()), 'sbev_query_order')
    return order


def sample_mix(levels, B, T, G, sampling_locations, scale_weights, params, out_points, frame_slots=None, n_slots=0, order=None):
    """Gather + adaptive mixing in one launch (sbev_sample_mix_f32): levels as for msmv_sampling_nhwc (or the ring's
    buffers with frame_slots / n_slots), params [B,Q,G*(C*C + out_points*T*P)] -> mixed [B,Q,G*out_points*C].
    Bit-identical to msmv_sampling_nhwc(..., OUT_MIX) followed by the mixing kernel.  order (query_order(); any permutation of
    the B*Q rows as int32): the workgroups' launch order -- a placement hint, the result does not depend on it."""
    feats = list(levels)
    _need_device(sampling_locations, scale_weights, params, *feats)
    _no_grad_only(sampling_locations, scale_weights, params, *feats)
    N = N_VIEWS
    Q, P = _check_sampling_args(feats, sampling_locations, scale_weights, B * T * G, 'sample_mix')
    GC = feats[0].shape[-1]
    C = GC // G
    L = len(feats)
    if not sample_mix_supported(L, C, P, T, G):
        raise RuntimeError('sample_mix: shape not covered by the fused kernel (L=%d C=%d P=%d T=%d)' % (L, C, P, T))
    hw = [(f.shape[1], f.shape[2]) for f in feats]
    params = params.contiguous()
    if params.dtype != torch.float32 or params.numel() != B * Q * G * (C * C + out_points * T * P):
        raise RuntimeError('sample_mix: params must be fp32 [B, Q, G*(C*C + out_points*T*P)]')
    y = torch.empty(B, Q, G * out_points * C, device=params.device, dtype=torch.float32)
    c_feats = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    c_hw = (ctypes.c_int32 * (2 * L))(*[v for pair in hw for v in pair])
    c_sbo = (ctypes.c_int64 * L)(*[N * h * w * GC for h, w in hw])
    c_sv = (ctypes.c_int64 * L)(*[h * w * GC for h, w in hw])
    c_slots = (ctypes.c_int32 * T)(*[int(v) for v in frame_slots]) if frame_slots is not None else None
    if order is not None:
        _need_device(order)
        if order.dtype != torch.int32 or order.numel() != B * Q or not order.is_contiguous():
            raise RuntimeError('sample_mix: order must be a contiguous int32 permutation of the B*Q rows')
    st = _lib.load().sbev_sample_mix_f32_ordered(c_feats, c_hw, L, _feat_dtype(feats), B, N, Q, T, G, P, C, c_sbo, C, c_sv, GC,
                                                 _ptr(sampling_locations.contiguous()), _ptr(scale_weights.contiguous()), c_slots, n_slots,
                                                 _ptr(params), _ptr(y), out_points, 1e-5, _ptr(order) if order is not None else None, _stream())
    _lib.check(st, 'sbev_sample_mix_f32')
    return y


def project_select(sample_points, lidar2img, image_h, image_w, G, P, eps=1e-5, dump=False):
    """Front half of sampling_4d (models/sparsebev_sampling.py:49-114) on device, bit-exact camera-hit
    mask.  sample_points ``[B,Q,T,G*P,3]``, lidar2img ``[B,T*6,4,4]`` -> loc ``[B*T*G,Q,P,3]`` and, with
    dump=True, the DUMP-tap tensors (uvh ``[B,T,6,Q,GP,3]``, valid uint8 ``[B,T,6,Q,GP]``, i_view int32
    ``[B,T,Q,GP]``)."""
    _need_device(sample_points, lidar2img)
    sample_points = sample_points.contiguous().float()
    lidar2img = lidar2img.contiguous().float()
    B, Q, T, GP, _ = sample_points.shape
    assert GP == G * P and lidar2img.shape[1] == T * N_VIEWS
    dev = sample_points.device
    loc = torch.empty(B * T * G, Q, P, 3, device=dev, dtype=torch.float32)
    uvh = valid = iview = None
    if dump:
        uvh = torch.empty(B, T, N_VIEWS, Q, GP, 3, device=dev, dtype=torch.float32)
        valid = torch.empty(B, T, N_VIEWS, Q, GP, device=dev, dtype=torch.uint8)
        iview = torch.empty(B, T, Q, GP, device=dev, dtype=torch.int32)
    st = _lib.load().sbev_project_select(_ptr(sample_points), _ptr(lidar2img), B, Q, T, N_VIEWS, G, P,
                                         float(image_h), float(image_w), float(eps),
                                         _ptr(loc), _ptr(uvh), _ptr(valid), _ptr(iview), _stream())
    _lib.check(st, 'sbev_project_select')
    return (loc, uvh, valid, iview) if dump else loc


def sampling_front(query_bbox, offset, scale_logits, time_diff, pc_range, T, G, P, L,
                   want_points=True, want_weights=True):
    """make_sample_points + velocity warp + level softmax + weight reorder (see sbev_sampling_front in
    include/sbev_hip.h).  offset [B,Q,>=G*P*3] and scale_logits [B,Q,>=G*P*L] may be column slices of one packed
    GEMM output (only their last-dim stride must be 1).
    Returns (sample_points [B,Q,T,G*P,3] | None, weights_bp [B*G*T,Q,P,L] | None)."""
    _need_device(query_bbox, offset, scale_logits, time_diff)
    B, Q = query_bbox.shape[:2]
    dev = query_bbox.device
    query_bbox = query_bbox.contiguous().float()

    def rows(t):
        if t is None:
            return None, 0
        if t.stride(-1) != 1 or t.stride(0) != Q * t.stride(1):
            t = t.contiguous()
        return t, t.stride(1)

    offset, ld_off = rows(offset if want_points else None)
    scale_logits, ld_lg = rows(scale_logits if want_weights else None)
    td = time_diff.contiguous().float() if want_points else None
    pts = torch.empty(B, Q, T, G * P, 3, device=dev, dtype=torch.float32) if want_points else None
    wbp = torch.empty(B * G * T, Q, P, L, device=dev, dtype=torch.float32) if want_weights else None
    pc = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    st = _lib.load().sbev_sampling_front(_ptr(query_bbox), _ptr(offset), ld_off, _ptr(scale_logits), ld_lg, _ptr(td), pc,
                                         B, Q, T, G, P, L, _ptr(pts), _ptr(wbp), _stream())
    _lib.check(st, 'sbev_sampling_front')
    return pts, wbp
