"""Python face of the C++ decoder runtime (csrc/decoder.hip): one ctypes call enqueues the whole 6-layer
forward.  Mirrors include/sbev_hip.h's sbev_decoder_config / sbev_decoder_weights field for field."""
import ctypes

import torch

from . import _lib

MAX_LEVELS = 5
GEMM_F32, GEMM_BF16X3, GEMM_BF16X6, GEMM_BF16X3S, GEMM_F16X3, GEMM_F16X4 = 0, 1, 2, 3, 4, 5      # sbev_gemm_mode
# fp16 modes: how many binades a typical (4 sigma, median |gamma|) generator input may sit below the a-priori scale bound before the
# runtime refuses the split (lo stays a normal fp16 number down to 2^-17 of the bound; 2^-12 leaves 5 binades for the values' own spread)
F16_MAX_HEADROOM_LOG2 = 12
GEMM_MODES = {'f32': GEMM_F32, 'bf16x3': GEMM_BF16X3, 'bf16x6': GEMM_BF16X6, 'bf16x3s': GEMM_BF16X3S, 'f16x3': GEMM_F16X3, 'f16x4': GEMM_F16X4}
_f = ctypes.c_void_p


class DecoderConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('B', 'Q', 'T', 'N', 'G', 'P', 'L', 'D', 'H', 'ffn', 'num_classes',
                                              'code_size', 'num_layers', 'out_points', 'attn_in_rows', 'feat_dtype')] + \
               [('hw', (ctypes.c_int32 * 2) * MAX_LEVELS), ('image_h', ctypes.c_float), ('image_w', ctypes.c_float),
                ('eps_homo', ctypes.c_float), ('gemm_mode', ctypes.c_int32), ('n_slots', ctypes.c_int32),
                ('frame_slots', ctypes.c_int32 * 16), ('overlap', ctypes.c_int32),
                ('pc_range', ctypes.c_double * 6)]


_WEIGHT_FIELDS = ['pe0_w', 'pe0_b', 'pe1_g', 'pe1_b', 'pe3_w', 'pe3_b', 'pe4_g', 'pe4_b',
                  'attn_in_w', 'attn_in_b', 'attn_out_w', 'attn_out_b', 'samp_w', 'samp_b',
                  'pg_w', 'pg_b', 'op_w', 'op_b', 'pg_w2', 'op_w2', 'ffn0_w', 'ffn0_b', 'ffn1_w', 'ffn1_b',
                  'norm1_g', 'norm1_b', 'norm2_g', 'norm2_b', 'norm3_g', 'norm3_b',
                  'cls0_w', 'cls0_b', 'cls1_g', 'cls1_b', 'cls3_w', 'cls3_b', 'cls4_g', 'cls4_b', 'cls6_w', 'cls6_b',
                  'reg0_w', 'reg0_b', 'reg2_w', 'reg2_b', 'reg4_w', 'reg4_b', 'chain_pack', 'pg_ws', 'op_wp', 'pg_wdown', 'op_nscale', 'pg_xscale']


class DecoderWeights(ctypes.Structure):
    _fields_ = [(n, _f) for n in _WEIGHT_FIELDS]


class LazyFeats(ctypes.Structure):
    """sbev_lazy_feats: the NCHW sources of the on-demand relayout (a device pointer table + indices, or direct pointers)"""
    _fields_ = [('table', _f), ('index', ctypes.c_int32 * MAX_LEVELS), ('src', _f * MAX_LEVELS)]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class DecoderRuntime:
    """Binds a SparseBEVTransformerDecoder's parameters (by pointer) to the C++ runtime and owns the workspace.
    Re-binds automatically when a parameter is replaced or modified in place (``_version`` / ``data_ptr`` change)."""

    def __init__(self, decoder, gemm_mode=0, overlap=False):
        import os
        if os.environ.get('SBEV_NO_SAMPLE_MIX') == '1':
            fuse_sample_mix(False)
        if os.environ.get('SBEV_NO_ROW_CHAIN') == '1':
            row_chain(False)
        self.decoder = decoder
        self.gemm_mode = gemm_mode
        self.overlap = overlap
        self._sig = None
        self._keep = None          # tensors whose storage the weight struct points into
        self._weights = None
        self._ws = None
        self._ws_key = None
        self._params = None        # cached parameter SLOTS (module._parameters dict, name); re-collected every 64 signature checks
        self._graph_ws = {}        # (device, size, stream) -> [workspace of the captured step graphs of that size replayed on that stream, graphs using it]
        self._sig_calls = 0
        self.mode_eff = gemm_mode
        self.f16_headroom_log2 = 0.0
        self.step_graphs = StepGraphs(self)

    # -- weights -----------------------------------------------------------------------------------------
    def _signature(self):
        """((data_ptr, _version) of every parameter).  Walking the module tree costs ~0.15 ms per call, so the (module._parameters
        dict, name) SLOTS are cached and each call only looks the current Parameter object of every slot up -- a replaced
        Parameter (``load_state_dict(assign=True)``, ``lin.weight = nn.Parameter(...)``, pruning / parametrize) is a different
        object with its own (data_ptr, _version) and re-binds at once; the slot list itself (modules added or removed) is
        re-collected every 64th call."""
        self._sig_calls += 1
        if self._params is None or (self._sig_calls & 63) == 0:
            self._params = [(m._parameters, n) for m in self.decoder.modules() for n, p in m._parameters.items() if p is not None]
        try:
            ps = [d[n] for d, n in self._params]
            return tuple((p.data_ptr(), p._version) for p in ps)
        except (KeyError, AttributeError):          # a slot vanished or became None: re-collect now
            self._params = [(m._parameters, n) for m in self.decoder.modules() for n, p in m._parameters.items() if p is not None]
            return tuple((d[n].data_ptr(), d[n]._version) for d, n in self._params)

    def _ensure_bound(self):
        """(Re-)bind when a parameter was replaced or modified in place; returns the current signature."""
        sig = self._signature()
        if sig != self._sig:
            rebind = self._sig is not None            # (() = invalidated by hand: also a re-bind)
            self._bind()
            self._sig = sig
            if rebind:
                self.step_graphs.clear()    # graphs of the previous weight images can never be hit again: free them now
        return sig

    def _bind(self):
        layer = self.decoder.decoder_layer
        for p in layer.parameters():
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError('sparsebev_amd runtime needs fp32 parameters on the device (call .to("cuda"))')
        pe, sa, smp, mix = layer.position_encoder, layer.self_attn, layer.sampling, layer.mixing
        att = sa.attention.attn
        D, H = layer.embed_dims, sa.num_heads
        pad = (-(3 * D + H)) % 4
        with torch.no_grad():
            attn_in_w = torch.cat([att.in_proj_weight, sa.gen_tau.weight] + ([att.in_proj_weight.new_zeros(pad, D)] if pad else []), 0).contiguous()
            attn_in_b = torch.cat([att.in_proj_bias, sa.gen_tau.bias] + ([att.in_proj_bias.new_zeros(pad)] if pad else []), 0).contiguous()
            samp_w = torch.cat([smp.sampling_offset.weight, smp.scale_weights.weight], 0).contiguous()
            samp_b = torch.cat([smp.sampling_offset.bias, smp.scale_weights.bias], 0).contiguous()
        cb, rb, ffn = layer.cls_branch, layer.reg_branch, layer.ffn.layers
        t = dict(
            pe0_w=pe[0].weight, pe0_b=pe[0].bias, pe1_g=pe[1].weight, pe1_b=pe[1].bias,
            pe3_w=pe[3].weight, pe3_b=pe[3].bias, pe4_g=pe[4].weight, pe4_b=pe[4].bias,
            attn_in_w=attn_in_w, attn_in_b=attn_in_b, attn_out_w=att.out_proj.weight, attn_out_b=att.out_proj.bias,
            samp_w=samp_w, samp_b=samp_b,
            pg_w=mix.parameter_generator.weight, pg_b=mix.parameter_generator.bias,
            op_w=mix.out_proj.weight, op_b=mix.out_proj.bias,
            ffn0_w=ffn[0][0].weight, ffn0_b=ffn[0][0].bias, ffn1_w=ffn[1].weight, ffn1_b=ffn[1].bias,
            norm1_g=layer.norm1.weight, norm1_b=layer.norm1.bias, norm2_g=layer.norm2.weight, norm2_b=layer.norm2.bias,
            norm3_g=layer.norm3.weight, norm3_b=layer.norm3.bias,
            cls0_w=cb[0].weight, cls0_b=cb[0].bias, cls1_g=cb[1].weight, cls1_b=cb[1].bias,
            cls3_w=cb[3].weight, cls3_b=cb[3].bias, cls4_g=cb[4].weight, cls4_b=cb[4].bias,
            cls6_w=cb[6].weight, cls6_b=cb[6].bias,
            reg0_w=rb[0].weight, reg0_b=rb[0].bias, reg2_w=rb[2].weight, reg2_b=rb[2].bias,
            reg4_w=rb[4].weight, reg4_b=rb[4].bias)
        keep = {k: v.detach().contiguous() for k, v in t.items()}
        self.mode_eff = self.gemm_mode      # what the launches run in: an fp16 mode falls back to the exact kernels when norm1 is unfit (below)
        if self.gemm_mode in (GEMM_F16X3, GEMM_F16X4):
            import math
            # The generator's input is norm1's output; its fp16 scale is fixed per bind from |LayerNorm(x) g + b| <= sqrt(D - 1) max|g| +
            # max|b| (no pass over the activations).  ONE power of two serves the whole tensor, so a channel whose |g| sits far below the
            # largest one lives that many binades under the fp16 range's top: hi stays normal down to 2^-29 of the bound, lo down to 2^-17.
            # A checkpoint with an outlier in norm1 (max|g| or max|b| thousands of times the typical |g|) would push the bulk of the
            # channels below that and quietly lose their lo image: refuse the split then and run the exact f32 kernels.
            g_abs = keep['norm1_g'].abs().float()
            gmax, gmed, bmax = float(g_abs.max()), float(g_abs.median()), float(keep['norm1_b'].abs().max())
            bound = math.sqrt(D - 1) * gmax + bmax
            typical = 4.0 * gmed             # a 4-sigma element of a typical channel
            self.f16_headroom_log2 = math.log2(bound / typical) if (typical > 0 and math.isfinite(bound) and bound > 0) else float('inf')
            if not (self.f16_headroom_log2 <= F16_MAX_HEADROOM_LOG2):
                self.mode_eff = GEMM_F32
                if not getattr(DecoderRuntime, '_warned_f16_bound', False):
                    import warnings
                    warnings.warn('sparsebev_amd: norm1 puts typical generator inputs %.1f binades below the a-priori fp16 scale bound '
                                  '(max|gamma| %.3g, median|gamma| %.3g, max|beta| %.3g; limit %d): the two mixing GEMMs run on the exact f32 '
                                  'kernels instead of %s' % (self.f16_headroom_log2, gmax, gmed, bmax, F16_MAX_HEADROOM_LOG2,
                                                              'f16x3' if self.gemm_mode == GEMM_F16X3 else 'f16x4'))
                    DecoderRuntime._warned_f16_bound = True
        if self.gemm_mode == GEMM_BF16X3:      # one-off (hi, lo) bf16 images of the two big weight matrices
            lib = _lib.load()
            for name in ('pg_w', 'op_w'):
                src = keep[name]
                img = torch.empty(src.shape[0], src.shape[1] * 2, device=src.device, dtype=torch.int16)
                _lib.check(lib.sbev_split_bf16x3_weights(_ptr(src), _ptr(img), src.shape[0], src.shape[1],
                                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                           'sbev_split_bf16x3_weights')
                keep[name + '2'] = img
        if self.gemm_mode in (GEMM_BF16X6, GEMM_BF16X3S):      # split-bf16 images for csrc/gemm_bf16s.hip (3 or 2 images)
            from . import dense
            nimg = 3 if self.gemm_mode == GEMM_BF16X6 else 2
            keep['pg_ws'] = dense.pack_bf16s_frags(keep['pg_w'], nimg)
            keep['op_wp'] = dense.pack_bf16s_frags(keep['op_w'], nimg)
        if self.mode_eff in (GEMM_F16X3, GEMM_F16X4):          # scaled fp16 hi + lo images (csrc/gemm_bf16s.hip), one power of two per weight row
            from . import dense
            lib = _lib.load()
            keep['pg_ws'], pg_sc = dense.pack_f16s_frags(keep['pg_w'])
            keep['op_wp'], op_sc = dense.pack_f16s_frags(keep['op_w'])
            keep['pg_wdown'] = pg_sc[1].contiguous()
            c0 = DecoderConfig()
            c0.G, c0.D, c0.out_points = smp.num_groups, D, mix.out_points
            # the generator's input scale: the power of two of the bound computed above, fixed per bind
            e = 0 if not (bound > 0 and math.isfinite(bound)) else max(-100, min(100, math.floor(math.log2(65504.0 / bound) - 1e-9)))
            keep['pg_xscale'] = torch.tensor([2.0 ** e, 2.0 ** -e], device=pg_sc.device, dtype=torch.float32)
            keep['op_nscale'] = torch.empty(D, device=pg_sc.device, dtype=torch.float32)
            _lib.check(lib.sbev_f16s_out_scale(_ptr(op_sc[1].contiguous()), lib.sbev_decoder_mixed_up_log2(ctypes.byref(c0)), _ptr(keep['op_nscale']), D,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'sbev_f16s_out_scale')
        w = DecoderWeights()
        for k in _WEIGHT_FIELDS:
            setattr(w, k, keep[k].data_ptr() if k in keep else None)
        # lane-ordered image of the small Linears' weights for the row-chain kernels (csrc/row_chain.hip), where they cover
        # the layer's shape; re-made with every re-bind
        lib = _lib.load()
        cfg = DecoderConfig()
        cfg.B = cfg.Q = 1
        cfg.T, cfg.N, cfg.G, cfg.P, cfg.L = smp.num_frames, 6, smp.num_groups, smp.num_points, smp.num_levels
        cfg.D, cfg.H, cfg.ffn = D, H, ffn[0][0].weight.shape[0]
        cfg.num_classes, cfg.code_size, cfg.attn_in_rows = layer.num_classes, layer.code_size, attn_in_w.shape[0]
        n_pack = lib.sbev_decoder_chain_pack_floats(ctypes.byref(cfg))
        if n_pack > 0:
            pack = torch.empty(n_pack, device=attn_in_w.device, dtype=torch.float32)
            st = lib.sbev_decoder_chain_pack(ctypes.byref(cfg), ctypes.byref(w), _ptr(pack),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if st == 0:
                keep['chain_pack'] = pack
                w.chain_pack = pack.data_ptr()
            elif not getattr(DecoderRuntime, '_warned_chain', False):
                # e.g. a device without 160 KB of LDS per workgroup: the op-by-op launches compute the same layer
                import warnings
                warnings.warn('sparsebev_amd: row-chain weight image unavailable (%s): running the op-by-op launches'
                              % lib.sbev_last_error().decode('utf-8', 'replace'))
                DecoderRuntime._warned_chain = True
        self._keep, self._weights = keep, w
        self._attn_in_rows = attn_in_w.shape[0]

    # -- forward -----------------------------------------------------------------------------------------
    def _prepare(self, query_bbox, query_feat, pyramid, ctx, attn_mask, own_workspace=False):
        sig = self._ensure_bound()
        dec, layer = self.decoder, self.decoder.decoder_layer
        smp = layer.sampling
        B, Q, D = query_feat.shape
        dev = query_feat.device
        cfg = DecoderConfig()
        cfg.B, cfg.Q, cfg.T, cfg.N, cfg.G, cfg.P, cfg.L = B, Q, smp.num_frames, 6, smp.num_groups, smp.num_points, smp.num_levels
        cfg.D, cfg.H, cfg.ffn = D, layer.self_attn.num_heads, layer.ffn.layers[0][0].weight.shape[0]
        cfg.num_classes, cfg.code_size, cfg.num_layers = layer.num_classes, layer.code_size, dec.num_layers
        cfg.out_points, cfg.attn_in_rows = layer.mixing.out_points, self._attn_in_rows
        cfg.feat_dtype = {torch.bfloat16: 1, torch.float16: 2}.get(pyramid.levels[0].dtype, 0)
        cfg.gemm_mode = self.mode_eff
        slots = getattr(pyramid, 'frame_slots', None)
        if slots is not None:
            cfg.n_slots = pyramid.n_slots
            for t, sl in enumerate(slots):
                cfg.frame_slots[t] = int(sl)
        cfg.overlap = int(self.overlap)
        if len(pyramid.levels) != cfg.L or pyramid.T != cfg.T or pyramid.B != B:
            raise RuntimeError('feature pyramid (L=%d, T=%d, B=%d) does not match the decoder config (L=%d, T=%d, B=%d)'
                               % (len(pyramid.levels), pyramid.T, pyramid.B, cfg.L, cfg.T, B))
        for l, f in enumerate(pyramid.levels):
            cfg.hw[l][0], cfg.hw[l][1] = f.shape[1], f.shape[2]
        cfg.image_h, cfg.image_w, cfg.eps_homo = float(ctx.image_h), float(ctx.image_w), 1e-5
        for i, v in enumerate(dec.pc_range):
            cfg.pc_range[i] = float(v)
        lib = _lib.load()
        need = lib.sbev_decoder_workspace_bytes(ctypes.byref(cfg))
        if need < 0:
            raise _lib.SbevError('sbev_decoder_workspace_bytes: ' + lib.sbev_last_error().decode())
        key = (str(dev), need)
        if own_workspace:                   # captured graphs keep a workspace of their own (eager calls use the runtime's): ONE per size
            # AND per stream -- graphs replayed on one stream run one after the other and may share the pair counters / exchange rows /
            # order buffer carved from it; two streams (multi-stream serving) would race on them.  Released with the last
            # graph that uses it (StepGraphs._release_ws).
            key = key + (torch.cuda.current_stream(dev).cuda_stream,)
            slot = self._graph_ws.get(key)
            if slot is None:
                slot = self._graph_ws[key] = [torch.zeros(need + 256, device=dev, dtype=torch.uint8), 0]      # (zeroed once: the on-demand relayout's flag words start clean)
            slot[1] += 1
            ws = slot[0]
            self._last_graph_ws_key = key
        else:
            if self._ws is None or self._ws_key != key:
                self._ws = torch.zeros(need + 256, device=dev, dtype=torch.uint8)
                self._ws_key = key
            ws = self._ws
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        cls = torch.empty(cfg.num_layers, B, Q, cfg.num_classes, device=dev, dtype=torch.float32)
        box = torch.empty(cfg.num_layers, B, Q, cfg.code_size, device=dev, dtype=torch.float32)
        feats = (ctypes.c_void_p * cfg.L)(*[f.data_ptr() for f in pyramid.levels])
        mask = attn_mask.to(device=dev, dtype=torch.uint8).contiguous() if attn_mask is not None else None
        qb, qf = query_bbox.contiguous(), query_feat.contiguous()
        args = (ctypes.byref(cfg), ctypes.byref(self._weights), feats, _ptr(qb), _ptr(qf),
                _ptr(ctx.time_diff), _ptr(ctx.lidar2img), _ptr(ctx.vel_div), _ptr(mask),
                _ptr(cls), _ptr(box), ctypes.c_void_p(ws_ptr), need)
        keep = (cfg, feats, qb, qf, mask, pyramid, ctx, ws, self._keep)
        return args, keep, cls, box

    def launches_per_layer(self, B, Q):
        """What sbev_decoder_forward would enqueue per layer for a [B, Q] call with the bound weights (asks the library)."""
        self._ensure_bound()
        dec, layer = self.decoder, self.decoder.decoder_layer
        smp = layer.sampling
        cfg = DecoderConfig()
        cfg.B, cfg.Q, cfg.T, cfg.N, cfg.G, cfg.P, cfg.L = B, Q, smp.num_frames, 6, smp.num_groups, smp.num_points, smp.num_levels
        cfg.D, cfg.H, cfg.ffn = layer.embed_dims, layer.self_attn.num_heads, layer.ffn.layers[0][0].weight.shape[0]
        cfg.num_classes, cfg.code_size, cfg.num_layers = layer.num_classes, layer.code_size, dec.num_layers
        cfg.out_points, cfg.attn_in_rows = layer.mixing.out_points, self._attn_in_rows
        cfg.gemm_mode, cfg.overlap = self.mode_eff, int(self.overlap)
        return int(_lib.load().sbev_decoder_launches_per_layer(ctypes.byref(cfg), ctypes.byref(self._weights)))

    def forward(self, query_bbox, query_feat, pyramid, ctx, attn_mask=None, finish=False):
        """pyramid: transformer.FeaturePyramid; ctx: transformer.DecoderContext.  Returns (cls, bbox) stacked over
        layers -- raw, or with ``finish`` nan_to_num'ed by one more launch of this library (sbev_finish_outputs: what
        SparseBEVTransformer.forward applies, models/sparsebev_transformer.py:35-36)."""
        if _STATE['chain_pair']:
            check_pair_faults()             # an earlier step lost a pair hand-off: raise before anything is enqueued on top of it
        args, _keep, cls, box = self._prepare(query_bbox, query_feat, pyramid, ctx, attn_mask)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib = _lib.load()
        st = lib.sbev_decoder_forward(*args, stream)
        _lib.check(st, 'sbev_decoder_forward')
        if finish:
            cls_o, box_o = torch.empty_like(cls), torch.empty_like(box)
            _lib.check(lib.sbev_finish_outputs(_ptr(cls), _ptr(box), _ptr(cls_o), _ptr(box_o), cls.numel(), box.numel(), stream), 'sbev_finish_outputs')
            return cls_o, box_o
        return cls, box

    def lazy_ok(self, mlvl_feats):
        """whether a feature LIST qualifies for the on-demand relayout of the eager step: the switch is on, every level is a contiguous,
        16-byte aligned NCHW device tensor [B, T*6, 256, H, W] of one dtype (what StepGraphs stages), 4 groups of 64 channels"""
        if not _STATE['lazy'] or hasattr(mlvl_feats, 'levels') or len(mlvl_feats) == 0:
            return False
        f0 = mlvl_feats[0]
        return all(torch.is_tensor(f) and f.is_cuda and StepGraphs._relayout_ok(f) and f.dtype == f0.dtype and f.shape[2] == 256 for f in mlvl_feats) \
            and self.decoder.decoder_layer.sampling.num_groups == 4

    def forward_lazy(self, query_bbox, query_feat, mlvl_feats, ctx, attn_mask=None, buffers=None, finish=False):
        """The eager step on the reference's NCHW feature list ``[B, T*6, 256, H_l, W_l]`` WITHOUT a dense relayout
        (sbev_decoder_forward_lazy): channels-last buffers (``buffers``, or new uninitialised ones) receive only the units the sample
        points read.  Bit-identical to ``forward`` on ``FeaturePyramid(mlvl_feats)``.  Returns (cls, box, pyramid-of-buffers)."""
        from . import transformer as TR
        if _STATE['chain_pair']:
            check_pair_faults()
        pyramid = buffers if buffers is not None else TR.FeaturePyramid.empty_like_nchw(mlvl_feats)
        args, _keep, cls, box = self._prepare(query_bbox, query_feat, pyramid, ctx, attn_mask)
        lib = _lib.load()
        if not lib.sbev_decoder_lazy_supported(args[0]):
            raise _lib.SbevError('sbev_decoder_forward_lazy does not cover this pyramid (4 groups x 64 channels; no frame ring)')
        lz = LazyFeats()
        for l, f in enumerate(mlvl_feats):
            if not (f.is_cuda and f.is_contiguous() and f.data_ptr() % 16 == 0 and f.dtype == pyramid.levels[l].dtype):
                raise RuntimeError('forward_lazy needs contiguous, 16-byte aligned device NCHW levels of the buffers\' dtype')
            lz.src[l] = f.data_ptr()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        st = lib.sbev_decoder_forward_lazy(args[0], args[1], args[2], ctypes.byref(lz), *args[3:], stream)
        _lib.check(st, 'sbev_decoder_forward_lazy')
        if finish:
            cls_o, box_o = torch.empty_like(cls), torch.empty_like(box)
            _lib.check(lib.sbev_finish_outputs(_ptr(cls), _ptr(box), _ptr(cls_o), _ptr(box_o), cls.numel(), box.numel(), stream), 'sbev_finish_outputs')
            return cls_o, box_o, pyramid
        return cls, box, pyramid

    def capture(self, query_bbox, query_feat, pyramid, ctx, attn_mask=None):
        """Record one decoder step into a hipGraph (sbev_decoder_capture) and return a DecoderGraph.  The graph reads
        its inputs through the captured pointers: refresh ``query_bbox`` / ``query_feat`` / the pyramid levels /
        ``ctx`` tensors IN PLACE (``.copy_()``), call ``replay()``, read ``graph.cls`` / ``graph.box``.  The runtime's
        workspace belongs to the graph while it is alive: use another DecoderRuntime for un-captured calls."""
        if not (query_bbox.is_contiguous() and query_feat.is_contiguous()):
            raise RuntimeError('capture needs contiguous query tensors (they are read in place on every replay)')
        args, keep, cls, box = self._prepare(query_bbox, query_feat, pyramid, ctx, attn_mask)
        side = torch.cuda.Stream(device=query_feat.device)
        side.wait_stream(torch.cuda.current_stream())
        handle = ctypes.c_void_p()
        st = _lib.load().sbev_decoder_capture(*args, ctypes.c_void_p(side.cuda_stream), ctypes.byref(handle))
        _lib.check(st, 'sbev_decoder_capture')
        return DecoderGraph(handle, keep, cls, box)


class StepGraphs:
    """hipGraph replay of the whole per-call step: staging of the inputs, feature relayout and all decoder launches are ONE captured
    graph from the second call of a kind on; per call the host packs the per-sample constants (time_diff, lidar2img) and the
    input POINTERS into one small array, refreshes the graph's device copy of it through the pinned upload ring and launches.
    Results are bit-identical to the eager path (same kernels, same order).

    What a graph is keyed on -- round 4: shapes, not addresses.  The reference's loops hand ``model(...)`` freshly allocated
    tensors every step (timing.py:77-96, val.py; mmdet's eval loop: new backbone outputs, new query tensors out of
    ``SparseBEVHead.forward``), so:
      * queries and the attention mask are always STAGED: one in-graph copy launch reads their addresses from the refreshed
        table (sbev_copy_indirect) into buffers the graph owns;
      * fp32 NCHW feature lists (the reference's layout) go through the in-graph relayout, whose source address also comes from
        the table (sbev_nchw_to_nhwc_f32_indirect) -- any tensors of the same shapes replay the same graph, and the graph pins
        none of the caller's tensors;
      * inputs that are read IN PLACE by the decoder kernels -- channels-last lists, FeaturePyramid / the online ring's buffers --
        keep their addresses in the key.  A tensor list is only captured when the SAME tensor objects come back (weak
        references from the first sighting: a recycled address of a dead tensor is not "the same input"), and while
        ``MAX_WASTED`` address-keyed graphs stand captured but never replayed no further one is captured (the ring's own
        buffers are persistent by construction and exempt).
    A weight update, other switches, another layer count or box convention miss the cache.  At most ``MAX`` graphs are kept (least
    recently used first out); the graphs of one workspace size replayed on one stream share one workspace (DecoderRuntime._graph_ws),
    freed with the last of them.
    Replaces per-call Python + launch overhead of the reference's eager module chain (models/sparsebev_transformer.py:86-97)."""

    MAX = 8
    MAX_WASTED = 3
    RETRY_EVERY = 64

    def __init__(self, runtime):
        self.rt = runtime
        self.entries = {}          # key -> dict(graph, ctx_buf, ...) or a _FirstSighting (seen once, not captured yet)
        self.replays = 0
        self.captures = 0
        self.wasted = 0            # address-keyed graphs evicted without a single replay (forgiven by a replay of one, or slowly by time)
        self._refused = 0          # calls refused a capture because of them
        self._warned = False

    def clear(self):
        for e in self.entries.values():
            if isinstance(e, dict):
                e['graph'].destroy()
                self._release_ws(e)
        self.entries.clear()

    def _release_ws(self, e):
        """one graph less on its workspace; the last one frees it"""
        slot = self.rt._graph_ws.get(e.get('ws_key'))
        if slot is not None:
            slot[1] -= 1
            if slot[1] <= 0:
                del self.rt._graph_ws[e['ws_key']]

    def _drop(self, key):
        e = self.entries.pop(key)
        if isinstance(e, dict):
            # "captured and never replayed" only counts against the caller when the graph could still have been hit: an entry of a
            # PREVIOUS weight signature (key[4]; every optimizer step makes one) was orphaned by the weight update, not by the caller
            # bringing new buffers (evaluate-between-optimizer-steps loops switched replay off for good)
            if e['pinned'] and e['replays'] == 0 and key[4] == self.rt._sig:
                self.wasted += 1
            e['graph'].destroy()
            self._release_ws(e)

    def _never_replayed(self):
        """keys of the live address-keyed graphs of the CURRENT weight signature that were captured and never replayed (an entry of an older
        signature was orphaned by the weight update, not by the caller bringing new buffers: ADVICE r5)"""
        return [k for k, e in self.entries.items() if isinstance(e, dict) and e['pinned'] and e['replays'] == 0 and k[4] == self.rt._sig]

    def _unproven(self):
        """address-keyed graphs (alive or already evicted) that were captured and never replayed afterwards"""
        return max(self.wasted, 0) + len(self._never_replayed())

    @staticmethod
    def _relayout_ok(f):
        return f.dim() == 5 and f.dtype in (torch.float32, torch.bfloat16, torch.float16) and f.is_contiguous() and f.data_ptr() % 16 == 0 and \
            not f.permute(0, 1, 3, 4, 2).is_contiguous()

    def _feat_key(self, feats):
        """(key part, the tensors whose identity an address-keyed entry depends on, staged?)"""
        if hasattr(feats, 'levels'):        # FeaturePyramid / RingPyramid: resident channels-last buffers (+ the ring's slot table);
            # their level tensors are views made per call, so there is no object identity to remember: MAX_WASTED bounds a caller
            # that builds a new pyramid over new buffers every step
            return ('pyr', tuple((f.data_ptr(), tuple(f.shape), f.dtype) for f in feats.levels),
                    tuple(getattr(feats, 'frame_slots', ())), getattr(feats, 'n_slots', 0)), [], False
        if all(self._relayout_ok(f) for f in feats):
            return ('nchw', tuple((tuple(f.shape), f.dtype) for f in feats)), [], True
        return ('list', tuple((f.data_ptr(), tuple(f.shape), tuple(f.stride()), f.dtype) for f in feats)), list(feats), False

    def run(self, query_bbox, query_feat, mlvl_feats, attn_mask, img_metas, finish=False):
        """(cls, box) of the graph's own output buffers (the caller clones or post-processes out of place), or None when this
        call has to take the eager path.  ``finish``: the graph's last node (sbev_finish_outputs_indirect) writes the nan_to_num'ed
        outputs into tensors allocated for THIS call, whose addresses travel in the pointer table -- nothing runs after the replay,
        and what is returned belongs to the caller."""
        rt = self.rt
        if _STATE['profile'] or not (query_bbox.is_cuda and query_bbox.is_contiguous() and query_feat.is_contiguous()):
            return None                      # (bracketing launches with HIP events needs the eager enqueue)
        if query_bbox.data_ptr() % 16 or query_feat.data_ptr() % 16:
            return None
        if attn_mask is not None and not (attn_mask.is_cuda and attn_mask.dtype == torch.uint8 and attn_mask.is_contiguous() and attn_mask.data_ptr() % 16 == 0):
            return None                      # the runtime would convert it into a temporary: nothing stable to stage from
        if not hasattr(mlvl_feats, 'levels') and not all(torch.is_tensor(f) and f.is_cuda for f in mlvl_feats):
            return None
        from . import transformer as TR
        if _STATE['chain_pair']:
            check_pair_faults()             # (before a capture or replay is built on an invalid step; one host-memory read)
        sig = rt._ensure_bound()
        fkey, ident, staged = self._feat_key(mlvl_feats)
        key = (tuple(query_bbox.shape), tuple(query_feat.shape), fkey, None if attn_mask is None else tuple(attn_mask.shape), sig,
               torch.cuda.current_device(), _STATE['row_chain'], _STATE['chain_pair'], _STATE['fuse'], _STATE['order'], _lib.load().sbev_get_box_convention(),
               rt.decoder.num_layers, tuple(rt.decoder.pc_range),
               torch.cuda.current_stream(query_bbox.device).cuda_stream,      # per stream: a graph's workspace belongs to the stream it replays on
               bool(finish), _STATE['relayout_multi'], _STATE['lazy'], _STATE['out_fold'])
        e = self.entries.get(key, False)
        if e is False or (isinstance(e, _FirstSighting) and not e.same(ident)):
            # first sighting (or an address whose tensor died and was recycled): eager this time, capture if it comes again
            if not staged and not hasattr(mlvl_feats, 'frame_slots') and self._unproven() >= self.MAX_WASTED:
                # not for the life of the runtime: every RETRY_EVERY-th refused call forgives one never-replayed capture, so a caller
                # that starts re-using its buffers later gets its graph after all (one probe capture per RETRY_EVERY calls otherwise)
                self._refused += 1
                if self._refused % self.RETRY_EVERY == 0:
                    if self.wasted > 0:
                        self.wasted -= 1
                    else:
                        stale = next(iter(self._never_replayed()), None)
                        if stale is not None:
                            self._drop(stale)         # (counts it as wasted ...)
                            self.wasted = max(self.wasted - 1, 0)      # ... which this probe forgives: net zero, never negative
                if not self._warned:
                    import warnings
                    warnings.warn('sparsebev_amd: %d step graphs keyed on input addresses were captured and never replayed (the caller '
                                  'passes new channels-last / pyramid buffers every call): no further address-keyed graphs; pass fp32 '
                                  'NCHW feature lists or reuse the buffers' % self._unproven())
                    self._warned = True
                return None
            if e is False and len(self.entries) >= self.MAX:
                self._drop(next(iter(self.entries)))
            self.entries[key] = _FirstSighting(ident)
            return None
        B = query_bbox.shape[0]
        packed, layout, ih, iw = TR.DecoderContext.pack(img_metas, B)
        outs = None
        if finish:                           # this call's own output tensors (the graph's last node writes them through the table)
            nl, nc, cs = rt.decoder.num_layers, rt.decoder.decoder_layer.num_classes, rt.decoder.decoder_layer.code_size
            Q = query_bbox.shape[1]
            outs = (torch.empty(nl, B, Q, nc, device=query_bbox.device, dtype=torch.float32),
                    torch.empty(nl, B, Q, cs, device=query_bbox.device, dtype=torch.float32))
        if isinstance(e, _FirstSighting):
            e = self._capture(key, query_bbox, query_feat, mlvl_feats, attn_mask, B, packed, layout, ih, iw, staged, outs)
            if e is None:
                return None
        else:
            if layout != e['layout'] or (ih, iw) != e['image']:
                return None                  # other camera count / image size: not this graph's step
            TR._upload(self._with_table(packed, e, query_bbox, query_feat, mlvl_feats, attn_mask, outs), query_bbox.device, out=e['ctx_buf'])
        self.entries[key] = self.entries.pop(key)            # most recently used last
        e['graph'].replay()
        e['replays'] += 1
        self.replays += 1
        if e['pinned'] and e['replays'] > 0:
            self.wasted = 0                  # the caller DOES bring its buffers back: earlier never-replayed captures are forgiven
        return outs if outs is not None else (e['graph'].cls, e['graph'].box)

    @staticmethod
    def _with_table(packed, e, query_bbox, query_feat, mlvl_feats, attn_mask, outs=None):
        """packed per-sample constants + the pointer table of this call (int64 words viewed as fp32 pairs) in ONE upload.
        Table: 0 query_bbox, 1 query_feat, 2 mask, [3 .. 3 + L) the NCHW levels (staged entries), then -- ``finish`` entries -- the
        call's output tensors cls, bbox."""
        import numpy as np
        ptrs = [query_bbox.data_ptr(), query_feat.data_ptr(), attn_mask.data_ptr() if attn_mask is not None else 0]
        if e['staged']:
            ptrs += [f.data_ptr() for f in mlvl_feats]
        if outs is not None:
            ptrs += [outs[0].data_ptr(), outs[1].data_ptr()]
        full = np.empty(e['n_packed'] + 2 * len(ptrs), dtype=np.float32)
        full[:packed.size] = packed
        full[packed.size:e['n_packed']] = 0.0
        full[e['n_packed']:].view(np.int64)[:] = ptrs
        return full

    def _capture(self, key, query_bbox, query_feat, mlvl_feats, attn_mask, B, packed, layout, ih, iw, staged, outs=None):
        import numpy as np
        from . import transformer as TR
        rt, lib = self.rt, _lib.load()
        dev = query_bbox.device
        # graph-owned staging buffers of the queries / mask (the decoder kernels read these)
        qb = torch.empty_like(query_bbox)
        qf = torch.empty_like(query_feat)
        mask = torch.empty_like(attn_mask) if attn_mask is not None else None
        relayout = []                                         # (table index, resident NHWC buffer, n_images, channels, hw)
        if hasattr(mlvl_feats, 'levels'):
            pyramid = mlvl_feats
        else:
            pyramid = TR.FeaturePyramid.__new__(TR.FeaturePyramid)
            f0 = mlvl_feats[0]
            pyramid.B, TN, pyramid.GC = f0.shape[0], f0.shape[1], f0.shape[2]
            pyramid.T = TN // TR.N_VIEWS
            pyramid.levels, pyramid.copied = [], 0
            for l, f in enumerate(mlvl_feats):
                nhwc = f.permute(0, 1, 3, 4, 2)
                if staged:
                    buf = torch.empty(f.shape[0], TN, f.shape[3], f.shape[4], pyramid.GC, device=dev, dtype=f.dtype)
                    relayout.append((3 + l, buf, f.shape[0] * TN, pyramid.GC, f.shape[3] * f.shape[4]))
                    nhwc = buf
                    pyramid.copied += 1
                elif not nhwc.is_contiguous():
                    return None              # a layout neither read in place nor taken by the in-graph relayout kernel: eager path
                pyramid.levels.append(nhwc.reshape(pyramid.B * TN, f.shape[3], f.shape[4], pyramid.GC))
        n_packed = (packed.size + 3) // 4 * 4                 # the table behind the constants, 16-byte aligned
        # 'replays' counts launches AFTER the capturing call's own; 'pinned': keyed on addresses of buffers a caller may not bring back
        # (the online ring's buffers are persistent by construction)
        e = {'staged': staged, 'n_packed': n_packed, 'layout': layout, 'image': (ih, iw), 'replays': -1,
             'pinned': not staged and not hasattr(mlvl_feats, 'frame_slots')}
        full = self._with_table(packed, e, query_bbox, query_feat, mlvl_feats, attn_mask, outs)
        ctx = TR.DecoderContext.from_packed(full, layout, ih, iw, dev)     # the graph reads constants AND table from this tensor on every replay
        table = ctypes.c_void_p(ctx.buffer.data_ptr() + 4 * n_packed)
        args, keep, cls, box = rt._prepare(qb, qf, pyramid, ctx, mask, own_workspace=True)
        e['ws_key'] = rt._last_graph_ws_key
        try:                                 # (anything that raises before the capture proper must give the workspace reference back: ADVICE r5)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            sp = ctypes.c_void_p(side.cuda_stream)
            segs = [(0, qb), (1, qf)] + ([(2, mask)] if mask is not None else [])
            c_idx = (ctypes.c_int32 * len(segs))(*[i for i, _ in segs])
            c_dst = (ctypes.c_void_p * len(segs))(*[t.data_ptr() for _, t in segs])
            c_nb = (ctypes.c_int64 * len(segs))(*[t.numel() * t.element_size() for _, t in segs])
            _lib.check(lib.sbev_capture_begin(sp), 'sbev_capture_begin')
        except BaseException:
            self._release_ws(e)
            raise
        ok = True
        st_fwd = 0
        try:
            ok = lib.sbev_copy_indirect(table, len(segs), c_idx, c_dst, c_nb, sp) == 0
            # on-demand relayout: no dense pass at all -- every layer moves the units its sample points marked (sbev_decoder_forward_lazy)
            lazy = bool(relayout) and staged and len(relayout) == len(mlvl_feats) and _STATE['lazy'] and bool(lib.sbev_decoder_lazy_supported(args[0]))
            multi = (not lazy and len(relayout) > 1 and all(r[1].dtype == torch.float32 and r[2] == relayout[0][2] and r[3] == relayout[0][3] and r[3] % 4 == 0
                                               and r[4] % 4 == 0 for r in relayout) and len(relayout) <= MAX_LEVELS and _STATE['relayout_multi'])
            if multi:                        # every fp32 level of the pyramid in ONE launch (the coarse levels ride in the finest one's tail)
                n = len(relayout)
                c_ix = (ctypes.c_int32 * n)(*[r[0] for r in relayout])
                c_out = (ctypes.c_void_p * n)(*[r[1].data_ptr() for r in relayout])
                c_hw = (ctypes.c_int32 * n)(*[r[4] for r in relayout])
                ok = ok and lib.sbev_nchw_to_nhwc_f32_multi_indirect(table, n, c_ix, c_out, relayout[0][2], relayout[0][3], c_hw, sp) == 0
            elif not lazy:
                for idx, buf, n_img, ch, hw in relayout:
                    fn = lib.sbev_nchw_to_nhwc_f32_indirect if buf.dtype == torch.float32 else lib.sbev_nchw_to_nhwc_b16_indirect
                    ok = ok and fn(table, idx, _ptr(buf), n_img, ch, hw, sp) == 0
            if lazy:
                lz = LazyFeats()
                lz.table = table
                for l, r in enumerate(relayout):
                    lz.index[l] = r[0]
                st_fwd = lib.sbev_decoder_forward_lazy(args[0], args[1], args[2], ctypes.byref(lz), *args[3:], sp) if ok else 0
            else:
                st_fwd = lib.sbev_decoder_forward(*args, sp) if ok else 0
            ok = ok and st_fwd == 0
            if outs is not None:
                i_out = 3 + (len(mlvl_feats) if staged else 0)
                ok = ok and lib.sbev_finish_outputs_indirect(table, i_out, i_out + 1, _ptr(cls), _ptr(box), cls.numel(), box.numel(), sp) == 0
        finally:
            handle = ctypes.c_void_p()
            st = lib.sbev_capture_end(sp, ctypes.byref(handle) if ok else None)
        if not ok or st != 0:
            self._release_ws(e)
        if st_fwd == _lib.EFAULT:            # an EARLIER step lost a pair hand-off: the caller sees PairFaultError (pair mode off, acknowledged), as on the eager path
            _lib.check(st_fwd, 'sbev_decoder_forward (capture)')
        if not ok:
            raise _lib.SbevError('graph capture of the decoder step failed: ' + lib.sbev_last_error().decode())
        _lib.check(st, 'sbev_capture_end')
        torch.cuda.current_stream().wait_stream(side)
        # what the graph's launches read or write: its own buffers, and -- address-keyed entries only -- the caller's feature buffers
        pinned = None if staged else (mlvl_feats, pyramid)
        graph = DecoderGraph(handle, (keep, [r[1] for r in relayout], qb, qf, mask, ctx, pinned), cls, box)
        e.update(graph=graph, ctx_buf=ctx.buffer)
        self.entries[key] = e
        self.captures += 1
        return e


class _FirstSighting:
    """An input combination seen once.  Address-keyed entries remember WHAT MEMORY the tensors were (weakly): the second sighting only
    counts when the same live storages come back at the same offsets / shapes / strides -- a recycled address of a freed tensor is a
    different input, while a caller that re-creates its views over persistent channels-last buffers every call (``permute`` /
    ``reshape`` per forward: new tensor OBJECTS, same storage) still matches (round 4 compared object identity and left that caller
    eager for good, without a word -- ADVICE r4).  A storage's Python object lives exactly as long as some tensor uses it."""

    def __init__(self, tensors):
        import weakref
        self.refs = [(weakref.ref(t.untyped_storage()), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype) for t in tensors]

    def same(self, tensors):
        return len(tensors) == len(self.refs) and all(
            r[0]() is not None and r[0]() is t.untyped_storage() and r[1:] == (t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)
            for r, t in zip(self.refs, tensors))


class DecoderGraph:
    """An instantiated hipGraph of one decoder step (see DecoderRuntime.capture)."""

    def __init__(self, handle, keep, cls, box):
        self._h, self._keep, self.cls, self.box = handle, keep, cls, box
        self.num_nodes = int(_lib.load().sbev_graph_num_nodes(handle))

    def replay(self):
        if self._h is None:
            raise RuntimeError('DecoderGraph was destroyed')
        st = _lib.load().sbev_graph_launch(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(st, 'sbev_graph_launch')
        return self.cls, self.box

    def destroy(self):
        if self._h is not None:
            _lib.load().sbev_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# process-wide switches mirrored here so that a captured step is only replayed under the settings it was recorded with
import os as _os
_STATE = {'row_chain': True, 'chain_pair': not _os.environ.get('SBEV_NO_CHAIN_PAIR'), 'fuse': True, 'profile': 0,
          'out_fold': bool(_os.environ.get('SBEV_OUT_FOLD')),          # A/B (off: measured slower): the out-projection folds its split-K slabs inside its launch
          'lazy': not _os.environ.get('SBEV_NO_SPARSE_RELAYOUT'),      # staged NCHW pyramids: on-demand relayout of the units the sample points read
          'relayout_multi': not _os.environ.get('SBEV_NO_RELAYOUT_MULTI'),      # staged fp32 NCHW pyramids: all levels in one launch (A/B switch)
          'order': (lambda v: 2 if v == 2 else int(v != 0))(int(_os.environ.get('SBEV_QUERY_ORDER', '0') or 0))}


def out_fold(enable):
    """The out-projection GEMM folds its split-K slabs inside its launch (the chunk-workgroups of a row tile meet at a counter; fp16 GEMM
    modes, row chains, <= ~1000 rows; follows ``chain_pair``) so that the tail chain reads one row block instead of 32 slabs.  Bit-identical
    results either way.  OFF by default -- at config 2 it costs the out-projection 12.6 us and saves the tail 4;
    ``SBEV_OUT_FOLD=1`` starts with it on.  Returns the previous setting."""
    prev = bool(_lib.load().sbev_decoder_out_fold(int(bool(enable))))
    _STATE['out_fold'] = bool(enable)
    return prev


def lazy_relayout(enable):
    """Replayable steps on NCHW feature lists move only the feature units the sample points read (on-demand relayout, default on;
    ``SBEV_NO_SPARSE_RELAYOUT=1`` starts with the dense relayout).  Bit-identical results either way.  Returns the previous setting."""
    prev = _STATE['lazy']
    _STATE['lazy'] = bool(enable)
    return prev


def query_order(enable):
    """The fused gather + mixing launch walks its items in sbev_query_order's order (one group and one arc of the camera ring per
    XCD: 20 % fewer fabric reads at config 2, not faster -- DESIGN_HISTORY.md section 10.8; off by default).  Bit-identical results either
    way.  Returns the previous setting."""
    mode = 2 if enable == 2 and enable is not True else int(bool(enable))      # 2: sorted once per step (from the input boxes) instead of every layer
    prev = int(_lib.load().sbev_decoder_query_order(mode))
    _STATE['order'] = mode
    return prev


def fuse_sample_mix(enable):
    """Gather + adaptive mixing as one launch inside sbev_decoder_forward where the fused kernel covers the shape (default on;
    results are bit-identical either way).  ``SBEV_NO_SAMPLE_MIX=1`` in the environment switches it off for A/B runs."""
    _lib.check(_lib.load().sbev_decoder_fuse_sample_mix(int(bool(enable))), 'sbev_decoder_fuse_sample_mix')
    _STATE['fuse'] = bool(enable)


def row_chain(enable):
    """The row-local op chains of a layer (position encoder, attention projections, norms, ffn, branches, refine_bbox) as three
    launches with the rows in LDS instead of one launch per op (default on where the kernels cover the layer's shape; results
    agree to fp32 round-off, not bit for bit).  ``SBEV_NO_ROW_CHAIN=1`` in the environment switches it off for A/B runs."""
    _lib.check(_lib.load().sbev_decoder_row_chain(int(bool(enable))), 'sbev_decoder_row_chain')
    _STATE['row_chain'] = bool(enable)


def chain_pair(enable):
    """The tail chain on PAIRS of workgroups per row block (half the weight stream per CU, two in-launch hand-offs; default on
    where both members of every pair fit the device in one round: <= ~2000 rows on 256 CUs; results agree with the
    single-workgroup tail to fp32 round-off).  ``SBEV_NO_CHAIN_PAIR=1`` in the environment starts with it off.  Returns the
    previous setting."""
    prev = _lib.load().sbev_decoder_chain_pair(int(bool(enable)))
    _STATE['chain_pair'] = bool(enable)
    return bool(prev)


def _on_pair_fault():
    """SBEV_EFAULT came back (``_lib.check``): mirror what the library did (pair mode off -- captured pair-mode steps are keyed on
    the switch and are not hit again) and acknowledge, so that the caller's repeated step runs on the single-workgroup tail."""
    lib = _lib.load()
    lib.sbev_decoder_chain_pair(0)
    _STATE['chain_pair'] = False
    lib.sbev_decoder_chain_pair_faults_ack()
    import warnings
    warnings.warn('sparsebev_amd: a pair-mode tail hand-off timed out (GPU shared / preempted?); pair mode is off for the rest of the process')


def check_pair_faults():
    """For callers that synchronise themselves: right AFTER ``torch.cuda.synchronize()`` (or an event / ``.item()`` on the step's
    outputs) this tells whether the step just finished -- or any since the last check -- lost a pair hand-off; raises
    ``_lib.PairFaultError`` (pair mode off, acknowledged) if so.  Costs one read of pinned host memory; never synchronises.  Without
    it a fault surfaces at the NEXT decoder call instead (``sbev_decoder_forward`` / graph replay refuse with SBEV_EFAULT)."""
    n = int(_lib.load().sbev_decoder_chain_pair_faults())
    if n > 0:
        _on_pair_fault()
        raise _lib.PairFaultError('%d pair-mode hand-off(s) timed out since the last check: the step(s) since then hold invalid rows; '
                                  'pair mode is off now, repeat them' % n)


def chain_pair_timeouts():
    """Pair hand-offs whose partner did not arrive within the poll bound since the library was loaded (0 on a GPU of our own;
    synchronises the device)."""
    n = _lib.load().sbev_decoder_chain_pair_timeouts()
    if n < 0:
        raise RuntimeError('sbev_decoder_chain_pair_timeouts: HIP error')
    return int(n)


def profile_sampler(enable):
    """enable: False / True (sampler launches only) or an int mask (1 sampler | 2 generator GEMM | 4 out-projection GEMM |
    8 fused gather + mixing)."""
    _lib.load().sbev_profile_sampler(int(enable))
    _STATE['profile'] = int(enable)


def profile_stride(every_n_calls):
    """Bracket the launches of only every n-th decoder call (the event records themselves cost ~2 % of a step)."""
    _lib.check(_lib.load().sbev_profile_stride(int(every_n_calls)), 'sbev_profile_stride')


def read_kernel_ms(kind, max_n=4096):
    """Elapsed ms of the bracketed launches of one kind (0 sampler, 1 generator GEMM, 2 out-projection GEMM, 3 fused gather +
    mixing)."""
    buf = (ctypes.c_float * max_n)()
    n = _lib.load().sbev_profile_read(kind, buf, max_n)
    return [buf[i] for i in range(n)]


def read_sampler_ms(max_n=4096):
    return read_kernel_ms(0, max_n)
