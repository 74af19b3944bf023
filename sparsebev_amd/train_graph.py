"""One training step of the decoder -- forward + backward through all layers -- as ONE hipGraph replay.

The differentiable path (transformer.forward_differentiable, sparsebev_amd.autograd) is ~550 kernel launches per step at config 2;
issued one by one from Python and the autograd engine they cost 10.5-11.2 ms for 8.6 ms of kernel time.  Captured once and replayed,
the same launches take 8.2 ms (tools/bench_train.py --graph; profiles/r4_train_step.log), with gradients bit-identical to the eager
step's.  This is torch.cuda.graphs' whole-network recipe; what this module adds is what the decoder needs to be capturable:

  * the per-call constants (camera matrices, time stamps: host data in ``img_metas``) are uploaded OUTSIDE the graph into one device
    buffer that later steps refresh in place (transformer.DecoderContext(..., out=buffer));
  * everything -- warm-up, capture, replay -- runs on ONE side stream: the parameters' AccumulateGrad nodes remember the stream they
    were created on, and a backward under capture that has to hop to another, non-capturing stream does not end with an error but
    with a crash inside hipStreamEndCapture; the warm-up watches for torch's stream-mismatch warning and refuses to capture instead;
  * dropout: the differentiable path draws one host seed per layer call, which a capture freezes -- so with dropout on, the step is
    captured with a DEVICE seed installed (autograd.device_seed): every dropout launch hashes host seed + one int64 word in device
    memory, and ``replay`` draws a new word (one small torch launch, torch's CUDA generator) before it replays the graph.  Forward and
    backward of a step see the same word.

Inputs are STATIC tensors: write the next batch into them with ``copy_()`` (the features, the queries), pass the next ``img_metas`` to
``replay``.  Gradients land in the same ``.grad`` tensors every replay and OVERWRITE them (they were ``None`` at capture): run the
optimizer after ``replay()``, and never ``zero_grad(set_to_none=True)`` -- that would detach ``.grad`` from the graph's buffers.

Reference counterpart: the training loop around SparseBEVTransformer.forward (models/sparsebev_transformer.py:86-101) and its
backward through msmv_sampling (models/csrc/wrapper.py:51-61); the reference has no graph capture.
"""
import warnings

import torch

from . import autograd as AG
from .transformer import DecoderContext, FeaturePyramid, _upload
from .utils import VERSION

class CapturedTrainStep:
    """``step = CapturedTrainStep(model, query_bbox, query_feat, mlvl_feats, img_metas, loss_fn)`` captures
    ``loss_fn(*model(...)).backward()``; ``loss, cls_scores, bbox_preds = step.replay(img_metas)`` runs it again on the current contents
    of the (static) input tensors.  ``step.grads`` maps parameter names to the static gradient tensors, ``step.input_grads`` holds
    ``query_feat.grad`` / the feature gradients where those inputs require grad."""

    def __init__(self, model, query_bbox, query_feat, mlvl_feats, img_metas, loss_fn, attn_mask=None, warmup=3):
        if not torch.is_grad_enabled():
            raise RuntimeError('CapturedTrainStep: grad is disabled')
        VERSION.require_supported()
        self.model = model
        self.decoder = dec = model.decoder if hasattr(model, 'decoder') else model
        layer = dec.decoder_layer
        self.dropout = bool(dec.training and (layer.self_attn.attn_drop > 0 or layer.ffn_drop > 0))
        if not (query_bbox.is_cuda and query_bbox.dtype == torch.float32 and query_bbox.is_contiguous()
                and query_feat.dtype == torch.float32 and query_feat.is_contiguous()):
            raise ValueError('CapturedTrainStep: query_bbox / query_feat must be contiguous fp32 CUDA tensors (they are read in place)')
        self.query_bbox, self.query_feat, self.attn_mask = query_bbox, query_feat, attn_mask
        self.mlvl_feats = list(mlvl_feats)
        self.loss_fn = loss_fn
        self.B = query_bbox.shape[0]
        self.device = query_bbox.device
        self.ctx = DecoderContext(img_metas, self.B, self.device)          # host -> device: outside the graph
        # the part of the dropout seeds that changes from replay to replay (see the module text); None without dropout
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device).random_() if self.dropout else None
        self._leaves = [p for p in dec.parameters() if p.requires_grad]
        self._leaves += [t for t in [query_feat, query_bbox] + self.mlvl_feats if t.requires_grad]
        self.stream = torch.cuda.Stream(device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            mismatch = False
            for i in range(max(1, warmup)):
                self._clear_grads()
                with warnings.catch_warnings(record=True) as seen:
                    warnings.simplefilter('always')
                    self._step()
                mismatch = any('AccumulateGrad' in str(w.message) and 'stream' in str(w.message) for w in seen)
            if mismatch:
                raise RuntimeError("CapturedTrainStep: a parameter's AccumulateGrad node still belongs to another stream (an autograd graph "
                                   'of an earlier eager step is alive); capturing now would crash inside HIP.  Drop the references to earlier '
                                   'losses / outputs, or build the captured step before the first eager step.')
            self._clear_grads()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss, self.cls_scores, self.bbox_preds = self._step()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.grads = {n: p.grad for n, p in dec.named_parameters() if p.grad is not None}
        self.input_grads = {'query_feat': query_feat.grad, 'mlvl_feats': [f.grad for f in self.mlvl_feats]}

    def _clear_grads(self):
        for t in self._leaves:
            t.grad = None

    def _step(self):
        with AG.device_seed(self.seed_dev):
            return self._run()

    def _run(self):
        dec = self.decoder
        pyr = FeaturePyramid(list(self.mlvl_feats))          # the NCHW -> NHWC relayout: device work, part of the step
        cls, box = dec.forward_differentiable(self.query_bbox, self.query_feat, list(self.mlvl_feats), pyr, self.attn_mask, self.ctx)
        if dec is not self.model:                           # SparseBEVTransformer.forward's nan_to_num
            cls, box = torch.nan_to_num(cls), torch.nan_to_num(box)
        loss = self.loss_fn(cls, box)
        loss.backward()
        return loss.detach(), cls.detach(), box.detach()

    def replay(self, img_metas=None, new_masks=True):
        """Run the captured step on the current contents of the static inputs; ``img_metas`` (same batch size, frame and camera
        counts) refreshes the camera matrices / time stamps first; with dropout on, new masks are drawn unless ``new_masks=False``
        (``seed_dev`` may also be set by hand).  Returns the graph's own (loss, cls_scores, bbox_preds) tensors -- clone what has to
        survive the next replay."""
        if img_metas is not None:
            packed, layout, image_h, image_w = DecoderContext.pack(img_metas, self.B)
            if layout != self.ctx.layout or (image_h, image_w) != (self.ctx.image_h, self.ctx.image_w):
                raise ValueError('CapturedTrainStep.replay: img_metas with another shape than the captured step (frames, cameras, image size)')
            _upload(packed, self.device, out=self.ctx.buffer)          # in place: the graph reads this buffer
        if self.seed_dev is not None and new_masks:
            self.seed_dev.random_()
        self.graph.replay()
        return self.loss, self.cls_scores, self.bbox_preds
