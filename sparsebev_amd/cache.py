"""Online (streaming) per-frame feature ring -- SURVEY.md section 8f rank 2.

The reference's published FPS is measured in online mode: features of past frames are cached per frame and only
the 6 new images go through the backbone (models/sparsebev.py:255-321) -- but every step it still ``torch.cat``s
all T cached frames (:297-303) and the decoder then regroup-copies them again (models/sparsebev_transformer.py:
73-85).  Here each level is ONE resident channels-last buffer ``[B, n_slots, 6, H, W, C]``; a new frame is
relayouted (NCHW -> NHWC, one launch per level) straight into the slot of the evicted frame and the sampler reads
logical frame t through a slot table (``sbev_msmv_fwd_ring``), so nothing older than the newest frame is ever moved.
"""
import ctypes

import torch

from . import _lib, ops

N_VIEWS = 6


class FrameFeatureCache:
    """``dtype``: the ring's STORAGE type.  fp32 (default) takes fp32 NCHW frames and channels-last fp32 / fp16 / bf16 frames (widened);
    ``torch.float16`` / ``torch.bfloat16`` keep an fp16 backbone's / a bf16 neck's frames as they are (half the memory and the relayout
    traffic; the sampler widens a tap exactly) and take frames of that type only, NCHW (2-byte relayout) or channels-last (copied)."""

    def __init__(self, num_frames, n_slots=None, dtype=torch.float32):
        if dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise ValueError('ring storage must be fp32, fp16 or bf16')
        self.dtype = dtype
        self.T = num_frames
        self.n_slots = n_slots or num_frames
        if not self.T <= self.n_slots <= 16:
            raise ValueError('need num_frames <= n_slots <= 16 (the reference evicts its cache at 16 frames)')
        self.buffers = None            # list[L] of [B, n_slots, 6, H, W, C]
        self.order = []                # physical slots, newest first
        self.B = None

    def _alloc(self, frame_feats):
        f0 = frame_feats[0]
        self.B = f0.shape[0]
        self.buffers = [torch.empty(self.B, self.n_slots, N_VIEWS, f.shape[3], f.shape[4], f.shape[2], device=f.device, dtype=self.dtype)
                        for f in frame_feats]

    def push(self, frame_feats):
        """frame_feats: list[L] of [B, 6, C, H_l, W_l] device tensors = the neck's output for the 6 NEW images.
        fp32 NCHW memory is relayouted by the transpose kernel; channels-last memory (what a channels_last conv stack
        emits; fp32 / fp16 / bf16) is already in the ring's layout and only copied (and widened) into its slot."""
        if self.buffers is None:
            self._alloc(frame_feats)
        slot = len(self.order) if len(self.order) < self.n_slots else self.order.pop()      # free slot, else evict the oldest
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for f, buf in zip(frame_feats, self.buffers):
            if not f.is_cuda or f.shape[0] != self.B or f.shape[1] != N_VIEWS:
                raise RuntimeError('frame features must be device tensors [B, 6, C, H, W]')
            if self.dtype != torch.float32:                     # 2-byte ring: frames of the ring's own type only, moved as bytes
                if f.dtype != self.dtype:
                    raise RuntimeError('a %s ring takes %s frames only (got %s)' % (self.dtype, self.dtype, f.dtype))
                if f.stride(2) == 1 and f[0].is_contiguous(memory_format=torch.channels_last):
                    for b in range(self.B):
                        buf[b, slot].copy_(f[b].permute(0, 2, 3, 1))      # both sides contiguous [6, H, W, C]: a device memcpy
                else:
                    f = f.contiguous()
                    C, H, W = f.shape[2:]
                    for b in range(self.B):
                        st = lib.sbev_nchw_to_nhwc_b16(ctypes.c_void_p(f[b].data_ptr()), ctypes.c_void_p(buf[b, slot].data_ptr()),
                                                       N_VIEWS, C, H * W, stream)
                        _lib.check(st, 'sbev_nchw_to_nhwc_b16')
                continue
            if f.stride(2) == 1 and f[0].is_contiguous(memory_format=torch.channels_last):      # NHWC memory: zero relayout
                code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}.get(f.dtype)
                if code is None:
                    raise RuntimeError('channels-last frame features must be fp32 / fp16 / bf16')
                for b in range(self.B):          # f[b] is one contiguous [6, H, W, C] run in memory; so is buf[b, slot]
                    st = lib.sbev_copy_widen_f32(ctypes.c_void_p(f[b].data_ptr()), code, ctypes.c_void_p(buf[b, slot].data_ptr()),
                                                 buf[b, slot].numel(), stream)
                    _lib.check(st, 'sbev_copy_widen_f32')
                continue
            if f.dtype != torch.float32:
                raise RuntimeError('NCHW frame features must be fp32 (channels-last inputs may be fp16 / bf16)')
            f = f.contiguous()
            C, H, W = f.shape[2:]
            for b in range(self.B):
                st = lib.sbev_nchw_to_nhwc_f32(ctypes.c_void_p(f[b].data_ptr()), ctypes.c_void_p(buf[b, slot].data_ptr()),
                                               N_VIEWS, C, H * W, stream)
                _lib.check(st, 'sbev_nchw_to_nhwc_f32')
        self.order.insert(0, slot)
        del self.order[self.n_slots:]

    def pyramid(self):
        """View of the newest T frames for the decoder (drop-in for transformer.FeaturePyramid)."""
        if len(self.order) < self.T:
            raise RuntimeError('only %d of %d frames cached' % (len(self.order), self.T))
        return RingPyramid(self)


class RingPyramid:
    def __init__(self, cache):
        self.B, self.T = cache.B, cache.T
        self.n_slots = cache.n_slots
        self.frame_slots = list(cache.order[:cache.T])
        self.levels = [b.reshape(cache.B * cache.n_slots * N_VIEWS, b.shape[3], b.shape[4], b.shape[5]) for b in cache.buffers]
        self.GC = cache.buffers[0].shape[-1]
        self.copied = 0

    def sample(self, loc, w_bp, T, G):
        return ops.msmv_sampling_ring(self.levels, self.B, T, G, self.frame_slots, self.n_slots, loc, w_bp)
