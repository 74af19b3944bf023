"""Module-level switches the reference exposes next to the hot path (models/utils.py:309-325)."""
import os
import tempfile

import torch


class DumpConfig:
    """The ``DUMP`` debug taps (models/utils.py:309-317).  When ``enabled`` the decoder writes, per stage,
    ``sample_points_cam_stage{i}.pth`` ([B,T,N,Q,GP,3] = u, v, max(homo,eps)),
    ``sample_points_cam_valid_mask_stage{i}.pth`` ([B,T,N,Q,GP] float 0/1), ``sasa_tau_stage{i}.pth``,
    ``query_bbox_stage{i}.pth``, ``bbox_pred_stage{i}.pth`` and ``cls_score_stage{i}.pth`` into ``out_dir``
    -- the files viz_sample_points.py:83-105 reads."""

    def __init__(self):
        self.enabled = False
        self.out_dir = tempfile.mkdtemp()
        self.stage_count = 0
        self.frame_count = 0

    def save(self, name, tensor):
        torch.save(tensor.detach().cpu(), os.path.join(self.out_dir, '%s_stage%d.pth' % (name, self.stage_count)))


DUMP = DumpConfig()


class Version:
    """Checkpoint convention switch (models/utils.py:320-325): ``VERSION.name`` is 'v1.0.0' (default) or 'v0.17.1'
    (set from ``checkpoint['version']``, val.py:128-129).  It flips the rotation sign of the sample offsets
    (rotation_3d_in_axis, models/utils.py:66-77) and the box layout of ``get_bboxes`` (models/sparsebev_head.py:472-476);
    assigning it forwards the choice to the library (``sbev_set_box_convention``), which reads it at launch time."""
    _CODES = {'v1.0.0': 0, 'v0.17.1': 1}

    def __init__(self):
        self._name = 'v1.0.0'

    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, value):
        if value not in self._CODES:
            raise NotImplementedError("unknown box convention %r (the reference knows 'v1.0.0' and 'v0.17.1')" % (value,))
        from . import _lib
        _lib.check(_lib.load().sbev_set_box_convention(self._CODES[value]), 'sbev_set_box_convention')
        self._name = value

    def require_supported(self):
        """The library and the Python switch must agree (someone may have called sbev_set_box_convention directly)."""
        from . import _lib
        if _lib.load().sbev_get_box_convention() != self._CODES[self._name]:
            raise RuntimeError('VERSION.name = %r but libsbev_hip.so is set to convention %d'
                               % (self._name, _lib.load().sbev_get_box_convention()))


VERSION = Version()
