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
    """Checkpoint convention switch (models/utils.py:320-325).  Only 'v1.0.0' (the default, and the only
    convention any shipped config uses) is implemented by the HIP kernels."""

    def __init__(self):
        self.name = 'v1.0.0'

    def require_supported(self):
        """Called on every forward: a checkpoint that sets the old convention (val.py:128-129 of the reference) must not be
        run silently with the wrong rotation sign / box layout."""
        if self.name != 'v1.0.0':
            raise NotImplementedError("sparsebev_amd implements the 'v1.0.0' box / rotation convention only (VERSION.name = %r): "
                                      "rotation_3d_in_axis and get_bboxes differ for 'v0.17.1' (models/utils.py:66-77, "
                                      "models/sparsebev_head.py:472-476)" % (self.name,))


VERSION = Version()
