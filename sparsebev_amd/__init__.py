"""sparsebev_amd -- MI355X (gfx950) native implementation of SparseBEV's decoder hot path:
adaptive spatio-temporal sampling + adaptive mixing + scale-adaptive self attention, behind the
reference's own operator / registry interfaces (msmv_sampling, SparseBEVTransformer).

Python here is host glue (tensors, streams, torch.distributed); the work is done by hand-written HIP
kernels in ``csrc/libsbev_hip.so`` bound through a C ABI (``include/sbev_hip.h``).  Importing the
operator modules without the built library raises -- there is no CPU / PyTorch fallback.
"""
from ._lib import load as load_library, SbevError, LIB_PATH   # noqa: F401

__all__ = ['load_library', 'SbevError', 'LIB_PATH']
