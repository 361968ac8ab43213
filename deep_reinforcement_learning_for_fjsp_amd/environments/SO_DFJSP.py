"""Host-side mirror of the reference's environments/SO_DFJSP.py (the environment agents/DA3C instantiates):
SO_FJSSP.py line for line, but over class_FJSP.py instead of class_FJSSP.py -- every job's due date is its
order's delivery time (class_FJSP.py:229 instead of the rounded per-job dates of class_FJSSP.py:214-218) and
Machine.gap_ave has no 1e-18 in its divisor (:159).  Same actions ([6, 5]), state (20) and kernels; the library
variant only changes the due dates the packer writes.
"""
from ..batch import VARIANT_SO_DFJSP
from .SO_FJSSP import BatchedSOFJSSP, SO_FJSSP_Environment


class BatchedSODFJSP(BatchedSOFJSSP):
    """Vectorised SO_DFJSP: reset() -> f64[N,20]; step(actions u8[N,2]) -> (state, reward, done) device tensors."""
    variant = VARIANT_SO_DFJSP


class SO_DFJSP_Environment(SO_FJSSP_Environment):
    """Drop-in for environments/SO_DFJSP.py:13 (N = 1 view of the batched kernels)."""
    environment_name = "Single object DFJSP"          # SO_DFJSP.py:15
    variant = VARIANT_SO_DFJSP
