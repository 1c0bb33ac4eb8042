"""Monotonic alignment search on the MI355X: mirror of `monotonic_align.maximum_path`
(training/vits2/monotonic_align/__init__.py:6-20), whose Cython core (core.pyx:7-42) is the reference repo's only
native code.  Same call: `maximum_path(neg_cent, mask)` with neg_cent, mask [b, t_t, t_s]; returns the 0/1 path in
neg_cent's type (torch tensor -> torch tensor on the same device and dtype, numpy -> numpy float32).  The DP runs in a
HIP kernel behind `vits_mas_maximum_path` (include/vits_mi355.h); there is no CPU fallback.
"""
import numpy as np

from .capi import VitsLib

_lib = None


def _get_lib():
    global _lib
    if _lib is None:
        _lib = VitsLib()
    return _lib


def maximum_path(neg_cent, mask, device=0, lib=None):
    is_torch = hasattr(neg_cent, "detach")
    if is_torch:
        t_dev, t_dtype = neg_cent.device, neg_cent.dtype
        values = neg_cent.detach().cpu().numpy().astype(np.float32)
        m = mask.detach().cpu().numpy()
    else:
        values = np.asarray(neg_cent, dtype=np.float32)
        m = np.asarray(mask)
    # t_t_max = mask.sum(1)[:, 0], t_s_max = mask.sum(2)[:, 0]   (__init__.py:17-18)
    t_ys = m.sum(1)[:, 0].astype(np.int32)
    t_xs = m.sum(2)[:, 0].astype(np.int32)
    path = (lib or _get_lib()).mas_maximum_path(values, t_ys, t_xs, device)
    if is_torch:
        import torch

        return torch.from_numpy(path).to(device=t_dev, dtype=t_dtype)
    return path.astype(np.float32)
