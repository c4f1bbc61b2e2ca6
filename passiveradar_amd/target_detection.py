"""Drop-in for the one function of the reference's ``passiveRadar/target_detection.py`` that sits next
to the range-Doppler path (SURVEY 8f "next" #3): ``CFAR_2D`` (:683-703), applied per frame to
``|xambg|`` by range_doppler_plot.py:56-57.  The Kalman trackers of that module are out of scope."""
from __future__ import annotations

import numpy as np

from . import _lib, engine
from ._lib import check, lib

__all__ = ["CFAR_2D"]


def CFAR_2D(X, fw, gw, thresh=None):
    """Constant false alarm rate filter (same contract as target_detection.py:683-703): X is a 2-D
    (Doppler x range) magnitude map -- or a stack [nframes, H, W] / torch device tensor for batches;
    returns the CFAR ratio in float64 (boolean if ``thresh`` is given), like the reference."""
    if _lib.is_device_tensor(X):
        import torch
        x = X.to(torch.float32).contiguous()
        frames = 1 if x.dim() == 2 else x.shape[0]
        H, W = x.shape[-2], x.shape[-1]
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):          # launch on the tensor's device and torch's stream there
            check(lib().prc_cfar2d(x.data_ptr(), H, W, int(fw), int(gw), int(thresh is not None),
                                   float(thresh or 0.0), out.data_ptr(), frames, _lib.torch_stream_ptr(x.device)))
        return out.bool() if thresh is not None else out
    x = np.ascontiguousarray(X, dtype=np.float32)
    frames = 1 if x.ndim == 2 else x.shape[0]
    H, W = x.shape[-2], x.shape[-1]
    st = engine.staging()
    dx = st.get("cfar_x", x.nbytes)
    do = st.get("cfar_o", x.nbytes)
    dx.upload(x)
    check(lib().prc_cfar2d(dx.ptr, H, W, int(fw), int(gw), int(thresh is not None), float(thresh or 0.0),
                           do.ptr, frames, None))
    out = do.download(x.shape, np.float32)
    return out > 0.5 if thresh is not None else out.astype(np.float64)
