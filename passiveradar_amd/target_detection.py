"""Drop-in for the one function of the reference's ``passiveRadar/target_detection.py`` that sits next
to the range-Doppler path (SURVEY 8f "next" #3): ``CFAR_2D`` (:683-703), applied per frame to
``|xambg|`` by range_doppler_plot.py:56-57.  The Kalman trackers of that module are out of scope."""
from __future__ import annotations

import numpy as np

from . import _lib, engine
from ._lib import check, lib

__all__ = ["CFAR_2D", "CFAR_2D_abs"]


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


def CFAR_2D_abs(xambg, fw, gw, thresh=None):
    """``CFAR_2D(np.abs(xambg), fw, gw, thresh)`` -- the call of range_doppler_plot.py:56-57 -- in ONE kernel: the
    complex64 range-Doppler map (2-D, a stack [nframes, H, W], or torch device tensors of those shapes) goes in as it
    is and |X| is taken while the CFAR tiles are loaded, so the complex map is read once and no magnitude map is written.  Same return convention as
    CFAR_2D."""
    if _lib.is_device_tensor(xambg):
        import torch
        x = xambg.to(torch.complex64).contiguous()
        if x.dim() not in (2, 3):
            raise ValueError("CFAR_2D_abs takes a 2-D map or a stack [nframes, H, W]")
        frames = 1 if x.dim() == 2 else x.shape[0]
        H, W = x.shape[-2], x.shape[-1]
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().prc_cfar2d_c64(x.data_ptr(), H, W, int(fw), int(gw), int(thresh is not None),
                                       float(thresh or 0.0), out.data_ptr(), frames, _lib.torch_stream_ptr(x.device)))
        return out.bool() if thresh is not None else out
    x = np.ascontiguousarray(xambg, dtype=np.complex64)
    if x.ndim not in (2, 3):
        raise ValueError("CFAR_2D_abs takes a 2-D map or a stack [nframes, H, W]")
    frames = 1 if x.ndim == 2 else x.shape[0]
    H, W = x.shape[-2], x.shape[-1]
    st = engine.staging()
    dx = st.get("cfar_xc", x.nbytes)
    do = st.get("cfar_o", x.nbytes // 2)
    dx.upload(x)
    check(lib().prc_cfar2d_c64(dx.ptr, H, W, int(fw), int(gw), int(thresh is not None), float(thresh or 0.0),
                               do.ptr, frames, None))
    out = do.download(x.shape, np.float32)
    return out > 0.5 if thresh is not None else out.astype(np.float64)
