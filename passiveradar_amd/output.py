"""Output side of main.py:200-224 (SURVEY 8f "next" #4): the range-Doppler maps as a zarr v2
directory store with the reference's array geometry -- shape (F, R+1, nframes), dtype complex64, C order,
chunks (F, R+1, 1) -- plus the ``.npz`` axis metadata, so that the reference's plot / tracker scripts
(``zarr.load(fname)``, range_doppler_plot.py:43-49) read it unchanged.

The frame block this package produces, [nframes][F][R+1] complex64 C order, is byte-for-byte one zarr
chunk per frame, so writing is a plain dump of each frame.  Chunks are stored uncompressed
(``"compressor": null``): zarr / numcodecs are not a dependency here, and the reference only relies on
zarr's defaults when *reading*.
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["save_range_doppler_zarr", "load_range_doppler_zarr", "save_metadata"]


def save_range_doppler_zarr(path, frames):
    """frames: [nframes][F][R+1] complex64 (NumPy array or torch tensor) -> zarr v2 store at ``path``
    holding the (F, R+1, nframes) array of main.py:216-224."""
    if hasattr(frames, "cpu"):
        frames = frames.cpu().numpy()
    frames = np.ascontiguousarray(frames, dtype=np.complex64)
    nframes, F, cols = frames.shape
    os.makedirs(path, exist_ok=True)
    meta = {"zarr_format": 2, "shape": [F, cols, nframes], "chunks": [F, cols, 1], "dtype": "<c8",
            "compressor": None, "fill_value": None, "order": "C", "filters": None}
    with open(os.path.join(path, ".zarray"), "w") as fh:
        json.dump(meta, fh, indent=1)
    with open(os.path.join(path, ".zattrs"), "w") as fh:
        fh.write("{}")
    for i in range(nframes):
        frames[i].tofile(os.path.join(path, f"0.0.{i}"))       # chunk (F, R+1, 1) == frame i, C order
    return path


def load_range_doppler_zarr(path):
    """Read a store written by save_range_doppler_zarr back as the (F, R+1, nframes) array."""
    meta = json.load(open(os.path.join(path, ".zarray")))
    F, cols, nframes = meta["shape"]
    assert meta["chunks"] == [F, cols, 1] and meta["dtype"] == "<c8" and meta["compressor"] is None
    out = np.empty((F, cols, nframes), dtype=np.complex64)
    for i in range(nframes):
        out[:, :, i] = np.fromfile(os.path.join(path, f"0.0.{i}"), dtype=np.complex64).reshape(F, cols)
    return out


def save_metadata(config, nframes, fname=None):
    """The ``.npz`` written at main.py:200-206 (frame_timestamps, range_bins, doppler_bins -- the latter
    with the reference's 2F entries, kept as is)."""
    F, cols = config["num_doppler_cells"], config["num_range_cells"] + 1
    frame_timestamps = np.arange(nframes) * config["frame_interval"]
    range_bins = np.arange(cols) * config["range_cell_width"]
    doppler_bins = np.arange(-1 * F, F) * config["doppler_cell_width"]
    fname = fname or config["meta_fname"]
    np.savez(fname, frame_timestamps=frame_timestamps, range_bins=range_bins, doppler_bins=doppler_bins)
    return fname
