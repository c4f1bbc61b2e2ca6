"""Output side of main.py:200-224 (SURVEY 8f "next" #4): the range-Doppler maps as a zarr v2
directory store with the reference's array geometry -- shape (F, R+1, nframes), dtype complex64, C order,
chunks (F, R+1, 1) -- plus the ``.npz`` axis metadata, so that the reference's plot / tracker scripts
(``zarr.load(fname)``, range_doppler_plot.py:43-49) read it unchanged.

The frame block this package produces, [nframes][F][R+1] complex64 C order, is byte-for-byte one zarr
chunk per frame, so writing is a plain dump of each frame.  Chunks are stored uncompressed
(``"compressor": null``): zarr / numcodecs are not a dependency here, and the reference only relies on
zarr's defaults when *reading*.  zarr itself is not installed in this image, so the store is checked against
the zarr v2 storage specification instead: ``validate_zarr_v2_metadata`` (required ``.zarray`` fields and their
types) and ``read_zarr_v2``, a reader written from the specification (chunk grid, chunk file names, in-chunk
order, edge chunks, fill value) that does not share code with the writer.  HDF5 output (main.py:208-214) is
NOT built: h5py is unavailable and an HDF5 container could not be verified against a real reader here.
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["save_range_doppler", "save_range_doppler_zarr", "load_range_doppler_zarr", "save_metadata",
           "validate_zarr_v2_metadata", "read_zarr_v2"]


def validate_zarr_v2_metadata(meta):
    """Check a ``.zarray`` document against the zarr storage specification version 2 (the fields every v2 reader,
    zarr-python's ``zarr.load`` of range_doppler_plot.py:43-49 included, requires).  Raises ValueError."""
    required = ("zarr_format", "shape", "chunks", "dtype", "compressor", "fill_value", "order", "filters")
    missing = [k for k in required if k not in meta]
    if missing:
        raise ValueError(f".zarray lacks required keys {missing}")
    if meta["zarr_format"] != 2:
        raise ValueError("zarr_format must be the integer 2")
    shape, chunks = meta["shape"], meta["chunks"]
    if not (isinstance(shape, list) and all(isinstance(v, int) and v >= 0 for v in shape)):
        raise ValueError("shape must be a list of non-negative integers")
    if not (isinstance(chunks, list) and len(chunks) == len(shape) and all(isinstance(v, int) and v > 0 for v in chunks)):
        raise ValueError("chunks must be a list of positive integers, one per dimension")
    try:
        dt = np.dtype(meta["dtype"])
    except TypeError as e:
        raise ValueError(f"dtype {meta['dtype']!r} is not a NumPy typestr") from e
    if isinstance(meta["dtype"], str) and meta["dtype"][0] not in "<>|":
        raise ValueError("dtype must carry an explicit byte order (<, > or |)")
    if meta["order"] not in ("C", "F"):
        raise ValueError("order must be 'C' or 'F'")
    comp = meta["compressor"]
    if comp is not None and not (isinstance(comp, dict) and "id" in comp):
        raise ValueError("compressor must be null or an object with an 'id'")
    if meta["filters"] is not None and not isinstance(meta["filters"], list):
        raise ValueError("filters must be null or a list")
    if meta.get("dimension_separator", ".") not in (".", "/"):
        raise ValueError("dimension_separator must be '.' or '/'")
    fv = meta["fill_value"]
    if dt.kind == "c" and fv is not None and not (isinstance(fv, list) and len(fv) == 2):
        raise ValueError("a complex fill_value is encoded as [real, imag]")
    return dt


def read_zarr_v2(path):
    """Spec-driven reader of an UNCOMPRESSED zarr v2 array in a directory store: any chunk grid, C or F order inside
    a chunk, edge chunks stored at full chunk size, missing chunks = fill_value.  Independent of how the store was
    written -- the tests read save_range_doppler_zarr's output through it."""
    meta = json.load(open(os.path.join(path, ".zarray")))
    dt = validate_zarr_v2_metadata(meta)
    if meta["compressor"] is not None or meta["filters"]:
        raise NotImplementedError("compressed / filtered chunks need numcodecs")
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    fv = meta["fill_value"]
    if fv is None:
        fill = 0
    elif dt.kind == "c":
        fill = complex(float(fv[0]), float(fv[1]))
    else:
        fill = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(fv, fv) if isinstance(fv, str) else fv
    out = np.full(shape, fill, dtype=dt)
    sep = meta.get("dimension_separator", ".")
    grid = [range(-(-s // c)) for s, c in zip(shape, chunks)]
    import itertools
    for idx in itertools.product(*grid):
        f = os.path.join(path, sep.join(str(i) for i in idx))
        if not os.path.exists(f):
            continue
        raw = np.fromfile(f, dtype=dt)
        if raw.size != int(np.prod(chunks)):
            raise ValueError(f"chunk {idx}: {raw.size} items, the chunk shape holds {int(np.prod(chunks))}")
        block = raw.reshape(chunks, order=meta["order"])
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = block[tuple(slice(0, sl.stop - sl.start) for sl in sel)]
    return out


def save_range_doppler(config, frames):
    """main.py:208-227: write the maps in the format ``config['range_doppler_map_ftype']`` names, to
    ``config['range_doppler_map_fname']`` (dict of passiveradar_amd.config.getConfiguration).  'zarr' is built;
    'hdf5' is not (h5py is not available to this build and an HDF5 container written without it could not be
    verified against a real reader): NotImplementedError.  Anything else: the reference's ValueError."""
    ftype = config["range_doppler_map_ftype"]
    if ftype == "zarr":
        return save_range_doppler_zarr(config["range_doppler_map_fname"], frames)
    if ftype == "hdf5":
        raise NotImplementedError("HDF5 output (main.py:208-214, dataset '/xambg') is not built: h5py is not "
                                  "available here; use range_doppler_map_ftype: 'zarr'")
    raise ValueError("Unsupported output file type. Enter 'hdf5' or 'zarr'")


def save_range_doppler_zarr(path, frames):
    """frames: [nframes][F][R+1] complex64 (NumPy array or torch tensor) -> zarr v2 store at ``path``
    holding the (F, R+1, nframes) array of main.py:216-224."""
    if hasattr(frames, "cpu"):
        frames = frames.cpu().numpy()
    frames = np.ascontiguousarray(frames, dtype=np.complex64)
    nframes, F, cols = frames.shape
    os.makedirs(path, exist_ok=True)
    # what zarr.open(mode='w', shape=..., chunks=(F, R+1, 1), dtype=complex64) of main.py:216-221 records, except for
    # the compressor (zarr's default is Blosc; null = raw chunks, which every v2 reader accepts)
    meta = {"zarr_format": 2, "shape": [F, cols, nframes], "chunks": [F, cols, 1], "dtype": "<c8",
            "compressor": None, "fill_value": [0.0, 0.0], "order": "C", "filters": None}
    validate_zarr_v2_metadata(meta)
    with open(os.path.join(path, ".zarray"), "w") as fh:
        json.dump(meta, fh, indent=1)
    with open(os.path.join(path, ".zattrs"), "w") as fh:
        fh.write("{}")
    for i in range(nframes):
        frames[i].tofile(os.path.join(path, f"0.0.{i}"))       # chunk (F, R+1, 1) == frame i, C order
    return path


def load_range_doppler_zarr(path):
    """Read a store written by save_range_doppler_zarr back as the (F, R+1, nframes) array (through the generic
    spec-driven reader)."""
    return read_zarr_v2(path)


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
