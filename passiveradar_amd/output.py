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
order, edge chunks, fill value) that does not share code with the writer.  HDF5 output (main.py:208-214): h5py is
not installed either, but the HDF5 C library is, so the file is written by libhdf5 itself through ctypes (see below).
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["save_range_doppler", "save_range_doppler_zarr", "load_range_doppler_zarr", "save_metadata",
           "validate_zarr_v2_metadata", "read_zarr_v2", "write_zarr_v2", "zarray_json", "save_range_doppler_hdf5", "load_range_doppler_hdf5",
           "hdf5_available"]


# ---- HDF5 (main.py:208-214) through the HDF5 C library itself ---------------------------------------------------
# h5py is not installed here, but libhdf5 is (conda's copy in this image; any system copy elsewhere): the file is
# written by the real library through ctypes -- dataset '/xambg', shape (F, R+1, nframes), complex64 stored the way
# h5py stores NumPy complex numbers (compound {'r': float32, 'i': float32}), contiguous layout -- so that
# range_doppler_plot.py:43-47 (`np.abs(f['/xambg'])`) reads it unchanged.
_H5 = {"lib": None, "tried": False}


def _hdf5_lib():
    import ctypes as C
    import ctypes.util
    import glob
    if _H5["tried"]:
        return _H5["lib"]
    _H5["tried"] = True
    cands = [os.environ.get("PRC_HDF5_LIB"), ctypes.util.find_library("hdf5")]
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat))
    for c in cands:
        if not c:
            continue
        try:
            h = C.CDLL(c)
            if h.H5open() < 0:
                continue
            # hid_t is 64 bits wide from HDF5 1.10 on (32 bits in 1.8.x, where the signatures below would pass garbage
            # handles): older libraries are refused rather than bound
            maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
            if h.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel)) < 0 or (maj.value, mnr.value) < (1, 10):
                continue
        except (OSError, AttributeError):
            continue
        hid, hsz, pp = C.c_int64, C.c_uint64, C.POINTER(C.c_uint64)
        sig = {"H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
               "H5Fclose": (C.c_int, [hid]), "H5Screate_simple": (hid, [C.c_int, pp, pp]), "H5Sclose": (C.c_int, [hid]),
               "H5Sselect_hyperslab": (C.c_int, [hid, C.c_int, pp, pp, pp, pp]),
               "H5Sget_simple_extent_ndims": (C.c_int, [hid]), "H5Sget_simple_extent_dims": (C.c_int, [hid, pp, pp]),
               "H5Tcreate": (hid, [C.c_int, C.c_size_t]), "H5Tinsert": (C.c_int, [hid, C.c_char_p, C.c_size_t, hid]),
               "H5Tclose": (C.c_int, [hid]), "H5Tget_class": (C.c_int, [hid]), "H5Tget_nmembers": (C.c_int, [hid]),
               "H5Tget_member_name": (C.c_void_p, [hid, C.c_uint]), "H5Tget_size": (C.c_size_t, [hid]),
               "H5free_memory": (C.c_int, [C.c_void_p]),
               "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]),
               "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]), "H5Dclose": (C.c_int, [hid]),
               "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
               "H5Dread": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p])}
        try:
            for name, (res, args) in sig.items():
                fn = getattr(h, name)
                fn.restype, fn.argtypes = res, args
            h._f32le = C.c_int64.in_dll(h, "H5T_IEEE_F32LE_g").value
        except (AttributeError, ValueError):
            continue
        _H5["lib"] = h
        break
    return _H5["lib"]


def hdf5_available():
    return _hdf5_lib() is not None


def _h5_complex64(h):
    t = h.H5Tcreate(6, 8)                                   # H5T_COMPOUND, 8 bytes
    if t < 0 or h.H5Tinsert(t, b"r", 0, h._f32le) < 0 or h.H5Tinsert(t, b"i", 4, h._f32le) < 0:
        raise OSError("HDF5: could not build the complex64 compound type")
    return t


def save_range_doppler_hdf5(path, frames, dataset="/xambg"):
    """frames: [nframes][F][R+1] complex64 -> HDF5 file at ``path`` holding dataset '/xambg' of shape
    (F, R+1, nframes) (main.py:208-214).  Frame i is written into the hyperslab [:, :, i]; nothing is transposed in
    host memory."""
    import ctypes as C
    h = _hdf5_lib()
    if h is None:
        raise NotImplementedError("HDF5 output needs the HDF5 C library (libhdf5), which was not found; set "
                                  "PRC_HDF5_LIB or use range_doppler_map_ftype: 'zarr'")
    if hasattr(frames, "cpu"):
        frames = frames.cpu().numpy()
    frames = np.ascontiguousarray(frames, dtype=np.complex64)
    nframes, F, cols = frames.shape
    A3 = C.c_uint64 * 3
    A2 = C.c_uint64 * 2
    f = h.H5Fcreate(os.fsencode(path), 2, 0, 0)             # H5F_ACC_TRUNC
    if f < 0:
        raise OSError(f"HDF5: cannot create {path}")
    t = fs = ms = d = -1
    try:
        t = _h5_complex64(h)
        fs = h.H5Screate_simple(3, A3(F, cols, nframes), None)
        ms = h.H5Screate_simple(2, A2(F, cols), None)
        d = h.H5Dcreate2(f, dataset.encode(), t, fs, 0, 0, 0)
        if min(fs, ms, d) < 0:
            raise OSError("HDF5: dataset creation failed")
        for i in range(nframes):
            if h.H5Sselect_hyperslab(fs, 0, A3(0, 0, i), None, A3(F, cols, 1), None) < 0 or \
                    h.H5Dwrite(d, t, ms, fs, 0, frames[i].ctypes.data) < 0:
                raise OSError(f"HDF5: writing frame {i} failed")
    finally:
        for closer, hid in ((h.H5Dclose, d), (h.H5Sclose, ms), (h.H5Sclose, fs), (h.H5Tclose, t)):
            if hid >= 0:
                closer(hid)
        h.H5Fclose(f)
    return path


def load_range_doppler_hdf5(path, dataset="/xambg"):
    """Read '/xambg' back as the (F, R+1, nframes) complex64 array; checks that it is stored as h5py stores complex
    numbers (compound of two float32 members named 'r' and 'i')."""
    import ctypes as C
    h = _hdf5_lib()
    if h is None:
        raise NotImplementedError("the HDF5 C library (libhdf5) was not found")
    f = h.H5Fopen(os.fsencode(path), 0, 0)                  # H5F_ACC_RDONLY
    if f < 0:
        raise OSError(f"HDF5: cannot open {path}")
    d = sp = ft = mt = -1
    try:
        d = h.H5Dopen2(f, dataset.encode(), 0)
        if d < 0:
            raise KeyError(dataset)
        sp, ft = h.H5Dget_space(d), h.H5Dget_type(d)
        nd = h.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        h.H5Sget_simple_extent_dims(sp, dims, None)
        names = []
        for m in range(max(h.H5Tget_nmembers(ft), 0)):
            p = h.H5Tget_member_name(ft, m)
            names.append(C.string_at(p).decode())
            h.H5free_memory(p)
        if h.H5Tget_class(ft) != 6 or names != ["r", "i"] or h.H5Tget_size(ft) != 8:
            raise ValueError(f"{dataset} is not a complex64 dataset in h5py's convention (members {names})")
        out = np.empty(tuple(int(v) for v in dims), dtype=np.complex64)
        mt = _h5_complex64(h)
        if h.H5Dread(d, mt, 0, 0, 0, out.ctypes.data) < 0:    # H5S_ALL
            raise OSError("HDF5: read failed")
        return out
    finally:
        for closer, hid in ((h.H5Tclose, mt), (h.H5Tclose, ft), (h.H5Sclose, sp), (h.H5Dclose, d)):
            if hid >= 0:
                closer(hid)
        h.H5Fclose(f)


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


def _zarr_fill(meta, dt):
    fv = meta["fill_value"]
    if fv is None:
        return 0
    if dt.kind == "c":
        return complex(float(fv[0]), float(fv[1]))
    return {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(fv, fv) if isinstance(fv, str) else fv


def _zarr_codec(meta):
    """(decode, encode) of the chunk compressor: null (raw chunks) or zlib -- the codec the v2 specification's own
    storage example uses, available in the standard library; anything else needs numcodecs"""
    comp = meta["compressor"]
    if meta["filters"]:
        raise NotImplementedError("filtered chunks need numcodecs")
    if comp is None:
        return (lambda b: b), (lambda b: b)
    if comp.get("id") == "zlib":
        import zlib
        level = int(comp.get("level", 1))
        return zlib.decompress, (lambda b: zlib.compress(b, level))
    raise NotImplementedError(f"compressor {comp.get('id')!r} needs numcodecs")


def zarray_json(meta):
    """the bytes of a ``.zarray`` document as zarr-python 2.x writes them (sorted keys, four-space indent, ASCII)"""
    return json.dumps(meta, indent=4, sort_keys=True, ensure_ascii=True, separators=(",", ": ")).encode("ascii")


def read_zarr_v2(path):
    """Spec-driven reader of a zarr v2 array in a directory store: any chunk grid, C or F order inside a chunk, edge
    chunks stored at full chunk size, missing chunks = fill_value, raw or zlib-compressed chunks.  Independent of how
    the store was written -- the tests read save_range_doppler_zarr's output through it."""
    meta = json.load(open(os.path.join(path, ".zarray")))
    dt = validate_zarr_v2_metadata(meta)
    decode, _ = _zarr_codec(meta)
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    out = np.full(shape, _zarr_fill(meta, dt), dtype=dt)
    sep = meta.get("dimension_separator", ".")
    grid = [range(-(-s // c)) for s, c in zip(shape, chunks)]
    import itertools
    for idx in itertools.product(*grid):
        f = os.path.join(path, sep.join(str(i) for i in idx))
        if not os.path.exists(f):
            continue
        raw = np.frombuffer(decode(open(f, "rb").read()), dtype=dt)
        if raw.size != int(np.prod(chunks)):
            raise ValueError(f"chunk {idx}: {raw.size} items, the chunk shape holds {int(np.prod(chunks))}")
        block = raw.reshape(chunks, order=meta["order"])
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = block[tuple(slice(0, sl.stop - sl.start) for sl in sel)]
    return out


def write_zarr_v2(path, array, meta):
    """The general writer behind the store of main.py:216-224: ``array`` into a directory store described by the
    ``.zarray`` document ``meta`` (any chunk grid, C / F order, '.' or '/' separators, raw or zlib chunks; edge chunks
    are stored at full chunk size, padded with the fill value; every chunk is written, also those equal to it)."""
    import itertools
    dt = validate_zarr_v2_metadata(meta)
    array = np.asarray(array)
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    if tuple(array.shape) != shape:
        raise ValueError(f"write_zarr_v2: array of shape {array.shape}, .zarray says {shape}")
    _, encode = _zarr_codec(meta)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, ".zarray"), "wb") as fh:
        fh.write(zarray_json(meta))
    sep = meta.get("dimension_separator", ".")
    fill = _zarr_fill(meta, dt)
    for idx in itertools.product(*[range(-(-s // c)) for s, c in zip(shape, chunks)]):
        block = np.full(chunks, fill, dtype=dt)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        block[tuple(slice(0, sl.stop - sl.start) for sl in sel)] = array[sel]
        f = os.path.join(path, sep.join(str(i) for i in idx))
        os.makedirs(os.path.dirname(f), exist_ok=True)
        with open(f, "wb") as fh:
            fh.write(encode(block.tobytes(order=meta["order"])))
    return path


def save_range_doppler(config, frames):
    """main.py:208-227: write the maps in the format ``config['range_doppler_map_ftype']`` names, to
    ``config['range_doppler_map_fname']`` (dict of passiveradar_amd.config.getConfiguration).  'zarr': directory
    store written here; 'hdf5': dataset '/xambg' written through the HDF5 C library (NotImplementedError if libhdf5
    cannot be found).  Anything else: the reference's ValueError."""
    ftype = config["range_doppler_map_ftype"]
    if ftype == "zarr":
        return save_range_doppler_zarr(config["range_doppler_map_fname"], frames)
    if ftype == "hdf5":
        return save_range_doppler_hdf5(config["range_doppler_map_fname"], frames)
    raise ValueError("Unsupported output file type. Enter 'hdf5' or 'zarr'")


class ZarrFrameWriter:
    """The zarr v2 store of main.py:216-224, (F, R+1, nframes) complex64 in chunks (F, R+1, 1), written frame by frame:
    chunk '0.0.i' IS frame i in C order, so frames can be stored in any order and as they arrive (the reference's
    dask graph stores chunk by chunk as well); nothing is transposed in host memory."""

    def __init__(self, path, F, cols, nframes):
        self.path, self.F, self.cols, self.nframes = path, int(F), int(cols), int(nframes)
        os.makedirs(path, exist_ok=True)
        # what zarr.open(mode='w', shape=..., chunks=(F, R+1, 1), dtype=complex64) of main.py:216-221 records, except for
        # the compressor (zarr's default is Blosc; null = raw chunks, which every v2 reader accepts)
        meta = {"zarr_format": 2, "shape": [self.F, self.cols, self.nframes], "chunks": [self.F, self.cols, 1], "dtype": "<c8",
                "compressor": None, "fill_value": [0.0, 0.0], "order": "C", "filters": None}
        validate_zarr_v2_metadata(meta)
        with open(os.path.join(path, ".zarray"), "wb") as fh:
            fh.write(zarray_json(meta))
        with open(os.path.join(path, ".zattrs"), "w") as fh:
            fh.write("{}")

    def write(self, first, frames):
        """frames: [m][F][R+1] complex64 (NumPy) = frames first .. first + m - 1 of the stream"""
        frames = np.asarray(frames)
        if frames.dtype != np.complex64 or frames.shape[1:] != (self.F, self.cols) or first < 0 or first + frames.shape[0] > self.nframes:
            raise ValueError(f"ZarrFrameWriter.write: block {frames.shape} {frames.dtype} at frame {first} does not fit "
                             f"({self.nframes}, {self.F}, {self.cols}) complex64")
        for k in range(frames.shape[0]):
            np.ascontiguousarray(frames[k]).tofile(os.path.join(self.path, f"0.0.{first + k}"))


def save_range_doppler_zarr(path, frames):
    """frames: [nframes][F][R+1] complex64 (NumPy array or torch tensor) -> zarr v2 store at ``path``
    holding the (F, R+1, nframes) array of main.py:216-224."""
    if hasattr(frames, "cpu"):
        frames = frames.cpu().numpy()
    frames = np.ascontiguousarray(frames, dtype=np.complex64)
    nframes, F, cols = frames.shape
    ZarrFrameWriter(path, F, cols, nframes).write(0, frames)
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
