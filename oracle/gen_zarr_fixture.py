#!/usr/bin/env python3
"""tests/golden/zarr_v2_spec_example/: the array of the zarr v2 storage specification's worked example ("storing a
single array": zarr.create(shape=(20, 20), chunks=(10, 10), dtype='i4', fill_value=42, compressor=Zlib(level=1))),
written by hand from the specification's rules with nothing but the standard library -- NOT by zarr (not installed
here), not by passiveradar_amd.output (this is what its reader and writer are held against).

  .zarray   the document the specification prints for that array: sorted keys, four-space indent
  i.j       chunk (i, j) of the 2 x 2 grid: zlib (level 1) of the chunk's 100 little-endian int32 in C order
            holding arange(400).reshape(20, 20)

    python oracle/gen_zarr_fixture.py
"""
import os
import struct
import zlib

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "zarr_v2_spec_example")
ZARRAY = """{
    "chunks": [
        10,
        10
    ],
    "compressor": {
        "id": "zlib",
        "level": 1
    },
    "dtype": "<i4",
    "fill_value": 42,
    "filters": null,
    "order": "C",
    "shape": [
        20,
        20
    ],
    "zarr_format": 2
}"""

if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    with open(os.path.join(HERE, ".zarray"), "w") as fh:
        fh.write(ZARRAY)
    for i in range(2):
        for j in range(2):
            vals = [(10 * i + r) * 20 + (10 * j + c) for r in range(10) for c in range(10)]
            with open(os.path.join(HERE, f"{i}.{j}"), "wb") as fh:
                fh.write(zlib.compress(struct.pack("<100i", *vals), 1))
    with open(os.path.join(HERE, "zlib_version.txt"), "w") as fh:
        fh.write(zlib.ZLIB_RUNTIME_VERSION + "\n")
    print("wrote", sorted(os.listdir(HERE)))
