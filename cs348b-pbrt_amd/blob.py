"""Named-array container used for scene blobs and golden fixtures (tests/golden/*.bin).

Layout (little endian): b"PVB1", u32 n_entries, then per entry
  char name[48] (NUL padded), u32 dtype (0=f32 1=u32 2=i32 3=u64), u64 count, payload.
A C++ reader/writer of the same layout lives with the test tooling.
"""
import struct

import numpy as np

_DTYPES = {0: np.float32, 1: np.uint32, 2: np.int32, 3: np.uint64}
_CODES = {np.dtype(np.float32): 0, np.dtype(np.uint32): 1, np.dtype(np.int32): 2, np.dtype(np.uint64): 3}


def load(path):
    """Read a blob file into an ordered dict name -> 1-D numpy array."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"PVB1":
        raise ValueError("%s: not a PVB1 blob" % path)
    (n,) = struct.unpack_from("<I", data, 4)
    off = 8
    for _ in range(n):
        name = data[off:off + 48].split(b"\0", 1)[0].decode()
        dtype, count = struct.unpack_from("<IQ", data, off + 48)
        off += 60
        dt = np.dtype(_DTYPES[dtype])
        nbytes = dt.itemsize * count
        out[name] = np.frombuffer(data, dtype=dt, count=count, offset=off).copy()
        off += nbytes
    return out


def save(path, arrays):
    """Write a dict name -> array (float32 / uint32 / int32 / uint64) as a blob file."""
    with open(path, "wb") as f:
        f.write(b"PVB1")
        f.write(struct.pack("<I", len(arrays)))
        for name, arr in arrays.items():
            a = np.ascontiguousarray(arr).reshape(-1)
            if a.dtype not in _CODES:
                raise TypeError("%s: unsupported dtype %s" % (name, a.dtype))
            nm = name.encode()
            if len(nm) > 47:
                raise ValueError("name too long: %s" % name)
            f.write(nm.ljust(48, b"\0"))
            f.write(struct.pack("<IQ", _CODES[a.dtype], a.size))
            f.write(a.tobytes())
