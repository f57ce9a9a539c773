"""Reader for the golden-fixture container written by oracle/blobio.h (format "SMGF1")."""
import struct

import numpy as np

_DT = {b"f": np.float32, b"d": np.float64, b"q": np.int64, b"i": np.int32, b"B": np.uint8,
       b"I": np.uint32}


def load_blob(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    assert data[:6] == b"SMGF1\n", "bad magic in %s" % path
    p = 6
    while p < len(data):
        (nl,) = struct.unpack_from("<I", data, p); p += 4
        name = data[p:p + nl].decode(); p += nl
        dt = data[p:p + 1]; p += 1
        (nd,) = struct.unpack_from("<I", data, p); p += 4
        dims = struct.unpack_from("<%dq" % nd, data, p); p += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        dtype = np.dtype(_DT[dt])
        arr = np.frombuffer(data, dtype=dtype, count=n, offset=p).reshape(dims).copy()
        p += n * dtype.itemsize
        out[name] = arr
    return out
