"""MXNet ``nd.save`` list container (the ``.params`` weight files of the reference).

The reference loads its generator from ``stylegan-{gan}.params`` (reference
image_generator.py:21-22) and its decoder from ``checkpoints/*.params``
(reference seg_solver.py:331-349).  Both are plain MXNet NDArray-list files; the
byte layout restated here is SURVEY.md Appendix C (MXNet 1.5 ``NDArray::Save``):

    u64 0x112, u64 0, u64 n_arrays,
    n_arrays x { u32 magic(V2 0xF993FAC9 | V3 0xF993FACA), i32 stype(0 dense),
                 u32 ndim, i64 dims[ndim], i32 dev_type, i32 dev_id,
                 i32 type_flag, raw C-order data },
    u64 n_names, n_names x { u64 len, bytes }

MXNet is not available in this environment, so this reader/writer pair is pinned
by a hand-assembled byte fixture (tests/golden/handmade.params) and round trips.
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
NDARRAY_V1_MAGIC = 0xF993FAC8
NDARRAY_V2_MAGIC = 0xF993FAC9
NDARRAY_V3_MAGIC = 0xF993FACA

# mshadow type flags
_TYPE_FLAGS = {
    0: np.dtype("<f4"), 1: np.dtype("<f8"), 2: np.dtype("<f2"), 3: np.dtype("u1"),
    4: np.dtype("<i4"), 5: np.dtype("i1"), 6: np.dtype("<i8"),
}
_FLAG_OF = {v: k for k, v in _TYPE_FLAGS.items()}


class ParamsFormatError(ValueError):
    pass


def _strip_prefix(name):
    # Module-style checkpoints prefix keys with "arg:" / "aux:".
    for p in ("arg:", "aux:"):
        if name.startswith(p):
            return name[len(p):]
    return name


def load_params(path):
    """Read a ``.params`` file -> ``dict[str, np.ndarray]`` in file order."""
    with open(path, "rb") as f:
        buf = f.read()
    return loads_params(buf)


def loads_params(buf):
    mv = memoryview(buf)
    pos = 0

    def take(fmt):
        nonlocal pos
        size = struct.calcsize(fmt)
        if pos + size > len(mv):
            raise ParamsFormatError("truncated .params file at byte %d" % pos)
        vals = struct.unpack_from(fmt, mv, pos)
        pos += size
        return vals

    magic, _reserved = take("<QQ")
    if magic != LIST_MAGIC:
        raise ParamsFormatError("bad list magic 0x%x (expected 0x112)" % magic)
    (n_arrays,) = take("<Q")
    arrays = []
    for _ in range(n_arrays):
        (nd_magic,) = take("<I")
        if nd_magic in (NDARRAY_V2_MAGIC, NDARRAY_V3_MAGIC):
            (stype,) = take("<i")
            if stype != 0:
                raise ParamsFormatError("sparse storage type %d is not supported" % stype)
            (ndim,) = take("<I")
            dims = take("<%dq" % ndim) if ndim else ()
        elif nd_magic == NDARRAY_V1_MAGIC:
            (ndim,) = take("<I")
            dims = take("<%dq" % ndim) if ndim else ()
        else:
            # legacy (pre-V1) arrays: the u32 just read is ndim, dims are u32
            ndim = nd_magic
            if ndim > 32:
                raise ParamsFormatError("bad ndarray magic 0x%x" % nd_magic)
            dims = take("<%dI" % ndim) if ndim else ()
        if ndim == 0:
            arrays.append(np.zeros((), np.float32))
            continue
        _dev_type, _dev_id = take("<ii")
        (type_flag,) = take("<i")
        if type_flag not in _TYPE_FLAGS:
            raise ParamsFormatError("unknown type flag %d" % type_flag)
        dt = _TYPE_FLAGS[type_flag]
        count = int(np.prod(dims, dtype=np.int64))
        nbytes = count * dt.itemsize
        if pos + nbytes > len(mv):
            raise ParamsFormatError("truncated tensor data at byte %d" % pos)
        arr = np.frombuffer(mv, dtype=dt, count=count, offset=pos).reshape(dims).copy()
        pos += nbytes
        arrays.append(arr)
    (n_names,) = take("<Q")
    if n_names not in (0, n_arrays):
        raise ParamsFormatError("%d names for %d arrays" % (n_names, n_arrays))
    names = []
    for _ in range(n_names):
        (ln,) = take("<Q")
        if pos + ln > len(mv):
            raise ParamsFormatError("truncated name at byte %d" % pos)
        names.append(_strip_prefix(bytes(mv[pos:pos + ln]).decode("utf-8")))
        pos += ln
    if not names:
        names = [str(i) for i in range(n_arrays)]
    return dict(zip(names, arrays))


def dumps_params(tensors, magic=NDARRAY_V2_MAGIC):
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(tensors))]
    for arr in tensors.values():
        arr = np.ascontiguousarray(arr)
        dt = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
        if np.dtype(dt) not in _FLAG_OF:
            raise ParamsFormatError("dtype %s has no MXNet type flag" % arr.dtype)
        out.append(struct.pack("<IiI", magic, 0, arr.ndim))
        out.append(struct.pack("<%dq" % arr.ndim, *arr.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAG_OF[np.dtype(dt)]))
        out.append(arr.astype(dt, copy=False).tobytes())
    out.append(struct.pack("<Q", len(tensors)))
    for name in tensors:
        b = name.encode("utf-8")
        out.append(struct.pack("<Q", len(b)))
        out.append(b)
    return b"".join(out)


def save_params(path, tensors, magic=NDARRAY_V2_MAGIC):
    """Write ``dict[str, np.ndarray]`` as an MXNet NDArray-list file."""
    with open(path, "wb") as f:
        f.write(dumps_params(tensors, magic))
