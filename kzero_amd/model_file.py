"""KZM1 model container: the hand-off format between the trainer side and the HIP executor.

The reference hands the self-play server an ONNX *path* (`Command::NewNetwork`,
rust/kz-selfplay/src/server/protocol.rs:36) which `load_graph` parses and
optimizes (rust/kz-selfplay/src/server/server_alphazero.rs:126-128).  Until the
C++ ONNX reader lands the engine accepts this container
instead.  It stores the *unfolded* PyTorch `state_dict` tensors of
`PredictionHeads(ResTower, ScalarHead, <policy head>)`
(python/lib/model/post_act.py:187-211) under their state_dict names plus an
architecture descriptor; Conv+BN folding happens in the C++ loader, like
`optimize_graph` does in the reference.

Layout (little endian):
    char  magic[8] = "KZMODEL1"
    u32   n_meta
    n_meta x { u16 key_len; key bytes; u8 kind; value }
        kind 0: i64   kind 1: f64   kind 2: u32 len + utf-8 bytes
    u32   n_tensors
    n_tensors x { u16 name_len; name; u8 dtype (0 = f32, 1 = i64); u32 ndim;
                  u64 dims[ndim]; u64 byte_offset; u64 byte_len }
    u64   data_len
    data  (byte_offset is relative to the start of data; every tensor 64-byte aligned)
"""
import struct
from typing import Dict, Tuple, Union

import numpy as np

MAGIC = b"KZMODEL1"

MetaValue = Union[int, float, str]


def write_model(meta: Dict[str, MetaValue], tensors: Dict[str, np.ndarray]) -> bytes:
    out = bytearray()
    out += MAGIC
    out += struct.pack("<I", len(meta))
    for key, value in meta.items():
        kb = key.encode()
        out += struct.pack("<H", len(kb)) + kb
        if isinstance(value, bool):
            value = int(value)
        if isinstance(value, (int, np.integer)):
            out += struct.pack("<Bq", 0, int(value))
        elif isinstance(value, (float, np.floating)):
            out += struct.pack("<Bd", 1, float(value))
        elif isinstance(value, str):
            vb = value.encode()
            out += struct.pack("<BI", 2, len(vb)) + vb
        else:
            raise TypeError(f"unsupported meta value for '{key}': {type(value)}")

    data = bytearray()
    out += struct.pack("<I", len(tensors))
    for name, arr in tensors.items():
        if arr.dtype == np.float32:
            dtype = 0
        elif arr.dtype == np.int64:
            dtype = 1
        else:
            raise TypeError(f"tensor '{name}' has unsupported dtype {arr.dtype}")
        arr = np.ascontiguousarray(arr)
        while len(data) % 64:
            data.append(0)
        offset = len(data)
        raw = arr.tobytes()
        data += raw
        nb = name.encode()
        out += struct.pack("<H", len(nb)) + nb
        out += struct.pack("<BI", dtype, arr.ndim)
        for d in arr.shape:
            out += struct.pack("<Q", d)
        out += struct.pack("<QQ", offset, len(raw))
    out += struct.pack("<Q", len(data))
    out += data
    return bytes(out)


def read_model(blob: bytes) -> Tuple[Dict[str, MetaValue], Dict[str, np.ndarray]]:
    assert blob[:8] == MAGIC, "not a KZMODEL1 container"
    pos = 8

    def take(fmt):
        nonlocal pos
        vals = struct.unpack_from(fmt, blob, pos)
        pos += struct.calcsize(fmt)
        return vals

    def take_bytes(n):
        nonlocal pos
        b = blob[pos:pos + n]
        pos += n
        return b

    meta: Dict[str, MetaValue] = {}
    (n_meta,) = take("<I")
    for _ in range(n_meta):
        (klen,) = take("<H")
        key = take_bytes(klen).decode()
        (kind,) = take("<B")
        if kind == 0:
            (meta[key],) = take("<q")
        elif kind == 1:
            (meta[key],) = take("<d")
        elif kind == 2:
            (vlen,) = take("<I")
            meta[key] = take_bytes(vlen).decode()
        else:
            raise ValueError(f"bad meta kind {kind}")

    (n_tensors,) = take("<I")
    entries = []
    for _ in range(n_tensors):
        (nlen,) = take("<H")
        name = take_bytes(nlen).decode()
        dtype, ndim = take("<BI")
        dims = [take("<Q")[0] for _ in range(ndim)]
        offset, nbytes = take("<QQ")
        entries.append((name, dtype, dims, offset, nbytes))
    (data_len,) = take("<Q")
    data = blob[pos:pos + data_len]
    assert len(data) == data_len, "truncated container"

    tensors = {}
    for name, dtype, dims, offset, nbytes in entries:
        np_dtype = np.float32 if dtype == 0 else np.int64
        tensors[name] = np.frombuffer(data, dtype=np_dtype, count=nbytes // np.dtype(np_dtype).itemsize,
                                      offset=offset).reshape(dims)
    return meta, tensors
