"""Serialise a `CompiledModel` into the flat "RGMODEL1" blob that crosses the C ABI
(`rg_model_create`, include/rgstep.h).

Layout (little endian):
    char     magic[8]  = "RGMODEL1"
    uint32   nentries
    uint32   reserved
    entry[nentries]:  char name[40]; uint32 dtype (0=f64, 1=i32, 2=f32); uint32 count; uint64 offset
    payload, each array 8-byte aligned, `offset` from the start of the blob
"""
import struct

import numpy as np

_DT = {np.dtype(np.float64): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2}


def pack_model(model) -> bytes:
    items = []
    for name, arr in sorted(model.arrays.items()):
        a = np.ascontiguousarray(arr)
        if a.dtype not in _DT:
            if np.issubdtype(a.dtype, np.integer):
                a = a.astype(np.int32)
            else:
                a = a.astype(np.float64)
        if len(name) >= 40:
            raise ValueError("array name too long: %s" % name)
        items.append((name, a))
    header = 16 + 56 * len(items)
    offset = (header + 7) // 8 * 8
    directory, payload = [], []
    for name, a in items:
        raw = a.tobytes()
        directory.append(struct.pack("<40sIIQ", name.encode(), _DT[a.dtype], a.size, offset))
        pad = (-len(raw)) % 8
        payload.append(raw + b"\0" * pad)
        offset += len(raw) + pad
    head = struct.pack("<8sII", b"RGMODEL1", len(items), 0) + b"".join(directory)
    head += b"\0" * ((-len(head)) % 8)
    return head + b"".join(payload)
