"""Torch-free reader of PyTorch checkpoints (`.pth` written by `torch.save` of a `state_dict`) -- the ingestion point of the
reference's `load_torch_weights` (utils.py:159-171 calls `torch.load(filepath, map_location="cpu")`) without needing the
`torch` package on the box that serves the model.

Format (torch >= 1.6, zip container): `<name>/data.pkl` is a protocol-2 pickle whose tensors are
`torch._utils._rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad, backward_hooks)` calls and whose storages
are persistent ids `('storage', <storage class>, key, location, numel)`; the raw little-endian bytes of storage `key` are the zip
member `<name>/data/<key>`.  Only what a state_dict needs is interpreted: any other global raises `pickle.UnpicklingError`, so a
checkpoint cannot run code here.  The legacy (non-zip) format is refused with a clear message.
"""
from __future__ import annotations

import collections
import pickle
import zipfile
from typing import Dict

import numpy as np

_DTYPES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16, "LongStorage": np.int64,
    "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8, "ByteStorage": np.uint8, "BoolStorage": np.bool_,
    "BFloat16Storage": "bf16",
}


class _StorageType:
    def __init__(self, name):
        self.name = name


class _Storage:
    __slots__ = ("dtype", "raw")

    def __init__(self, dtype, raw):
        self.dtype, self.raw = dtype, raw


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    if storage.dtype == "bf16":                              # widen bf16 bits to fp32 (exact)
        base = (np.frombuffer(storage.raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
    else:
        base = np.frombuffer(storage.raw, dtype=np.dtype(storage.dtype).newbyteorder("<"))
    size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
    storage_offset = int(storage_offset)
    # the view must stay inside the storage: offset, sizes and strides come from the (untrusted) pickle, and as_strided checks nothing
    if len(size) != len(stride) or storage_offset < 0 or any(v < 0 for v in size) or any(v < 0 for v in stride):
        raise pickle.UnpicklingError(f"tensor view with offset {storage_offset}, size {size}, stride {stride} is not a plain forward view")
    if all(v > 0 for v in size):
        last = storage_offset + sum((n - 1) * st for n, st in zip(size, stride))
        if last >= base.size:
            raise pickle.UnpicklingError(f"tensor view (offset {storage_offset}, size {size}, stride {stride}) reaches element {last} of a "
                                         f"{base.size}-element storage")
    else:
        return np.zeros(size, dtype=base.dtype)
    if len(size) == 0:
        return np.array(base[storage_offset])
    item = base.itemsize
    view = np.lib.stride_tricks.as_strided(base[storage_offset:], shape=size, strides=tuple(s * item for s in stride), writeable=False)
    return np.ascontiguousarray(view)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, read_storage):
        super().__init__(f)
        self._read = read_storage

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _DTYPES:
            return _StorageType(name)
        if module == "torch" and name == "Size":
            return tuple
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}: only plain state_dicts of tensors are read")

    def persistent_load(self, pid):
        if not isinstance(pid, tuple) or pid[0] != "storage":
            raise pickle.UnpicklingError(f"unknown persistent id {pid!r}")
        _, stype, key, _location, _numel = pid[:5]
        name = stype.name if isinstance(stype, _StorageType) else getattr(stype, "__name__", str(stype))
        if name not in _DTYPES:
            raise pickle.UnpicklingError(f"unsupported storage type {name}")
        return _Storage(_DTYPES[name], self._read(str(key)))


def load_state_dict(path: str) -> "Dict[str, np.ndarray]":
    """Ordered `{name: numpy array}` of a `torch.save`d state_dict, without importing torch."""
    if not zipfile.is_zipfile(path):
        raise ValueError(f"{path}: not a zip-format PyTorch checkpoint (legacy torch < 1.6 files: re-save them with a current torch)")
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if not pkl:
            raise ValueError(f"{path}: no data.pkl in the archive")
        root = pkl[0][: -len("data.pkl")]
        cache = {}

        def read(key):
            if key not in cache:
                cache[key] = z.read(f"{root}data/{key}")
            return cache[key]

        with z.open(pkl[0]) as f:
            obj = _Unpickler(f, read).load()
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: the checkpoint is a {type(obj).__name__}, expected a state_dict")
    out = collections.OrderedDict()
    for k, v in obj.items():
        if not isinstance(v, np.ndarray):
            raise ValueError(f"{path}: entry {k!r} is a {type(v).__name__}, expected a tensor")
        out[k] = v
    return out
