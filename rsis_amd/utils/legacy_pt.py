"""Reader for the checkpoint files the reference wrote (reference src/utils/utils.py:89-95 `torch.save(state_dict)`
under torch 0.2 / python 2) -- torch's pre-zip "magic number" format:

    pickle(MAGIC_NUMBER) pickle(PROTOCOL_VERSION) pickle(sys_info) pickle(obj; storages as persistent ids)
    pickle(sorted storage keys) then per key: int64 element count + the raw elements

with `obj` a protocol-2 python-2 pickle: py2 `str` keys (SHORT_BINSTRING / BINSTRING opcodes: bytes that python 3 must decode),
tensors as `NEWOBJ(torch.FloatTensor | torch.cuda.FloatTensor)` + `BUILD((storage, offset, size, stride))` (torch <= 0.3; torch 0.4-1.5
wrote `torch._utils._rebuild_tensor[_v2]` REDUCE calls into the same container -- read too), an OrderedDict of them for a module,
plain dict / list / tuple / number / str for an optimizer.

Why not `torch.load`: `weights_only=True` cannot instantiate `torch.cuda.FloatTensor` on a host without that device and refuses the
format's older corners, `weights_only=False` executes whatever the file names.  This reader resolves a CLOSED list of globals to inert
stand-ins, never imports or calls anything the stream names, and builds the tensors itself from the raw storage bytes (always on the
CPU: the caller moves them; that is what the reference's `map_location=lambda storage, location: storage` did).
"""
import io
import pickle
import struct
from collections import OrderedDict

import numpy as np
import torch

MAGIC_NUMBER = 0x1950A86A20F9469CFC6C
PROTOCOL_VERSION = 1001

_DTYPES = {"FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16, "LongStorage": np.int64,
           "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8, "ByteStorage": np.uint8,
           # storages torch gained after the reference's 0.2 but still writes into the pre-zip container (a bf16 checkpoint of this
           # build saved with _use_new_zipfile_serialization=False): bf16 is read as 16-bit words and re-viewed, bool as bytes
           "BFloat16Storage": np.int16, "BoolStorage": np.bool_}
_VIEW_AS = {"BFloat16Storage": torch.bfloat16}


def is_legacy_file(path):
    with open(path, "rb") as f:
        head = f.read(4)
    return head[:2] == b"\x80\x02" and head[2:3] == b"\x8a"      # PROTO 2, LONG1: the pickled magic number (a zip file starts with PK)


class _StorageType(object):
    def __init__(self, name):
        self.name = name


class _StorageRef(object):
    """a persistent id ('storage', type, root_key, location, numel, view_metadata) until the raw bytes are read"""

    def __init__(self, stype, key, location, numel, view):
        if not isinstance(stype, _StorageType) or stype.name not in _DTYPES:
            raise pickle.UnpicklingError("legacy checkpoint: unknown storage type %r" % (stype,))
        self.dtype, self.key, self.location, self.numel = np.dtype(_DTYPES[stype.name]), str(key), location, int(numel)
        self.view_as = _VIEW_AS.get(stype.name)
        self.offset = 0
        if view is not None:             # (view_key, offset, view_size): a storage that is a window of a larger root storage
            self.offset = int(view[1])
        self.data = None


class _LegacyTensor(object):
    """stands in for torch.FloatTensor & co.: NEWOBJ makes an empty one, BUILD hands over (storage, offset, size, stride)"""

    def __init__(self, *args):
        self.state = tuple(args) if args else None

    def __setstate__(self, state):
        self.state = tuple(state)


def _rebuild_tensor(storage, storage_offset, size, stride, *_ignored):
    """torch._utils._rebuild_tensor / _rebuild_tensor_v2 (torch 0.4-1.5 files): same four leading arguments"""
    t = _LegacyTensor()
    t.state = (storage, storage_offset, tuple(size), tuple(stride))
    return t


def _rebuild_parameter(data, requires_grad, _hooks):
    return data


def _ordered_dict(*args):
    return OrderedDict(*args)


_TENSOR_NAMES = {"FloatTensor", "DoubleTensor", "HalfTensor", "LongTensor", "IntTensor", "ShortTensor", "CharTensor", "ByteTensor",
                 "BFloat16Tensor", "BoolTensor"}


class _Unpickler(pickle.Unpickler):
    def __init__(self, f):
        super().__init__(f, encoding="latin1")       # py2 str payloads: bytes 0-255 map 1:1 (keys are ASCII; nothing is lost)
        self.refs = []

    def find_class(self, module, name):
        if module in ("torch", "torch.cuda"):
            if name in _TENSOR_NAMES:
                return _LegacyTensor
            if name in _DTYPES:
                return _StorageType(name)
            if module == "torch" and name == "Size":
                return tuple
        if (module, name) in (("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2")):
            return _rebuild_tensor
        if (module, name) == ("torch._utils", "_rebuild_parameter"):
            return _rebuild_parameter
        if (module, name) == ("collections", "OrderedDict"):
            return _ordered_dict
        if module in ("__builtin__", "builtins") and name in ("dict", "list", "tuple", "set", "long", "int", "float", "bool", "str", "unicode"):
            return {"long": int, "unicode": str}.get(name) or getattr(__import__("builtins"), name)
        raise pickle.UnpicklingError("legacy checkpoint: refusing to resolve %s.%s" % (module, name))

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) == 6 and pid[0] == "storage"):
            raise pickle.UnpicklingError("legacy checkpoint: unexpected persistent id %r" % (pid[:1] if isinstance(pid, tuple) else pid,))
        ref = _StorageRef(*pid[1:])
        self.refs.append(ref)
        return ref


def _plain_load(f):
    class U(pickle.Unpickler):
        def find_class(self, module, name):
            raise pickle.UnpicklingError("legacy checkpoint header: refusing to resolve %s.%s" % (module, name))
    return U(f, encoding="latin1").load()


def _materialise(obj, roots):
    if isinstance(obj, _LegacyTensor):
        state = getattr(obj, "state", None)     # (NEWOBJ does not run __init__: an empty tensor has no state at all)
        if state is None:
            return torch.empty(0)
        ref, off, size, stride = state[:4]
        if not isinstance(ref, _StorageRef):
            raise pickle.UnpicklingError("legacy checkpoint: tensor without a storage")
        base = roots[ref.key]
        if base.dtype != ref.dtype:
            raise pickle.UnpicklingError("legacy checkpoint: storage %s read as %s, referenced as %s" % (ref.key, base.dtype, ref.dtype))
        size, stride, start = tuple(int(v) for v in size), tuple(int(v) for v in stride), ref.offset + int(off)
        span = 1 + sum((n - 1) * s for n, s in zip(size, stride)) if all(n > 0 for n in size) else 0
        if start < 0 or any(s < 0 for s in stride) or start + span > base.size:
            raise pickle.UnpicklingError("legacy checkpoint: tensor reaches outside its storage")
        t = torch.from_numpy(base)
        t = torch.as_strided(t, size, stride, start).clone() if size else t[start].clone()
        return t.view(ref.view_as) if ref.view_as is not None else t
    if isinstance(obj, OrderedDict):
        return OrderedDict((k, _materialise(v, roots)) for k, v in obj.items())
    if isinstance(obj, dict):
        return {k: _materialise(v, roots) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_materialise(v, roots) for v in obj)
    return obj


def load(path):
    """the object a reference-era `torch.save` wrote, tensors on the CPU"""
    with open(path, "rb") as f:
        if _plain_load(f) != MAGIC_NUMBER:
            raise pickle.UnpicklingError("%s: not a torch legacy checkpoint (bad magic number)" % path)
        ver = _plain_load(f)
        if ver != PROTOCOL_VERSION:
            raise pickle.UnpicklingError("%s: legacy checkpoint protocol %r (expected %d)" % (path, ver, PROTOCOL_VERSION))
        info = _plain_load(f)
        if isinstance(info, dict) and info.get("little_endian", True) is not True:
            raise pickle.UnpicklingError("%s: big-endian checkpoint" % path)
        up = _Unpickler(f)
        obj = up.load()
        keys = _plain_load(f)
        want = {}
        for r in up.refs:
            if r.key in want and want[r.key][0] != r.dtype:
                raise pickle.UnpicklingError("legacy checkpoint: storage %s referenced with two element types" % r.key)
            want.setdefault(r.key, (r.dtype, r.numel))
        roots = {}
        for k in keys:
            k = str(k)
            (n,) = struct.unpack("<q", f.read(8))
            dt = want.get(k, (np.uint8, n))[0]
            nbytes = n * np.dtype(dt).itemsize
            raw = f.read(nbytes)
            if n < 0 or len(raw) != nbytes:
                raise pickle.UnpicklingError("legacy checkpoint: storage %s is truncated" % k)
            roots[k] = np.frombuffer(raw, dtype=dt).copy()
        missing = [k for k in want if k not in roots]
        if missing:
            raise pickle.UnpicklingError("legacy checkpoint: %d referenced storages are not in the file" % len(missing))
    return _materialise(obj, roots)


def loads(raw):
    import os
    import tempfile
    fd, p = tempfile.mkstemp()
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(raw)
        return load(p)
    finally:
        os.unlink(p)
