"""Writes a checkpoint directory byte for byte the way the REFERENCE's environment wrote it -- TEST INFRASTRUCTURE (see
rsis_oracle.py header): python 2.7 + torch 0.2 `save_checkpoint` (reference src/utils/utils.py:89-95), from tensors this
container can produce.  No python 2 and no old torch exist here, so the streams are emitted opcode by opcode:

* `encoder.pt` / `decoder.pt` / `*_opt.pt`: torch's pre-zip container (torch/serialization.py of 0.2: pickled MAGIC_NUMBER,
  PROTOCOL_VERSION, sys_info, the object with storages as persistent ids, the sorted storage keys, then per storage an int64 count +
  raw elements), every pickle protocol 2 as python 2's cPickle writes it: `str` as SHORT_BINSTRING / BINSTRING (NOT unicode),
  every container memoised with BINPUT / LONG_BINPUT, OrderedDict as REDUCE(collections.OrderedDict, ([[k, v], ...],)), tensors as
  NEWOBJ(torch.cuda.FloatTensor) + BUILD((storage, offset, size tuple, stride tuple)) -- `_TensorBase.__getstate__` of that era --
  `module.`-prefixed keys when `parallel` (nn.DataParallel, train.py:269-274), no `num_batches_tracked` (torch < 0.4.1).
* `args.pkl`: `pickle.dump(args, open(..., 'wb'))` of python 2 = PROTOCOL 0 (text opcodes: copy_reg._reconstructor, S'..' strings,
  I01 booleans, numpy.float64 through numpy.core.multiarray.scalar with an escaped S'' payload).

What pins it: `tests/test_checkpoint.py` disassembles the streams with pickletools (opcode set == the python-2 set, no BINUNICODE
anywhere), loads them with STOCK `torch.load` (torch's own legacy reader accepts the container) and with the product's reader.
"""
import os
import struct

import numpy as np

MAGIC_NUMBER = 0x1950A86A20F9469CFC6C
PROTOCOL_VERSION = 1001


class Py2Pickle2(object):
    """protocol-2 emitter with python 2's choices (cPickle.dump(obj, f, 2))"""

    def __init__(self):
        self.b = bytearray(b"\x80\x02")
        self.n = 0
        self.memo = {}

    def put(self, key=None):
        i = self.n
        self.n += 1
        self.b += (b"q" + bytes([i])) if i < 256 else (b"r" + struct.pack("<I", i))
        if key is not None:
            self.memo[key] = i
        return i

    def get(self, i):
        self.b += (b"h" + bytes([i])) if i < 256 else (b"j" + struct.pack("<I", i))

    def str_(self, s):
        key = ("s", s)
        if key in self.memo:                 # (python 2 interns / memoises identical str objects of a dict: second use is a BINGET)
            return self.get(self.memo[key])
        raw = s.encode("latin1")
        self.b += (b"U" + bytes([len(raw)]) + raw) if len(raw) < 256 else (b"T" + struct.pack("<i", len(raw)) + raw)
        self.put(key)

    def int_(self, v):
        if isinstance(v, bool):
            self.b += b"\x88" if v else b"\x89"
        elif 0 <= v < 256:
            self.b += b"K" + bytes([v])
        elif 0 <= v < 65536:
            self.b += b"M" + struct.pack("<H", v)
        elif -2 ** 31 <= v < 2 ** 31:
            self.b += b"J" + struct.pack("<i", v)
        else:                                # python 2 `long`: LONG1
            raw = v.to_bytes((v.bit_length() + 8) // 8, "little", signed=True)
            self.b += b"\x8a" + bytes([len(raw)]) + raw

    def float_(self, v):
        self.b += b"G" + struct.pack(">d", v)

    def none(self):
        self.b += b"N"

    def glob(self, module, name):
        key = ("g", module, name)
        if key in self.memo:
            return self.get(self.memo[key])
        self.b += b"c" + module.encode() + b"\n" + name.encode() + b"\n"
        self.put(key)

    def tuple_(self, items, emit):
        if len(items) == 0:
            self.b += b")"
            return
        if len(items) > 3:
            self.b += b"("
        for it in items:
            emit(it)
        self.b += {1: b"\x85", 2: b"\x86", 3: b"\x87"}.get(len(items), b"t")
        self.put()

    def value(self, v):
        """plain python-2 values: None / bool / int / float / str / tuple / list / dict"""
        if v is None:
            self.none()
        elif isinstance(v, (bool, int, np.integer)):
            self.int_(v if isinstance(v, bool) else int(v))
        elif isinstance(v, float):
            self.float_(v)
        elif isinstance(v, str):
            self.str_(v)
        elif isinstance(v, tuple):
            self.tuple_(v, self.value)
        elif isinstance(v, list):
            self.b += b"]"
            self.put()
            if v:
                self.b += b"("
                for it in v:
                    self.value(it)
                self.b += b"e"
        elif isinstance(v, dict):
            self.b += b"}"
            self.put()
            if v:
                self.b += b"("
                for k, it in v.items():
                    self.value(k)
                    self.value(it)
                self.b += b"u"
        elif isinstance(v, TensorRef):
            v.emit(self)
        else:
            raise TypeError(type(v))

    def done(self):
        return bytes(self.b + b".")


class TensorRef(object):
    """one tensor of the object being saved (numpy array, C-contiguous) and the storage it owns"""

    def __init__(self, arr, key, cuda=True):
        self.arr, self.key, self.cuda = np.ascontiguousarray(arr), str(key), cuda

    def emit(self, p):
        a = self.arr
        names = {np.dtype(np.float32): "Float", np.dtype(np.float64): "Double", np.dtype(np.int64): "Long"}[a.dtype]
        mod = "torch.cuda" if self.cuda else "torch"
        p.glob(mod, names + "Tensor")
        p.b += b")\x81"                                   # EMPTY_TUPLE NEWOBJ
        p.put()
        p.b += b"("                                       # state tuple: 4 items
        p.b += b"("                                       # persistent id: 6 items
        p.str_("storage")
        p.glob(mod, names + "Storage")
        p.str_(self.key)
        p.str_("cuda:0" if self.cuda else "cpu")
        p.int_(a.size)
        p.none()                                          # view_metadata: the tensor owns its whole storage
        p.b += b"t"
        p.put()
        p.b += b"Q"                                       # BINPERSID
        p.int_(0)                                         # storage_offset
        size = tuple(int(v) for v in a.shape)
        stride = tuple(int(s // a.itemsize) for s in a.strides)
        p.tuple_(size, p.int_)
        p.tuple_(stride, p.int_)
        p.b += b"t"
        p.put()
        p.b += b"b"                                       # BUILD -> __setstate__


def _py2_dumps2(v):
    p = Py2Pickle2()
    p.value(v)
    return p.done()


def torch02_save(obj, path, cuda=True):
    """obj: OrderedDict[str, ndarray] (a state_dict) or a dict / list tree with ndarrays as tensors (an optimizer state_dict)"""
    refs = []

    def wrap(v):
        if isinstance(v, np.ndarray):
            r = TensorRef(v, 94000000000000 + 4096 * len(refs), cuda)      # root_key = str(storage._cdata): a heap address
            refs.append(r)
            return r
        if isinstance(v, dict) and type(v) is dict:
            return {k: wrap(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(wrap(x) for x in v)
        return v

    p = Py2Pickle2()
    if hasattr(obj, "items") and type(obj) is not dict:      # OrderedDict.__reduce__ of python 2: (cls, ([[k, v], ...],))
        p.glob("collections", "OrderedDict")
        p.b += b"]"
        p.put()
        items = list(obj.items())
        if items:
            p.b += b"("
            for k, v in items:
                p.b += b"]"
                p.put()
                p.b += b"("
                p.str_(k)
                wrap(v).emit(p)
                p.b += b"e"
            p.b += b"e"
        p.b += b"\x85"
        p.put()
        p.b += b"R"
        p.put()
    else:
        p.value(wrap(obj))
    body = p.done()
    sys_info = {"protocol_version": PROTOCOL_VERSION, "little_endian": True, "type_sizes": {"short": 2, "int": 4, "long": 4}}
    keys = sorted(r.key for r in refs)
    by_key = {r.key: r for r in refs}
    with open(path, "wb") as f:
        f.write(_py2_dumps2(MAGIC_NUMBER))
        f.write(_py2_dumps2(PROTOCOL_VERSION))
        f.write(_py2_dumps2(sys_info))
        f.write(body)
        f.write(_py2_dumps2(keys))
        for k in keys:                                       # THPStorage_(writeFileRaw): int64 count, then the elements
            a = by_key[k].arr
            f.write(struct.pack("<q", a.size))
            f.write(a.tobytes())


def _repr_py2_str(raw):
    """python 2's repr() of a str, as protocol 0 writes S'...' payloads"""
    out = []
    for c in raw:
        if c in (0x27, 0x5C):
            out.append("\\" + chr(c))
        elif c == 0x0A:
            out.append("\\n")
        elif c == 0x0D:
            out.append("\\r")
        elif c == 0x09:
            out.append("\\t")
        elif 32 <= c < 127:
            out.append(chr(c))
        else:
            out.append("\\x%02x" % c)
    return "'" + "".join(out) + "'"


def py2_args_pickle(ns_dict):
    """`pickle.dump(Namespace(**ns_dict), f)` of python 2.7: protocol 0.  Values: None / bool / int / float / str / list / numpy.float64"""
    out, memo = [], [0]

    def put():
        out.append("p%d\n" % memo[0])
        memo[0] += 1

    def val(v):
        if v is None:
            out.append("N")
        elif isinstance(v, bool):
            out.append("I01\n" if v else "I00\n")
        elif isinstance(v, np.float64):
            out.append("cnumpy.core.multiarray\nscalar\n")
            put()
            out.append("(cnumpy\ndtype\n")
            put()
            out.append("(S'f8'\n")
            put()
            out.append("I0\nI1\ntRp%d\n" % memo[0])
            memo[0] += 1
            out.append("(I3\nS'<'\n")
            put()
            out.append("NNNI-1\nI-1\nI0\ntbS%s\n" % _repr_py2_str(struct.pack("<d", float(v))))
            put()
            out.append("tR")
            put()
        elif isinstance(v, int):
            out.append("I%d\n" % v)
        elif isinstance(v, float):
            out.append("F%s\n" % repr(v))
        elif isinstance(v, str):
            out.append("S%s\n" % _repr_py2_str(v.encode("latin1")))
            put()
        elif isinstance(v, list):
            out.append("(l")
            put()
            for it in v:
                val(it)
                out.append("a")
        else:
            raise TypeError(type(v))

    out.append("ccopy_reg\n_reconstructor\n")
    put()
    out.append("(cargparse\nNamespace\n")
    put()
    out.append("c__builtin__\nobject\n")
    put()
    out.append("Ntp%d\nRp%d\n" % (memo[0], memo[0] + 1))
    memo[0] += 2
    out.append("(dp%d\n" % memo[0])
    memo[0] += 1
    for k, v in ns_dict.items():
        out.append("S%s\n" % _repr_py2_str(k.encode("latin1")))
        put()
        val(v)
        out.append("s")
    out.append("b.")
    return "".join(out).encode("latin1")


def write_reference_checkpoint(root, model_name, enc_sd, dec_sd, args_dict, parallel=True, cuda=True, enc_opt=None, dec_opt=None):
    """the five files of reference save_checkpoint (utils/utils.py:89-95) under root/model_name, as python 2 + torch 0.2 wrote them.
    enc_sd / dec_sd: mapping key -> torch tensor or ndarray in the REFERENCE layout (modern `num_batches_tracked` entries are dropped)."""
    from collections import OrderedDict
    d = os.path.join(root, model_name)
    os.makedirs(d, exist_ok=True)

    def conv(sd):
        out = OrderedDict()
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            out[("module." if parallel else "") + k] = a
        return out

    torch02_save(conv(enc_sd), os.path.join(d, "encoder.pt"), cuda)
    torch02_save(conv(dec_sd), os.path.join(d, "decoder.pt"), cuda)
    empty = {"state": {}, "param_groups": []}
    torch02_save(enc_opt if enc_opt is not None else empty, os.path.join(d, "enc_opt.pt"), cuda)
    torch02_save(dec_opt if dec_opt is not None else empty, os.path.join(d, "dec_opt.pt"), cuda)
    with open(os.path.join(d, "args.pkl"), "wb") as f:
        f.write(py2_args_pickle(args_dict))
    return d
