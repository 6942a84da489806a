"""ctypes binding of libeadgan_hip.so.

Prototypes are parsed from ``include/eadgan_hip.h`` so the header is the single source of truth for the
C ABI.  There is NO fallback: if the shared library is missing or a symbol is absent this module raises,
and every op in :mod:`ops` raises ``RuntimeError`` when the native call reports an error.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "eadgan_hip.h")
LIB_PATH = os.path.join(_HERE, "libeadgan_hip.so")

EG_F32, EG_BF16, EG_F16 = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = range(5)
OUT_NHWC, OUT_NCHW_F32 = 0, 1
# eg_epilogue.nt_variant (include/eadgan_hip.h: EG_NT_*)
NT_AUTO, NT_REG, NT_BUF128, NT_PERS, NT_S8, NT_S8P, NT_S8H = range(7)
# eg_epilogue.stat_mode (EG_STAT_*): column statistics of the stored tile, fused into the convolution's epilogue
STAT_NONE, STAT_MOMENTS, STAT_BN_BWD, STAT_SN_BIAS = range(4)


class EgConv(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "H", "W", "Cin", "Cout", "k", "stride", "pad", "up")]


class EgEpilogue(ctypes.Structure):
    _fields_ = [("bias", ctypes.c_void_p), ("bias_mod", ctypes.c_int), ("sigma", ctypes.c_void_p),
                ("act", ctypes.c_int), ("slope", ctypes.c_float), ("mask", ctypes.c_void_p),
                ("mask_act", ctypes.c_int), ("mask_slope", ctypes.c_float), ("out_mode", ctypes.c_int), ("sigma_rows", ctypes.c_int),
                ("splitk_ws", ctypes.c_void_p), ("splitk_ws_bytes", ctypes.c_size_t),
                ("nt_variant", ctypes.c_int), ("nt_splitk", ctypes.c_int),
                ("stat_mode", ctypes.c_int), ("stat_out", ctypes.c_void_p), ("stat_aux", ctypes.c_void_p),
                ("stat_p0", ctypes.c_void_p), ("stat_p1", ctypes.c_void_p), ("stat_p2", ctypes.c_void_p), ("stat_p3", ctypes.c_void_p),
                ("stat_act", ctypes.c_int), ("stat_slope", ctypes.c_float)]


class EgSnLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("sigma", ctypes.c_void_p),
                ("u_snap", ctypes.c_void_p), ("v_snap", ctypes.c_void_p), ("R", ctypes.c_int), ("Kd", ctypes.c_int)]


class EgRngSeg(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("out", ctypes.c_void_p), ("n", ctypes.c_size_t), ("a", ctypes.c_float), ("b", ctypes.c_float),
                ("stream_id", ctypes.c_uint), ("onehot", ctypes.c_void_p), ("onehot_n", ctypes.c_int)]


class EgHead(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("partials", ctypes.c_void_p), ("nslice", ctypes.c_int),
                ("y", ctypes.c_void_p), ("dout", ctypes.c_void_p),
                ("dx", ctypes.c_void_p), ("sigma", ctypes.c_void_p),
                ("B", ctypes.c_int), ("T", ctypes.c_int), ("K", ctypes.c_int), ("Kpad", ctypes.c_int), ("N", ctypes.c_int), ("mode", ctypes.c_int),
                ("target", ctypes.c_float * 3), ("scale", ctypes.c_float * 3),
                ("c_cont", ctypes.c_int), ("n_cont", ctypes.c_int), ("n_cat", ctypes.c_int), ("code", ctypes.c_void_p), ("ldc", ctypes.c_int),
                ("labels", ctypes.c_void_p), ("lcat", ctypes.c_float), ("lcon", ctypes.c_float), ("laff", ctypes.c_float),
                ("loss", ctypes.c_void_p), ("terms", ctypes.c_void_p), ("counter", ctypes.c_void_p), ("mask_act", ctypes.c_int),
                ("mask_slope", ctypes.c_float)]


_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
            "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong, "unsigned int": ctypes.c_uint,
            "eg_stream_t": ctypes.c_void_p}


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p
    base = re.sub(r"\bconst\b", "", decl).strip()
    base = " ".join(base.split()[:-1]) if len(base.split()) > 1 and base.split()[-1] not in ("long", "int") else base
    for k, v in _SCALARS.items():
        if base == k:
            return v
    raise ValueError(f"unhandled C type in header: {decl!r}")


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct.*?}\s*\w+;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?m)^\s*(const char\*|size_t|int)\s+(eg_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        argtypes = [] if args in ("void", "") else [_ctype_of(a) for a in args.split(",")]
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = restype, argtypes

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed (rc={rc}): {self.cdll.eg_last_error().decode()}")

    def query(self, name, *args):
        return getattr(self.cdll, name)(*args)


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
