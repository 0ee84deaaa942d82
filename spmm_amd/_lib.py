"""ctypes binding of libspmm_hip.so.  Signatures are parsed from include/spmm_hip.h so the
header stays the single source of truth for the C ABI.  There is NO fallback: if the shared
library is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "spmm_hip.h")
LIB_PATH = os.environ.get("SPMM_HIP_LIB") or os.path.join(_HERE, "libspmm_hip.so")      # (override: instrumented builds of tools/)

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "uint64_t": ctypes.c_uint64,
       "spmm_stream_t": ctypes.c_void_p, "void": None}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.startswith("const char") else ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _CT[base]


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every `spmm_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(spmm_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = [] if args in ("void", "") else [_ctype(re.sub(r"\s*\w+$", "", a.strip()) if "*" not in a else a)
                                                    for a in args.split(",")]
        protos[name] = (_ctype(ret), argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C spmm_amd/csrc`).  spmm_amd has no CPU / eager fallback.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        self.cdll.spmm_last_error.restype = ctypes.c_char_p

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed (rc={rc}): {self.cdll.spmm_last_error().decode()}")


_lib = None
_device = None


def bind_device(index: int) -> None:
    """One GPU per process (the launch model of the package: one process per GPU under torch.distributed).  The kernels' dynamic-LDS
    opt-ins (hipFuncSetAttribute) are made once per process, on the device current at that moment; a second device in the same process
    would launch the 128-KiB GEMM and the large attention kernels without them.  Called by every Engine; raises on a second device."""
    global _device
    if _device is None:
        _device = int(index)
    elif _device != int(index):
        raise RuntimeError(f"spmm_amd: this process already runs on cuda:{_device}; a second device (cuda:{index}) needs its own process "
                           "(kernel attributes are set once per process, spmm_amd/_lib.py::bind_device)")


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
