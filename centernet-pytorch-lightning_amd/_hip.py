"""ctypes binding of libcenternet_hip.so (the C ABI declared in include/centernet_hip.h).

The prototypes are parsed from the header itself, so the binding can never drift from the ABI.
There is NO fallback: if the shared library is missing the import of any op fails loudly.
PyTorch is used only as plumbing here (device memory, current stream).
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "centernet_hip.h")
LIB_PATH = os.environ.get("CN_LIB_PATH") or os.path.join(_PKG, "libcenternet_hip.so")   # CN_LIB_PATH: A/B against another build

CN_F32, CN_BF16 = 0, 1

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "void": None,
}


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every `cn_*` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\s*\*)\s+(cn_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, alist)
    return protos


def _ctype(t):
    if "*" in t:
        return ctypes.c_void_p
    return _CT[t.replace("const ", "").strip()]


def parse_struct(name, path=HEADER):
    """-> [(field, type_str), ...] of `typedef struct <name> { ... } <name>;` in declaration order"""
    src = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if decl:
            mm = re.match(r"(.*?)(\w+)$", decl)
            fields.append((mm.group(2), mm.group(1).strip()))
    return fields


class Hooks(ctypes.Structure):
    """`cn_hooks` (include/centernet_hip.h): the optional extras of ONE call — input pre-affine, BatchNorm statistics sinks of the
    output, weight-gradient grid — handed to the `_h` twin of an entry point.  The field list is read from the header, like the
    prototypes.  Host memory, read (and `bn_taken` / `bnb_taken` written) during the call only: nothing stays armed in the library."""
    _fields_ = [(f, _ctype(t)) for f, t in parse_struct("cn_hooks")]

    def set(self, **kw):
        for k, v in kw.items():
            setattr(self, k, _arg(v))
        return self


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C centernet-pytorch-lightning_amd/csrc -j8`). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (ret, args) in _protos.items():
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_char_p if "char" in ret else _CT[ret]
            fn.argtypes = [_ctype(t) for t, _ in args]
        # the ctypes mirror of cn_hooks is generated from the header; a STALE library built against an older struct would misread device
        # pointers and sink fields (GPU faults or corrupt BN statistics, not an error): refuse it at load time (round-5 ADVICE)
        if _lib.cn_hooks_size() != ctypes.sizeof(Hooks):
            n, _lib = _lib.cn_hooks_size(), None
            raise RuntimeError(f"{LIB_PATH} was built against a different cn_hooks ({n} bytes, include/centernet_hip.h says "
                               f"{ctypes.sizeof(Hooks)}): rebuild it (`make -C centernet-pytorch-lightning_amd/csrc -j8`)")
    return _lib


def dtype_code(dt):
    if dt == torch.float32:
        return CN_F32
    if dt == torch.bfloat16:
        return CN_BF16
    raise TypeError(f"unsupported activation dtype {dt}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise RuntimeError("centernet_hip: tensor argument is not on the GPU (no CPU path exists)")
        return a.data_ptr()
    return a


TRACE = False      # debugging aid: name every call on stderr before it is issued (with HIP_LAUNCH_BLOCKING=1 the last line is the culprit)


def _hooks_args(name, hooks):
    """(entry-point name, leading part of its trailing arguments): with per-call hooks (`Hooks`) the `_h` twin is called"""
    if hooks is None:
        return name, ()
    return name + "_h", (ctypes.addressof(hooks),)


def call(name, *args, hooks=None):
    """Invoke cn_<name> (its `_h` twin when `hooks` — a `Hooks` — is given); tensors become device pointers; the current torch
    stream is appended."""
    L = lib()
    name, tail = _hooks_args(name, hooks)
    fn = getattr(L, name)
    if TRACE:
        import sys
        sys.stderr.write("CALL " + name + " " + " ".join(str(tuple(a.shape)) if isinstance(a, torch.Tensor) else str(a) for a in args) + "\n")
        sys.stderr.flush()
    rc = fn(*[_arg(a) for a in args], *tail, stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {L.cn_last_error().decode()}")


def try_call(name, *args, hooks=None):
    """Like `call`, for entry points that decline shapes: -> False on CN_EUNSUPPORTED (the caller runs its general path)."""
    L = lib()
    name, tail = _hooks_args(name, hooks)
    rc = getattr(L, name)(*[_arg(a) for a in args], *tail, stream_ptr())
    if rc == -2:
        return False
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {L.cn_last_error().decode()}")
    return True


def query(name, *args):
    """size_t / int queries without a stream argument."""
    return getattr(lib(), name)(*args)


_ws = {}
_ws_retired = []     # outgrown buffers stay allocated: a captured hipGraph (or a kernel still queued) may hold their address


def workspace(nbytes, device, tag="ws"):
    """Persistent scratch buffer per (tag, device, stream): reuse is stream-ordered, so a buffer is never shared between two
    streams (the decode of `TrainStep.post_forward` runs on its own stream next to backward).  When a larger size is asked for,
    the old buffer is retired, not freed — hipGraphs captured earlier keep replaying into it."""
    key = (tag, torch.device(device).index, torch.cuda.current_stream().cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf
