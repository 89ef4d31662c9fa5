"""ctypes binding of libdig_hip.so (the C-ABI boundary declared in include/dig_hip.h).

The argument structs are generated from the header text itself, so the Python mirror cannot drift from
the C declarations.  There is NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "dig_hip.h")
LIB_PATH = os.environ.get("DIG_HIP_LIB", os.path.join(_HERE, "csrc", "libdig_hip.so"))  # override only for A/B kernel experiments

DIG_BF16, DIG_F32 = 0, 1

_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong}


def _parse_header(text):
    """-> (structs {name: [(field, ctype)]}, functions [name])"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "*" in decl:  # pointer field(s): "const void* a" (one per declaration in this header)
                fname = decl.rsplit("*", 1)[1].strip()
                fields.append((fname, ctypes.c_void_p))
                continue
            for base in ("long long", "int", "float"):
                if decl.startswith(base + " "):
                    for item in decl[len(base):].split(","):
                        item = item.strip()
                        m = re.match(r"(\w+)\[(\d+)\]$", item)
                        if m:
                            fields.append((m.group(1), _SCALARS[base] * int(m.group(2))))
                        else:
                            fields.append((item, _SCALARS[base]))
                    break
            else:
                raise RuntimeError(f"dig_hip.h: cannot parse field declaration {decl!r}")
        structs[name] = fields
    funcs = re.findall(r"^\s*(?:int|const char\*)\s+(dig_\w+)\s*\(", text, flags=re.M)
    return structs, funcs


with open(HEADER) as _f:
    _STRUCT_FIELDS, FUNCTIONS = _parse_header(_f.read())

STRUCTS = {}
for _name, _fields in _STRUCT_FIELDS.items():
    STRUCTS[_name] = type(_name, (ctypes.Structure,), {"_fields_": _fields})

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built - never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C dig_amd/csrc` (or __graft_entry__.build()). "
                "dig_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for fn in FUNCTIONS:
            f = getattr(L, fn)  # AttributeError if a declared symbol is not exported
            f.restype = ctypes.c_char_p if fn == "dig_last_error" else ctypes.c_int
        _lib = L
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()  # torch.Tensor


_STRUCT_CACHE = {}
_SNAME = {"gemm_nt": "dig_gemm_args", "qsample_stem": "dig_stem_args", "bilinear_ac_concat": "dig_bilinear_ac_args",
          "nchw_f32_to_nhwc": "dig_layout_args", "nhwc_to_nchw_f32": "dig_layout_args"}


def _struct_for(fn):
    """(ctypes struct class, {field: ctype}) of dig_<fn>'s argument block; built once per entry point (the per-launch host
    cost is what bounds small batches)."""
    hit = _STRUCT_CACHE.get(fn)
    if hit is None:
        S = STRUCTS[_SNAME.get(fn, f"dig_{fn}_args")]
        hit = _STRUCT_CACHE[fn] = (S, dict(S._fields_))
    return hit


TRACE = None  # bench.py's per-launch HIP-event hook: callable(fn, kw, phase) with phase 0 before / 1 after the launch


def call(fn, stream=None, **kw):
    """Call `int dig_<fn>(const dig_<fn>_args*, void* stream)` with keyword fields; tensors become pointers.
    The launch goes to the device the tensor arguments live on (they must all share one HIP device; CPU tensors and
    mixed devices raise before anything is launched), on that device's current torch stream unless `stream` is given.
    Raises RuntimeError on a non-zero return."""
    import torch

    L = lib()
    dev = None
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            if not v.is_cuda:
                raise RuntimeError(f"dig_{fn}: argument {k!r} is a {v.device} tensor; the kernels take HIP-device pointers only "
                                   "(no CPU fallback) - move the module and its inputs to the device first")
            if dev is None:
                dev = v.device
            elif v.device != dev:
                raise RuntimeError(f"dig_{fn}: argument {k!r} lives on {v.device} but another argument on {dev}")
    S, ftypes = _struct_for(fn)
    args = S()
    for k, v in kw.items():
        ftype = ftypes.get(k)
        if ftype is None:
            raise TypeError(f"{S.__name__} has no field {k!r}")
        if ftype is ctypes.c_void_p:
            setattr(args, k, v if v is None or isinstance(v, int) else v.data_ptr())
        elif hasattr(ftype, "_length_"):
            setattr(args, k, ftype(*v))
        else:
            setattr(args, k, v)
    tr = TRACE
    if tr is not None:
        tr(fn, kw, 0)
    if dev is None or dev.index in (None, torch.cuda.current_device()):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(L, "dig_" + fn)(ctypes.byref(args), ctypes.c_void_p(stream))
    else:  # tensors on a device that is not current: the launchers key their per-device state on hipGetDevice()
        with torch.cuda.device(dev):
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            rc = getattr(L, "dig_" + fn)(ctypes.byref(args), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"dig_{fn} failed ({rc}): {L.dig_last_error().decode()}")
    if tr is not None:
        tr(fn, kw, 1)


def dtype_code(torch_dtype):
    import torch

    if torch_dtype == torch.bfloat16:
        return DIG_BF16
    if torch_dtype == torch.float32:
        return DIG_F32
    raise ValueError(f"unsupported dtype {torch_dtype}; the HIP path stores activations as bfloat16 or float32")


def _tver(t):
    """Version counter of a tensor, 0 for an inference tensor: tensors created under torch.inference_mode() (the reference
    decorates sample_from_t / ddim_sample_from_t with it, ddp.py:770,827) track no version and raise on `_version`; they
    cannot be written in place outside inference mode either, so a constant is a sound stand-in for the cache key."""
    return 0 if t.is_inference() else t._version


def _pver(params):
    """Sum of the parameters' version counters (the re-pack key); inference tensors count 0, see _tver."""
    return sum(_tver(q) for q in params)
