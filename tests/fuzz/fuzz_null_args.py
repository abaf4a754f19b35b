"""Every entry point of include/rgc_hip.h called with nothing: NULL for every pointer, 0 for every number -- without a context, with a fresh
context, and with a context that holds clouds.  A C-ABI answers that with a status, not with a crash.  The prototypes are read from the header.
    python tests/fuzz/fuzz_null_args.py"""
import sys, os, re, json, ctypes as C, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np
from rgc_slam_amd import _lib, registration as reg
import rgc_slam_amd.synth as synth

hdr = open(os.path.join(ROOT, "include", "rgc_hip.h")).read()
hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
protos = re.findall(r"RGC_API\s+([\w\s\*]+?)\s*\b(rgc_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
L = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)    # a second handle without the mirror's prototypes: arguments as this script builds them


def ctype_of(arg):
    a = " ".join(arg.split())
    if a in ("void", ""):
        return None
    if "*" in a or "[" in a:
        return C.c_void_p
    base = a.rsplit(" ", 1)[0] if " " in a else a
    if "double" in base: return C.c_double
    if "float" in base: return C.c_float
    if "size_t" in base or "long long" in base: return C.c_longlong
    return C.c_int


SKIP = {"rgc_create", "rgc_destroy", "rgc_host_free", "rgc_device_free"}   # (life-cycle: exercised below on their own)
v = reg.odometer_vgicp(0)
w = reg.odometer_vgicp(0)
world, base = synth.make_world_and_map(5000, seed=3)
w.setInputTarget(base.astype(np.float32)); w.setInputSource(base[::3].astype(np.float32) + np.float32(0.01))
rep = {"functions": 0, "calls": 0, "statuses": {}, "failures": []}
for ret, name, args in protos:
    if name in SKIP:
        continue
    arg_list = [a for a in args.split(",")] if args.strip() not in ("", "void") else []
    types = [ctype_of(a) for a in arg_list]
    if any(t is None for t in types):
        types = [t for t in types if t is not None]
    fn = getattr(raw, name)
    fn.argtypes = types
    fn.restype = C.c_void_p if "*" in ret else (C.c_int if "int" in ret else None)
    takes_ctx = bool(arg_list) and "rgc_ctx" in arg_list[0]
    rep["functions"] += 1
    for variant, ctx in (("no context", None), ("fresh context", v._h), ("context with clouds", w._h)):
        if not takes_ctx and variant != "no context":
            continue
        vals = []
        for i, (a, t) in enumerate(zip(arg_list, types)):
            if i == 0 and takes_ctx:
                vals.append(ctx)
            elif t is C.c_void_p:
                vals.append(None)
            elif t in (C.c_double, C.c_float):
                vals.append(0.0)
            else:
                vals.append(0)
        print("calling", name, variant, flush=True, file=sys.stderr)
        try:
            r = fn(*vals)
            rep["calls"] += 1
            if fn.restype is C.c_int:
                rep["statuses"][str(r)] = rep["statuses"].get(str(r), 0) + 1
        except Exception as e:
            rep["failures"].append(dict(function=name, variant=variant, error=repr(e)))
# the contexts are none the worse for it (some of the calls above were legitimate: rgc_clear_source with nothing else to say clears the scan)
try:
    w.setInputTarget(base.astype(np.float32)); w.setInputSource(base[::3].astype(np.float32) + np.float32(0.01))
    w.align(np.eye(4, dtype=np.float32), want_output=False)
    rep["context_still_works"] = bool(np.all(np.isfinite(w.getFinalTransformation())))
except Exception as e:
    rep["failures"].append(dict(function="rgc_align afterwards", error=repr(e)))
# life-cycle entries with nothing
raw.rgc_destroy.argtypes = [C.c_void_p]; raw.rgc_destroy(None)
raw.rgc_host_free.argtypes = [C.c_void_p]; raw.rgc_host_free(None)
raw.rgc_device_free.argtypes = [C.c_void_p, C.c_void_p]; raw.rgc_device_free(None, None); raw.rgc_device_free(v._h, None)
raw.rgc_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p]; raw.rgc_create.restype = C.c_int
rep["create_with_null_out"] = raw.rgc_create(0, None, None)
rep["create_on_device_99"] = raw.rgc_create(99, None, C.byref(C.c_void_p()))
v.close(); w.close()
print(json.dumps(rep))
