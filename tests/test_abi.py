"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/rgc_hip.h declares, and refuses
to work without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rgc_slam_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("rgc_build", os.path.join(ROOT, "rgc-slam_amd", "build.py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        m.build()
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "rgc_hip.h")).read()
    declared = sorted(set(re.findall(r"RGC_API[^;(]*?\b(rgc_\w+)\s*\(", hdr)))
    assert len(declared) >= 30
    assert sorted(lib.SYMBOLS) == declared, "binding list and header disagree"
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), f"librgc_hip.so does not export {name}"
    assert b"gfx950" in L.rgc_version()


def test_default_params(lib):
    p = lib.default_params()
    assert (p.voxel_res, p.max_iterations, p.lm_max_iterations, p.k_correspondences) == (1.0, 25, 10, 20)
    assert (p.rotation_eps, p.translation_eps, p.lm_init_lambda_factor) == (2e-3, 1e-6, 1e-9)
    assert p.neighbor_method == lib.DIRECT1


def test_no_cpu_fallback(lib):
    """without a GPU the product must fail loudly, never silently compute on the host"""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from rgc_slam_amd import registration as r\n"
            "try:\n    r.FastVGICP(0); print('CREATED')\nexcept r.RgcError as e:\n    print('ERR', e.status)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout
    assert "ERR -2" in out, out


def test_product_does_not_touch_oracle():
    """nothing under the package or the C-ABI sources may reference oracle/"""
    pkg = os.path.join(ROOT, "rgc-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rgc_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_null_arguments_are_rejected_without_a_gpu(lib):
    """the entry points that take contexts check them before touching HIP: a null context is RGC_ERR_INVALID (-1), not a crash"""
    import ctypes as C
    L = lib.load()
    g = (C.c_float * 16)(*([0.0] * 16))
    assert L.rgc_align_begin(None, g, 0) == -1
    assert L.rgc_align_end(None, None, None, None, None, None, None) == -1
    assert L.rgc_share_target(None, None) == -1
    assert L.rgc_align(None, g, None, None, None, None, None, None) == -1


def test_knob_inventory_is_current():
    """rgc-slam_amd/csrc/KNOBS.md lists every build flag (#ifndef RGC_X / #define RGC_X default) and every environment variable of the library as the
    sources define them now (scripts/make_knobs.py regenerates it), and README.md names every environment variable."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_knobs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr or r.stdout
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_knobs
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert [n for n, _ in make_knobs.env_vars() if n not in readme] == []
    assert len(make_knobs.build_flags()) >= 30


def test_every_entry_point_selects_its_device():
    """One process may hold contexts on several GPUs and call them from threads whose current device is another one (a new thread starts on device 0).
    A static audit of csrc/rgc_api.hip, where every entry point lives: each exported rgc_* function from which a HIP runtime call or a kernel launch can
    be reached calls hipSetDevice(c->device) itself or in a function it calls directly.  (A one-GPU box cannot show the difference; round 6 found
    rgc_set_params and rgc_get_stats re-preparing clouds -- kernel launches -- on whatever device the caller's thread had current.)"""
    import re
    src = open(os.path.join(ROOT, "rgc-slam_amd", "csrc", "rgc_api.hip")).read()
    funcs = {}
    for m in re.finditer(r'^(?:extern "C" )?(?:RGC_API |static |inline )*[\w:<>\*& ]+?\b(\w+)\s*\(([^;{}]*?)\)\s*(?:const\s*)?\{', src, re.M):
        if m.group(1) in ("for", "if", "while", "switch", "catch"):
            continue
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += (src[i] == "{") - (src[i] == "}")
            i += 1
        funcs.setdefault(m.group(1), src[m.end():i])
    hip_call = re.compile(r"\bhip(?!SetDevice|Success|Error|GetErrorString|GetLastError|Stream_t|Event_t)[A-Z]\w*\s*\(|<<<|hipLaunchKernelGGL|hipExtLaunch")

    def reaches_hip(name, seen=frozenset()):
        body = funcs.get(name)
        if body is None or name in seen:
            return False
        return bool(hip_call.search(body)) or any(reaches_hip(c, seen | {name}) for c in set(re.findall(r"\b(\w+)\s*\(", body)) if c in funcs and c != name)

    exported = [n for n in funcs if n.startswith("rgc_")]
    assert len(exported) >= 75
    no_context = {"rgc_host_alloc", "rgc_host_free"}                    # pinned host memory, hipHostMallocPortable: no context in the signature
    bad = []
    for n in exported:
        if n in no_context or n.startswith("rgc_lab_") or not reaches_hip(n):
            continue
        body = funcs[n]
        if "hipSetDevice" in body or any("hipSetDevice" in funcs[c] for c in set(re.findall(r"\b(\w+)\s*\(", body)) if c in funcs and c != n):
            continue
        bad.append(n)
    assert bad == [], bad
    # ... and selects it BEFORE the first launch or allocation it can reach, in the order of the text (rgc_align_end scored the general route's pose first)
    device_work = re.compile(r"<<<|hipLaunchKernelGGL|hipExtLaunch|hipMalloc\b|hipMallocAsync|hipFree\b|hipEventCreate|hipStreamCreate|hipMemset")

    def launches(name, seen=frozenset()):
        body = funcs.get(name)
        if body is None or name in seen:
            return False
        return bool(device_work.search(body)) or any(launches(c, seen | {name}) for c in set(re.findall(r"\b(\w+)\s*\(", body)) if c in funcs and c != name)

    late = []
    for n in exported:
        if n in no_context or n.startswith("rgc_lab_"):
            continue
        first_set = first_work = None
        for m in re.finditer(r"\b(\w+)\s*\(|<<<", funcs[n]):
            tok = m.group(1) or "<<<"
            sets_it = tok == "hipSetDevice" or (tok in funcs and "hipSetDevice" in funcs[tok])
            if sets_it and first_set is None:
                first_set = m.start()
            if first_work is None and not sets_it and (tok == "<<<" or device_work.match(tok) or (tok in funcs and tok != n and launches(tok))):
                first_work = m.start()
        if first_work is not None and (first_set is None or first_work < first_set):
            late.append(n)
    assert late == [], late


def test_integration_names_every_entry_point():
    """INTEGRATION.md's appendix (scripts/make_abi_index.py, generated from include/rgc_hip.h) is current and names every symbol the header declares."""
    import re
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_abi_index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr or r.stdout
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    syms = set(re.findall(r"\b(rgc_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "rgc_hip.h")).read()))
    assert len(syms) >= 85 and [s for s in sorted(syms) if "`" + s + "`" not in doc] == []
