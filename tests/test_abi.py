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


def test_python_mirror_prototypes_match_the_header(lib):
    """The ctypes prototypes the tests and bench.py call through (rgc_slam_amd/_lib.py) against include/rgc_hip.h: every declared function has argtypes,
    as many as the header's parameters, pointer where the header has a pointer or an array and scalar where it has a scalar (a mismatch is silent in ctypes)."""
    L = lib.load()
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rgc_hip.h")).read(), flags=re.S)
    checked = 0
    for m in re.finditer(r"RGC_API\s+[\w\s\*]+?\b(rgc_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", h, re.S):
        name, args = m.group(1), m.group(2).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        f = getattr(L, name)
        if not params:
            assert not f.argtypes, name
            continue
        assert f.argtypes is not None and len(f.argtypes) == len(params), (name, len(params), f.argtypes)
        for p, t in zip(params, f.argtypes):
            is_ptr = "*" in p or "[" in p
            t_ptr = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or issubclass(t, (C._Pointer, C.Array)) or (hasattr(t, "_type_") and not isinstance(t._type_, str))
            assert is_ptr == bool(t_ptr), (name, p, t)
            if not is_ptr and re.match(r"(const\s+)?(double|float)\b", p):
                assert t in (C.c_double, C.c_float) and (t is C.c_double) == ("double" in p), (name, p, t)
        checked += 1
    assert checked >= 80


def test_python_mirror_struct_layouts_match_the_header(lib, tmp_path):
    """sizeof and every field's offset of the ctypes Structures (rgc_slam_amd/_lib.py) against the C compiler's view of include/rgc_hip.h's structs
    (a generated C program prints offsetof for each field, in the header's order)."""
    import subprocess
    pairs = {"rgc_params": lib.Params, "rgc_stats": lib.Stats, "rgc_fuse_in": lib.FuseIn, "rgc_imu_filter": lib.ImuFilter, "rgc_ground_gate": lib.GroundGate,
             "rgc_fe_params": lib.FeParams, "rgc_fe_out": lib.FeOut, "rgc_mapreg_report": lib.MapregReport, "rgc_icp_params": lib.IcpParams,
             "rgc_icp_result": lib.IcpResult, "rgc_pc2_layout": lib.Pc2Layout, "rgc_pc2_field": lib.Pc2Field, "rgc_mapreg_ground": lib.MapregGround,
             "rgc_mapreg_imu": lib.MapregImu, "rgc_map_info": lib.MapInfo}
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rgc_hip.h")).read(), flags=re.S)
    fields = {}
    for m in re.finditer(r"typedef struct (rgc_\w+)\s*\{(.*?)\}\s*\1\s*;", h, re.S):
        names = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                for part in decl.split(","):          # `double a, b[3]` declares two fields
                    names.append(re.sub(r"\[.*", "", part.strip().split()[-1].lstrip("*")))
        fields[m.group(1)] = names
    assert set(pairs) <= set(fields), sorted(set(pairs) - set(fields))
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "rgc_hip.h"', "int main(void) {"]
    for s, names in fields.items():
        if s in pairs:
            src.append(f'  printf("{s} %zu", sizeof({s}));')
            src += [f'  printf(" %zu", offsetof({s}, {n}));' for n in names]
            src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(tmp_path / "layout.c"), "-o", str(tmp_path / "layout")])
    out = subprocess.run([str(tmp_path / "layout")], capture_output=True, text=True, check=True).stdout
    seen = 0
    for line in out.splitlines():
        s, size, *offs = line.split()
        cls = pairs[s]
        assert C.sizeof(cls) == int(size), (s, C.sizeof(cls), size)
        assert [getattr(cls, f[0]).offset for f in cls._fields_] == [int(o) for o in offs], (s, [f[0] for f in cls._fields_], fields[s])
        seen += 1
    assert seen == len(pairs)


def test_oracle_mirror_matches_its_header(tmp_path):
    """The checker's own binding (oracle/oracle.py over oracle/rgc_oracle.h) held to the same two checks: a ctypes mismatch in the ORACLE would corrupt what
    every parity test compares against.  Prototypes: count, pointer / scalar, float / double.  Structs: sizeof and every field's offset."""
    import subprocess
    from oracle import oracle as orc
    L = orc.lib()
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "oracle", "rgc_oracle.h")).read(), flags=re.S)
    checked = 0
    for m in re.finditer(r"^\s*([\w\s\*]+?)\b(orc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, re.S | re.M):
        name, args = m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        f = getattr(L, name)
        assert f.argtypes is not None and len(f.argtypes) == len(params), (name, len(params), f.argtypes)
        for p, t in zip(params, f.argtypes):
            is_ptr = "*" in p or "[" in p
            t_ptr = t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (hasattr(t, "_type_") and not isinstance(t._type_, str))
            assert is_ptr == bool(t_ptr), (name, p, t)
            if not is_ptr and re.match(r"(const\s+)?(double|float)\b", p):
                assert (t is C.c_double) == ("double" in p) and t in (C.c_double, C.c_float), (name, p, t)
        checked += 1
    assert checked >= 40
    pairs = {"orc_params": orc.Params, "orc_lm_trace": orc.LmTrace, "orc_fe_params": orc.FeParams, "orc_fe_out": orc.FeOut, "orc_edge_factor": orc.EdgeFactor,
             "orc_plane_factor": orc.PlaneFactor, "orc_mapreg_trace": orc.MapregTrace, "orc_mapreg_ground": orc.MapregGround, "orc_mapreg_imu": orc.MapregImu,
             "orc_icp_params": orc.IcpParams, "orc_icp_result": orc.IcpResult}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "rgc_oracle.h"', "int main(void) {"]
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(orc_\w+)\s*;", h, re.S):
        if m.group(2) not in pairs:
            continue
        names = []
        for decl in m.group(1).split(";"):
            if decl.strip():
                for part in decl.strip().split(","):
                    names.append(re.sub(r"\[.*", "", part.strip().split()[-1].lstrip("*")))
        src.append(f'  printf("{m.group(2)} %zu", sizeof({m.group(2)}));')
        src += [f'  printf(" %zu", offsetof({m.group(2)}, {n}));' for n in names]
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "oracle"), str(tmp_path / "layout.c"), "-o", str(tmp_path / "layout")])
    out = subprocess.run([str(tmp_path / "layout")], capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(pairs)
    for line in out:
        s, size, *offs = line.split()
        cls = pairs[s]
        assert C.sizeof(cls) == int(size), (s, C.sizeof(cls), size)
        assert [getattr(cls, f[0]).offset for f in cls._fields_] == [int(o) for o in offs], (s, [f[0] for f in cls._fields_])
