"""Are the reference citations (file:line) of this repository real?  Every `name.ext:line[-line]` in the given files (default: include/rgc_hip.h, the
oracle's sources, the kernels) whose file name is one of the reference's (/root/reference, read-only; present in the build
container only) is resolved there and its line range checked against the file's length.  Reports unknown files and ranges past the end; exit code 1 if any.
    python scripts/check_citations.py [files ...]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DEFAULT = ["include/rgc_hip.h", "oracle/rgc_oracle.c", "oracle/rgc_oracle_aux.c", "oracle/rgc_oracle_map.c", "oracle/rgc_oracle.h", "oracle/py_oracle.py", "oracle/py_frontend.py",
           "oracle/py_fusion.py", "oracle/py_icp.py", "oracle/py_mapreg.py", "oracle/py_odometer.py", "rgc-slam_amd/csrc/rgc_kernels.hip", "rgc-slam_amd/csrc/rgc_api.hip",
           "rgc-slam_amd/csrc/rgc_frontend.hip", "rgc-slam_amd/csrc/rgc_pre.hip", "rgc-slam_amd/csrc/rgc_host.cpp", "rgc-slam_amd/cpp/odometry_node.hpp",
           "rgc-slam_amd/cpp/fast_vgicp_hip.hpp", "rgc-slam_amd/odometry.py", "rgc-slam_amd/registration.py", "DESIGN.md", "INTEGRATION.md", "SURVEY.md"]
CITE = re.compile(r"([\w/\.]*\b[\w]+\.(?:cpp|hpp|h|cu|cuh|launch|yaml|msg)):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")


def reference_files():
    by = {}
    for d, _, fs in os.walk(REF):
        for f in fs:
            by.setdefault(f, []).append(os.path.join(d, f))
    return by


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing checked")
        return 0
    by = reference_files()
    length = {}
    files = sys.argv[1:] or DEFAULT
    total, bad, own = 0, [], 0
    for rel in files:
        p = os.path.join(ROOT, rel)
        if not os.path.exists(p):
            continue
        for ln, line in enumerate(open(p, errors="replace").read().splitlines(), 1):
            for m in CITE.finditer(line):
                name, ranges = m.group(1), m.group(2)
                base = os.path.basename(name)
                if base not in by:
                    if os.path.exists(os.path.join(ROOT, name)) or any(base == os.path.basename(x) for x in DEFAULT) or base.startswith(("rgc_", "test_", "fuzz_")):
                        own += 1            # a citation of this repository's own file
                    else:
                        bad.append((rel, ln, name, ranges, "no such file in the reference"))
                    continue
                cands = [c for c in by[base] if c.endswith(name)] or by[base]
                total += 1
                last = max(int(x) for x in re.findall(r"\d+", ranges))
                ok = False
                for c in cands:
                    if c not in length:
                        length[c] = sum(1 for _ in open(c, errors="replace"))
                    ok = ok or last <= length[c]
                if not ok:
                    bad.append((rel, ln, name, ranges, "past the end (%s lines)" % "/".join(str(length[c]) for c in cands)))
    print(f"{total} citations of reference files checked in {len(files)} files ({own} of the repository's own files skipped); {len(bad)} do not resolve")
    for b in bad[:60]:
        print("  %s:%d  %s:%s  -- %s" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
