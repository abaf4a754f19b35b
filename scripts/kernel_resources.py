"""Static resources of every gfx950 kernel in the built library, read from the code objects' metadata (no GPU): VGPRs (+ AGPRs: one file on
gfx950), SGPRs, LDS and scratch bytes per workgroup / lane, spills, the largest workgroup, and the waves per SIMD the register allocation admits
(MI355X_MICROARCH.md: allocation in steps of 8, min(8, 512 / allocated)).  The library is copied to a scratch directory first: llvm-objdump
--offloading extracts next to its input.
    python scripts/kernel_resources.py [library ...] > profiles/r06_kernel_resources.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(lib):
    """[{name, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill, max_wg}] of every kernel in `lib`'s gfx950 code objects"""
    out = []
    with tempfile.TemporaryDirectory() as td:
        cp = os.path.join(td, os.path.basename(lib))
        shutil.copy(lib, cp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            doc = notes[notes.index("---"):]
            doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
            for k in yaml.safe_load(doc)["amdhsa.kernels"]:
                out.append(dict(name=k[".name"], vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"], lds=k[".group_segment_fixed_size"],
                                scratch=k[".private_segment_fixed_size"], vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                max_wg=k[".max_flat_workgroup_size"]))
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"\(.*$", "", n).replace("rgck::", "").replace("void ", "")
    return out


def waves_per_simd(k):
    alloc = -(-(k["vgpr"] + k.get("agpr", 0)) // 8) * 8     # vgpr_count already includes the AGPRs on a unified file when the compiler says so
    alloc = max(alloc, 8)
    return min(8, 512 // alloc)


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "rgc-slam_amd", "librgc_hip.so")]
    for lib in libs:
        ks = kernels_of(lib)
        print(f"# {os.path.relpath(lib, ROOT)}: {len(ks)} gfx950 kernels (code-object metadata; waves/SIMD = what the VGPR allocation admits, LDS and launch bounds aside)")
        print("# LDS B: the static part (the kNN kernels' per-lane columns are dynamic LDS, sized at launch); scratch B: per lane; SGPR spills go to VGPR lanes, not to memory")
        print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS B':>7s} {'scratch B':>9s} {'VGPR spills':>11s} {'SGPR spills':>11s} {'max WG':>6s} {'waves/SIMD':>10s}")
        for k in sorted(ks, key=lambda k: k["demangled"]):
            print(f"{k['demangled'][:72]:72s} {k['vgpr']:5d} {k.get('agpr', 0):5d} {k['sgpr']:5d} {k['lds']:7d} {k['scratch']:9d} "
                  f"{k['vgpr_spill']:11d} {k.get('sgpr_spill', 0):11d} {k['max_wg']:6d} {waves_per_simd(k):10d}")
        print()


if __name__ == "__main__":
    main()
