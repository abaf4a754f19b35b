"""Do two builds of the library hold the same gfx950 device code?  Extracts the code objects of both (llvm-objdump --offloading, in scratch directories) and
compares, per code object, the SHA-256 of the .text section and the kernel descriptors' resources (scripts/kernel_resources.py).  Used to show that the
in-tree librgc_hip.so is what the committed sources build to, and that a change meant to leave the product alone (a comment, a default-off flag) did.  No GPU.
    python scripts/same_device_code.py <library A> <library B>"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def text_hashes(lib):
    out = []
    with tempfile.TemporaryDirectory() as td:
        cp = os.path.join(td, "lib.so")
        shutil.copy(lib, cp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            sec = os.path.join(td, f + ".text")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", os.path.join(td, f), sec], check=True, capture_output=True)
            data = open(sec, "rb").read()
            out.append((len(data), hashlib.sha256(data).hexdigest()))
    return out


def main():
    a, b = sys.argv[1], sys.argv[2]
    ha, hb = text_hashes(a), text_hashes(b)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_resources as kr
    strip = lambda ks: sorted((k["name"], k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"]) for k in ks)
    ra, rb = strip(kr.kernels_of(a)), strip(kr.kernels_of(b))
    same_text, same_res = sorted(ha) == sorted(hb), ra == rb
    print(f"{a}: {len(ha)} gfx950 code objects, .text {sum(x[0] for x in ha)} bytes, {len(ra)} kernels")
    print(f"{b}: {len(hb)} gfx950 code objects, .text {sum(x[0] for x in hb)} bytes, {len(rb)} kernels")
    print("device code (.text, SHA-256 per code object):", "IDENTICAL" if same_text else "DIFFERENT")
    print("kernel resources (registers, LDS, scratch, spills per kernel):", "IDENTICAL" if same_res else "DIFFERENT")
    if not same_res:
        da, db = dict((r[0], r[1:]) for r in ra), dict((r[0], r[1:]) for r in rb)
        for n in sorted(set(da) | set(db)):
            if da.get(n) != db.get(n):
                print("  ", n[:90], da.get(n), "->", db.get(n))
    sys.exit(0 if same_text and same_res else 1)


if __name__ == "__main__":
    main()
