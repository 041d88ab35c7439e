#!/usr/bin/env python3
"""sha256 of the sections of libpm_engine.so that hold code and data — the gfx950 code object's .text / .rodata / .note
(kernel metadata: registers, LDS, arguments) and the host's .text / .rodata / .data.  The library as a whole differs from
build to build (symbol names carry a per-compilation id); these six do not, so a refactor that must not change the
product (moving code between files, renaming, comments) is one whose hashes are the same before and after:

    python tools/section_hashes.py [library]          # default: protocol_amd/libpm_engine.so, built if stale

(How round 4's split of pm_kernels.hip into pm_*.inc was checked.)"""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"


def section(path: str, name: str) -> bytes:
    with tempfile.NamedTemporaryFile() as f:
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section={name}={f.name}", path, "/dev/null"])
        return open(f.name, "rb").read()


def hashes(lib: str) -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat, dev = os.path.join(d, "fatbin"), os.path.join(d, "dev.co")
        open(fat, "wb").write(section(lib, ".hip_fatbin"))
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        for s in (".text", ".rodata", ".note"):
            out["gfx950 " + s] = hashlib.sha256(section(dev, s)).hexdigest()[:16]
    for s in (".text", ".rodata", ".data"):
        out["host " + s] = hashlib.sha256(section(lib, s)).hexdigest()[:16]
    return out


def toolchain() -> str:
    """the compiler the hashes belong to: hipcc's HIP and clang version lines"""
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return "unknown"
    hip = [l.split(":", 1)[1].strip() for l in out.splitlines() if l.startswith("HIP version")]
    clang = [l.strip() for l in out.splitlines() if "clang version" in l]
    return "; ".join((hip[:1] or ["?"]) + [c.split("(")[0].strip() for c in clang[:1]])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        lib = sys.argv[1]
    else:
        from protocol_amd import build as B
        lib = B.build()
    for k, v in hashes(lib).items():
        print(f"  {k:16s} {v}")
    print(f"  toolchain        {toolchain()}")
