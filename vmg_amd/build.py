"""Builds vmg_amd/libvmg_hip.so (gfx950) from vmg_amd/csrc/*.hip with hipcc.  In-tree, no JIT cache."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DIAG = os.environ.get("VMG_DIAG") == "1"  # diagnostics build (in-kernel stamps, ablation bits): a library of its own, never the shipped one
LIB = os.path.join(HERE, "libvmg_hip_diag.so" if DIAG else "libvmg_hip.so")
OBJDIR = os.path.join(HERE, "build_diag" if DIAG else "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Wshadow", "-Werror=shadow"] + (["-DVMG_DIAG"] if DIAG else [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    objs = []
    procs = []
    os.makedirs(OBJDIR, exist_ok=True)
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(
                os.path.getmtime(src), *(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))),
                os.path.getmtime(os.path.join(HERE, "..", "include", "vmg_hip.h"))):
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force=False, verbose=True))
