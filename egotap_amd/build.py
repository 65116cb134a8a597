"""Builds libegotap_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
hipcc cross-compiles for gfx950 without a GPU, so this also runs in the build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.environ.get("EGOTAP_LIB") or os.path.join(PKG, "libegotap_hip.so")   # EGOTAP_LIB: experiment builds
SOURCES = ["egotap_abi.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the egotap_amd HIP library cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", f) for f in ("egotap.h", "egotap_debug.h")]
    return any(os.path.getmtime(d) > t for d in deps)


PARTS = (0, 1, 2, 3)   # egotap_abi.hip compiled as four translation units in parallel (-DEGOTAP_PART=n), then linked


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    # experiment builds (EGOTAP_LIB=<other .so>, EGOTAP_CXXFLAGS="-DEGOTAP_ABL=1 ..."): objects next to their library, extra flags
    objdir = os.path.join(PKG, "build") if "EGOTAP_LIB" not in os.environ else LIB + ".build"
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    common += os.environ.get("EGOTAP_CXXFLAGS", "").split()
    procs, objs = [], []
    for src in SOURCES:
        for part in PARTS:
            obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}_part{part}.o")
            cmd = common + [f"-DEGOTAP_PART={part}", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            for _, other in procs:
                if other.poll() is None:
                    other.kill()
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(link))
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
