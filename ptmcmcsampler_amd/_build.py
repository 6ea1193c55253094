"""Builds libptmi.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ptmi_kernels.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "ptmi_device.h"), os.path.join(os.path.dirname(HERE), "include", "ptmi.h")]
OUT = os.path.join(HERE, "libptmi.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
