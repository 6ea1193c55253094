"""Builds libptmi.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The per-chain kernel templates are compiled per shape, likelihood family and part (ptmi_shape.hip with -DPTMI_G/-DPTMI_E/-DPTMI_L/-DPTMI_PART), all
translation units in parallel, then linked into one shared library."""
import concurrent.futures
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.path.join(HERE, "libptmi.so")


def deps():
    """Every source the library is built from: all of csrc/*.hip, csrc/*.h and the public header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(HERE), "include", "ptmi.h")]


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"] + os.environ.get("PTMI_EXTRA_CXXFLAGS", "").split()


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def shapes():
    txt = open(os.path.join(CSRC, "ptmi_common.h")).read()
    line = re.search(r"#define PTMI_SHAPE_LIST\(X\)(.*)", txt).group(1)
    return [(int(g), int(e)) for g, e in re.findall(r"X\((\d+),\s*(\d+)\)", line)]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in deps())


def _compile(job):
    src, obj, defs = job
    r = subprocess.run([hipcc()] + FLAGS + defs + ["-c", src, "-o", obj], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        hint = ""
        if "-amdgpu-use-amdgpu-trackers" in defs and "amdgpu-use-amdgpu-trackers" in r.stderr:
            hint = ("\nThis hipcc does not know -mllvm -amdgpu-use-amdgpu-trackers (LLVM's AMDGPU register-pressure trackers, ROCm >= 7): "
                    "the iso / dense step kernels are tuned and parity-tested with it; build with ROCm 7.x.")
        raise RuntimeError("hipcc failed on %s %s:\n%s%s" % (os.path.basename(src), " ".join(defs), r.stderr[-4000:], hint))
    return obj


def build(force=False, verbose=False, jobs=None):
    if not force and not stale():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    work = [(os.path.join(CSRC, "ptmi_abi.hip"), os.path.join(OBJ, "abi.o"), []),
            (os.path.join(CSRC, "ptmi_split.hip"), os.path.join(OBJ, "split.o"), [])]         # the split path's row kernels (shape-independent)
    for g, e in sorted(shapes(), key=lambda s: -s[0] * s[1]):       # biggest units first
        for fam in (1, 0, 2, 3):
            if fam == 3 and e > 8:          # the interval family (PTMI_LOGL_INTERVAL): the gradient-jump shapes only (PTMI_GJ_SHAPE_LIST)
                continue
            # the iso / dense step kernels sit at the register limit of their occupancy: with LLVM's AMDGPU register-pressure
            # trackers the scheduler spills 20 instead of 96 bytes in the config-2 kernel (1.107 -> 1.072 ms per 100 steps,
            # dense 12.8 -> 12.5); the curved family (gradient jumps) measured 1.5 % slower with them and keeps the default
            # + the max-ILP scheduling strategy for the kernels of SCAM-only cycles of the same two families (part 0 of a shape:
            # config-2 kernel 0.797 -> 0.782 ms per 100 steps, dense 5.81 -> 5.72); the kernels of cycles with AM / DE entries
            # (part 1) measured 1-2 % slower with it and keep the default strategy.  max-memory-clause, metric bias 0, relaxed
            # occupancy and no post-RA scheduling measured within 0.5 % of the default or worse.  PTMI_NO_ILP=1: an A/B build without it.
            track = ["-mllvm", "-amdgpu-use-amdgpu-trackers"] if fam < 2 else []
            ilp = [] if os.environ.get("PTMI_NO_ILP") else (["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if fam < 2 else [])
            defs = ["-DPTMI_G=%d" % g, "-DPTMI_E=%d" % e, "-DPTMI_L=%d" % fam]
            work.append((os.path.join(CSRC, "ptmi_shape.hip"), os.path.join(OBJ, "shape_%d_%d_%d.o" % (g, e, fam)), defs + ["-DPTMI_PART=0"] + track + ilp))
            work.append((os.path.join(CSRC, "ptmi_shape.hip"), os.path.join(OBJ, "shape_full_%d_%d_%d.o" % (g, e, fam)), defs + ["-DPTMI_PART=1"] + track))
    jobs = jobs or min(len(work), os.cpu_count() or 1)
    if verbose:
        print("compiling %d translation units with %d workers" % (len(work), jobs))
    with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(_compile, work))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT + ".tmp"] + objs)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
