"""Build libvaura_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vaura_amd.csrc.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemv.hip", "gemv3.hip", "attention.hip", "step.hip", "api.hip", "dac.hip", "post.hip", "vit.hip"]
HEADERS = ["common.h", "gemv_kernel.h", "gemv3_kernel.h", "mlp_engine.h", os.path.join("..", "..", "include", "vaura_hip.h")]
# csrc/experiments/*.h (the measured-negative engines of rounds 4-5) are compiled only with -DVAURA_EXPERIMENT_ENGINES=1:
#   python -m vaura_amd.csrc.build --tag engines -DVAURA_EXPERIMENT_ENGINES=1   -> libvaura_hip_engines.so (never loaded by the package)
LIB = os.path.join(HERE, "libvaura_hip.so")
# diagnostic build (--stamps): the same sources with -DVAURA_STAMPS (in-kernel s_memrealtime stamps, common.h); never loaded by the
# package, only by tools/pmc_driver --stamps
LIB_STAMPS = os.path.join(HERE, "libvaura_hip_stamps.so")
# experiment build (--plain-stores): -DVAURA_PLAIN_STORES, ordinary instead of write-through output stores (common.h); timed against
# the product by tools/pmc_driver
LIB_WT = os.path.join(HERE, "libvaura_hip_plain.so")
ARCH = "gfx950"
# -amdgpu-kernarg-preload-count: the first kernel arguments arrive in SGPRs at wave launch (gfx94x/gfx950); the
# compiler keeps a compatibility prologue that loads them the old way when the firmware does not preload
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src: str, force: bool, stamps=False) -> str:
    # stamps: False = product, True = -DVAURA_STAMPS, "wt" = -DVAURA_PLAIN_STORES, ("tag", "-DX=1", ..) = an experiment build
    if isinstance(stamps, tuple):
        obj = os.path.join(HERE, src.replace(".hip", f".{stamps[0]}.o"))
        cmd = [_hipcc(), *FLAGS, *stamps[1:], "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj
    obj = os.path.join(HERE, src.replace(".hip", ".wt.o" if stamps == "wt" else (".stamps.o" if stamps else ".o")))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS]
    if force or _stale(obj, deps):
        cmd = [_hipcc(), *FLAGS, *(["-DVAURA_PLAIN_STORES"] if stamps == "wt" else (["-DVAURA_STAMPS"] if stamps else [])), "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, stamps=False) -> str:
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    LIB = LIB_WT if stamps == "wt" else (LIB_STAMPS if stamps else globals()["LIB"])
    if isinstance(stamps, tuple):
        LIB = os.path.join(HERE, f"libvaura_hip_{stamps[0]}.so")
        force = True
    with cf.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, stamps), srcs))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    if "--tag" in sys.argv:      # experiment build: --tag NAME -DX=1 ... -> libvaura_hip_NAME.so (tools/experiment.sh; never loaded by the package)
        build(verbose=True, stamps=(sys.argv[sys.argv.index("--tag") + 1], *[a for a in sys.argv if a.startswith("-D")]))
    else:
        build(force="--force" in sys.argv, verbose=True, stamps="wt" if "--plain-stores" in sys.argv else ("--stamps" in sys.argv))
