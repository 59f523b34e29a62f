"""Build libbitdance_hip.so (gfx950) in-tree with hipcc.  `python -m bitdance_amd.build`"""
from __future__ import annotations

import fcntl
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libbitdance_hip.so")
SOURCES = ["bd_gemm.hip", "bd_gemm_tile.hip", "bd_gemm_half.hip", "bd_gemm8.hip", "bd_rows.hip", "bd_conv.hip", "bd_attn.hip", "bd_comm.hip", "bd_sp.hip", "bd_api.hip"]
# -ffp-contract=off: the row kernels restate separately-rounded torch ops (bit-exact sampler update); HIP's
# __fadd_rn/__fmul_rn are plain operators, so contraction must be disabled at the compiler level.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required)")
    return exe


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True, stamp: bool = False) -> str:
    """Compile what is out of date and link.  Safe to call from every rank of a multi-process launch: one process builds
    under an exclusive file lock, the others wait and then find everything up to date; the library is replaced atomically.
    ``stamp``: the MEASUREMENT build libbitdance_hip_stamp.so (-DBD_GEMM_STAMP: in-kernel phase stamps, csrc/bd_common.h) that
    tools/launch_anatomy.py loads instead of the product library; the product build compiles none of that code."""
    obj = OBJ + ("_stamp" if stamp else "")
    os.makedirs(obj, exist_ok=True)
    with open(os.path.join(obj, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, obj, LIB.replace(".so", "_stamp.so") if stamp else LIB,
                                 FLAGS + (["-DBD_GEMM_STAMP"] if stamp else []))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, OBJ: str = OBJ, LIB: str = LIB, FLAGS: list = FLAGS) -> str:
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "bitdance_hip.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append((s, o))
    if not jobs and _newer(LIB, objs):
        return LIB
    hipcc = _hipcc()

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda so: run([hipcc, *FLAGS, "-c", so[0], "-o", so[1]]), jobs))
    tmp = LIB + f".tmp{os.getpid()}"
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp])
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamp="--stamp" in sys.argv))
