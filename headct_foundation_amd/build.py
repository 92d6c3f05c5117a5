"""Build recipe for libheadct_hip.so (hipcc, gfx950 only).  In-tree so the .so travels with the snapshot."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libheadct_hip.so")
SOURCES = ["elementwise.hip", "optim.hip", "gemm.hip", "attention_simple.hip", "attention_mfma.hip", "attention.hip",
           "mae_plan.hip", "heads.hip", "prof.hip", "dino.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "prof.h"), os.path.join(os.path.dirname(HERE), "include", "headct_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-inline-asm", "-ffp-contract=off"]
# packed f32 VALU (v_pk_mul_f32 / v_pk_fma_f32 from the SLP vectoriser) issues slower than the two scalar operations beside MFMAs
FILE_FLAGS = {"attention_mfma.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(op, [sp] + HEADERS):
            jobs.append((sp, op))

    def cc(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return op

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_stamps_library(tag: str = "stamps", defines=("-DHCT_STAMPS",), src: str = "gemm.hip") -> str:
    """Diagnostic variant for scripts/stamp_gemm.py: gemm.hip with -DHCT_STAMPS (in-kernel timestamps), rest unchanged."""
    build_library()
    hipcc = _hipcc()
    op = os.path.join(OBJ, f"{src[:-4]}_{tag}.o")
    r = subprocess.run([hipcc] + FLAGS + FILE_FLAGS.get(src, []) + list(defines) + ["-c", os.path.join(CSRC, src), "-o", op], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed (stamps):\n{r.stdout}\n{r.stderr}")
    objs = [op if s == src else os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    out = os.path.join(HERE, f"libheadct_hip_{tag}.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed (stamps):\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps_library())
        sys.exit(0)
    if "--variant" in sys.argv:  # python -m headct_foundation_amd.build --variant TAG -DX=1 -DY=2   (diagnostic builds)
        i = sys.argv.index("--variant")  # ... --variant TAG [--src attention_mfma.hip] -DX=1
        rest = sys.argv[i + 2:]
        src = "gemm.hip"
        if "--src" in rest:
            j = rest.index("--src")
            src = rest[j + 1]
            rest = rest[:j] + rest[j + 2:]
        print(build_stamps_library(sys.argv[i + 1], tuple(rest), src))
        sys.exit(0)
    print(build_library(force="--force" in sys.argv, verbose=True))
