"""Builds trace_amd/libtrace_hip.so (gfx950, bf16 elements) and trace_amd/libtrace_hip_f16.so (the same sources with -DTRACE_F16: IEEE fp16
elements, the reference's own inference dtype) from trace_amd/csrc/*.hip with hipcc.

In-tree build on purpose: the .so travels with the repo snapshot to the GPU box (a JIT cache would not).
`python -m trace_amd.build [--force]`
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libtrace_hip.so")
LIB_F16 = os.path.join(HERE, "libtrace_hip_f16.so")
SOURCES = ["gemm", "gemm_ldr", "gemm_pers", "gemm_w4", "norm", "vit", "patch_embed", "attn", "slot_pool", "llm", "decode", "fp8", "stc", "preproc", "engine"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (this package has no CPU fallback)")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "trace_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(name: str, force: bool, f16: bool = False) -> str:
    src, obj = os.path.join(CSRC, name + ".hip"), os.path.join(OBJ, name + ("_f16.o" if f16 else ".o"))
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _newest_header())):
        return obj
    cmd = [_hipcc(), *FLAGS, *(["-DTRACE_F16"] if f16 else []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}.hip{' (-DTRACE_F16)' if f16 else ''}:\n{r.stderr[-4000:]}")
    return obj


def _link(lib: str, objs, force: bool, verbose: bool) -> None:
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print("linked", lib)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    jobs = [(n, False) for n in SOURCES] + [(n, True) for n in SOURCES]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    _link(LIB, objs[:len(SOURCES)], force, verbose)
    _link(LIB_F16, objs[len(SOURCES):], force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
