"""Builds libtmpnn.so (hand-written HIP for gfx950) in-tree with hipcc. No JIT cache, no torch extension:
the library is a plain C-ABI shared object loaded with ctypes (thermompnn_amd/_lib.py)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtmpnn.so")
SOURCES = ["tmpnn_api.hip", "tmpnn_graph.hip", "tmpnn_layers.hip", "tmpnn_head.hip", "tmpnn_split.hip", "tmpnn_pdb.cpp"]
HEADERS = ["tmpnn_common.h", "tmpnn_split.h", "tmpnn_internal.h", os.path.join("..", "..", "include", "tmpnn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtmpnn.so cannot be built on this machine")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(o)
        cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
