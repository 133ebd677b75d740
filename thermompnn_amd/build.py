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
# The debug variant (-DTMPNN_DEBUG_BUILD): the same kernels plus the TMPNN_* kernel-form switches and per-phase timers that the
# shipped library does not contain (its launchers never read the environment, allocate or synchronise). A/B tests, tools/dbg_*.py
# and tools/phase_prof.py load it through TMPNN_LIB.
DEBUG_LIB = os.path.join(HERE, "libtmpnn_debug.so")
SOURCES = ["tmpnn_api.hip", "tmpnn_graph.hip", "tmpnn_layers.hip", "tmpnn_head.hip", "tmpnn_split.hip", "tmpnn_edge.hip", "tmpnn_msg.hip",
           "tmpnn_edge_msg.hip", "tmpnn_edge_wave.hip", "tmpnn_node.hip", "tmpnn_pdb.cpp", "tmpnn_csv.cpp"]
HEADERS = ["tmpnn_common.h", "tmpnn_split.h", "tmpnn_internal.h", "tmpnn_head_body.h", os.path.join("..", "..", "include", "tmpnn.h"),
           os.path.join("..", "..", "include", "tmpnn_debug.h"), "tmpnn_host_guard.hpp"]
# -mcode-object-version=5: tm_nblk() / tm_bdim() (tmpnn_common.h) read gridDim / blockDim at fixed offsets of the v5
# implicit-argument block; pinned here and checked on the device by tmpnn_selftest.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-mcode-object-version=5"]
# kernels only (.hip): no sNaN-quieting "v_max x, x" in front of every v_min / v_max (the GELU clamps). Device code never
# relies on NaNs (range checks test the exponent bits); the host-side PDB reader (.cpp) does and keeps IEEE semantics.
DEVICE_FLAGS = ["-mno-amdgpu-ieee", "-fno-honor-nans"]
# ... except the f16x2 per-edge / node kernels (tmpnn_edge / _msg / _edge_msg / _node.hip, and tmpnn_split.hip with the device self-test
# of exactly this property): their GELU clamps are gfx950's NaN-PROPAGATING v_minimum3_f32 /
# v_maximum3_f32 (IEEE-754-2019; no canonicalising op in front of them either), which hipcc emits from __builtin_elementwise_minimum /
# maximum only in a translation unit that honours NaNs (under -fno-honor-nans they degrade to v_min / v_max). See gelu2, TM_GELU_NAN3.
NAN3_FILES = ("tmpnn_split.hip", "tmpnn_edge.hip", "tmpnn_msg.hip", "tmpnn_edge_msg.hip", "tmpnn_edge_wave.hip", "tmpnn_node.hip")
FILE_FLAGS = {f: ["-DTM_GELU_NAN3=1"] for f in NAN3_FILES}
# tmpnn_edge_wave.hip (one wavefront per SIMD, 512 registers): MFMA accumulators in VGPRs, so that the accumulation half of the register
# file is free for the register-resident W13 (256 AGPRs, read by the MFMAs directly)
FILE_FLAGS["tmpnn_edge_wave.hip"] = FILE_FLAGS["tmpnn_edge_wave.hip"] + ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtmpnn.so cannot be built on this machine")
    return exe


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB, tag: str = "", only=()) -> str:
    """``extra_flags`` / ``out`` / ``tag`` build an A/B variant (e.g. ``-DWT_PIPE=0``) beside the shipped library;
    select it at run time with TMPNN_LIB=<path> (thermompnn_amd/_lib.py)."""
    if not force and out == LIB and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        if only and s not in only:          # variant of a few files: link the shipped objects of the others
            objs.append(os.path.join(CSRC, os.path.splitext(s)[0] + ".o"))
            continue
        o = os.path.join(CSRC, os.path.splitext(s)[0] + tag + ".o")
        objs.append(o)
        per_file = FILE_FLAGS.get(s, DEVICE_FLAGS if s.endswith(".hip") else [])
        cmd = [_hipcc(), *FLAGS, *per_file, *extra_flags, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{log}")
        if verbose and log.strip():
            print(log, file=sys.stderr)
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    if tag:
        for o in objs:
            if o.endswith(tag + ".o"):
                os.remove(o)
    return out


def build_debug_library(force: bool = False, verbose: bool = False) -> str:
    """libtmpnn_debug.so: every source again with -DTMPNN_DEBUG_BUILD (own object files)."""
    if not force and not needs_build(DEBUG_LIB):
        return DEBUG_LIB
    return build_library(force=True, verbose=verbose, extra_flags=["-DTMPNN_DEBUG_BUILD"], out=DEBUG_LIB, tag=".debug")


def build_pdb_sanitizer_driver(out: str | None = None) -> str:
    """The host-side PDB reader (csrc/tmpnn_pdb.cpp, untrusted text in) + tests/native/pdb_fuzz_driver.cpp as ONE executable
    under AddressSanitizer + UndefinedBehaviorSanitizer (g++; a report aborts the process). Used by the malformed-input test."""
    repo = os.path.dirname(HERE)
    out = out or os.path.join(repo, "tests", "native", "pdb_fuzz_driver")
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found: the sanitizer driver cannot be built on this machine")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-Wall", "-pthread", os.path.join(CSRC, "tmpnn_pdb.cpp"), os.path.join(CSRC, "tmpnn_csv.cpp"),
           os.path.join(repo, "tests", "native", "pdb_fuzz_driver.cpp"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed on the sanitizer driver:\n" + r.stdout)
    return out


def build_native_example(out: str | None = None) -> str:
    """examples/scan_native.cpp: a host without Python (PDB files -> CSV through the C-ABI + the HIP runtime API only).
    Plain host C++ (g++); the ROCm root is taken from where hipcc lives."""
    repo = os.path.dirname(HERE)
    out = out or os.path.join(repo, "examples", "scan_native")
    if not os.path.exists(LIB):
        build_library()
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(_hipcc())))
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found: examples/scan_native cannot be built on this machine")
    cmd = [gxx, "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(repo, "include"),
           "-I", os.path.join(rocm, "include"), os.path.join(repo, "examples", "scan_native.cpp"), "-L", HERE, "-ltmpnn",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-pthread", "-Wl,-rpath,$ORIGIN/../thermompnn_amd",
           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("building examples/scan_native failed:\n" + r.stdout)
    return out


if __name__ == "__main__":
    # python -m thermompnn_amd.build [--force] [--variant NAME -DFLAG ...]  (variant -> thermompnn_amd/libtmpnn_NAME.so)
    if "--pdb-sanitizer-driver" in sys.argv:
        print(build_pdb_sanitizer_driver())
    elif "--variant" in sys.argv:
        k = sys.argv.index("--variant")
        name, flags = sys.argv[k + 1], sys.argv[k + 2:]
        only = [f[7:] for f in flags if f.startswith("--only=")]      # --only=tmpnn_wt.hip: recompile just that file
        flags = [f for f in flags if not f.startswith("--only=")]
        print(build_library(force=True, verbose=True, extra_flags=flags, out=os.path.join(HERE, f"libtmpnn_{name}.so"), tag="." + name, only=only))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
        print(build_debug_library(force="--force" in sys.argv, verbose=True))
        print(build_native_example())
