#!/usr/bin/env python3
"""Static instruction counts of the shipped kernels' tile loops (the input of bench.py's ISSUE roof).

Compiles every .hip of thermompnn_amd/csrc to gfx950 assembly with the flags of the shipped build
(hipcc --offload-device-only -S), finds in each kernel the largest loop (the persistent tile loop: the backward branch whose
span holds the most instructions) and counts what one wavefront issues per trip: MFMA by shape, VALU (packed, transcendental
and the rest separately — they issue at different rates), SALU, LDS, vector memory, barriers / waits.
    python tools/isa_counts.py [out.json]        (default profiles/r06_isa_counts.json; needs hipcc, no GPU)
The file is stamped with the hash of the kernel sources: bench.py ignores a file measured on other sources."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from thermompnn_amd import build as tm_build  # noqa: E402

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op: str) -> str:
    if op.startswith("v_mfma_") or op.startswith("v_smfmac_"):
        return "mfma:" + op
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu_trans"
        if op.startswith("v_pk_"):
            return "valu_packed"
        if op in ("v_nop",):
            return "nop"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt":
        return "waitcnt"
    if op in ("s_nop",):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels_of(asm: str):
    """-> {symbol: [(line_no, label | None, op | None)]} for every kernel (.amdhsa_kernel symbols)."""
    syms = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, flags=re.M))
    out, cur = {}, None
    for n, line in enumerate(asm.splitlines()):
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m and m.group(1) in syms:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if m:
            out[cur].append((n, m.group(1), None))
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            if t.startswith(".Lfunc_end"):
                cur = None
            continue
        op = t.split()[0]
        out[cur].append((n, None, op + " " + " ".join(t.split()[1:2])))
        if op == "s_endpgm":
            pass
    return out


def main_loop(body):
    """The persistent tile loop: among the backward branches, the span that holds the most MFMAs (all of a tile's) and, among
    those, the SHORTEST one (an outer span would add the prologue and the first-tile code in front of the loop)."""
    labels = {lab: k for k, (_, lab, _) in enumerate(body) if lab}
    best = None
    for k, (_, lab, ins) in enumerate(body):
        if ins and ins.startswith(("s_cbranch", "s_branch")):
            tgt = ins.split()[1].rstrip(",") if len(ins.split()) > 1 else ""
            if tgt in labels and labels[tgt] < k:
                span = [x for x in body[labels[tgt]:k + 1] if x[2]]
                mf = sum(1 for x in span if x[2].startswith(("v_mfma_", "v_smfmac_")))
                key = (mf, -len(span))
                if best is None or key > best[3]:
                    best = (labels[tgt], k, len(span), key)
    return best[:3] if best else None


def count(items):
    c = {}
    for _, _, ins in items:
        if ins:
            key = classify(ins.split()[0])
            c[key] = c.get(key, 0) + 1
    return c


def valu_histogram(items):
    """opcode -> count of the VALU instructions (the issue-roof bracket of bench.py prices them per opcode with the costs
    tools/probe/valu_cost_probe.hip measured)."""
    h = {}
    for _, _, ins in items:
        if ins:
            op = ins.split()[0]
            if classify(op).startswith("valu"):
                h[op] = h.get(op, 0) + 1
    return dict(sorted(h.items()))


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names) + "\n", capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "r06_isa_counts.json")
    import bench
    res = {"source_stamp": bench.kernel_source_stamp(), "flags": tm_build.FLAGS, "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in tm_build.SOURCES:
            if not src.endswith(".hip"):
                continue
            per_file = tm_build.FILE_FLAGS.get(src, tm_build.DEVICE_FLAGS)
            s_path = os.path.join(tmp, src + ".s")
            cmd = [tm_build._hipcc(), *tm_build.FLAGS, *per_file, "--offload-device-only", "-S", os.path.join(tm_build.CSRC, src), "-o", s_path]
            procs.append((src, s_path, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for src, s_path, p in procs:
            log, _ = p.communicate()
            if p.returncode != 0:
                raise SystemExit(f"hipcc failed on {src}:\n{log}")
            asm = open(s_path).read()
            ks = kernels_of(asm)
            names = demangle(list(ks))
            meta = {}
            for m in re.finditer(r"\.amdhsa_kernel\s+(\S+)(.*?)\.end_amdhsa_kernel", asm, flags=re.S):
                g = lambda key: (re.search(r"\.amdhsa_" + key + r"\s+(\d+)", m.group(2)) or [None, None])[1]
                meta[m.group(1)] = {"next_free_vgpr": g("next_free_vgpr"), "accum_offset": g("accum_offset"), "lds_bytes": g("group_segment_fixed_size"),
                                    "scratch_bytes": g("private_segment_fixed_size")}
            for sym, body in ks.items():
                loop = main_loop(body)
                entry = {"file": src, "whole_kernel": count(body), **meta.get(sym, {})}
                if loop:
                    entry["tile_loop"] = count(body[loop[0]:loop[1] + 1])
                    entry["tile_loop_valu_ops"] = valu_histogram(body[loop[0]:loop[1] + 1])
                    entry["tile_loop_instructions"] = loop[2]
                res["kernels"][names[sym]] = entry
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for name, e in sorted(res["kernels"].items()):
        tl = e.get("tile_loop", {})
        mf = sum(v for k, v in tl.items() if k.startswith("mfma:"))
        if mf:
            print(f"{name[:100]:100s} loop: mfma {mf:4d} valu {tl.get('valu', 0):4d} pk {tl.get('valu_packed', 0):4d} trans {tl.get('valu_trans', 0):3d} "
                  f"salu {tl.get('salu', 0):4d} lds {tl.get('lds', 0):3d} vmem {tl.get('vmem', 0):3d} bar {tl.get('barrier', 0)}")


if __name__ == "__main__":
    main()
