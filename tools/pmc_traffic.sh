#!/bin/bash
# HBM traffic per launch (run on the GPU box): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
# tools/pmc_workload.py, per-kernel means -> gpurun_out/pmc_traffic/{FETCH_SIZE,WRITE_SIZE}_per_kernel.csv + pmc_traffic.json
# (copy those into profiles/). FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads — the device copy in the
# workload is the calibration: it must come out at 402.7 MB read / 402.7 MB written), WRITE_SIZE is exact; unit KB = 1024 B.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o $c -- python $R/tools/pmc_workload.py > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$R")
from bench import kernel_source_stamp
LOGICAL = [("enc_edge", "enc_edge"), ("msg8_wave_kernel<false", "enc_msg"), ("msg8_wave_kernel<true", "dec_msg"),
           ("msg8_rp_kernel<SplitH2, false", "enc_msg_remainder"), ("msg8_rp_kernel<SplitH2, true", "dec_msg_remainder"),
           ("featurize", "featurize"), ("gather_rows_kernel", "gather_rows"), ("copyBuffer", "device_copy_calibration"),
           ("node_update", "node_update"), ("knn_kernel", "knn"), ("head8_split", "head"), ("node_proj", "node_proj")]
T, E = 16384, 16384 * 48
EB = E * 128 * 4
# algorithmic bytes per launch: edge tiles + node projections [T,256] + neighbour lists + small per-node arrays
ALG = {"enc_edge": 2 * EB + T * 256 * 4 + E * 4, "enc_msg": EB + T * 256 * 4 + E * 4 + T * 128 * 4 + T * 4 * 2,
       "dec_msg": EB + T * 256 * 4 + E * 4 + T * 128 * 4 + T * 4 * 2, "featurize": EB + E * 8 + T * 56,
       "gather_rows": E * 4 + T * 128 * 4 + EB, "device_copy_calibration": 2 * EB}
mean = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k in acc:                      # the runtime's copy kernel also moves the weights: keep the three 402.7 MB copies
        if "copyBuffer" in k: acc[k] = sorted(acc[k])[-3:]
    with open("$OUT/%s_per_kernel.csv" % c, "w") as o:
        o.write("Kernel_Name,Launches,Mean_%s_KB\n" % c)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            o.write('"%s",%d,%.1f\n' % (k.split("(")[0], len(v), sum(v) / len(v)))
            mean[(c, k)] = (sum(v) / len(v), len(v))
kern = {}
for (c, k), (m, n) in mean.items():
    for pat, name in LOGICAL:
        if pat in k:
            d = kern.setdefault(name, {"kernel_symbol": k.split("(")[0], "launches": n})
            d["fetch_bytes" if c == "FETCH_SIZE" else "write_bytes"] = m * 1024 * (2 if c == "FETCH_SIZE" else 1)
            break
for name, d in kern.items():
    d["traffic_bytes"] = d.get("fetch_bytes", 0) + d.get("write_bytes", 0)
    if name in ALG: d["algorithmic_bytes"] = ALG[name]; d["ratio"] = d["traffic_bytes"] / ALG[name]
json.dump({"kernel_source_stamp": kernel_source_stamp(), "units": "bytes per launch at T=16384 residues (64 x L=256); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section "
           "(gfx950 reports 1/2 of wide coalesced reads; confirmed on the 402.7 MB device copy), WRITE_SIZE exact (KB = 1024 B)",
           "kernels": kern}, open("$OUT/pmc_traffic.json", "w"), indent=1)
for n, d in kern.items(): print(n, d.get("traffic_bytes"), d.get("algorithmic_bytes"), d.get("ratio"))
PY
