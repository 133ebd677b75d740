#!/bin/bash
# SQ counter passes over the bench batch (run on the GPU box): per-kernel sums -> gpurun_out/pmc_sq_summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $R/tools/pmc_workload.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"): n[(k, r["Counter_Name"])] += 1
with open("$R/gpurun_out/pmc_sq_summary.txt", "w") as o:
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        o.write(k + "\n")
        for c, v in sorted(d.items()): o.write("   %-28s %.4g\n" % (c, v))
# effective shader clock per kernel = GRBM_GUI_ACTIVE (per dispatch, max over the XCD instances) / kernel duration
import re
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p4/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$R/gpurun_out/pmc_sq_summary.txt", "a") as o:
    o.write("\n# kernel: launches, avg duration us (pass 4), GRBM_GUI_ACTIVE sum, GRBM_COUNT sum\n")
    for k, d in dur.items():
        o.write("%-60s %4d %9.1f %.4g %.4g\n" % (k, len(d), sum(d) / len(d) / 1e3, agg[k].get("GRBM_GUI_ACTIVE", 0), agg[k].get("GRBM_COUNT", 0)))
print(open("$R/gpurun_out/pmc_sq_summary.txt").read()[:9000])
PY
