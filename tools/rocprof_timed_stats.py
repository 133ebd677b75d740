"""Per-kernel statistics of the TIMED launches only, from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --no-extras`.
rocprofv3 --stats averages over every launch of the process — the ~100 clock warm-up forwards and the W warm-up steps included
(the first launches run at a lower clock: round 2's CSV sat 4 % above the HIP-event figure of the same run). This keeps the LAST
K x (launches per step) launches of every kernel, i.e. exactly the region bench.py times.
    python tools/rocprof_timed_stats.py <kernel_trace.csv> <steps> > profiles/rNN_bench_kernel_stats_timed.csv"""
import collections
import csv
import sys

PER_STEP = {"knn_kernel": 1, "featurize": 1, "msg8_wave_kernel<false": 3, "msg8_wave_kernel<true": 3,
            "msg8_rp_kernel<SplitH2, false": 3, "msg8_rp_kernel<SplitH2, true": 3, "enc_edge8": 3, "node_update8": 6, "head8": 1}


def main(path, steps):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel_Name", "TimedLaunches", "AvgNs", "MinNs", "MaxNs", "AllLaunches", "AvgNsAllLaunches"])
    out = []
    for name, v in rows.items():
        per = next((n for pat, n in PER_STEP.items() if pat in name), None)
        if per is None:
            continue
        v.sort()
        d = [x[1] for x in v[-per * steps:]]
        out.append((sum(d), name.split("(")[0], len(d), sum(d) / len(d), min(d), max(d), len(v), sum(x[1] for x in v) / len(v)))
    for _, *r in sorted(out, reverse=True):
        w.writerow([r[0], r[1], "%.1f" % r[2], r[3], r[4], r[5], "%.1f" % r[6]])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
