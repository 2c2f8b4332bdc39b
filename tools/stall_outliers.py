"""Attribute the stall outliers of the encoder chain's small GEMMs inside the training step (`gemm_direct*`
launches of 24-41 us mean with maxima of ~400 us).  Input: a rocprofv3 --kernel-trace CSV of bench.py.  For every gemm_direct* launch
whose duration exceeds 4x the median of its kernel name the report says: where in the step it started, on which queue (HIP stream), its
index in that queue's launch order, which conv kernels were running when it started and whether its END coincides (within 8 us) with the
end of one of them -- the signature of a work-group that could not be placed until a persistent conv work-group retired.

usage: stall_outliers.py kernel_trace.csv > report.txt"""
import collections
import csv
import re
import statistics
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0][:52]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if "adamw" in r["n"]]
    if len(adam) < 3:
        sys.exit("need at least three optimizer launches (steps) in the trace")
    by_name = collections.defaultdict(list)
    for r in rows[adam[0]:]:
        if r["n"].startswith("gemm_direct"):
            by_name[r["n"]].append(r["e"] - r["s"])
    med = {k: statistics.median(v) for k, v in by_name.items()}
    print("kernel".ljust(54), "launches  median us   p99 us    max us")
    for k, v in sorted(by_name.items()):
        v = sorted(v)
        print(k.ljust(54), f"{len(v):8d} {med[k] / 1e3:10.1f} {v[int(0.99 * (len(v) - 1))] / 1e3:8.1f} {v[-1] / 1e3:9.1f}")
    convs = [r for r in rows if r["n"].startswith(("conv", "wino"))]
    print("\noutliers (> 4 x median of their kernel), per step:")
    total = 0
    for si in range(len(adam) - 1):
        step = rows[adam[si] + 1:adam[si + 1] + 1]
        t0 = step[0]["s"]
        order = collections.defaultdict(int)
        for r in step:
            order[r["Queue_Id"]] += 1
            r["qi"] = order[r["Queue_Id"]]
        for r in step:
            if not r["n"].startswith("gemm_direct") or r["e"] - r["s"] <= 4 * med[r["n"]]:
                continue
            total += 1
            running = [c for c in convs if c["s"] <= r["s"] < c["e"]]
            ends_with = [c for c in convs if abs(c["e"] - r["e"]) <= 8000 and c["s"] < r["s"]]
            # the kernel launched just before it on the same queue (its dependency) and how long after that one's end it started
            prev = [p for p in step if p["Queue_Id"] == r["Queue_Id"] and p["qi"] == r["qi"] - 1]
            gap = (r["s"] - prev[0]["e"]) / 1e3 if prev else float("nan")
            print(f"  step {si}: +{(r['s'] - t0) / 1e3:7.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f} us (median {med[r['n']] / 1e3:5.1f})  queue {r['Queue_Id']} launch #{r['qi']:3d}  "
                  f"grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {r['n']}\n"
                  f"           after {prev[0]['n'] if prev else '-'} (gap {gap:.1f} us); running beside it at start: {', '.join(sorted({c['n'] for c in running})) or 'no conv kernel'}; "
                  f"ends within 8 us of: {', '.join(sorted({c['n'] for c in ends_with})) or 'nothing'}")
    steps = len(adam) - 1
    print(f"\n{total} outliers in {steps} steps ({total / steps:.1f} per step)")


if __name__ == "__main__":
    main()
