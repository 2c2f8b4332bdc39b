"""Per-queue timeline of one training step from a rocprofv3 --kernel-trace CSV: for each HIP stream's queue, the busy time
(sum of kernel durations), the idle gaps between consecutive kernels, and the slowest kernels.  Usage: trace_timeline.py CSV [step]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # steps are delimited by the AdamW launch (one per step)
    adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 2
    lo, hi = adam[which] + 1, adam[which + 1] + 1
    step = rows[lo:hi]
    t0, t1 = step[0]["s"], max(r["e"] for r in step)
    print(f"step {which}: {len(step)} kernels, {(t1 - t0) / 1e3:.1f} us wall")
    byq = collections.defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    for q, ks in sorted(byq.items()):
        busy = sum(r["e"] - r["s"] for r in ks)
        gaps = [max(0, b["s"] - a["e"]) for a, b in zip(ks, ks[1:])]
        print(f"queue {q}: {len(ks)} kernels, first start +{(ks[0]['s'] - t0) / 1e3:.1f} us, last end +{(ks[-1]['e'] - t0) / 1e3:.1f} us, "
              f"busy {busy / 1e3:.1f} us, gaps {sum(gaps) / 1e3:.1f} us (mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f})")
        agg = collections.defaultdict(lambda: [0, 0])
        for r in ks:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += r["e"] - r["s"]
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"    {t / 1e3:9.1f} us  {n:4d} x {t / n / 1e3:8.2f} us  {name}")




def sequence(path, which=None, lo_name="head_a_kernel", count=40):
    """Kernel sequence of the queue that runs `lo_name`, from its first launch in the step: start offset, duration, gap."""
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    which = len(adam) - 2 if which is None else which
    step = rows[adam[which] + 1:adam[which + 1] + 1]
    t0 = step[0]["s"]
    first = next(r for r in step if lo_name in r["Kernel_Name"])
    ks = [r for r in step if r["Queue_Id"] == first["Queue_Id"] and r["s"] >= first["s"]][:count]
    prev = None
    for r in ks:
        gap = (r["s"] - prev["e"]) / 1e3 if prev else 0.0
        print(f"  +{(r['s'] - t0) / 1e3:8.1f} us  dur {(r['e'] - r['s']) / 1e3:8.1f}  gap {gap:7.1f}  {short(r['Kernel_Name'])}")
        prev = r


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "seq":
        sequence(sys.argv[1], int(sys.argv[2]))
    else:
        main()
