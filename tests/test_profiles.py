"""The committed evidence under profiles/ is internally consistent: the traffic summary is what tools/pmc_traffic.py
derives from the committed PMC dumps, and the bench line carries the fields the measurement contract asks for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_pmc_traffic_is_reproducible_from_the_dumps(tmp_path):
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(PROF, "r01_pmc_fetch_size.csv"),
                    os.path.join(PROF, "r01_pmc_write_size.csv"), str(out)], check=True, capture_output=True)
    got, want = json.load(open(out)), json.load(open(os.path.join(PROF, "r01_pmc_traffic.json")))
    for k in ("conv2_fwd", "conv2_dgrad", "conv2_wgrad", "conv1_fwd", "conv1_wgrad"):
        assert abs(got[k] - want[k]) <= 1e-6 * want[k], k
        # HBM traffic can only exceed the compulsory bytes; conv2 forward: 268 MB in + 168 MB out
    assert got["conv2_fwd"] >= 436e6 and got["conv2_fwd"] < 2 * 436e6


def test_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(PROF, "r01_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["n_gpus"] == 1 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # consistency: value = batch / step time
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3


def test_rocprof_kernel_table_agrees_with_the_bench_line():
    """The measurement contract: the rocprofv3 --kernel-trace --stats average of the roofline kernel agrees with the duration
    bench.py measured with its own HIP events (profiles are collected by tools/profile_round.sh with --no-isolated, so every
    launch in the table is an in-step one)."""
    import csv
    d = json.load(open(os.path.join(PROF, "r01_bench_n1.json")))
    r = d["roofline"]
    pattern = {"conv2_wgrad": "conv_wgrad32_kernel", "conv2_fwd": "wino_conv_kernel<0>", "conv2_dgrad": "wino_conv_kernel<1>"}[r["kernel"]]
    rows = [x for x in csv.DictReader(open(os.path.join(PROF, "r01_kernel_stats.csv"))) if pattern in x["Name"]]
    assert len(rows) == 1
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - r["ms_per_launch"]) <= 0.05 * r["ms_per_launch"], (avg_ms, r["ms_per_launch"])
    # and the PMC traffic of that kernel is what the bench line reports
    t = json.load(open(os.path.join(PROF, "r01_pmc_traffic.json")))
    assert abs(t[r["kernel"]] - r["traffic"]) <= 0.05 * r["traffic"]
