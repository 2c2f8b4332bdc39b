"""The committed evidence under profiles/ is internally consistent: the traffic summary is what tools/pmc_traffic.py
derives from the committed PMC dumps, every configuration's bench line carries the fields the measurement contract asks for, and
the rocprofv3 kernel table of the same command agrees with the duration bench.py measured with its own HIP events."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
ROUND = "r04"
# compulsory HBM bytes per launch of the conv kernels at batch B (fp32 NCHW, u8 masks)
CONV_BYTES = {"conv2_fwd": lambda B: B * (32 * 64 * 64 * 4 + 64 * 32 * 32 * 5), "conv1_fwd": lambda B: B * (3 * 128 * 128 * 4 + 32 * 64 * 64 * 5)}


def bench(config):
    return json.load(open(os.path.join(PROF, f"{ROUND}_bench_config{config}.json")))


def test_pmc_traffic_is_reproducible_from_the_dumps(tmp_path):
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(PROF, f"{ROUND}_pmc_fetch_size.csv"),
                    os.path.join(PROF, f"{ROUND}_pmc_write_size.csv"), str(out)], check=True, capture_output=True)
    got, want = json.load(open(out)), json.load(open(os.path.join(PROF, f"{ROUND}_pmc_traffic_config3.json")))
    for k in ("conv2_fwd", "conv2_dgrad", "conv2_wgrad", "conv1_fwd", "conv1_wgrad"):
        assert abs(got[k] - want[k]) <= 1e-6 * want[k], k
    # HBM traffic can only exceed the compulsory bytes, and the XCD-aware strip mapping keeps it within 5 % of them:
    # conv2 forward at B = 512: 268 MB in + 168 MB out
    for k, f in CONV_BYTES.items():
        assert f(512) <= want[k] < 1.05 * f(512), (k, want[k], f(512))


@pytest.mark.parametrize("config", [1, 2, 3, 4, 5])
def test_bench_lines_have_the_contract_fields(config):
    d = bench(config)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and "workload" in d["config"] and d["config"]["baseline_config"] == config
    assert d["dtype"] == ("f64" if config == 1 else "f32") and d["scaling"] == "weak" and d["data"] == "synthetic"
    r = d["roofline"]
    # configs 2, 3, 5: the dominant kernel runs in the split-bf16 form, priced against the bf16 pipe's ceiling (2500 / 6 TFLOP/s of float32
    # products); config 1's float64 MLP grid against the f32 MFMA peak as before; config 4's dominant kernel is whichever section has
    # the largest total time per step -- the fused small-head attention backward (f32 MFMA 16x16x4) in round 3
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    split = config in (2, 3, 5) or (config == 4 and not r["kernel"].startswith("attn"))
    assert r["peak"] == (416.7 if split else 157.3)
    if config != 1:
        assert abs(r["frac_of_f32_mfma_peak"] - r["achieved"] / 157.3) < 1e-3
        if split:
            assert abs(r["executed_bf16_tflops"] - 6 * r["achieved"]) < 0.1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    if config != 1:
        # consistency: value = batch / step time
        B = d["config"]["per_gpu_batch"]
        assert B == {2: 256, 3: 512, 4: 512, 5: 4096}[config]
        assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
        assert abs(r["achieved"] - r["flops_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12) < 0.02 * r["achieved"]
        # only the dominant candidate is bracketed by events in the timed region; an untimed pass over all candidates chose it
        # (config 5, forward only, has a single candidate)
        if config != 5:
            cands = r["candidates_untimed_pass"]
            assert r["kernel"] == max(cands, key=lambda k: cands[k]["ms_per_step"]) and "largest total time per step" in r["dominant_chosen_by"]


@pytest.mark.parametrize("config", [2, 3, 5])
def test_rocprof_kernel_table_agrees_with_the_bench_line(config):
    """The measurement contract: the rocprofv3 --kernel-trace --stats average of the roofline kernel agrees with the duration
    bench.py measured with its own HIP events (profiles are collected by tools/profile_round.sh with --no-isolated, so every
    launch in the table is an in-step one), and the PMC traffic quoted in the line is the committed pass of that configuration."""
    d = bench(config)
    r = d["roofline"]
    # (round 4: the kernel is templated over its geometry -- "conv_b3_kernel<0, B3Geom<32, 64, 64> >")
    # (round 4, late: training plans with an encoder run the forward as "conv_b3p_kernel<0, B3Geom<32, 64, 64>, true>")
    pattern = {"conv2_wgrad": "conv_b3_wgrad_", "conv2_fwd": "conv_b3_kernel<0" if config in (2, 5) else "conv_b3p_kernel<0",
               "conv2_dgrad": "conv_b3_kernel<1"}[r["kernel"]]
    rows = [x for x in csv.DictReader(open(os.path.join(PROF, f"{ROUND}_kernel_stats_config{config}.csv"))) if pattern in x["Name"]]
    assert len(rows) == 1
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    # config 5 (B = 4096 inference): 13 launches of 3.2 .. 5.6 ms depending on what the encoder stream runs beside them -- two short
    # runs' averages agree to ~15 %; the training configurations (tens of launches of one repeating step) to 5-8 % (the bench line is a
    # separate run of the same command on the same box: round 3's pair differs by 6 %, which is why the band went from 5 to 8 % in round 3)
    tol = 0.15 if config == 5 else 0.08
    assert abs(avg_ms - r["ms_per_launch"]) <= tol * r["ms_per_launch"], (avg_ms, r["ms_per_launch"])
    t = json.load(open(os.path.join(PROF, f"{ROUND}_pmc_traffic_config{config}.json")))
    assert r["traffic_source"] == f"{ROUND}_pmc_traffic_config{config}.json" and t["config"] == config
    assert abs(t[r["kernel"]] - r["traffic"]) <= 1e-6 * r["traffic"]
    # traffic close to the algorithmic bytes: nothing is re-read from HBM
    B = d["config"]["per_gpu_batch"]
    # (round 4: forward-only plans -- config 5 -- keep no pooling decisions: 4 bytes per pooled element instead of 5)
    conv2_bytes = B * (32 * 64 * 64 * 4 + 64 * 32 * 32 * (4 if config == 5 else 5))
    assert conv2_bytes <= r["traffic"] < 1.12 * conv2_bytes, (r["traffic"], conv2_bytes)


def test_config4_dominant_kernel_is_chosen_by_total_time_and_agrees_with_rocprof():
    """Config 4 (F = 2048): the roofline kernel is the section with the largest TOTAL time per step among every encoder GEMM / attention
    kernel and the conv2 kernels (not a kernel picked by fiat), the line says how it was chosen and how often it runs,
    and the rocprofv3 table of the same command agrees with its per-launch time."""
    d = bench(4)
    r = d["roofline"]
    cands = r["candidates_untimed_pass"]
    assert len(cands) >= 15 and r["kernel"] == max(cands, key=lambda k: cands[k]["ms_per_step"])
    assert r["launches_per_step"] >= 1 and abs(r["ms_per_step"] - r["ms_per_launch"] * r["launches_per_step"]) <= 0.02 * r["ms_per_step"]
    assert 0.05 < r["share_of_step"] < 0.6 and "largest total time per step" in r["dominant_chosen_by"]
    # round 1: 19.5 ms; round 2: 9.51 ms; round 3: wide LayerNorm + single-sweep attention backward
    assert d["ms_per_step"] < 8.9
    pattern = {"attn_bwd": "attn_small_bwd1_kernel", "attn_fwd": "attn_small_fwd_kernel"}.get(r["kernel"])
    if pattern is not None:
        rows = [x for x in csv.DictReader(open(os.path.join(PROF, f"{ROUND}_kernel_stats_config4.csv"))) if pattern in x["Name"]]
        assert len(rows) == 1
        avg_ms = float(rows[0]["AverageNs"]) * 1e-6
        # The bench's HIP events bracket the launch on its stream: they include the time the kernel's work-groups (118 KB of LDS each)
        # wait for a CU that the image branch's persistent conv work-groups (2 x 66 KB) hold; rocprofv3's duration starts at the first
        # wave.  So the event time is the larger one, by up to the length of a conv strip loop's tail (~0.05 ms of 0.27).
        assert avg_ms <= r["ms_per_launch"] * 1.03 and r["ms_per_launch"] - avg_ms <= 0.30 * r["ms_per_launch"], (avg_ms, r["ms_per_launch"])
    rows = [x for x in csv.DictReader(open(os.path.join(PROF, f"{ROUND}_kernel_stats_config4.csv"))) if "layernorm_fwd_wide_kernel" in x["Name"]]
    assert len(rows) == 1 and float(rows[0]["AverageNs"]) < 25e3, "the work-group-per-row LayerNorm ran in the profiled command (39 us per call in round 2)"
