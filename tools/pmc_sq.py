"""Matrix-pipe utilisation of the conv kernels from one rocprofv3 --pmc pass over SQ counters.

    pmc_sq.py counter_collection.csv out.txt

SQ_VALU_MFMA_BUSY_CYCLES is 64 cycles per v_mfma_f32_32x32x2_f32 (32 per v_mfma_f32_32x32x16_bf16) summed over all SIMDs;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 4 * CUs).  sustained_ghz = GUI_ACTIVE / 8 /
kernel time is not available here (no durations in a counter pass): see the clock the bench line reports.  Counter collection
serialises the dispatches."""
import collections
import csv
import sys

KERNELS = {"conv2_fwd (winograd)": "wino_conv_kernel<0>", "conv2_dgrad (winograd)": "wino_conv_kernel<1>", "conv2_wgrad (f32)": "conv_wgrad32_kernel",
           "conv1_fwd": "conv3x3_kernel<3, 32", "conv1_wgrad (f32)": "conv_wgrad3_kernel", "conv2_fwd (direct f32)": "conv3x3_kernel<32, 64",
           "conv2_dgrad (direct f32)": "conv3x3_kernel<64, 32", "conv2_fwd (split-bf16)": "conv_b3_kernel<0>", "conv2_dgrad (split-bf16)": "conv_b3_kernel<1>",
           "conv2_wgrad (split-bf16)": "conv_b3_wgrad_kernel", "conv2_wgrad (split-bf16, 2:4 structured-sparse MFMA)": "conv_b3_wgrad_sp_kernel", "conv1_wgrad (split-bf16)": "conv_b3_wgrad3_kernel", "imgfc / large GEMM NT (split-bf16)": "gemm_b3_kernel<0>",
           "large GEMM NN (split-bf16)": "gemm_b3_kernel<1>", "large GEMM TN (split-bf16)": "gemm_b3_kernel<2>"}
CUS, XCDS = 256, 8


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(sys.argv[2], "w") as out:
        out.write("# " + __doc__.strip().replace("\n", "\n# ") + "\n")
        for key, v in acc.items():
            m = {c: sum(x) / len(x) for c, x in v.items()}
            util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / XCDS * 4 * CUS) if m.get("GRBM_GUI_ACTIVE") else float("nan")
            out.write(f"{key}: mfma_pipe_utilisation={util:.3f} " + " ".join(f"{c}={x:.4g}" for c, x in sorted(m.items())) + "\n")


if __name__ == "__main__":
    main()
