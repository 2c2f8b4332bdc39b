"""HBM traffic per launch of the conv kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json [CONFIG]
The output records a hash of the kernel sources it was collected with (bench.py quotes the traffic only for those sources).
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters
are in KiB; on gfx950 FETCH_SIZE counts 64-byte units of the 128-byte wide reads as one, so fetched bytes = raw x 1024 x 2.
"""
import collections
import csv
import hashlib
import json
import os
import sys


def sources_sha():
    """Same hash as bench.py: sources_sha()."""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bbbp-multi-modal-deep-ensemble-framework_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("conv") or f == "common.h":          # the sources of the kernels whose traffic is quoted (conv*.hip)
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

KERNELS = {            # a section maps to whichever of its kernels ran (direct f32, Winograd or split-bf16 form)
    "conv2_fwd": ("conv3x3_kernel<32, 64, 64, 0>", "wino_conv_kernel<0>", "conv_b3_kernel<0", "conv_b3p_kernel<0"),
    "conv2_dgrad": ("conv3x3_kernel<64, 32, 64, 1>", "wino_conv_kernel<1>", "conv_b3_kernel<1", "conv_b3p_kernel<1"),
    "conv2_wgrad": ("conv_wgrad32_kernel<64, 64, 32, 64>", "conv_b3_wgrad_kernel", "conv_b3_wgrad_sp_kernel"),
    "conv1_fwd": ("conv3x3_kernel<3, 32, 128, 0>", "conv1_b3_fwd_kernel", "conv1_b3p_fwd_kernel"),
    "conv1_wgrad": ("conv_wgrad3_kernel<128>", "conv_b3_wgrad3_kernel"),
}


def mean_counter(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key, pat in KERNELS.items():
            if any(p_ in r["Kernel_Name"] for p_ in ((pat,) if isinstance(pat, str) else pat)):
                acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if v}


def main():
    fetch = mean_counter(sys.argv[1], "FETCH_SIZE")
    write = mean_counter(sys.argv[2], "WRITE_SIZE")
    out = {"_detail": {}}
    for k in KERNELS:
        if k in fetch and k in write:
            fb, wb = fetch[k] * 1024 * 2, write[k] * 1024
            out[k] = fb + wb
            out["_detail"][k] = dict(fetch_bytes_corrected=fb, write_bytes=wb, hbm_bytes=fb + wb, raw_FETCH_SIZE_KiB=fetch[k],
                                     raw_WRITE_SIZE_KiB=write[k])
    out["sources_sha"] = sources_sha()
    out["config"] = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    out["_note"] = ("per launch at the configuration's batch size; FETCH_SIZE KiB x1024 x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM), "
                    "WRITE_SIZE KiB x1024; separate --pmc passes")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print({k: round(v / 1e6, 1) for k, v in out.items() if k in KERNELS}, "MB per launch")


if __name__ == "__main__":
    main()
