#!/usr/bin/env python3
"""Headline benchmark: molecules/s of the 3-branch MixedInputModel training step (forward + MSE + backward +
fused AdamW; BASELINE.json config 3: MACCS width F=167, 3x128x128 images, batch 512 per GPU, train mode) on N GPUs
of one node, one process per GPU, gradients all-reduced over RCCL when N > 1 (weak scaling: 512 molecules per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the build prompt) with two extra objects:
  roofline     -- the dominant kernel (a conv2 implicit-GEMM), timed with HIP events on its own stream inside the
                  timed region; achieved = algorithmic FLOPs per launch / mean launch time; peak = 157.3 TFLOP/s
                  (dense f32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle (a port of the reference arithmetic, oracle/reference_cpu.py) timed on the host
                  cores for the same step on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

F_DIM, BATCH, IMG_FLAT = 167, 512, 3 * 128 * 128
PEAK_F32_MFMA_TFLOPS = 157.3


def fwd_flops(B, F, L=6, dff=2048):
    """algorithmic forward FLOPs (2 * MACs) per batch."""
    enc = L * (2 * B * F * 3 * F + 4 * B * B * F + 2 * B * F * F + 4 * B * F * dff)
    conv1 = 2 * B * 32 * 128 * 128 * 27
    conv2 = 2 * B * 64 * 64 * 64 * 288
    return dict(encoder=enc, fp_fc=2 * B * F * 128, conv1=conv1, conv2=conv2, img_fc=2 * B * 65536 * 128,
                fusion=2 * B * 4 * (256 * 128 + 128), head=2 * B * (256 * 256 + 256 * 128 + 128 * 64 + 64))


def synthetic_b3db(n, F, seed, device):
    """Bernoulli(0.25) MACCS bits (bit 0 unused) column-standardised; white images with ~6 % dark
    bond pixels on three equal channels, standardised; labels N(-0.1, 0.8^2) clipped to [-2, 1.7]."""
    g = torch.Generator().manual_seed(seed)
    bits = (torch.rand(n, F, generator=g) < 0.25).float()
    bits[:, 0] = 0
    fp = (bits - bits.mean(0)) / bits.std(0).clamp_min(1e-6)
    fp[:, 0] = 0
    plane = torch.ones(n, 128 * 128)
    dark = torch.rand(n, 128 * 128, generator=g) < 0.06
    plane[dark] = torch.rand(int(dark.sum()), generator=g) * 0.5
    img = plane.repeat(1, 3)
    img = (img - img.mean()) / img.std()
    y = (torch.randn(n, generator=g) * 0.8 - 0.1).clamp(-2.0, 1.7)
    return fp.to(device), img.to(device), y.to(device)


def cpu_baseline(fp, img, y, state, iters=3):
    """Time the CPU oracle for the same train step (forward + MSE + backward + AdamW) on the host cores."""
    from oracle import reference_cpu as oracle
    # a 1-GPU box shares a 256-thread host: its CPU share is 16 cores, and oversubscribing costs 20x
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in state.items()}
    keys = [k for k, v in p.items() if v.requires_grad]
    m = {k: torch.zeros_like(p[k]) for k in keys}
    v2 = {k: torch.zeros_like(p[k]) for k in keys}
    fp, img, y = fp.cpu(), img.cpu(), y.cpu()
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        for k in keys:
            p[k].grad = None
        loss = oracle.mse_loss(oracle.mixed_input_forward(p, fp, img, training=True, bn_state={}), y)
        loss.backward()
        with torch.no_grad():
            for k in keys:
                oracle.adamw_step(p[k], p[k].grad, m[k], v2[k], it + 1)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / iters
    return dict(value=round(fp.shape[0] / t, 2), unit="molecules/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{iters} steps of B={fp.shape[0]} forward+MSE+backward+AdamW after 1 warm-up ({t:.2f} s/step), "
                       "oracle/reference_cpu.py on torch CPU fp32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 0.75 s of timed work: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the untimed overlap-off leg (profiles: every launch of the rocprofv3 kernel table is then an in-step one)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Rehearsal knob (not used by the driver): BBBP_BENCH_BACKEND=gloo lets N ranks share the GPUs that exist
    # (a 1-GPU box) to exercise the multi-rank control flow; the real runs use RCCL, one GPU per rank.
    backend = os.environ.get("BBBP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import bbbp_amd
    from bbbp_amd import _lib
    from bbbp_amd import distributed as D
    from bbbp_amd.optim import AdamW

    torch.manual_seed(20250113)           # same initial weights on every rank
    model = bbbp_amd.MixedInputModel(F_DIM, 128).to(dev).train()
    opt = AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    crit = bbbp_amd.MSELoss()             # nn.MSELoss semantics, value + gradient in one kernel
    fp, img, y = synthetic_b3db(2 * BATCH, F_DIM, 20250113 + rank, dev)
    params = list(model.parameters())
    # N > 1: the image-FC weight gradient (62 % of the bytes) is all-reduced under the rest of the backward pass, the
    # remainder after it (distributed.OverlappedGradAllReduce); BBBP_BENCH_PLAIN_ALLREDUCE=1 selects the single collective
    reducer = None
    if world > 1 and os.environ.get("BBBP_BENCH_PLAIN_ALLREDUCE", "0") != "1":
        reducer = D.OverlappedGradAllReduce(model)

    opt_events = []                       # (start, end) around the optimizer step, only while `time_opt` is set (untimed pass)
    time_opt = [False]

    def step(i, collective=True):
        s = (i % 2) * BATCH
        out = model(fp[s:s + BATCH], img[s:s + BATCH]).squeeze()
        loss = crit(out, y[s:s + BATCH])
        loss.backward()
        if time_opt[0]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            opt.step(grad_scale=1.0 / world)
            e1.record()
            opt_events.append((e0, e1))
            opt.zero_grad(set_to_none=True)
            return loss
        if world > 1 and collective:
            # ONE RCCL sum over xGMI when the gradients are one flat buffer (they are); 1/world folded into AdamW
            if reducer is not None:
                reducer(params, average=False)
            else:
                D.allreduce_gradients(params, average=False)
        opt.step(grad_scale=1.0 / world)
        opt.zero_grad(set_to_none=True)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    L = _lib.lib()
    nsec = L.bbbp_profile_num_sections()
    names = [L.bbbp_profile_section_name(i).decode() for i in range(nsec)]
    # HIP events on the launch stream, recorded INSIDE the timed region, around the three conv2 kernels only (the
    # candidates for the dominant kernel); the full per-section breakdown comes from an untimed pass below
    L.bbbp_profile_select(sum(1 << i for i, n in enumerate(names) if n.startswith("conv2_")))
    L.bbbp_profile_enable(1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    ms_sum = (ctypes.c_float * nsec)()
    cnt = (ctypes.c_int * nsec)()
    _lib.check(L.bbbp_profile_collect(ms_sum, cnt), "bbbp_profile_collect")     # events of the timed region itself
    L.bbbp_profile_enable(0)
    sections = {L.bbbp_profile_section_name(i).decode(): ms_sum[i] / cnt[i] for i in range(nsec) if cnt[i]}
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    L.bbbp_profile_select(0)
    # outside the timed region: every section with the overlap on (where the step goes) ...
    if rank == 0:
        L.bbbp_profile_enable(1)
        time_opt[0] = True
        for i in range(5):
            step(i, collective=False)        # rank 0 only: no collective here, the other ranks wait in the barrier below
        time_opt[0] = False
        torch.cuda.synchronize()
        _lib.check(L.bbbp_profile_collect(ms_sum, cnt), "bbbp_profile_collect")
        L.bbbp_profile_enable(0)
        for i in range(nsec):
            if cnt[i] and names[i] not in sections:
                sections[names[i]] = ms_sum[i] / cnt[i]
    # ... and the same kernels with the branch overlap off, i.e. each conv kernel alone on the GPU
    isolated = {}
    clock = {}
    if rank == 0 and not args.no_isolated:
        old = L.bbbp_set_overlap(0)
        for i in range(2):
            step(i, collective=False)
        L.bbbp_profile_enable(1)
        for i in range(5):
            step(i, collective=False)
        torch.cuda.synchronize()
        _lib.check(L.bbbp_profile_collect(ms_sum, cnt), "bbbp_profile_collect")
        L.bbbp_profile_enable(0)
        L.bbbp_set_overlap(old)
        isolated = {L.bbbp_profile_section_name(i).decode(): ms_sum[i] / cnt[i] for i in range(nsec) if cnt[i]}
        # the last forward / data-gradient conv launch of a step is conv2's data gradient: its work-group 0 counted
        # shader cycles and 100 MHz wall ticks (include/bbbp_hip.h: bbbp_conv_last_clock)
        cyc, ticks = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _lib.check(L.bbbp_conv_last_clock(ctypes.byref(cyc), ctypes.byref(ticks)), "bbbp_conv_last_clock")
        clock = dict(cycles=int(cyc.value), ghz=(cyc.value / (ticks.value * 10.0)) if ticks.value else None)
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * BATCH * args.steps / elapsed
        fl = fwd_flops(BATCH, F_DIM)
        conv2 = fl["conv2"]
        cand = {k: sections[k] for k in ("conv2_fwd", "conv2_dgrad", "conv2_wgrad") if k in sections}
        # conv2's forward / data gradient may run as Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36.  The
        # roofline line is the slowest conv2 kernel priced at the algorithmic (direct) flop count of SURVEY.md 8(d); for the
        # Winograd kernels `winograd` below also gives the MFMA flops they actually execute.
        wmask = L.bbbp_get_conv_winograd()
        wino = {k: bool(wmask & bit) for k, bit in (("conv2_fwd", 1), ("conv2_dgrad", 2))}
        roofline = None
        if cand:
            dom = max(cand, key=cand.get)
            achieved = conv2 / (cand[dom] * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(dom)
            roofline = dict(bound="mfma", kernel=dom, achieved=round(achieved, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic,
                            flops_per_launch=conv2, ms_per_launch=round(cand[dom], 4),
                            note="timed inside the training step, where the kernel shares the GPU with the fingerprint "
                                 "branch's side-stream kernels; *_isolated = same kernel, overlap off, after the timed region",
                            ms_per_launch_isolated=round(isolated.get(dom, 0.0), 4),
                            frac_isolated=round(conv2 / (isolated[dom] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if isolated.get(dom) else None,
                            sections_ms={k: round(v, 4) for k, v in sections.items()})
            roofline["algorithm"] = "winograd F(2x2,3x3) f32" if wino.get(dom) else "direct implicit GEMM f32"
            if any(wino.values()):
                roofline["winograd"] = {
                    k: dict(ms_per_launch=round(sections[k], 4), ms_per_launch_isolated=round(isolated.get(k, 0.0), 4),
                            executed_flops_per_launch=conv2 * 16 // 36,
                            direct_equivalent_tflops=round(conv2 / (sections[k] * 1e-3) / 1e12, 2),
                            executed_frac_of_peak=round(conv2 * 16 / 36 / (sections[k] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
                    for k in wino if wino[k] and k in sections}
            if clock.get("ghz"):
                # one v_mfma_f32_32x32x2_f32 = 4096 flop and occupies its SIMD's matrix pipe for 64 cycles
                n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
                executed = conv2 * 16 / 36 if wino["conv2_dgrad"] else conv2
                roofline["conv2_dgrad_isolated_clock"] = dict(
                    sustained_ghz=round(clock["ghz"], 3), kernel_cycles=clock["cycles"],
                    mfma_pipe_busy=round(executed / 4096 * 64 / n_simd / clock["cycles"], 4),
                    note="shader clock while the kernel runs alone; the 157.3 TFLOP/s peak assumes 2.4 GHz")
        total_flops = sum(fl.values()) * 3 - fl["conv1"]          # bwd = 2 * fwd - conv1 dgrad
        result = {
            "metric": "molecules/sec fwd+bwd (3-branch ensemble, B=512)", "value": round(value, 1), "unit": "molecules/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MixedInputModel F=167 (MACCS) + 3x128x128 image, 6-layer encoder + 2-stage CNN + fusion + "
                                   "BN head; forward + MSE + backward + fused AdamW, train mode (dropout 0.1)",
                       "global_batch": BATCH * world, "per_gpu_batch": BATCH, "parallelism": f"dp{world}",
                       "gflop_per_step_per_gpu": round(total_flops / 1e9, 1)},
            "model_tflops_per_gpu": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
            # the metric's "+ optimizer step reported separately": fused AdamW over the flat parameter buffer (one launch,
            # 16 B read + 12 B written per parameter), included in ms_per_step
            "optimizer_ms_per_step": round(sum(a.elapsed_time(b) for a, b in opt_events) / len(opt_events), 4) if opt_events else None,
            "final_loss": round(float(loss.detach()), 5),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(fp[:BATCH], img[:BATCH], y[:BATCH], model.state_dict())
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
