"""Step time of the other BASELINE configs on one GPU (not the headline bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bbbp_amd
from bbbp_amd.optim import AdamW
dev = torch.device("cuda:0")
def run(F, B, train, steps=5):
    torch.manual_seed(0)
    m = bbbp_amd.MixedInputModel(F, 128).to(dev)
    m.train(train)
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
    def step():
        if train:
            torch.nn.MSELoss()(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
        else:
            with torch.no_grad(): m(fp, img)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"F={F} B={B} {'train' if train else 'eval '}: {dt*1e3:8.2f} ms/step  {B/dt:10.0f} molecules/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del m, opt; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
run(167, 512, True)
run(167, 256, True)
run(2048, 512, True)
run(167, 4096, False)


def run_variant(name, make, B, steps=20, fused_opt=False):
    """The other model scripts (bbbp_amd.variants), composed from per-op autograd nodes on the same HIP kernels."""
    torch.manual_seed(0)
    m = make().to(dev).train()
    opt = (AdamW if fused_opt else torch.optim.AdamW)(m.parameters(), lr=1e-4, weight_decay=1e-5)      # fused_opt: the one-launch AdamW of bench.py
    F = m.fingerprint_size if hasattr(m, "fingerprint_size") else 167
    fp = torch.randn(B, F, device=dev); img = torch.randn(B, 49152, device=dev); y = torch.randn(B, device=dev)
    def step():
        torch.nn.MSELoss()(m(fp, img).squeeze(), y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name} B={B} train: {dt*1e3:8.2f} ms/step  {B/dt:10.0f} molecules/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del m, opt; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()


if os.environ.get("BBBP_TIME_VARIANTS", "1") != "0":
    from bbbp_amd import variants
    run_variant("wide/deep (12-layer encoder, 3-stage CNN)", lambda: variants.WideDeepMixedInputModel(167, 128), 256)
    # exact-global-batch mode at world size 1 (no process group: the collectives are identities): what the per-op composition costs
    # against the fused engine on the same arithmetic
    run_variant("exact-global-batch mode, one rank (fused engine, round 3; fused AdamW as in bench.py)", lambda: variants.ExactBatchMixedInputModel(167, 128), 512, fused_opt=True)
    run_variant("exact-global-batch mode, one rank (fused engine, round 3; torch.optim.AdamW)", lambda: variants.ExactBatchMixedInputModel(167, 128), 512)
    run_variant("exact-global-batch mode, one rank (per-op autograd, rounds 1-2)", lambda: variants.PerOpExactBatchMixedInputModel(167, 128), 512)
