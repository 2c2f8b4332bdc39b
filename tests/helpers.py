"""Shared test helpers: golden loading, seeded inputs, tolerant comparisons."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-4      # BASELINE.json north_star: predictions within 1e-4 relative of the reference CPU path


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def synth_inputs(seed, B, F, I):
    """Same seeded inputs as tools/make_golden.py:synth_inputs."""
    g = torch.Generator().manual_seed(seed)
    fp = torch.randn(B, F, generator=g)
    img = torch.randn(B, I, generator=g)
    y = torch.randn(B, generator=g) * 0.8 - 0.1
    return fp, img, y


def check_param_checksums(g, state_dict):
    assert list(g["meta/keys"]) == list(state_dict.keys())
    for k, v in state_dict.items():
        if v.dtype.is_floating_point:
            s = g["param/" + k]
            d = v.detach().double().cpu()
            assert abs(float(d.sum()) - s[0]) <= 1e-9 * max(1.0, abs(s[0])), k
            assert abs(float(d.abs().sum()) - s[1]) <= 1e-9 * max(1.0, s[1]), k


def assert_close(actual, expected, rtol=RTOL, atol_frac=1e-5, what=""):
    """|a - e| <= rtol * |e| + atol_frac * max|e| elementwise (the floor absorbs cancellation noise)."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    scale = float(np.max(np.abs(e))) if e.size else 0.0
    err = np.abs(a - e)
    tol = rtol * np.abs(e) + atol_frac * scale + 1e-30
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{e.size} elements out of tolerance; max err {err.max():.3e} "
                           f"at scale {scale:.3e}; worst ratio {(err / tol).max():.2f}")


def check_summary(g, prefix, tensor, rtol=RTOL, atol_frac=2e-5):
    """Compare a tensor with the (stats, head, samp) summary stored by tools/make_golden.py."""
    d = tensor.detach().double().cpu().flatten()
    stats = g[prefix + "/stats"]
    l2 = float(d.norm())
    assert abs(l2 - stats[2]) <= (2 * rtol + atol_frac) * max(stats[2], 1e-30) + 1e-12, f"{prefix}: l2 {l2} vs {stats[2]}"
    scale = stats[2] / max(np.sqrt(d.numel()), 1.0)          # rms of the golden tensor
    head = g[prefix + "/head"]
    idx = g[prefix + "/idx"]
    for got, exp, nm in ((d[:64].numpy(), head, "head"), (d[torch.from_numpy(idx)].numpy(), g[prefix + "/samp"], "samp")):
        err = np.abs(got - exp)
        tol = rtol * np.abs(exp) + atol_frac * max(scale, float(np.max(np.abs(exp)))) + 1e-30
        assert (err <= tol).all(), f"{prefix}/{nm}: max err {err.max():.3e}, worst ratio {(err / tol).max():.2f}"


def check_summary_adam(g, prefix, tensor, lr, steps, rtol=1e-5, min_frac=0.9, tight_lr_frac=0.02):
    """Parameters after AdamW steps.  Where the exact gradient is ~0 (|g| <~ eps) AdamW turns rounding noise into
    +-lr moves, so a minority of elements may differ by up to ~2*lr*steps between any two implementations; the
    rest must agree tightly."""
    d = tensor.detach().double().cpu().flatten()
    idx = g[prefix + "/idx"]
    got = np.concatenate([d[:64].numpy(), d[torch.from_numpy(idx)].numpy()])
    exp = np.concatenate([g[prefix + "/head"], g[prefix + "/samp"]])
    err = np.abs(got - exp)
    assert (err <= 2.5 * lr * steps + rtol * np.abs(exp)).all(), f"{prefix}: max err {err.max():.3e}"
    tight = err <= rtol * np.abs(exp) + tight_lr_frac * lr
    assert tight.mean() >= min_frac, f"{prefix}: only {tight.mean():.2%} of sampled elements agree tightly"


def assert_close_or_as_accurate_as_fp32(actual, ref64, ref32, rtol=RTOL, atol_frac=5e-5, slack=2.0, what=""):
    """For sums that are ill-conditioned in float32 (the conv weight/bias gradients: 10^6..10^7 cancelling products per element
    behind max-pools): pass when ``actual`` meets the float64 oracle element-wise, OR when its worst error against float64 is
    no larger than ``slack`` x the worst error of the SAME arithmetic done by torch on the CPU in float32 (``ref32``: the
    reference's own path) -- i.e. the GPU is held to the accuracy the reference itself has, not to one float32 cannot give."""
    a = np.asarray(actual, dtype=np.float64); e = np.asarray(ref64, dtype=np.float64); r = np.asarray(ref32, dtype=np.float64)
    assert a.shape == e.shape == r.shape, f"{what}: shapes {a.shape} {e.shape} {r.shape}"
    scale = float(np.max(np.abs(e))) if e.size else 0.0
    err = np.abs(a - e)
    if (err <= rtol * np.abs(e) + atol_frac * scale + 1e-30).all():
        return
    ref_err = float(np.abs(r - e).max())
    assert err.max() <= slack * ref_err + atol_frac * scale, (
        f"{what}: max err {err.max():.3e} vs float64 at scale {scale:.3e}; torch-CPU float32 is off by {ref_err:.3e}")


def oracle_train(state, fp, img, y, orders, batch_size, faithful, test, lrs=None, dtype=torch.float32):
    """The reference's per-fold loop (Models/...20250113.py:165-241, incl. the train-once / eval-thereafter quirk when ``faithful``)
    run with the CPU oracle and the oracle's AdamW, in ``dtype`` (float32 = the reference's own precision; float64 = the yardstick
    both float32 implementations are measured against).  Returns (mean training loss per epoch, predictions on ``test``)."""
    from oracle import reference_cpu as oracle
    cast = (lambda t: t.detach().cpu().to(dtype)) if dtype != torch.float32 else (lambda t: t.detach().cpu())
    p = {k: (cast(v) if v.dtype.is_floating_point else v.detach().cpu()).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in state.items()}
    fp, img, y = cast(fp), cast(img), cast(y)
    test = (cast(test[0]), cast(test[1]))
    keys = [k for k, v in p.items() if v.requires_grad]
    m = {k: torch.zeros_like(p[k]) for k in keys}; v2 = {k: torch.zeros_like(p[k]) for k in keys}
    step, training_mode, losses = 0, True, []
    for ep, order in enumerate(orders):
        if not faithful:
            training_mode = True
        tot, nb = 0.0, 0
        for i in range(0, len(order), batch_size):
            idx = torch.as_tensor(order[i:i + batch_size])
            for k in keys:
                p[k].grad = None
            st = {}
            loss = oracle.mse_loss(oracle.mixed_input_forward(p, fp[idx], img[idx], training=training_mode, bn_state=st), y[idx])
            loss.backward()
            step += 1
            with torch.no_grad():
                for k in keys:
                    oracle.adamw_step(p[k], p[k].grad, m[k], v2[k], step, **({} if lrs is None else {"lr": lrs[ep]}))
                for k, val in st.items():
                    p[k] = val
            tot += float(loss.detach()); nb += 1
        losses.append(tot / nb)
        training_mode = False                        # the validation pass leaves the model in eval mode
    with torch.no_grad():
        preds = torch.cat([oracle.mixed_input_forward(p, test[0][i:i + batch_size], test[1][i:i + batch_size], training=False).reshape(-1)
                           for i in range(0, test[0].shape[0], batch_size)])
    return losses, preds


# ---- multi-process launcher for the distributed tests ---------------------------------------------------------------------------
_RENDEZVOUS_ERRORS = ("Address already in use", "EADDRINUSE", "Connection refused", "connection refused", "failed to connect",
                      "The server socket has failed to listen")


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_entry(worker, rank, world, port, q, done, args):
    """Runs in the child: call ``worker(rank, world, port, *args)`` and report ("ok", result) or ("error", traceback).  The
    child then stays until the parent has read every report: tensors in a result travel as shared-memory handles."""
    try:
        q.put((rank, "ok", worker(rank, world, port, *args)))
    except BaseException as e:                     # noqa: BLE001 -- report instead of leaving the peer hanging in a collective
        import traceback
        q.put((rank, "error", "".join(traceback.format_exception(type(e), e, e.__traceback__))[-6000:]))
    done.wait(timeout=120)


def run_ranks(worker, world, args=(), timeout=300, launches=3):
    """Spawn ``world`` processes running ``worker(rank, world, port, *args)`` and return their results by rank.
    The LAUNCH is repeated only when a rank reports a rendezvous error (the probed port was taken before the store bound it).
    A rank that times out, dies without reporting or raises anything else fails the test AT ONCE, with every traceback and
    exit code of that one launch -- a hang or crash is never re-run."""
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(launches):
        port = _free_port()
        q, done = ctx.Queue(), ctx.Event()
        procs = [ctx.Process(target=_rank_entry, args=(worker, r, world, port, q, done, tuple(args))) for r in range(world)]
        for p in procs:
            p.start()
        msgs, timed_out = [], False
        for _ in procs:
            try:
                msgs.append(q.get(timeout=timeout))
            except queue.Empty:
                timed_out = True
                break
        done.set()
        for p in procs:
            p.join(timeout=5 if timed_out else 60)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
        errors = [(r, m) for r, kind, m in msgs if kind == "error"]
        report = "\n".join([f"--- rank {r} ---\n{m}" for r, m in errors] + [f"exit codes: {[p.exitcode for p in procs]}"])
        if timed_out or len(msgs) < world:
            raise AssertionError(f"{world - len(msgs)} of {world} ranks did not report within {timeout} s (not retried)\n{report}")
        if not errors:
            return [m for _, _, m in sorted(msgs, key=lambda t: t[0])]
        if all(any(s in m for s in _RENDEZVOUS_ERRORS) for _, m in errors) and attempt + 1 < launches:
            continue                                  # rendezvous only: try another port
        raise AssertionError(report)
    raise AssertionError("unreachable")
