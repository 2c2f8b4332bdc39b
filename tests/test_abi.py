"""CPU: the C-ABI library loads, exports every symbol include/bbbp_hip.h declares, and the product path refuses to
run without a GPU instead of falling back to anything."""
import ctypes

import pytest
import torch

import bbbp_amd
from bbbp_amd import _lib, ops


def test_library_exports_every_declared_symbol():
    declared = _lib.declared_symbols()
    assert len(declared) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/bbbp_hip.h but not exported by libbbbp_hip.so"
    # every declared symbol has a ctypes signature and vice versa
    assert sorted(_lib._SIGNATURES) == declared
    assert _lib.lib().bbbp_abi_version() == 1


def test_descriptor_and_plan_functions_without_gpu():
    L = _lib.lib()
    d = _lib.MixedDesc(batch=512, fingerprint_size=167, nhead=1, num_layers=6, dim_feedforward=2048, training=1,
                       dropout_p=0.1, seed=1, need_input_grad=0)
    assert L.bbbp_mixed_num_params(ctypes.byref(d)) == 106 == len(list(bbbp_amd.MixedInputModel(167, 128).parameters()))
    ws = L.bbbp_mixed_workspace_bytes(ctypes.byref(d))
    assert 0.9e9 < ws < 4e9                 # ~1.0 GB of saved activations + scratch at B = 512
    bad = _lib.MixedDesc(batch=4, fingerprint_size=167, nhead=8, num_layers=6, dim_feedforward=2048)
    assert L.bbbp_mixed_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"divisible" in L.bbbp_last_error()
    assert L.bbbp_gemm_workspace_bytes(512, 128, 65536, 1) > 0          # split-K slabs for the image FC


def test_no_cpu_fallback():
    m = bbbp_amd.MixedInputModel(64, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 64), torch.zeros(2, 49152))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(ValueError):
        bbbp_amd.MixedInputModel(64, 64)        # the reference's forward hard-codes 3x128x128


def test_module_surface_matches_reference_contract():
    m = bbbp_amd.MixedInputModel(167, 128)
    keys = list(m.state_dict().keys())
    assert keys[0] == "fingerprint_transformer.layers.0.self_attn.in_proj_weight"
    assert "image_cnn.7.weight" in keys and "attention_fusion.attention_heads.3.2.bias" in keys
    assert "fc.2.running_mean" in keys and "fc.2.num_batches_tracked" in keys and "fc.7.bias" in keys
    assert sum(p.numel() for p in m.parameters()) == 13_464_087
    assert m.nhead == 1 and bbbp_amd.MixedInputModel(64, 128).nhead == 8
    # parameters live in one flat buffer, in named_parameters order, and stay so after dtype/device moves
    from bbbp_amd.models import flat_view_of
    assert flat_view_of(list(m.parameters())) is not None
    m.float()
    assert flat_view_of(list(m.parameters())) is not None
    ds = bbbp_amd.MixedDataset([[1.0, 2.0]], [[3.0]], [0.5])
    fp, img, y = ds[0]
    assert len(ds) == 1 and fp.dtype == img.dtype == y.dtype == torch.float32 and y.dim() == 0


def test_overlapped_reducer_schedule_tiles_the_flat_gradient_buffer():
    """distributed.OverlappedGradAllReduce fixes its collective schedule at construction from the parameter layout alone (host
    side, no GPU): image-FC weight, encoder layers from the last to the first, the two remaining slices, the conv tensors --
    together exactly the flat buffer, once."""
    import bbbp_amd
    from bbbp_amd import distributed as D
    for cls, F, layers in ((bbbp_amd.MixedInputModel, 64, 6), (bbbp_amd.ConcatMixedInputModel, 167, 6), (bbbp_amd.TwoBranchConcatModel, 167, 0)):
        m = cls(F, 128)
        r = D.OverlappedGradAllReduce(m)
        sched = r.schedule()
        assert [b for _, b, _, _ in sched if b is not None and b != 1] == [0] + list(range(1 + layers, 1, -1))
        spans = sorted((lo, hi) for _, _, lo, hi in sched if hi > lo)
        total = sum(p.numel() for p in m.parameters())
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        w = m.image_cnn[7].weight
        assert (sched[0][3] - sched[0][2]) == w.numel() == 128 * 65536
        if layers:
            per_layer = sum(p.numel() for p in m.fingerprint_transformer.layers[0].parameters())
            assert all(hi - lo == per_layer for _, b, lo, hi in sched if b is not None and b >= 2)


def test_graph_replay_stays_off_while_any_reducer_is_alive():
    """closing one reducer must not switch HIP-graph replay back on under another live reducer -- the first reducer
    remembers the process's mode, the LAST one to close restores it (host side only: bbbp_set_graphs is a flag)."""
    import bbbp_amd
    from bbbp_amd import _lib, distributed as D
    L = _lib.lib()
    before = L.bbbp_set_graphs(1)                     # pretend the process runs with replay on
    try:
        m = bbbp_amd.TwoBranchConcatModel(167, 128)
        a, b = D.OverlappedGradAllReduce(m), D.OverlappedGradAllReduce(m)
        assert L.bbbp_set_graphs(0) == 0              # off while both live (the query sets 0, which is what it already is)
        a.close()
        assert L.bbbp_set_graphs(0) == 0              # still off: b is alive
        b.close()
        assert L.bbbp_set_graphs(1) == 1              # restored by the last close
        a.close(); b.close()                          # idempotent
        assert D._LIVE["count"] == 0
    finally:
        L.bbbp_set_graphs(before)


def test_round4_entry_points_validate_their_arguments_before_touching_the_gpu():
    """Argument errors are reported as BBBP_ERR_ARG with a message, before any HIP call (so: testable without a GPU)."""
    import ctypes
    from bbbp_amd import _lib
    L = _lib.lib()
    ERR_ARG = 1
    fake = ctypes.c_void_p(4096)                         # never dereferenced: validation comes first
    # multi-tensor AdamW: tensor count, null pointers, alignment of the flat buffers, 1-based step
    assert L.bbbp_adamw_step_multi(None, fake, fake, fake, 16, fake, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == ERR_ARG
    assert b"n_tensors" in L.bbbp_last_error()
    assert L.bbbp_adamw_step_multi(None, fake, fake, fake, 16, None, 2, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == ERR_ARG
    assert L.bbbp_adamw_step_multi(None, ctypes.c_void_p(4100), fake, fake, 16, fake, 2, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == ERR_ARG
    assert b"aligned" in L.bbbp_last_error()
    assert L.bbbp_adamw_step_multi(None, fake, fake, fake, 16, fake, 2, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, None) == ERR_ARG
    assert L.bbbp_adamw_step_multi(None, fake, fake, fake, 0, None, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == 0          # empty: nothing to do
    assert L.bbbp_adamw_hyper_store(None, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0) == ERR_ARG
    assert L.bbbp_adamw_step(None, fake, fake, fake, fake, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0) == ERR_ARG
    assert L.bbbp_adamw_step_deferred(None, fake, fake, fake, fake, 16, 8, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0) == ERR_ARG   # lo > hi
    # per-thread setters are plain flags
    assert L.bbbp_set_seed_base(None) == 0
    prev = L.bbbp_set_conv_wgrad_beside_encoder(1)
    assert L.bbbp_set_conv_wgrad_beside_encoder(prev) == 1
    # MLP profile: group count bound
    buf = (ctypes.c_ulonglong * 3)()
    assert L.bbbp_mlp_profile_groups(ctypes.cast(buf, ctypes.c_void_p), 5000) == ERR_ARG
    assert L.bbbp_mlp_profile_groups(None, 1) == ERR_ARG
