"""`python bench.py --gpus N` typed as is (N > 1, no WORLD_SIZE): the process must become a launcher that starts N ranks as child
processes through torch.distributed.run, never import torch or touch the GPU itself, pass rank 0's JSON line through and return the
children's exit code.  A stub rank program (gloo, CPU) stands in for the GPU worker."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_bench_rank.py")


def run(*argv, env=None):
    e = dict(os.environ, BBBP_BENCH_WORKER=STUB)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_typed_as_is_launches_two_ranks_and_forwards_one_json_line():
    r = run("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "sum": 3.0, "launched_by": "bench.py"}
    # the parent never loaded torch (hence no torch.cuda, no HIP) and used the contract's launcher form on 127.0.0.1
    assert "launcher imported torch: False" in r.stderr
    assert "-m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1" in r.stderr


def test_launcher_returns_the_ranks_exit_code():
    r = run("--gpus", "2", "--steps", "13")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_command_is_the_contract_form():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.torch is None                    # importing bench.py alone does not import torch either
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "5"], port=29511)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[9].endswith("bench.py") and cmd[10:] == ["--gpus", "4", "--steps", "5"]
    # every other option reaches the ranks untouched (the exact-global-batch mode shards the global batch over them)
    assert bench.launcher_command(2, ["--gpus", "2", "--exact-batch", "--config", "4"], port=1)[10:] == ["--gpus", "2", "--exact-batch", "--config", "4"]
