"""Stand-in rank program for tests/test_bench_launcher.py: what bench.py's launcher starts when BBBP_BENCH_WORKER names this
file.  Joins a gloo group from the torch.distributed.run environment, sums a value over the ranks and lets rank 0 print one JSON
line; `--steps 13` makes every rank exit with code 3 (return-code propagation)."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
args, _ = ap.parse_known_args()
if args.steps == 13:
    sys.exit(3)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == args.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t), "launched_by": os.environ.get("BBBP_BENCH_LAUNCHED_BY")}), flush=True)
dist.destroy_process_group()
