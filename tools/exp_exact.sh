#!/bin/bash
# one rank, B = 512: the exact-global-batch mode on the fused engine against the per-op composition and the replica engine
timeout -k 10 400 python tools/time_configs.py > gpurun_out/r03_time_configs.txt 2>&1; tail -12 gpurun_out/r03_time_configs.txt
