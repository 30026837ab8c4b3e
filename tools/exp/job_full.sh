#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python tools/csc_probe.py > gpurun_out/csr2csc_ab.txt 2>&1; tail -8 gpurun_out/csr2csc_ab.txt
timeout 300 python tools/sage_bench.py --captured --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/sage_captured.txt
timeout 600 python tools/gat_bench.py > gpurun_out/gat_bench.txt 2>&1; tail -18 gpurun_out/gat_bench.txt | cut -c1-200
