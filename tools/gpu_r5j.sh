#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/longgrid_ab.py > gpurun_out/r5j_longgrid_ab.txt 2>&1; tail -9 gpurun_out/r5j_longgrid_ab.txt
timeout 300 python tools/epoch_host_profile.py > gpurun_out/r5j_host_profile.txt 2>&1; head -60 gpurun_out/r5j_host_profile.txt | cut -c1-200
timeout 300 python -m pytest tests/test_captured_step_gpu.py -q -x -p no:cacheprovider -k "without_nodes" 2>&1 | tail -2
