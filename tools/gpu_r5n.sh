#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/papers_variants.py > gpurun_out/r5n_papers_variants.txt 2>&1; tail -18 gpurun_out/r5n_papers_variants.txt
