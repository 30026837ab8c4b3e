#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/rowqueue_ab.py > gpurun_out/r5m_rowqueue_ab.txt 2>&1; tail -16 gpurun_out/r5m_rowqueue_ab.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
