cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_big_dispatch_gpu.py tests/test_bigcsr_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -30 | cut -c1-300
