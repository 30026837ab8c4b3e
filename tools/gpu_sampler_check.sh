cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sample_gpu.py tests/test_captured_step_gpu.py tests/test_pipeline_gpu.py tests/test_config4_gpu.py tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "not edge_softmax" > gpurun_out/r5q_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5q_tests.log
COGDL_AMD_TUNING=11=1 timeout 600 python -m pytest tests/test_sample_gpu.py tests/test_captured_step_gpu.py tests/test_pipeline_gpu.py tests/test_config4_gpu.py -q -x -p no:cacheprovider > gpurun_out/r5q_tests_sortform.log 2>&1; echo "sort-form tests rc=$?"; tail -4 gpurun_out/r5q_tests_sortform.log
timeout 300 python tools/sampler_bench.py 2>&1 | tail -8
