#!/bin/bash
# One gpurun call.  Usage: bash tools/gpu_round.sh [stage ...]   stages: pmcgat smoke tests variants bench prof pmc pmc2 pmcops ops epoch e2e sampler hunt gat pmccombined nocache_tests timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
STAGES="${@:-smoke tests variants bench prof}"
has() { [[ " $STAGES " == *" $1 "* ]]; }
(lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; free -g | head -2; rocminfo | grep -E "gfx|Compute Unit" | head -6) > gpurun_out/host.txt 2>&1
if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -2 gpurun_out/smoke.log
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
if has variants; then
  timeout 900 python tools/spmm_variants.py > gpurun_out/variants.txt 2>&1; timeout 600 python tools/spmm_variants.py big >> gpurun_out/variants.txt 2>&1
  tail -5 gpurun_out/variants.txt
fi
if has bench; then
  timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
  cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
fi
if has prof; then
  rm -rf gpurun_out/prof_kt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_kt" -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu --no-shard-base) > gpurun_out/bench_prof.log 2>&1
  find gpurun_out/prof_kt -name "*stats*" | head; 
fi
if has ops; then
  timeout 1500 python tools/ops_bench.py --json gpurun_out/ops_bench.json > gpurun_out/ops_bench.txt 2>&1; echo "ops rc=$?" >> gpurun_out/ops_bench.txt
  tail -60 gpurun_out/ops_bench.txt
fi
if has epoch; then
  rm -rf gpurun_out/prof_epoch
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_epoch" -o ep -- python "$GRAFT_REPO_ROOT/tools/epoch_probe.py") > gpurun_out/epoch_probe.log 2>&1
  python tools/epoch_breakdown.py "$(find gpurun_out/prof_epoch -name '*kernel_stats.csv' | head -1)" > gpurun_out/epoch_breakdown.txt 2>&1
  tail -3 gpurun_out/epoch_probe.log; cat gpurun_out/epoch_breakdown.txt
fi
if has e2e; then
  (echo "# tools/sage_bench.py (configs[3]) and tools/gat_bench.py (configs[2]) on one MI355X"; timeout 400 python tools/sage_bench.py --inference 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --batch 8192 --steps 30 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --features host 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --features host --pipeline 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --pipeline 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --features host --pipeline --batch 8192 --steps 30 2>&1 | tail -1; timeout 600 python tools/gat_bench.py 2>&1 | tail -5) > gpurun_out/e2e.txt 2>&1
  cat gpurun_out/e2e.txt | cut -c1-300
fi
if has sampler; then
  timeout 600 python tools/sampler_bench.py > gpurun_out/sampler_bench.txt 2>&1; tail -10 gpurun_out/sampler_bench.txt
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    tag=$(echo $c | tr ' ' '_')
    rm -rf gpurun_out/pmc_$tag
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_probe.py") > gpurun_out/pmc_$tag.log 2>&1
  done
  python tools/pmc_summarize.py gpurun_out > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.err; cat gpurun_out/pmc_summary.json | head -50
fi
if has pmc2; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc2_$c
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc2_$c" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_probe_es.py") > gpurun_out/pmc2_$c.log 2>&1
  done
  python tools/pmc_by_kernel.py gpurun_out pmc2_ > gpurun_out/pmc2_summary.json 2> gpurun_out/pmc2_summary.err; head -c 3000 gpurun_out/pmc2_summary.json
fi
if has pmcops; then
  # ONE operator per process, each under its own short timeout (a combined run faulted and hung under the profiler in
  # round 3: tools/pmc_probe_ops.py's header); a pass that times out loses 75 s, not the call
  for op in csr_spmm mhspmm gat_fwd mhsddmm csr_sddmm gat_bwd; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/pmcops_${op}_$c
      (cd /tmp && timeout 75 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcops_${op}_$c" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_probe_ops.py" $op) > gpurun_out/pmcops_${op}_$c.log 2>&1
      echo "pmcops $op $c rc=$?"
    done
  done
fi
if has pmcgat; then
  # round 6: the fused GAT operator of configs[2] (bf16, H = 8 x F = 8, forward + backward, without / with dropout) under the
  # counters, XCD-partitioned plan off and on; one operator per process, one counter set per pass
  for mode in off auto; do
    for op in gat_bf16 gat_drop_bf16; do
      for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
        tag=$(echo $c | tr ' ' '_')
        rm -rf gpurun_out/pmcgat_${mode}_${op}_$tag
        (cd /tmp && COGDL_AMD_XCD=$mode timeout 120 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcgat_${mode}_${op}_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_probe_ops.py" $op) > gpurun_out/pmcgat_${mode}_${op}_$tag.log 2>&1
        echo "pmcgat $mode $op $tag rc=$?"
      done
    done
    python tools/pmc_by_kernel.py gpurun_out pmcgat_${mode}_ GatFwd GatBwd > gpurun_out/pmcgat_${mode}.json 2> gpurun_out/pmcgat_${mode}.err
  done
  python - <<'PYEOF'
import json
for mode in ("off", "auto"):
    r = json.load(open("gpurun_out/pmcgat_%s.json" % mode))
    print(mode, r.get("calibration"))
    for k, v in r["kernels"].items():
        hit, miss = v.get("TCC_HIT_sum"), v.get("TCC_MISS_sum")
        print("  %-100s %8.1f us  read %6.2f GB  write %6.2f GB  L2 hit %s" % (k[:100], v["duration_us_profiled"], (v.get("hbm_read_bytes") or 0) / 1e9, (v.get("hbm_write_bytes") or 0) / 1e9, "%.3f" % (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss else "-"))
PYEOF
fi
if has gat; then
  timeout 600 python tools/gat_bench.py > gpurun_out/gat_bench.txt 2>&1; tail -25 gpurun_out/gat_bench.txt | cut -c1-400
fi
if has hunt; then
  # The memory fault of round 3's combined PMC probe: the SAME operator sequence in one process, without
  # the profiler, under allocation layouts / dispatch modes that expose out-of-bounds accesses the caching allocator hides:
  #   plain      as is
  #   nocache    every tensor its own hipMalloc (no neighbours from the caching allocator's big blocks)
  #   serialize  AMD_SERIALIZE_KERNEL=3: one kernel at a time, as the counter collection runs them
  # each under a hard 200 s limit (-s KILL); the logs say which operator was running when a fault hit.
  for mode in plain nocache serialize; do
    case $mode in
      plain) envs="" ;;
      nocache) envs="PYTORCH_NO_CUDA_MEMORY_CACHING=1" ;;
      serialize) envs="AMD_SERIALIZE_KERNEL=3" ;;
    esac
    env $envs timeout -s KILL 200 python tools/pmc_probe_ops.py > gpurun_out/hunt_$mode.log 2>&1
    echo "hunt $mode rc=$? $(grep -c -i 'memory access fault' gpurun_out/hunt_$mode.log) fault line(s); last: $(grep done gpurun_out/hunt_$mode.log | tail -1)"
  done
fi
if has pmccombined; then
  # Round 3's fault, re-run: ALL operators of tools/pmc_probe_ops.py in ONE process under the counters (one pass per
  # counter), hard 400 s limit each.  The log's first lines carry the profiler's own timestamps: a fault within a second
  # of "HSA version ... initialized" cannot come from operators that only start after seconds of graph generation.
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcall_$c
    (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcall_$c" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_probe_ops.py") > gpurun_out/pmcall_$c.log 2>&1
    echo "pmccombined $c rc=$? faults=$(grep -c -i 'memory access fault' gpurun_out/pmcall_$c.log) $(grep done gpurun_out/pmcall_$c.log | tr '\n' ' ')"
  done
  python tools/pmc_by_kernel.py gpurun_out pmcall_ SpmmOp GatFwd GatBwd SddmmOp MhsddmmOp > gpurun_out/pmcall_summary.json 2> gpurun_out/pmcall_summary.err; python -c "import json; r=json.load(open('gpurun_out/pmcall_summary.json')); [print('%-110s %8.1f us  %7.2f GB  %5.2f TB/s' % (k[:110], v['duration_us_profiled'], v.get('hbm_bytes_per_launch',0)/1e9, v.get('hbm_GBs',0)/1e3)) for k,v in r['kernels'].items() if 'main' in k]"
fi
if has nocache_tests; then
  # (stream capture needs the caching allocator's private pools: the hipGraph tests are deselected; failures listed, not -x)
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -s KILL 1800 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider -k "not captur and not replay and not hipgraph and not graph_launch" > gpurun_out/pytest_nocache.log 2>&1; echo "nocache pytest rc=$?" >> gpurun_out/pytest_nocache.log
  grep -E "^FAILED|^ERROR|passed|failed|rc=" gpurun_out/pytest_nocache.log | tail -25
fi
if has timeline; then
  rm -rf gpurun_out/prof_tl
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_tl" -o tl -- python "$GRAFT_REPO_ROOT/tools/sage_bench.py" --captured --steps 60 --warmup 10) > gpurun_out/timeline.log 2>&1
  python tools/step_timeline.py "$(find gpurun_out/prof_tl -name '*kernel_trace.csv' | head -1)" > gpurun_out/captured_step_timeline.txt 2>&1
  (timeout 300 python tools/sage_bench.py --captured --steps 200 --warmup 20 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --captured --batch 128 --steps 200 --warmup 20 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --captured --steps 200 --warmup 20 --side-stream 2>&1 | tail -1; timeout 300 python tools/sage_bench.py --captured --batch 128 --steps 200 --warmup 20 --side-stream 2>&1 | tail -1) > gpurun_out/captured_ms.txt 2>&1
  cut -c1-140 gpurun_out/captured_step_timeline.txt | tail -70; grep -o '"ms_per_step": [0-9.]*' gpurun_out/captured_ms.txt
fi
