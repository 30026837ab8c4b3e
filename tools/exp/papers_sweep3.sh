#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
summ() { python - "$1" "$2" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("directed","symmetrised"):
    if k in r and "forward" in r[k]:
        d=r[k]; print(sys.argv[2], k, "fwd %.1f ms (%.3f)  bwd alone %.1f ms" % (d["forward"]["ms"], d["forward"]["frac"], d["backward_alone"]["ms"]), flush=True)
PY
}
for cfg in "desc 65536" "asc 65536" "asc 16384" "asc 262144" "asc 1048576" "asc 1073741824" "desc 65536"; do
  set -- $cfg
  COGDL_AMD_ROW_SCHED=$1 COGDL_AMD_ROW_WINDOW=$2 python tools/papers_bench.py --steps 3 > gpurun_out/ps3.json 2>/dev/null; summ gpurun_out/ps3.json "schedule=$1 window=$2"
done
COGDL_AMD_ROW_ORDER=0 python tools/papers_bench.py --steps 3 > gpurun_out/ps3.json 2>/dev/null; summ gpurun_out/ps3.json "no schedule"
