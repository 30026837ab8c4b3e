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
for m in desc asc mix; do
  COGDL_AMD_ROW_SCHED=$m python tools/papers_bench.py --steps 3 > gpurun_out/ps_m$m.json 2>/dev/null; summ gpurun_out/ps_m$m.json "schedule=$m"
done
