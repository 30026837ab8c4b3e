#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
summ() { python - "$1" "$2" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d=r["symmetrised"]; print(sys.argv[2], "symmetrised fwd %.1f ms (%.3f)  bwd alone %.1f ms" % (d["forward"]["ms"], d["forward"]["frac"], d["backward_alone"]["ms"]), flush=True)
PY
}
for w in 4096 16384 262144 1048576 1073741824; do
  COGDL_AMD_ROW_WINDOW=$w python tools/papers_bench.py --only symmetrised --steps 3 > gpurun_out/ps_w$w.json 2>/dev/null; summ gpurun_out/ps_w$w.json "window=$w"
done
for t in 256 1024 2048; do
  COGDL_AMD_TUNING="1=$t" python tools/papers_bench.py --only symmetrised --steps 3 > gpurun_out/ps_t$t.json 2>/dev/null; summ gpurun_out/ps_t$t.json "window=65536 long-row threshold=$t"
done
