#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err ) 2>&1 | tail -3
python -c "
import json; r=json.load(open('gpurun_out/r5g_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'])
ro=r['roofline']; print({k:ro[k] for k in ro if k.startswith('measured') or k.startswith('frac_of') or k in ('torch_copy_GBs','kernel_copy_GBs')})
print('hbm_resident', json.dumps(ro.get('hbm_resident'))[:600])
p=r.get('configs4_papers_1gpu',{}); print('papers', json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('forward','backward_alone','forward_backward','peak_allocated_GB','segments','nnz','error')}) for k,v in p.items()})[:1800])
print('gnn_epoch', json.dumps({k:v for k,v in r['gnn_epoch'].items() if k!='trainer'})[:600])
print('trainer', {k:(v.get('train_step_ms_median') if isinstance(v,dict) else v) for k,v in r['gnn_epoch'].get('trainer',{}).items()})
"
tail -3 gpurun_out/r5g_bench.err
