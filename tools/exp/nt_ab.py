"""A/B: csr_spmm with non-temporal loads of colind / val and non-temporal stores of the output rows (build flag
COGDL_NT_STREAM, variant library tools/exp/lib_nt/libcogdl_hip.so) against the default library, one process per library:
papers100M-shaped graph forward (both graphs) and the arxiv / Reddit-shaped graphs."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
sys.argv = ["papers_bench", "--steps", "3"]
from cogdl_amd import _lib
if os.environ.get("EXP_LIB"):
    _lib.HIP_LIB_PATH = os.environ["EXP_LIB"]
import torch
from cogdl_amd import synth
from cogdl_amd.operators.spmm import csr_spmm_raw
from tools.ops_bench import timeit
dev = "cuda:0"
for name, g in (("arxiv-uniform", synth.arxiv_like(0)), ("arxiv-rmat", synth.arxiv_like(0, "rmat")), ("reddit", synth.reddit_like(0, norm="sym"))):
    rp, ci, w = g.rowptr.to(dev), g.colind.to(dev), g.weight.to(dev)
    for f in (128, 64, 40):
        x = torch.randn(g.num_nodes, f, device=dev)
        t = timeit(lambda: csr_spmm_raw(rp, ci, w, x), 20) * 1e3
        print("%%-14s F=%%3d fp32  %%8.1f us" %% (name, f, t), flush=True)
    del rp, ci, w, x
import tools.papers_bench as pb
pb.main()
''' % ROOT
for tag, lib in (("default", ""), ("nt", os.path.join(ROOT, "tools/exp/lib_nt/libcogdl_hip.so")), ("default", ""), ("nt", os.path.join(ROOT, "tools/exp/lib_nt/libcogdl_hip.so"))):
    env = dict(os.environ, EXP_LIB=lib)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print("=====", tag, "rc", p.returncode, flush=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            try:
                r = json.loads(ln)
                for k in ("directed", "symmetrised"):
                    if k in r and "forward" in r[k]:
                        print("papers %-12s forward %8.1f ms (%.3f)  fwd+bwd %8.1f ms" % (k, r[k]["forward"]["ms"], r[k]["forward"]["frac"], r[k]["forward_backward"]["ms"]), flush=True)
                    elif k in r:
                        print("papers", k, r[k])
            except Exception as e:
                print("parse", e)
        else:
            print(ln)
    if p.returncode:
        print(p.stderr[-2000:])
