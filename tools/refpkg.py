"""Locate and drive the REAL reference package (CogDL) -- measurement/test infrastructure only, never imported by
cogdl_amd/.

Where the package comes from:
  * oracle/_ref/pkg/cogdl   staged by `make -C oracle ref` in the build container (git-ignored, travels to the GPU box
                            with the snapshot exactly like oracle/_ref/O3/spmm_cpu.so);
  * $COGDL_REFERENCE or /root/reference (build container only) as a fall-back, through a scratch copy: importing the
    reference writes into its own tree.
The four third-party modules the reference imports but this image lacks (optuna, numba, grave, turtle) are the no-op
stubs of tests/golden/_stubs.
"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "oracle", "_ref", "pkg")
STUBS = os.path.join(ROOT, "tests", "golden", "_stubs")


def available():
    ref = os.environ.get("COGDL_REFERENCE", "/root/reference")
    return os.path.isdir(os.path.join(STAGED, "cogdl")) or os.path.isdir(os.path.join(ref, "cogdl"))


def setup(install=True, linear=False):
    """Make `import cogdl` resolve to the unchanged reference (optionally on top of cogdl_amd.install()).
    Returns the directory that was put on sys.path."""
    sys.dont_write_bytecode = True
    if os.path.isdir(os.path.join(STAGED, "cogdl")):
        pkg = STAGED
    else:
        ref = os.environ.get("COGDL_REFERENCE", "/root/reference")
        pkg = tempfile.mkdtemp(prefix="cogdl_refcopy_")
        shutil.copytree(os.path.join(ref, "cogdl"), os.path.join(pkg, "cogdl"))
    for p in (pkg, STUBS, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    if install:
        import cogdl_amd

        cogdl_amd.install(linear=linear)
        import cogdl  # noqa: F401

        cogdl_amd.install(linear=linear)  # again: rebinds coo2csr_index in the modules that are imported by now
    return pkg


def node_dataset(num_nodes, num_pairs, num_features, num_classes, seed=0, sparse_features=False, device="cpu"):
    """A synthetic stand-in for a Planetoid / OGB node-classification dataset in the reference's own types
    (cogdl.data.Graph inside cogdl.datasets.NodeDataset): symmetrised random edges, random labels, 5 % / 20 % / 40 %
    train / val / test masks.  Self loops and normalisation are added by the reference's own pre_transform
    (cogdl/wrappers/data_wrapper/node_classification/node_classification_dw.py:19-20)."""
    import torch
    from cogdl.data import Graph
    from cogdl.datasets import NodeDataset

    gen = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, num_nodes, (2, num_pairs), generator=gen)
    ei = ei[:, ei[0] != ei[1]]
    ei = torch.unique(torch.cat([ei, ei.flip(0)], 1), dim=1)  # symmetrise + coalesce (cogdl/datasets/ogb.py:50-55)
    if sparse_features:
        x = (torch.rand(num_nodes, num_features, generator=gen) < 0.01).float()
    else:
        x = torch.randn(num_nodes, num_features, generator=gen)
    y = torch.randint(0, num_classes, (num_nodes,), generator=gen)
    g = Graph(x=x, edge_index=ei, y=y)
    perm = torch.randperm(num_nodes, generator=gen)
    n_tr, n_va, n_te = num_nodes // 20, num_nodes // 5, (num_nodes * 2) // 5
    for name, lo, hi in (("train_mask", 0, n_tr), ("val_mask", n_tr, n_tr + n_va),
                         ("test_mask", n_tr + n_va, n_tr + n_va + n_te)):
        mask = torch.zeros(num_nodes, dtype=torch.bool)
        mask[perm[lo:hi]] = True
        setattr(g, name, mask)
    # (the dataset class saves itself to `path` when it is built: a scratch file, not ./data.pt in the caller's directory)
    path = os.path.join(tempfile.mkdtemp(prefix="cogdl_ds_"), "data.pt")
    return NodeDataset(path=path, data=g, metric="accuracy")


def cora_like(seed=0):
    """BASELINE.json configs[0]: 2,708 nodes, 10,556 directed edges, 1,433 features, 7 classes (graph.rst:115-117)."""
    return node_dataset(2708, 5300, 1433, 7, seed=seed, sparse_features=True)


def arxiv_like(seed=0):
    """BASELINE.json configs[1]: 169,343 nodes, ~2.33 M directed edges (+ self loops), 128 features, 40 classes."""
    return node_dataset(169_343, 1_166_243, 128, 40, seed=seed)


class StepTimer:
    """Wall time of every cogdl.trainer.Trainer.train_step call (trainer.py:500-540), fenced with
    torch.cuda.synchronize() on both sides when a GPU is in use.  Instruments the unchanged class from outside."""

    def __init__(self):
        self.ms = []
        self.losses = []
        self._orig = None

    def __enter__(self):
        import torch
        from cogdl.trainer import trainer as T

        orig = T.Trainer.train_step
        times, losses = self.ms, self.losses

        def timed(this, model_w, train_loader, optimizers, lr_schedulers, device, scaler):
            cuda = torch.cuda.is_available() and str(device) != "cpu"
            if cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = orig(this, model_w, train_loader, optimizers, lr_schedulers, device, scaler)
            if cuda:
                torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            losses.append(float(out))
            return out

        self._orig = (T.Trainer, orig)
        T.Trainer.train_step = timed
        return self

    def __exit__(self, *exc):
        cls, orig = self._orig
        cls.train_step = orig
        return False


def run_experiment(dataset, model="gcn", epochs=5, cpu=False, seed=0, **kw):
    """cogdl.experiment(...) in a scratch working directory (the Trainer writes ./checkpoints/model.pt).  Returns
    (result dict of the single variant + "train_losses" per epoch, [train_step ms])."""
    from cogdl import experiment

    cwd = os.getcwd()
    scratch = tempfile.mkdtemp(prefix="cogdl_exp_")
    os.chdir(scratch)
    try:
        with StepTimer() as timer:
            res = experiment(dataset=dataset, model=model, epochs=epochs, cpu=cpu, seed=[seed], **kw)
    finally:
        os.chdir(cwd)
        shutil.rmtree(scratch, ignore_errors=True)
    (variant,) = list(res.values())
    out = dict(variant[0])
    out["train_losses"] = timer.losses
    return out, timer.ms
