"""HIP-graph capture of a training step (SURVEY.md section 8f rank 3 / round-1 verdict item 6): a full-graph GNN
epoch is ~100 short kernels, i.e. bound by the host's launch rate, not by the GPU.  The C ABI neither allocates nor
synchronises, so a whole step -- forward, backward, optimizer -- can be captured once and replayed as ONE graph launch.

    step = cogdl_amd.graphs.capture(train_step)      # train_step(): no arguments, reads/writes persistent tensors
    for epoch in range(n): step()

What the helper takes care of (torch.cuda.CUDAGraph does the capture itself):
  * the graph-plan lookups of the backward passes (cached transposes, keyed on a structure hash that has to be read on
    the host) are recorded during an eager run and replayed from that record while capturing (cogdl_amd.plan.PlanTape);
    so are the launches' own decisions (plan.taped_choice): the recorded run waits for every structure's key, so skewed
    structures run over their XCD-partitioned, length-ordered plans (cogdl_amd/xcdplan.py) in the captured step -- forward
    calls too, which an eager call with its hash still in flight cannot;
  * warm-up runs on a side stream, as torch's capture rules ask, so that every lazily created piece of state
    (plans, memoised transposed weights, cuBLAS/hipBLASLt workspaces, optimizer state) exists before the capture.
What the step must respect: static shapes and addresses (index with precomputed index tensors, not boolean masks), no
host synchronisation (`.item()`, `.cpu()`, printing a loss), an optimizer created with `capturable=True`.
"""
import torch

from . import plan as _plan


class CapturedStep:
    def __init__(self, graph, outputs, keepalive=()):
        self.graph, self.outputs = graph, outputs
        # The captured kernels read the step's graph plans (colptr / rowind / perm) and their memoised transposed
        # weights through RAW pointers.  Those tensors were allocated eagerly during the warm-up runs, i.e. outside
        # the graph's private pool, and the plan cache (an LRU with a byte budget, PLANS.clear(), or a later
        # transposed_values() call with another weight tensor, which replaces plan._val_t) would otherwise be their
        # only owner: this tuple keeps every one of them alive for as long as the graph can be replayed.
        self._keepalive = tuple(keepalive)

    def __call__(self):
        self.graph.replay()
        return self.outputs


def capture(step, warmup=3, pool=None):
    """Run `step()` `warmup` times eagerly (side stream), record its plan lookups, capture it, return a callable that
    replays the captured graph and returns the (static) outputs of the captured call."""
    if warmup < 1:
        raise ValueError("at least one eager run is needed to record the plan lookups")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    tape = _plan.PlanTape()
    with torch.cuda.stream(side):
        for i in range(warmup):
            if i == warmup - 1:
                _plan.set_tape(tape)
            try:
                step()
            finally:
                _plan.set_tape(None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    tape.mode, tape.pos, tape.cpos = "replay", 0, 0
    graph = torch.cuda.CUDAGraph()
    _plan.set_tape(tape)
    try:
        with torch.cuda.graph(graph, pool=pool):
            outputs = step()
    finally:
        _plan.set_tape(None)
    if tape.pos != len(tape.plans):
        raise RuntimeError("hipGraph capture: the captured step used %d of the %d recorded plan lookups"
                           % (tape.pos, len(tape.plans)))
    if tape.cpos != len(tape.choices):
        raise RuntimeError("hipGraph capture: the captured step took %d of the %d recorded launch decisions"
                           % (tape.cpos, len(tape.choices)))
    keep = []
    for plan in tape.plans:
        keep.append((plan, plan.colptr, plan.rowind, plan.perm, plan._val_t, plan._val_src))

    def xplans_of(value):  # the XCD-partitioned plans inside a recorded decision (None | plan | tuples of them)
        if isinstance(value, (tuple, list)):
            for v in value:
                yield from xplans_of(v)
        elif hasattr(value, "vrowptr"):
            yield value

    for _, value in tape.choices:
        for xp in xplans_of(value):
            # (the plan object owns its index tensors and the struct the launches pass by pointer; its memoised permuted weights
            #  are replaced when another weight tensor comes by: the captured kernels read the ones of the capture)
            keep.append((xp, xp._val_p, xp._val_src))
    return CapturedStep(graph, outputs, keep)
