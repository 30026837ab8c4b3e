"""cogdl_amd.operators.ops on the CPU: the module mirrors cogdl/operators/ops.py name for name; CPU tensors run the
reference's own torch expressions, so every result equals the reference's golden output bit for bit
(tests/golden/message_ops.npz, written by the reference itself) -- and the checks of the reference's own
tests/test_ops.py:31-135 hold."""
import types

import numpy as np
import torch

from cogdl_amd.operators import ops


def _graph(z):
    g = types.SimpleNamespace()
    g.edge_index = (torch.from_numpy(z["row"]), torch.from_numpy(z["col"]))
    g.edge_weight = torch.from_numpy(z["w"])
    return g


def test_module_exports_every_reference_name():
    for name in ("scatter_add", "op_src_edge", "op_aggr", "src_op_e_aggr_coo", "s_add_e_sum", "s_mul_e_sum",
                 "s_sub_e_sum", "s_add_e_mean", "s_mul_e_mean", "s_sub_e_mean", "s_add_e", "s_sub_e", "s_mul_e",
                 "src_op_target_coo", "s_add_t", "s_mul_t", "s_sub_t", "s_dot_t", "s_div_t", "message_passing"):
        assert callable(getattr(ops, name)), name
        assert getattr(ops, name).__name__ == name


def test_cpu_results_equal_reference_goldens(golden):
    z = golden("message_ops")
    g, n = _graph(z), int(z["n"])
    x, ef, es = (torch.from_numpy(z[k]) for k in ("x", "ef", "es"))
    for op1 in ("add", "sub", "mul"):
        for op2 in ("sum", "mean"):
            fn = getattr(ops, "s_%s_e_%s" % (op1, op2))
            assert fn(g, x, ef).numpy().tobytes() == z["%s_%s" % (op1, op2)].tobytes()
            assert fn(g, x, ef, weight=True).numpy().tobytes() == z["%s_%s_w" % (op1, op2)].tobytes()
    assert ops.s_mul_e_sum(g, x, es).numpy().tobytes() == z["mul_sum_scalar"].tobytes()
    assert ops.scatter_add(ef, g.edge_index[0], n).numpy().tobytes() == z["scatter_add"].tobytes()
    G = torch.from_numpy(z["G"])
    xg, eg = x.clone().requires_grad_(), ef.clone().requires_grad_()
    (ops.s_mul_e_mean(g, xg, eg, weight=True) * G).sum().backward()
    np.testing.assert_allclose(xg.grad.numpy(), z["grad_x_mul_mean_w"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(eg.grad.numpy(), z["grad_e_mul_mean_w"], rtol=1e-6, atol=1e-6)


def test_elementwise_ops_like_reference_tests(golden):
    # the exact-equality checks of the reference's tests/test_ops.py:31-57,113-135
    z = golden("message_ops")
    g = _graph(z)
    x, ef = torch.from_numpy(z["x"]), torch.from_numpy(z["ef"])
    row, col = g.edge_index
    src, dst = x[col], x[row]
    assert (ops.s_add_t(g, x) == src + dst).all() and (ops.s_sub_t(g, x) == src - dst).all()
    assert (ops.s_mul_t(g, x) == src * dst).all() and (ops.s_div_t(g, x) == src / dst).all()
    assert (ops.s_dot_t(g, x) == (src * dst).sum(dim=-1, keepdim=True)).all()
    assert (ops.s_add_e(g, x, ef) == src + ef).all() and (ops.s_sub_e(g, x, ef) == src - ef).all()
    assert (ops.s_mul_e(g, x, ef) == src * ef).all()
    zero = torch.zeros(3, 2)
    assert (ops.src_op_target_coo("div", None, torch.ones(3, 2), zero) == 0).all()  # x / 0 -> 0 (ops.py:143-145)


def test_op_aggr_mean_of_an_empty_destination_is_zero():
    msg = torch.tensor([[2.0, 4.0], [6.0, 8.0]])
    out = ops.op_aggr("mean", msg, torch.tensor([1, 1]), 3)
    assert out.tolist() == [[0.0, 0.0], [4.0, 6.0], [0.0, 0.0]]
