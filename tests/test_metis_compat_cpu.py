"""cogdl_amd.metis_compat without a GPU: the input forms of metis.part_graph, the loud failure where the partitioner
cannot run, and install(metis=True) serving `import metis` only when the real package is absent."""
import subprocess
import sys

import numpy as np
import pytest
import torch

from cogdl_amd import _lib, metis_compat


def test_adjacency_forms_give_the_same_csr():
    adj = [np.array([1, 2]), np.array([0]), np.array([0, 3]), np.array([2])]
    xadj, adjncy = metis_compat._csr_of(adj)
    assert xadj.tolist() == [0, 2, 3, 5, 6] and adjncy.tolist() == [1, 2, 0, 0, 3, 2]
    x2, a2 = metis_compat._csr_of((xadj, adjncy))
    assert x2.tolist() == xadj.tolist() and a2.tolist() == adjncy.tolist()
    x3, a3 = metis_compat._csr_of([[1, 2], [0], [0, 3], [2]])  # plain lists, as metis accepts them
    assert x3.tolist() == xadj.tolist() and a3.tolist() == adjncy.tolist()
    xe, ae = metis_compat._csr_of([[], []])
    assert xe.tolist() == [0, 0, 0] and ae.size == 0


def test_trivial_calls_need_no_device_and_everything_else_fails_loudly_without_one():
    adj = [np.array([1]), np.array([0])]
    assert metis_compat.part_graph(adj, 1) == (0, [0, 0])
    assert metis_compat.part_graph([], 3) == (0, [])
    with pytest.raises(_lib.BackendError):
        metis_compat.part_graph(adj, 0)
    with pytest.raises(_lib.BackendError):
        metis_compat.part_graph([np.array([7]), np.array([0])], 2)  # neighbour id out of range
    if not torch.cuda.is_available():
        with pytest.raises(_lib.BackendError, match="GPU"):
            metis_compat.part_graph(adj, 2)


def test_install_serves_import_metis_only_when_asked_and_absent():
    code = ("import sys; import cogdl_amd\n"
            "cogdl_amd.install()\n"
            "assert 'metis' not in sys.modules\n"
            "cogdl_amd.install(metis=True)\n"
            "import metis\n"
            "print(metis.__name__, callable(metis.part_graph))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["cogdl_amd.metis_compat", "True"]
