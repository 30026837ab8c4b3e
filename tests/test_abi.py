"""The C-ABI boundary: both shared libraries load and export every symbol their headers declare,
and the ctypes signature tables cover exactly those symbols.  No compute, no GPU."""
import ctypes
import os
import re

from cogdl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("cogdl_hip.h", "cogdl_hip_")
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.HIP_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.HIP_SIGNATURES) == names
    assert _lib.hip().cogdl_hip_abi_version() == 9
    assert _lib.hip().cogdl_hip_strerror(3) == b"misaligned pointer"


def test_host_library_exports_every_declared_symbol():
    names = declared("cogdl_host.h", "cogdl_host_")
    lib = ctypes.CDLL(_lib.HOST_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.HOST_SIGNATURES) == names


def test_host_library_is_hip_free():
    """The sampler runs in forked DataLoader workers: libcogdl_host must not link the HIP runtime."""
    import subprocess

    out = subprocess.run(["ldd", _lib.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "hsa" not in out, out


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest

    monkeypatch.setattr(_lib, "_hip", None)
    monkeypatch.setattr(_lib, "HIP_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.BackendError):
        _lib.hip()


def test_gpu_ops_refuse_cpu_tensors():
    import pytest
    import torch

    from cogdl_amd.operators.spmm import csrspmm

    rowptr = torch.tensor([0, 1], dtype=torch.int32)
    colind = torch.tensor([0], dtype=torch.int32)
    with pytest.raises(_lib.BackendError):
        csrspmm(rowptr, colind, torch.ones(1, 4), torch.ones(1))


def test_segments_struct_layout_matches_the_c_header(tmp_path):
    """cogdl_hip_segments crosses the boundary by address: the ctypes mirror (cogdl_amd/_lib.py: Segments) must have the
    layout the C compiler gives the header's struct (plain C: the header must also compile as C)."""
    import subprocess

    from cogdl_amd import _lib

    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cogdl_hip.h"\n#include "cogdl_host.h"\n'
                   'int main(void) { printf("%zu %zu %zu %d %lld\\n", sizeof(cogdl_hip_segments), offsetof(cogdl_hip_segments, row), '
                   'offsetof(cogdl_hip_segments, edge), COGDL_HIP_MAX_SEGMENTS, (long long)COGDL_HIP_SEGMENT_MAX_EDGES); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, off_row, off_edge, max_seg, max_edges = (int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert size == ctypes.sizeof(_lib.Segments)
    assert off_row == _lib.Segments.row.offset and off_edge == _lib.Segments.edge.offset
    assert max_seg == _lib.MAX_SEGMENTS and max_edges == 2 ** 31 - 2 ** 20
