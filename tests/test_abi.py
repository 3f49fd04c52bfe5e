"""The C-ABI library builds for gfx950, loads, and exports every symbol include/oc_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from optimal_crowds_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from optimal_crowds_amd import _lib
    hdr = open(os.path.join(REPO, "include", "oc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(oc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name)


def test_abi_version_and_error_string(lib):
    assert lib.oc_abi_version() == 3
    assert isinstance(lib.oc_last_error(), bytes)


def test_struct_layout_matches_header():
    """ctypes mirrors of oc_config / oc_hjb_stats / oc_agents have the C sizes (checked with gcc)."""
    import subprocess
    import tempfile
    from optimal_crowds_amd import _lib
    src = ('#include <stdio.h>\n#include "oc_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(oc_config), '
           'sizeof(oc_hjb_stats), sizeof(oc_agents));printf("%zu %zu\\n", sizeof(oc_hjb_ring_job), sizeof(oc_hjb_ring));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.OcConfig), ctypes.sizeof(_lib.OcHjbStats), ctypes.sizeof(_lib.OcAgents),
                     ctypes.sizeof(_lib.OcHjbRingJob), ctypes.sizeof(_lib.OcHjbRing)]


def test_no_gpu_fails_loudly(lib):
    """Without a HIP device the product path raises instead of falling back to a CPU implementation."""
    import json
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from optimal_crowds_amd import _lib
    from optimal_crowds_amd.engine import Engine
    cfg = json.load(open(os.path.join(REPO, "optimal_crowds_amd", "config.json")))
    with pytest.raises(_lib.OcError):
        Engine(cfg, 5, 4, 100, 80, np.linspace(0, 5, 100), np.linspace(0, 4, 80), 1, 10)
