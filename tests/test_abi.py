"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares."""
import os
import re

import pytest

from gabotorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """The .so is a build artefact (git-ignored): build it with hipcc when a fresh checkout has none."""
    if not os.path.exists(_lib.LIB_PATH):
        from gabotorch_amd import _build
        _build.build()


def _declared():
    text = open(os.path.join(ROOT, "include", "gabo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(gabo_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert names, "no declarations parsed from include/gabo_hip.h"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gabo_hip.h but not exported by libgabo_hip.so"
    assert names == set(_lib.SIGNATURES), (names ^ set(_lib.SIGNATURES))


def test_host_side_argument_errors_need_no_gpu():
    lib = _lib.load()
    assert lib.gabo_version() >= 100
    assert lib.gabo_spd_ai_workspace_bytes(2, 3, 4, 10) == 2 * 7 * 55 * 8
    assert lib.gabo_spd_ai_workspace_bytes(2, 3, 4, 20) == 2 * 2 * 3 * 400 * 8      # wave-per-pair fallback layout
    # argument validation happens before any HIP call
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 4, 4, 40, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 4, 4, 3, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_ERR_ARG
    assert lib.gabo_spd_ai_pairwise(None, None, None, None, 1, 0, 4, 3, 0, 0, 1.0, 0, None, 0, None, None) == _lib.GABO_OK
    assert lib.gabo_sphere_pairwise(None, None, None, 1, 4, 4, 0, 0, 0, 1.0, 0, 0, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_mandel_to_matrix(None, None, 3, 65, None) == _lib.GABO_ERR_DIM
    assert lib.gabo_mandel_to_matrix(None, None, 0, 5, None) == _lib.GABO_OK
