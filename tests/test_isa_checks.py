"""Static checks on the compiled gfx950 code of the Cholesky kernels (hipcc cross-compiles here, no GPU): the properties
DESIGN.md states about the hot loops must hold for the code that ships, not for the code that was measured once."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_scratch_access_in_the_k_loops_and_none_in_the_wide_kernel():
    import check_isa

    problems = check_isa.check(verbose=False)
    assert not problems, "\n".join(problems)
