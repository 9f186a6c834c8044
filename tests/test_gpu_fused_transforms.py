"""With a log_scale parameter the scale factor Omega does not depend on the flux (spectrum_model.py:316-318), so the rows of
X (spectrum_model.py:312), the reconstruction (:313), the rescaling (transforms.py:231), the residual (:402) and the rank-m
factor Y come out of ONE pass over the pixels (k_eval_resid_y) instead of k_eval_rows -> k_scale -> k_resid_y.  The fused
kernel performs the same operations in the same order: every output of the forward call and the likelihood are bit-identical
to the three launches (tuning build, SF_TRANSFORM_UNFUSED).  Needs an MI355X: run with -m gpu."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TUNING_LIB = os.path.join(ROOT, "starfish_amd", "libstarfish_amd_tuning.so")

_CODE = r"""
import json, hashlib, numpy as np
from starfish_amd import synth
out = {}
for name, kw in (("cfg1", dict(N=1024, m=4, seed=5)), ("odd", dict(N=1500, m=6, seed=9))):
    o = synth.make_order(**kw)
    model = synth.build_model(o)
    P = synth.walker_ball(o, B=12, seed=3)
    P[5, synth.LABELS.index("T")] = 1.0e6  # a walker the emulator refuses (outside the grid): its rows are zeroed on both paths
    lnl, info = model.log_likelihood_batch(P, return_info=True)
    model.set_param_vector(P[2])
    flux, cov = model()
    h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    out[name] = dict(lnl=[float(v) if np.isfinite(v) else None for v in lnl], info=info.tolist(), flux=h(flux), cov=h(cov),
                     scalar=float(model.log_likelihood()))
print(json.dumps(out))
"""


def _run(**env):
    if not os.path.exists(TUNING_LIB):  # (normally built by __graft_entry__.build() and shipped with the tree)
        subprocess.run(["make", "-C", os.path.join(ROOT, "starfish_amd", "csrc"), "-j8", "TUNING=1"], check=True,
                       capture_output=True)
    e = dict(os.environ, SF_LIB_PATH=TUNING_LIB, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _CODE], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_one_pass_rows_scale_residual_equals_the_three_launches_bit_for_bit():
    fused = _run()
    three = _run(SF_TRANSFORM_UNFUSED="1")
    assert fused == three
    for case in fused.values():
        assert case["info"][5] != 0 and case["lnl"][5] is None
        assert all(i == 0 for k, i in enumerate(case["info"]) if k != 5)
