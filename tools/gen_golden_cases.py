"""Case tables shared by tools/gen_golden.py (reference side) and the parity tests."""
import numpy as np

from starfish_amd import synth

SMALL_CASES = {
    "full": {},
    "renorm": {"drop": ["log_scale"]},
    "bare": {"drop": ["vz", "vsini", "cheb", "global_cov", "local_cov"]},
    "no_local": {"drop": ["local_cov"]},
    "no_global": {"drop": ["global_cov"]},
    "two_local": {"extra_local": True},
    "cheb4": {"cheb": [0.01, -0.02, 0.005, 0.001]},
    "norm": {"norm": True},
    "norm_renorm": {"norm": True, "drop": ["log_scale"]},
}


FULL_COV_CASES = ("full", "two_local")
COV_ROWS = [0, 17, 85, 100, 255]


def small_case_params(o, spec):
    c = synth.centre_params(o)
    for k in spec.get("drop", []):
        c.pop(k)
    if spec.get("extra_local"):
        N = len(o["wave"])
        c["local_cov"] = c["local_cov"] + [
            dict(mu=float(o["wave"][(2 * N) // 3]), log_amp=-7.5, log_sigma=float(np.log(8.0)))
        ]
    if "cheb" in spec:
        c["cheb"] = spec["cheb"]
    return c


