"""Helpers shared by the GPU parity tests: build a DeviceOrder from a synthetic order and pack
oracle-style parameter dicts into C-ABI parameter rows."""
import numpy as np

from oracle import sf_oracle as O
from starfish_amd import _device as D


def oracle_order(o, **kw):
    return O.OracleOrder(
        o["wave"], o["flux"], o["sigma"], o["emu_wl"], o["eigenspectra"], o["flux_mean"],
        o["flux_std"], o["grid_points"], o["w_hat"], **kw
    )


def device_order(oo):
    """DeviceOrder fed with exactly the static arrays the oracle holds (same bulk_fluxes, v11)."""
    return D.DeviceOrder(
        oo.wave, oo.flux, oo.sigma, oo.min_dv_wave, oo.bulk_fluxes, oo.grid_points, oo.variances,
        oo.lengthscales, oo.v11, oo.w_hat,
    )


def model_desc(dev_order, p):
    return dev_order.model_desc(
        "vsini" in p, "vz" in p, "log_scale" in p, "global_cov" in p, len(p.get("local_cov", [])),
        len(p.get("cheb", [])), use_sigma_w=p.get("emulator_cov", "code") == "paper", has_av="Av" in p,
    )


def pack_rows(dev_order, plist):
    """plist: list of oracle parameter dicts with identical structure."""
    p0 = plist[0]
    md = model_desc(dev_order, p0)
    stride = dev_order.param_stride(md)
    rows = np.zeros((len(plist), stride))
    P = dev_order.P
    for r, p in zip(rows, plist):
        r[0] = p.get("vsini", 0.0)
        r[1] = p.get("vz", 0.0)
        r[2] = p.get("log_scale", 0.0)
        r[3] = p.get("norm", 1.0)
        if "global_cov" in p:
            r[4], r[5] = p["global_cov"]
        r[6 : 6 + P] = p["grid"]
        nc = len(p.get("cheb", []))
        r[6 + P : 6 + P + nc] = p.get("cheb", [])
        nl = len(p.get("local_cov", []))
        for k, (mu, la, ls) in enumerate(p.get("local_cov", [])):
            r[6 + P + nc + 3 * k : 6 + P + nc + 3 * k + 3] = (mu, la, ls)
        if "Av" in p:
            r[6 + P + nc + 3 * nl] = p["Av"]
    return md, rows
