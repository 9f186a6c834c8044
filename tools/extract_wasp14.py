"""
Extract ONE order of the reference's bundled WASP14 spectrum (data file, not source) into a small
npz fixture:  /opt/conda/bin/python3.9 tools/extract_wasp14.py   (that interpreter has h5py).

Source data: /root/reference/data/WASP14/WASP14-2009-06-14.hdf5 (legacy keys wls/fls/sigmas,
fp32 flux/sigma) and WASP14_23.mask.npy  (cfg 1).  Flux and sigma are divided
by the median flux so that sigma^2 is not swamped by the absolute 1e-10 jitter.
"""
import os

import h5py
import numpy as np

ORDER = 23
here = os.path.dirname(os.path.abspath(__file__))
src = "/root/reference/data/WASP14"
with h5py.File(os.path.join(src, "WASP14-2009-06-14.hdf5"), "r") as f:
    wl = np.asarray(f["wls"][ORDER], dtype=np.float64)
    fl = np.asarray(f["fls"][ORDER], dtype=np.float64)
    sg = np.asarray(f["sigmas"][ORDER], dtype=np.float64)
mask = np.load(os.path.join(src, f"WASP14_{ORDER}.mask.npy")).astype(bool)
med = np.median(fl)
out = os.path.join(here, "..", "tests", "golden", "wasp14_order23.npz")
np.savez_compressed(out, wave=wl, flux=fl / med, sigma=sg / med, mask=mask, order=ORDER)
print("wrote", out, wl.shape, mask.sum(), wl[mask].min(), wl[mask].max())
