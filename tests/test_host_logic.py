"""CPU-only tests: parameter book-keeping of SpectrumModel (mirrors the reference's
tests/test_models/test_models.py parameter tests), FlatterDict, grids, the TOML round trip, and that
the product refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from starfish_amd import Spectrum, synth
from starfish_amd._flatdict import FlatterDict
from starfish_amd.emulator import Emulator
from starfish_amd.models import SpectrumModel
from starfish_amd.utils import calculate_dv, create_log_lam_grid

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def make_model(**over):
    o = synth.make_order(N=128, m=3, seed=9)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    emu._trained = True
    data = Spectrum(o["wave"], o["flux"], sigmas=o["sigma"])
    c = synth.centre_params(o)
    c.update(over)
    gp = c.pop("grid_params")
    return SpectrumModel(emu, data, grid_params=gp, **c)


def test_label_order_matches_reference_convention():
    m = make_model()
    assert m.labels == synth.LABELS  # kwargs order, cheb moved last, then emulator param names
    np.testing.assert_allclose(m.get_param_vector(), synth.centre_vector(dict(wave=m.data.wave)))


def test_get_set_param_vector_roundtrip_and_length_check():
    m = make_model()
    P0 = m.get_param_vector()
    m.set_param_vector(P0 + 1)
    np.testing.assert_allclose(m.get_param_vector(), P0 + 1)
    with pytest.raises(ValueError):
        m.set_param_vector(P0[:-1])


def test_freeze_thaw_groups():
    m = make_model()
    m.freeze("global_cov")
    assert "global_cov:log_amp" not in m.labels and "global_cov" in m.frozen
    m.freeze("local_cov")
    assert not any(k.startswith("local_cov") for k in m.labels)
    m.freeze("cheb")
    assert not any(k.startswith("cheb") for k in m.labels)
    before = m["global_cov:log_amp"]
    m.set_param_dict({"global_cov": {"log_amp": 100.0}})
    assert m["global_cov:log_amp"] == before  # frozen values are not overwritten
    m.thaw(["global_cov", "local_cov", "cheb"])
    assert m.labels == synth.LABELS
    m.freeze("all")
    assert m.labels == ()
    m.thaw("all")
    assert m.frozen == []
    m.freeze("vz")
    m.set_param_vector(m.get_param_vector())
    assert "vz" not in m.labels and m["vz"] == 10.0


def test_setitem_getitem_delitem():
    m = make_model()
    m["vsini"] = 12.0
    assert m["vsini"] == 12.0
    m["global_cov:log_amp"] = -3.0
    assert m.params["global_cov"]["log_amp"] == -3.0
    m["cheb:4"] = 0.5
    assert m["cheb"] == [0.01, -0.02, 0, 0.5]
    with pytest.raises(KeyError):
        m["cheb:0"] = 1.0
    with pytest.raises(KeyError):
        m["garbage"] = 1.0
    with pytest.raises(KeyError):
        m["global_cov:garbage"] = 1.0
    m["cheb"] = [0.3]
    assert m["cheb"] == [0.3]
    del m["global_cov"]
    assert "global_cov" not in m.params and m._glob_cov is None
    with pytest.raises(KeyError):
        del m["global_cov"]
    m["T"] = 6100.0
    np.testing.assert_allclose(m.grid_params, [6100.0, 4.2, -0.3])


def test_emulator_covariance_form_switch_is_validated():
    """``emulator_cov``: "code" (X^T Sigma_w^-1 X, spectrum_model.py:334-335: default, parity target) or "paper"
    (Phi Sigma_w Phi^T, docs/api/emulator.rst:105); anything else is rejected at construction."""
    from starfish_amd import synth

    o = synth.make_order(N=128, m=4, seed=1)
    assert synth.build_model(o).emulator_cov == "code"
    assert synth.build_model(o, emulator_cov="paper").emulator_cov == "paper"
    with pytest.raises(ValueError, match="emulator_cov"):
        synth.build_model(o, emulator_cov="docs")


def test_multi_order_data_is_rejected():
    o = synth.make_order(N=64, m=2)
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"],
                   o["w_hat"], o["flux_mean"], o["flux_std"], o["factors"])
    data = Spectrum(np.vstack([o["wave"]] * 2), np.vstack([o["flux"]] * 2))
    with pytest.raises(ValueError):
        SpectrumModel(emu, data, grid_params=[6050, 4.2, -0.3])


def test_toml_roundtrip(tmp_path):
    m = make_model()
    m.freeze("vz")
    path = tmp_path / "model.toml"
    m.save(path, metadata={"note": "unit test"})
    m2 = make_model(vz=99.0)
    m2.load(path)
    assert m2.params == m.params
    assert m2.frozen == m.frozen
    assert m2["local_cov:0:mu"] == m["local_cov:0:mu"]


def test_flatterdict_semantics():
    fd = FlatterDict({"a": 1, "g": {"x": 2, "y": 3}, "l": [{"mu": 1.0}, {"mu": 2.0}]})
    assert fd.keys() == ["a", "g:x", "g:y", "l:0:mu", "l:1:mu"]
    assert "g" in fd and "g:x" in fd and "g:z" not in fd
    assert fd["l:1:mu"] == 2.0
    fd["g:y"] = 30
    assert fd["g"]["y"] == 30
    assert fd.as_dict()["l"] == [{"mu": 1.0}, {"mu": 2.0}]
    child = fd["g"]
    child["x"] = -2
    assert fd["g:x"] == -2
    del fd["g:x"]
    assert fd.keys() == ["a", "g:y", "l:0:mu", "l:1:mu"]
    flat = FlatterDict()
    flat["p:0:q"] = 5
    assert flat.as_dict() == {"p": {"0": {"q": 5}}}
    assert FlatterDict({"a": {"b": 1}}) == FlatterDict({"a": {"b": 1}})


def test_spectrum_containers():
    w = np.linspace(1e4, 4e4, 24).reshape(2, 12)
    s = Spectrum(w, np.ones_like(w), name="x")
    assert s.shape == (2, 12) and len(s) == 2 and s.sigmas.shape == (2, 12)
    r = s.reshape((4, 6))
    assert r.shape == (4, 6)
    s.shape = (1, 24)
    assert s.shape == (1, 24)
    mask = np.zeros(24, bool)
    mask[5:15] = True
    s1 = Spectrum(w.reshape(-1), np.arange(24.0), masks=mask)
    assert s1[0].wave.shape == (10,) and s1[0].flux[0] == 5.0


def test_log_lambda_grid_properties():
    g = create_log_lam_grid(2.0, 5000.0, 5100.0)
    wl = g["wl"]
    assert len(wl) & (len(wl) - 1) == 0
    assert calculate_dv(wl) <= 2.0
    assert abs(wl[0] - 5000.0) < 1e-9 and abs(wl[-1] - 5100.0) < 1e-9
    with pytest.raises(ValueError):
        create_log_lam_grid(2.0, 5100.0, 5000.0)
    with pytest.raises(ValueError):
        create_log_lam_grid(2.0, -1.0, 5000.0)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from starfish_amd import _lib

    lib = _lib.load()  # loading needs no GPU
    header = open(os.path.join(ROOT, "include", "starfish_amd.h")).read()
    declared = set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/starfish_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.sf_version().startswith(b"starfish_amd")


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from starfish_amd import _lib, transforms
    from starfish_amd.models.kernels import global_covariance_matrix

    with pytest.raises(_lib.StarfishAMDError):
        global_covariance_matrix(np.linspace(5000, 5001, 8), 1.0, 1.0)
    with pytest.raises(_lib.StarfishAMDError):
        transforms.rotational_broaden(np.linspace(5000, 5001, 8), np.ones(8), 10.0)
    with pytest.raises(_lib.StarfishAMDError):
        make_model().log_likelihood()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "starfish_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|sf_oracle\s+import|oracle/", text, re.M), \
                    f"{f} reaches into oracle/"


def test_band_halfwidth_bound_covers_the_true_support():
    """Host-side bound used to route walkers to the banded solver: never smaller than the true support of
    the reference kernels (oracle restatement of models/kernels.py), and tight to a few pixels."""
    from oracle import sf_oracle as O
    from starfish_amd._device import band_halfwidth_bound

    rng = np.random.default_rng(11)
    for n, dv in ((300, 2.0), (257, 3.7)):
      for jitter in (0.0, 0.2):
          wave = 5000.0 * np.exp(np.arange(n) * dv / 2.99792458e5)
          wave[1:-1] += rng.uniform(-jitter, jitter, n - 2) * np.diff(wave).min()  # not exactly log-uniform
          for ls, sig, mu_i in ((3.0, 6.0, n // 3), (9.0, 2.0, 5), (0.7, 11.0, n - 4)):
              row = np.zeros(6 + 3 + 2 + 3)
              row[5] = np.log(ls)
              row[6 + 3 + 2:] = (wave[mu_i] + 0.01, -8.0, np.log(sig))
              K = O.matern32_global(wave, 1.0, ls)
              L = O.gaussian_local(wave, 1.0, wave[mu_i] + 0.01, sig)
              for has_g, n_loc, M in ((1, 0, K), (0, 1, L), (1, 1, K + L)):
                  ii, jj = np.nonzero(M)
                  true_hw = int(np.abs(ii - jj).max())
                  got = int(band_halfwidth_bound(wave, row, 3, has_g, n_loc, 2)[0])
                  # always a bound; tight to a few pixels on a regular grid (it uses the smallest spacing)
                  assert true_hw <= got, (n, ls, sig, has_g, n_loc, true_hw, got)
                  if jitter == 0 and mu_i == n // 3:  # (a patch cut by the array edge is narrower than the bound)
                      assert got <= true_hw + 4, (n, ls, sig, has_g, n_loc, true_hw, got)
    # a grid that is not strictly increasing cannot use the banded solver at all
    assert band_halfwidth_bound(wave[::-1], row, 3, 1, 1, 2)[0] == np.iinfo(np.int32).max


def test_every_status_code_of_the_header_has_a_python_message():
    """include/starfish_amd.h is the contract: each negative per-item status (SF_INFO_*) must be known to the
    Python layer, which turns it into the reference's exceptions / messages."""
    import re

    from starfish_amd import _device as D

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "starfish_amd.h")).read()
    codes = {name: int(val) for name, val in re.findall(r"#define\s+(SF_INFO_[A-Z_]+)\s+\((-\d+)\)", text)}
    assert len(codes) >= 5, codes
    for name, val in codes.items():
        assert val in D.INFO_MESSAGES, (name, val)
    assert D.INFO_BANDWIDTH == codes["SF_INFO_BANDWIDTH"]


def test_echelle_from_orders_checks_labels_and_synth_helpers():
    """Host-only: EchelleModel.from_orders refuses orders whose thawed labels differ; the multi-order synthetic
    helpers keep the shared / per-order split the cfg 3 goldens were generated with."""
    from starfish_amd import synth
    from starfish_amd.models import EchelleModel

    orders = synth.make_echelle(3, N=64, m=2)
    assert orders[1]["wave"][0] == pytest.approx(5000.0 * 1.02)
    models = [synth.build_model(o, freeze=("local_cov",)) for o in orders]
    em = EchelleModel.from_orders(models)
    assert len(em) == 3 and em.labels == synth.SHARED_LABELS
    P = synth.shared_ball(orders[0], B=4)
    assert P.shape == (4, len(synth.SHARED_LABELS))
    p = synth.shared_to_oracle_params(orders[2], P[1])
    assert p["local_cov"][0][0] == pytest.approx(orders[2]["wave"][64 // 3])  # the order's OWN local kernel
    assert p["grid"] == list(P[1][-3:])
    models[1].thaw("local_cov")
    with pytest.raises(ValueError):
        EchelleModel.from_orders(models)
    with pytest.raises(ValueError):
        EchelleModel.from_orders([])
    # freeze / thaw bookkeeping of a group (table-driven implementation): members are listed once, thaw restores
    m = models[0]
    before = m.labels
    m.thaw("local_cov")
    assert "local_cov:0:mu" in m.labels and "local_cov" not in m.frozen
    m.freeze("local_cov")
    assert m.labels == before and m.frozen.count("local_cov:0:mu") == 1
    with pytest.raises(ValueError):
        m.thaw("global_cov")  # not frozen: list.remove raises, like the reference


def test_bench_builds_through_the_product_and_keeps_the_oracle_in_the_baseline_leg():
    """bench.py must not depend on the test tree, and may touch the oracle only inside the cpu_baseline functions."""
    import ast

    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "gpu_helpers" not in src and '"tests"' not in src
    tree = ast.parse(src)
    allowed = {"_cpu_pool_worker", "cpu_baseline"}
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
        name = getattr(fn, "name", "<module>")
        body = fn.body if isinstance(fn, ast.Module) else fn.body
        for node in body if isinstance(fn, ast.Module) else ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                assert name in allowed, f"oracle imported in {name}"
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                assert name in allowed, f"oracle imported in {name}"


def test_release_library_has_no_environment_switches():
    """The shipped library must not contain a knob that changes its results or launch sequence from the environment
    (round-2 finding: SF_PANEL_SKIP returned wrong likelihoods by design).  The tuning switches are compiled only
    under -DSF_TUNING (`make TUNING=1` -> libstarfish_amd_tuning.so); the sources reach the environment through
    the SF_TUNE_* macros alone."""
    from starfish_amd import _lib

    blob = open(os.path.join(ROOT, "starfish_amd", "libstarfish_amd.so"), "rb").read()
    for name in (b"SF_PANEL_SKIP", b"SF_CHOL_UNFUSED", b"SF_CHOL_SPLIT", b"SF_NO_LOOKAHEAD", b"SF_MULTI_FIRST",
                 b"SF_BAND_NO_TWIST", b"SF_BAND_TILES_POISON", b"SF_DIAG_SCRATCH"):
        assert name not in blob, name
    assert os.path.basename(_lib.LIB_PATH) == "libstarfish_amd.so"
    csrc = os.path.join(ROOT, "starfish_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            for ln in open(os.path.join(csrc, f)):
                if "getenv" in ln:
                    assert f == "sf_common.h" and "#define SF_TUNE_" in ln, (f, ln)


def test_a_library_that_lacks_a_declared_entry_point_does_not_load_silently(monkeypatch):
    """Advisor, round 5: with SF_LIB_PATH set, load() used to skip any missing symbol, so a stale build loaded without
    complaint and failed later inside an error path.  A missing symbol now raises at load time; an OLDER build for a
    same-box A/B has to be asked for (SF_ALLOW_OLD_LIB=1) and every skipped name is reported."""
    from starfish_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setitem(_lib.SIGNATURES, "sf_entry_point_of_a_newer_header", (None, []))
    monkeypatch.delenv("SF_ALLOW_OLD_LIB", raising=False)
    with pytest.raises(_lib.StarfishAMDError, match="does not export sf_entry_point_of_a_newer_header"):
        _lib.load()
    monkeypatch.setenv("SF_LIB_PATH", _lib.LIB_PATH)
    with pytest.raises(_lib.StarfishAMDError, match="stale or mismatched"):
        _lib.load()  # (the development hook alone does not relax it)
    monkeypatch.setenv("SF_ALLOW_OLD_LIB", "1")
    with pytest.warns(RuntimeWarning, match="lacks sf_entry_point_of_a_newer_header"):
        assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", None)


def test_header_documents_no_environment_switch_and_the_persistent_kernel_status():
    text = open(os.path.join(ROOT, "include", "starfish_amd.h")).read()
    assert "environment switches (tuning aids, read once)" not in text
    assert "reads NO environment variable" in text
    assert "sf_persistent_potrf_status" in text and "25 ms" in text
    blob = open(os.path.join(ROOT, "starfish_amd", "libstarfish_amd.so"), "rb").read()
    for name in (b"SF_DF_STALL_MS", b"SF_DF_MISS_CLAIMS", b"SF_DF_TIMEOUT_S", b"SF_WIDE_HEAD", b"getenv"):
        assert name not in blob, name


def test_multiplan_collect_raises_when_the_internal_status_survives_the_retry(monkeypatch):
    """Advisor, round 5: MultiPlan.collect re-ran an aborted call once and then handed a SECOND -5 back as an ordinary
    per-unit status.  Host logic only (no device): collect_multi is replaced by a fake that keeps returning -5."""
    from starfish_amd import _device as D

    class FakeLib:
        def __init__(self):
            self.switched = []

        def sf_persistent_potrf_status(self, buf):
            for i, v in enumerate((1, 2, 400, 512, 3_000_000, 77, 1, 1)):
                buf[i] = v
            return 0

        def sf_persistent_potrf(self, v):
            self.switched.append(v)
            return 1

    plan = D.MultiPlan.__new__(D.MultiPlan)
    plan.lib, plan.quad, plan.info, plan.sizes = FakeLib(), None, None, [2]
    calls = []
    plan.enqueue = lambda: calls.append("enqueue")
    bad = [dict(lnl=np.full(2, -np.inf), logdet=np.zeros(2), sqmah=np.zeros(2), log_scale=np.zeros(2),
                info=np.full(2, D.INFO_INTERNAL, dtype=np.int32))]
    monkeypatch.setattr(D, "collect_multi", lambda quad, info, sizes: bad)
    with pytest.warns(RuntimeWarning, match="no task of the launch completed for 25 ms; only 400 of its 512 workgroups had started"):
        with pytest.raises(RuntimeError, match="internal error"):
            plan.collect()
    assert calls == ["enqueue"] and plan.lib.switched == [0]
    # ... and a retry that succeeds is returned
    good = [dict(bad[0], info=np.zeros(2, dtype=np.int32), lnl=np.ones(2))]
    seq = iter([bad, good])
    monkeypatch.setattr(D, "collect_multi", lambda quad, info, sizes: next(seq))
    with pytest.warns(RuntimeWarning, match="disabled for this whole process"):
        out = plan.collect()
    assert (out[0]["info"] == 0).all()
