"""Behaviours of the reference's SpectrumModel object API (parameter store, freeze / thaw, persistence, error
conventions, caches, str) that its own suite exercises (tests/test_models/test_models.py), restated against this
package's SpectrumModel on a small synthetic order.  Book-keeping checks run on CPU (the device state is built
lazily); everything that evaluates the model is marked gpu."""
import os
from datetime import datetime

import numpy as np
import pytest

from starfish_amd import Spectrum, synth
from starfish_amd._flatdict import FlatterDict
from starfish_amd.emulator import Emulator
from starfish_amd.models import SpectrumModel

GP = [6050.0, 4.2, -0.3]


def make_emulator(o, name="mock emulator"):
    emu = Emulator(o["grid_points"], o["param_names"], o["emu_wl"], o["weights"], o["eigenspectra"], o["w_hat"],
                   o["flux_mean"], o["flux_std"], 1.0 + 0.01 * np.arange(len(o["grid_points"])), name=name)
    emu._trained = True
    return emu


@pytest.fixture
def model():
    o = synth.make_order(N=200, m=4, seed=8)
    w = o["wave"]
    data = Spectrum(w, o["flux"], sigmas=o["sigma"], name="mock order")
    return SpectrumModel(
        make_emulator(o), data, grid_params=GP, vz=0, Av=0, log_scale=-10, vsini=30,
        global_cov=dict(log_amp=-9.0, log_ls=2.0),
        local_cov=[dict(mu=float(w[60]), log_amp=-8.0, log_sigma=2.0), dict(mu=float(w[150]), log_amp=-8.5, log_sigma=2.0)],
        cheb=[0.1, -0.2],
    )


# ------------------------------------------------------------------------------------------ parameter store (CPU)
def test_item_access_mirrors_the_constructor(model):
    assert [model[k] for k in ("T", "logg", "Z")] == GP
    assert (model["vz"], model["Av"], model["log_scale"], model["vsini"]) == (0, 0, -10, 30)
    assert model["cheb"] == [0.1, -0.2]
    assert np.all(model.grid_params == GP)
    assert [k for k in model.params if k.startswith("cheb")] == ["cheb:1", "cheb:2"]
    assert set(model["global_cov"]) == {"log_amp", "log_ls"}
    assert len(model.params.as_dict()["local_cov"]) == 2 and "log_sigma" in model["local_cov"]["1"]
    flat = model.get_param_dict(flat=True)
    for key in ("global_cov:log_amp", "local_cov:0:log_amp", "local_cov:1:mu"):
        assert key in flat
    assert sorted(model.labels) == sorted(flat)


def test_cheb_assignment_and_gap_filling(model):
    model["cheb"] = [-0.2, 0.1]
    assert (model["cheb:1"], model["cheb:2"]) == (-0.2, 0.1)
    with pytest.raises(KeyError):
        model["cheb:0"] = 1  # the constant term is fixed
    model["cheb:4"] = 0.05  # skipped index: the gap is filled with zeros
    assert list(model.cheb) == [-0.2, 0.1, 0, 0.05] and model["cheb:3"] == 0


@pytest.mark.parametrize("bad", ["garbage", "global_cov:not quite", "global_cov:garbage", "local_cov:garbage"])
def test_unknown_keys_are_rejected(model, bad):
    with pytest.raises(KeyError):
        model[bad] = -4


def test_setitem_rebuilds_the_same_store(model):
    original, model.params = model.params, FlatterDict()
    for key, value in original.items():
        model[key] = value
    assert list(model.params.values()) == list(original.values())


@pytest.mark.parametrize("flat", [False, True])
def test_param_dict_and_vector_round_trips(model, flat):
    P0 = model.get_param_dict(flat=flat)
    model.set_param_dict(P0)
    assert model.get_param_dict(flat=flat) == P0
    store = model.params
    v = model.get_param_vector()
    model.set_param_vector(v)
    assert np.allclose(model.get_param_vector(), v) and model.params == store
    v[2] = 7
    model.set_param_vector(v)
    assert model[model.labels[2]] == 7
    with pytest.raises(ValueError):
        model.set_param_vector(np.append(v, 7))


@pytest.mark.parametrize("names", ["vsini", "logg", "global_cov:log_amp", "local_cov:0:log_amp",
                                   ["global_cov:log_amp", "global_cov:log_ls"]])
def test_freeze_removes_from_the_vector_and_thaw_restores(model, names):
    listed = [names] if isinstance(names, str) else names
    before = {k: model[k] for k in listed}
    assert all(k in model.labels for k in listed)
    model.freeze(names)
    assert all(k not in model.labels and k not in model.get_param_dict(flat=True) for k in listed)
    model.thaw(names)
    assert all(k in model.labels and model[k] == before[k] for k in listed)


def test_frozen_parameters_ignore_set_param_dict(model):
    P0 = model.get_param_dict()
    model.freeze("Z")
    P0["Z"] = 7
    model.set_param_dict(P0)
    assert model["Z"] == GP[2]


@pytest.mark.parametrize("group", ["global_cov", "local_cov", "cheb"])
def test_group_freeze_lists_the_group_and_its_members(model, group):
    members = [l for l in model.labels if l.startswith(group)]
    model.freeze(group)
    assert group in model.frozen and all(l in model.frozen for l in members)
    model.thaw(group)
    assert group not in model.frozen and not any(l in model.frozen for l in members)


def test_freeze_all_and_unknown_names(model):
    labels = model.labels
    model.freeze("all")
    assert set(model.frozen) == set(labels + ("global_cov", "local_cov", "cheb")) and model.labels == ()
    model.thaw("all")
    assert set(model.labels) == set(labels)
    fr = list(model.frozen)
    model.freeze("pinguino")
    model.thaw("pinguino")
    assert model.frozen == fr


def test_toml_persistence(model, tmp_path):
    path = os.path.join(tmp_path, "model.toml")
    model.freeze(["logg", "vsini", "global_cov"])
    model.set_param_vector(model.get_param_vector())  # values become numpy scalars: must still be written as numbers
    store, frozen, thawed = model.params, list(model.frozen), model.get_param_dict()
    model.save(path, metadata={"name": "Test Model", "date": datetime.today()})
    text = open(path).read()
    assert "[metadata]" in text and 'name = "Test Model"' in text
    model.load(path)
    assert model.params == store and model.frozen == frozen and model.get_param_dict() == thawed


def test_construction_errors(model):
    o = synth.make_order(N=64, m=2, seed=1)
    two = Spectrum(np.vstack([o["wave"], o["wave"] + 500]), np.vstack([o["flux"], o["flux"]]),
                   np.vstack([o["sigma"], o["sigma"]]))
    with pytest.raises(ValueError):
        SpectrumModel(make_emulator(o), two, grid_params=GP)

    class NoLogpdf:
        pass

    with pytest.raises(ValueError):
        model.train({"penguin": NoLogpdf()}, options={"maxiter": 1})  # not a parameter of the model
    with pytest.raises(ValueError):
        model.train({"T": lambda x: 1 / x}, options={"maxiter": 1})  # no logpdf method


# ------------------------------------------------------------------------------------------ evaluation (GPU)
@pytest.mark.gpu
def test_call_likelihood_and_perfect_fit(model):
    flux, cov = model()
    assert flux.shape == model.data.wave.shape and cov.shape == (len(flux), len(flux))
    lnprob = model.log_likelihood()
    assert np.isfinite(lnprob)
    model.data._flux = flux  # a model that reproduces the data exactly is more likely
    assert model.log_likelihood() > lnprob


@pytest.mark.gpu
def test_covariance_caches_follow_freeze_and_delete(model):
    assert model._glob_cov is None and model._loc_cov is None
    model()
    glob, loc = model._glob_cov, model._loc_cov
    assert glob.shape == loc.shape
    model.freeze("local_cov")
    assert model._loc_cov is None  # freezing a group forgets its cache ...
    model()
    assert np.allclose(model._loc_cov, loc) and np.allclose(model._glob_cov, glob)  # ... the next call refills it
    model.freeze("global_cov")
    assert model._glob_cov is None and np.allclose(model._loc_cov, loc)
    model()
    assert np.allclose(model._glob_cov, glob)
    del model["global_cov"]
    assert "global_cov" not in model.params and "global_cov" not in model.frozen and model._glob_cov is None


@pytest.mark.gpu
def test_norm_multiplies_the_flux_by_the_interpolated_factor(model):
    F1, _ = model()
    model.norm = True
    F2, _ = model()
    assert np.allclose(F1 * model.emulator.norm_factor(model.grid_params), F2)


@pytest.mark.gpu
def test_str_lists_thawed_then_frozen_and_the_fitted_scale(model):
    model.freeze("logg")
    text = str(model)
    lines = text.splitlines()
    assert lines[0] == "SpectrumModel" and lines[1] == "-" * 13
    assert f"Data: {model.data_name}" in lines and f"Emulator: {model.emulator.name}" in lines
    assert "  cheb: [0.1, -0.2]" in lines and "  T: 6050.0" in lines
    assert any(l.startswith("    0: mu: ") and "log_amp: -8.0, log_sigma: 2.0" in l for l in lines)
    assert lines[-2:] == ["Frozen Parameters", "  logg: 4.2"]
    del model["log_scale"]  # the scale is then fitted per call and reported as such
    model.log_likelihood()
    assert f"  log_scale: {model._log_scale} (fit)" in str(model).splitlines()
