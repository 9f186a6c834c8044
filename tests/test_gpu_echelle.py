"""Multi-order path (BASELINE cfg 3): EchelleModel through sf_loglike_multi_batch against
per-order values of the REAL reference (tests/golden/model_cfg3.npz, tools/gen_golden.py cfg3) and against the oracle.
Run with -m gpu."""
import threading

import numpy as np
import pytest

from conftest import load_golden
from gpu_helpers import oracle_order
from oracle import sf_oracle as O
from starfish_amd import _device as D
from starfish_amd import synth

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-8):
    return np.all(np.abs(np.asarray(a) - np.asarray(b)) <= rtol * np.abs(b) + 1e-8)


def test_cfg3_echelle_vs_reference_goldens():
    """25 orders x N = 3000: per-order lnL and their sum against the reference's SpectrumModel values."""
    g = load_golden("model_cfg3.npz")
    n_orders, N = int(g["n_orders"][0]), int(g["N"][0])
    orders = synth.make_echelle(n_orders, N, seed0=int(g["seed0"][0]))
    em = synth.build_echelle(orders)
    assert em.labels == tuple(g["labels"]) == synth.SHARED_LABELS
    P = g["P"]
    total, info, per_order = em.log_likelihood_batch(P, return_info=True, return_orders=True)
    assert (info == 0).all()
    assert per_order.shape == (n_orders, len(P))
    assert close(per_order, g["lnl"]), np.max(np.abs(per_order - g["lnl"]) / np.abs(g["lnl"]))
    assert close(total, g["lnl"].sum(axis=0))
    # the scalar API, one order at a time (the reference's way to use the model), gives the same sum
    em.set_param_vector(P[0])
    assert close(em.log_likelihood(), g["lnl"][:, 0].sum())
    # and the multi-order pass equals the per-order batched passes to rounding (same kernels and padding; the
    # split-K factor of the under-filled launches depends on the batch size, so not bit for bit)
    serial = np.array([m.log_likelihood_batch(P) for m in em.orders])
    np.testing.assert_allclose(serial, per_order, rtol=1e-12)


def test_echelle_orders_of_different_length_vs_oracle():
    """Orders with different pixel counts share one factorisation (identity padding to the longest); checked
    against the CPU oracle, order by order."""
    sizes = [200, 333, 256, 129]
    orders = [synth.make_order(N=n, m=4, seed=40 + i, wave0=5000.0 * 1.02**i) for i, n in enumerate(sizes)]
    em = synth.build_echelle(orders)
    P = synth.shared_ball(orders[0], B=5, seed=9)
    total, per_order = em.log_likelihood_batch(P, return_orders=True)
    for i, o in enumerate(orders):
        oo = oracle_order(o)
        want = np.array([O.log_likelihood(oo, synth.shared_to_oracle_params(o, p)) for p in P])
        assert close(per_order[i], want), (i, per_order[i], want)
    assert close(total, per_order.sum(axis=0), rtol=1e-14)
    # out-of-grid walker: -inf for the model, code -1, the others untouched
    P2 = P.copy()
    P2[2, synth.SHARED_LABELS.index("T")] = 9000.0
    t2, info2 = em.log_likelihood_batch(P2, return_info=True)
    assert t2[2] == -np.inf and info2[2] == -1
    keep = [0, 1, 3, 4]
    np.testing.assert_allclose(t2[keep], total[keep], rtol=1e-12)


def test_echelle_structured_solver_matches_the_dense_model():
    """solver="auto" on the orders of a multi-order model: band + Woodbury per order, same sum as the one-pass dense
    evaluation (and the same handling of a walker that leaves the grid)."""
    sizes = [512, 300, 448]
    orders = [synth.make_order(N=n, m=4, seed=60 + i, wave0=5000.0 * 1.02**i) for i, n in enumerate(sizes)]
    em = synth.build_echelle(orders)
    P = synth.shared_ball(orders[0], B=6, seed=3)
    P[4, synth.SHARED_LABELS.index("T")] = 9000.0
    dense, info_d, per_d = em.log_likelihood_batch(P, return_info=True, return_orders=True)
    for m in em.orders:
        m.solver = "auto"
    auto, info_a, per_a = em.log_likelihood_batch(P, return_info=True, return_orders=True)
    ok = np.arange(6) != 4
    np.testing.assert_allclose(auto[ok], dense[ok], rtol=1e-10)
    np.testing.assert_allclose(per_a[:, ok], per_d[:, ok], rtol=1e-10)
    assert auto[4] == dense[4] == -np.inf and info_a[4] == info_d[4] == -1 and (info_a[ok] == 0).all()


def test_multi_chunking_matches_single_pass():
    """A workspace cap forces the unit list through several sf_loglike_multi_batch calls: same values (to the
    rounding of a different split-K factor)."""
    orders = [synth.make_order(N=192, m=4, seed=60 + i, wave0=5000.0 * 1.02**i) for i in range(3)]
    em = synth.build_echelle(orders)
    P = synth.shared_ball(orders[0], B=7, seed=3)
    packed = [m._pack(P, update_caches=False) for m in em.orders]
    devs = [p[0] for p in packed]
    rows = [p[2] for p in packed]
    md = packed[0][1]
    one = D.loglike_multi(devs, md, rows)
    for cap in (4, 7, 10):
        many = D.loglike_multi(devs, md, rows, max_units=cap)
        for a, b in zip(one, many):
            np.testing.assert_allclose(a["lnl"], b["lnl"], rtol=1e-12)
            np.testing.assert_array_equal(a["info"], b["info"])


def test_two_contexts_on_two_threads_match_serial():
    """contexts are independent -- two models driven from two host threads at the same time give
    bit-identical results to the same calls made one after the other."""
    o1 = synth.make_order(N=640, m=4, seed=71)
    o2 = synth.make_order(N=512, m=8, seed=72, wave0=5200.0)
    m1, m2 = synth.build_model(o1), synth.build_model(o2)
    P1, P2 = synth.walker_ball(o1, B=24, seed=5), synth.walker_ball(o2, B=16, seed=6)
    want1 = [m1.log_likelihood_batch(P1) for _ in range(2)]
    want2 = [m2.log_likelihood_batch(P2) for _ in range(2)]
    np.testing.assert_array_equal(want1[0], want1[1])
    got, errs = {}, []

    def run(key, model, P):
        try:
            import torch

            with torch.cuda.stream(torch.cuda.Stream()):
                got[key] = [model.log_likelihood_batch(P) for _ in range(6)]
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=run, args=(1, m1, P1)), threading.Thread(target=run, args=(2, m2, P2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for r in got[1]:
        np.testing.assert_array_equal(r, want1[0])
    for r in got[2]:
        np.testing.assert_array_equal(r, want2[0])


def test_orders_with_different_descriptors_vs_oracle():
    """orders whose parameter-row layouts differ -- another number of frozen local kernels, no
    local kernel at all -- expose the same thawed labels but must not share one sf_loglike_multi_batch call (the
    C-ABI reads all segments of a call with ONE ModelDesc).  Per-order values against the oracle."""
    orders = [synth.make_order(N=n, m=4, seed=80 + i, wave0=5000.0 * 1.02**i) for i, n in enumerate((256, 320, 192, 256))]
    kernels = []
    models = []
    for i, o in enumerate(orders):
        c = dict(synth.centre_params(o))
        w = o["wave"]
        if i == 1:  # two local kernels
            c["local_cov"] = list(c["local_cov"]) + [dict(mu=float(w[len(w) // 2]), log_amp=-8.5, log_sigma=float(np.log(10.0)))]
        if i == 2:  # none
            del c["local_cov"]
        kernels.append([(k["mu"], k["log_amp"], k["log_sigma"]) for k in c.get("local_cov", [])])
        models.append(synth.build_model(o, params=c, freeze=("local_cov",) if "local_cov" in c else ()))
    from starfish_amd.models import EchelleModel

    em = EchelleModel.from_orders(models)
    assert em.labels == synth.SHARED_LABELS
    strides = {m._pack(None)[2].shape[1] for m in models}
    assert len(strides) == 3  # really different row layouts
    P = synth.shared_ball(orders[0], B=5, seed=4)
    total, info, per_order = em.log_likelihood_batch(P, return_info=True, return_orders=True)
    assert (info == 0).all()
    for i, o in enumerate(orders):
        oo = oracle_order(o)
        for b, p in enumerate(P):
            q = synth.shared_to_oracle_params(o, p)
            q["local_cov"] = kernels[i]
            if not kernels[i]:
                del q["local_cov"]
            want = O.log_likelihood(oo, q)
            assert close(per_order[i, b], want), (i, b, per_order[i, b], want)
    assert close(total, per_order.sum(axis=0), rtol=1e-14)
    # a MultiPlan asked to mix the layouts refuses
    packed = [m._pack(P, update_caches=False) for m in models]
    with pytest.raises(ValueError):
        D.MultiPlan([p[0] for p in packed], packed[0][1], [p[2] for p in packed])
