"""GPU: simulated_annealing end to end (scores from the HIP engine, chains in
the engine library) against the graphs the reference returned for the same
data, seeds and parameters (tests/golden), through the C ABI and through the
drop-in module."""
import os

import numpy as np
import pytest

from openbnsl_amd import engine, synth

pytestmark = pytest.mark.gpu

FRAMES = {
    "asia20k": lambda: synth.asia(20000, seed=0)[:2],
    "mixed12_3k": lambda: synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 2], 3000, seed=11,
                                          n_arcs=16)[:2],
}


@pytest.fixture(scope="module")
def datasets():
    out = {}
    for name, make in FRAMES.items():
        codes, card = make()
        out[name] = (codes, card, engine.Data(codes, card))
    yield out
    for _, _, d in out.values():
        d.close()


@pytest.mark.parametrize("mode", ["tables", "steps"])
def test_matches_reference_graphs(golden, datasets, mode, monkeypatch):
    """mode: complete score tables in one GPU pass, or one GPU table per step."""
    monkeypatch.setenv("OBNSL_ANNEAL_SCORES", mode)
    meta, _ = golden
    for c in meta["simulated_annealing"]["cases"]:
        if mode == "steps" and c["max_iters"] > 3000:
            continue
        _, _, d = datasets[c["frame"]]
        have, total = d.simulated_annealing(c["max_parents"], c["max_iters"], c["init_temp"],
                                            c["cooling_rate"], c["is_deterministic"], c["seed"],
                                            c["num_chains"], ess=c["ess"])
        assert have == c["parents"], (mode, c)
        assert np.isfinite(total) and total < 0


def test_total_is_the_score_of_the_returned_graph(datasets):
    import oracle
    codes, card, d = datasets["mixed12_3k"]
    have, total = d.simulated_annealing(3, 4000, seed=9, num_chains=4, is_deterministic=True)
    want = 0.0
    for v, ps in enumerate(have):
        vs = sorted(ps + [v])
        cnt = oracle.count(codes, card, vs)
        want += oracle.local_score(cnt, card[vs], vs.index(v))
    assert abs(total - want) <= 1e-9 * abs(want)


def test_bic_chain_runs_and_respects_the_bound(datasets):
    codes, card, d = datasets["asia20k"]
    have, total = d.simulated_annealing(2, 2000, seed=3, num_chains=2, kind=engine.BIC)
    assert all(len(p) <= 2 for p in have) and total < 0


def test_drop_in_module(golden, datasets):
    from openbnsl_amd import openbnsllib as ob
    meta, _ = golden
    codes, card, _ = datasets["asia20k"]
    names = [f"c{v:02d}" for v in range(8)]
    df = synth.to_dataframe(codes, names, labels=[[f"s{j}" for j in range(int(c))] for c in card])
    w = ob.base.DataframeWrapper(df)
    for c in meta["simulated_annealing"]["cases"]:
        if c["frame"] != "asia20k":
            continue
        g = ob.structure_learning.simulated_annealing(
            w, ob.score.BDeu(c["ess"]), max_parents=c["max_parents"], max_iters=c["max_iters"],
            init_temp=c["init_temp"], cooling_rate=c["cooling_rate"],
            is_deterministic=c["is_deterministic"], seed=c["seed"], num_chains=c["num_chains"])
        have = [[p for p in range(8) if g.has_edge(p, v)] for v in range(8)]
        assert have == c["parents"]
    with pytest.raises(ValueError, match="max_parents out of range"):
        ob.structure_learning.simulated_annealing(w, ob.score.BDeu(1.0), max_parents=8)
    with pytest.raises(ValueError, match="Unsupported score type"):
        ob.structure_learning.simulated_annealing(w, ob.score.K2(), max_parents=2)
    # defaults of the reference's binding
    g = ob.structure_learning.simulated_annealing(w, ob.score.BDeu(1.0), 2)
    assert g.num_vars == 8
