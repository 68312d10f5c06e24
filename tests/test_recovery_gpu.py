"""GPU: the reference's structure-recovery test
(pytest/structure_learning/test_exhaustive_search.py:13-30) through the module
under its own name: cancer / asia / a 20-variable CHILD-shaped network, N = 1e5,
BDeu(1.0), max_parents = 3, SHD == 0 against the generating graph.  The data
come from this package's generators (pgmpy's example models are not available
offline); the distance is the reference's (helpers/structural_distance.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(name):
    from openbnsl_amd import synth
    if name == "cancer":
        codes, card, names, truth = synth.cancer(100_000, seed=0)
        labels = [["False", "True"], ["False", "True"], ["high", "low"], ["False", "True"],
                  ["negative", "positive"]]
    elif name == "asia":
        codes, card, names = synth.asia(100_000, seed=0)
        idx = {k: i for i, k in enumerate(names)}
        truth = [[] for _ in names]
        for a, b in synth.ASIA_EDGES:
            truth[idx[b]].append(idx[a])
        labels = [["no", "yes"]] * 8
    else:
        codes, card, names, truth = synth.child_shaped(100_000, seed=0)
        labels = None
    return codes, card, names, truth, labels


@pytest.mark.parametrize("model_name", ["cancer", "asia", "child"])
def test_exhaustive_search(model_name):
    import openbnsllib                                   # the reference's module name
    from openbnsllib.base import PDAG
    from openbnsl_amd import graphs, synth
    codes, card, names, truth, labels = _case(model_name)
    samples = synth.to_dataframe(codes, names, labels=labels)
    samples = samples[sorted(samples.columns)]
    df_wrapper = openbnsllib.base.DataframeWrapper(samples)
    assert df_wrapper.num_of_values == card.tolist()
    _pdag = openbnsllib.structure_learning.exhaustive_search(
        df_wrapper, openbnsllib.score.BDeu(1.0), max_parents=3)
    assert isinstance(_pdag, PDAG) and _pdag.num_vars == len(names)
    errors = graphs.structural_errors(truth, _pdag)
    assert errors["SHD"] == 0, f"Failed for model {model_name} with Errors: {errors}"
    # edge extraction as helpers/pgmpy_bridge.py:25-28
    pairs = graphs.directed_pairs(_pdag)
    assert len(pairs) == sum(len(p) for p in graphs.parent_lists(_pdag))


def test_device_search_on_22_variables_equals_the_host_restatement():
    """The subset dynamic program on the device (one launch per subset size) against
    the host restatement of the reference's loop on the same score tables: same
    graph, same total, at a size where the tables are 22 * 2^21 entries per child."""
    import openbnsllib
    from openbnsl_amd import engine, graphs, synth
    cards = [2, 3, 2, 2, 3, 2, 4, 2, 3, 2, 2, 3, 2, 4, 2, 3, 2, 2, 3, 2, 2, 3]
    codes, card, _, _ = synth.random_bn(cards, 3_000, seed=22)
    n, mp = len(cards), 2
    d = engine.Data(codes, card)
    parents, total = d.exhaustive_search(mp)
    tables = [d.score_all(k).ravel().tolist() for k in range(1, mp + 2)]
    g, host_total = openbnsllib.structure_learning.search_from_tables(n, mp, tables)
    assert parents == graphs.parent_lists(g)
    assert total == host_total
    assert graphs.is_acyclic(parents) and all(len(p) <= mp for p in parents)
    d.close()
