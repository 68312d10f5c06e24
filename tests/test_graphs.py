"""CPU: the structural distance of the reference's harness
(helpers/structural_distance.py:43-129, helpers/pgmpy_bridge.py:25-28) restated
without pgmpy, on hand-checked cases, and the recovery criterion of
pytest/structure_learning/test_exhaustive_search.py:25-30 on the oracle's search."""
import pytest

from openbnsl_amd import graphs


class _G:
    """Minimal PDAG-like object (num_vars, has_edge)."""

    def __init__(self, n, edges):
        self.num_vars, self.e = n, set(edges)

    def has_edge(self, u, v):
        return (u, v) in self.e


def test_pattern_orients_only_unshielded_colliders():
    # 0 -> 2 <- 1, 2 -> 3: the collider is compelled, 2 - 3 stays undirected in the
    # graph the reference compares (no propagation of orientations)
    d, u = graphs.pattern({(0, 2), (1, 2), (2, 3)})
    assert d == {(0, 2), (1, 2)} and u == {frozenset((2, 3))}
    # shielded: 0 -> 2 <- 1 with 0 -> 1: nothing is compelled
    d, u = graphs.pattern({(0, 2), (1, 2), (0, 1)})
    assert d == set() and len(u) == 3
    # an edge given in both directions is undirected and takes no part in colliders
    d, u = graphs.pattern({(0, 2), (2, 0), (1, 2)})
    assert d == set() and u == {frozenset((0, 2)), frozenset((1, 2))}


def test_structural_errors_known_answers():
    truth = [[], [], [0, 1], [2]]                       # 0 -> 2 <- 1, 2 -> 3
    assert graphs.structural_errors(truth, truth)["SHD"] == 0
    # Markov-equivalent member (3 -> 2 is not: it would add a collider); same class: reverse nothing
    assert graphs.structural_errors(truth, [[], [], [0, 1], [2]]) == \
        {"SHD": 0, "ME": 0, "EE": 0, "DE": 0, "ED": 0, "MD": 0, "RD": 0}
    # missing edge 2 -> 3, extra edge 0 -> 3
    e = graphs.structural_errors(truth, [[], [], [0, 1], [0]])
    assert (e["ME"], e["EE"], e["DE"], e["SHD"]) == (1, 1, 0, 2)
    # collider lost: 0 -> 2 -> 1 makes both edges undirected: two missing directions
    e = graphs.structural_errors(truth, [[], [2], [0], [2]])
    assert (e["MD"], e["RD"], e["ED"], e["SHD"]) == (2, 0, 0, 2)
    # reversed collider: truth 0 -> 2 <- 1; predicted 1 -> 0 <- 2 ... 0 - 1 is extra, 2 - 1 missing
    e = graphs.structural_errors([[], [], [0, 1]], [[1, 2], [], []])
    assert e == {"SHD": 3, "ME": 1, "EE": 1, "DE": 1, "ED": 0, "MD": 0, "RD": 1}
    # extra direction: truth chain 0 - 1 - 2 (no collider), predicted 0 -> 1 <- 2
    e = graphs.structural_errors([[], [0], [1]], [[], [0, 2], []])
    assert (e["ED"], e["SHD"]) == (2, 2)
    # PDAG-like objects and parent lists mix; differing node sets are refused
    assert graphs.structural_errors(_G(4, {(0, 2), (1, 2), (2, 3)}), truth)["SHD"] == 0
    with pytest.raises(ValueError):
        graphs.structural_errors(truth, [[], []])


def test_directed_pairs_follows_has_edge():
    g = _G(3, {(0, 1), (1, 0), (2, 1)})
    assert graphs.directed_pairs(g) == {(0, 1), (1, 0), (2, 1)}
    assert graphs.directed_pairs([[1], [0, 2], []]) == {(1, 0), (0, 1), (2, 1)}


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_search_recovers_the_cancer_network(seed):
    """The reference's criterion (SHD == 0 against the generating graph, N = 1e5,
    max_parents = 3) on the restated search."""
    import oracle
    from openbnsl_amd import synth
    codes, card, names, truth = synth.cancer(100_000, seed)
    assert names == ["Cancer", "Dyspnoea", "Pollution", "Smoker", "Xray"]
    parents, _ = oracle.exhaustive_search(codes, card, 3)
    assert graphs.structural_errors(truth, parents)["SHD"] == 0
