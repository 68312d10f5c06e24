"""Graph helpers for checking learned structures without pgmpy.

The reference's tests compare structures through pgmpy and
helpers/structural_distance.py (SHD between CPDAGs, reference:
pytest/structure_learning/test_exhaustive_search.py:25-30;
helpers/pgmpy_bridge.py:25-28 for the edge extraction).  Two DAGs are Markov
equivalent -- same CPDAG, SHD 0 -- exactly when they have the same skeleton and
the same v-structures (Verma & Pearl), which is what these helpers test.
"""


def parent_lists(pdag, n=None):
    """PDAG -> list of sorted parent lists (edge j -> v iff pdag.has_edge(j, v))."""
    n = pdag.num_vars if n is None else n
    return [[j for j in range(n) if pdag.has_edge(j, v)] for v in range(n)]


def edge_list(parents, names=None):
    out = []
    for v, ps in enumerate(parents):
        for p in ps:
            out.append((names[p], names[v]) if names else (p, v))
    return sorted(out)


def skeleton(parents):
    return {frozenset((p, v)) for v, ps in enumerate(parents) for p in ps}


def v_structures(parents):
    """{(a, c, b)}: a -> c <- b with a < b not adjacent."""
    skel = skeleton(parents)
    out = set()
    for c, ps in enumerate(parents):
        ps = sorted(ps)
        for i, a in enumerate(ps):
            for b in ps[i + 1:]:
                if frozenset((a, b)) not in skel:
                    out.add((a, c, b))
    return out


def markov_equivalent(parents_a, parents_b):
    return (skeleton(parents_a) == skeleton(parents_b)
            and v_structures(parents_a) == v_structures(parents_b))


def directed_pairs(graph, n=None):
    """{(u, v)}: every ordered pair with graph.has_edge(u, v), the extraction the
    reference performs in helpers/pgmpy_bridge.py:25-28.  `graph` is a PDAG-like
    object (num_vars, has_edge) or a list of parent lists; an undirected edge is a
    pair present in both orders."""
    if hasattr(graph, "has_edge"):
        n = graph.num_vars if n is None else n
        return {(u, v) for u in range(n) for v in range(n) if u != v and graph.has_edge(u, v)}
    return {(p, v) for v, ps in enumerate(graph) for p in ps if p != v}


def pattern(pairs):
    """(directed, undirected) edge sets of the graph the reference compares
    (helpers/structural_distance.py:43-76): every edge of the skeleton starts
    undirected; for every node with two parents (by edges present in one
    direction only) that are not adjacent, the two edges into it become directed.
    No orientation is propagated beyond the unshielded colliders."""
    pairs = set(pairs)
    one_way = {(u, v) for (u, v) in pairs if (v, u) not in pairs}
    adjacent = {frozenset(e) for e in pairs}
    both = {(u, v) for e in adjacent for (u, v) in (tuple(e), tuple(e)[::-1])}
    parents = {}
    for u, v in one_way:
        parents.setdefault(v, []).append(u)
    for x, ps in parents.items():
        ps = sorted(ps)
        for i, y in enumerate(ps):
            for z in ps[i + 1:]:
                if frozenset((y, z)) not in adjacent:
                    both.discard((x, y))
                    both.discard((x, z))
    directed = {(u, v) for (u, v) in both if (v, u) not in both}
    undirected = {frozenset((u, v)) for (u, v) in both if (v, u) in both}
    return directed, undirected


def structural_errors(truth, predicted, n=None):
    """The reference's structural distance (helpers/structural_distance.py:79-129)
    between two graphs given as PDAG-like objects or parent lists:
      SHD = ME + EE + DE, ME / EE = edges missing from / extra in the predicted
      skeleton, DE = ED + MD + RD: an edge undirected in the truth but directed
      in the prediction, directed in the truth but undirected in the prediction,
      directed both ways round."""
    if hasattr(truth, "num_vars") and hasattr(predicted, "num_vars") and truth.num_vars != predicted.num_vars:
        raise ValueError("Node sets differ.")
    if not hasattr(truth, "num_vars") and not hasattr(predicted, "num_vars") and len(truth) != len(predicted):
        raise ValueError("Node sets differ.")
    dir_t, und_t = pattern(directed_pairs(truth, n))
    dir_p, und_p = pattern(directed_pairs(predicted, n))
    skel_t = und_t | {frozenset(e) for e in dir_t}
    skel_p = und_p | {frozenset(e) for e in dir_p}
    me, ee = len(skel_t - skel_p), len(skel_p - skel_t)
    ed = sum(1 for e in und_t for (u, v) in [tuple(e)] if ((u, v) in dir_p) ^ ((v, u) in dir_p))
    md = sum(1 for (u, v) in dir_t if frozenset((u, v)) in und_p)
    rd = sum(1 for (u, v) in dir_t if (v, u) in dir_p)
    de = ed + md + rd
    return {"SHD": me + ee + de, "ME": me, "EE": ee, "DE": de, "ED": ed, "MD": md, "RD": rd}


def is_acyclic(parents):
    n = len(parents)
    state = [0] * n

    def visit(v):
        if state[v] == 1:
            return False
        if state[v] == 2:
            return True
        state[v] = 1
        ok = all(visit(p) for p in parents[v])
        state[v] = 2
        return ok

    return all(visit(v) for v in range(n))
