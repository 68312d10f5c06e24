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
