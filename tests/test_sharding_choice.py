"""CPU: how ranks decide, each on its own, between splitting the subsets and splitting
the rows, and where the row slices lie."""
import bench
from openbnsl_amd import pipeline


def test_sharding_choice_is_a_function_of_the_problem():
    """What bench.py's ranks decide on their own (no collective): rows for the short passes,
    subsets where the count tables are large or a size is counted per row."""
    want = {"alarm37_1e6_bdeu_mp3": "rows", "alarm37_1e6_bic_mp4": "subsets", "hepar70_1e7_bdeu_mp3": "rows",
            "uniform128x4_1e7_bdeu_mp3": "subsets", "asia8_1e5_bdeu_mp2": "subsets"}
    for name, mode in want.items():
        spec = bench.WORKLOADS[name]
        codes, card, _ = bench.make_data(dict(spec, N=50_000))    # (the cardinalities only)
        assert pipeline.choose_sharding(card, spec["N"], spec["max_parents"], 8) == mode, name
        assert pipeline.choose_sharding(card, spec["N"], spec["max_parents"], 1) == "subsets"
    edges = [pipeline.row_slice(1_000_000, r, 8) for r in range(8)]
    assert edges[0][0] == 0 and edges[-1][1] == 1_000_000
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:])) and all(lo % 2048 == 0 for lo, _ in edges)
