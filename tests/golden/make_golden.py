"""Generates tests/golden/*.npz + *.json from the REFERENCE's own C++ hot path.

Run in the build container only (it needs oracle/_ref, i.e. /root/reference):
    make -C oracle && python3 tests/golden/make_golden.py
The outputs are data (inputs and the reference's outputs), committed so that
the parity tests run where the reference is absent.  Scores are stored as
float.hex() strings (exact).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from openbnsl_amd import synth  # noqa: E402

ref = oracle.ref()
assert ref is not None, "oracle/_ref not built"


def frame_labels(codes, card, labels):
    names = [f"c{v:02d}" for v in range(codes.shape[0])]
    return synth.to_dataframe(codes, names, labels=labels), names


def subsets_for(n, rng, sizes=(0, 1, 2, 3, 4, 5), per_size=4):
    out = []
    for k in sizes:
        if k > n:
            continue
        if k == 0:
            out.append([])
            continue
        seen = set()
        for _ in range(per_size * 4):
            s = tuple(sorted(rng.choice(n, size=k, replace=False).tolist()))
            if s not in seen:
                seen.add(s)
                out.append(list(s))
            if len(seen) == per_size:
                break
    return out


def table_record(dfw, vs, ess_list=(1.0, 10.0)):
    ct = ref.ContingencyTable(vs, dfw)
    rec = {"var_ids": vs, "cardinalities": list(ct.cardinalities),
           "counts": {str(k): int(v) for k, v in ct.counts.items()}, "scores": {}}
    for ess in ess_list:
        rec["scores"][repr(ess)] = [
            float(ref.calculate_local_score(ch, [v for v in vs if v != ch], ct,
                                            ref.BDeu(ess))).hex()
            for ch in vs]
    return rec


def main():
    frames = {}
    arrays = {}

    # A: ASIA-shaped binary data, labels yes/no
    codes, card, names = synth.asia(5000, seed=3)
    frames["asia5k"] = (codes, [["no", "yes"]] * 8)
    # B: mixed cardinalities 2..4, rare states so that some cells stay empty
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3], 3000, seed=11,
                                        n_arcs=14, min_prob=0.01)
    frames["mixed3k"] = (codes, [[f"s{j}" for j in range(int(c))] for c in card])
    # C: multi-digit numeric labels: lexicographic coding ("10" < "2")
    rng = np.random.default_rng(5)
    raw = np.stack([rng.integers(0, 12, 500), rng.integers(0, 3, 500),
                    rng.integers(8, 11, 500), rng.integers(0, 2, 500)]).astype(np.int64)
    frames["lexi500"] = (raw, None)

    meta = {}
    for name, (codes, labels) in frames.items():
        n = codes.shape[0]
        names = [f"c{v:02d}" for v in range(n)]
        if labels is None:
            import pandas as pd
            df = pd.DataFrame({nm: codes[v].astype(str).astype(object)
                               for v, nm in enumerate(names)})
            arrays[name + "_raw"] = codes.astype(np.int64)
        else:
            df = synth.to_dataframe(codes, names, labels=labels)
        dfw = ref.DataframeWrapper(df)
        ref_codes = np.array(dfw.data_column_major, dtype=np.uint8)
        arrays[name + "_codes"] = ref_codes
        rng = np.random.default_rng(hash(name) % 1000)
        rec = {
            "columns": names,
            "labels": labels,
            "num_of_values": list(dfw.num_of_values),
            "val_idx2str": [list(v) for v in dfw.val_idx2str],
            "tables": [table_record(dfw, vs) for vs in subsets_for(n, np.random.default_rng(7))],
        }
        meta[name] = rec

    # full score table of the ASIA plumbing configuration (8 vars, N=1e5,
    # BDeu(1.0), subset sizes 1..3): 8 + 56 + 168 = 232 scores
    codes, card, names = synth.asia(100_000, seed=0)
    df = synth.to_dataframe(codes, names, labels=[["no", "yes"]] * 8)
    dfw = ref.DataframeWrapper(df)
    full = {}
    for k in (1, 2, 3):
        subs = oracle.enumerate_subsets(8, k).tolist()
        secs, scores = ref.score_subsets_timed(dfw, subs, 1.0)
        full[str(k)] = [[float(x).hex() for x in row] for row in scores]
    meta["asia100k_full"] = {
        "generator": "openbnsl_amd.synth.asia(100000, seed=0)",
        "sha256_codes": hashlib.sha256(codes.tobytes()).hexdigest(),
        "columns": names,
        "scores_by_size": full,
        # edge list the reference's exhaustive_search returned for this data set
        # with max_parents 2 and 3 (recorded when the reference was surveyed; the
        # search itself cannot be compiled in this image, it needs Boost)
        "exhaustive_search_edges": [["bronc", "dysp"], ["either", "dysp"], ["either", "xray"],
                                    ["lung", "either"], ["tub", "either"], ["smoke", "bronc"],
                                    ["lung", "smoke"], ["tub", "asia"]],
    }

    # simulated_annealing of the reference (src/structure_learning/simulated_annealing.cpp),
    # a caller of the same table + score pair: parents per node of the returned graph
    sa_frames = {
        "asia20k": ("openbnsl_amd.synth.asia(20000, seed=0)", synth.asia(20000, seed=0)[:2]),
        "mixed12_3k": ("openbnsl_amd.synth.random_bn([2,3,4,2,3,4,3,2,4,3,2,2], 3000, seed=11, n_arcs=16)",
                       synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 2], 3000, seed=11, n_arcs=16)[:2]),
    }
    sa_cases = []
    for fname, mp, iters, t0, cool, seed, chains in [
            ("asia20k", 3, 1000, 1.0, 0.9995, 0, 1), ("asia20k", 2, 2000, 1.0, 0.9995, 5, 3),
            ("asia20k", 1, 500, 2.0, 0.99, 7, 2), ("mixed12_3k", 3, 3000, 1.0, 0.9995, 1, 4),
            ("mixed12_3k", 4, 3000, 1.0, 0.9995, 2, 2), ("mixed12_3k", 2, 10000, 1.0, 0.9995, 0, 1)]:
        codes, card = sa_frames[fname][1]
        nms = [f"c{v:02d}" for v in range(codes.shape[0])]
        sdf = synth.to_dataframe(codes, nms, labels=[[f"s{j}" for j in range(int(c))] for c in card])
        sdfw = ref.DataframeWrapper(sdf)
        for det in (True, False):
            _, parents = ref.simulated_annealing(sdfw, 1.0, mp, iters, t0, cool, det, seed, chains)
            sa_cases.append({"frame": fname, "ess": 1.0, "max_parents": mp, "max_iters": iters,
                             "init_temp": t0, "cooling_rate": cool, "is_deterministic": det,
                             "seed": seed, "num_chains": chains,
                             "parents": [list(map(int, p)) for p in parents]})
    meta["simulated_annealing"] = {
        "frames": {k: {"generator": g, "sha256_codes": hashlib.sha256(c[0].tobytes()).hexdigest()}
                   for k, (g, c) in sa_frames.items()},
        "cases": sa_cases,
    }

    # error behaviour of the reference
    errs = {}
    try:
        ref.ContingencyTable([1, 0], dfw)
    except Exception as e:  # noqa: BLE001
        errs["unsorted_var_ids"] = [type(e).__name__, str(e)]
    ct = ref.ContingencyTable([0, 1], dfw)
    try:
        ref.calculate_local_score(5, [0], ct, ref.BDeu(1.0))
    except Exception as e:  # noqa: BLE001
        errs["child_not_in_table"] = [type(e).__name__, str(e)]
    for nm, st in (("BIC", ref.BIC()), ("K2", ref.K2()), ("AIC", ref.AIC())):
        try:
            ref.calculate_local_score(0, [1], ct, st)
        except Exception as e:  # noqa: BLE001
            errs["unsupported_" + nm] = [type(e).__name__, str(e)]
    try:
        ref.simulated_annealing(dfw, 1.0, 8, 10, 1.0, 0.9995, True, 0, 1)
    except Exception as e:  # noqa: BLE001
        errs["sa_max_parents_out_of_range"] = [type(e).__name__, str(e)]
    meta["errors"] = errs

    np.savez_compressed(os.path.join(HERE, "frames.npz"), **arrays)
    with open(os.path.join(HERE, "reference_outputs.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays;", {k: len(v.get("tables", [])) for k, v in meta.items()
                                           if isinstance(v, dict)})


if __name__ == "__main__":
    main()
