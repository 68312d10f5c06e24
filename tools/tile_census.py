"""Tile census of the matrix-core launch of one subset size (host arithmetic only, no GPU):
rows / columns of the contraction, tiles by number of valid 32-column blocks, executed
against counted cells.  Mirrors build_plan's row order and tile limits."""
import sys
from itertools import combinations
from math import comb

import numpy as np

sys.path.insert(0, ".")
from openbnsl_amd import synth  # noqa: E402


def census(card, k, tb=128, tbc=None):
    tbc = tbc or tb
    n = len(card)
    kc, kr = k // 2, k - k // 2
    red = [c - 1 for c in card]
    # column groups in colex order
    cg = sorted(combinations(range(n), kc), key=lambda s: sum(comb(v, j + 1) for j, v in enumerate(s)))
    cells = [int(np.prod([red[v] for v in s])) if s else 1 for s in cg]
    before = np.concatenate([[0], np.cumsum(cells)])
    def ncols_below(u):
        g = 1 if kc == 0 else comb(u, kc)
        return int(before[min(g, len(cg))]) if g else 0
    rs = [s for s in sorted(combinations(range(n), kr), key=lambda s: sum(comb(v, j + 1) for j, v in enumerate(s))) if s[0] >= kc]
    rs.sort(key=lambda s: s[0])
    rows = []   # ncols per row
    need = 0
    for s in rs:
        M = int(np.prod([red[v] for v in s]))
        rows += [ncols_below(s[0])] * M
        need += M * ncols_below(s[0])
    nrows = len(rows)
    tiles = {}
    executed = 0
    for i in range(0, nrows, tb):
        lim = rows[min(i + tb, nrows) - 1]
        for j in range(0, lim, tbc):
            blk = (min(lim - j, tbc) + 31) // 32
            tiles[blk] = tiles.get(blk, 0) + 1
            executed += tb * 32 * blk
    return dict(rows=nrows, cols=int(before[-1]), need=need, executed=executed, ratio=executed / max(need, 1), tiles=tiles,
                ntiles=sum(tiles.values()))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "alarm"
    if which == "alarm":
        rng = np.random.default_rng(1000)
        cards = list(synth.ALARM_CARDS)
        rng.shuffle(cards)
    elif which == "hepar2":
        rng = np.random.default_rng(2000)
        cards = [2] * 48 + [3] * 13 + [4] * 9
        rng.shuffle(cards)
    else:
        cards = [4] * int(which)
    print(cards)
    for k in range(1, 6):
        for tb, tbc in ((128, 128), (128, 256), (256, 128), (256, 256), (64, 128)):
            print(k, (tb, tbc), census(cards, k, tb, tbc))


def block_floor(card, k, rb=32, cb=32):
    """Cells executed when only the rb x cb blocks that hold a valid cell are computed."""
    n = len(card)
    kc, kr = k // 2, k - k // 2
    red = [c - 1 for c in card]
    cg = sorted(combinations(range(n), kc), key=lambda s: sum(comb(v, j + 1) for j, v in enumerate(s)))
    cells = [int(np.prod([red[v] for v in s])) if s else 1 for s in cg]
    before = np.concatenate([[0], np.cumsum(cells)])
    rs = [s for s in combinations(range(n), kr) if s[0] >= kc]
    rs.sort(key=lambda s: (s[0], sum(comb(v, j + 1) for j, v in enumerate(s))))
    rows = []
    for s in rs:
        M = int(np.prod([red[v] for v in s]))
        g = 1 if kc == 0 else comb(s[0], kc)
        rows += [int(before[min(g, len(cg))])] * M
    ex = 0
    for i in range(0, len(rows), rb):
        lim = rows[min(i + rb, len(rows)) - 1]
        ex += rb * ((lim + cb - 1) // cb) * cb
    return ex
