#!/usr/bin/env python3
"""The owner function of the map-sharded factor: rank = (bx + A[P] * by + B[P] * bz) mod P over block coordinates — a lattice
colouring.  Neighbouring blocks never share a rank, and any compact set of blocks (the handful of blocks next to the sensor that
hold most of a scan's points) spreads over the ranks as evenly as a static function can; the XOR hash it replaces put
neighbouring heavy blocks on the same rank at random (1.65 x the fair share of queries on the fullest of 8 ranks).

(A, B) per world size P.  LiDAR maps are sheets of blocks — floors and, in a gravity-aligned frame, mostly axis-aligned walls —
so the sections of the lattice {v : v . (1, A, B) = 0 mod P} with the three coordinate planes decide how a wall's blocks spread:
maximise the smallest of the three sections' shortest non-zero vectors, then their sum, then the shortest vector of the 3-D
lattice; ties to the smallest (A, B).  Prints the two tables (index = P; mimosa_amd/csrc/shard_kernels.hip carries them as C
initialisers, tests/dist_harness.py and tools/shard_table.py import this module's TABLE_A / TABLE_B)."""
import itertools
import sys

R = 9  # |coordinate| bound of the vectors searched: the shortest vector of a determinant-P lattice, P <= 64, is well inside


def shortest(coeffs, P):
    best = None
    for v in itertools.product(range(-R, R + 1), repeat=len(coeffs)):
        if any(v) and sum(c * x for c, x in zip(coeffs, v)) % P == 0:
            d = sum(x * x for x in v)
            best = d if best is None or d < best else best
    return best


def pick(P):
    if P == 1:
        return 0, 0
    one = {c: shortest((1, c), P) for c in range(P)}
    keyed = {}
    for a in range(P):
        for b in range(P):
            sec = (one[a], one[b], shortest((a, b), P) or 0)
            keyed[(a, b)] = (min(sec), sum(sec))
    top = max(keyed.values())
    best = None
    for (a, b), k in sorted(keyed.items()):
        if k != top:
            continue
        s3 = shortest((1, a, b), P)
        if best is None or s3 > best[0]:
            best = (s3, a, b)
    return best[1], best[2]


def tables(n=64):
    ab = [pick(P) for P in range(1, n + 1)]
    return [0] + [a for a, _ in ab], [0] + [b for _, b in ab]


# python tools/lattice_table.py regenerates these two lines
TABLE_A = [0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 3, 3, 5, 7, 5, 6, 5, 4, 7, 5, 6, 6, 5, 5, 5, 5, 5, 6, 6, 6, 5, 5, 7, 6, 7, 5, 7, 6, 7, 8, 7, 6, 7, 7, 9, 8, 8, 7, 7, 10, 10, 7, 7, 8, 7, 8, 8, 8, 8]
TABLE_B = [0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 5, 6, 4, 5, 7, 8, 8, 8, 8, 10, 9, 7, 11, 8, 12, 11, 12, 6, 7, 14, 13, 10, 6, 7, 11, 16, 15, 6, 12, 7, 13, 17, 10, 14, 18, 11, 11, 14, 23, 22, 16, 12, 21, 8, 9, 13, 11, 11, 17, 14, 19]


def owner_of_block(b, world):
    """numpy: rank of block coordinates b (..., 3), int64."""
    import numpy as np
    b = np.asarray(b, np.int64)
    return ((b[..., 0] + TABLE_A[world] * b[..., 1] + TABLE_B[world] * b[..., 2]) % world).astype(np.int16)


if __name__ == "__main__":
    A, B = tables()
    print("TABLE_A =", A)
    print("TABLE_B =", B)
