"""Synthetic inputs: FCC crystal + Maxwell momenta (host logic).

Mirrors what the reference's generator produces for ``coordname _fcc``
(generate_fcc, imd_generate.c:370-560): sites ((x+.5)a/2,(y+.5)a/2,(z+.5)a/2)
with x+y+z even, generated x-outer / z-inner, numbered from 0 in that order;
box = box_param*a (init_cubic, imd_generate.c:289-368).  Momenta are drawn from
a Maxwell distribution at `temp` (eV) with the centre-of-mass momentum removed
(maxwell, imd_maxwell.c:45-323); the random stream is numpy's, not drand48.
"""
import numpy as np


def fcc_positions(box_param, a=4.05, x_range=None):
    """Positions (n,3) of FCC sites; optional x_range=(lo,hi) in half-cell units
    restricts generation to a slab of the x loop (one rank's share)."""
    bx, by, bz = (2 * int(v) for v in box_param)
    bu = a / 2.0
    x0, x1 = (0, bx) if x_range is None else x_range
    x = np.arange(x0, x1, dtype=np.int64)[:, None, None]
    y = np.arange(by, dtype=np.int64)[None, :, None]
    z = np.arange(bz, dtype=np.int64)[None, None, :]
    mask = ((x + y + z) % 2) == 0
    X, Y, Z = np.broadcast_arrays(x, y, z)
    pos = np.stack([(X[mask] + 0.5) * bu, (Y[mask] + 0.5) * bu, (Z[mask] + 0.5) * bu], axis=1)
    return pos


def fcc_box(box_param, a=4.05):
    return np.diag([box_param[0] * a, box_param[1] * a, box_param[2] * a]).astype(np.float64)


def maxwell_momenta(n, mass, temp, seed=12345):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n, 3)) * np.sqrt(mass * temp)
    p -= p.mean(axis=0, keepdims=True)
    return p
