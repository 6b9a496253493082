"""Helpers to turn tests/golden/*.npz (dumps of the real reference's memory) into
oracle systems / product inputs.  Test infrastructure."""
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(case):
    return np.load(os.path.join(GOLDEN, case + ".npz"), allow_pickle=False)


def table(G, key):
    if key + "_ncols" not in G:
        return None
    return {f: G["%s_%s" % (key, f)] for f in ("ncols", "maxsteps", "begin", "end", "step", "invstep", "len", "table")}


def inner_counts(G, step):
    return G["s%d_cell_n" % step].astype(np.int64)[G["cells"]]


def oracle_from_golden(G, step, nbl=True):
    """Oracle system holding exactly the reference's cell arrangement of `step`
    (inner cells only; ghost cells are refilled by the oracle's own send_cells)."""
    pre = "s%d_" % step
    cell_n = G[pre + "cell_n"].astype(np.int64)
    cells = G["cells"]
    dim = G["cell_dim"]
    if not nbl:                      # cell-pair build has no ghost layer: embed it
        dim = dim + 2
    cap = int(cell_n.max() + 8)
    s = po.OracleSystem(dim, cap, int(G["ntypes"]), G["box"], G["pbc_dirs"], float(G["cellsz"]),
                        float(G["nbl_margin"]), float(G["timestep"]))
    if not nbl:
        cells = s.inner.copy()       # same x-outer/z-inner order
        cn = cell_n
    else:
        cn = cell_n[cells]
    k = 0
    for c, m in zip(cells, cn):
        m = int(m)
        s.n[c] = m
        s.ort[c, :m] = G[pre + "ort"][k:k + m]
        s.impuls[c, :m] = G[pre + "impuls"][k:k + m]
        s.masse[c, :m] = G[pre + "masse"][k:k + m]
        s.sorte[c, :m] = G[pre + "sorte"][k:k + m]
        s.nummer[c, :m] = G[pre + "nummer"][k:k + m]
        if pre + "nbl_pos" in G:
            s.nbl_pos[c, :m] = G[pre + "nbl_pos"][k:k + m]
        k += m
    for which, key in ((0, "pair_pot"), (1, "rho_h_tab"), (2, "embed_pot")):
        t = table(G, key)
        if t is not None:
            s.set_table(which, t)
    return s


def oracle_ttm_from_golden(G, s):
    """TTM oracle bound to oracle system `s`, EOS tables from the fixture."""
    t = po.OracleTTM(s, G["ttm_fd_ext"], int(G["ttm_fd_min_atoms"]), float(G["ttm_atomic_weight"]),
                     float(G["ttm_atomic_charge"]), float(G["pair_pot_end"][0]))
    nr, nt = int(G["eos_nr"]), int(G["eos_nt"])
    t.set_eos(0, nr, nt, G["eos_r"][0], G["eos_r"][-1], G["eos_t"][0], G["eos_t"][-1], G["eos_cve"])
    t.set_eos(1, nr, nt, G["eos_r"][0], G["eos_r"][-1], G["eos_ts"][0], G["eos_ts"][-1], G["eos_ee"])
    t.t.omega_laser = float(G["ttm_omega_laser"])
    return t


def ttm_stage(G, step, tag):
    pre = "t%d_%s_" % (step, tag)
    return {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}
