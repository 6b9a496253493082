"""The static part of the multi-rank plan built in C (imdgpu_dist_plan_geometry, imd_amd/csrc/imdgpu_dist.hip: what the
native rebuild imdgpu_dist_rebuild uses) against the host-side statement of the same plan (imd_amd.parallel.direct_segments),
for the rank grids of bench.py --gpus 2/4/8, a z split and slabs with open x.  Pure host code: runs without a GPU.
Reference: rank grid and neighbours of setup_mpi_topology (imd_geom_mpi_3d.c:32-90), shift added by the sender
(imd_comm_force_3d.c:247-266)."""
import ctypes as C

import numpy as np
import pytest

from imd_amd import lib
from imd_amd.parallel import Decomposition, direct_segments


@pytest.mark.parametrize("cpu_dim,pbc,gdim", [
    ((2, 1, 1), (1, 1, 1), (8, 4, 6)), ((1, 1, 2), (1, 1, 1), (4, 6, 8)), ((2, 2, 1), (1, 1, 1), (8, 8, 4)),
    ((2, 2, 2), (1, 1, 1), (8, 8, 8)), ((2, 2, 2), (0, 1, 1), (12, 8, 8)), ((4, 2, 1), (0, 1, 0), (16, 8, 5)),
    ((1, 1, 1), (1, 1, 1), (4, 4, 4)), ((3, 1, 2), (1, 0, 1), (9, 5, 6))])
def test_c_plan_geometry_equals_host_plan(cpu_dim, pbc, gdim):
    L = lib.load()
    box = np.diag([40.5, 32.4, 28.35]) + np.array([[0, 0, 0], [0.0, 0, 0], [0, 0, 0]])
    world = int(np.prod(cpu_dim))
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    for rank in range(world):
        dec = Decomposition(rank, cpu_dim, gdim, box, pbc)
        _, segs = direct_segments(dec)
        seg = np.zeros(26 * 6, np.int32); sh = np.zeros(26 * 3)
        ns, nr = C.c_int(), C.c_int()

        def arr(v):
            return np.ascontiguousarray(v, np.int32)
        cd, co, dm, pb = arr(cpu_dim), arr(dec.coord), arr(dec.dim), arr(pbc)
        bx = np.ascontiguousarray(box, np.float64)
        assert L.imdgpu_dist_plan_geometry(cd.ctypes.data_as(ip), co.ctypes.data_as(ip), dm.ctypes.data_as(ip), pb.ctypes.data_as(ip),
                                           bx.ctypes.data_as(dp), seg.ctypes.data_as(ip), sh.ctypes.data_as(dp), None, C.byref(ns), None,
                                           C.byref(nr)) == 0
        sc = np.zeros(max(ns.value, 1), np.int32); rc = np.zeros(max(nr.value, 1), np.int32)
        assert L.imdgpu_dist_plan_geometry(cd.ctypes.data_as(ip), co.ctypes.data_as(ip), dm.ctypes.data_as(ip), pb.ctypes.data_as(ip),
                                           bx.ctypes.data_as(dp), seg.ctypes.data_as(ip), sh.ctypes.data_as(dp), sc.ctypes.data_as(ip),
                                           C.byref(ns), rc.ctypes.data_as(ip), C.byref(nr)) == 0
        seg = seg.reshape(26, 6); sh = sh.reshape(26, 3)
        want_s = np.concatenate([g["send_cells"] for g in segs if g["to"] >= 0] + [np.zeros(0, np.int64)])
        want_r = np.concatenate([g["recv_cells"] for g in segs if g["frm"] >= 0] + [np.zeros(0, np.int64)])
        for q, g in enumerate(segs):
            assert tuple(seg[q, :3]) == tuple(g["d"])
            assert seg[q, 3] == g["to"] and seg[q, 4] == g["frm"], (rank, q, seg[q], g["to"], g["frm"])
            assert seg[q, 5] == len(g["send_cells"]) == len(g["recv_cells"])
            assert np.array_equal(sh[q], g["shift"])
        assert ns.value == len(want_s) and nr.value == len(want_r)
        assert np.array_equal(sc[:ns.value], want_s) and np.array_equal(rc[:nr.value], want_r)
