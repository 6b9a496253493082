"""Spatial decomposition over several GPUs: one rank per GPU, ghost-cell halo
exchange and atom migration over torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

This is the host-side restatement of the reference's MPI layer:
  - rank grid and neighbours: setup_mpi_topology (imd_geom_mpi_3d.c:32-90),
  - forward halo with the periodic shift added by the sending boundary rank: send_cells
    (imd_comm_force_3d.c:222-396).  The reference fills the ghost shell in Plimpton's three stages
    (z, y, x; 6 messages, edges travel twice, corners three times); here every one of the 26 ghost
    regions comes DIRECTLY from the rank that owns it (DirectHaloPlan): the 8 GPUs of a node are
    fully connected by xGMI, so one group of point-to-point messages replaces three dependent stages,
  - atom migration when the neighbour list is rebuilt: fix_cells -> send_atoms
    (imd_fix_cells_3d.c:43-306,436-542),
  - list validity as a global MAX: check_nblist (imd_forces_nbl.c:1087).
Because the device kernels gather over full neighbour lists, only positions and
dF travel (forward); the reference's two reverse exchanges (rho, forces) do not
exist here.  Every ghost direction is filled (the non-AR shell).

Division of labour: the REBUILD (migration, ghost layout, index lists; every ~17 steps) is driven from
here over torch.distributed; the STEP itself is native -- imdgpu_dist_md_step (imd_amd/csrc/imdgpu_dist.hip)
posts the halos as RCCL groups from C, keeps the validity flag on the device and returns only the
"rebuild needed" bit.

`ops` is the per-rank device: GpuOps (libimdgpu through the C ABI) in production;
the CPU tests plug in a numpy stand-in with the same methods, which is how the
N>1 logic is exercised without a GPU.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def rank_to_coord(rank, cpu_dim):
    pz = cpu_dim[2]
    py = cpu_dim[1]
    return (rank // (py * pz), (rank // pz) % py, rank % pz)


def coord_to_rank(c, cpu_dim):
    return (c[0] * cpu_dim[1] + c[1]) * cpu_dim[2] + c[2]


def choose_cpu_dim(world):
    """Most cubic factorisation: 8 -> 2x2x2 so that the 6 faces use 6 different xGMI links
    (calc_cpu_dim, imd_geom_mpi_3d.c:201, picks the same for powers of two)."""
    best = (world, 1, 1)
    for a in range(1, world + 1):
        if world % a:
            continue
        for b in range(1, world // a + 1):
            if (world // a) % b:
                continue
            c = world // a // b
            cand = tuple(sorted((a, b, c), reverse=True))
            if max(cand) - min(cand) < max(best) - min(best):
                best = cand
    return best


class Decomposition:
    """Geometry of one rank: local cell array (with ghost layer), neighbours, stage cell lists."""

    def __init__(self, rank, cpu_dim, global_cell_dim, box, pbc):
        self.rank, self.cpu_dim = rank, tuple(int(v) for v in cpu_dim)
        self.coord = rank_to_coord(rank, self.cpu_dim)
        self.gdim = tuple(int(v) for v in global_cell_dim)
        for d in range(3):
            if self.gdim[d] % self.cpu_dim[d]:
                raise ValueError("global_cell_dim must be a multiple of cpu_dim")
        self.dim = tuple(self.gdim[d] // self.cpu_dim[d] + 2 for d in range(3))       # cell_dim incl. ghosts
        self.box = np.asarray(box, dtype=np.float64).reshape(3, 3)
        self.pbc = tuple(int(v) for v in pbc)
        self.nall = int(np.prod(self.dim))
        self.lo, self.hi, self.shift_lo, self.shift_hi = [], [], [], []
        for d in range(3):
            P, c = self.cpu_dim[d], self.coord[d]
            lo = hi = None
            if c > 0 or self.pbc[d]:
                n = list(self.coord); n[d] = (c - 1) % P; lo = coord_to_rank(n, self.cpu_dim)
            if c < P - 1 or self.pbc[d]:
                n = list(self.coord); n[d] = (c + 1) % P; hi = coord_to_rank(n, self.cpu_dim)
            self.lo.append(lo); self.hi.append(hi)
            # periodic image shift added by the SENDING boundary rank (imd_comm_force_3d.c:247-266)
            self.shift_lo.append(self.box[d] * (1.0 if (c == 0 and self.pbc[d]) else 0.0))
            self.shift_hi.append(self.box[d] * (-1.0 if (c == P - 1 and self.pbc[d]) else 0.0))

    def lin(self, i, j, k):
        return (i * self.dim[1] + j) * self.dim[2] + k

    def stage_cells(self, d):
        """Cell lists of stage d: (send_lo, send_hi, recv_lo, recv_hi) as linear indices.
        Stage z covers inner x,y; stage y inner x and ALL z; stage x ALL y,z (edges travel twice,
        corners three times: imd_comm_force_3d.c:213-217)."""
        D = self.dim
        rng = {0: None, 1: None, 2: None}
        if d == 2:
            rng[0], rng[1] = range(1, D[0] - 1), range(1, D[1] - 1)
        elif d == 1:
            rng[0], rng[2] = range(1, D[0] - 1), range(0, D[2])
        else:
            rng[1], rng[2] = range(0, D[1]), range(0, D[2])
        others = [q for q in range(3) if q != d]
        a, b = np.meshgrid(np.array(rng[others[0]]), np.array(rng[others[1]]), indexing="ij")
        a, b = a.ravel(), b.ravel()

        def layer(v):
            idx = [None, None, None]
            idx[d] = np.full_like(a, v)
            idx[others[0]], idx[others[1]] = a, b
            return ((idx[0] * D[1] + idx[1]) * D[2] + idx[2]).astype(np.int64)
        return layer(1), layer(D[d] - 2), layer(0), layer(D[d] - 1)


def _staged():
    """gloo moves host memory only: CUDA buffers are staged through the host (used when several ranks
    share one GPU in tests); with nccl (= RCCL) device buffers travel directly over xGMI."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def all_gather_flat(t, world):
    if world == 1:
        return t
    if _staged() and t.is_cuda:
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(out, h)
        return torch.cat(out).to(t.device)
    out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


def all_reduce_(t, op=None):
    op = op or dist.ReduceOp.SUM
    if _staged() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def _exchange(dec, d, buf_lo, buf_hi, like_lo, like_hi, group=None, out_lo=None, out_hi=None):
    """Send buf_lo to the low neighbour and buf_hi to the high one; returns (from_lo, from_hi) where
    from_lo is the low neighbour's HIGH layer and from_hi the high neighbour's LOW layer.
    like_*: (numel, dtype) of what arrives; out_*: optional tensors to receive into (views of one buffer).
    Self-neighbours (cpu_dim[d]==1) short-circuit."""
    lo, hi = dec.lo[d], dec.hi[d]
    if dec.cpu_dim[d] == 1:
        return (buf_hi if lo is not None else None), (buf_lo if hi is not None else None)
    ops, from_lo, from_hi = [], None, None
    dev = buf_lo.device if buf_lo is not None else buf_hi.device
    stage = _staged() and dev.type == "cuda"
    tdev = torch.device("cpu") if stage else dev
    if stage:
        buf_lo = buf_lo.cpu() if buf_lo is not None else None
        buf_hi = buf_hi.cpu() if buf_hi is not None else None
    # empty messages are not posted at all: both sides know the sizes (counts travel first), and a zero-byte
    # point-to-point operation is the one thing collectives libraries disagree about
    if lo is not None and buf_lo.numel() > 0:
        ops.append(dist.P2POp(dist.isend, buf_lo, lo, group))
    if hi is not None and buf_hi.numel() > 0:
        ops.append(dist.P2POp(dist.isend, buf_hi, hi, group))
    if hi is not None:
        from_hi = out_hi if (out_hi is not None and not stage) else torch.empty(like_hi[0], dtype=like_hi[1], device=tdev)
        if like_hi[0] > 0:
            ops.append(dist.P2POp(dist.irecv, from_hi, hi, group))
    if lo is not None:
        from_lo = out_lo if (out_lo is not None and not stage) else torch.empty(like_lo[0], dtype=like_lo[1], device=tdev)
        if like_lo[0] > 0:
            ops.append(dist.P2POp(dist.irecv, from_lo, lo, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if stage:
        if from_lo is not None:
            from_lo = out_lo.copy_(from_lo) if out_lo is not None else from_lo.to(dev)
        if from_hi is not None:
            from_hi = out_hi.copy_(from_hi) if out_hi is not None else from_hi.to(dev)
    return from_lo, from_hi


class _Neighbours:
    """view of a Decomposition with other neighbours along one direction (do_COMMFLUX: no periodic wrap)"""

    def __init__(self, dec, d, lo, hi):
        self.cpu_dim = dec.cpu_dim
        self.lo = list(dec.lo); self.hi = list(dec.hi)
        self.lo[d], self.hi[d] = lo, hi


WHAT_WIDTH = {0: (4, torch.float64), 1: (1, torch.float64), 2: (1, torch.int32), 3: (3, torch.float64)}


class HaloPlan:
    """Index lists of one rebuild: per stage what to gather for each neighbour and where arrivals go."""

    def __init__(self, dec, ops):
        self.dec, self.stages = dec, []
        cs, cc = ops.cell_tables()
        cs, cc = cs.astype(np.int64), cc.astype(np.int64)
        nlocal = ops.nlocal()
        if getattr(dec, "_static", None) is None:     # geometry only: ghost-cell mask and the stage cell lists
            ghost_mask = np.ones(dec.nall, bool)
            D = dec.dim
            ii, jj, kk = np.meshgrid(range(1, D[0] - 1), range(1, D[1] - 1), range(1, D[2] - 1), indexing="ij")
            ghost_mask[((ii * D[1] + jj) * D[2] + kk).ravel()] = False
            dec._static = (np.flatnonzero(ghost_mask), {d: dec.stage_cells(d) for d in (0, 1, 2)})
        ghost_cells, stage_cells = dec._static
        cc[ghost_cells] = 0
        nghost = 0
        dev = ops.device
        for d in (2, 1, 0):
            s_lo, s_hi, r_lo, r_hi = stage_cells[d]
            c_lo, c_hi = torch.from_numpy(cc[s_lo].astype(np.int32)).to(dev), torch.from_numpy(cc[s_hi].astype(np.int32)).to(dev)
            f_lo, f_hi = _exchange(dec, d, c_lo, c_hi, (len(r_lo), torch.int32), (len(r_hi), torch.int32))
            st = dict(d=d)
            for name, cells, cnts in (("lo", r_lo, f_lo), ("hi", r_hi, f_hi)):
                if cnts is None:
                    st["recv_" + name] = None
                    continue
                cn = cnts.cpu().numpy().astype(np.int64)
                start = nlocal + nghost + np.concatenate([[0], np.cumsum(cn)[:-1]])
                cc[cells], cs[cells] = cn, start
                tot = int(cn.sum())
                st["recv_" + name] = (nlocal + nghost, tot)
                nghost += tot
            for name, cells, nb in (("lo", s_lo, dec.lo[d]), ("hi", s_hi, dec.hi[d])):
                if nb is None:
                    st["send_" + name] = None
                    continue
                n = cc[cells]
                # atoms of the listed cells, cell after cell (the same order the receiver laid its ghosts out in)
                idx = np.repeat(cs[cells] - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum()))
                st["send_" + name] = torch.from_numpy(idx.astype(np.int32)).to(dev)
            self.stages.append(st)
        self.nghost = nghost
        ops.set_ghost_layout(cs.astype(np.int32), cc.astype(np.int32), nghost)
        for st in self.stages:       # arrivals land in one contiguous range per neighbour
            for name in ("lo", "hi"):
                if st["recv_" + name] is not None:
                    a, n = st["recv_" + name]
                    st["ridx_" + name] = torch.arange(a, a + n, dtype=torch.int32, device=dev)
            # one gather and one scatter launch per stage: send list = [low face; high face]; arrivals are laid out
            # as [from the high neighbour; from the low neighbour], which for a self-neighbour IS the send buffer
            e = torch.empty(0, dtype=torch.int32, device=dev)
            s_lo = st["send_lo"] if st["send_lo"] is not None else e
            s_hi = st["send_hi"] if st["send_hi"] is not None else e
            st["send_cat"], st["n_send"] = torch.cat([s_lo, s_hi]), (int(s_lo.numel()), int(s_hi.numel()))
            r_hi = st["ridx_hi"] if st["recv_hi"] is not None else e
            r_lo = st["ridx_lo"] if st["recv_lo"] is not None else e
            st["ridx_cat"], st["n_recv"] = torch.cat([r_hi, r_lo]), (int(r_hi.numel()), int(r_lo.numel()))

    def exchange(self, ops, what):
        """Forward halo of `what` (0 positions+dF, 1 dF, 2 types) in the stages z, y, x."""
        w, dt = WHAT_WIDTH[what]
        dec = self.dec
        for st in self.stages:
            d = st["d"]
            n_lo, n_hi = st["n_send"]
            m_hi, m_lo = st["n_recv"]
            if n_lo + n_hi == 0 and m_hi + m_lo == 0:
                continue
            buf = ops.gather2(what, st["send_cat"], n_lo, dec.shift_lo[d], dec.shift_hi[d])
            b_lo = buf[:n_lo * w] if st["send_lo"] is not None else None
            b_hi = buf[n_lo * w:] if st["send_hi"] is not None else None
            if dec.cpu_dim[d] == 1:
                if dec.lo[d] is not None and dec.hi[d] is not None:       # periodic self-neighbour: [from_hi; from_lo] = [b_lo; b_hi]
                    ops.scatter(what, st["ridx_cat"], buf)
                continue
            recv = torch.empty((m_hi + m_lo) * w, dtype=dt, device=buf.device)
            _exchange(dec, d, b_lo, b_hi, (m_lo * w, dt), (m_hi * w, dt), out_lo=recv[m_hi * w:], out_hi=recv[:m_hi * w])
            ops.scatter(what, st["ridx_cat"], recv)


DIRECTIONS = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]


def direct_segments(dec):
    """The static part of the direct halo: per direction (in the fixed order of DIRECTIONS) the peers, the periodic shift
    the sender adds and the cell layers.  The library builds the same in C (imdgpu_dist_plan_geometry, imdgpu_dist.hip);
    tests/test_dist_plan_geometry.py holds the two against each other."""
    D = dec.dim
    ghost_mask = np.ones(dec.nall, bool)
    ii, jj, kk = np.meshgrid(range(1, D[0] - 1), range(1, D[1] - 1), range(1, D[2] - 1), indexing="ij")
    ghost_mask[((ii * D[1] + jj) * D[2] + kk).ravel()] = False
    segs = []
    for d in DIRECTIONS:
        peer_to, peer_from, shift, ok_to, ok_from = [], [], np.zeros(3), True, True
        for k in range(3):
            P, c = dec.cpu_dim[k], dec.coord[k]
            t, f = c + d[k], c - d[k]
            if t < 0 or t >= P:
                if not dec.pbc[k]:
                    ok_to = False
                elif d[k]:
                    shift = shift + dec.box[k] * (1.0 if t < 0 else -1.0)      # added by the SENDER (:247-266)
            if f < 0 or f >= P:
                if not dec.pbc[k]:
                    ok_from = False
            peer_to.append(t % P); peer_from.append(f % P)

        def layer(sel):
            r = [range(1, D[k] - 1) if sel[k] is None else [sel[k]] for k in range(3)]
            a, b, c = np.meshgrid(np.array(list(r[0])), np.array(list(r[1])), np.array(list(r[2])), indexing="ij")
            return ((a * D[1] + b) * D[2] + c).ravel().astype(np.int64)
        send_cells = layer([None if d[k] == 0 else (1 if d[k] < 0 else D[k] - 2) for k in range(3)])
        # arrivals of direction d come from coord - d and fill my ghost region on the side they come from
        recv_cells = layer([None if d[k] == 0 else (D[k] - 1 if d[k] < 0 else 0) for k in range(3)])
        segs.append(dict(d=d, to=coord_to_rank(peer_to, dec.cpu_dim) if ok_to else -1,
                         frm=coord_to_rank(peer_from, dec.cpu_dim) if ok_from else -1, shift=shift,
                         send_cells=send_cells, recv_cells=recv_cells))
    return np.flatnonzero(ghost_mask), segs


class DirectHaloPlan:
    """The 26-message halo of one neighbour list.  For a direction d in {-1,0,1}^3 this rank SENDS the atoms of its
    boundary cell layer S(d) (cell index 1 where d_k = -1, dim_k - 2 where d_k = +1, all inner cells where d_k = 0) to
    the rank at coord + d, with the periodic image shift added when that step crosses the box boundary, and RECEIVES
    from the rank at coord - d the atoms of ITS S(d), which fill this rank's ghost region R(-d) (cell index dim_k - 1
    where d_k = -1, 0 where d_k = +1).  Directions are walked in one fixed order on every rank, so the messages between
    one pair of ranks pair up by position (both transports match point-to-point messages in posting order).
    Ghost atoms are laid out region after region in that order: the receive buffer of a halo IS the ghost range."""

    def __init__(self, dec, ops):
        self.dec = dec
        timing = os.environ.get("IMD_DIST_TIMING") == "1"
        if timing:
            import time
            tm = [("start", time.perf_counter())]
        cs, cc = ops.cell_tables()
        if timing:
            tm.append(("cell_tables", time.perf_counter()))
        cs, cc = cs.astype(np.int64), cc.astype(np.int64)
        nlocal = ops.nlocal()
        D = dec.dim
        if getattr(dec, "_direct", None) is None:          # geometry only
            dec._direct = direct_segments(dec)
        ghost_cells, segs = dec._direct
        cc[ghost_cells] = 0
        self.segs = segs
        dev = ops.device
        # per-cell counts travel first: one batch of point-to-point messages in direction order (host arrays in, host arrays
        # out: segments to this rank itself never leave the host, the others share one device buffer each way)
        sends = [cc[g["send_cells"]].astype(np.int32) if g["to"] >= 0 else None for g in segs]
        recvs = _exchange_direct(dec, segs, sends, [len(g["recv_cells"]) for g in segs], dev)
        if timing:
            tm.append(("counts", time.perf_counter()))
        nghost = 0
        send_peer, send_cnt, send_shift, send_idx, recv_peer, recv_cnt = [], [], [], [], [], []
        for g, cnts in zip(segs, recvs):
            if g["frm"] < 0:
                continue
            cn = cnts.astype(np.int64)
            cells = g["recv_cells"]
            cc[cells] = cn
            cs[cells] = nlocal + nghost + np.concatenate([[0], np.cumsum(cn)[:-1]])
            recv_peer.append(g["frm"]); recv_cnt.append(int(cn.sum()))
            nghost += int(cn.sum())
        for g in segs:
            if g["to"] < 0:
                continue
            cells = g["send_cells"]
            n = cc[cells]
            idx = np.repeat(cs[cells] - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum()))
            send_peer.append(g["to"]); send_cnt.append(int(n.sum())); send_shift.append(g["shift"]); send_idx.append(idx)
        self.nghost = nghost
        if timing:
            tm.append(("index lists", time.perf_counter()))
        ops.set_ghost_layout(cs.astype(np.int32), cc.astype(np.int32), nghost)
        if timing:
            tm.append(("set_ghost_layout", time.perf_counter()))
        idx = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32)
        ops.dist_set_plan(send_peer, send_cnt, np.array(send_shift).reshape(-1, 3), torch.from_numpy(idx).to(dev),
                          recv_peer, recv_cnt)
        if timing:
            tm.append(("dist_set_plan", time.perf_counter()))
            print("[dist plan] ms: " + ", ".join("%s %.2f" % (tm[q][0], 1e3 * (tm[q][1] - tm[q - 1][1])) for q in range(1, len(tm))), file=sys.stderr)

    def exchange(self, ops, what):
        ops.dist_halo(what)


def _exchange_direct(dec, segs, sends, recv_len, dev):
    """one batch of point-to-point int32 messages in direction order (rebuild time: the per-cell counts), numpy in and out.
    Self-peers short-circuit on the host: my k-th self-send is my k-th self-receive.  Everything for other ranks is packed
    into ONE buffer per direction of travel (one host-to-device copy, the P2P batch on slices of it, one copy back)."""
    me = dec.rank
    out = [None] * len(segs)
    stage = _staged() or dev.type != "cuda"
    self_sends, remote_s, remote_r = [], [], []
    for g, b in zip(segs, sends):
        if g["to"] < 0:
            continue
        if g["to"] == me:
            self_sends.append(b)
        elif b.size > 0:
            remote_s.append((g["to"], b))
    k = 0
    for q, g in enumerate(segs):
        if g["frm"] < 0:
            continue
        if g["frm"] == me:
            out[q] = self_sends[k]; k += 1
        elif recv_len[q] > 0:
            remote_r.append((g["frm"], q))
        else:
            out[q] = np.zeros(0, np.int32)
    if remote_s or remote_r:
        tdev = torch.device("cpu") if stage else dev
        sbuf = torch.from_numpy(np.concatenate([b for _, b in remote_s]) if remote_s else np.zeros(0, np.int32)).to(tdev)
        rbuf = torch.empty(sum(recv_len[q] for _, q in remote_r), dtype=torch.int32, device=tdev)
        p2p, o = [], 0
        for peer, b in remote_s:
            p2p.append(dist.P2POp(dist.isend, sbuf[o:o + b.size], peer)); o += b.size
        o = 0
        for peer, q in remote_r:
            p2p.append(dist.P2POp(dist.irecv, rbuf[o:o + recv_len[q]], peer)); o += recv_len[q]
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
        host = rbuf.cpu().numpy()
        o = 0
        for peer, q in remote_r:
            out[q] = host[o:o + recv_len[q]]; o += recv_len[q]
    return out


def attach_ttm_mailbox(ops, rank, world, nbr):
    """Create (rank 0) / open the shared-memory file of the TTM face mailbox, map it and hand it to the library.
    Returns the objects that must stay alive.  One node only (the ranks share /dev/shm), like bench.py's contract."""
    import mmap
    per = ops.ttm_mailbox_bytes()
    name = [None]
    if rank == 0:
        name[0] = "/dev/shm/imdgpu_ttm_%d_%d" % (os.getpid(), int.from_bytes(os.urandom(4), "little"))
        with open(name[0], "wb") as f:
            f.truncate(per * world)                         # zero-filled
    if world > 1:
        dist.broadcast_object_list(name, src=0)
    f = open(name[0], "r+b")
    mm = mmap.mmap(f.fileno(), per * world)
    ops.ttm_mailbox_attach(mm, per, rank, world, nbr)
    if world > 1:
        dist.barrier()
    if rank == 0:
        os.unlink(name[0])                                  # the mappings keep it alive
    return f, mm


class DistributedMD:
    """The MD loop of one rank among several.  Rebuilds (migration, re-binning, ghost layout, halo plan, list) are
    driven from here; the step itself is ops.dist_md_step = imdgpu_dist_md_step (C + RCCL, imdgpu_dist.hip), which
    returns whether some rank's list became invalid (main_loop body, imd_main_3d.c:406-812)."""

    def __init__(self, dec, ops, ttm=False, local_group=None):
        """local_group: a LocalGroup -- the ranks are threads of THIS process (in-process transport of the library, no
        torch.distributed); otherwise the transport follows torch.distributed's backend (nccl -> RCCL, gloo -> host callback)."""
        self.dec, self.ops, self.ttm = dec, ops, ttm
        self.plan = None
        self.valid = False
        self.nrebuild = 0
        self.local_group = local_group
        if local_group is not None:
            self.world = local_group.world
        else:
            self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.ttm_decomposed = bool(ttm and getattr(ops, "ttm_is_decomposed", lambda: False)())
        self.ttm_adv = bool(ttm and getattr(ops, "ttm_has_adv", lambda: False)())
        # the rebuild is native (imdgpu_dist_rebuild) wherever the device layer offers it; the host-side plan below remains
        # for device stand-ins (CPU tests) and as the readable statement of the same plan (IMD_DIST_PY_REBUILD=1 forces it)
        self.native_rebuild = hasattr(ops, "dist_rebuild") and os.environ.get("IMD_DIST_PY_REBUILD") != "1"
        if local_group is not None:
            ops.dist_init_local(local_group, dec.rank)
        else:
            ops.dist_init(dec.rank, self.world)
        if ttm and self.ttm_decomposed:
            # every rank diffuses its own block; ghost faces go GPU -> host shared memory -> GPU (imdgpu_ttm_mailbox_*)
            nbr = []
            for d in range(3):
                nbr += [dec.lo[d] if dec.lo[d] is not None else -1, dec.hi[d] if dec.hi[d] is not None else -1]
            if local_group is not None:
                self._mailbox = local_group.attach_ttm_mailbox(ops, dec.rank, nbr)
            else:
                self._mailbox = attach_ttm_mailbox(ops, dec.rank, self.world, nbr)
        elif ttm and not self.native_rebuild:
            offs = []
            for r in range(self.world):
                c = rank_to_coord(r, dec.cpu_dim)
                offs += [c[d] * ops.ttm_block_dims()[d] for d in range(3)]
            ops.dist_set_ttm_blocks(offs)

    # -- rebuild: migration, re-binning, ghost layout, neighbour list ------------------------
    def _migrate(self):
        dec, ops = self.dec, self.ops
        for d in range(3):
            if dec.cpu_dim[d] == 1:
                continue
            n_lo, n_hi = ops.migrate_scan(d)
            cnt = torch.tensor([n_lo, n_hi], dtype=torch.int32, device=ops.device)
            f_lo, f_hi = _exchange(dec, d, cnt[0:1].clone() if dec.lo[d] is not None else None,
                                   cnt[1:2].clone() if dec.hi[d] is not None else None, (1, torch.int32), (1, torch.int32))
            in_lo = int(f_lo.item()) if f_lo is not None else 0       # what the low neighbour sends up to me
            in_hi = int(f_hi.item()) if f_hi is not None else 0
            b_lo = ops.migrate_gather(0) if dec.lo[d] is not None else None
            b_hi = ops.migrate_gather(1) if dec.hi[d] is not None else None
            if (dec.lo[d] is None and n_lo) or (dec.hi[d] is None and n_hi):
                raise RuntimeError("atoms left the simulation box through an open boundary")
            r_lo, r_hi = _exchange(dec, d, b_lo, b_hi, (in_lo * 10, torch.float64), (in_hi * 10, torch.float64))
            ops.migrate_apply(r_lo, in_lo, r_hi, in_hi)

    def rebuild(self):
        ops = self.ops
        if self.native_rebuild:
            ops.dist_rebuild()                   # migration, re-binning, plan, first halo, list: all in C (imdgpu_dist.hip)
            self.plan = _NativePlan()
            self.valid = True
            self.nrebuild += 1
            return
        timing = os.environ.get("IMD_DIST_TIMING") == "1"      # experiment: where a rebuild's time goes (stderr)
        if timing:
            import sys, time
            sync = getattr(ops, "synchronize", lambda: None)
            sync(); t0 = time.perf_counter(); marks = []

            def mark(name):
                sync(); marks.append((name, time.perf_counter()))
        else:
            def mark(name):
                pass
        if self.nrebuild > 0:
            self._migrate()
        mark("migrate")
        ops.fix_cells()
        mark("fix_cells")
        self.plan = DirectHaloPlan(self.dec, ops)
        mark("plan")
        changed = 1 if ops.autotile() else 0             # first rebuild: tile shape chosen from the real occupancies
        if dist.is_initialized() and self.nrebuild == 0:   # building a plan is collective (counts travel): every rank must
            t = torch.tensor([changed], dtype=torch.int32, device=ops.device)    # agree -- a rank in vacuum never re-tiles
            all_reduce_(t, dist.ReduceOp.MAX)
            changed = int(t.item())
        if changed:
            self.plan = DirectHaloPlan(self.dec, ops)    # local atoms were re-sorted -> new index lists
        self.plan.exchange(ops, 0)
        if ops.ntypes > 1:
            self.plan.exchange(ops, 2)
        mark("exchange")
        ops.make_nblist()
        mark("make_nblist")
        if timing:
            prev = t0
            out = []
            for name, t in marks:
                out.append("%s %.2f" % (name, 1e3 * (t - prev))); prev = t
            print("[dist rebuild %d] ms: %s" % (self.nrebuild, ", ".join(out)), file=sys.stderr)
        self.valid = True
        self.nrebuild += 1

    # -- one MD step ------------------------------------------------------------------------------
    def md_step(self, step):
        ops = self.ops
        if not self.valid:
            self.rebuild()
        if ops.dist_md_step(step):                   # some rank's list became invalid during this step
            self.valid = False

    def calc_forces(self):
        """forces of the CURRENT positions, complete (all three passes with their halos), e.g. before a download"""
        ops = self.ops
        if not self.valid:
            self.rebuild()
        else:
            self.plan.exchange(ops, 3)
        ops.forces_phase(1); ops.forces_phase(2)
        self.plan.exchange(ops, 1)
        ops.forces_phase(3)

    def global_sums(self):
        s = self.ops.get_sums()
        if self.local_group is not None:
            v = self.ops.dist_allreduce_sum([s["tot_pot_energy"], s["virial"], s["tot_kin_energy"], float(self.ops.nlocal())])
            return dict(tot_pot_energy=v[0], virial=v[1], tot_kin_energy=v[2], natoms=int(v[3]))
        t = torch.tensor([s["tot_pot_energy"], s["virial"], s["tot_kin_energy"], float(self.ops.nlocal())],
                         dtype=torch.float64, device=self.ops.device)
        if dist.is_initialized():
            all_reduce_(t)
        return dict(tot_pot_energy=float(t[0]), virial=float(t[1]), tot_kin_energy=float(t[2]), natoms=int(t[3]))


class _NativePlan:
    """after a native rebuild the plan lives in the library: halos go through imdgpu_dist_halo"""

    def exchange(self, ops, what):
        ops.dist_halo(what)


class LocalGroup:
    """N ranks as N threads of one process (imdgpu_local_group_*): how many ranks rehearse on ONE GPU when the box admits only
    a few processes, e.g. the 2x2x2 grid of BASELINE config 4 on a one-GPU box.  run(fn) starts fn(rank) on a thread per rank
    (ctypes releases the GIL inside the library) and returns the results in rank order."""

    def __init__(self, world):
        import ctypes as C
        import threading
        from . import lib
        self.world, self.L = int(world), lib.load()
        g = C.c_void_p()
        if self.L.imdgpu_local_group_create(self.world, C.byref(g)):
            raise RuntimeError("imdgpu_local_group_create failed")
        self.handle = g
        self._lock = threading.Lock()
        self._shared = {}

    def barrier(self):
        if self.L.imdgpu_local_group_barrier(self.handle):
            raise RuntimeError("local group: a rank failed or did not arrive")

    def attach_ttm_mailbox(self, ops, rank, nbr):
        """the face mailbox of the decomposed lattice: ordinary host memory shared by the threads"""
        import ctypes as C
        per = ops.ttm_mailbox_bytes()
        p = C.c_void_p()
        if self.L.imdgpu_local_group_shared(self.handle, b"ttm_mailbox", per * self.world, C.byref(p)):
            raise RuntimeError("imdgpu_local_group_shared failed")
        ops.ttm_mailbox_attach(p, per, rank, self.world, nbr)
        self.barrier()
        return p

    def run(self, fn, timeout=600):
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException:
                import traceback
                err[r] = traceback.format_exc()
        th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout)
        bad = [r for r in range(self.world) if err[r] is not None or th[r].is_alive()]
        if bad:
            raise RuntimeError("local ranks failed: " + "; ".join("rank %d: %s" % (r, err[r] or "still running") for r in bad))
        return out

    def close(self):
        if self.handle:
            self.L.imdgpu_local_group_destroy(self.handle)
            self.handle = None


def make_host_transport(ops, rank, world):
    """imdgpu_exchange_fn for rehearsals without RCCL: device buffers are staged through the host and travel over
    torch.distributed (gloo).  Messages between one pair of ranks match in posting order, like RCCL's."""
    import ctypes as C
    L = ops.L
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    LL = C.POINTER(C.c_longlong)
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), LL, LL, C.c_int,
                     C.POINTER(C.c_int), LL, LL, C.c_int)

    def d2h(ptr, nbytes):
        a = np.empty(nbytes, np.uint8)
        if nbytes and hip.hipMemcpy(a.ctypes.data, ptr, nbytes, 2):
            raise RuntimeError("hipMemcpy D2H failed")
        return a

    def h2d(ptr, a):
        a = np.ascontiguousarray(a)
        if a.nbytes and hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1):
            raise RuntimeError("hipMemcpy H2D failed")

    def fn(user, op, dsend, drecv, nsend, sp, so, sc, nrecv, rp, ro, rc, eb):
        try:
            if op == 0:
                reqs, outs = [], []
                for k in range(nsend):
                    buf = torch.from_numpy(d2h(dsend + so[k] * eb, sc[k] * eb))
                    reqs.append(dist.P2POp(dist.isend, buf, int(sp[k])))
                for k in range(nrecv):
                    buf = torch.empty(rc[k] * eb, dtype=torch.uint8)
                    outs.append((ro[k], buf))
                    reqs.append(dist.P2POp(dist.irecv, buf, int(rp[k])))
                if reqs:
                    for w in dist.batch_isend_irecv(reqs):
                        w.wait()
                for off, buf in outs:
                    h2d(drecv + off * eb, buf.numpy())
            elif op in (1, 3):
                dt = np.int32 if op == 1 else np.float64
                t = torch.from_numpy(d2h(dsend, nsend * eb).view(dt).copy())
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
                h2d(drecv, t.numpy())
            elif op == 2:
                t = torch.from_numpy(d2h(dsend, nsend * eb).view(np.float64).copy())
                out = [torch.empty_like(t) for _ in range(world)]
                if world > 1:
                    dist.all_gather(out, t)
                else:
                    out = [t]
                h2d(drecv, torch.cat(out).numpy())
            else:
                return 1
            return 0
        except BaseException:
            import traceback
            traceback.print_exc()
            return 1
    return FN(fn)


class GpuOps:
    """libimdgpu behind the interface DistributedMD / HaloPlan need; buffers are torch CUDA tensors."""

    def __init__(self, sysm, device):
        import ctypes as C
        self.C, self.s, self.L, self.h = C, sysm, sysm.L, sysm.h
        self.device = torch.device("cuda", device)
        self.ntypes = sysm.ntypes
        # one side stream shared by torch (halo buffers, RCCL ordering) and the library (kernels, hipGraph
        # capture of the TTM sub-steps is not allowed on the legacy default stream)
        self.stream = torch.cuda.Stream(self.device)
        torch.cuda.set_stream(self.stream)
        sysm.set_stream(self.stream.cuda_stream)

    def synchronize(self):
        self.s.synchronize()

    def _ck(self, rc):
        self.s._ck(rc)

    def nlocal(self):
        return self.s.num_atoms()[0]

    def cell_tables(self):
        cs = np.zeros(self.s.nallcells, np.int32); cc = np.zeros(self.s.nallcells, np.int32)
        C = self.C
        self._ck(self.L.imdgpu_get_cell_tables(self.h, cs.ctypes.data_as(C.POINTER(C.c_int)), cc.ctypes.data_as(C.POINTER(C.c_int))))
        return cs, cc

    def set_ghost_layout(self, cs, cc, nghost):
        C = self.C
        cs, cc = np.ascontiguousarray(cs, np.int32), np.ascontiguousarray(cc, np.int32)
        self._ck(self.L.imdgpu_set_ghost_layout(self.h, cs.ctypes.data_as(C.POINTER(C.c_int)),
                                                cc.ctypes.data_as(C.POINTER(C.c_int)), int(nghost)))

    def gather(self, what, idx, shift):
        C = self.C
        w, dt = WHAT_WIDTH[what]
        out = torch.empty(idx.numel() * w, dtype=dt, device=self.device)
        sh = np.ascontiguousarray(shift, np.float64)
        self._ck(self.L.imdgpu_halo_gather(self.h, what, C.c_void_p(idx.data_ptr()), idx.numel(),
                                           sh.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(out.data_ptr())))
        return out

    def gather2(self, what, idx, n_lo, shift_lo, shift_hi):
        C = self.C
        w, dt = WHAT_WIDTH[what]
        out = torch.empty(idx.numel() * w, dtype=dt, device=self.device)
        lo, hi = np.ascontiguousarray(shift_lo, np.float64), np.ascontiguousarray(shift_hi, np.float64)
        dp = C.POINTER(C.c_double)
        self._ck(self.L.imdgpu_halo_gather2(self.h, what, C.c_void_p(idx.data_ptr()), int(n_lo), idx.numel(),
                                            lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), C.c_void_p(out.data_ptr())))
        return out

    def scatter(self, what, idx, buf):
        C = self.C
        self._ck(self.L.imdgpu_halo_scatter(self.h, what, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(buf.data_ptr())))

    def migrate_scan(self, d):
        C = self.C
        a, b = C.c_int(), C.c_int()
        self._ck(self.L.imdgpu_migrate_scan(self.h, d, C.byref(a), C.byref(b)))
        self._mig = (a.value, b.value)
        return a.value, b.value

    def migrate_gather(self, direction):
        C = self.C
        out = torch.empty(self._mig[direction] * 10, dtype=torch.float64, device=self.device)
        self._ck(self.L.imdgpu_migrate_gather(self.h, direction, C.c_void_p(out.data_ptr())))
        return out

    def migrate_apply(self, r_lo, n_lo, r_hi, n_hi):
        C = self.C
        self._ck(self.L.imdgpu_migrate_apply(self.h, C.c_void_p(r_lo.data_ptr() if n_lo else 0), n_lo,
                                             C.c_void_p(r_hi.data_ptr() if n_hi else 0), n_hi))

    # -- native step (imdgpu_dist.hip) -------------------------------------------------------------------
    def dist_init(self, rank, world):
        """RCCL communicators created by the library itself when torch.distributed runs on nccl (or alone on one rank);
        under gloo (several ranks sharing one GPU in the tests) the same C step logic gets a host-staged transport."""
        C = self.C
        use_rccl = (not dist.is_initialized() and world == 1) or (dist.is_initialized() and dist.get_backend() == "nccl")
        if os.environ.get("IMDGPU_DIST_TRANSPORT") == "callback":
            use_rccl = False
        if use_rccl:
            uid = [None]
            if rank == 0:
                buf = C.create_string_buffer(384)          # three ids: halo + flags, lattice, the position halo under pass 1
                self._ck(self.L.imdgpu_dist_unique_id(buf, 384))
                uid[0] = bytes(buf.raw)
            if dist.is_initialized():
                dist.broadcast_object_list(uid, src=0)
            self._ck(self.L.imdgpu_dist_init_rccl(self.h, world, rank, uid[0], 384))
            self.transport = "rccl"
        else:
            self._cb = make_host_transport(self, rank, world)
            self._ck(self.L.imdgpu_dist_init_callback(self.h, world, rank, C.cast(self._cb, C.c_void_p), None))
            self.transport = "host callback (%s)" % (dist.get_backend() if dist.is_initialized() else "single rank")

    def dist_init_local(self, group, rank):
        self._ck(self.L.imdgpu_dist_init_local(self.h, group.handle, int(rank)))
        self.transport = "in-process (%d ranks on threads)" % group.world

    def dist_rebuild(self):
        self._ck(self.L.imdgpu_dist_rebuild(self.h))

    def dist_allreduce_sum(self, vals):
        a = np.ascontiguousarray(vals, np.float64)
        self._ck(self.L.imdgpu_dist_allreduce_sum(self.h, a.ctypes.data_as(self.C.POINTER(self.C.c_double)), len(a)))
        return [float(v) for v in a]

    def dist_set_plan(self, send_peer, send_cnt, send_shift, send_idx, recv_peer, recv_cnt):
        C = self.C
        ip = C.POINTER(C.c_int)
        sp, sc = np.ascontiguousarray(send_peer, np.int32), np.ascontiguousarray(send_cnt, np.int32)
        rp, rc = np.ascontiguousarray(recv_peer, np.int32), np.ascontiguousarray(recv_cnt, np.int32)
        sh = np.ascontiguousarray(send_shift, np.float64).reshape(-1)
        self._ck(self.L.imdgpu_dist_set_plan(self.h, len(sp), sp.ctypes.data_as(ip), sc.ctypes.data_as(ip),
                                             sh.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(send_idx.data_ptr()),
                                             len(rp), rp.ctypes.data_as(ip), rc.ctypes.data_as(ip)))

    def dist_halo(self, what):
        self._ck(self.L.imdgpu_dist_halo(self.h, int(what)))

    def dist_set_ttm_blocks(self, offs):
        a = np.ascontiguousarray(offs, np.int32)
        self._ck(self.L.imdgpu_dist_set_ttm_blocks(self.h, a.ctypes.data_as(self.C.POINTER(self.C.c_int))))

    def dist_md_step(self, step):
        v = self.C.c_int()
        self._ck(self.L.imdgpu_dist_md_step(self.h, int(step), self.C.byref(v)))
        return bool(v.value)

    def fix_cells(self): self.s.fix_cells()

    def autotile(self):
        v = self.C.c_int()
        self._ck(self.L.imdgpu_autotile(self.h, self.C.byref(v)))
        return bool(v.value)
    def make_nblist(self): self.s.make_nblist()
    def forces_phase(self, k): self._ck(self.L.imdgpu_forces_phase(self.h, k))
    def integrate(self): self._ck(self.L.imdgpu_integrate(self.h))
    def invalidate(self): self._ck(self.L.imdgpu_invalidate(self.h))
    def get_sums(self): return self.s.get_sums()

    def get_invalid(self):
        v = self.C.c_int()
        self._ck(self.L.imdgpu_get_invalid(self.h, self.C.byref(v)))
        return v.value

    # TTM (replicated lattice)
    def ttm_block_dims(self):
        return tuple((self.s.cell_dim[d] - 2) // self.s.fd_ext[d] for d in range(3))

    def ttm_is_decomposed(self):
        return bool(getattr(self.s, "ttm_decomposed", False))

    def ttm_mailbox_bytes(self):
        C = self.C
        n = C.c_size_t()
        self._ck(self.L.imdgpu_ttm_mailbox_bytes(self.h, C.byref(n)))
        return int(n.value)

    def ttm_mailbox_attach(self, mm, per, rank, world, nbr):
        C = self.C
        base = mm.value if isinstance(mm, C.c_void_p) else C.addressof(C.c_char.from_buffer(mm))
        arr = (C.c_int * 6)(*[int(v) for v in nbr])
        self._ck(self.L.imdgpu_ttm_mailbox_attach(self.h, C.c_void_p(base), per, int(rank), int(world), arr))

    def ttm_has_adv(self):
        return bool(self.s.get_info("ttm_adv"))

    def ttm_local_block(self, step):
        C = self.C
        n = C.c_int()
        self._ck(self.L.imdgpu_ttm_local_block(self.h, step, None, C.byref(n)))
        out = torch.empty(n.value * self.s.get_info("ttm_block_width"), dtype=torch.float64, device=self.device)
        self._ck(self.L.imdgpu_ttm_local_block(self.h, step, C.c_void_p(out.data_ptr()), C.byref(n)))
        return out

    def ttm_apply_blocks(self, step, allb, nblocks, offs):
        C = self.C
        self._ck(self.L.imdgpu_ttm_apply_blocks(self.h, step, C.c_void_p(allb.data_ptr()), nblocks, C.c_void_p(offs.data_ptr())))

    def ttm_lattice_step(self, step):
        self._ck(self.L.imdgpu_ttm_lattice_step(self.h, step))
