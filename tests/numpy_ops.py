"""TEST INFRASTRUCTURE: a numpy stand-in for the per-rank device (imd_amd.parallel.GpuOps) so that the
multi-rank host logic (decomposition, staged halo, migration, step sequence) runs under gloo on CPUs.
Forces are brute force over local + ghost atoms with the same 3-point table rule."""
import numpy as np
import torch


def tab_eval(t, r2):
    """value and 2*derivative of a one-column pot_table dict (PAIR_INT2, potaccess.h:323-354)."""
    a = np.minimum(r2, t["end"][0]) - t["begin"][0]
    a = np.maximum(a, 0.0) * t["invstep"][0]
    k = a.astype(np.int64)
    chi = a - k
    tb = t["table"]
    p0, p1, p2 = tb[k], tb[k + 1], tb[k + 2]
    dv, d2v = p1 - p0, p2 - 2 * p1 + p0
    return p0 + chi * dv + 0.5 * chi * (chi - 1) * d2v, 2 * t["invstep"][0] * (dv + (chi - 0.5) * d2v)


class NumpyOps:
    def __init__(self, dec, pair, rho, emb, cellsz, nbl_margin, timestep):
        self.dec, self.pair, self.rho, self.emb = dec, pair, rho, emb
        self.cellsz, self.margin, self.dt = cellsz, nbl_margin, timestep
        self.device = torch.device("cpu")
        self.ntypes = 1
        self.n = 0
        self.invalid = 0
        self.sums = dict(tot_pot_energy=0.0, virial=0.0, tot_kin_energy=0.0)

    # -- atoms ---------------------------------------------------------------------
    def upload(self, pos, mom, mass, ident):
        self.n = len(pos)
        self.pos = np.zeros((self.n, 4)); self.pos[:, :3] = pos
        self.p, self.mass, self.id = mom.copy(), np.broadcast_to(mass, (self.n,)).copy(), ident.copy()
        self.type = np.zeros(self.n, np.int32)

    def nlocal(self):
        return self.n

    def _local_lin(self, pos):
        dec = self.dec
        t = np.linalg.inv(dec.box).T
        s = pos[:, :3] @ t.T
        c = np.clip((np.array(dec.gdim) * s).astype(np.int64), 0, np.array(dec.gdim) - 1)
        loc = c - np.array(dec.coord) * (np.array(dec.dim) - 2) + 1
        return loc

    def fix_cells(self):
        dec = self.dec
        t = np.linalg.inv(dec.box).T
        for d in range(3):
            if dec.pbc[d]:
                s = -np.floor(self.pos[:self.n, :3] @ t[d])
                self.pos[:self.n, :3] += s[:, None] * dec.box[d]
        loc = self._local_lin(self.pos[:self.n])
        assert ((loc >= 1) & (loc <= np.array(dec.dim) - 2)).all(), "atom outside this rank's domain after migration"
        lin = (loc[:, 0] * dec.dim[1] + loc[:, 1]) * dec.dim[2] + loc[:, 2]
        o = np.argsort(lin, kind="stable")
        for name in ("pos", "p", "mass", "id", "type"):
            a = getattr(self, name); a[:self.n] = a[:self.n][o]
        self.cc = np.bincount(lin, minlength=dec.nall).astype(np.int32)
        self.cs = np.concatenate([[0], np.cumsum(self.cc)[:-1]]).astype(np.int32)
        self.invalid = 0

    def autotile(self):
        return False

    def cell_tables(self):
        return self.cs.copy(), self.cc.copy()

    def set_ghost_layout(self, cs, cc, nghost):
        self.cs, self.cc, self.nghost = cs.copy(), cc.copy(), nghost
        allp = np.zeros((self.n + nghost, 4)); allp[:self.n] = self.pos[:self.n]
        self.pos = allp

    def gather(self, what, idx, shift):
        i = idx.numpy().astype(np.int64)
        if what == 0:
            out = self.pos[i].copy(); out[:, :3] += np.asarray(shift)
            return torch.from_numpy(out.reshape(-1))
        if what == 1:
            return torch.from_numpy(self.pos[i, 3].copy())
        if what == 3:
            return torch.from_numpy((self.pos[i, :3] + np.asarray(shift)).reshape(-1))
        return torch.zeros(len(i), dtype=torch.int32)

    def gather2(self, what, idx, n_lo, shift_lo, shift_hi):
        lo = self.gather(what, idx[:n_lo], shift_lo)
        hi = self.gather(what, idx[n_lo:], shift_hi)
        return torch.cat([lo, hi])

    def scatter(self, what, idx, buf):
        i = idx.numpy().astype(np.int64)
        if what == 0:
            self.pos[i] = buf.numpy().reshape(-1, 4)
        elif what == 1:
            self.pos[i, 3] = buf.numpy()
        elif what == 3:
            self.pos[i, :3] = buf.numpy().reshape(-1, 3)

    # -- the native step's interface (imdgpu_dist_*), restated for the stand-in: same plan, same sequence ------------
    def dist_init(self, rank, world):
        self.rank, self.world = rank, world
        self._halo_ready = False
        self._ttm_offs = None

    def dist_set_plan(self, send_peer, send_cnt, send_shift, send_idx, recv_peer, recv_cnt):
        self._plan = dict(sp=list(send_peer), sc=list(send_cnt), sh=np.asarray(send_shift, float).reshape(-1, 3),
                          idx=send_idx.numpy().astype(np.int64), rp=list(recv_peer), rc=list(recv_cnt))
        assert sum(recv_cnt) == self.nghost
        self._halo_ready = False

    def dist_halo(self, what):
        import torch.distributed as dist
        P = self._plan
        w = {0: 4, 1: 1, 2: 1, 3: 3}[what]
        off, bufs = 0, []
        for k, n in enumerate(P["sc"]):
            i = P["idx"][off:off + n]; off += n
            if what == 0:
                b = self.pos[i].copy(); b[:, :3] += P["sh"][k]
            elif what == 1:
                b = self.pos[i, 3].copy()
            elif what == 2:
                b = self.type[i].astype(np.float64) if len(self.type) > i.max(initial=-1) else np.zeros(n)
            else:
                b = self.pos[i, :3] + P["sh"][k]
            bufs.append(np.ascontiguousarray(b, dtype=np.float64).reshape(-1))
        reqs, selfq, outs = [], [], []
        for k, n in enumerate(P["sc"]):
            if n == 0:
                continue
            if P["sp"][k] == self.rank:
                selfq.append(bufs[k])
            else:
                reqs.append(dist.P2POp(dist.isend, torch.from_numpy(bufs[k]), P["sp"][k]))
        at, q = self.n, 0
        for k, n in enumerate(P["rc"]):
            if n == 0:
                continue
            if P["rp"][k] == self.rank:
                outs.append((at, n, torch.from_numpy(selfq[q]))); q += 1
            else:
                t = torch.empty(n * w, dtype=torch.float64)
                reqs.append(dist.P2POp(dist.irecv, t, P["rp"][k]))
                outs.append((at, n, t))
            at += n
        if reqs:
            for r in dist.batch_isend_irecv(reqs):
                r.wait()
        for at, n, t in outs:
            a = t.numpy()
            if what == 0:
                self.pos[at:at + n] = a.reshape(-1, 4)
            elif what == 1:
                self.pos[at:at + n, 3] = a
            elif what == 3:
                self.pos[at:at + n, :3] = a.reshape(-1, 3)

    def dist_set_ttm_blocks(self, offs):
        self._ttm_offs = torch.tensor(list(offs), dtype=torch.int32)

    def dist_md_step(self, step):
        import torch.distributed as dist
        if not self._halo_ready:
            self.dist_halo(3)
        self._halo_ready = False
        self.forces_phase(1)
        if self._ttm_offs is not None:               # replicated lattice: block -> all-gather -> fold in -> lattice step
            blk = self.ttm_local_block(step)
            out = [torch.empty_like(blk) for _ in range(self.world)]
            dist.all_gather(out, blk)
            self.ttm_apply_blocks(step, torch.cat(out), self.world, self._ttm_offs)
            self.ttm_lattice_step(step)
        self.forces_phase(2)
        self.dist_halo(1)
        self.forces_phase(3)
        self.integrate()
        self.dist_halo(3)
        flag = torch.tensor([self.invalid], dtype=torch.int32)
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            return True
        self._halo_ready = True
        return False

    # -- migration -------------------------------------------------------------------
    def migrate_scan(self, d):
        dec = self.dec
        t = np.linalg.inv(dec.box).T
        s = self.pos[:self.n, :3] @ t[d]
        if dec.pbc[d]:
            s = s - np.floor(s)
        c = np.clip((dec.gdim[d] * s).astype(np.int64), 0, dec.gdim[d] - 1)
        dr = c // (dec.dim[d] - 2) - dec.coord[d]
        P = dec.cpu_dim[d]
        if dec.pbc[d]:
            dr = np.where(2 * dr > P, dr - P, dr); dr = np.where(2 * dr < -P, dr + P, dr)
        self._flag = np.where(dr < 0, 1, np.where(dr > 0, 2, 0))
        return int((self._flag == 1).sum()), int((self._flag == 2).sum())

    def migrate_gather(self, direction):
        m = self._flag == direction + 1
        rec = np.zeros((int(m.sum()), 10))
        rec[:, :3], rec[:, 3:6], rec[:, 6], rec[:, 8] = self.pos[:self.n][m, :3], self.p[m], self.mass[m], self.id[m]
        return torch.from_numpy(rec.reshape(-1))

    def migrate_apply(self, r_lo, n_lo, r_hi, n_hi):
        stay = self._flag == 0
        recs = [r.numpy().reshape(-1, 10) for r, n in ((r_lo, n_lo), (r_hi, n_hi)) if n]
        inc = np.concatenate(recs) if recs else np.zeros((0, 10))
        pos = np.zeros((stay.sum() + len(inc), 4)); pos[:stay.sum(), :3] = self.pos[:self.n][stay, :3]; pos[stay.sum():, :3] = inc[:, :3]
        self.p = np.concatenate([self.p[stay], inc[:, 3:6]])
        self.mass = np.concatenate([self.mass[stay], inc[:, 6]])
        self.id = np.concatenate([self.id[stay], inc[:, 8].astype(self.id.dtype)])
        self.type = np.zeros(len(pos), np.int32)
        self.pos, self.n = pos, len(pos)

    # -- forces --------------------------------------------------------------------------
    def make_nblist(self):
        """pair candidates (i local, j local or ghost, i != j) inside the list radius, found once per rebuild by a
        chunked all-pairs search; the force phases then work on these sparse pair arrays only"""
        self.ref = self.pos[:self.n, :3].copy()
        X = self.pos[:, :3]
        ii, jj = [], []
        for a in range(0, self.n, 256):
            b = min(a + 256, self.n)
            d = X[None, :, :] - X[a:b, None, :]
            r2 = (d * d).sum(-1)
            r2[np.arange(b - a), np.arange(a, b)] = 1e30
            i, j = np.nonzero(r2 < self.cellsz)
            ii.append(i + a); jj.append(j)
        self._ii, self._jj = np.concatenate(ii), np.concatenate(jj)

    def _acc(self, w):
        return np.bincount(self._ii, weights=w, minlength=self.n)

    def forces_phase(self, k):
        n = self.n
        ii, jj = self._ii, self._jj
        if k == 1:
            X = self.pos[:, :3]
            self._d = X[jj] - X[ii]
            self._r2 = r2 = (self._d * self._d).sum(-1)
            pot, grad = tab_eval(self.pair, r2)
            inp = r2 <= self.pair["end"][0]
            grad = np.where(inp, grad, 0.0); pot = np.where(inp, pot, 0.0)
            self.f = np.stack([self._acc(self._d[:, c] * grad) for c in range(3)], axis=1)
            self.epot = 0.5 * self._acc(pot)
            val, _ = tab_eval(self.rho, r2)
            self.rhoh = self._acc(np.where(r2 < self.rho["end"][0], val, 0.0))
            self.vir = -0.5 * (r2 * grad).sum()
        elif k == 2:
            pot, dF = tab_eval(self.emb, self.rhoh)
            self.epot += pot
            self.pos[:n, 3] = dF
        else:
            r2 = self._r2
            _, rp = tab_eval(self.rho, r2)
            w = self.pos[:, 3]
            grad = np.where(r2 < self.rho["end"][0], 0.5 * (w[ii] + w[jj]) * rp, 0.0)
            g = self._d * grad[:, None]
            self.f += np.stack([self._acc(g[:, c]) for c in range(3)], axis=1)
            self.vir += -0.5 * (self._d * g).sum()
            self.sums["tot_pot_energy"], self.sums["virial"] = float(self.epot.sum()), float(self.vir)

    def integrate(self):
        n = self.n
        e1 = (self.p ** 2).sum(1)
        self.p += self.dt * self.f
        e2 = (self.p ** 2).sum(1)
        self.sums["tot_kin_energy"] = float(((e1 + e2) / (4 * self.mass)).sum())
        self.pos[:n, :3] += (self.dt / self.mass)[:, None] * self.p
        dis = ((self.pos[:n, :3] - self.ref) ** 2).sum(1).max()
        self.invalid = int(dis > (0.5 * self.margin) ** 2)

    def get_invalid(self):
        return self.invalid

    def invalidate(self):
        pass

    def get_sums(self):
        return dict(self.sums)
