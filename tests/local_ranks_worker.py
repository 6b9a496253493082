"""Runs N ranks of the multi-rank MD as N THREADS of this one process on cuda:0 (imd_amd.parallel.LocalGroup: the library's
in-process transport, native rebuild and native step) and stores what every rank holds at the end.  Started by
tests/test_gpu_multirank_local.py as ONE child process per case -- a GPU box admits only a few processes on its card, so
the 2x2x2 grid of BASELINE config 4 cannot be rehearsed with a process per rank.

usage: python tests/local_ranks_worker.py <case> <cx> <cy> <cz> <nsteps> <ttm 0|1|2|3> <decomposed 0|1> <out.npz>
(ttm: 1 = the 3D lattice, 2 = imd_ttm.c's 1D model, 3 = the 3D lattice with do_ADV / do_COMMFLUX in every step)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, cpu_dim, nsteps = sys.argv[1], tuple(int(v) for v in sys.argv[2:5]), int(sys.argv[5])
    ttm, decomposed, outpath = bool(int(sys.argv[6])), bool(int(sys.argv[7])), sys.argv[8]
    ttm1d = int(sys.argv[6]) == 2                 # imd_ttm.c's 1D model (with its laser source / electron pressure if the fixture has them)
    adv = int(sys.argv[6]) == 3                   # calc_ttm with do_ADV(1.0) after update_fd, as the reference compiles it (ADVMODE 1)
    if ttm1d:
        ttm = False
    from tests import golden_util as gu
    from imd_amd.parallel import Decomposition, DistributedMD, GpuOps, LocalGroup
    from imd_amd.system import GpuSystem
    G = gu.load(case)
    world = int(np.prod(cpu_dim))
    group = LocalGroup(world)
    pos, mom, mass, ident = G["s0_ort"], G["s0_impuls"], G["s0_masse"], G["s0_nummer"]
    ntypes = int(G["ntypes"])
    gdim = tuple(int(v) for v in G["global_cell_dim"])
    t = np.linalg.inv(G["box"]).T

    def rank_body(rank):
        dec = Decomposition(rank, cpu_dim, gdim, G["box"], G["pbc_dirs"])
        c = np.clip((np.array(dec.gdim) * (pos @ t.T)).astype(int), 0, np.array(dec.gdim) - 1)
        mine = np.all(c // (np.array(dec.dim) - 2) == np.array(dec.coord), axis=1)
        s = GpuSystem(G["box"], G["pbc_dirs"], dec.dim, ntypes, float(G["cellsz"]), float(G["nbl_margin"]), float(G["timestep"]),
                      device=0, global_cell_dim=dec.gdim, cpu_dim=cpu_dim, my_coord=dec.coord, ttm_decomposed=decomposed,
                      count_ttm_neighbours=ttm or ttm1d)
        s.upload_tables(gu.table(G, "pair_pot"), gu.table(G, "rho_h_tab"), gu.table(G, "embed_pot"))
        if adv:
            s.set_option("ttm_adv", 1)
        if ttm:
            s.init_ttm(G["ttm_fd_ext"], int(G["ttm_fd_min_atoms"]), int(G["ttm_diff_substeps"]), float(G["ttm_atomic_weight"]),
                       float(G["ttm_atomic_charge"]), dict(x=G["eos_r"], y=G["eos_t"], z=G["eos_cve"]),
                       dict(x=G["eos_r"], y=G["eos_ts"], z=G["eos_ee"]), float(G["ttm_omega_laser"]))
        s.upload_atoms(pos[mine], mom[mine], mass[mine], G["s0_sorte"][mine].astype(np.int32) if ntypes > 1 else None, ident[mine])
        if ttm1d:
            s.init_ttm1d(int(G["ttm_global_fd_dim"]), int(G["ttm_fd_min_atoms"]), int(G["ttm_fd_n_timesteps"]), int(G["ttm_vlatdim"]),
                         int(G["ttm_vlatbuffer"]), float(G["ttm_vlatdens"]), float(G["ttm_atomic_weight"]), float(G["ttm_atomic_charge"]),
                         dict(x=G["eos_r"], y=G["eos_t"], z=G["eos_cve"]), dict(x=G["eos_r"], y=G["eos_ts"], z=G["eos_ee"]),
                         ref_ranks=int(G["ref_ranks"]), adv=True)
            if "tmm_I0" in G:
                s.init_ttm1d_tmm(float(G["tmm_I0"]), float(G["tmm_lam"]), float(G["tmm_sigma_t"]), float(G["tmm_t0"]), G["tmm_eps_bb"], G["tmm_K12"],
                                 dict(x=G["eos_r"], y=G["eos_t"], z=G["tmm_phase_z"]), tmm_threshold=float(G["tmm_abs_threshold"]))
            if "epress_z" in G:
                s.ttm1d_set_epress(dict(x=G["eos_r"], y=G["eos_t"], z=G["epress_z"]))
        miss = os.environ.get("IMD_TEST_PERSIST_MISS", "")         # "<rank>:<step>": that rank's persistent sub-step launch of that step steps aside
        if miss and int(miss.split(":")[0]) == rank:
            s.set_option("ttm_persist_miss", int(miss.split(":")[1]))
        for kv in filter(None, os.environ.get("IMD_TEST_OPTIONS", "").split(",")):      # "name=value,...": library options for this run
            s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        if os.environ.get("IMD_TEST_HALO_OVERLAP"):                # pass 1 in its queue form under the position halo
            s.set_option("halo_overlap", int(os.environ["IMD_TEST_HALO_OVERLAP"]))
        ops = GpuOps(s, 0)
        md = DistributedMD(dec, ops, ttm=ttm, local_group=group)
        assert md.native_rebuild
        trace = {}
        for st in range(nsteps):
            md.md_step(st)
            if os.environ.get("IMD_TEST_TRACE"):                  # diagnostics: what every step left behind (see tools/)
                for k, v in s.download(fields=("impuls", "kraft", "pot_eng", "eam_rho", "eam_dF", "nummer")).items():
                    trace["s%d_%s" % (st, k)] = v
        md.calc_forces()
        sums = md.global_sums()
        d = s.download(fields=("ort", "impuls", "kraft", "nummer", "numneighs"))
        redone = s.get_info("ttm_persist_redone") if ttm else 0
        lat = s.ttm_download(fields=("temp", "xi", "natoms") + (("U", "temp2", "U2", "flux") if adv else ())) if ttm else {}
        if ttm1d:                              # calc_ttm of the step after the last: collective (the slab sums), every rank calls it
            s.ttm1d_calc(nsteps)
            L1 = s.ttm1d_download()
            lat = {k: L1[k] for k in ("temp", "xi", "natoms", "md_temp", "dens", "U", "source", "natomsglobal", "xiarr_global")}
        info = dict(nrebuild=md.nrebuild, transport=ops.transport, list_mode=s.get_info("list_mode"),
                    persist=s.get_info("ttm_persist_mode") if ttm else -9, remote=s.get_info("ttm_remote") if ttm else -9,
                    runs=s.get_info("ttm_persist_runs") if ttm else -9, redone=redone, overlapped=s.get_info("halo_overlap_passes"),
                    interior=s.get_info("halo_overlap_interior_tiles"), ntiles=s.get_info("ntiles"))
        group.barrier()                       # nobody tears its context down while another rank still runs kernels against it
        s.close()
        d.update(trace)
        return dict(d=d, lat=lat, sums=sums, coord=dec.coord, info=info)

    res = group.run(rank_body, timeout=400)
    out = {}
    for r, R in enumerate(res):
        for k, v in R["d"].items():
            out["r%d_%s" % (r, k)] = v
        for k, v in R["lat"].items():
            out["r%d_lat_%s" % (r, k)] = v
        out["r%d_coord" % r] = np.array(R["coord"])
        out["r%d_nrebuild" % r] = R["info"]["nrebuild"]
        out["r%d_list_mode" % r] = R["info"]["list_mode"]
        out["r%d_persist" % r] = R["info"]["persist"]
        out["r%d_remote" % r] = R["info"]["remote"]
        out["r%d_persist_runs" % r] = R["info"]["runs"]
        out["r%d_persist_redone" % r] = R["info"]["redone"]
        out["r%d_overlapped" % r] = R["info"]["overlapped"]
        out["r%d_interior_tiles" % r] = R["info"]["interior"]
        out["r%d_ntiles" % r] = R["info"]["ntiles"]
        for k in ("tot_pot_energy", "tot_kin_energy", "virial", "natoms"):
            out["r%d_%s" % (r, k)] = R["sums"][k]
    out["transport"] = np.array(res[0]["info"]["transport"])
    np.savez(outpath, **out)
    group.close()


if __name__ == "__main__":
    main()
