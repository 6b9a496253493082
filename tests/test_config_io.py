"""Configuration files (.chkpt / coordname input) against the reference binary (CPU tests).

Fixture tests/golden/chkpt.npz (tests/golden/make_golden.py chkpt): the ASCII and the binary checkpoint the reference
wrote after step 10 of a 256-atom EAM run, the reference's memory around that step (probe dumps), a binary file written
by imd_amd.config_io.write_config, and the reference's memory after READING that file."""
import os

import numpy as np

from imd_amd import config_io
from tests import golden_util as gu


def _files(tmp_path):
    G = gu.load("chkpt")
    pa, pl = str(tmp_path / "A.chkpt"), str(tmp_path / "L.chkpt")
    open(pa, "wb").write(bytes(G["chkpt_A"])); open(pl, "wb").write(bytes(G["chkpt_L"]))
    return G, pa, pl


def _no_stamp(b):
    return [l for l in b.split(b"\n") if not (l.startswith(b"## Generated on") or l.startswith(b"## by"))]


def test_binary_checkpoint_reads_back_the_reference_memory_exactly(tmp_path):
    """read_header/read_atoms for format 'L' (imd_io.c:3426, imd_io_3d.c:273-290): a checkpoint is written after
    move_atoms of step 10 (imd_main_3d.c:701-705), so positions and momenta are those the reference holds in
    calc_forces(11); Epot and eam_rho are those of step 10.  All values bit-exact (velocity = impuls / masse)."""
    G, _, pl = _files(tmp_path)
    c = config_io.read_config(pl, ntypes=1)
    h = c["header"]
    assert (h["format"], h["n_items"], h["n_data"]) == ("L", 11, 2) and h["contents"][-2:] == ["Epot", "eam_rho"]
    assert np.array_equal(h["box"], G["box"]) and h["pbc"] == [1, 1, 1]
    o, g11, g10 = np.argsort(c["number"]), np.argsort(G["s11_nummer"]), np.argsort(G["s10_nummer"])
    assert np.array_equal(c["number"][o], G["s11_nummer"][g11])
    assert np.array_equal(c["mass"][o], G["s11_masse"][g11])
    box = np.diag(G["box"])
    dx = c["pos"][o] - G["s11_ort"][g11]
    assert np.abs(dx - np.round(dx / box) * box).max() == 0.0        # same atom image or an exact box shift
    assert np.array_equal(c["vel"][o], G["s11_impuls"][g11] / G["s11_masse"][g11][:, None])
    assert np.array_equal(c["data"]["Epot"][o], G["s10_pot_eng"][g10])
    assert np.array_equal(c["data"]["eam_rho"][o], G["s10_eam_rho"][g10])


def test_ascii_checkpoint(tmp_path):
    """format 'A': the same state through printf("%f") -- six decimals."""
    G, pa, pl = _files(tmp_path)
    a, b = config_io.read_config(pa, ntypes=1), config_io.read_config(pl, ntypes=1)
    assert a["header"]["format"] == "A" and np.array_equal(a["number"], b["number"]) and np.array_equal(a["vtype"], b["vtype"])
    for x, y in ((a["mass"], b["mass"]), (a["pos"], b["pos"]), (a["vel"], b["vel"]), (a["data"]["Epot"], b["data"]["Epot"]),
                 (a["data"]["eam_rho"], b["data"]["eam_rho"])):
        assert np.abs(x - y).max() <= 0.5e-6 * (1 + 1e-9)


def test_writer_reproduces_the_reference_files_byte_for_byte(tmp_path):
    """write_header_config + write_atoms_config (imd_io.c:3196, imd_io_3d.c:721): from the state of the binary file,
    the ASCII and the binary writer give the reference's files byte for byte (the two lines with date and program
    name aside)."""
    G, pa, pl = _files(tmp_path)
    c = config_io.read_config(pl, ntypes=1)
    for binary, ref in ((False, bytes(G["chkpt_A"])), (True, bytes(G["chkpt_L"]))):
        p = str(tmp_path / "w.chkpt")
        config_io.write_config(p, c["number"], c["vtype"], c["mass"], c["pos"], c["impuls"], c["data"]["Epot"],
                               c["header"]["box"], c["header"]["pbc"], eam_rho=c["data"]["eam_rho"], binary=binary)
        mine = open(p, "rb").read()
        if binary:          # header lines differ in length: compare header lines and the record block separately
            hm, hr = mine.index(b"#E\n") + 3, ref.index(b"#E\n") + 3
            assert _no_stamp(mine[:hm]) == _no_stamp(ref[:hr]) and mine[hm:] == ref[hr:]
        else:
            assert _no_stamp(mine) == _no_stamp(ref)


def test_reference_reads_what_the_writer_wrote(tmp_path):
    """The other direction: the fixture holds the binary file write_config produced when the fixture was made and the
    reference's memory after reading it as `coordname` with box_from_header: the writer still produces that file, and
    the reference held exactly the written positions, masses and box, momenta = velocity * mass (read_atoms :391-397)."""
    G, _, pl = _files(tmp_path)
    c = config_io.read_config(pl, ntypes=1)
    p = str(tmp_path / "mine.chkpt")
    config_io.write_config(p, c["number"], c["vtype"], c["mass"], c["pos"], c["impuls"], c["data"]["Epot"],
                           c["header"]["box"], c["header"]["pbc"], eam_rho=c["data"]["eam_rho"], binary=True, stamp="(fixture)")
    assert open(p, "rb").read() == bytes(G["mine_L"])
    o, g = np.argsort(c["number"]), np.argsort(G["reread_nummer"])
    assert np.array_equal(c["number"][o], G["reread_nummer"][g])
    box = np.diag(G["box"])
    dx = G["reread_ort"][g] - c["pos"][o]
    assert np.abs(dx - np.round(dx / box) * box).max() < 4e-15       # back_into_box may add/subtract a box row
    assert np.array_equal(G["reread_masse"][g], c["mass"][o])
    assert np.array_equal(G["reread_impuls"][g], c["vel"][o] * c["mass"][o][:, None])
    assert np.array_equal(G["reread_box"], c["header"]["box"])


def test_headerless_and_velocity_free_files(tmp_path):
    """read_atoms' fall-backs: no header = `number type mass x y z vx vy vz` (imd_io_3d.c:189-204); lines without
    velocities are accepted (:311) and mean 'Maxwell start' (vel None); a wrong column count is an error."""
    p = str(tmp_path / "x.conf")
    open(p, "w").write("1 0 26.98 0.5 0.5 0.5 0.1 0.2 0.3\n2 1 26.98 1.5 0.5 0.5 0 0 0\n\n")
    c = config_io.read_config(p, ntypes=1)
    assert c["header"]["format"] is None and c["type"].tolist() == [0, 0] and c["vtype"].tolist() == [0, 1]
    assert np.allclose(c["impuls"][0], np.array([0.1, 0.2, 0.3]) * 26.98)
    open(p, "w").write("#F A 1 1 0 3 3 0\n#C number type x y z vx vy vz\n#E\n1 0 0.5 0.5 0.5\n2 0 1.5 0.5 0.5\n")
    c = config_io.read_config(p, ntypes=1, masses=[26.98])
    assert c["vel"] is None and c["mass"].tolist() == [26.98, 26.98] and c["pos"][1, 0] == 1.5
    open(p, "w").write("#F A 1 1 1 3 3 0\n#E\n1 0 26.98 0.5 0.5\n")
    try:
        config_io.read_config(p)
        assert False
    except ValueError as e:
        assert "incorrect line" in str(e)


def test_ttm_file_writer_and_reader(tmp_path):
    """ttm_writeout / ttm_read (imd_ttm_3D.c:1836-2118, :2127-2330): from the lattice the reference held after calc_ttm
    of step 2, write_ttm gives the file the reference wrote byte for byte; read_ttm returns those values at printf's
    %e / %f precision."""
    from imd_amd import ttm_io
    G = gu.load("ttm_file")
    lat = {k[4:]: G[k] for k in G.files if k.startswith("lat_")}
    p = str(tmp_path / "out.2.ttm")
    ttm_io.write_ttm(p, lat, proc=0)
    assert open(p, "rb").read() == bytes(G["ttm_file"])
    r = ttm_io.read_ttm(p)
    inner = (slice(1, -1),) * 3
    assert np.array_equal(r["natoms"], lat["natoms"][inner])
    for f in ("temp", "md_temp", "U", "xi", "dens", "fd_k", "fd_g", "Z", "vcom"):
        assert np.allclose(r[f], lat[f][inner], rtol=1e-6, atol=0), f
    assert np.abs(r["Ce"] - lat["Ce"][inner]).max() <= 0.5e-6
