"""IMD configuration files (`.chkpt`, the input `coordname` file): reader and writer.

Host-side data format next to the hot path, so that the reference's own inputs and checkpoints
go in and come out unchanged.  Mirrors, for the 3D / DOUBLE / EAM2 build the force path targets:

* read_header          imd_io.c:3426-3560   `#F fmt n_number n_type n_mass n_pos n_vel n_data`, `#C` column names,
                                            `#X/#Y/#Z` box rows, `#E` end of header
* read_atoms           imd_io_3d.c:44-545   ASCII lines `number type [mass] x y z [vx vy vz] data...`; binary records
                                            ('l'/'b' float, 'L'/'B' double; little/big endian): for doubles the first
                                            8-byte word holds the two ints (number, type).  Momenta are velocity * mass
                                            (:391-397); a file without velocities asks for a Maxwell start.
* write_header_config  imd_io.c:3196-3361
* write_atoms_config   imd_io_3d.c:721-975  ASCII with " %f" per value (6 decimals: an ASCII checkpoint is lossy, a binary
                                            one is exact), velocities = impuls / masse, then Epot and (EAM2) eam_rho.

Pinned by tests/test_config_io.py against files written and read by the reference binary (tests/golden/chkpt.npz)."""
import time

import numpy as np

_BIN = {"l": ("<", 4), "b": (">", 4), "L": ("<", 8), "B": (">", 8)}


def read_header(path):
    """-> dict(format, n_number, n_type, n_mass, n_pos, n_vel, n_data, n_items, contents, box, pbc, data_offset);
    format None when the file has no header (read_atoms then assumes `number type mass x y z vx vy vz`, :189-204)."""
    info = dict(format=None, contents=None, box=None, pbc=None, data_offset=0)
    box = {}
    with open(path, "rb") as f:
        while True:
            at = f.tell()
            line = f.readline()
            if not line.startswith(b"#"):
                info["data_offset"] = at
                break
            tag, rest = line[1:2], line[2:].decode("ascii", "replace")
            if tag == b"F":
                tok = rest.split()
                if len(tok) < 7:
                    raise ValueError("Format line of file %s corrupt" % path)
                n, t, m, npos, nvel, ndat = (int(v) for v in tok[1:7])
                info.update(format=tok[0][0], n_number=n, n_type=t, n_mass=m, n_pos=npos, n_vel=nvel, n_data=ndat,
                            n_items=n + t + m + npos + nvel + ndat)
            elif tag == b"C":
                info["contents"] = rest.split()
            elif tag in (b"X", b"Y", b"Z"):
                box[tag.decode()] = [float(v) for v in rest.split()]
            elif line.startswith(b"##PBC"):
                info["pbc"] = [int(v) for v in line[5:].split()]
            elif tag == b"E":
                info["data_offset"] = f.tell()
                break
    if len(box) == 3:
        info["box"] = np.array([box["X"], box["Y"], box["Z"]])
    if info["format"] is None:
        info.update(format=None, n_number=1, n_type=1, n_mass=1, n_pos=3, n_vel=3, n_data=0, n_items=9, data_offset=0)
    return info


def read_config(path, ntypes=None, masses=None):
    """-> dict(header, number, vtype, type, mass, pos, vel (None without velocities), impuls, data {column: array}).

    `ntypes`: type = vtype mod ntypes as read_atoms does (:318); `masses`: per-type masses for files without a mass
    column (:322).  Positions are returned as stored (the reference then applies back_into_box)."""
    h = read_header(path)
    fmt, ni = h["format"], h["n_items"]
    if fmt in (None, "A"):
        rows = []
        with open(path, "rb") as f:
            f.seek(h["data_offset"])
            for line in f:
                if line.startswith(b"#") or not line.strip():
                    continue
                rows.append(line.split())
        nshort = ni - h["n_vel"]
        for r in rows:
            if len(r) < ni and len(r) != nshort:
                raise ValueError("incorrect line in configuration file.")        # :311-313
        has_vel = h["n_vel"] > 0 and all(len(r) >= ni for r in rows)
        width = ni if has_vel else nshort
        a = np.array([[float(v) for v in r[:width]] for r in rows], dtype=np.float64).reshape(len(rows), width)
        number, vtype, d = a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2:]
    else:
        if fmt not in _BIN:
            raise ValueError("unknown configuration format %r" % fmt)
        end, size = _BIN[fmt]
        raw = np.fromfile(path, dtype=np.uint8, offset=h["data_offset"])
        if size == 8:               # (n_items - 1) words of 8 bytes; word 0 = the two ints
            rec = np.dtype([("i", end + "i4", 2), ("d", end + "f8", ni - 2)])
        else:
            rec = np.dtype([("i", end + "i4", 2), ("d", end + "f4", ni - 2)])
        n = len(raw) // rec.itemsize
        a = raw[: n * rec.itemsize].view(rec)
        number, vtype = a["i"][:, 0].astype(np.int32), a["i"][:, 1].astype(np.int32)
        d = a["d"].astype(np.float64).reshape(n, ni - 2)
        has_vel = h["n_vel"] > 0
    typ = vtype % ntypes if ntypes else vtype.copy()
    c = 0
    if h["n_mass"]:
        mass = d[:, 0].copy(); c = 1
    else:
        if masses is None:
            raise ValueError("configuration file has no mass column: pass masses")
        mass = np.asarray(masses, dtype=np.float64)[typ]
    if (mass <= 0).any():
        raise ValueError("Mass zero or negative.")
    pos = d[:, c:c + 3].copy(); c += h["n_pos"]
    vel = None
    if has_vel:
        vel = d[:, c:c + 3].copy(); c += h["n_vel"]
    names = (h["contents"] or [])[ni - h["n_data"]:] if h["contents"] else []
    data = {}
    for q in range(d.shape[1] - c):
        data[names[q] if q < len(names) else "data%d" % q] = d[:, c + q].copy()
    return dict(header=h, number=number, vtype=vtype, type=typ, mass=mass, pos=pos, vel=vel,
                impuls=None if vel is None else vel * mass[:, None], data=data)


def _header(fmt, box, pbc, with_rho, progname, date, stamp):
    npar = 2 if with_rho else 1
    s = "#F %s 1 1 1 3 3 %d\n" % (fmt, npar)
    s += "#C number type mass x y z vx vy vz Epot" + (" eam_rho" if with_rho else "") + "\n"
    for tag, row in zip("XYZ", np.asarray(box, dtype=np.float64).reshape(3, 3)):
        s += "#%s \t%.16e %.16e %.16e\n" % (tag, row[0], row[1], row[2])
    s += "##PBC %d %d %d\n" % tuple(int(v) for v in pbc)
    s += "## Generated on %s\n" % (stamp if stamp is not None else time.ctime())
    s += "## by %s (version of %s)\n" % (progname, date)
    return s + "#E\n"


def write_config(path, number, vtype, mass, pos, impuls, epot, box, pbc, eam_rho=None, binary=False,
                 progname="imd_amd", date="", stamp=None):
    """Checkpoint in the reference's layout (write_header_config + write_atoms_config): velocities are impuls/masse;
    `binary` writes little-endian double records (format 'L'), else ASCII with printf's "%f"."""
    number = np.asarray(number, dtype=np.int32); vtype = np.asarray(vtype, dtype=np.int32)
    mass = np.asarray(mass, dtype=np.float64); pos = np.asarray(pos, dtype=np.float64)
    vel = np.asarray(impuls, dtype=np.float64) / mass[:, None]
    epot = np.asarray(epot, dtype=np.float64)
    cols = [mass, pos[:, 0], pos[:, 1], pos[:, 2], vel[:, 0], vel[:, 1], vel[:, 2], epot]
    if eam_rho is not None:
        cols.append(np.asarray(eam_rho, dtype=np.float64))
    head = _header("L" if binary else "A", box, pbc, eam_rho is not None, progname, date, stamp).encode("ascii")
    with open(path, "wb") as f:
        f.write(head)
        if binary:
            rec = np.zeros(len(number), dtype=np.dtype([("i", "<i4", 2), ("d", "<f8", len(cols))]))
            rec["i"][:, 0], rec["i"][:, 1] = number, vtype
            rec["d"] = np.stack(cols, axis=1)
            f.write(rec.tobytes())
        else:
            fmt = "%d %d" + " %f" * len(cols) + "\n"
            a = np.stack(cols, axis=1)
            f.write("".join(fmt % ((int(n), int(t)) + tuple(r)) for n, t, r in zip(number, vtype, a)).encode("ascii"))
