#!/bin/bash
# build_ref.sh -- TEST INFRASTRUCTURE.  Compiles the *real* reference (IMD/lasMD)
# force path from its own C sources where they lie under $REF (default
# /root/reference) with plain gcc -- the reference's Makefile is not run; the
# source list and -D flags below are what its target-name rules
# (Makefile:141-160,283-296,1000-1050) select for the named target.  Nothing is
# copied: objects and binaries go to oracle/_ref/ only (git-ignored, but they
# travel to the GPU box).  The only non-reference TU linked in is
# oracle/ref_probe.c, a read-only observation probe attached with ld --wrap.
#
# usage: oracle/build_ref.sh [target ...]
#   targets: eam_nve_nbl  nve_nbl  eam_nve  nve  mpi_eam_nve_nbl  mpi_eam_ttm3d_nbl  mpi_eam_ttm3d_laser_nbl  mpi_eam_ttm1d_nbl  mpi_eam_ttm1d_tmm_nbl  mpi_eam_ttm1d_tmm_ep_nbl  fdtd_probe  tricub_probe  tri_probe  dropin
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${IMD_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
CC="${CC:-gcc}"
# -fcommon: globals.h defines a few structs without EXTERN (GCC >= 10 needs it).
# No -march=native / -ffast-math: parity is defined against plain IEEE fp64.
BASEFLAGS="-O2 -fcommon -w -I$REF"

if [ ! -d "$REF" ]; then
  echo "build_ref: $REF not present (GPU box?) -- using prebuilt oracle/_ref" >&2
  exit 0
fi
mkdir -p "$OUT"

COMMON="imd_maxwell.c imd_integrate.c imd_misc.c imd_param.c imd_alloc.c imd_io.c
        imd_io_3d.c imd_potential.c imd_time.c imd_generate.c imd_distrib.c
        imd_main_3d.c imd_geom_3d.c imd_pictures_3d.c imd.c"
NBLSRC="imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c imd_forces_nbl.c"

build() {
  local name="$1" defs="$2" srcs="$3" cc="$4" libs="$5"
  local obj="$OUT/obj_$name"
  mkdir -p "$obj"
  local objs=""
  for s in $srcs; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then
      $cc $BASEFLAGS $defs -c "$REF/$s" -o "$o" &
    fi
    objs="$objs $o"
  done
  wait
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  local wrap="-Wl,--wrap=calc_forces" extra=""
  if [[ "$defs" == *-DTTM1D* ]]; then
    $cc $BASEFLAGS $defs -c "$HERE/ref_probe_ttm1d.c" -o "$obj/ref_probe_ttm1d.o"
    extra="$obj/ref_probe_ttm1d.o"
    if [[ "$srcs" == *imd_tmm.c* ]]; then wrap="$wrap -Wl,--wrap=calc_ttm"          # the reference's own tmm_init / do_tmm
    else wrap="$wrap -Wl,--wrap=calc_ttm -Wl,--wrap=tmm_init -Wl,--wrap=do_tmm"; fi
  elif [[ "$defs" == *-DTTM* ]]; then
    $cc $BASEFLAGS $defs -c "$HERE/ref_probe_ttm.c" -o "$obj/ref_probe_ttm.o"
    extra="$obj/ref_probe_ttm.o"
    wrap="$wrap -Wl,--wrap=calc_ttm -Wl,--wrap=tmm_init -Wl,--wrap=do_tmm"
  fi
  $cc -o "$OUT/imd_$name" $objs "$obj/ref_probe.o" $extra $wrap \
      "$REF/nn_interpol/libnn.a" -lm $libs
  echo "built $OUT/imd_$name"
}

TARGETS="${*:-eam_nve_nbl nve_nbl eam_nve mpi_eam_nve_nbl mpi_eam_ttm3d_nbl mpi_eam_ttm3d_laser_nbl mpi_eam_ttm1d_nbl mpi_eam_ttm1d_tmm_nbl mpi_eam_ttm1d_tmm_ep_nbl fdtd_probe tricub_probe tri_probe dropin}"
for t in $TARGETS; do
  case "$t" in
    eam_nve_nbl) build "$t" "-DNBL -DEAM2 -DNVE" "$COMMON $NBLSRC" "$CC" "" ;;
    nve_nbl)     build "$t" "-DNBL -DNVE"        "$COMMON $NBLSRC" "$CC" "" ;;
    eam_nve)     build "$t" "-DEAM2 -DNVE" "$COMMON imd_main_risc_3d.c imd_forces.c imd_forces_eam2.c" "$CC" "" ;;
    nve)         build "$t" "-DNVE"        "$COMMON imd_main_risc_3d.c imd_forces.c" "$CC" "" ;;
    mpi_eam_nve_nbl)
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DNVE" "$COMMON $NBLSRC imd_mpi_util.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    mpi_eam_ttm3d_nbl)
      # 3D TTM (imd_ttm_3D.c swapped in for imd_ttm.c as the reference's manual says);
      # needs MPI, -DTMM (see ref_probe_ttm.c) and -DEOSMODE=1 (only imd_ttm.c defines it).
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1" \
            "$COMMON $NBLSRC imd_mpi_util.c imd_ttm_3D.c imd_interpol.c fminbnd3.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    mpi_eam_ttm1d_nbl)
      # the 1D TTM the reference's Makefile builds (imd_ttm.c with -DTTM1D: position-binned slabs, RHOMIN, virtual lattice);
      # do_tmm / tmm_init answered by oracle/ref_probe_ttm1d.c (the TMM solver's input files are not in the tree)
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DTTM -DTMM -DTTM1D" \
            "$COMMON $NBLSRC imd_mpi_util.c imd_ttm.c imd_interpol.c fminbnd3.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    mpi_eam_ttm1d_tmm_nbl)
      # the same with the reference's own transfer-matrix Helmholtz solver as laser source (imd_tmm.c: tmm_init, do_tmm,
      # tmm_get_epsilon): the configuration the reference's Makefile target imd_mpi_eam_ttm_tmm_nbl describes.  Its input
      # files (../alu_eps_bb.dat, ../K12.dat) are not in the tree: the fixtures write synthetic ones (tests/golden/make_golden.py)
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DTTM -DTMM -DTTM1D" \
            "$COMMON $NBLSRC imd_mpi_util.c imd_ttm.c imd_interpol.c fminbnd3.c imd_tmm.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    mpi_eam_ttm1d_tmm_ep_nbl)
      # the same with -DELECPRESS (commented out in imd_ttm.c:60 and imd_integrate.c:25: the command line defines it for both):
      # do_electronic_pressure (imd_ttm.c:553-623) and the extra force of move_atoms_ttm (imd_integrate.c:657-743)
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DTTM -DTMM -DTTM1D -DELECPRESS" \
            "$COMMON $NBLSRC imd_mpi_util.c imd_ttm.c imd_interpol.c fminbnd3.c imd_tmm.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    mpi_eam_ttm3d_laser_nbl)
      # the same with -DLASER (imd_laser.c: Beer-Lambert source laser_rescale_ttm, init_laser, get_surface)
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      build "$t" "-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1 -DLASER" \
            "$COMMON $NBLSRC imd_mpi_util.c imd_ttm_3D.c imd_interpol.c fminbnd3.c imd_laser.c" \
            "$CC -I$MPIINC/include" "-L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi" ;;
    fdtd_probe)
      # the reference's 1D Maxwell solver at function level (see oracle/ref_probe_fdtd.c): imd_fdtd.c + imd_misc.c
      # (error()) from the reference, the harness as MAIN; imd_ttm_3D.c does not compile with -DFDTD as shipped
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      obj="$OUT/obj_fdtd_probe"; mkdir -p "$obj"
      FD="-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1 -DFDTD"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$REF/imd_fdtd.c" -o "$obj/imd_fdtd.o"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$REF/imd_misc.c" -o "$obj/imd_misc.o"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$HERE/ref_probe_fdtd.c" -o "$obj/ref_probe_fdtd.o"
      $CC -o "$OUT/ref_probe_fdtd" "$obj/ref_probe_fdtd.o" "$obj/imd_fdtd.o" "$obj/imd_misc.o" \
          -L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi -lm
      echo "built $OUT/ref_probe_fdtd" ;;
    tricub_probe)
      # the reference's tricubic table look-up behind fitDL at function level (see oracle/ref_probe_tricub.c):
      # imd_interpol.c + imd_misc.c (error()) from the reference and the libnn.a it ships, the harness as MAIN
      MPIINC="${MPI_HOME:-/opt/conda}"
      if [ ! -f "$MPIINC/include/mpi.h" ]; then echo "no MPI at $MPIINC, skipping $t" >&2; continue; fi
      obj="$OUT/obj_tricub_probe"; mkdir -p "$obj"
      FD="-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1 -DFDTD"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$REF/imd_interpol.c" -o "$obj/imd_interpol.o"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$REF/imd_misc.c" -o "$obj/imd_misc.o"
      $CC -I$MPIINC/include $BASEFLAGS $FD -c "$HERE/ref_probe_tricub.c" -o "$obj/ref_probe_tricub.o"
      $CC -o "$OUT/ref_probe_tricub" "$obj/ref_probe_tricub.o" "$obj/imd_interpol.o" "$obj/imd_misc.o" \
          "$REF/nn_interpol/libnn.a" -L$MPIINC/lib -Wl,-rpath,$MPIINC/lib -lmpi -lm
      echo "built $OUT/ref_probe_tricub" ;;
    tri_probe)
      # the reference's triangulator (nn-c inside nn_interpol/libnn.a) asked directly which diagonal splits every cell of a grid
      $CC -O1 -w -I"$REF/nn_interpol" -o "$OUT/ref_probe_tri" "$HERE/ref_probe_tri.c" "$REF/nn_interpol/libnn.a" -lm
      echo "built $OUT/ref_probe_tri" ;;
    dropin) ;;    # built below
    *) echo "unknown target $t" >&2; exit 1 ;;
  esac
done

# ---------------------------------------------------------------------------------------------
# Drop-in integration build (TEST INFRASTRUCTURE): the real reference program with its
# imd_forces_nbl.o REPLACED by imd_amd/host/imd_shim.c + libimdgpu.so.  Proves that the library
# links in place of the reference's force object and that the unmodified IMD main loop
# (integrator, fix_cells, I/O) runs on top of the GPU forces.  Output: oracle/_ref/imd_eam_nve_nbl_gpu
# (needs the GPU box to run; tests/test_gpu_dropin.py compares it with oracle/_ref/imd_eam_nve_nbl).
# ---------------------------------------------------------------------------------------------
build_dropin() {
  local name="eam_nve_nbl_gpu" defs="-DNBL -DEAM2 -DNVE"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $CC $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  wait
  $CC $BASEFLAGS $defs -DIMD_SHIM_WRAP_WRITERS -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $CC $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  # --wrap=write_config...: level 2 hands the device's atoms to every writer of the reference (imd_shim.c: IMD_SHIM_WRAP_WRITERS)
  $CC -o "$OUT/imd_$name" $objs "$obj/imd_shim.o" "$obj/ref_probe.o" -Wl,--wrap=calc_forces \
      -Wl,--wrap=write_config -Wl,--wrap=write_config_select -Wl,--wrap=write_distrib -Wl,--wrap=write_pictures \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin ;; esac

# The same for the cell-pair force loop (no -DNBL): calc_forces of imd_main_risc_3d.c is renamed at compile time
# (-Dcalc_forces=calc_forces_host, no source change) so that the shim's calc_forces links; do_forces,
# do_forces_eam2 and do_embedding_energy (imd_forces.c, imd_forces_eam2.c) stay linked only because the renamed,
# never-called host loop references them.
build_dropin_cellpair() {
  local name="eam_nve_gpu" defs="-DEAM2 -DNVE"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_forces.c imd_forces_eam2.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $CC $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  $CC $BASEFLAGS $defs -Dcalc_forces=calc_forces_host -c "$REF/imd_main_risc_3d.c" -o "$obj/imd_main_risc_3d.o" &
  wait
  $CC $BASEFLAGS $defs -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $CC $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $CC -o "$OUT/imd_$name" $objs "$obj/imd_main_risc_3d.o" "$obj/imd_shim.o" "$obj/ref_probe.o" -Wl,--wrap=calc_forces \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_cellpair ;; esac

# The cell-pair program with do_forces / do_forces_eam2 / do_embedding_energy forwarded ONE BY ONE: the reference's own
# calc_forces (imd_main_risc_3d.c, unchanged) walks the cell pairs and calls the shim's versions of these names
# (-DIMD_SHIM_CELLPAIR_FUNCTIONS; imd_forces.c and imd_forces_eam2.c are NOT linked; the shim's whole-step calc_forces is
# compiled under another name).  tests/test_gpu_dropin.py compares it with oracle/_ref/imd_eam_nve.
build_dropin_cellpair_fn() {
  local name="eam_nve_gpufn" defs="-DEAM2 -DNVE"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_main_risc_3d.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $CC $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  wait
  $CC $BASEFLAGS $defs -DIMD_SHIM_CELLPAIR_FUNCTIONS -Dcalc_forces=calc_forces_shim_whole_step -fopenmp -I"$repo/include" \
      -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $CC $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $CC -o "$OUT/imd_$name" $objs "$obj/imd_shim.o" "$obj/ref_probe.o" -Wl,--wrap=calc_forces \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_cellpair_fn ;; esac

# The 3D TTM program on the device: forces through the shim as above AND calc_ttm of the shim (imd_ttm_3D.c compiled
# with -Dcalc_ttm=calc_ttm_host, no source change; its init_ttm, ttm_writeout, ... stay).  The probe only adds the
# lattice dump (-DPROBE_TTM_DROPIN) and the two TMM entry points the 3D file needs (see ref_probe_ttm.c).
# tests/test_gpu_dropin.py compares it with oracle/_ref/imd_mpi_eam_ttm3d_nbl.
build_dropin_ttm() {
  local name="mpi_eam_ttm3d_nbl_gpu" defs="-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  local MPIINC="${MPI_HOME:-/opt/conda}"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  [ -f "$MPIINC/include/mpi.h" ] || { echo "no MPI at $MPIINC, skipping $name" >&2; return 0; }
  local cc="$CC -I$MPIINC/include"
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c imd_mpi_util.c imd_interpol.c fminbnd3.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $cc $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  $cc $BASEFLAGS $defs -Dcalc_ttm=calc_ttm_host -c "$REF/imd_ttm_3D.c" -o "$obj/imd_ttm_3D.o" &
  wait
  $cc $BASEFLAGS $defs -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $cc $BASEFLAGS $defs -DPROBE_TTM_DROPIN -c "$HERE/ref_probe_ttm.c" -o "$obj/ref_probe_ttm.o"
  $cc -o "$OUT/imd_$name" $objs "$obj/imd_ttm_3D.o" "$obj/imd_shim.o" "$obj/ref_probe.o" "$obj/ref_probe_ttm.o" \
      -Wl,--wrap=calc_forces -Wl,--wrap=calc_ttm -Wl,--wrap=tmm_init -Wl,--wrap=do_tmm \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp \
      -Wl,-rpath-link,/usr/lib/x86_64-linux-gnu "$MPIINC/lib/libmpi.so" -Wl,-rpath,"$MPIINC/lib" -lm
  # (libmpi by path and the system directory first for indirect dependencies: the MPI prefix carries an older
  #  libstdc++ than the HIP runtime needs; at run time libimdgpu.so is listed, hence loaded, before libmpi)
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_ttm ;; esac

# The 3D TTM program with the lattice functions forwarded ONE BY ONE (imd_shim.c, -DIMD_SHIM_TTM_FUNCTIONS): imd_ttm_3D.c is
# compiled with its update_fd, do_ADV, do_cell_activation, do_FILLMESH, ttm_fill_ghost_layers, do_DIFF, CFL_maxdt renamed
# (-Dname=name_host, no source change), so the probe's calc_ttm sequence (ref_probe_ttm.c: the reference's own call order,
# function by function) calls the shim's versions of exactly these names.  tests/test_gpu_dropin.py compares it with the CPU
# build, whose probe drives the reference's own functions in the same order.
# The reference's own Makefile configuration -- imd_ttm.c's 1D model with the transfer-matrix laser source, here also with
# -DELECPRESS -- with calc_forces / calc_ttm on the GPU: imd_ttm.c compiled with -Dcalc_ttm=calc_ttm_host, its
# ttm_fill_ghost_layers weakened in the object file (no source change), the shim provides both (imd_shim.c, TTM1D block).
build_dropin_ttm1d() {
  local name="mpi_eam_ttm1d_tmm_ep_nbl_gpu" defs="-DMPI -DNBL -DEAM2 -DTTM -DTMM -DTTM1D -DELECPRESS"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  local MPIINC="${MPI_HOME:-/opt/conda}"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  [ -f "$MPIINC/include/mpi.h" ] || { echo "no MPI at $MPIINC, skipping $name" >&2; return 0; }
  local cc="$CC -I$MPIINC/include"
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c imd_mpi_util.c imd_interpol.c fminbnd3.c imd_tmm.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $cc $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  # calc_ttm: renamed by -D (its only caller is the main loop).  ttm_fill_ghost_layers is also called from INSIDE imd_ttm.c
  # (init_ttm, :1193-1199): compiled position-independent, so that those calls bind at link time, and its definition made weak
  # -- the shim's definition is the one every call reaches
  $cc $BASEFLAGS $defs -fPIC -Dcalc_ttm=calc_ttm_host -c "$REF/imd_ttm.c" -o "$obj/imd_ttm.o" &
  wait
  objcopy --weaken-symbol=ttm_fill_ghost_layers "$obj/imd_ttm.o"
  $cc $BASEFLAGS $defs -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $cc -o "$OUT/imd_$name" $objs "$obj/imd_ttm.o" "$obj/imd_shim.o" "$obj/ref_probe.o" -Wl,--wrap=calc_forces \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp \
      -Wl,-rpath-link,/usr/lib/x86_64-linux-gnu "$MPIINC/lib/libmpi.so" -Wl,-rpath,"$MPIINC/lib" -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_ttm1d ;; esac

build_dropin_ttm_fn() {
  local name="mpi_eam_ttm3d_nbl_gpufn" defs="-DMPI -DNBL -DEAM2 -DTTM -DTMM -DEOSMODE=1"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  local MPIINC="${MPI_HOME:-/opt/conda}"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  [ -f "$MPIINC/include/mpi.h" ] || { echo "no MPI at $MPIINC, skipping $name" >&2; return 0; }
  local cc="$CC -I$MPIINC/include"
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c imd_mpi_util.c imd_interpol.c fminbnd3.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $cc $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  local ren="-Dcalc_ttm=calc_ttm_host -Dupdate_fd=update_fd_host -Ddo_ADV=do_ADV_host -Ddo_cell_activation=do_cell_activation_host"
  ren="$ren -Ddo_FILLMESH=do_FILLMESH_host -Dttm_fill_ghost_layers=ttm_fill_ghost_layers_host -Ddo_DIFF=do_DIFF_host -DCFL_maxdt=CFL_maxdt_host"
  $cc $BASEFLAGS $defs $ren -c "$REF/imd_ttm_3D.c" -o "$obj/imd_ttm_3D.o" &
  wait
  $cc $BASEFLAGS $defs -DIMD_SHIM_TTM_FUNCTIONS -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe_ttm.c" -o "$obj/ref_probe_ttm.o"
  $cc -o "$OUT/imd_$name" $objs "$obj/imd_ttm_3D.o" "$obj/imd_shim.o" "$obj/ref_probe.o" "$obj/ref_probe_ttm.o" \
      -Wl,--wrap=calc_forces -Wl,--wrap=calc_ttm -Wl,--wrap=tmm_init -Wl,--wrap=do_tmm \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp \
      -Wl,-rpath-link,/usr/lib/x86_64-linux-gnu "$MPIINC/lib/libmpi.so" -Wl,-rpath,"$MPIINC/lib" -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_ttm_fn ;; esac

# The MPI force program on several ranks with the shim: the reference keeps its own decomposition, migration and halo
# exchange (send_cells inside the shim's calc_forces where imd_forces_nbl.c has them); every rank drives a GPU.
# tests/test_gpu_dropin.py runs it with mpiexec -n 2 beside oracle/_ref/imd_mpi_eam_nve_nbl.
build_dropin_mpi() {
  local name="mpi_eam_nve_nbl_gpu" defs="-DMPI -DNBL -DEAM2 -DNVE"
  local obj="$OUT/obj_$name" repo="$(cd "$HERE/.." && pwd)"
  local MPIINC="${MPI_HOME:-/opt/conda}"
  [ -f "$repo/imd_amd/csrc/libimdgpu.so" ] || { echo "libimdgpu.so not built, skipping drop-in" >&2; return 0; }
  [ -f "$MPIINC/include/mpi.h" ] || { echo "no MPI at $MPIINC, skipping $name" >&2; return 0; }
  local cc="$CC -I$MPIINC/include"
  mkdir -p "$obj"
  local objs=""
  for s in $COMMON imd_geom_mpi_3d.c imd_comm_force_3d.c imd_fix_cells_3d.c imd_mpiio.c imd_mpi_util.c; do
    local o="$obj/${s%.c}.o"
    if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $cc $BASEFLAGS $defs -c "$REF/$s" -o "$o" & fi
    objs="$objs $o"
  done
  wait
  $cc $BASEFLAGS $defs -fopenmp -I"$repo/include" -c "$repo/imd_amd/host/imd_shim.c" -o "$obj/imd_shim.o"
  $cc $BASEFLAGS $defs -c "$HERE/ref_probe.c" -o "$obj/ref_probe.o"
  $cc -o "$OUT/imd_$name" $objs "$obj/imd_shim.o" "$obj/ref_probe.o" -Wl,--wrap=calc_forces \
      "$REF/nn_interpol/libnn.a" -L"$repo/imd_amd/csrc" -limdgpu -Wl,-rpath,/root/repo/imd_amd/csrc -fopenmp \
      -Wl,-rpath-link,/usr/lib/x86_64-linux-gnu "$MPIINC/lib/libmpi.so" -Wl,-rpath,"$MPIINC/lib" -lm
  echo "built $OUT/imd_$name"
}
case " $TARGETS " in *" dropin "*) build_dropin_mpi ;; esac
