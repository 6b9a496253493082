/*
 * imd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the force + TTM hot path of IMD/lasMD
 * (fmqeisfeld/IMD), used only as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The shipped
 * library (imd_amd/csrc, include/imdgpu.h) never includes, links or calls
 * anything in this directory.
 *
 * Parity status: PINNED for cells / Verlet list / pair + EAM forces / energies
 * / NVE integrator against the real reference binary (oracle/_ref, built by
 * oracle/build_ref.sh from /root/reference) through the fixtures in
 * tests/golden/ (tests/test_oracle_vs_reference.py).  TTM parts: see
 * oracle/ttm_oracle.h.
 *
 * Data model: "padded cells".  The local cell array has dim[0]*dim[1]*dim[2]
 * cells including the one-cell ghost layer (reference BUFCELLS layout,
 * imd_geom_3d.c:253-255); cell (i,j,k) has linear index (i*dim[1]+j)*dim[2]+k
 * (imd_geom_3d.c:898-900); every cell owns `cap` slots, atom (c,s) lives at
 * index c*cap+s of each per-atom array.
 */
#ifndef IMD_ORACLE_H
#define IMD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* restatement of pot_table_t (reference types.h:537-549) */
typedef struct {
  int     ncols, maxsteps;
  double *begin, *end, *step, *invstep;
  int    *len;
  double *table;             /* table[k*ncols+col], maxsteps+2 rows */
} orc_table;

typedef struct {
  int dim[3];                /* local cell array incl. ghost layer     */
  int gdim[3];               /* global cell array (== dim-2 for 1 rank)*/
  int pbc[3];
  int ntypes, cap, nall, ninner;
  double box[9];             /* rows: box_x, box_y, box_z              */
  double tbox[9];            /* rows: tbox_x, tbox_y, tbox_z           */
  double cellsz;             /* squared list radius (r_c+margin)^2     */
  double nbl_margin, timestep;
  int    *n;                 /* [nall] atoms per cell                  */
  int    *inner;             /* [ninner] reference `cells` list        */
  int    *nq;                /* [ninner*14] reference cnbrs[].nq       */
  double *ort, *impuls, *kraft, *nbl_pos;       /* [nall*cap*3] */
  double *masse, *pot_eng, *rho, *dF;           /* [nall*cap]   */
  int    *sorte, *nummer, *numneighs;           /* [nall*cap]   */
  int    *cl_off;            /* [nall+1]                               */
  int    *tl, *tb;           /* Verlet list, reference numbering       */
  int     tl_cap, tb_cap, tb_len;
  int     have_valid_nbl, nbl_count;
  orc_table pair, rhotab, embed;
  int     have_pair, have_eam, count_ttm_neigh;
  double  tot_pot_energy, virial, tot_kin_energy;
  int     is_short;
  double  restrictions[3];   /* single vsorte: (1,1,1)                 */
  int    *fdi, *fdj, *fdk;   /* [nall*cap] global FD cell of the atom at the last update_fd (cell.fdi.., imd_alloc.c:645) */
} orc_sys;

orc_sys *orc_new(const int dim[3], int cap, int ntypes, const double box[9],
                 const int pbc[3], double cellsz, double nbl_margin,
                 double timestep);
void orc_free(orc_sys *s);
void orc_set_table(orc_sys *s, int which /*0 pair,1 rho,2 embed*/, int ncols,
                   int maxsteps, const double *begin, const double *end,
                   const double *step, const double *invstep, const int *len,
                   const double *table);

/* table evaluation, reference potaccess.h:323-354,465-495,591-621 */
void orc_pair_int(const orc_table *pt, int col, int inc, double r2,
                  double *pot, double *grad, int *is_short);
double orc_val_func(const orc_table *pt, int col, int inc, double r2, int *is_short);
double orc_deriv_func(const orc_table *pt, int col, int inc, double r2, int *is_short);
/* end padding, reference imd_potential.c:1260-1276 */
void orc_init_threepoint(double *table, int ncols, const int *len);

/* geometry */
void orc_global_cell_dim(const double box[9], double cellsz, const int cd[3], int out[3]);
void orc_make_cell_lists(orc_sys *s);
void orc_cell_coord(const orc_sys *s, double x, double y, double z, int out[3]);
int  orc_insert_atom(orc_sys *s, int nummer, int sorte, double masse,
                     const double pos[3], const double imp[3]);

/* hot path */
void orc_do_boundaries(orc_sys *s);
void orc_fix_cells(orc_sys *s);
void orc_send_cells(orc_sys *s, int what /*0 positions, 1 dF*/, int full_shell);
void orc_send_forces(orc_sys *s, int what /*0 rho, 1 forces*/);
void orc_make_nblist(orc_sys *s);
void orc_calc_forces(orc_sys *s, int steps);
void orc_check_nblist(orc_sys *s);
void orc_move_atoms_nve(orc_sys *s);
void orc_set_nblist(orc_sys *s, const int *tl, int ntl, const int *tb, int ntb);

#ifdef __cplusplus
}
#endif
#endif
