// imdgpu_dist.hip -- the multi-GPU MD step of one rank, native: C host code + RCCL over xGMI.
//
// What the reference does per step with MPI (send_cells x 2, Plimpton's three stages each, imd_comm_force_3d.c:222-396;
// MPI_Allreduce of the list validity, imd_forces_nbl.c:1087) is restated for one node of eight fully connected GPUs:
//
//   * DIRECT halo: every rank has a point-to-point xGMI link to every other rank of the node, so the 26 ghost regions
//     (6 faces, 12 edges, 8 corners) are filled by 26 direct messages posted as ONE RCCL group -- one exchange latency per
//     halo instead of three dependent stages, and the faces of different directions travel over different links at the
//     same time.  (Plimpton's staging exists to save message count on a torus; here it would only serialise the links.)
//   * one gather launch (all 26 send lists, periodic shift per segment) -> one ncclGroup of sends/recvs on the context's
//     stream -> one linear unpack launch: the ghost atoms of the 26 regions lie in the order of the receive segments, so
//     the receive buffer maps 1:1 onto the ghost atoms [nlocal, nlocal + nghost).
//   * the list-validity flag stays on the device: ncclAllReduce(MAX) of one int, consumed by the GUARDED pass-1 kernel of
//     the next step, which is enqueued before the host has looked at the flag.  The host reads the flag while that kernel
//     runs; on "invalid" the guarded kernel has done nothing and the step is re-issued after the rebuild.
//   * TTM with the replicated lattice: ncclAllGather of the ranks' update_fd blocks on the lattice stream.
//
// The transport is a small vtable: RCCL (production; librccl is dlopen'ed, so the library has no link-time dependency
// on it) or a host callback (tests: several ranks sharing ONE GPU cannot form an RCCL communicator, so the identical
// step logic is driven with a host-staged exchange; imd_amd/parallel.py:CallbackTransport).
// The rebuild path (migration, counts, index lists: every ~17 steps) stays in the host layer.
#include "imdgpu_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

static inline int nblk(long long n, int b) { return n <= 0 ? 1 : (int) ((n + b - 1) / b); }

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static const char *rccl_load()
{
  if (g_rccl.h) return nullptr;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return "librccl.so not found (dlopen)";
#define SYM(field, name) do { *(void **) (&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return "librccl: symbol " name " missing"; } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.h = h;
  return nullptr;
}

#define MAXSEG 32

struct DistState {
  int nranks = 1, rank = 0;
  ncclComm_t comm = nullptr, comm2 = nullptr;      // comm: halo + flag on the main stream; comm2: TTM all-gather on the lattice stream
  imdgpu_exchange_fn cb = nullptr; void *cb_user = nullptr;
  // plan of the current list (atoms)
  int nsend = 0, nrecv = 0;
  std::vector<int> send_peer, recv_peer;
  std::vector<long long> send_cnt, send_off, recv_cnt, recv_off;
  std::vector<double> send_shift;
  long long nsend_atoms = 0, nrecv_atoms = 0;
  int *d_send_idx = nullptr; size_t send_idx_cap = 0;
  int *d_seg_of = nullptr;                     // per send atom: its segment (for the shift)
  int *d_self_dst = nullptr;                   // per send atom: the ghost atom it becomes on THIS rank (self segments), or -1
  unsigned char *d_recv_self = nullptr; size_t recv_self_cap = 0;   // per ghost atom: filled directly by the gather (self segment)
  long long nself_atoms = 0;
  double *d_shift = nullptr;                   // 3 * MAXSEG
  double *sendbuf = nullptr, *recvbuf = nullptr; size_t sendbuf_cap = 0, recvbuf_cap = 0;   // doubles
  int *d_flag = nullptr;                       // [0] global "list invalid" of the step just integrated
  int *h_flag = nullptr;                       // pinned
  hipEvent_t ev_flag = nullptr;
  bool have_plan = false, halo_ready = false, spec_pass1 = false, self_comm = false;
  int ttm_block_cells = 0;
  double *ttm_blk = nullptr, *ttm_all = nullptr; size_t ttm_blk_cap = 0, ttm_all_cap = 0;
  int *d_ttm_offs = nullptr;
  hipEvent_t ev_gathered = nullptr;
};

static DistState *DS(imdgpu_ctx *ctx) { return (DistState *) ctx->dist; }

void imdgpu_dist_free(imdgpu_ctx *ctx)
{
  DistState *D = DS(ctx);
  if (!D) return;
  if (D->comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(D->comm);
  if (D->comm2 && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(D->comm2);
  void *ptrs[] = {D->d_send_idx, D->d_seg_of, D->d_self_dst, D->d_recv_self, D->d_shift, D->sendbuf, D->recvbuf, D->d_flag, D->ttm_blk, D->ttm_all, D->d_ttm_offs};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  if (D->h_flag) (void) hipHostFree(D->h_flag);
  if (D->ev_flag) (void) hipEventDestroy(D->ev_flag);
  if (D->ev_gathered) (void) hipEventDestroy(D->ev_gathered);
  delete D;
  ctx->dist = nullptr;
}

static int dist_new(imdgpu_ctx *ctx, int nranks, int rank)
{
  if (nranks < 1 || rank < 0 || rank >= nranks) return imdgpu_fail(ctx, "dist_init: bad rank %d of %d", rank, nranks);
  if (!ctx->external_halo) return imdgpu_fail(ctx, "dist_init: the context needs the option external_halo (ghost cells owned by the halo layer)");
  imdgpu_dist_free(ctx);
  DistState *D = new DistState();
  D->nranks = nranks; D->rank = rank;
  ctx->dist = D;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMalloc((void **) &D->d_flag, 4 * sizeof(int)));
  HIPCHK(hipMemset(D->d_flag, 0, 4 * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &D->h_flag, 4 * sizeof(int)));
  HIPCHK(hipMalloc((void **) &D->d_shift, 3 * MAXSEG * sizeof(double)));
  HIPCHK(hipEventCreateWithFlags(&D->ev_flag, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_gathered, hipEventDisableTiming));
  D->self_comm = getenv("IMDGPU_DIST_SELF_COMM") != nullptr;   // rehearsal: self-neighbour segments go through the transport too
  return 0;
}

// rank 0: two RCCL unique ids (2 x 128 bytes), to be broadcast to the other ranks by the launcher (MPI_Bcast, a file, ...)
extern "C" int imdgpu_dist_unique_id(void *out, int nbytes)
{
  const char *e = rccl_load();
  if (e || nbytes < 2 * (int) sizeof(ncclUniqueId)) return 1;
  for (int k = 0; k < 2; k++) {
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return 1;
    memcpy((char *) out + k * sizeof(id), &id, sizeof(id));
  }
  return 0;
}

extern "C" int imdgpu_dist_init_rccl(imdgpu_ctx *ctx, int nranks, int rank, const void *unique_id, int nbytes)
{
  const char *e = rccl_load();
  if (e) return imdgpu_fail(ctx, "dist_init_rccl: %s", e);
  if (nbytes != 2 * (int) sizeof(ncclUniqueId)) return imdgpu_fail(ctx, "dist_init_rccl: unique id must be %d bytes (imdgpu_dist_unique_id)", 2 * (int) sizeof(ncclUniqueId));
  if (dist_new(ctx, nranks, rank)) return 1;
  for (int k = 0; k < 2; k++) {
    ncclUniqueId id;
    memcpy(&id, (const char *) unique_id + k * sizeof(id), sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(k ? &DS(ctx)->comm2 : &DS(ctx)->comm, nranks, id, rank);
    if (r != ncclSuccess) return imdgpu_fail(ctx, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  return 0;
}

extern "C" int imdgpu_dist_init_callback(imdgpu_ctx *ctx, int nranks, int rank, imdgpu_exchange_fn fn, void *user)
{
  if (!fn) return imdgpu_fail(ctx, "dist_init_callback: no function");
  if (dist_new(ctx, nranks, rank)) return 1;
  DS(ctx)->cb = fn; DS(ctx)->cb_user = user;
  return 0;
}

#define NEED_DIST()                                                                            \
  DistState *D = DS(ctx);                                                                      \
  if (!D) return imdgpu_fail(ctx, "imdgpu_dist_init_rccl / _callback first");                 \
  HIPCHK(hipSetDevice(ctx->device))
#define NCCLCHK(call)                                                                          \
  do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return imdgpu_fail(ctx, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); } while (0)

// the plan of one neighbour list: what to send where (atom index lists on the device) and what arrives
extern "C" int imdgpu_dist_set_plan(imdgpu_ctx *ctx, int nsend, const int *send_peer, const int *send_count,
                                    const double *send_shift, const int *dev_send_idx, int nrecv, const int *recv_peer,
                                    const int *recv_count)
{
  NEED_DIST();
  if (nsend > MAXSEG || nrecv > MAXSEG) return imdgpu_fail(ctx, "dist_set_plan: at most %d segments", MAXSEG);
  D->nsend = nsend; D->nrecv = nrecv;
  D->send_peer.assign(send_peer, send_peer + nsend); D->recv_peer.assign(recv_peer, recv_peer + nrecv);
  D->send_cnt.resize(nsend); D->send_off.resize(nsend); D->recv_cnt.resize(nrecv); D->recv_off.resize(nrecv);
  D->send_shift.assign(send_shift, send_shift + 3 * nsend);
  long long a = 0;
  for (int s = 0; s < nsend; s++) { D->send_off[s] = a; D->send_cnt[s] = send_count[s]; a += send_count[s]; }
  D->nsend_atoms = a;
  a = 0;
  for (int s = 0; s < nrecv; s++) { D->recv_off[s] = a; D->recv_cnt[s] = recv_count[s]; a += recv_count[s]; }
  D->nrecv_atoms = a;
  if (D->nrecv_atoms != ctx->nghost) return imdgpu_fail(ctx, "dist_set_plan: %lld atoms arrive, the ghost layout holds %d", D->nrecv_atoms, ctx->nghost);
  // the k-th self segment sent must be the k-th self segment received, with the same length
  {
    std::vector<long long> a1, a2;
    for (int s = 0; s < nsend; s++) if (send_peer[s] == D->rank) a1.push_back(send_count[s]);
    for (int s = 0; s < nrecv; s++) if (recv_peer[s] == D->rank) a2.push_back(recv_count[s]);
    if (a1 != a2) return imdgpu_fail(ctx, "dist_set_plan: self segments of the send and receive lists do not pair up");
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if ((size_t) D->nsend_atoms > D->send_idx_cap) {
    if (D->d_send_idx) (void) hipFree(D->d_send_idx);
    if (D->d_seg_of) (void) hipFree(D->d_seg_of);
    if (D->d_self_dst) (void) hipFree(D->d_self_dst);
    D->send_idx_cap = (size_t) (D->nsend_atoms * 1.2) + 1024;
    HIPCHK(hipMalloc((void **) &D->d_send_idx, D->send_idx_cap * sizeof(int)));
    HIPCHK(hipMalloc((void **) &D->d_seg_of, D->send_idx_cap * sizeof(int)));
    HIPCHK(hipMalloc((void **) &D->d_self_dst, D->send_idx_cap * sizeof(int)));
  }
  if ((size_t) D->nrecv_atoms + 1 > D->recv_self_cap) {
    if (D->d_recv_self) (void) hipFree(D->d_recv_self);
    D->recv_self_cap = (size_t) (D->nrecv_atoms * 1.2) + 1024;
    HIPCHK(hipMalloc((void **) &D->d_recv_self, D->recv_self_cap));
  }
  if (D->nsend_atoms > 0) HIPCHK(hipMemcpy(D->d_send_idx, dev_send_idx, D->nsend_atoms * sizeof(int), hipMemcpyDeviceToDevice));
  {
    std::vector<int> seg((size_t) D->nsend_atoms);
    for (int s = 0; s < nsend; s++) for (long long q = 0; q < D->send_cnt[s]; q++) seg[D->send_off[s] + q] = s;
    if (D->nsend_atoms > 0) HIPCHK(hipMemcpy(D->d_seg_of, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice));
    // Segments this rank sends to itself (a periodic direction with one rank: all 26 at one rank, the y/z faces of a 2x1x1 grid)
    // never touch the buffers: the gather kernel writes those atoms, shifted, straight into the ghost range, the k-th self
    // segment sent being the k-th self segment received; unpack skips them.
    std::vector<int> dst((size_t) D->nsend_atoms, -1);
    std::vector<unsigned char> rs((size_t) D->nrecv_atoms + 1, 0);
    D->nself_atoms = 0;
    if (!D->self_comm) {
      std::vector<int> self_recv;
      for (int r = 0; r < nrecv; r++) if (recv_peer[r] == D->rank) self_recv.push_back(r);
      size_t k = 0;
      for (int q = 0; q < nsend; q++) {
        if (send_peer[q] != D->rank) continue;
        const int r = self_recv[k++];
        for (long long a2 = 0; a2 < D->send_cnt[q]; a2++) { dst[D->send_off[q] + a2] = ctx->nlocal + (int) (D->recv_off[r] + a2); rs[D->recv_off[r] + a2] = 1; }
        D->nself_atoms += D->send_cnt[q];
      }
    }
    if (D->nsend_atoms > 0) HIPCHK(hipMemcpy(D->d_self_dst, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D->d_recv_self, rs.data(), rs.size(), hipMemcpyHostToDevice));
    double sh[3 * MAXSEG] = {0};
    for (int s = 0; s < 3 * nsend; s++) sh[s] = send_shift[s];
    HIPCHK(hipMemcpy(D->d_shift, sh, sizeof(sh), hipMemcpyHostToDevice));
  }
  const size_t ns = (size_t) D->nsend_atoms * 4 + 8, nr = (size_t) D->nrecv_atoms * 4 + 8;
  if (ns > D->sendbuf_cap) { if (D->sendbuf) (void) hipFree(D->sendbuf); D->sendbuf_cap = (size_t) (ns * 1.2); HIPCHK(hipMalloc((void **) &D->sendbuf, D->sendbuf_cap * sizeof(double))); }
  if (nr > D->recvbuf_cap) { if (D->recvbuf) (void) hipFree(D->recvbuf); D->recvbuf_cap = (size_t) (nr * 1.2); HIPCHK(hipMalloc((void **) &D->recvbuf, D->recvbuf_cap * sizeof(double))); }
  D->have_plan = true; D->halo_ready = false; D->spec_pass1 = false;
  return 0;
}

// what: 0 = (x,y,z,dF) 4 doubles; 1 = dF; 2 = type (as double: one element width for every message); 3 = x,y,z
static __host__ __device__ __forceinline__ int what_width(int what) { return what == 0 ? 4 : what == 3 ? 3 : 1; }

__global__ void k_dist_gather(int what, long long n, const int *idx, const int *seg_of, const int *self_dst, const double *shift,
                              double4 *pos, int *type, double *out)
{
  const long long m = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const int i = idx[m], g = self_dst[m];                // g >= 0: this atom is its own rank's ghost atom g (nothing to send)
  if (what == 1) { if (g >= 0) pos[g].w = pos[i].w; else out[m] = pos[i].w; return; }
  if (what == 2) { if (g >= 0) type[g] = type[i]; else out[m] = (double) type[i]; return; }
  const double *s = shift + 3 * seg_of[m];
  const double4 r = pos[i];
  if (g >= 0) {
    if (what == 0) pos[g] = make_double4(r.x + s[0], r.y + s[1], r.z + s[2], r.w);
    else { pos[g].x = r.x + s[0]; pos[g].y = r.y + s[1]; pos[g].z = r.z + s[2]; }
    return;
  }
  if (what == 0) { double *o = out + 4 * m; o[0] = r.x + s[0]; o[1] = r.y + s[1]; o[2] = r.z + s[2]; o[3] = r.w; }
  else { double *o = out + 3 * m; o[0] = r.x + s[0]; o[1] = r.y + s[1]; o[2] = r.z + s[2]; }
}
// the receive buffer IS the ghost atoms in order: atom nlocal + m (those of self segments are already in place)
__global__ void k_dist_unpack(int what, long long n, int nlocal, const unsigned char *recv_self, const double *in, double4 *pos, int *type)
{
  const long long m = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n || recv_self[m]) return;
  const long long i = nlocal + m;
  if (what == 1) pos[i].w = in[m];
  else if (what == 2) type[i] = (int) in[m];
  else if (what == 0) { const double *q = in + 4 * m; pos[i] = make_double4(q[0], q[1], q[2], q[3]); }
  else { const double *q = in + 3 * m; pos[i].x = q[0]; pos[i].y = q[1]; pos[i].z = q[2]; }
}
// one exchange of `w` doubles per atom following the plan, on the context's stream
static int dist_exchange(imdgpu_ctx *ctx, DistState *D, int w)
{
  std::vector<long long> so, sc, ro, rc; std::vector<int> sp, rp;
  for (int s = 0; s < D->nsend; s++) {
    if (D->send_cnt[s] == 0 || (D->send_peer[s] == D->rank && !D->self_comm)) continue;      // self segments: done by the gather
    sp.push_back(D->send_peer[s]); so.push_back(D->send_off[s] * w); sc.push_back(D->send_cnt[s] * w);
  }
  for (int s = 0; s < D->nrecv; s++) {
    if (D->recv_cnt[s] == 0 || (D->recv_peer[s] == D->rank && !D->self_comm)) continue;
    rp.push_back(D->recv_peer[s]); ro.push_back(D->recv_off[s] * w); rc.push_back(D->recv_cnt[s] * w);
  }
  if (sp.empty() && rp.empty()) return 0;
  if (D->comm) {
    NCCLCHK(g_rccl.GroupStart());
    for (size_t s = 0; s < sp.size(); s++) NCCLCHK(g_rccl.Send(D->sendbuf + so[s], (size_t) sc[s], ncclDouble, sp[s], D->comm, ctx->stream));
    for (size_t s = 0; s < rp.size(); s++) NCCLCHK(g_rccl.Recv(D->recvbuf + ro[s], (size_t) rc[s], ncclDouble, rp[s], D->comm, ctx->stream));
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));             // host transport (tests): the buffers must be complete
  if (D->cb(D->cb_user, 0, D->sendbuf, D->recvbuf, (int) sp.size(), sp.data(), so.data(), sc.data(), (int) rp.size(), rp.data(),
            ro.data(), rc.data(), 8))
    return imdgpu_fail(ctx, "dist: the host transport failed");
  return 0;
}

extern "C" int imdgpu_dist_halo(imdgpu_ctx *ctx, int what)
{
  NEED_DIST();
  if (!D->have_plan) return imdgpu_fail(ctx, "dist_halo: no plan (imdgpu_dist_set_plan after the ghost layout)");
  if (what < 0 || what > 3) return imdgpu_fail(ctx, "dist_halo: what must be 0..3");
  const int w = what_width(what);
  ProfScope ps(ctx, PROF_GHOSTS);
  if (D->nsend_atoms > 0)
    hipLaunchKernelGGL(k_dist_gather, dim3(nblk(D->nsend_atoms, 256)), dim3(256), 0, ctx->stream, what, D->nsend_atoms, D->d_send_idx,
                       D->d_seg_of, D->d_self_dst, D->d_shift, ctx->pos, ctx->type, D->sendbuf);
  if (dist_exchange(ctx, D, w)) return 1;
  if (D->nrecv_atoms > D->nself_atoms)                    // something did arrive through the buffers
    hipLaunchKernelGGL(k_dist_unpack, dim3(nblk(D->nrecv_atoms, 256)), dim3(256), 0, ctx->stream, what, D->nrecv_atoms, ctx->nlocal,
                       D->d_recv_self, D->recvbuf, ctx->pos, ctx->type);
  HIPCHK(hipGetLastError());
  return 0;
}

// MAX over the ranks of this rank's list-validity flag (check_nblist's MPI_Allreduce, imd_forces_nbl.c:1087) -> d_flag[0]
__global__ void k_dist_flag_in(const int *counters, int *flag) { flag[0] = counters[CNT_INVALID]; }
static int dist_flag_allreduce(imdgpu_ctx *ctx, DistState *D)
{
  hipLaunchKernelGGL(k_dist_flag_in, dim3(1), dim3(1), 0, ctx->stream, ctx->d_counters, D->d_flag);
  if (D->nranks == 1 && !D->self_comm) return 0;
  if (D->comm) { NCCLCHK(g_rccl.AllReduce(D->d_flag, D->d_flag, 1, ncclInt, ncclMax, D->comm, ctx->stream)); return 0; }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (D->cb(D->cb_user, 1, D->d_flag, D->d_flag, 1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 4))
    return imdgpu_fail(ctx, "dist: the host transport failed (all-reduce)");
  return 0;
}

// replicated TTM lattice: FD-cell offset of every rank's block (3 ints per rank, host) -- once
extern "C" int imdgpu_dist_set_ttm_blocks(imdgpu_ctx *ctx, const int *offs)
{
  NEED_DIST();
  if (!D->d_ttm_offs) HIPCHK(hipMalloc((void **) &D->d_ttm_offs, 3 * (size_t) D->nranks * sizeof(int)));
  HIPCHK(hipMemcpy(D->d_ttm_offs, offs, 3 * (size_t) D->nranks * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

// the lattice half of a step on the side stream (forked after pass 1, joined before the integrator)
static int dist_ttm_chain(imdgpu_ctx *ctx, DistState *D, int steps)
{
  HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  hipStream_t main = ctx->stream;
  ctx->stream = ctx->stream2;
  int rc = 0;
  if (ctx->ttm_decomposed) {
    if (ctx->ttm_adv) rc = imdgpu_fail(ctx, "dist_md_step: decomposed lattice with advection is driven by the host layer");
    else rc = imdgpu_ttm_local_step(ctx, steps);
  } else {
    int nb = 0;
    long long width = 0;
    rc = imdgpu_ttm_local_block(ctx, steps, nullptr, &nb) || imdgpu_get_info(ctx, "ttm_block_width", &width);
    const size_t per = (size_t) nb * (size_t) width;
    if (!rc && per > D->ttm_blk_cap) {
      if (D->ttm_blk) (void) hipFree(D->ttm_blk);
      if (D->ttm_all) (void) hipFree(D->ttm_all);
      D->ttm_blk = D->ttm_all = nullptr;
      if (hipMalloc((void **) &D->ttm_blk, per * sizeof(double)) != hipSuccess ||
          hipMalloc((void **) &D->ttm_all, per * D->nranks * sizeof(double)) != hipSuccess) rc = imdgpu_fail(ctx, "dist: out of memory (TTM blocks)");
      D->ttm_blk_cap = per;
    }
    if (!rc && !D->d_ttm_offs) rc = imdgpu_fail(ctx, "dist_md_step: imdgpu_dist_set_ttm_blocks first");
    if (!rc) rc = imdgpu_ttm_local_block(ctx, steps, D->ttm_blk, &nb);
    if (!rc) {
      if (D->nranks == 1 && !D->self_comm) { if (hipMemcpyAsync(D->ttm_all, D->ttm_blk, per * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = 1; }
      else if (D->comm) { if (g_rccl.AllGather(D->ttm_blk, D->ttm_all, per, ncclDouble, D->comm2, ctx->stream) != ncclSuccess) rc = imdgpu_fail(ctx, "ncclAllGather failed"); }
      else {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 1;
        if (!rc && D->cb(D->cb_user, 2, D->ttm_blk, D->ttm_all, (int) per, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 8))
          rc = imdgpu_fail(ctx, "dist: the host transport failed (all-gather)");
      }
    }
    // the EAM force pass is persistent and fills every CU: a collective that is not resident by then would wait for its
    // end and the whole chain with it, so pass 3 waits for the gather's completion (not for the chain)
    if (!rc && hipEventRecord(D->ev_gathered, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) rc = imdgpu_ttm_apply_blocks(ctx, steps, D->ttm_all, D->nranks, D->d_ttm_offs);
    if (!rc && ctx->ttm_adv) rc = imdgpu_ttm_adv(ctx, 1.0, steps);
    if (!rc) rc = imdgpu_ttm_lattice_step(ctx, steps);
  }
  ctx->stream = main;
  if (rc) return 1;
  HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
  return 0;
}

// One MD step of this rank.  *rebuild_needed != 0: some rank's list became invalid DURING this step (the step itself is
// complete and valid); the caller migrates / re-bins / re-plans before the next call.
extern "C" int imdgpu_dist_md_step(imdgpu_ctx *ctx, int steps, int *rebuild_needed)
{
  NEED_DIST();
  if (!D->have_plan || !ctx->have_valid_nbl) return imdgpu_fail(ctx, "dist_md_step: rebuild first (ghost layout, imdgpu_dist_set_plan, imdgpu_make_nblist)");
  imdgpu_tl_mark(ctx, "begin");
  if (!D->spec_pass1) {
    if (!D->halo_ready && imdgpu_dist_halo(ctx, 3)) return 1;
    if (imdgpu_launch_forces_phase(ctx, 1)) return 1;
  }
  D->spec_pass1 = false; D->halo_ready = false;
  if (ctx->ttm && dist_ttm_chain(ctx, D, steps)) return 1;
  if (imdgpu_launch_forces_phase(ctx, 2)) return 1;
  if (ctx->rho.present && imdgpu_dist_halo(ctx, 1)) return 1;          // dF of the ghost atoms  [send_cells(copy_dF), :786]
  if (ctx->ttm && !ctx->ttm_decomposed) HIPCHK(hipStreamWaitEvent(ctx->stream, D->ev_gathered, 0));
  if (imdgpu_launch_forces_phase(ctx, 3)) return 1;
  if (ctx->ttm) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  if (imdgpu_launch_integrate(ctx)) return 1;
  // next step's position halo and the global validity flag; then pass 1 of the next step, guarded by that flag, so the
  // device has work while the host reads the flag
  if (imdgpu_dist_halo(ctx, 3)) return 1;
  if (dist_flag_allreduce(ctx, D)) return 1;
  HIPCHK(hipMemcpyAsync(D->h_flag, D->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipEventRecord(D->ev_flag, ctx->stream));
  const bool spec = !getenv("IMDGPU_NO_SPECULATION");
  if (spec) {
    ctx->pass1_guard = D->d_flag;
    int rc = imdgpu_launch_forces_phase(ctx, 1);
    ctx->pass1_guard = nullptr;
    if (rc) return 1;
  }
  HIPCHK(hipEventSynchronize(D->ev_flag));
  const int invalid = D->h_flag[0];
  if (invalid) {
    ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1; D->have_plan = false;
    if (spec)                       // the guarded launch did nothing: it must not count as a pass-1 launch in the kernel timings
      for (size_t q = ctx->ev_pending.size(); q-- > 0;)
        if (ctx->ev_pending[q].id == PROF_PASS1) {
          ctx->ev_pool.push_back(ctx->ev_pending[q].a); ctx->ev_pool.push_back(ctx->ev_pending[q].b);
          ctx->ev_pending.erase(ctx->ev_pending.begin() + q);
          break;
        }
  }
  else { D->halo_ready = true; D->spec_pass1 = spec; }
  if (rebuild_needed) *rebuild_needed = invalid;
  return 0;
}

// a state change from outside (upload, re-binning, new list) voids what was enqueued ahead
void imdgpu_dist_cancel_ahead(imdgpu_ctx *ctx)
{
  DistState *D = DS(ctx);
  if (D) { D->spec_pass1 = false; D->halo_ready = false; }
}

// sums over the ranks of n doubles on the host (energies; not on the per-step path)
extern "C" int imdgpu_dist_allreduce_sum(imdgpu_ctx *ctx, double *vals, int n)
{
  NEED_DIST();
  if (D->nranks == 1 || n <= 0) return 0;
  if (n > 16) return imdgpu_fail(ctx, "dist_allreduce_sum: at most 16 values");
  double *d = D->sendbuf ? D->sendbuf : nullptr;
  if (!d) return imdgpu_fail(ctx, "dist_allreduce_sum: no plan yet");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(d, vals, n * sizeof(double), hipMemcpyHostToDevice));
  if (D->comm) { NCCLCHK(g_rccl.AllReduce(d, d, (size_t) n, ncclDouble, ncclSum, D->comm, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
  else if (D->cb(D->cb_user, 3, d, d, n, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 8)) return imdgpu_fail(ctx, "dist: the host transport failed (sum)");
  HIPCHK(hipMemcpy(vals, d, n * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
