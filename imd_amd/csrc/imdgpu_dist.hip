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
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
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
  SYM(CommCount, "ncclCommCount"); SYM(CommCuDevice, "ncclCommCuDevice");
#undef SYM
  g_rccl.h = h;
  return nullptr;
}

#define MAXSEG 32

struct DistState {
  int nranks = 1, rank = 0;
  ncclComm_t comm = nullptr, comm2 = nullptr;      // comm: halo + flag on the main stream; comm2: TTM all-gather on the lattice stream
  ncclComm_t comm3 = nullptr;                      // the position halo on its own stream under interior tiles of pass 1 (option halo_overlap)
  hipStream_t stream3 = nullptr; hipEvent_t ev_fork3 = nullptr, ev_halo = nullptr;
  int remote[3]{0, 0, 0};                          // dimension d has a neighbour rank other than this one (or the self-message rehearsal is on)
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
  double *fluxbuf = nullptr; size_t flux_cap = 0;   // do_COMMFLUX: [to lo | to hi | from lo | from hi], 5 doubles per face cell
};

static DistState *DS(imdgpu_ctx *ctx) { return (DistState *) ctx->dist; }

void imdgpu_dist_plan_free(imdgpu_ctx *ctx);
void imdgpu_dist_free(imdgpu_ctx *ctx)
{
  DistState *D = DS(ctx);
  imdgpu_dist_plan_free(ctx);
  if (!D) return;
  if (D->comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(D->comm);
  if (D->comm2 && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(D->comm2);
  if (D->comm3 && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(D->comm3);
  if (D->stream3) (void) hipStreamDestroy(D->stream3);
  if (D->ev_fork3) (void) hipEventDestroy(D->ev_fork3);
  if (D->ev_halo) (void) hipEventDestroy(D->ev_halo);
  void *ptrs[] = {D->d_send_idx, D->d_seg_of, D->d_self_dst, D->d_recv_self, D->d_shift, D->sendbuf, D->recvbuf, D->d_flag, D->ttm_blk, D->ttm_all, D->d_ttm_offs, D->fluxbuf};
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
  HIPCHK(hipMalloc((void **) &D->d_flag, 8 * sizeof(int)));       // [0..3]: the step's flags (dist_flag_allreduce), [4]: scratch of dist_max_int
  HIPCHK(hipMemset(D->d_flag, 0, 8 * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &D->h_flag, 8 * sizeof(int)));
  HIPCHK(hipMalloc((void **) &D->d_shift, 3 * MAXSEG * sizeof(double)));
  HIPCHK(hipEventCreateWithFlags(&D->ev_flag, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_gathered, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_fork3, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&D->ev_halo, hipEventDisableTiming));
  HIPCHK(hipStreamCreateWithFlags(&D->stream3, hipStreamNonBlocking));
  D->self_comm = cfg_env("IMDGPU_DIST_SELF_COMM") != nullptr;   // rehearsal: self-neighbour segments go through the transport too
  return 0;
}

// rank 0: RCCL unique ids of 128 bytes each, as many as the buffer holds (two at least, three at most), to be broadcast to the other
// ranks by the launcher (MPI_Bcast, a file, ...).  The third one gives the position halo its own communicator (option halo_overlap).
extern "C" int imdgpu_dist_unique_id(void *out, int nbytes)
{
  const char *e = rccl_load();
  if (e || nbytes < 2 * (int) sizeof(ncclUniqueId)) return 1;
  const int n = nbytes >= 3 * (int) sizeof(ncclUniqueId) ? 3 : 2;
  for (int k = 0; k < n; k++) {
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
  const int n = nbytes / (int) sizeof(ncclUniqueId);
  if ((n != 2 && n != 3) || nbytes != n * (int) sizeof(ncclUniqueId))
    return imdgpu_fail(ctx, "dist_init_rccl: unique id must be 2 or 3 x %d bytes (imdgpu_dist_unique_id)", (int) sizeof(ncclUniqueId));
  if (dist_new(ctx, nranks, rank)) return 1;
  for (int k = 0; k < n; k++) {
    ncclUniqueId id;
    memcpy(&id, (const char *) unique_id + k * sizeof(id), sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(k == 0 ? &DS(ctx)->comm : k == 1 ? &DS(ctx)->comm2 : &DS(ctx)->comm3, nranks, id, rank);
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
  if (D->nsend_atoms > 0) HIPCHK(hipMemcpyAsync(D->d_send_idx, dev_send_idx, D->nsend_atoms * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
  {
    std::vector<int> seg((size_t) D->nsend_atoms);
    for (int s = 0; s < nsend; s++) for (long long q = 0; q < D->send_cnt[s]; q++) seg[D->send_off[s] + q] = s;
    if (D->nsend_atoms > 0) { HIPCHK(hipMemcpyAsync(D->d_seg_of, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
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
    if (D->nsend_atoms > 0) HIPCHK(hipMemcpyAsync(D->d_self_dst, dst.data(), dst.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(D->d_recv_self, rs.data(), rs.size(), hipMemcpyHostToDevice, ctx->stream));
    double sh[3 * MAXSEG] = {0};
    for (int s = 0; s < 3 * nsend; s++) sh[s] = send_shift[s];
    HIPCHK(hipMemcpyAsync(D->d_shift, sh, sizeof(sh), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));            // (the host vectors above go out of scope)
  }
  const size_t ns = (size_t) D->nsend_atoms * 4 + 8, nr = (size_t) D->nrecv_atoms * 4 + 8;
  if (ns > D->sendbuf_cap) { if (D->sendbuf) (void) hipFree(D->sendbuf); D->sendbuf_cap = (size_t) (ns * 1.2); HIPCHK(hipMalloc((void **) &D->sendbuf, D->sendbuf_cap * sizeof(double))); }
  if (nr > D->recvbuf_cap) { if (D->recvbuf) (void) hipFree(D->recvbuf); D->recvbuf_cap = (size_t) (nr * 1.2); HIPCHK(hipMalloc((void **) &D->recvbuf, D->recvbuf_cap * sizeof(double))); }
  D->have_plan = true; D->halo_ready = false; D->spec_pass1 = false;
  return 0;
}

// what: 0 = (x,y,z,dF) 4 doubles; 1 = dF; 2 = type (as double: one element width for every message); 3 = x,y,z
static __host__ __device__ __forceinline__ int what_width(int what) { return what == 0 ? 4 : what == 3 ? 3 : 1; }

// (Round 5: in the eight-ranks-on-one-GPU rehearsal, about once in 150 runs beside another process on the GPU, the two workgroups of
// k_dist_unpack or k_dist_gather that ONE XCD ran left no trace, or acted like the launch before them, with every input verified
// (IMDGPU_LOCAL_VERIFY).  System-scope loads / write-through stores / fences here did not change that and cost 0.4 ms per step at
// 10 M atoms, so the kernels are plain again; docs/VARIANTS.md has the record.)
// (`what` is a template argument: launches for different kinds of data are different kernels, so that a workgroup which -- as in that
// anomaly -- ran with the arguments of the previous launch of ITS kernel would still move the right kind of data.)
template <int what>
__global__ void k_dist_gather(long long n, const int *idx, const int *seg_of, const int *self_dst, const double *shift,
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
template <int what>
__global__ void k_dist_unpack(long long n, int nlocal, const unsigned char *recv_self, const double *in, double4 *pos, int *type)
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
static int dist_exchange(imdgpu_ctx *ctx, DistState *D, int w, ncclComm_t comm = nullptr)
{
  if (!comm) comm = D->comm;
  std::vector<long long> so, sc, ro, rc; std::vector<int> sp, rp;
  // consecutive segments of one peer are one message (the plan lays the segments out peer by peer); an empty segment does
  // not break a stretch (it occupies no buffer space), both sides see the same counts and merge alike
  for (int s = 0; s < D->nsend; s++) {
    if (D->send_cnt[s] == 0 || (D->send_peer[s] == D->rank && !D->self_comm)) continue;      // self segments: done by the gather
    if (!sp.empty() && sp.back() == D->send_peer[s] && so.back() + sc.back() == D->send_off[s] * w) { sc.back() += D->send_cnt[s] * w; continue; }
    sp.push_back(D->send_peer[s]); so.push_back(D->send_off[s] * w); sc.push_back(D->send_cnt[s] * w);
  }
  for (int s = 0; s < D->nrecv; s++) {
    if (D->recv_cnt[s] == 0 || (D->recv_peer[s] == D->rank && !D->self_comm)) continue;
    if (!rp.empty() && rp.back() == D->recv_peer[s] && ro.back() + rc.back() == D->recv_off[s] * w) { rc.back() += D->recv_cnt[s] * w; continue; }
    rp.push_back(D->recv_peer[s]); ro.push_back(D->recv_off[s] * w); rc.push_back(D->recv_cnt[s] * w);
  }
  if (sp.empty() && rp.empty()) return 0;
  if (D->comm) {
    NCCLCHK(g_rccl.GroupStart());
    for (size_t s = 0; s < sp.size(); s++) NCCLCHK(g_rccl.Send(D->sendbuf + so[s], (size_t) sc[s], ncclDouble, sp[s], comm, ctx->stream));
    for (size_t s = 0; s < rp.size(); s++) NCCLCHK(g_rccl.Recv(D->recvbuf + ro[s], (size_t) rc[s], ncclDouble, rp[s], comm, ctx->stream));
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));             // host transport (tests): the buffers must be complete
  if (D->cb(D->cb_user, 0, D->sendbuf, D->recvbuf, (int) sp.size(), sp.data(), so.data(), sc.data(), (int) rp.size(), rp.data(),
            ro.data(), rc.data(), 8))
    return imdgpu_fail(ctx, "dist: the host transport failed");
  return 0;
}

// parts: 1 = the gather (send buffer; ghost atoms this rank supplies to itself are written in place), 2 = messages + unpack
static int dist_halo_parts(imdgpu_ctx *ctx, DistState *D, int what, int parts, ncclComm_t comm = nullptr)
{
  if (!D->have_plan) return imdgpu_fail(ctx, "dist_halo: no plan (imdgpu_dist_set_plan after the ghost layout)");
  if (what < 0 || what > 3) return imdgpu_fail(ctx, "dist_halo: what must be 0..3");
  const int w = what_width(what);
  ProfScope ps(ctx, PROF_GHOSTS);
#define HALO_K(K, ...) do { if (what == 0) hipLaunchKernelGGL(K<0>, __VA_ARGS__); else if (what == 1) hipLaunchKernelGGL(K<1>, __VA_ARGS__); \
                            else if (what == 2) hipLaunchKernelGGL(K<2>, __VA_ARGS__); else hipLaunchKernelGGL(K<3>, __VA_ARGS__); } while (0)
  if ((parts & 1) && D->nsend_atoms > 0)
    HALO_K(k_dist_gather, dim3(nblk(D->nsend_atoms, 256)), dim3(256), 0, ctx->stream, D->nsend_atoms, D->d_send_idx,
           D->d_seg_of, D->d_self_dst, D->d_shift, ctx->pos, ctx->type, D->sendbuf);
  static const bool verify = cfg_env("IMDGPU_LOCAL_VERIFY") != nullptr;      // diagnostics: the gather's output against a host evaluation
  if (verify && (parts & 1) && D->nsend_atoms > 0 && (what == 1 || what == 3)) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t) D->nsend_atoms;
    std::vector<double> s1(n * w), s2(n * w), sh(3 * MAXSEG);
    std::vector<int> idx(n), sd(n), seg(n);
    std::vector<double4> pos((size_t) ctx->nlocal + ctx->nghost);
    HIPCHK(hipMemcpy(s1.data(), D->sendbuf, s1.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(idx.data(), D->d_send_idx, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sd.data(), D->d_self_dst, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(seg.data(), D->d_seg_of, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sh.data(), D->d_shift, sh.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pos.data(), ctx->pos, pos.size() * sizeof(double4), hipMemcpyDeviceToHost));
    size_t lo = n, hi = 0, nbad = 0;
    auto expect = [&](size_t m, int c) { const double4 r = pos[idx[m]]; return what == 1 ? r.w : (c == 0 ? r.x : c == 1 ? r.y : r.z) + sh[3 * seg[m] + c]; };
    for (size_t m = 0; m < n; m++) {
      if (sd[m] >= 0) continue;
      for (int c = 0; c < w; c++) if (s1[m * w + c] != expect(m, c)) { lo = std::min(lo, m); hi = m + 1; nbad++; break; }
    }
    if (nbad) {
      usleep(20000);
      HIPCHK(hipMemcpy(s2.data(), D->sendbuf, s2.size() * 8, hipMemcpyDeviceToHost));
      size_t still = 0;
      for (size_t m = lo; m < hi; m++) { if (sd[m] >= 0) continue; for (int c = 0; c < w; c++) if (s2[m * w + c] != expect(m, c)) { still++; break; } }
      fprintf(stderr, "[imdgpu local verify] rank %d GATHER what %d: %zu of %zu atoms differ from pos[] right after the stream's synchronisation, atoms [%zu, %zu); 20 ms later %zu still differ; first bad: got %.6g expected %.6g\n",
              D->rank, what, nbad, n, lo, hi, still, s1[lo * w], expect(lo, 0));
    }
  }
  if (parts & 2) {
    if (dist_exchange(ctx, D, w, comm)) return 1;
    if (parts == 2) imdgpu_tl_mark(ctx, "halo3_msgs");
    if (D->nrecv_atoms > D->nself_atoms)                    // something did arrive through the buffers
      HALO_K(k_dist_unpack, dim3(nblk(D->nrecv_atoms, 256)), dim3(256), 0, ctx->stream, D->nrecv_atoms, ctx->nlocal,
             D->d_recv_self, D->recvbuf, ctx->pos, ctx->type);
    if (verify && D->nrecv_atoms > D->nself_atoms && (what == 1 || what == 3)) {        // ... and the unpacked ghost atoms against the receive buffer
      HIPCHK(hipStreamSynchronize(ctx->stream));
      const size_t n = (size_t) D->nrecv_atoms;
      std::vector<double> in(n * w);
      std::vector<double4> gp(n);
      std::vector<unsigned char> rs(n);
      HIPCHK(hipMemcpy(in.data(), D->recvbuf, in.size() * 8, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(rs.data(), D->d_recv_self, n, hipMemcpyDeviceToHost));
      auto count_bad = [&](size_t &lo, size_t &hi) {
        size_t nb = 0; lo = n; hi = 0;
        for (size_t m = 0; m < n; m++) {
          if (rs[m]) continue;
          const double4 q = gp[m];
          const bool bad = what == 1 ? q.w != in[m] : (q.x != in[3 * m] || q.y != in[3 * m + 1] || q.z != in[3 * m + 2]);
          if (bad) { nb++; lo = std::min(lo, m); hi = m + 1; }
        }
        return nb;
      };
      HIPCHK(hipMemcpy(gp.data(), ctx->pos + ctx->nlocal, n * sizeof(double4), hipMemcpyDeviceToHost));
      size_t lo, hi;
      const size_t nb = count_bad(lo, hi);
      if (nb) {
        const double got = what == 1 ? gp[lo].w : gp[lo].x, want = what == 1 ? in[lo] : in[3 * lo];
        usleep(20000);
        HIPCHK(hipMemcpy(gp.data(), ctx->pos + ctx->nlocal, n * sizeof(double4), hipMemcpyDeviceToHost));
        size_t lo2, hi2;
        const size_t nb2 = count_bad(lo2, hi2);
        fprintf(stderr, "[imdgpu local verify] rank %d UNPACK what %d: %zu of %zu ghost atoms differ from the receive buffer after the stream's synchronisation, atoms [%zu, %zu); 20 ms later %zu; first bad: got %.6g expected %.6g\n",
                D->rank, what, nb, n, lo, hi, nb2, got, want);
      }
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int imdgpu_dist_halo(imdgpu_ctx *ctx, int what)
{
  NEED_DIST();
  return dist_halo_parts(ctx, D, what, 3);
}

// MAX over the ranks of this rank's list-validity flag (check_nblist's MPI_Allreduce, imd_forces_nbl.c:1087) -> d_flag[0]
// flag[1]: a time-out of the lattice's hand-offs (persistent sub-step kernel, mailbox) on ANY rank fails the step everywhere
// flag[2]: the persistent sub-step kernel of some rank stepped aside in trusted mode (its status word; nothing of that rank's
// lattice was touched and its integrator returned at once): every rank repeats the tail of the step after that rank has
// re-issued its sub-steps.  flag[3] = flag[0] | flag[2]: what the guarded pass 1 enqueued ahead looks at.
__global__ void k_dist_flag_in(const int *counters, const int *pstat, int *flag)
{
  const int soft = (pstat && *pstat == 1) ? 1 : 0;
  flag[0] = counters[CNT_INVALID]; flag[1] = counters[CNT_TTM_TIMEOUT]; flag[2] = soft; flag[3] = (counters[CNT_INVALID] || soft) ? 1 : 0;
}
static int dist_flag_allreduce(imdgpu_ctx *ctx, DistState *D)
{
  hipLaunchKernelGGL(k_dist_flag_in, dim3(1), dim3(1), 0, ctx->stream, ctx->d_counters, ctx->ttm ? imdgpu_ttm_persist_guard(ctx) : nullptr, D->d_flag);
  if (D->nranks == 1 && !D->self_comm) return 0;
  if (D->comm) { NCCLCHK(g_rccl.AllReduce(D->d_flag, D->d_flag, 4, ncclInt, ncclMax, D->comm, ctx->stream)); return 0; }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (D->cb(D->cb_user, 1, D->d_flag, D->d_flag, 4, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 4))
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
static int dist_ttm1d_chain(imdgpu_ctx *ctx, DistState *D, int steps);
static int dist_commflux(imdgpu_ctx *ctx, DistState *D);
static int dist_ttm_chain(imdgpu_ctx *ctx, DistState *D, int steps)
{
  HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  hipStream_t main = ctx->stream;
  ctx->stream = ctx->stream2;
  int rc = 0;
  if (ctx->ttm_decomposed) {
    // calc_ttm's order (imd_ttm_3D.c:57-60): update_fd, do_ADV -- whose first act is do_COMMFLUX (:1325) --, then the rest.
    // The exchange is a collective's worth of transport kernels: like the replicated lattice's all-gather it must be
    // through before the persistent EAM force pass takes every CU, so that pass waits for ev_gathered (dist_md_step)
    rc = imdgpu_ttm_update_fd(ctx, steps);
    if (!rc && ctx->ttm_adv) rc = dist_commflux(ctx, D);
    if (!rc && hipEventRecord(D->ev_gathered, ctx->stream) != hipSuccess) rc = 1;
    if (!rc && ctx->ttm_adv) rc = imdgpu_ttm_adv(ctx, 1.0, steps);
    if (!rc) rc = imdgpu_ttm_lattice_step(ctx, steps);
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
  if (imdgpu_launch_forces_phase(ctx, 2)) return 1;
  // Two RCCL communicators in flight at once with no fixed device order across ranks is safe only while both transport kernels are
  // co-resident on every rank -- which no multi-GPU run has shown yet: on the RCCL transport with more than one rank the
  // dF halo therefore stays IN FRONT of the fork (the order of the replicated lattice, below) unless option df_halo_beside_head = 1
  // asks for the overlapped order; the other transports (in-process, host callback, a one-rank communicator) have no such hazard.
  const bool head_first = ctx->ttm && ctx->ttm_decomposed && !ctx->ttm1d &&
                          (ctx->df_halo_beside_head > 0 || (ctx->df_halo_beside_head < 0 && !(D->comm && D->nranks > 1)));
  if (head_first) {
    // Decomposed lattice: the chain's head (update_fd, with do_ADV the do_COMMFLUX batch on the lattice stream's own
    // communicator) is forked FIRST and the dF halo runs beside it on the main stream -- between pass 1 and the EAM force pass
    // the chip is otherwise idle, so the halo's 8 bytes per ghost atom travel behind update_fd's 0.2 ms instead of in front
    // of them, and transport kernels of both communicators find room side by side whatever their order on a rank.  The force
    // pass, persistent on every CU, starts once that head is through (ev_gathered): update_fd beside it would crawl
    //  and an exchange not yet resident would wait for the pass to end.
    if (dist_ttm_chain(ctx, D, steps)) return 1;
    if (ctx->rho.present && imdgpu_dist_halo(ctx, 1)) return 1;        // dF of the ghost atoms  [send_cells(copy_dF), :786]
    imdgpu_tl_mark(ctx, "dF_halo");
    HIPCHK(hipStreamWaitEvent(ctx->stream, D->ev_gathered, 0));
  } else {
    if (ctx->rho.present && imdgpu_dist_halo(ctx, 1)) return 1;        // dF of the ghost atoms  [send_cells(copy_dF), :786]
    // Replicated lattice / 1D model: the chain is forked AFTER the dF halo: its all-gather (second communicator, lattice
    // stream) then follows the halo group in device order on EVERY rank, so the two communicators can never wait for each
    // other crosswise (the embedding pass is folded into pass 1, so nothing of substance sits between pass 1 and this
    // point; the EAM force pass below waits for the gather anyway).
    if (ctx->ttm1d) {
      if (dist_ttm1d_chain(ctx, D, steps)) return 1;
      HIPCHK(hipStreamWaitEvent(ctx->stream, D->ev_gathered, 0));
    }
    else if (ctx->ttm) {
      if (dist_ttm_chain(ctx, D, steps)) return 1;
      HIPCHK(hipStreamWaitEvent(ctx->stream, D->ev_gathered, 0));
    }
  }
  if (imdgpu_launch_forces_phase(ctx, 3)) return 1;
  if (ctx->ttm || ctx->ttm1d) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  if (imdgpu_launch_integrate(ctx)) return 1;
  imdgpu_tl_mark(ctx, "move");
  // next step's position halo and the global validity flag; then pass 1 of the next step, guarded by that flag, so the
  // device has work while the host reads the flag
  const bool spec = !exp_env("IMDGPU_NO_SPECULATION");
  auto drop_ahead = [&]() {           // the guarded launch did nothing: it must not count as a pass-1 launch in the kernel timings
    for (int id : {(int) PROF_PASS1B, (int) PROF_PASS1})
      for (size_t q = ctx->ev_pending.size(); q-- > 0;)
        if (ctx->ev_pending[q].id == id) {
          ctx->ev_pool.push_back(ctx->ev_pending[q].a); ctx->ev_pool.push_back(ctx->ev_pending[q].b);
          ctx->ev_pending.erase(ctx->ev_pending.begin() + q);
          break;
        }
  };
  // Option halo_overlap: the position halo travels on its own stream (and communicator) while pass 1 -- in its queue form, two
  // launches -- works through the tiles whose halo cells hold no atom of another rank; see k_pass1_tile_merged<.., DYN>.
  // Order on the main stream: gather (it also writes the ghost atoms this rank supplies to itself), the flags' all-reduce (the
  // guard of both launches), launch A (interior tiles); on the halo stream behind the gather: messages, unpack; launch B (boundary
  // tiles) behind that.
  if (spec && ctx->halo_overlap > 0 && ctx->dyn_ntiles != ctx->ntiles && imdgpu_pass1_dyn_prepare(ctx, D->remote)) return 1;
  const bool overlap = spec && imdgpu_pass1_dyn_ready(ctx) && (!D->comm || D->comm3);
  for (int round = 0;; round++) {
    if (overlap) { if (dist_halo_parts(ctx, D, 3, 1)) return 1; }
    else if (imdgpu_dist_halo(ctx, 3)) return 1;
    if (dist_flag_allreduce(ctx, D)) return 1;
    HIPCHK(hipMemcpyAsync(D->h_flag, D->d_flag, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipEventRecord(D->ev_flag, ctx->stream));
    if (overlap) {
      HIPCHK(hipEventRecord(D->ev_fork3, ctx->stream));
      ctx->pass1_guard = D->d_flag + 3;
      static const int dexp = exp_env_int("IMDGPU_DYN_EXP", 0);    // experiments: 2 = the halo in front of launch A on the main stream
      int rc = (dexp & 2) ? 0 : imdgpu_launch_pass1_dyn(ctx, 0);
      hipStream_t main = ctx->stream;
      if (!rc) {
        if (!(dexp & 2)) ctx->stream = D->stream3;
        if (hipStreamWaitEvent(ctx->stream, D->ev_fork3, 0) != hipSuccess) rc = imdgpu_fail(ctx, "hipStreamWaitEvent failed");
        if (!rc) rc = dist_halo_parts(ctx, D, 3, 2, D->comm3);
        if (!rc && hipEventRecord(D->ev_halo, ctx->stream) != hipSuccess) rc = imdgpu_fail(ctx, "hipEventRecord failed");
        imdgpu_tl_mark(ctx, "halo3");
        ctx->stream = main;
      }
      if (!rc && (dexp & 2)) rc = imdgpu_launch_pass1_dyn(ctx, 0);
      if (!rc && hipStreamWaitEvent(ctx->stream, D->ev_halo, 0) != hipSuccess) rc = imdgpu_fail(ctx, "hipStreamWaitEvent failed");
      if (!rc) rc = imdgpu_launch_pass1_dyn(ctx, 1);
      ctx->pass1_guard = nullptr;
      if (rc) return 1;
    } else
    if (spec) {
      ctx->pass1_guard = D->d_flag + 3;
      int rc = imdgpu_launch_forces_phase(ctx, 1);
      ctx->pass1_guard = nullptr;
      if (rc) return 1;
    }
    HIPCHK(hipEventSynchronize(D->ev_flag));
    // how this rank's persistent sub-step kernel went (replicated lattice, trusted mode): a launch that stepped aside has
    // its sub-steps issued through the launch chain now, the integrator -- which returned at once on the status word -- again;
    // EVERY rank then repeats the halo, the flags and the pass enqueued ahead, because some rank's positions were not moved
    const int r = ctx->ttm ? imdgpu_ttm_persist_settle(ctx) : 0;
    if (r < 0) return 1;
    if (r == 1 && imdgpu_launch_integrate(ctx)) return 1;
    if (!D->h_flag[2]) break;
    if (round > 0) return imdgpu_fail(ctx, "dist_md_step: the persistent sub-step kernel's status did not clear after the step was re-issued");
    if (spec) drop_ahead();
  }
  if (ctx->tl_on) { imdgpu_tl_mark(ctx, "halo+flag+pass1ahead"); imdgpu_tl_flush(ctx, steps); }
  if (D->h_flag[1]) {               // like imdgpu_md_step: the lattice of this step was not (or only half) advanced on some rank
    HIPCHK(hipMemsetAsync(ctx->d_counters + CNT_TTM_TIMEOUT, 0, sizeof(int), ctx->stream));      // reported once
    if (D->h_flag[1] >= 2)
      return imdgpu_fail(ctx, "TTM, decomposed lattice: the active region of some rank's block is not supported by the persistent sub-step "
                              "kernel (more blocks of 4 x 4 x 4 FD cells than one launch brings, or a lattice one layer thin): the lattice of this "
                              "step is not valid; use the replicated lattice (option ttm_decomposed 0) for this run");
    return imdgpu_fail(ctx, "TTM: a hand-off between lattice blocks timed out on some rank (persistent sub-step kernel or mailbox): the lattice of this step is not valid");
  }
  const int invalid = D->h_flag[0];
  if (invalid) {
    ctx->have_valid_nbl = 0; ctx->need_fix_cells = 1; D->have_plan = false;
    if (spec) drop_ahead();
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

// The 1D model on several ranks: every rank bins ITS atoms into the slabs, the slab sums are added up over the ranks (what the
// reference does with MPI_Allreduce on natomsglobal, vcom*global, ..., imd_ttm.c:325-334,340-347,404-407) and every rank then
// runs the same lattice launch on the whole row.  Lattice stream, second communicator, like the replicated 3D lattice's gather.
int imdgpu_dist_sum_dev(imdgpu_ctx *ctx, double *d, int n)
{
  DistState *D = DS(ctx);
  if (!D || D->nranks == 1 || n <= 0) return 0;
  if (D->comm) { NCCLCHK(g_rccl.AllReduce(d, d, (size_t) n, ncclDouble, ncclSum, D->comm2, ctx->stream)); }
  else {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (D->cb(D->cb_user, 3, d, d, n, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 8)) return imdgpu_fail(ctx, "dist: the host transport failed (sum)");
  }
  HIPCHK(hipEventRecord(D->ev_gathered, ctx->stream));       // the EAM force pass waits for the collectives, not for the lattice
  return 0;
}
static int dist_ttm1d_chain(imdgpu_ctx *ctx, DistState *D, int steps)
{
  HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  hipStream_t main = ctx->stream;
  ctx->stream = ctx->stream2;
  HIPCHK(hipEventRecord(D->ev_gathered, ctx->stream));         // one rank: nothing to wait for
  const int rc = imdgpu_ttm1d_calc(ctx, steps);
  ctx->stream = main;
  if (rc) return 1;
  HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
  return 0;
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

// =====================================================================================================================
// The REBUILD, native as well: what fix_cells -> send_atoms (imd_fix_cells_3d.c:43-306,436-542; records of
// imd_mpi_util.c:349-650) and setup_buffers / the first send_cells of a new list (imd_comm_force_3d.c:222-396,
// imd_mpi_util.c:850-1059) do in the reference, for the direct 26-neighbour halo of this file.  A C host needs nothing
// but imdgpu_dist_step(): migration along x, y, z, re-binning, per-cell counts to the 26 peers, ghost layout, send index
// lists (built on the device), tile choice, first halo, neighbour list -- then steps until some rank's list goes stale.
// Every exchange goes through the transport of the step (RCCL / host callback / the in-process transport below).
// =====================================================================================================================
struct PlanSeg {                 // one of the 26 directions; static for a geometry
  int d[3];
  int to = -1, frm = -1;         // rank this direction's boundary layer goes to / rank whose layer fills my ghost region
  double shift[3] = {0, 0, 0};   // periodic image shift, added by the SENDER (imd_comm_force_3d.c:247-266)
  int nsc = 0, nrc = 0;          // cells of the send layer S(d) / of the ghost region R(-d)
  int soff = 0, roff = 0;        // their offsets in the concatenated cell lists (valid when to >= 0 / frm >= 0)
};
struct PlanState {
  bool ready = false;
  int dim[3] = {0, 0, 0};
  PlanSeg seg[26];
  int nsc = 0, nrc = 0;                                  // send cells over all segments with to >= 0; recv cells likewise
  int *d_send_cells = nullptr, *d_recv_cells = nullptr;  // linear cell indices, segment after segment
  int *d_cnt_send = nullptr, *d_cnt_recv = nullptr;      // atoms per listed cell
  int *h_cnt_send = nullptr, *h_cnt_recv = nullptr;      // pinned mirrors
  int *h_aux = nullptr, *d_aux = nullptr;                // per send cell: off, dst, seg; per recv cell: start   (3 nsc + nrc ints)
  int lo[3] = {-1, -1, -1}, hi[3] = {-1, -1, -1};        // migration neighbours (imd_geom_mpi_3d.c:32-90)
  double *mig_send = nullptr, *mig_recv = nullptr; size_t mig_send_cap = 0, mig_recv_cap = 0;
  int *d_mig_cnt = nullptr;                              // 4 ints: n_lo, n_hi out; from_hi, from_lo in
  int nrebuild = 0;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int timing = 0;
};
static std::map<imdgpu_ctx *, PlanState *> g_plans;      // one per context; guarded by g_plans_mu (contexts may live on several threads)
static std::mutex g_plans_mu;
static PlanState *PS(imdgpu_ctx *ctx)
{
  std::lock_guard<std::mutex> lk(g_plans_mu);
  auto it = g_plans.find(ctx);
  if (it != g_plans.end()) return it->second;
  PlanState *p = new PlanState();
  p->timing = cfg_env("IMD_DIST_TIMING") != nullptr;
  g_plans[ctx] = p;
  return p;
}
void imdgpu_dist_plan_free(imdgpu_ctx *ctx)
{
  PlanState *p = nullptr;
  { std::lock_guard<std::mutex> lk(g_plans_mu); auto it = g_plans.find(ctx); if (it == g_plans.end()) return; p = it->second; g_plans.erase(it); }
  void *ptrs[] = {p->d_send_cells, p->d_recv_cells, p->d_cnt_send, p->d_cnt_recv, p->d_aux, p->mig_send, p->mig_recv, p->d_mig_cnt};
  for (void *q : ptrs) if (q) (void) hipFree(q);
  if (p->h_cnt_send) (void) hipHostFree(p->h_cnt_send);
  if (p->h_cnt_recv) (void) hipHostFree(p->h_cnt_recv);
  if (p->h_aux) (void) hipHostFree(p->h_aux);
  delete p;
}

static inline int rank_of(const int c[3], const int P[3]) { return (c[0] * P[1] + c[1]) * P[2] + c[2]; }

// The geometry of the 26 segments in a fixed order (x slowest, then y, z; (0,0,0) skipped) -- the same order on every rank,
// so that the messages between one pair of ranks pair up by position.  Pure host code: exported for the CPU tests
// (imdgpu_dist_plan_geometry) and used by the rebuild.
static void plan_geometry(const int cpu_dim[3], const int coord[3], const int dim[3], const int pbc[3], const double box[9], PlanSeg seg[26],
                          std::vector<int> *send_cells, std::vector<int> *recv_cells)
{
  int q = 0;
  for (int a = -1; a <= 1; a++) for (int b = -1; b <= 1; b++) for (int c = -1; c <= 1; c++) {
    if (!a && !b && !c) continue;
    PlanSeg &g = seg[q];
    g = PlanSeg();
    g.d[0] = a; g.d[1] = b; g.d[2] = c;
    int pt[3], pf[3]; bool ok_to = true, ok_from = true;
    for (int k = 0; k < 3; k++) {
      const int P = cpu_dim[k], t = coord[k] + g.d[k], f = coord[k] - g.d[k];
      if (t < 0 || t >= P) {
        if (!pbc[k]) ok_to = false;
        else if (g.d[k]) for (int x = 0; x < 3; x++) g.shift[x] += box[3 * k + x] * (t < 0 ? 1.0 : -1.0);
      }
      if ((f < 0 || f >= P) && !pbc[k]) ok_from = false;
      pt[k] = ((t % P) + P) % P; pf[k] = ((f % P) + P) % P;
    }
    g.to = ok_to ? rank_of(pt, cpu_dim) : -1;
    g.frm = ok_from ? rank_of(pf, cpu_dim) : -1;
    // S(d): cell index 1 where d_k = -1, dim_k - 2 where d_k = +1, all inner cells where d_k = 0;
    // R(-d) (filled by what arrives from coord - d): dim_k - 1 where d_k = -1, 0 where d_k = +1
    int lo_s[3], hi_s[3], lo_r[3], hi_r[3];
    for (int k = 0; k < 3; k++) {
      if (g.d[k] == 0) { lo_s[k] = lo_r[k] = 1; hi_s[k] = hi_r[k] = dim[k] - 2; }
      else if (g.d[k] < 0) { lo_s[k] = hi_s[k] = 1; lo_r[k] = hi_r[k] = dim[k] - 1; }
      else { lo_s[k] = hi_s[k] = dim[k] - 2; lo_r[k] = hi_r[k] = 0; }
    }
    g.nsc = g.nrc = 0;
    for (int i = lo_s[0]; i <= hi_s[0]; i++) for (int j = lo_s[1]; j <= hi_s[1]; j++) for (int k = lo_s[2]; k <= hi_s[2]; k++) {
      if (send_cells && g.to >= 0) send_cells->push_back((i * dim[1] + j) * dim[2] + k);
      g.nsc++;
    }
    for (int i = lo_r[0]; i <= hi_r[0]; i++) for (int j = lo_r[1]; j <= hi_r[1]; j++) for (int k = lo_r[2]; k <= hi_r[2]; k++) {
      if (recv_cells && g.frm >= 0) recv_cells->push_back((i * dim[1] + j) * dim[2] + k);
      g.nrc++;
    }
    q++;
  }
}

// For tests and for hosts that want to inspect the plan: per direction (26 x) {dx, dy, dz, to, from, cells}; shifts 26 x 3.
extern "C" int imdgpu_dist_plan_geometry(const int cpu_dim[3], const int my_coord[3], const int cell_dim[3], const int pbc[3],
                                         const double box[9], int *seg_out /*26*6*/, double *shift_out /*26*3*/,
                                         int *send_cells /*or NULL*/, int *nsend_cells, int *recv_cells /*or NULL*/, int *nrecv_cells)
{
  PlanSeg seg[26];
  std::vector<int> sc, rc;
  plan_geometry(cpu_dim, my_coord, cell_dim, pbc, box, seg, &sc, &rc);
  for (int q = 0; q < 26; q++) {
    int *o = seg_out + 6 * q;
    o[0] = seg[q].d[0]; o[1] = seg[q].d[1]; o[2] = seg[q].d[2]; o[3] = seg[q].to; o[4] = seg[q].frm; o[5] = seg[q].nsc;
    for (int x = 0; x < 3; x++) shift_out[3 * q + x] = seg[q].shift[x];
  }
  if (send_cells) memcpy(send_cells, sc.data(), sc.size() * sizeof(int));
  if (recv_cells) memcpy(recv_cells, rc.data(), rc.size() * sizeof(int));
  if (nsend_cells) *nsend_cells = (int) sc.size();
  if (nrecv_cells) *nrecv_cells = (int) rc.size();
  return 0;
}

static int plan_static(imdgpu_ctx *ctx, DistState *D, PlanState *P)
{
  if (P->ready && P->dim[0] == ctx->dim[0] && P->dim[1] == ctx->dim[1] && P->dim[2] == ctx->dim[2]) return 0;
  if (rank_of(ctx->my_coord, ctx->cpu_dim) != D->rank || ctx->cpu_dim[0] * ctx->cpu_dim[1] * ctx->cpu_dim[2] != D->nranks)
    return imdgpu_fail(ctx, "dist_rebuild: rank %d of %d does not match the geometry (cpu_dim %d %d %d, my_coord %d %d %d: ranks are "
                       "numbered x slowest, z fastest)", D->rank, D->nranks, ctx->cpu_dim[0], ctx->cpu_dim[1], ctx->cpu_dim[2],
                       ctx->my_coord[0], ctx->my_coord[1], ctx->my_coord[2]);
  std::vector<int> sc, rc;
  plan_geometry(ctx->cpu_dim, ctx->my_coord, ctx->dim, ctx->pbc, ctx->box, P->seg, &sc, &rc);
  int so = 0, ro = 0;
  for (int q = 0; q < 26; q++) {
    PlanSeg &g = P->seg[q];
    if (g.to >= 0) { g.soff = so; so += g.nsc; }
    if (g.frm >= 0) { g.roff = ro; ro += g.nrc; }
  }
  P->nsc = so; P->nrc = ro;
  for (int k = 0; k < 3; k++) {
    P->dim[k] = ctx->dim[k];
    const int Pk = ctx->cpu_dim[k], c = ctx->my_coord[k];
    int n[3] = {ctx->my_coord[0], ctx->my_coord[1], ctx->my_coord[2]};
    P->lo[k] = P->hi[k] = -1;
    if (c > 0 || ctx->pbc[k]) { n[k] = (c - 1 + Pk) % Pk; P->lo[k] = rank_of(n, ctx->cpu_dim); }
    if (c < Pk - 1 || ctx->pbc[k]) { n[k] = (c + 1) % Pk; P->hi[k] = rank_of(n, ctx->cpu_dim); }
  }
  void *old[] = {P->d_send_cells, P->d_recv_cells, P->d_cnt_send, P->d_cnt_recv, P->d_aux};
  for (void *p : old) if (p) (void) hipFree(p);
  if (P->h_cnt_send) (void) hipHostFree(P->h_cnt_send);
  if (P->h_cnt_recv) (void) hipHostFree(P->h_cnt_recv);
  if (P->h_aux) (void) hipHostFree(P->h_aux);
  const size_t ns = (size_t) std::max(so, 1), nr = (size_t) std::max(ro, 1);
  HIPCHK(hipMalloc((void **) &P->d_send_cells, ns * sizeof(int)));
  HIPCHK(hipMalloc((void **) &P->d_recv_cells, nr * sizeof(int)));
  HIPCHK(hipMalloc((void **) &P->d_cnt_send, ns * sizeof(int)));
  HIPCHK(hipMalloc((void **) &P->d_cnt_recv, nr * sizeof(int)));
  HIPCHK(hipMalloc((void **) &P->d_aux, (3 * ns + nr) * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &P->h_cnt_send, ns * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &P->h_cnt_recv, nr * sizeof(int)));
  HIPCHK(hipHostMalloc((void **) &P->h_aux, (3 * ns + nr) * sizeof(int)));
  if (so) HIPCHK(hipMemcpy(P->d_send_cells, sc.data(), (size_t) so * sizeof(int), hipMemcpyHostToDevice));
  if (ro) HIPCHK(hipMemcpy(P->d_recv_cells, rc.data(), (size_t) ro * sizeof(int), hipMemcpyHostToDevice));
  if (!P->d_mig_cnt) HIPCHK(hipMalloc((void **) &P->d_mig_cnt, 8 * sizeof(int)));
  P->ready = true;
  return 0;
}

// generic point-to-point batch of the rebuild: byte-addressed segments of two device buffers.  Messages to THIS rank are
// copied on the device (the k-th self send is the k-th self receive); empty messages are not posted (both sides know).
struct P2P { std::vector<int> sp, rp; std::vector<long long> so, sc, ro, rc; };
static int dist_p2p(imdgpu_ctx *ctx, DistState *D, char *sendbuf, char *recvbuf, const P2P &m, bool lattice_comm = false)
{
  ncclComm_t comm = lattice_comm ? D->comm2 : D->comm;    // the lattice stream has its own communicator (one stream per communicator)
  std::vector<int> sp, rp; std::vector<long long> so, sc, ro, rc;
  std::vector<size_t> self_s, self_r;
  for (size_t k = 0; k < m.sp.size(); k++) {
    if (m.sp[k] == D->rank) { self_s.push_back(k); continue; }
    if (m.sc[k] > 0) { sp.push_back(m.sp[k]); so.push_back(m.so[k]); sc.push_back(m.sc[k]); }
  }
  for (size_t k = 0; k < m.rp.size(); k++) {
    if (m.rp[k] == D->rank) { self_r.push_back(k); continue; }
    if (m.rc[k] > 0) { rp.push_back(m.rp[k]); ro.push_back(m.ro[k]); rc.push_back(m.rc[k]); }
  }
  if (self_s.size() != self_r.size()) return imdgpu_fail(ctx, "dist: self messages do not pair up (%zu sent, %zu expected)", self_s.size(), self_r.size());
  for (size_t k = 0; k < self_s.size(); k++) {
    const size_t a = self_s[k], b = self_r[k];
    if (m.sc[a] != m.rc[b]) return imdgpu_fail(ctx, "dist: self message %zu: %lld bytes sent, %lld expected", k, m.sc[a], m.rc[b]);
    if (m.sc[a] > 0) HIPCHK(hipMemcpyAsync(recvbuf + m.ro[b], sendbuf + m.so[a], (size_t) m.sc[a], hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (sp.empty() && rp.empty()) return 0;
  if (comm) {
    NCCLCHK(g_rccl.GroupStart());
    for (size_t s = 0; s < sp.size(); s++) NCCLCHK(g_rccl.Send(sendbuf + so[s], (size_t) sc[s], ncclChar, sp[s], comm, ctx->stream));
    for (size_t s = 0; s < rp.size(); s++) NCCLCHK(g_rccl.Recv(recvbuf + ro[s], (size_t) rc[s], ncclChar, rp[s], comm, ctx->stream));
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (D->cb(D->cb_user, 0, sendbuf, recvbuf, (int) sp.size(), sp.data(), so.data(), sc.data(), (int) rp.size(), rp.data(), ro.data(), rc.data(), 1))
    return imdgpu_fail(ctx, "dist: the host transport failed (point-to-point batch)");
  return 0;
}

// do_COMMFLUX (imd_ttm_3D.c:667-995, the part compiled without ADV2D) inside the native step, decomposed lattice: my boundary
// columns i = 1 / i = lx-2 of {flux[8], U} go into the x neighbours' ghost columns -- one pack launch per side, ONE batch of at
// most two sends and two receives on the lattice stream (its own communicator), one unpack launch per side.  Like the
// reference the ranks at the two ends of the x range exchange with their inner neighbour only: no wrap-around even with
// periodic x (:724-760), the ghost columns there keep flux 0 / U 0 of init_ttm (:1136-1157).  ctx->stream is the lattice stream.
static int dist_commflux(imdgpu_ctx *ctx, DistState *D)
{
  const int Px = ctx->cpu_dim[0], cx = ctx->my_coord[0];
  if (Px == 1 || D->nranks == 1) return 0;
  int n = 0;
  if (imdgpu_ttm_commflux(ctx, 0, 0, nullptr, &n)) return 1;
  const size_t seg = (size_t) n * 5;                      // doubles per face: eight int counters in four doubles, then U
  if (4 * seg > D->flux_cap) {
    if (D->fluxbuf) (void) hipFree(D->fluxbuf);
    D->fluxbuf = nullptr; D->flux_cap = 0;
    HIPCHK(hipMalloc((void **) &D->fluxbuf, 4 * seg * sizeof(double)));
    D->flux_cap = 4 * seg;
  }
  int nb[3] = {cx, ctx->my_coord[1], ctx->my_coord[2]};
  nb[0] = cx - 1; const int lo = cx > 0 ? rank_of(nb, ctx->cpu_dim) : -1;
  nb[0] = cx + 1; const int hi = cx < Px - 1 ? rank_of(nb, ctx->cpu_dim) : -1;
  double *b = D->fluxbuf;
  if (lo >= 0 && imdgpu_ttm_commflux(ctx, 0, 0, b, nullptr)) return 1;
  if (hi >= 0 && imdgpu_ttm_commflux(ctx, 1, 0, b + seg, nullptr)) return 1;
  P2P m;
  const long long bytes = (long long) (seg * sizeof(double));
  if (lo >= 0) { m.sp.push_back(lo); m.so.push_back(0); m.sc.push_back(bytes); m.rp.push_back(lo); m.ro.push_back(2 * bytes); m.rc.push_back(bytes); }
  if (hi >= 0) { m.sp.push_back(hi); m.so.push_back(bytes); m.sc.push_back(bytes); m.rp.push_back(hi); m.ro.push_back(3 * bytes); m.rc.push_back(bytes); }
  if (dist_p2p(ctx, D, (char *) b, (char *) b, m, true)) return 1;
  if (lo >= 0 && imdgpu_ttm_commflux(ctx, 0, 1, b + 2 * seg, nullptr)) return 1;
  if (hi >= 0 && imdgpu_ttm_commflux(ctx, 1, 1, b + 3 * seg, nullptr)) return 1;
  return 0;
}
// MAX over the ranks of one host int (collective decisions of the rebuild)
static int dist_max_int(imdgpu_ctx *ctx, DistState *D, int *v)
{
  if (D->nranks == 1) return 0;
  HIPCHK(hipMemcpyAsync(D->d_flag + 4, v, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (D->comm) NCCLCHK(g_rccl.AllReduce(D->d_flag + 4, D->d_flag + 4, 1, ncclInt, ncclMax, D->comm, ctx->stream));
  else {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (D->cb(D->cb_user, 1, D->d_flag + 4, D->d_flag + 4, 1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 4))
      return imdgpu_fail(ctx, "dist: the host transport failed (all-reduce)");
  }
  HIPCHK(hipMemcpyAsync(v, D->d_flag + 4, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

__global__ void k_plan_counts(int n, const int *cells, const int *cell_cnt, int *out)
{
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < n) out[m] = cell_cnt[cells[m]];
}
// every ghost cell empty (start = nlocal), then the received regions get their ranges
__global__ void k_plan_clear_ghost_cells(int d0, int d1, int d2, int nlocal, int *cell_start, int *cell_cnt)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d0 * d1 * d2) return;
  const int k = c % d2, j = (c / d2) % d1, i = c / (d2 * d1);
  if (i == 0 || j == 0 || k == 0 || i == d0 - 1 || j == d1 - 1 || k == d2 - 1) { cell_start[c] = nlocal; cell_cnt[c] = 0; }
}
__global__ void k_plan_set_ghost_cells(int n, const int *cells, const int *start, const int *cnt, int *cell_start, int *cell_cnt)
{
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < n) { cell_start[cells[m]] = start[m]; cell_cnt[cells[m]] = cnt[m]; }
}
// the send index lists, cell after cell (the order in which the receiver laid its ghost atoms out): one thread per cell
__global__ void k_plan_send_idx(int n, const int *cells, const int *cell_start, const int *cnt, const int *off, const int *dst,
                                const int *segof, int *idx, int *seg_of, int *self_dst)
{
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const int s = cell_start[cells[m]], c = cnt[m], o = off[m], g = dst[m], sg = segof[m];
  for (int a = 0; a < c; a++) { idx[o + a] = s + a; seg_of[o + a] = sg; self_dst[o + a] = g >= 0 ? g + a : -1; }
}

static double ms_since(std::chrono::steady_clock::time_point &t)
{
  auto n = std::chrono::steady_clock::now();
  double d = std::chrono::duration<double, std::milli>(n - t).count();
  t = n;
  return d;
}

// counts to the 26 peers, ghost layout, send lists, buffers: the plan of the list about to be built
static int dist_plan(imdgpu_ctx *ctx, DistState *D, PlanState *P)
{
  if (plan_static(ctx, D, P)) return 1;
  imdgpu_dist_cancel_ahead(ctx);
  const int nlocal = ctx->nlocal;
  if (P->nsc > 0)
    hipLaunchKernelGGL(k_plan_counts, dim3(nblk(P->nsc, 256)), dim3(256), 0, ctx->stream, P->nsc, P->d_send_cells, ctx->cell_cnt, P->d_cnt_send);
  {
    P2P m;
    for (int q = 0; q < 26; q++) {
      const PlanSeg &g = P->seg[q];
      if (g.to >= 0) { m.sp.push_back(g.to); m.so.push_back((long long) g.soff * 4); m.sc.push_back((long long) g.nsc * 4); }
      if (g.frm >= 0) { m.rp.push_back(g.frm); m.ro.push_back((long long) g.roff * 4); m.rc.push_back((long long) g.nrc * 4); }
    }
    if (dist_p2p(ctx, D, (char *) P->d_cnt_send, (char *) P->d_cnt_recv, m)) return 1;
  }
  if (P->nsc) HIPCHK(hipMemcpyAsync(P->h_cnt_send, P->d_cnt_send, (size_t) P->nsc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (P->nrc) HIPCHK(hipMemcpyAsync(P->h_cnt_recv, P->d_cnt_recv, (size_t) P->nrc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // host: prefix sums per segment
  int *a_off = P->h_aux, *a_dst = P->h_aux + P->nsc, *a_seg = P->h_aux + 2 * (size_t) P->nsc, *a_start = P->h_aux + 3 * (size_t) P->nsc;
  D->send_peer.clear(); D->recv_peer.clear(); D->send_cnt.clear(); D->send_off.clear(); D->recv_cnt.clear(); D->recv_off.clear(); D->send_shift.clear();
  long long run = 0;
  std::vector<int> self_recv;                   // indices (in the receive-segment list) of the segments from this rank
  // Segments are laid out PEER BY PEER (stable in the direction order inside a peer): on 2 x 2 x 2 ranks the 26 directions
  // lead to 7 peers, and what goes to one peer is then one contiguous stretch of the send buffer -- one ncclSend / ncclRecv
  // per peer instead of one per direction (dist_exchange merges them; measured with the 26 self segments of one rank going
  // through RCCL: 0.38 -> see profiles/r04 ms per step for both halos).  A direction's segment pairs with the SAME
  // direction on the other side, so both sides sort alike and messages between a pair of ranks still match by position.
  int ord_r[26], ord_s[26], n_r = 0, n_s = 0;
  for (int q = 0; q < 26; q++) { if (P->seg[q].frm >= 0) ord_r[n_r++] = q; if (P->seg[q].to >= 0) ord_s[n_s++] = q; }
  std::stable_sort(ord_r, ord_r + n_r, [&](int a, int b) { return P->seg[a].frm < P->seg[b].frm; });
  std::stable_sort(ord_s, ord_s + n_s, [&](int a, int b) { return P->seg[a].to < P->seg[b].to; });
  for (int oq = 0; oq < n_r; oq++) {
    const PlanSeg &g = P->seg[ord_r[oq]];
    const long long first = run;
    for (int m = 0; m < g.nrc; m++) { a_start[g.roff + m] = nlocal + (int) run; run += P->h_cnt_recv[g.roff + m]; }
    if (g.frm == D->rank) self_recv.push_back((int) D->recv_peer.size());
    D->recv_peer.push_back(g.frm); D->recv_off.push_back(first); D->recv_cnt.push_back(run - first);
  }
  const long long nghost = run;
  if ((long long) nlocal + nghost > 2000000000LL) return imdgpu_fail(ctx, "dist_rebuild: %lld atoms on one rank exceed the 32-bit atom index", (long long) nlocal + nghost);
  run = 0;
  size_t kself = 0;
  for (int oq = 0; oq < n_s; oq++) {
    const PlanSeg &g = P->seg[ord_s[oq]];
    const long long first = run;
    const int sidx = (int) D->send_peer.size();
    long long dst0 = -1;
    const bool self = g.to == D->rank && !D->self_comm;
    if (g.to == D->rank) {
      if (kself >= self_recv.size()) return imdgpu_fail(ctx, "dist_rebuild: self segments do not pair up");
      if (self) dst0 = nlocal + D->recv_off[self_recv[kself]];
    }
    for (int m = 0; m < g.nsc; m++) {
      a_off[g.soff + m] = (int) run; a_seg[g.soff + m] = sidx;
      a_dst[g.soff + m] = self ? (int) (dst0 + (run - first)) : -1;
      run += P->h_cnt_send[g.soff + m];
    }
    if (g.to == D->rank) {
      if (D->recv_cnt[self_recv[kself]] != run - first) return imdgpu_fail(ctx, "dist_rebuild: self segment %zu: %lld atoms sent, %lld expected", kself, run - first, D->recv_cnt[self_recv[kself]]);
      kself++;
    }
    D->send_peer.push_back(g.to); D->send_off.push_back(first); D->send_cnt.push_back(run - first);
    for (int x = 0; x < 3; x++) D->send_shift.push_back(g.shift[x]);
  }
  D->nsend = (int) D->send_peer.size(); D->nrecv = (int) D->recv_peer.size();
  D->nsend_atoms = run; D->nrecv_atoms = nghost;
  if (D->nsend > MAXSEG || D->nrecv > MAXSEG) return imdgpu_fail(ctx, "dist_rebuild: more than %d segments", MAXSEG);
  // device side of the layout
  if (imdgpu_ensure_capacity(ctx, nlocal, nlocal + (int) nghost)) return 1;
  ctx->nghost = (int) nghost; ctx->have_valid_nbl = 0;
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
  const size_t ns = (size_t) D->nsend_atoms * 4 + 8, nr = (size_t) D->nrecv_atoms * 4 + 8;
  if (ns > D->sendbuf_cap) { if (D->sendbuf) (void) hipFree(D->sendbuf); D->sendbuf_cap = (size_t) (ns * 1.2); HIPCHK(hipMalloc((void **) &D->sendbuf, D->sendbuf_cap * sizeof(double))); }
  if (nr > D->recvbuf_cap) { if (D->recvbuf) (void) hipFree(D->recvbuf); D->recvbuf_cap = (size_t) (nr * 1.2); HIPCHK(hipMalloc((void **) &D->recvbuf, D->recvbuf_cap * sizeof(double))); }
  const size_t naux = 3 * (size_t) P->nsc + P->nrc;
  if (naux) HIPCHK(hipMemcpyAsync(P->d_aux, P->h_aux, naux * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_plan_clear_ghost_cells, dim3(nblk(ctx->nall_cells, 256)), dim3(256), 0, ctx->stream, ctx->dim[0], ctx->dim[1], ctx->dim[2],
                     nlocal, ctx->cell_start, ctx->cell_cnt);
  if (P->nrc > 0)
    hipLaunchKernelGGL(k_plan_set_ghost_cells, dim3(nblk(P->nrc, 256)), dim3(256), 0, ctx->stream, P->nrc, P->d_recv_cells,
                       P->d_aux + 3 * (size_t) P->nsc, P->d_cnt_recv, ctx->cell_start, ctx->cell_cnt);
  if (P->nsc > 0)
    hipLaunchKernelGGL(k_plan_send_idx, dim3(nblk(P->nsc, 128)), dim3(128), 0, ctx->stream, P->nsc, P->d_send_cells, ctx->cell_start, P->d_cnt_send,
                       P->d_aux, P->d_aux + P->nsc, P->d_aux + 2 * (size_t) P->nsc, D->d_send_idx, D->d_seg_of, D->d_self_dst);
  HIPCHK(hipMemsetAsync(D->d_recv_self, 0, (size_t) D->nrecv_atoms + 1, ctx->stream));
  D->nself_atoms = 0;
  if (!D->self_comm)
    for (int r : self_recv) {
      if (D->recv_cnt[r] > 0) HIPCHK(hipMemsetAsync(D->d_recv_self + D->recv_off[r], 1, (size_t) D->recv_cnt[r], ctx->stream));
      D->nself_atoms += D->recv_cnt[r];
    }
  {
    double sh[3 * MAXSEG] = {0};
    for (size_t s = 0; s < D->send_shift.size(); s++) sh[s] = D->send_shift[s];
    HIPCHK(hipMemcpyAsync(D->d_shift, sh, sizeof(sh), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));          // sh and the host vectors are locals
  }
  HIPCHK(hipGetLastError());
  ctx->cells_epoch++;
  D->have_plan = true; D->halo_ready = false; D->spec_pass1 = false;
  return 0;
}

// fix_cells -> send_atoms (imd_fix_cells_3d.c:436-542): along x, y, z in turn, atoms whose cell now belongs to the lower /
// higher neighbour rank leave (10 doubles each) and the arrivals are appended; counts travel first
static int dist_migrate(imdgpu_ctx *ctx, DistState *D, PlanState *P)
{
  if (plan_static(ctx, D, P)) return 1;
  for (int d = 0; d < 3; d++) {
    if (ctx->cpu_dim[d] == 1) continue;
    int n_lo = 0, n_hi = 0;
    if (imdgpu_migrate_scan(ctx, d, &n_lo, &n_hi)) return 1;
    const int lo = P->lo[d], hi = P->hi[d];
    if ((lo < 0 && n_lo) || (hi < 0 && n_hi)) return imdgpu_fail(ctx, "atoms left the simulation box through an open boundary");
    int h[4] = {n_lo, n_hi, 0, 0};
    HIPCHK(hipMemcpyAsync(P->d_mig_cnt, h, 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(P->d_mig_cnt + 2, 0, 2 * sizeof(int), ctx->stream));
    {                                       // sends in the order (low, high), receives (from high, from low): pairs up on grids of two
      P2P m;
      if (lo >= 0) { m.sp.push_back(lo); m.so.push_back(0); m.sc.push_back(4); }
      if (hi >= 0) { m.sp.push_back(hi); m.so.push_back(4); m.sc.push_back(4); }
      if (hi >= 0) { m.rp.push_back(hi); m.ro.push_back(0); m.rc.push_back(4); }
      if (lo >= 0) { m.rp.push_back(lo); m.ro.push_back(4); m.rc.push_back(4); }
      if (dist_p2p(ctx, D, (char *) P->d_mig_cnt, (char *) (P->d_mig_cnt + 2), m)) return 1;
    }
    HIPCHK(hipMemcpyAsync(h + 2, P->d_mig_cnt + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int in_hi = hi >= 0 ? h[2] : 0, in_lo = lo >= 0 ? h[3] : 0;
    const size_t need_s = (size_t) (n_lo + n_hi) * 10 + 16, need_r = (size_t) (in_lo + in_hi) * 10 + 16;
    if (need_s > P->mig_send_cap) { if (P->mig_send) (void) hipFree(P->mig_send); P->mig_send_cap = need_s * 2; HIPCHK(hipMalloc((void **) &P->mig_send, P->mig_send_cap * sizeof(double))); }
    if (need_r > P->mig_recv_cap) { if (P->mig_recv) (void) hipFree(P->mig_recv); P->mig_recv_cap = need_r * 2; HIPCHK(hipMalloc((void **) &P->mig_recv, P->mig_recv_cap * sizeof(double))); }
    if (lo >= 0 && imdgpu_migrate_gather(ctx, 0, P->mig_send)) return 1;
    if (hi >= 0 && imdgpu_migrate_gather(ctx, 1, P->mig_send + (size_t) n_lo * 10)) return 1;
    {
      P2P m;
      if (lo >= 0) { m.sp.push_back(lo); m.so.push_back(0); m.sc.push_back((long long) n_lo * 80); }
      if (hi >= 0) { m.sp.push_back(hi); m.so.push_back((long long) n_lo * 80); m.sc.push_back((long long) n_hi * 80); }
      if (hi >= 0) { m.rp.push_back(hi); m.ro.push_back((long long) in_lo * 80); m.rc.push_back((long long) in_hi * 80); }
      if (lo >= 0) { m.rp.push_back(lo); m.ro.push_back(0); m.rc.push_back((long long) in_lo * 80); }
      if (dist_p2p(ctx, D, (char *) P->mig_send, (char *) P->mig_recv, m)) return 1;
    }
    if (imdgpu_migrate_apply(ctx, in_lo ? P->mig_recv : nullptr, in_lo, in_hi ? P->mig_recv + (size_t) in_lo * 10 : nullptr, in_hi)) return 1;
  }
  return 0;
}

void imdgpu_ttm_block_dims(imdgpu_ctx *ctx, int ldim[3]);

// all-gather of n doubles per rank through the transport of the step (host arrays in and out; rebuild-time set-up only)
static int dist_allgather_host(imdgpu_ctx *ctx, DistState *D, const double *mine, int n, double *all)
{
  if (D->nranks == 1) { memcpy(all, mine, n * sizeof(double)); return 0; }
  double *ds = nullptr, *dr = nullptr;
  HIPCHK(hipMalloc((void **) &ds, n * sizeof(double)));
  HIPCHK(hipMalloc((void **) &dr, (size_t) n * D->nranks * sizeof(double)));
  HIPCHK(hipMemcpy(ds, mine, n * sizeof(double), hipMemcpyHostToDevice));
  int rc = 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (D->comm) { if (g_rccl.AllGather(ds, dr, (size_t) n, ncclDouble, D->comm, ctx->stream) != ncclSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 1; }
  else if (D->cb(D->cb_user, 2, ds, dr, n, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 8)) rc = 1;
  if (!rc && hipMemcpy(all, dr, (size_t) n * D->nranks * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
  (void) hipFree(ds); (void) hipFree(dr);
  return rc ? imdgpu_fail(ctx, "dist: all-gather of the set-up data failed") : 0;
}

// Decomposed TTM lattice: every rank's ghost-record buffer is made known to the ranks beyond its six faces, which from then on
// write their boundary cells' (Te, kappa) straight into it from inside the persistent sub-step kernel (imdgpu_ttm_remote_*,
// k_ttm_chain_persist<.., REM>): what ttm_fill_ghost_layers does with six MPI_Sendrecv per sub-step (imd_ttm_3D.c:3314-3504)
// becomes stores over xGMI without a launch, a collective or the host.  Ranks in other processes are reached through
// hipIpc handles, ranks of this process (threads) by their pointer.  Collective.
extern "C" int imdgpu_ttm_remote_init(imdgpu_ctx *ctx, int coarse, void **dev_ptr, size_t *bytes);
extern "C" int imdgpu_ttm_remote_ipc_handle(imdgpu_ctx *ctx, void *handle64, int *fine);
extern "C" int imdgpu_ttm_remote_open(imdgpu_ctx *ctx, const void *handle64, void **dev_ptr);
extern "C" int imdgpu_ttm_remote_attach(imdgpu_ctx *ctx, void *const peer[6]);
extern "C" int imdgpu_ttm_remote_detach(imdgpu_ctx *ctx);
extern "C" int imdgpu_ttm_remote_ping(imdgpu_ctx *ctx, int phase, unsigned word, int *bad);
int imdgpu_ttm_remote_is_fine(imdgpu_ctx *ctx);
// Collective and failure-safe: a rank on which a step fails keeps taking part in the exchanges and every rank returns the
// same verdict (an error on ALL ranks if it failed on one), so that the host can fall back together -- e.g. to the
// replicated lattice -- instead of one rank leaving the others inside a collective.
extern "C" int imdgpu_dist_ttm_remote_setup(imdgpu_ctx *ctx)
{
  NEED_DIST();
  PlanState *P = PS(ctx);
  if (!ctx->ttm || !ctx->ttm_decomposed) return imdgpu_fail(ctx, "dist_ttm_remote_setup: needs a decomposed TTM lattice");
  if (plan_static(ctx, D, P)) return 1;
  const int N = D->nranks, W = 15;                       // per rank: pid, pointer, fine flag, failed flag, 2 x bus id, 8 x handle, device ordinal
  std::vector<double> mine(W, 0.0), all((size_t) W * N, 0.0);
  char bus[16] = {0};
  (void) hipDeviceGetPCIBusId(bus, 16, ctx->device);
  long long pid = (long long) getpid();
  int bad = 0;
  std::string why;
  auto failed = [&](const char *what) { if (!bad) { bad = 1; why = std::string(what) + ": " + ctx->err; } };
  // round 1: who lives where
  memcpy(&mine[0], &pid, 8); memcpy(&mine[4], bus, 16); mine[14] = ctx->device;
  if (dist_allgather_host(ctx, D, mine.data(), W, all.data())) return 1;          // the transport itself: fails everywhere
  bool all_same_gpu = true, other_process = false;
  for (int d = 0; d < 3; d++)
    for (int side = 0; side < 2; side++) {
      const int nb = side ? P->lo[d] : P->hi[d];
      if (nb < 0 || nb == D->rank) continue;
      long long p2; memcpy(&p2, &all[(size_t) W * nb], 8);
      if (memcmp(&all[(size_t) W * nb + 4], bus, 16) != 0) all_same_gpu = false;
      if (p2 != pid) other_process = true;
    }
  void *ptr = nullptr; size_t bytes = 0;
  const bool force_coarse = exp_env("IMDGPU_TTM_REMOTE_COARSE") != nullptr;
  unsigned char handle[64] = {0};
  int fine = 0;
  if (ctx->ttm_remote_fail == D->rank + 1) {   // test hook: this rank cannot set up
    imdgpu_fail(ctx, "IMDGPU_TTM_REMOTE_FAIL"); failed("test hook");
  }
  if (!bad && imdgpu_ttm_remote_init(ctx, force_coarse ? 1 : 0, &ptr, &bytes)) failed("buffer");
  if (!bad && other_process && imdgpu_ttm_remote_ipc_handle(ctx, handle, &fine)) {
    // no IPC handle for fine-grained memory on this driver: plain device memory is enough when every neighbour shares my GPU
    if (!all_same_gpu) failed("IPC handle");
    else if (imdgpu_ttm_remote_init(ctx, 1, &ptr, &bytes) || imdgpu_ttm_remote_ipc_handle(ctx, handle, &fine)) failed("IPC handle (coarse)");
  }
  fine = imdgpu_ttm_remote_is_fine(ctx);
  // A buffer that another GPU writes into while this GPU's kernel polls it must be FINE-GRAINED (coarse-grained device memory is
  // coherent at kernel boundaries only: the one-shot ping below would pass and the sub-step kernel's polls could still spin on
  // stale lines).  Plain memory is accepted only when every rank beyond my faces shares my GPU (one L2 serves all).  
  if (!bad && !all_same_gpu && !fine) { imdgpu_fail(ctx, "no fine-grained device memory for the ghost records (hipExtMallocWithFlags) and a neighbour rank lives on another GPU"); failed("buffer"); }
  memcpy(&mine[1], &ptr, 8); mine[2] = fine; mine[3] = bad; memcpy(&mine[6], handle, 64);
  if (dist_allgather_host(ctx, D, mine.data(), W, all.data())) return 1;
  for (int r = 0; r < N; r++) if (all[(size_t) W * r + 3] != 0.0 && !bad) { bad = 1; why = "rank " + std::to_string(r) + " could not provide its buffer"; }
  void *peer[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::map<int, void *> opened;
  for (int d = 0; d < 3 && !bad; d++)
    for (int side = 0; side < 2 && !bad; side++) {         // face 2d: towards +d (my high neighbour), 2d+1: towards -d
      const int nb = side ? P->lo[d] : P->hi[d], q = 2 * d + side;
      if (nb < 0 || nb == D->rank) continue;
      auto it = opened.find(nb);
      if (it != opened.end()) { peer[q] = it->second; continue; }
      long long p2; void *rp; memcpy(&p2, &all[(size_t) W * nb], 8); memcpy(&rp, &all[(size_t) W * nb + 1], 8);
      const bool same_gpu = memcmp(&all[(size_t) W * nb + 4], bus, 16) == 0;
      if (!same_gpu && all[(size_t) W * nb + 2] == 0.0) {   // the gathered flag: that rank's buffer is coarse-grained and I would write it from another GPU
        imdgpu_fail(ctx, "rank %d's ghost records are not fine-grained memory", nb); failed("neighbour buffer"); break;
      }
      void *mapped = rp;
      if (p2 != pid) { if (imdgpu_ttm_remote_open(ctx, &all[(size_t) W * nb + 6], &mapped)) { failed("opening a neighbour's IPC handle"); break; } }
      else if (!same_gpu) {
        // a rank of THIS process on another device: its pointer is only usable with peer access (the ping kernel's store
        // would otherwise fault instead of failing the set-up)
        const int pdev = (int) all[(size_t) W * nb + 14];
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctx->device, pdev) != hipSuccess || !can) { (void) hipGetLastError(); imdgpu_fail(ctx, "device %d cannot access device %d", ctx->device, pdev); failed("peer access"); break; }
        const hipError_t e = hipDeviceEnablePeerAccess(pdev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); imdgpu_fail(ctx, "hipDeviceEnablePeerAccess(%d): %s", pdev, hipGetErrorString(e)); failed("peer access"); break; }
        (void) hipGetLastError();
      }
      opened[nb] = mapped; peer[q] = mapped;
    }
  if (!bad && imdgpu_ttm_remote_attach(ctx, peer)) failed("attach");
  // nobody starts writing before every rank has zeroed and attached its buffer -- and everybody learns whether all could
  int anybad = bad;
  if (dist_max_int(ctx, D, &anybad)) return 1;
  // ping: every rank stores a word into the spare granule of each neighbour's buffer from a kernel (system-scope store, as
  // the sub-step kernel does), all ranks meet, every rank reads what its neighbours wrote with the kernel's own polling load
  if (!anybad) {
    const unsigned word = 0x5eed0000u + (unsigned) P->nrebuild;
    if (imdgpu_ttm_remote_ping(ctx, 0, word, nullptr)) failed("ping store");
    int one = bad;
    if (dist_max_int(ctx, D, &one)) return 1;
    int pbad = 0;
    if (!one && imdgpu_ttm_remote_ping(ctx, 1, word, &pbad)) failed("ping load");
    if (pbad && !bad) { bad = 1; why = "a neighbour's store into my ghost records did not arrive (peer access / coherence)"; }
    anybad = bad || one;
    if (dist_max_int(ctx, D, &anybad)) return 1;
  }
  if (anybad) {
    (void) imdgpu_ttm_remote_detach(ctx);
    return imdgpu_fail(ctx, "dist_ttm_remote_setup: ghost records in the neighbours' device memory are not available (%s)",
                       bad ? why.c_str() : "failed on another rank");
  }
  return 0;
}

// The whole rebuild of one rank (collective: every rank calls it).  After it the list is valid and imdgpu_dist_md_step runs.
extern "C" int imdgpu_dist_rebuild(imdgpu_ctx *ctx)
{
  NEED_DIST();
  PlanState *P = PS(ctx);
  auto t = std::chrono::steady_clock::now();
  const bool T = P->timing;
  if (T) HIPCHK(hipStreamSynchronize(ctx->stream));
  if (P->nrebuild > 0 && dist_migrate(ctx, D, P)) return 1;
  if (T) { HIPCHK(hipStreamSynchronize(ctx->stream)); P->ms[0] = ms_since(t); }
  if (imdgpu_fix_cells(ctx)) return 1;
  if (T) { HIPCHK(hipStreamSynchronize(ctx->stream)); P->ms[1] = ms_since(t); }
  if (dist_plan(ctx, D, P)) return 1;
  int changed = 0;
  if (imdgpu_autotile(ctx, &changed)) return 1;      // first rebuild: tile shape from the real occupancies
  if (P->nrebuild == 0 && dist_max_int(ctx, D, &changed)) return 1;   // planning is collective: a rank in vacuum never re-tiles by itself
  if (changed && dist_plan(ctx, D, P)) return 1;     // local atoms were re-sorted: new index lists
  if (T) { HIPCHK(hipStreamSynchronize(ctx->stream)); P->ms[2] = ms_since(t); }
  if (imdgpu_dist_halo(ctx, 0)) return 1;
  if (ctx->ntypes > 1 && imdgpu_dist_halo(ctx, 2)) return 1;
  if (T) { HIPCHK(hipStreamSynchronize(ctx->stream)); P->ms[3] = ms_since(t); }
  if (imdgpu_make_nblist(ctx)) return 1;
  for (int d = 0; d < 3; d++) D->remote[d] = ctx->cpu_dim[d] > 1 || D->self_comm;
  if (imdgpu_pass1_dyn_prepare(ctx, D->remote)) return 1;       // option halo_overlap: the queue order of the tiles
  if (T) {
    HIPCHK(hipStreamSynchronize(ctx->stream)); P->ms[4] = ms_since(t);
    fprintf(stderr, "[dist rebuild %d, native] ms: migrate %.2f, fix_cells %.2f, plan %.2f, exchange %.2f, make_nblist %.2f\n", P->nrebuild,
            P->ms[0], P->ms[1], P->ms[2], P->ms[3], P->ms[4]);
  }
  if (ctx->ttm && ctx->ttm_decomposed && P->nrebuild == 0 && ctx->ttm_remote_opt) {
    // decomposed lattice: faces through ghost records in the neighbours' memory; if that cannot be set up (the verdict is the
    // same on every rank) the host mailbox keeps carrying them -- slower, not wrong
    if (imdgpu_dist_ttm_remote_setup(ctx)) {
      if (ctx->err.find("dist_ttm_remote_setup") == std::string::npos) return 1;          // the transport itself failed
      if (D->rank == 0) fprintf(stderr, "imdgpu: %s; the decomposed lattice stays on the host mailbox\n", ctx->err.c_str());
    }
  }
  if (ctx->ttm && !ctx->ttm_decomposed && !D->d_ttm_offs) {          // replicated lattice: where every rank's update_fd block goes
    int ld[3];
    imdgpu_ttm_block_dims(ctx, ld);
    std::vector<int> offs(3 * (size_t) D->nranks);
    for (int r = 0; r < D->nranks; r++) {
      const int c[3] = {r / (ctx->cpu_dim[1] * ctx->cpu_dim[2]), (r / ctx->cpu_dim[2]) % ctx->cpu_dim[1], r % ctx->cpu_dim[2]};
      for (int x = 0; x < 3; x++) offs[3 * r + x] = c[x] * ld[x];
    }
    if (imdgpu_dist_set_ttm_blocks(ctx, offs.data())) return 1;
  }
  P->nrebuild++;
  return 0;
}
extern "C" int imdgpu_dist_rebuild_count(imdgpu_ctx *ctx, int *n) { if (n) *n = PS(ctx)->nrebuild; return 0; }

// One call per MD step for a C host: rebuild when the list is stale, then the step.
extern "C" int imdgpu_dist_step(imdgpu_ctx *ctx, int steps)
{
  NEED_DIST();
  if ((!D->have_plan || !ctx->have_valid_nbl) && imdgpu_dist_rebuild(ctx)) return 1;
  int stale = 0;
  return imdgpu_dist_md_step(ctx, steps, &stale);     // a stale list leaves have_plan / have_valid_nbl cleared for the next call
}
// complete forces of the CURRENT positions (all passes with their halos), e.g. before a download
extern "C" int imdgpu_dist_calc_forces(imdgpu_ctx *ctx)
{
  NEED_DIST();
  if (!D->have_plan || !ctx->have_valid_nbl) { if (imdgpu_dist_rebuild(ctx)) return 1; }
  else if (imdgpu_dist_halo(ctx, 3)) return 1;
  imdgpu_dist_cancel_ahead(ctx);
  if (imdgpu_launch_forces_phase(ctx, 1) || imdgpu_launch_forces_phase(ctx, 2)) return 1;
  if (ctx->rho.present && imdgpu_dist_halo(ctx, 1)) return 1;
  return imdgpu_launch_forces_phase(ctx, 3);
}

// ---- in-process transport: the ranks are contexts of ONE process, each driven by its own host thread ----------------------
// (several ranks rehearsing on one GPU without a process per rank -- a GPU box admits few processes --, or one process driving
// the GPUs of a node with peer copies).  Same semantics as the host callback: called with the stream drained; messages
// between a pair of ranks match in posting order; collectives over all ranks of the group.
struct LocalPost { int op = -1; char *send = nullptr, *recv = nullptr; int nsend = 0; std::vector<int> sp; std::vector<long long> so, sc; int eb = 0;
                   std::vector<char> snap; };        // IMDGPU_LOCAL_VERIFY (diagnostics): the sender's own copy of its send buffer
struct LocalGroup {
  int n = 0;
  std::mutex mu; std::condition_variable cv;
  int arrived = 0; unsigned long long gen = 0; bool broken = false;
  std::vector<LocalPost> post;
  std::map<std::string, std::pair<void *, size_t>> shared;   // named page-locked regions (the TTM face mailbox)
};
// The copies run on the RECEIVING context's current stream -- the one the caller has drained and launches the consumer of the data on --
// so producer (copy) and consumer (unpack) are ordered in one queue like RCCL's send / receive are.  (Round 5: with the copies on a
// stream of their own, ordered against the consumer through the host only, one run in ~100 of the 2 x 2 x 2 rehearsal beside another
// process on the GPU had single 256-atom blocks of a dF message arrive stale in k_dist_unpack.)
struct LocalMember { LocalGroup *g; int rank; int device; imdgpu_ctx *ctx = nullptr; };
static int local_barrier(LocalGroup *G)
{
  std::unique_lock<std::mutex> lk(G->mu);
  if (G->broken) return 1;
  const unsigned long long my = G->gen;
  if (++G->arrived == G->n) { G->arrived = 0; G->gen++; G->cv.notify_all(); return 0; }
  if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return G->gen != my || G->broken; })) { G->broken = true; G->cv.notify_all(); return 1; }
  return G->broken ? 1 : 0;
}
static int local_exchange(void *user, int op, void *dsend, void *drecv, int nsend, const int *sp, const long long *so, const long long *sc,
                          int nrecv, const int *rp, const long long *ro, const long long *rc, int eb)
{
  LocalMember *M = (LocalMember *) user;
  LocalGroup *G = M->g;
  if (hipSetDevice(M->device) != hipSuccess) return 1;
  const hipStream_t cs = M->ctx->stream;
  LocalPost &me = G->post[M->rank];
  me.op = op; me.send = (char *) dsend; me.recv = (char *) drecv; me.nsend = nsend; me.eb = eb;
  me.sp.clear(); me.so.clear(); me.sc.clear();
  if (op == 0) for (int k = 0; k < nsend; k++) { me.sp.push_back(sp[k]); me.so.push_back(so[k]); me.sc.push_back(sc[k]); }
  static const bool verify = cfg_env("IMDGPU_LOCAL_VERIFY") != nullptr;      // diagnostics: every message checked against the sender's snapshot
  if (verify && op == 0) {
    long long hi = 0;
    for (int k = 0; k < nsend; k++) hi = std::max(hi, so[k] + sc[k]);
    me.snap.resize((size_t) hi * eb);
    if (hi && hipMemcpy(me.snap.data(), dsend, me.snap.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  }
  if (local_barrier(G)) return 1;
  int rc_ = 0;
  if (op == 0) {
    std::vector<int> seen(G->n, 0);                      // how many messages from each peer I have matched so far
    for (int k = 0; k < nrecv && !rc_; k++) {
      const int p = rp[k];
      if (p < 0 || p >= G->n) { rc_ = 1; break; }
      const LocalPost &o = G->post[p];
      int want = seen[p]++, at = -1;
      for (size_t j = 0; j < o.sp.size(); j++) if (o.sp[j] == M->rank && want-- == 0) { at = (int) j; break; }
      if (at < 0 || o.sc[at] != rc[k] || o.eb != eb || o.op != 0) { rc_ = 1; break; }
      if (rc[k] > 0 && hipMemcpyAsync((char *) drecv + ro[k] * eb, o.send + o.so[at] * eb, (size_t) rc[k] * eb, hipMemcpyDeviceToDevice, cs) != hipSuccess) rc_ = 1;
    }
    if (hipStreamSynchronize(cs) != hipSuccess) rc_ = 1;
    if (verify && !rc_) {
      std::vector<int> seen2(G->n, 0);
      for (int k = 0; k < nrecv; k++) {
        const int p = rp[k];
        const LocalPost &o = G->post[p];
        int want = seen2[p]++, at = -1;
        for (size_t j = 0; j < o.sp.size(); j++) if (o.sp[j] == M->rank && want-- == 0) { at = (int) j; break; }
        const size_t bytes = (size_t) rc[k] * eb;
        std::vector<char> got(bytes), now(bytes);
        (void) hipMemcpy(got.data(), (char *) drecv + ro[k] * eb, bytes, hipMemcpyDeviceToHost);
        (void) hipMemcpy(now.data(), o.send + o.so[at] * eb, bytes, hipMemcpyDeviceToHost);
        const char *snap = o.snap.data() + o.so[at] * eb;
        size_t a1 = bytes, b1 = 0, a2 = bytes, b2 = 0;
        for (size_t q = 0; q < bytes; q++) {
          if (got[q] != snap[q]) { a1 = std::min(a1, q); b1 = q + 1; }
          if (now[q] != snap[q]) { a2 = std::min(a2, q); b2 = q + 1; }
        }
        if (b1 || b2)
          fprintf(stderr, "[imdgpu local verify] rank %d <- %d message %d (%zu bytes, eb %d): received differs from the sender's snapshot in [%zu, %zu), the sender's buffer NOW differs in [%zu, %zu)\n",
                  M->rank, p, k, bytes, eb, a1, b1, a2, b2);
      }
    }
  } else if (op == 1 || op == 3) {
    const size_t bytes = (size_t) nsend * eb;
    std::vector<char> mine(bytes), other(bytes);
    for (int r = 0; r < G->n && !rc_; r++) {
      if (G->post[r].op != op || G->post[r].nsend != nsend) { rc_ = 1; break; }
      if (hipMemcpy(r == 0 ? mine.data() : other.data(), G->post[r].send, bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc_ = 1; break; }
      if (r == 0) continue;
      if (op == 1) { int *a = (int *) mine.data(); const int *b = (const int *) other.data(); for (int i = 0; i < nsend; i++) a[i] = b[i] > a[i] ? b[i] : a[i]; }
      else { double *a = (double *) mine.data(); const double *b = (const double *) other.data(); for (int i = 0; i < nsend; i++) a[i] += b[i]; }   // rank order: the same sum on every rank
    }
    if (local_barrier(G)) return 1;                      // everyone has read every send buffer (in-place reductions)
    if (!rc_ && hipMemcpy(drecv, mine.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc_ = 1;
  } else if (op == 2) {
    for (int r = 0; r < G->n && !rc_; r++) {
      if (G->post[r].op != 2 || G->post[r].nsend != nsend) { rc_ = 1; break; }
      if (hipMemcpyAsync((char *) drecv + (size_t) r * nsend * eb, G->post[r].send, (size_t) nsend * eb, hipMemcpyDeviceToDevice, cs) != hipSuccess) rc_ = 1;
    }
    if (hipStreamSynchronize(cs) != hipSuccess) rc_ = 1;
  } else rc_ = 1;
  if (rc_) { std::lock_guard<std::mutex> lk(G->mu); G->broken = true; G->cv.notify_all(); return 1; }
  if (local_barrier(G)) return 1;                        // nobody reuses a send buffer before every reader is through
  return 0;
}
extern "C" int imdgpu_local_group_create(int nranks, void **group)
{
  if (nranks < 1 || !group) return 1;
  LocalGroup *G = new LocalGroup();
  G->n = nranks; G->post.resize(nranks);
  *group = G;
  return 0;
}
extern "C" void imdgpu_local_group_destroy(void *group)
{
  LocalGroup *G = (LocalGroup *) group;
  if (!G) return;
  for (auto &e : G->shared) (void) hipHostFree(e.second.first);
  delete G;
}
// a zero-filled page-locked host region all ranks of the group see under one name (first caller allocates, the size must agree)
extern "C" int imdgpu_local_group_shared(void *group, const char *name, size_t bytes, void **ptr)
{
  LocalGroup *G = (LocalGroup *) group;
  if (!G || !name || !ptr) return 1;
  std::lock_guard<std::mutex> lk(G->mu);
  auto it = G->shared.find(name);
  if (it == G->shared.end()) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return 1;
    memset(p, 0, bytes);
    it = G->shared.emplace(name, std::make_pair(p, bytes)).first;
  }
  if (it->second.second != bytes) return 1;
  *ptr = it->second.first;
  return 0;
}
extern "C" int imdgpu_local_group_barrier(void *group) { return local_barrier((LocalGroup *) group); }
extern "C" int imdgpu_dist_init_local(imdgpu_ctx *ctx, void *group, int rank)
{
  LocalGroup *G = (LocalGroup *) group;
  if (!G || rank < 0 || rank >= G->n) return imdgpu_fail(ctx, "dist_init_local: bad rank %d", rank);
  LocalMember *M = new LocalMember();                    // lives as long as the process (a few bytes per rank)
  M->g = G; M->rank = rank; M->device = ctx->device; M->ctx = ctx;
  HIPCHK(hipSetDevice(ctx->device));
  return imdgpu_dist_init_callback(ctx, G->n, rank, local_exchange, M);
}


// What RCCL itself says about the communicators this context created (bench.py prints it, so that "did RCCL see N ranks on N
// devices" is in the line): which = 0 -> the SMALLEST ncclCommCount over the communicators (0: none was created: another transport),
// 1 -> how many communicators exist, 2 -> the HIP device ncclCommCuDevice reports for the main one (-1: none)
long long imdgpu_dist_rccl_info(imdgpu_ctx *ctx, int which)
{
  DistState *D = (DistState *) ctx->dist;
  if (!D) return which == 2 ? -1 : 0;
  ncclComm_t cs[3] = {D->comm, D->comm2, D->comm3};
  int n = 0, mn = 0;
  for (ncclComm_t c : cs) {
    if (!c) continue;
    int cnt = 0;
    if (g_rccl.CommCount && g_rccl.CommCount(c, &cnt) == ncclSuccess) mn = n == 0 ? cnt : (cnt < mn ? cnt : mn);
    n++;
  }
  if (which == 0) return mn;
  if (which == 1) return n;
  int dev = -1;
  if (D->comm && g_rccl.CommCuDevice && g_rccl.CommCuDevice(D->comm, &dev) != ncclSuccess) dev = -1;
  return dev;
}
