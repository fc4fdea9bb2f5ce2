// C-ABI of the FEM hot path (include/fembrain_hip.h): handle life cycle, the per-step driver and the
// inspection entry points.  Kernels live in fem_device.hip.h, the host-side plan in fem_plan.cpp.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <chrono>

#include "comm.h"
#include "common.h"
#include "fem_device.hip.h"
#include "fem_plan.h"
#include "pcg_pipe.hip.h"
#include "pcg_pipe2.hip.h"
#include "plan_device.h"
#include "renumber.h"
#include "delta.h"

using namespace fb;

struct fb_fem_s {
  fb_fem_params prm;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;          // the slot-major assembly of the few slices too wide for the element-major kernel runs beside it (launch_rows)
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  fb_comm_s* comm = nullptr;  // not owned
  P2P* p2p = nullptr;         // direct peer mailboxes for halo refresh and dots (comm.h); null = collective library
  int xch_mode = FB_XCH_COLLECTIVE;     // how the exchanges of a sharded handle run (fb_fem_set_exchange_mode)
  DevBuf<int> send_off_dev, halo_off_dev;
  DevBuf<unsigned char> slice_halo;
  FemPlan plan;
  double lambda = 0, mu = 0;
  int grid = 8;
  bool f64 = false;
  // mesh
  DevBuf<int4> tets;
  DevBuf<double> x0, rest, fe;
  DevBuf<char> rec;  // MT[16] per tet
  DevBuf<char> kcorr;  // exact tangent (warp = 2): MT[144] per tet, the rotation-derivative terms of the element stiffness
  // Newmark (ImplicitNewmarkSparse): acceleration and the state at the start of the step
  DevBuf<double> qacc, q1, qvel1, qacc1;
  double nm_beta = 0.25, nm_gamma = 0.5, nm_eps = 1e-6;
  int nm_max_newton = 1;
  bool pcg_warm = false;  // the next solve starts from the x left by the previous one (Newmark), not from 0
  // matrix
  DevBuf<int> slice_off, colidx, slot_coff, slot_ccnt, send_local;
  DevBuf<uint32_t> contrib;
  DevBuf<short> coldelta;  // 16-bit column words (device-built plans); c16 says whether the SpMV may use them and in which form:
  int c16 = 0;             // 0 no, 1 column - row (unsharded), 2 the halo form of a shard (plan_device.hip k_plan_sell)
  DevBuf<int> halo_base;   // c16 == 2: lowest halo column of every slice
  PlanWorkspace plan_ws;  // the device plan builder's temporaries, kept for the next re-sync
  MeshDelta delta;        // fb_fem_resync_delta: the change on the device and its scratch
  DevBuf<int4> tets_next, tets_caller;   // (the element list being built; swapped with `tets`)
  DevBuf<double> x0_next;
  DevBuf<int> map_next_a, map_next_b;
  int last_resync_path = FB_RESYNC_FULL;
  int ren_nodes_at_build = 0;  // nodes when the internal order was last built from scratch
  DevBuf<int> flat_flag;
  DevBuf<int> inc_off;               // element-major assembly (k_assemble_tets): incidence lists per slice, see fem_device.hip.h
  DevBuf<uint32_t> inc, inc_slot;
  bool asm_tets = false;             // the assembly kernel in use
  int asm_lds = 0, asm_grid = 0, asm_max_width = 0;
  DevBuf<int> pipe_wg_first;         // persistent solver: the deal of the slices to the workgroups balanced by slots (pipe_deal), empty: equal numbers
  std::vector<int> pipe_wg_first_host;
  DevBuf<int4> pipe_tasks;           // persistent solver: per workgroup and wavefront its share of the slices (helpers, pcg_pipe.hip.h PipeArgs)
  int pipe_help_waves = 0;           // wavefronts launched beyond slices + service wavefront, for the helpers (0: none)
  DevBuf<unsigned long long> pipe_lines;  // k_gather_lines' three sums
  bool pipe_xyz = false;             // the published vector node by node instead of in planes (irregular meshes; k_pcg_pipe<..., XYZ>)
  int pipe_lines_nodes = 0;          // nodes of the plan the counts below were taken on
  double pipe_gather_lines[2] = {0, 0};  // cache lines a slot's gathers touch on average: three planes | 24-byte records (k_gather_lines; 0: not measured)
  int pipe_n_help = 0, pipe_help_tasks = 0;  // most helper tasks of a workgroup; all of them
  int asm_wide = 0;                  // slices wider than the element-major kernel takes (kIncMaxWidth slots): k_assemble_wide assembles those
  int asm_wide_slots = 0, asm_wide_grid = 0;  // the widest of them; workgroups of k_assemble_wide (each with a scratch area of asm_wide_slots slots)
  DevBuf<int> wide_list;             // their slice numbers
  DevBuf<double> res_all;            // Newmark, several Newton iterations: the residual of every DOF (AsmOut::res_all)
  DevBuf<double> wide_scratch;
  bool asm_staged = false;           // k_assemble_tets_st (records staged in LDS, mass entries precomputed) instead of k_assemble_tets
  int asm_lds_st = 0, asm_grid_st = 0;
  bool mass_valid = false;           // h->mblk holds the mass entries of the current rest data (k_mass_blocks)
  // locality renumbering behind the ABI (renumber.h): the handle works in its own node order, ids are mapped on the way in and out
  Renumbering ren;
  DevBuf<double> xyz_in;               // the caller-order rest positions the order was derived from
  DevBuf<double> io;                   // staging of a caller-order vector on its way to / from the internal order
  bool x0_ready = false;               // build_plan_on_device has put the (permuted) rest positions in place already
  bool masks_ready = false;            // ... and the constraint masks (device_constraint_masks)
  std::vector<int> l2c;                // a renumbered SHARDED handle: caller id of every local node (owned, then halo); empty otherwise
  unsigned long long order_sum = 0;    // ... and a checksum of the order, compared across the ranks
  int n_cu_device = 256;               // CUs of the device (FB_MATRIX_AUTO's size rule; setup_persist asks the device itself)
  bool shard_auto_on = false;          // FB_RENUMBER_AUTO on a sharded handle: the ranks voted for the internal order (vote_shard_order)
  int shard_vote_neighbours = 0;       // most neighbour ranks any rank would have had under the caller's numbering (what the vote saw)
  DevBuf<int> fixed_stage;
  std::vector<int> c_bptr, c_bcol, c_src;  // the pattern in the caller's numbering and the internal block behind each of its blocks (inspection entry points)
  bool caller_pattern = false;
  std::vector<double> x0_stage;  // host staging of the rest positions in local numbering (kept: a re-sync does not fault fresh pages)
  DevBuf<int> d_bptr, d_bcol, d_blk_slot;  // device-built plan only: pattern and slot table, fetched when an inspection entry point asks
  DevBuf<unsigned int> d_ucnt;             // ... and the pairs of every block (unsharded): with the three above and the contribution table, what fb_fem_resync_delta updates
  bool span_stale = false;                 // ren.span_after / mean_after are to be measured again (fb_fem_renumbering)
  bool csr_ready = false;                  // the four describe the current plan
  bool device_plan = false, host_pattern = true;
  DevBuf<uint8_t> dofmask;
  DevBuf<uint8_t> nodemask;  // the three dofmask bytes of a node as bits 0..2 (one gather per column in the assembly)
  DevBuf<char> vals;  // MT[n_slots][9][64]
  DevBuf<char> dlo;   // MT[n_slices][9][64]: low part of every row's diagonal block
  DevBuf<double> mblk;
  DevBuf<float> volf;                // rest volumes as the fp32 records hold them (k_tet_rest -> k_mass_blocks)
  DevBuf<double> invblk;  // FB_PCG_BLOCK_JACOBI: inverse 3x3 diagonal block per row
  // vectors (3*n_local each)
  DevBuf<double> q, qvel, fext, fint, rhs, x, r, d, Ad, invdiag, tmp, sendbuf;
  DevBuf<double> part_a, part_b, part_c, scal;
  DevBuf<CGState> st;
  DevBuf<int> counter;
  CGState* st_host = nullptr;  // pinned, 2 slots
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr}, ev_batch[2] = {nullptr, nullptr};
  bool system_valid = false;
  bool poisoned = false;  // a re-sync failed half way: the buffers no longer belong to one mesh; only a successful re-sync (or destroy) is accepted
  double last_assembly_s = 0, last_solve_s = 0;
  // one batch of 30 PCG iterations (29 merged + the exact-residual one) captured once and replayed: the launch sequence
  // and every kernel argument repeat from batch to batch, and on meshes of ~100k tets the host's launch rate, not the
  // device, would otherwise bound the iteration time
  int split = 0;       // wavefronts per slice of the SpMV on small / mid-size meshes (k_spmv_split): 0 (row kernel), 2 or 4
  int sgrid = 8;       // blocks (= partial sums) of the SpMV launches; equals grid unless split
  bool spmv_nt = false;  // stream the matrix values non-temporally (systems larger than the Infinity Cache, see k_spmv)
  int vgrid = 8;       // blocks of the merged vector pass: one 16-byte pair per thread
  hipGraphExec_t batch_graph = nullptr;
  const double* graph_rhs = nullptr;
  bool use_graph = true;
  bool literal_now = false, graph_literal = false;  // this solve runs the literal two-reduction sequence (tolerance below kPersistMinEps); what the captured batch runs
  // persistent solver (pcg_pipe.hip.h): one workgroup per CU, `persist_waves` wavefronts = slices each, the whole solve in one launch
  bool persist = false;
  int persist_blocks = 0, persist_waves = 0;
  DevBuf<long long> persist_timing;    // FEMBRAIN_PERSIST_TIMING=1 (development aid)
  DevBuf<unsigned long long> pipe_post;
  DevBuf<unsigned int> pipe_flags;     // [blocks padded to 4] flags, [+4] the error word, [+8..9] the two sequence numbers
  DevBuf<int> pipe_prod, pipe_prod_count, pipe_prod_xcd;
  DevBuf<unsigned int> pipe_xcc;        // per workgroup: the XCC id it announced in the current launch (k_pcg_pipe: plain-store publish)
  int pipe_plain_local = 0;            // interior workgroups publish with plain stores (FEMBRAIN_PIPE_PLAIN_STORES)
  DevBuf<double> pipe_planes, pipe_z, pipe_s, pipe_state;
  int pipe_klt = 0, pipe_wmax = 0;
  // sharded persistent solver (pcg_shard_box.hip.h; opt-in FEMBRAIN_SHARDED_PERSIST=1, unmeasured on multi-GPU hardware)
  bool shard_persist = false;
  bool persist_broken = false;  // a launch timed out: the handle runs the two-launch iteration until it is re-armed (below) or re-synced
  // Re-arming: one co-tenant burst must not cost a 1M-tet host 40 % of its speed for the life of the handle.  After
  // `rearm_after` clean two-launch solves an unsharded handle tries the persistent launch again; every further time-out doubles the
  // wait (32, 64, 128, ... solves; FEMBRAIN_PERSIST_REARM=n sets the first, 0 = never).  A sharded handle re-arms at a re-sync only
  // (the ranks must switch together).
  int rearm_after = 32, clean_solves = 0, persist_rearms = 0;
  char* sbox = nullptr;                // my box (fine-grained, mapped by the peers)
  void* sbox_opened[kP2PMaxRanks] = {nullptr};
  long long sbox_halo_cap = 0;
  DevBuf<char*> sbox_peers;
  DevBuf<int> sh_peer_seg, sh_halo_off, sh_row_send_off, sh_row_send_rank, sh_row_send_pos, sh_n_senders, sh_proxy_wg, sh_wg_duty;
  DevBuf<int2> sh_wg_range;
  int sh_n_proxy = 1, sh_relief = 0;
  DevBuf<unsigned int> sh_wg_send_mask;
  int pipe_flag_extra = 0;             // flag slots after the workgroups' (the proxies' flags of a sharded handle)
  int pipe_rows = 1;                   // rows per lane: 1 = k_pcg_pipe (up to 12 slices per CU), 2 = k_pcg_pipe2 (13..24)
  int pipe_max_producers = 0;          // longest producer list (-1: some workgroup polls all)
  bool pipe_stats_pending = false;     // ... still on the device (pipe_stats: [0] longest list, [1] someone polls all)
  DevBuf<int> pipe_owner, pipe_stats;
  DevBuf<unsigned int> pipe_mask;
  long long persist_timeout_ticks = 0; // wall_clock64 ticks (100 MHz) a wait inside a persistent launch may last
  int persist_fallbacks = 0;           // solves that had to be repeated with the two-launch form
  int persist_launches = 0;            // persistent launches made by this handle
  int last_pcg_path = 0;               // FB_PCG_PATH_* of the last solve
  int cu_limit = 0;                    // > 0: the stream is confined to this many CUs (FEMBRAIN_CU_MASK)
  hipEvent_t ev_p[2] = {nullptr, nullptr};  // around every persistent launch
  double persist_seconds = 0;          // device seconds of all persistent launches of this handle (HIP events on its stream)
  long long persist_iterations = 0;    // PCG iterations they ran
};

namespace {

size_t mt_size(const fb_fem_s* h) { return h->f64 ? sizeof(double) : sizeof(float); }

SellView sell_view(const fb_fem_s* h) {
  SellView sv;
  sv.slice_off = h->slice_off.p; sv.colidx = h->colidx.p; sv.n_slices = h->plan.n_slices; sv.n_owned = h->plan.n_owned;
  sv.coldelta = h->c16 ? h->coldelta.p : nullptr;
  sv.halo_base = h->c16 == 2 ? h->halo_base.p : nullptr;
  return sv;
}

// one wavefront per slice: does any of its columns lie in the halo?
__global__ __launch_bounds__(kBlock) void k_slice_halo(int n_slices, int n_owned, const int* __restrict__ slice_off, const int* __restrict__ colidx,
                                                       unsigned char* __restrict__ out) {
  const int s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  bool any = false;
  for (int k = slice_off[s]; k < slice_off[s + 1]; k++) any = any || colidx[(size_t)k * 64 + lane] >= n_owned;
  const unsigned long long b = __ballot(any);
  if (lane == 0) out[s] = b != 0ULL ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_widen_positions(long long n, const float* __restrict__ in, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// FB_RENUMBER_* of this handle: fb_fem_params.renumber unless FEMBRAIN_RENUMBER=0/1 says otherwise; a sharded handle renumbers on request only
// Zero fills of a (re-)build, batched: about twenty buffers are cleared, and a fill per buffer costs 4-5 us of launch each however
// small it is.  add() notes them (4-byte granularity), flush() clears up to kZeroMax per launch.
constexpr int kZeroMax = 24;
struct ZeroList { unsigned int* p[kZeroMax]; unsigned long long end[kZeroMax]; unsigned int tail_words[kZeroMax]; int n; };  // end: running total of 16-byte chunks
__global__ __launch_bounds__(kBlock) void k_zero_many(ZeroList z, unsigned long long total_chunks) {
  for (unsigned long long c = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; c < total_chunks; c += (unsigned long long)gridDim.x * kBlock) {
    int b = 0;
    while (c >= z.end[b]) b++;
    const unsigned long long local = c - (b ? z.end[b - 1] : 0ULL);
    if (c + 1 == z.end[b] && z.tail_words[b]) {  // the buffer's last chunk holds fewer than four words: the fill never leaves the buffer
      for (unsigned int k = 0; k < z.tail_words[b]; k++) z.p[b][4 * local + k] = 0u;
    } else {
      reinterpret_cast<uint4*>(z.p[b])[local] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}
struct ZeroBatch {
  ZeroList z;
  hipStream_t s;
  int rc = FB_OK;
  explicit ZeroBatch(hipStream_t stream) : s(stream) { z.n = 0; }
  // p: 16-byte aligned (the start of an allocation); bytes: a multiple of 4 (else the runtime's fill takes it)
  void add(void* p, size_t bytes) {
    if (!p || !bytes || rc != FB_OK) return;
    if (bytes & 3) {  // (not a whole number of words: the runtime's fill)
      if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) rc = fail(FB_EDEVICE, "hipMemsetAsync failed");
      return;
    }
    if (z.n == kZeroMax) flush();
    const unsigned long long chunks = (bytes + 15) / 16, before = z.n ? z.end[z.n - 1] : 0ULL;
    z.p[z.n] = static_cast<unsigned int*>(p);
    z.end[z.n] = before + chunks;
    z.tail_words[z.n] = (unsigned int)((bytes & 15) / 4);
    z.n++;
  }
  template <typename T>
  void add(DevBuf<T>& b) { add(b.p, b.n * sizeof(T)); }
  int flush() {
    if (rc != FB_OK) return rc;
    if (z.n == 0) return FB_OK;
    const unsigned long long total = z.end[z.n - 1];
    const unsigned int blocks = (unsigned int)std::min<unsigned long long>((total + kBlock - 1) / kBlock, 256ULL * 16);
    hipLaunchKernelGGL(k_zero_many, dim3(blocks), dim3(kBlock), 0, s, z, total);
    if (hipGetLastError() != hipSuccess) rc = fail(FB_EDEVICE, "k_zero_many failed to launch");
    z.n = 0;
    return rc;
  }
};

// slack of this handle's allocations (common.h): a quarter for a mesh that will be cut, or what reserve_nodes / reserve_elements ask for
int handle_slack(const fb_fem_s* h, int n_nodes, int n_tets) {
  if (!h->prm.expect_cuts) return 2;
  double f = 0.25;
  if (h->prm.reserve_nodes > n_nodes && n_nodes > 0) f = std::max(f, (double)h->prm.reserve_nodes / n_nodes - 1.0);
  if (h->prm.reserve_elements > n_tets && n_tets > 0) f = std::max(f, (double)h->prm.reserve_elements / n_tets - 1.0);
  return std::max(2, std::min(32, (int)std::ceil(f * 16.0)));
}

// FB_MATRIX_AUTO: fp32 values from 2 slices per CU on (setup_persist's `w`, the persistent solver's range) and where the persistent solver is
// asked for by name; fp64 below (include/fembrain_hip.h).  A shard decides by the WHOLE mesh, so that the sharded and the unsharded handle
// of one mesh store the same values; the sharded persistent solver, asked for through the environment, needs fp32 as the unsharded one does.
bool auto_matrix_f64(const fb_fem_s* h, int n_nodes, int n_ranks) {
  const int cus = h->cu_limit > 0 ? std::min(h->cu_limit, h->n_cu_device) : h->n_cu_device;
  const int nb = std::min(kPipeMaxBlocks, (cus / 8) * 8);
  const int w = nb >= 8 ? ceil_div(ceil_div(ceil_div(n_nodes, 64), 8), nb / 8) : 0;
  const bool shard_persist_asked = n_ranks > 1 && getenv("FEMBRAIN_SHARDED_PERSIST") && atoi(getenv("FEMBRAIN_SHARDED_PERSIST")) != 0;
  return h->prm.pcg_variant != FB_PCG_PERSISTENT && !shard_persist_asked && w < 2;
}

constexpr int kFreshOrderPercent = 2;
int fresh_order_percent() {
  const char* e = getenv("FEMBRAIN_FRESH_ORDER_PERCENT");
  return e ? std::max(0, atoi(e)) : kFreshOrderPercent;
}

int renumber_mode(const fb_fem_s* h) {
  if (const char* e = getenv("FEMBRAIN_RENUMBER")) return atoi(e) != 0 ? FB_RENUMBER_ON : FB_RENUMBER_OFF;
  if (h->prm.renumber == 0 && h->shard_auto_on) return FB_RENUMBER_ON;  // (a sharded handle under AUTO whose ranks voted for it)
  // a mesh that will be cut: the nodes a cut appends make any caller order wide, so the internal order is chosen at creation and the
  // first fb_fem_resync_delta merges its nodes into it instead of sending the mesh through the full builder
  if (h->prm.renumber == 0 && h->prm.expect_cuts && h->plan.n_ranks <= 1 && !h->comm) return FB_RENUMBER_ON;
  return h->prm.renumber > 0 ? FB_RENUMBER_ON : (h->prm.renumber < 0 ? FB_RENUMBER_OFF : FB_RENUMBER_AUTO);
}

int upload_masks(fb_fem_s* h) {
  const FemPlan& P = h->plan;
  FB_TRY(h->dofmask.upload(P.dofmask, h->stream));
  std::vector<uint8_t> nm((size_t)P.n_local);
  for (int l = 0; l < P.n_local; l++) nm[l] = (uint8_t)((P.dofmask[3 * (size_t)l] ? 1 : 0) | (P.dofmask[3 * (size_t)l + 1] ? 2 : 0) | (P.dofmask[3 * (size_t)l + 2] ? 4 : 0));
  return h->nodemask.upload(nm, h->stream);
}

void release_shard_persist(fb_fem_s* h) {
  for (auto& o : h->sbox_opened) { if (o) (void)hipIpcCloseMemHandle(o); o = nullptr; }
  if (h->sbox) (void)hipFree(h->sbox);
  h->sbox = nullptr;
  h->shard_persist = false;
}

// The rank-local half of the sharded persistent solver's set-up (pcg_shard_box.hip.h): planes over owned + halo columns, per-row send
// lists, producer lists with the proxies' flags.  The collective half (boxes, sender counts) is attach_pipe_shard.
int setup_persist_shard_local(fb_fem_s* h, int nb, int w) {
  const FemPlan& P = h->plan;
  hipStream_t s = h->stream;
  const int R = P.n_ranks;
  if (R > kP2PMaxRanks) return FB_OK;
  FB_TRY(h->sh_halo_off.upload(P.halo_off, s));
  // per slice: its owned column range and the ranks whose halo rows it gathers
  DevBuf<int4> range;
  FB_TRY(range.alloc((size_t)std::max(1, P.n_slices)));
  hipLaunchKernelGGL(k_slice_colrange_shard, dim3(ceil_div(std::max(1, P.n_slices), kWavesPerBlock)), dim3(kBlock), 0, s, P.n_slices, P.n_owned, R, h->slice_off.p,
                     h->colidx.p, h->sh_halo_off.p, range.p);
  FB_HIP(hipGetLastError());
  std::vector<int4> rg((size_t)std::max(1, P.n_slices));
  FB_TRY(range.download(rg.data(), rg.size(), s));
  // slice -> workgroup.  A workgroup whose slices gather halo rows gets its input later than the others by the length of the
  // cross-rank chain (drain -> counter -> proxy copy -> flag; ~4.5 us between two processes on one MI355X) EVERY iteration, and the
  // whole grid ends up at its pace.  So it is dealt fewer slices ("relief", FEMBRAIN_SHARD_RELIEF=n, default 35 % of the
  // slices per CU): its shorter product absorbs the wait.
  const int relief_env = getenv("FEMBRAIN_SHARD_RELIEF") ? atoi(getenv("FEMBRAIN_SHARD_RELIEF")) : -1;
  std::vector<int2> wg_range((size_t)nb, make_int2(0, 0));   // first slice and slice count of every workgroup
  // Default (profiles/r04_remote_delay.json: two and four ranks on CU shares of one GPU, remote signals delayed by 0 / 1 / 2 / 5 us): relief
  // pays where FEW workgroups gather halo rows -- 1M tets on two ranks, 2 of 28 planes per rank: 18.5 against 20.6 us per iteration, and
  // still 21.4 against 22.8 with 2 us added per hop -- and costs where many do: on four ranks (14 planes per rank, the geometry of the
  // 8M-tet mesh on eight GPUs) the even deal runs 20.4 against 23.5.  So: 35 % of the slices per CU while at most a twelfth of the slices
  // gather halo rows, none beyond.
  int halo_slices = 0;
  for (int sl = 0; sl < P.n_slices; sl++) halo_slices += rg[sl].z != 0 ? 1 : 0;
  const int relief_default = 12 * halo_slices <= P.n_slices ? std::max(0, (w * 35 + 50) / 100) : 0;
  int Q = w, relief = std::min(w - 1, relief_env >= 0 ? relief_env : relief_default);
  for (;; relief--) {
    Q = 0;
    if (relief <= 0 || (nb & 7)) {  // the even deal of the unsharded kernel
      relief = 0;
      for (int b = 0; b < nb; b++) {
        int first, count;
        pipe_slices(P.n_slices, nb, b, &first, &count);
        wg_range[b] = make_int2(first, count);
        Q = std::max(Q, count);
      }
      break;
    }
    // a slice that gathers halo rows weighs w / (w - relief) slices; equal weight per workgroup, contiguous, and in pipe_slices'
    // order: the workgroups of XCD b & 7 (round-robin dispatch) hold a contiguous eighth
    const double rho = (double)w / (double)(w - relief);
    double W = 0;
    for (int sl = 0; sl < P.n_slices; sl++) W += rg[sl].z != 0 ? rho : 1.0;
    std::vector<int> start((size_t)nb + 1, P.n_slices);
    double acc = 0;
    int pos = 0;
    for (int sl = 0; sl < P.n_slices; sl++) {
      while (pos < nb && acc >= W * pos / nb - 1e-9) start[pos++] = sl;   // position pos starts at the first slice at or past its share
      acc += rg[sl].z != 0 ? rho : 1.0;
    }
    const int per = nb >> 3;
    for (int p2 = 0; p2 < nb; p2++) {
      const int b = (p2 % per) * 8 + p2 / per;
      wg_range[b] = make_int2(start[p2], start[p2 + 1] - start[p2]);
      Q = std::max(Q, start[p2 + 1] - start[p2]);
    }
    if (Q <= 2 * (kPipeMaxWaves - 1)) break;
  }
  w = Q;
  h->pipe_rows = w < kPipeMaxWaves ? 1 : 2;   // 12..22 slices per CU: two rows per lane (k_pcg_pipe2_shard)
  h->pipe_wmax = h->pipe_rows == 2 ? 12 : (w < 8 ? 8 : 12);   // (one wavefront more than slices / slice pairs: the spare one serves the proxies and the sums)
  h->pipe_klt = h->pipe_rows == 2 ? std::min(kPipe2Klt, pipe2_lds_slots(w, false) / std::max(w, 1)) : std::min(h->pipe_wmax == 8 ? 8 : 6, pipe_lds_slots(false) / std::max(w, 1));  // (a shard's columns are 32-bit)
  h->persist_blocks = nb; h->persist_waves = w;
  h->sh_relief = relief;
  h->pipe_flag_extra = kP2PMaxRanks * kShardProxies;
  FB_TRY(h->sh_wg_range.upload(wg_range, s));
  FB_TRY(h->pipe_post.alloc((size_t)2 * nb * 4));
  FB_TRY(h->pipe_post.zero(s));
  FB_TRY(h->pipe_flags.alloc((size_t)nb + h->pipe_flag_extra + 16));
  FB_TRY(h->pipe_flags.zero(s));
  const size_t n_pad = (size_t)ceil_div(P.n_local, 64) * 64, nv = (size_t)3 * P.n_local + 2;
  FB_TRY(h->pipe_planes.alloc(2 * 3 * n_pad));
  FB_TRY(h->pipe_planes.zero(s));
  FB_TRY(h->pipe_z.alloc(nv));
  FB_TRY(h->pipe_s.alloc(nv));
  FB_TRY(h->pipe_state.alloc(2));
  FB_TRY(h->pipe_state.zero(s));
  h->persist_timing.release();
  std::vector<int> owner((size_t)P.n_slices, 0);
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < wg_range[b].y; k++) owner[wg_range[b].x + k] = b;
  // per-row send lists and per-workgroup destination masks from the plan's send lists (ascending owned ids per destination: the
  // position of a row in that list is its position in the destination's halo segment of this rank)
  std::vector<int> row_off((size_t)P.n_owned + 1, 0);
  for (int q = 0; q < R; q++)
    for (int k = P.send_off[q]; k < P.send_off[q + 1]; k++) row_off[(size_t)P.send_local[k] + 1]++;
  for (int a = 0; a < P.n_owned; a++) row_off[(size_t)a + 1] += row_off[a];
  std::vector<int> row_rank((size_t)std::max(1, row_off[P.n_owned])), row_pos(row_rank.size()), fill(row_off.begin(), row_off.end() - 1);
  std::vector<unsigned int> wg_mask((size_t)nb, 0u);
  for (int q = 0; q < R; q++)
    for (int k = P.send_off[q]; k < P.send_off[q + 1]; k++) {
      const int a = P.send_local[k];
      row_rank[fill[a]] = q; row_pos[fill[a]] = k - P.send_off[q]; fill[a]++;
      wg_mask[owner[a >> 6]] |= 1u << q;
    }
  FB_TRY(h->sh_row_send_off.upload(row_off, s));
  FB_TRY(h->sh_row_send_rank.upload(row_rank, s));
  FB_TRY(h->sh_row_send_pos.upload(row_pos, s));
  FB_TRY(h->sh_wg_send_mask.upload(wg_mask, s));
  // proxies: up to kShardProxies per rank I have halo nodes of, dealt to the workgroups from the last one downwards
  int n_src = 0;
  for (int q = 0; q < R; q++) n_src += P.halo_off[q + 1] > P.halo_off[q];
  const int K = std::max(1, std::min(kShardProxies, nb * kShardDuties / std::max(1, n_src)));
  h->sh_n_proxy = K;
  std::vector<int> proxy((size_t)R * K, -1), duty((size_t)nb * kShardDuties, -1), n_duty((size_t)nb, 0);
  for (int q = 0, idx = 0; q < R; q++)
    for (int k = 0; k < K; k++) {
      int lo, hi;
      shard_proxy_rows(P.halo_off[q], P.halo_off[q + 1], K, k, &lo, &hi);
      if (hi <= lo) continue;
      const int b = nb - 1 - (idx++ % nb);   // (idx < nb * kShardDuties by the choice of K)
      proxy[(size_t)q * K + k] = b;
      duty[(size_t)b * kShardDuties + n_duty[b]++] = q * K + k;
    }
  FB_TRY(h->sh_proxy_wg.upload(proxy, s));
  FB_TRY(h->sh_wg_duty.upload(duty, s));
  // producer lists: local workgroups from the owned column range of every slice, proxies from the ranks its halo columns belong to
  std::vector<int> prod((size_t)nb * kPipeMaxProducers, -1), cnt((size_t)nb, 0), far((size_t)nb, 1);
  std::vector<char> mark((size_t)nb + (size_t)R * K);
  h->pipe_max_producers = 0;
  h->pipe_stats_pending = false;
  for (int b = 0; b < nb; b++) {
    std::fill(mark.begin(), mark.end(), 0);
    int n = 0;
    for (int sl0 = wg_range[b].x; sl0 < wg_range[b].x + wg_range[b].y; sl0++) {
      const int4 r = rg[sl0];
      for (int sl = r.x >> 6; r.y >= r.x && sl <= (r.y >> 6) && sl < P.n_slices; sl++) {
        const int o = owner[sl];
        if (o != b && !mark[o]) { mark[o] = 1; n++; }
      }
      for (int q = 0; q < R; q++)
        if ((unsigned int)r.z >> q & 1u)
          for (int kk = 0; kk < K; kk++)   // (all proxies of a rank this slice gathers halo rows of)
            if (proxy[(size_t)q * K + kk] >= 0 && !mark[nb + q * K + kk]) { mark[nb + q * K + kk] = 1; n++; }
    }
    if (n > kPipeMaxProducers) { cnt[b] = -1; h->pipe_max_producers = -1; continue; }
    cnt[b] = n;
    if (h->pipe_max_producers >= 0) h->pipe_max_producers = std::max(h->pipe_max_producers, n);
    for (int o = 0, k = 0; o < nb + R * K; o++) if (mark[o]) prod[(size_t)b * kPipeMaxProducers + k++] = o;
  }
  FB_TRY(h->pipe_prod.upload(prod, s));
  FB_TRY(h->pipe_prod_count.upload(cnt, s));
  FB_TRY(h->pipe_prod_xcd.upload(far, s));
  h->pipe_plain_local = 0;
  h->shard_persist = true;  // (confirmed or withdrawn by attach_pipe_shard, collectively)
  return FB_OK;
}

// how long a wait inside a persistent launch may last before the launch gives up (read when a plan is built and when the solver is re-armed)
void read_persist_timeout(fb_fem_s* h) {
  const char* t = getenv("FEMBRAIN_PERSIST_TIMEOUT_MS");
  // default 50 ms: a whole 1M-tet solve is ~25 ms, one wait is microseconds.  A sharded handle also waits for the OTHER RANKS' launches
  // to begin, and those are separated by host jitter (first launch: code object load): 2 s there
  const double ms = t ? atof(t) : (h->plan.n_ranks > 1 ? 2000.0 : 50.0);
  h->persist_timeout_ticks = std::max(1LL, (long long)(ms * 1e5));  // 100 MHz
}

// Decides whether this handle solves inside persistent launches and allocates what they need (called for every (re)built plan).
int setup_persist(fb_fem_s* h) {
  const FemPlan& P = h->plan;
  hipStream_t s = h->stream;
  ZeroBatch zb(s);  // (the clears of this set-up in one launch)
  h->persist = false;
  // Nothing of the previous plan's sharded persistent solver survives a rebuild: a re-sync to a mesh that is no longer
  // eligible must not leave attach_pipe_shard a stale "keep", stale send lists or a stale workgroup deal.  setup_persist_shard_local
  // is the only place that sets shard_persist again; a fresh plan also gets a fresh chance after a time-out.
  h->shard_persist = false;
  h->persist_broken = false;
  h->pipe_stats_pending = false;
  h->clean_solves = 0;
  h->rearm_after = getenv("FEMBRAIN_PERSIST_REARM") ? std::max(0, atoi(getenv("FEMBRAIN_PERSIST_REARM"))) : 32;
  h->sh_halo_off.release(); h->sh_row_send_off.release(); h->sh_row_send_rank.release(); h->sh_row_send_pos.release(); h->sh_proxy_wg.release();
  h->sh_wg_duty.release(); h->sh_wg_range.release(); h->sh_wg_send_mask.release();
  hipDeviceProp_t prop;
  FB_HIP(hipGetDeviceProperties(&prop, h->prm.device));
  const int nb = std::min(kPipeMaxBlocks, ((h->cu_limit > 0 ? std::min(h->cu_limit, prop.multiProcessorCount) : prop.multiProcessorCount) / 8) * 8);
  const char* e = getenv("FEMBRAIN_PCG_PERSIST");
  int w = nb >= 8 ? ceil_div(ceil_div(P.n_slices, 8), nb / 8) : 0;
  // The deal of the slices to the workgroups.  Equal NUMBERS of slices per workgroup (pipe_slices) is equal work only where the slices are
  // alike; on a mesh with hull nodes of 40-60 neighbours the workgroups of the hull stream 2.5 x the slots of the others and every
  // iteration waits for them (606k-tet Delaunay probe: product 18 us there, 7.5 on average).  Where the fullest workgroup of the equal
  // deal has a quarter more slots than the average, the slices of every XCD's share are dealt to its workgroups by SLOTS instead
  // (contiguous runs still, at most 12 -- or 24 where the two-row kernel is the mesh's anyway -- slices each).  FEMBRAIN_PIPE_BALANCE=0/1.
  h->pipe_wg_first_host.clear();
  h->pipe_wg_first.release();
  if (P.n_ranks == 1 && nb >= 8 && w >= 1 && w <= 2 * kPipeMaxWaves && !P.slice_off.empty()) {
    const int per = nb / 8, chunk = ceil_div(P.n_slices, 8), cap = w <= kPipeMaxWaves ? kPipeMaxWaves : 2 * kPipeMaxWaves;
    long long total = 0, worst = 0;
    for (int b = 0; b < nb; b++) {
      int first, count;
      pipe_slices(P.n_slices, nb, b, &first, &count);
      const long long sl = count > 0 ? P.slice_off[first + count] - P.slice_off[first] : 0;
      total += sl;
      worst = std::max(worst, sl);
    }
    const char* eb = getenv("FEMBRAIN_PIPE_BALANCE");
    // (a fullest workgroup a quarter above the average -- or 15 % above it on a mesh whose widest slice is half again as wide as the mean:
    // the 56^3 cube after one cut, 200 slots against 171, 20.2 -> 19.3 us per iteration.  Forced on a uniform mesh the deal by slots is
    // SLOWER, 16.4 against 15.6 at 1M tets: some workgroups then hold 12 slices and a smaller LDS share each.)
    int widest = 0;
    for (int sl = 0; sl < P.n_slices; sl++) widest = std::max(widest, P.slice_off[sl + 1] - P.slice_off[sl]);
    const bool uneven = P.n_slices > 0 && (long long)widest * 2 * P.n_slices >= 3 * (long long)P.slice_off[P.n_slices];
    // (uneven WIDTHS only: on a small uniform mesh the fullest workgroup is a quarter above the average because it holds 2 slices where others
    // hold 1, which no deal changes -- and dealt by slots such a mesh was 2-5 % slower: 27^3 cube 8.65 -> 8.86 us per iteration, ventricle.blob 8.28 -> 8.62)
    const bool want = eb ? atoi(eb) != 0 : (uneven && (worst * nb * 4 > total * 5 || (w <= kPipeMaxWaves && worst * nb * 100 > total * 115)));  // (the two-row kernel at 13 slices per CU: 28.7 with the equal deal, 29.9 by slots)
    if (want) {
      std::vector<int> tab((size_t)2 * nb + 2, 0);
      int most = 0;
      for (int x = 0; x < 8; x++) {
        const int lo = std::min(x * chunk, P.n_slices), hi = std::min((x + 1) * chunk, P.n_slices);
        int at = lo;
        for (int j = 0; j < per; j++) {
          const int b = x + 8 * j, left = per - j;
          const long long rest = P.slice_off[hi] - P.slice_off[at];
          const double target = (double)rest / left;
          int n = 0;
          long long cum = 0;
          // at least what the workgroups behind cannot take, then slices while the run stays closer to its share than without the next one
          const int must = std::max(0, (hi - at) - (left - 1) * cap);
          while (at + n < hi && n < cap) {
            const long long wd = P.slice_off[at + n + 1] - P.slice_off[at + n];
            if (n >= must && left > 1 && (double)cum + 0.5 * (double)wd > target) break;
            cum += wd;
            n++;
          }
          tab[b] = at;
          tab[(size_t)nb + 1 + b] = n;
          at += n;
          most = std::max(most, n);
        }
      }
      tab[nb] = P.n_slices;
      h->pipe_wg_first_host = tab;
      FB_TRY(h->pipe_wg_first.upload(tab, s));
      if (getenv("FEMBRAIN_TIMING")) fprintf(stderr, "[fembrain] persistent solver: slices dealt by slots (fullest workgroup of the equal deal: %lld slots, average %.1f); up to %d slices per workgroup\n", worst, (double)total / nb, most);
      w = std::max(1, most);
    }
  }
  read_persist_timeout(h);
  const bool explicit_p = h->prm.pcg_variant == FB_PCG_PERSISTENT;
  const bool shard_opt = P.n_ranks > 1 && getenv("FEMBRAIN_SHARDED_PERSIST") && atoi(getenv("FEMBRAIN_SHARDED_PERSIST")) != 0;
  if (explicit_p && P.n_ranks > 1 && !shard_opt) return fail(FB_EINVAL, "FB_PCG_PERSISTENT on a sharded handle needs FEMBRAIN_SHARDED_PERSIST=1 (unmeasured on multi-GPU hardware)");
  if (explicit_p && h->f64) return fail(FB_EINVAL, "FB_PCG_PERSISTENT needs FB_MATRIX_F32 storage (part of the matrix is kept in LDS as fp32 words)");
  if (explicit_p && (nb < 8 || w < 1 || w > 2 * kPipeMaxWaves))
    return fail(FB_EINVAL, "FB_PCG_PERSISTENT needs at most %d slices per CU, this mesh has %d on %d CUs", 2 * kPipeMaxWaves, w, nb);
  // asked for explicitly (parameter or FEMBRAIN_PCG_PERSIST=1), or by default where it was measured faster than the
  // two-launch iteration: fp32 storage, up to 12 slices per CU
  // (a sharded handle: opt-in, one row per lane, and a spare wavefront per workgroup for the proxies and the sums)
  const bool eligible = !h->f64 && nb >= 8 && w >= 1 && (P.n_ranks == 1 ? w <= 2 * kPipeMaxWaves : (shard_opt && w <= 2 * (kPipeMaxWaves - 1)));  // (sharded: the spare wavefront leaves 11 for slices, two rows per lane from 12 slices on)
  // (us per iteration, two-launch vs persistent, on MI355X: 7.83 / 7.87 at 125 slices = 1 per CU, 8.74 / 8.74 at 308 and 8.98 / 8.64 at 466
  // = 2 per CU, 10.9 / 8.8 at 614 = 3 per CU, 14.0 / 10.3 at 792, 15.8 / 8.9 at 1,000, 27.4 / 15.75 at 2,744 = 1M tets)
  const int min_w = getenv("FEMBRAIN_PERSIST_MIN_WAVES") ? atoi(getenv("FEMBRAIN_PERSIST_MIN_WAVES")) : 2;
  // FB_PCG_BLOCK_JACOBI (opt-in, outside parity): the one-row persistent kernel with the block preconditioner, same rule
  const bool bj = h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI;
  const bool rows2_forced = getenv("FEMBRAIN_PERSIST_ROWS") && atoi(getenv("FEMBRAIN_PERSIST_ROWS")) == 2;
  const bool by_default = (h->prm.pcg_variant == FB_PCG_MERGED || bj) && w >= min_w;
  const bool want_p = e ? atoi(e) != 0 && (h->prm.pcg_variant == FB_PCG_MERGED || explicit_p || bj) : (explicit_p || by_default);
  if (!want_p || !eligible) return FB_OK;
  if (bj && (w > kPipeMaxWaves || rows2_forced || P.n_ranks > 1)) return FB_OK;  // (no two-row form with the block preconditioner)
  if (P.n_ranks > 1) return setup_persist_shard_local(h, nb, w);
  h->persist = true; h->persist_blocks = nb; h->persist_waves = w;
  // pipelined whole-solve kernel
  // up to 12 slices per CU: one row per lane (k_pcg_pipe); 13..24: two (k_pcg_pipe2, no LDS-resident slots).  FEMBRAIN_PERSIST_ROWS=2
  // forces the two-row kernel on a smaller system (tests).
  h->pipe_rows = w > kPipeMaxWaves || rows2_forced ? 2 : 1;
  // Up to 4 slices per CU (BASELINE config 2: 105k tets = 2 per CU) k_pcg_pipe<..,5,16> keeps the WHOLE slice in LDS -- 16 slots each where
  // (8, 8) keeps 8 of ~15 and streams the rest from L2 in every product.  Built, tested, measured -- and NOT the
  // default: 9.24 / 9.50 / 9.63 us per iteration at 27^3 / 33^3 / 37^3 against 8.82 / 9.06 / 9.58 with (8, 8).  The phase table says why
  // (profiles/r05_small_mesh_phase_table.txt): the product is 2.2 of the 8.9 us, the other 6.6 are the hand-offs between the CUs (drain of
  // the publish stores 1.0, flag + neighbour wait + acquire 3.2, sweep of the sums 2.0, recurrences 0.5), which no residency shortens.
  // FEMBRAIN_PIPE_SMALL=1 selects it.
  const bool small = h->pipe_rows == 1 && w <= 4 && !bj && P.n_ranks == 1 && getenv("FEMBRAIN_PIPE_SMALL") && atoi(getenv("FEMBRAIN_PIPE_SMALL")) != 0;
  h->pipe_wmax = h->pipe_rows == 2 ? 12 : (small ? 5 : (w <= 8 ? 8 : 12));
  // slots of every slice resident in LDS at least: the CU's 62 / 65 (16-bit columns) wavefront-slots dealt to the slices of a workgroup (k_pcg_pipe), at most 16 / 8 / 6
  const int lds_all = pipe_lds_slots(h->c16 != 0), lds_help = pipe_help_slots(h->c16 != 0);
  h->pipe_klt = h->pipe_rows == 2 ? std::min(kPipe2Klt, pipe2_lds_slots(w, h->c16 != 0) / std::max(w, 1)) : std::min(small ? 16 : (w <= 8 ? 8 : 6), lds_all / std::max(w, 1));
  // Helpers for very wide slices (unsharded, one row per lane, Jacobi): the widest slice at least half again as wide as the average and
  // wider than 24 slots.  A workgroup's wavefronts without a slice of their own -- those its neighbours' fuller deal leaves idle, and
  // the ones launched for the purpose: the 12-wavefront instantiation then serves fewer than 9 slices per CU too -- take the upper halves of
  // the longest streams, longest first, until none is left or no stream is longer than 2 x 8 slots.  FEMBRAIN_PIPE_HELPERS=0/1 overrides.
  h->pipe_tasks.release();
  h->pipe_help_waves = 0; h->pipe_n_help = 0; h->pipe_help_tasks = 0;
  if (!bj && !small && P.n_ranks == 1 && !P.slice_off.empty()) {  // (one row per lane: helpers and layout; two rows per lane: the layout)
    int mx = 0;
    long long tot = 0;
    for (int sl = 0; sl < P.n_slices; sl++) { const int wd = P.slice_off[sl + 1] - P.slice_off[sl]; mx = std::max(mx, wd); tot += wd; }
    const double mean = P.n_slices ? (double)tot / P.n_slices : 0.0;
    const char* eh = getenv("FEMBRAIN_PIPE_HELPERS");
    const bool want_h = h->pipe_rows == 1 && (eh ? atoi(eh) != 0 : (mx > 24 && (double)mx >= 1.5 * mean));
    // Where do the columns of a slot lie?  On a structured mesh the 64 rows of a slice have consecutive columns and a gather touches 4 lines
    // of each of the three planes of the published vector; on an unstructured one it touches ~50 of each, and a vector stored node by node
    // (24-byte records) costs half of those.  Sampled on the device (every 8th slice), decided here: node by node where the planes cost 30 lines
    // and more per slot and half again the records' (606k-tet Delaunay probe: 80 against 44 lines, 18.9 -> 16.4 us per iteration; the cube
    // after a cut: 19 against 15, and there the planes are the faster form, 19.4 against 20.6, each load touching 4 lines instead of 12).  FEMBRAIN_PIPE_XYZ=0/1 overrides.  (The table-driven instantiation carries the layout.)
    bool want_xyz = false;
    // (a re-sync that changes the mesh by a few per cent keeps the counts of the plan before it: the read-back is a host wait)
    const bool keep_lines = h->pipe_gather_lines[0] > 0.0 && h->pipe_lines_nodes > 0 && std::abs(P.n_local - h->pipe_lines_nodes) * 20 <= h->pipe_lines_nodes;
    if (keep_lines) {
      const char* ex = getenv("FEMBRAIN_PIPE_XYZ");
      want_xyz = P.n_local < (1 << 24) &&
                 (ex ? atoi(ex) != 0 : (h->pipe_gather_lines[0] >= 30.0 && h->pipe_gather_lines[0] >= 1.5 * h->pipe_gather_lines[1]));
    } else {
      h->pipe_gather_lines[0] = h->pipe_gather_lines[1] = 0.0;
      h->pipe_lines_nodes = 0;
    }
    if (!keep_lines && P.n_local < (1 << 24) && h->colidx.p) {
      FB_TRY(h->pipe_lines.alloc(4));
      FB_HIP(hipMemsetAsync(h->pipe_lines.p, 0, 4 * sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_gather_lines, dim3(64), dim3(256), 0, s, P.n_slices, 8, h->slice_off.p, h->colidx.p, h->pipe_lines.p);
      FB_HIP(hipGetLastError());
      unsigned long long got[3] = {0, 0, 0};
      FB_HIP(hipMemcpyAsync(got, h->pipe_lines.p, sizeof got, hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
      if (got[2]) { h->pipe_gather_lines[0] = 3.0 * (double)got[0] / (double)got[2]; h->pipe_gather_lines[1] = (double)got[1] / (double)got[2]; h->pipe_lines_nodes = P.n_local; }
      const char* ex = getenv("FEMBRAIN_PIPE_XYZ");
      want_xyz = ex ? atoi(ex) != 0 : (got[2] && h->pipe_gather_lines[0] >= 30.0 && h->pipe_gather_lines[0] >= 1.5 * h->pipe_gather_lines[1]);
    }
    h->pipe_xyz = h->pipe_rows == 2 && want_xyz;  // (the two-row kernel: a template parameter; the one-row kernel: with its task table, below)
    if (h->pipe_rows == 1 && (want_h || want_xyz) && w <= kPipeMaxWaves) {
      const int help_waves = std::max(0, kPipeMaxWaves - w - 1);   // (the 12-wavefront kernel: slices, helpers, the service wavefront)
      std::vector<int4> tasks((size_t)nb * kPipeTaskStride, make_int4(-1, 0, 0, 0));
      int most = 0, all = 0, deepest = 0;
      const int min_len = getenv("FEMBRAIN_PIPE_HELP_MINLEN") ? atoi(getenv("FEMBRAIN_PIPE_HELP_MINLEN")) : 16;  // (development)
      // FEMBRAIN_PIPE_LDS_BY_WIDTH=1 (measured and NOT the default): the workgroup's LDS dealt to its slices by width instead of in equal
      // shares, so that every wavefront streams about the same number of slots (the 27-slot slices of a cut among slices of 15 would keep 17
      // here instead of 5).  It is slower -- cut cube 21.9 against 20.1 us per iteration, Delaunay probe 22.0 against 18.6: a resident slot
      // still gathers its x entries from L2 / the fabric, and the compiled loop over resident slots has 6 slots' gathers in flight where
      // the hand-written stream never drains; on these meshes the product waits for gathers, not for matrix bytes.
      const bool even_share = !(getenv("FEMBRAIN_PIPE_LDS_BY_WIDTH") && atoi(getenv("FEMBRAIN_PIPE_LDS_BY_WIDTH")) != 0);
      // (with the node-by-node vector a resident slot's gathers cost a third of the lines, and slices that keep up to 12 slots in LDS instead
      // of 6 pay: sliver-free unstructured probe 16.2 -> 14.4 us per iteration, Delaunay probe 16.25 -> 15.9; 18 buys nothing more)
      // -- from 6 slices per CU on: with fewer, idle wavefronts split the streams (helpers) and a long LDS loop in the owner only delays them:
      // 27,000 / 46,656 / 64,000-node lattices 11.0 / 12.2 / 13.5 us per iteration with 6 slots against 12.9 / 13.4 / 14.1 with 12
      const int lds_cap = want_xyz && w >= 6 ? 12 : 6;
      for (int b = 0; b < nb; b++) {
        int first, count;
        pipe_deal(h->pipe_wg_first_host.empty() ? nullptr : h->pipe_wg_first_host.data(), P.n_slices, nb, b, &first, &count);
        int4* tk = &tasks[(size_t)b * kPipeTaskStride];
        std::vector<int> wd(count), res(count);
        for (int j = 0; j < count; j++) wd[j] = P.slice_off[first + j + 1] - P.slice_off[first + j];
        // LDS-resident slots: equal shares (the plain kernel's deal), or by width: the lowest level L with sum_j max(0, wd_j - L) <= the
        // workgroup's wavefront-slots -- every slice then streams min(wd, L) slots -- and what is left over one more for the first slices that
        // still stream.  First with the helpers' hand-over area set aside; a workgroup that gets no helper is dealt again with all of it.
        auto deal_lds = [&](int lds_slots) {
          if (even_share) {
            // (most slots of a slice in LDS: the unroll bound of the plain kernel, 6 -- or FEMBRAIN_PIPE_LDS_CAP, development: further groups of
            // six run in the table-driven kernel's second loop)
            const int cap = getenv("FEMBRAIN_PIPE_LDS_CAP") ? std::max(1, atoi(getenv("FEMBRAIN_PIPE_LDS_CAP"))) : lds_cap;
            const int lbase = std::min(cap, lds_slots / std::max(count, 1)), lrem = lbase < cap ? std::min(count, lds_slots - lbase * count) : 0;
            for (int j = 0; j < count; j++) res[j] = std::min(wd[j], lbase + (j < lrem ? 1 : 0));
            return;
          }
          int L = 0, widest = 0;
          for (int j = 0; j < count; j++) widest = std::max(widest, wd[j]);
          for (L = 0; L <= widest; L++) {
            int sum = 0;
            for (int j = 0; j < count; j++) sum += std::max(0, wd[j] - L);
            if (sum <= lds_slots) break;
          }
          int used = 0;
          for (int j = 0; j < count; j++) { res[j] = std::max(0, wd[j] - L); used += res[j]; }
          for (int j = 0; j < count && used < lds_slots; j++) if (res[j] < wd[j]) { res[j]++; used++; }
        };
        struct Stream { int wave, slice, k0, k1, floor; };
        std::vector<Stream> st;
        int n_h = 0;
        auto deal_streams = [&]() {
          st.clear();
          n_h = 0;
          for (int j = 0; j < kPipeTaskStride; j++) tk[j] = make_int4(-1, 0, 0, 0);
          for (int j = 0; j < count; j++) st.push_back({j, j, 0, wd[j], res[j]});
          std::vector<unsigned> mask(count, 0u);
          for (int hw = count; want_h && hw < w + help_waves && n_h < kPipeMaxHelpers; hw++) {  // idle slice wavefronts first, then the extra ones
            int best = -1, len = 0;
            for (int i = 0; i < (int)st.size(); i++) {
              const int l = st[i].k1 - std::max(st[i].k0, st[i].floor);  // what it streams
              if (l > len) { len = l; best = i; }
            }
            if (best < 0 || len < min_len) break;
            const int lo = std::max(st[best].k0, st[best].floor), mid = lo + (st[best].k1 - lo + 1) / 2;
            const Stream up = {hw, st[best].slice, mid, st[best].k1, mid};
            st[best].k1 = mid;
            st.push_back(up);
            tk[hw] = make_int4(up.slice, up.k0, up.k1, n_h);
            mask[up.slice] |= 1u << n_h;  // the owner adds this helper's partial sums
            n_h++;
          }
          for (const Stream& S : st) {
            if (S.wave < count) tk[S.wave] = make_int4(res[S.wave], 0, S.k1, (int)mask[S.wave]);  // owner: resident slots, (their place: below), end of its own stream, its helpers
            else { tk[S.wave].y = S.k0; tk[S.wave].z = S.k1; }                                   // (ends moved by later splits)
          }
          int at = 0;
          for (int j = 0; j < count; j++) { tk[j].y = at; at += res[j]; }
        };
        deal_lds(lds_all - lds_help);
        deal_streams();
        if (n_h == 0) { deal_lds(lds_all); deal_streams(); }
        for (int j = 0; j < count; j++) deepest = std::max(deepest, res[j]);
        if (const char* dbg = getenv("FEMBRAIN_PIPE_HELP_DEBUG")) {  // development: 1 = owners keep their whole slice, helpers get empty ranges (their zero sums are still added); 2 = owners keep it all and add nothing, helpers work for nothing
          for (int j = 0; j < count; j++) { tk[j].z = wd[j]; if (atoi(dbg) == 2) tk[j].w = 0; }
          if (atoi(dbg) == 1) for (int hw = count; hw < kPipeTaskStride; hw++) if (tk[hw].x >= 0) tk[hw].y = tk[hw].z;
        }
        if (const char* tr = getenv("FEMBRAIN_PIPE_TRUNCATE"))  // development, WRONG RESULTS: owners stop n slots short of their slice -- what an iteration would cost with that many fewer streamed slots per slice (half storage)
          for (int j = 0; j < count; j++) tk[j].z = std::max(tk[j].x, tk[j].z - atoi(tr));
        most = std::max(most, n_h);
        all += n_h;
      }
      if (all > 0 || deepest > 6 || want_xyz || (eh && atoi(eh) == 2)) {  // (=2, development: the task table without a single helper)
        FB_TRY(h->pipe_tasks.upload(tasks, s));
        h->pipe_wmax = 12;
        h->pipe_klt = deepest;
        h->pipe_help_waves = help_waves; h->pipe_n_help = most; h->pipe_help_tasks = all;
        h->pipe_xyz = want_xyz;
        if (getenv("FEMBRAIN_TIMING")) fprintf(stderr, "[fembrain] persistent solver: %d helper tasks (at most %d per workgroup), widest slice %d slots, mean %.1f, up to %d slots of a slice in LDS; a slot's gathers touch %.1f lines in planes, %.1f node by node -> %s\n", all, most, mx, mean, deepest, h->pipe_gather_lines[0], h->pipe_gather_lines[1], want_xyz ? "node by node" : "planes");
      }
    }
  }
  FB_TRY(h->pipe_post.alloc((size_t)2 * nb * 4));
  zb.add(h->pipe_post);
  h->pipe_flag_extra = P.n_ranks > 1 ? kP2PMaxRanks : 0;
  FB_TRY(h->pipe_flags.alloc((size_t)nb + h->pipe_flag_extra + 16));
  zb.add(h->pipe_flags);
  const size_t n_pad = (size_t)P.n_slices * 64, nv = (size_t)3 * P.n_local + 2;
  FB_TRY(h->pipe_planes.alloc(2 * 3 * n_pad));
  zb.add(h->pipe_planes);
  FB_TRY(h->pipe_z.alloc(nv));
  FB_TRY(h->pipe_s.alloc(nv));
  FB_TRY(h->pipe_state.alloc(2));
  zb.add(h->pipe_state);
  if (getenv("FEMBRAIN_PERSIST_TIMING")) {
    FB_TRY(h->persist_timing.alloc((size_t)nb * kPipeMaxWaves * 6));
    FB_TRY(h->persist_timing.zero(s));
  } else {
    h->persist_timing.release();
  }
  // producer lists: the workgroups that own the rows this workgroup's columns lie in, exactly (k_slice_producers)
  static_assert(kPipeMaxBlocks <= 256, "k_slice_producers holds 256 workgroups in its 8 mask words");
  const bool poll_all = getenv("FEMBRAIN_PERSIST_POLL_ALL") && atoi(getenv("FEMBRAIN_PERSIST_POLL_ALL")) != 0;  // development aid
  FB_TRY(h->pipe_owner.alloc((size_t)std::max(1, P.n_slices)));
  FB_TRY(h->pipe_mask.alloc((size_t)std::max(1, P.n_slices) * 8));
  FB_TRY(h->pipe_prod.alloc((size_t)nb * kPipeMaxProducers));
  FB_TRY(h->pipe_prod_count.alloc((size_t)nb));
  FB_TRY(h->pipe_prod_xcd.alloc((size_t)nb));
  FB_TRY(h->pipe_stats.alloc(2));
  zb.add(h->pipe_stats);
  FB_TRY(zb.flush());
  hipLaunchKernelGGL(k_slice_owner, dim3(ceil_div(nb, kBlock)), dim3(kBlock), 0, s, P.n_slices, nb, h->pipe_wg_first.p, h->pipe_owner.p);
  hipLaunchKernelGGL(k_slice_producers, dim3(ceil_div(std::max(1, P.n_slices), kWavesPerBlock)), dim3(kBlock), 0, s, P.n_slices, P.n_owned, h->slice_off.p, h->colidx.p,
                     h->pipe_owner.p, h->pipe_mask.p);
  hipLaunchKernelGGL(k_wg_producers, dim3(ceil_div(nb, kBlock)), dim3(kBlock), 0, s, P.n_slices, nb, h->pipe_wg_first.p, h->pipe_mask.p, poll_all ? 1 : 0, h->pipe_prod.p, h->pipe_prod_count.p,
                     h->pipe_prod_xcd.p, h->pipe_stats.p);
  FB_HIP(hipGetLastError());
  h->pipe_stats_pending = true;  // (the longest list is fetched when fb_fem_pcg_path asks)
  h->pipe_max_producers = 0;
  FB_TRY(h->pipe_xcc.alloc((size_t)nb));
  zb.add(h->pipe_xcc);
  FB_TRY(zb.flush());
  {
    // interior workgroups publish with plain stores where the iteration is latency-bound (measured, us per iteration with / without:
    // 9.5 / 10.2 at 466 slices, 9.5 / 9.9 at 1,000; 17.4 / 17.1 at 2,744 = 1M tets, where the matrix stream evicts the lines from
    // L2 anyway): up to 8 slices per CU.  FEMBRAIN_PIPE_PLAIN_STORES=0/1 overrides.
    const char* e = getenv("FEMBRAIN_PIPE_PLAIN_STORES");
    h->pipe_plain_local = e ? atoi(e) : (w <= 8 ? 1 : 0);
  }
  return FB_OK;
}

int upload_plan(fb_fem_s* h, const double* xyz_global, const float* xyz_device = nullptr, const double* xyz_device64 = nullptr) {
  const FemPlan& P = h->plan;
  hipStream_t s = h->stream;
  if (!h->device_plan) {  // (the device builder has put the tets and the plan arrays in place already)
    std::vector<int4> t4(P.n_tets);
    for (int e = 0; e < P.n_tets; e++) t4[e] = make_int4(P.tets[4 * (size_t)e], P.tets[4 * (size_t)e + 1], P.tets[4 * (size_t)e + 2], P.tets[4 * (size_t)e + 3]);
    FB_TRY(h->tets.upload(t4, s));
    FB_TRY(h->slice_off.upload(P.slice_off, s));
    FB_TRY(h->colidx.upload(P.colidx, s));
    FB_TRY(h->slot_coff.upload(P.slot_coff, s));
    FB_TRY(h->slot_ccnt.upload(P.slot_ccnt, s));
    FB_TRY(h->contrib.upload(P.contrib.data(), P.contrib.size(), s));
  }
  if (h->x0_ready) {
    // (renumbered: gathered into the internal order by build_plan_on_device)
  } else if (xyz_device64) {  // (fb_fem_resync_delta: the caller-order positions, already on this device)
    FB_TRY(h->x0.alloc((size_t)3 * P.n_local));
    FB_HIP(hipMemcpyAsync(h->x0.p, xyz_device64, sizeof(double) * 3 * (size_t)P.n_local, hipMemcpyDeviceToDevice, s));
  } else if (xyz_device) {  // mesh handed over on the device (fb_fem_create_from_poly): float positions widened in place
    const long long n3 = 3LL * P.n_local;
    FB_TRY(h->x0.alloc((size_t)n3));
    hipLaunchKernelGGL(k_widen_positions, dim3((unsigned)((n3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n3, xyz_device, h->x0.p);
    FB_HIP(hipGetLastError());
  } else {
    if (P.n_ranks == 1) {  // identity numbering
      FB_TRY(h->x0.upload(xyz_global, (size_t)3 * P.n_local, s));
    } else {
      std::vector<double>& x0 = h->x0_stage;
      x0.resize((size_t)3 * P.n_local);
      const bool mapped = !h->l2c.empty();  // (renumbered: the owned nodes are a range of the INTERNAL order, anywhere in the caller's)
      if (!mapped) memcpy(x0.data(), xyz_global + 3 * (size_t)P.node_lo, sizeof(double) * 3 * (size_t)P.n_owned);  // owned nodes are a contiguous global range
      for (int l = mapped ? 0 : P.n_owned; l < P.n_local; l++)
        for (int k = 0; k < 3; k++) x0[3 * (size_t)l + k] = xyz_global[3 * (size_t)(mapped ? h->l2c[l] : P.local2global[l]) + k];
      FB_TRY(h->x0.upload(x0, s));
    }
  }
  FB_TRY(h->rest.alloc((size_t)16 * P.n_tets));
  FB_TRY(h->volf.alloc((size_t)std::max(1, P.n_tets)));
  FB_TRY(h->fe.alloc((size_t)12 * P.n_tets));
  FB_TRY(h->rec.alloc((size_t)16 * P.n_tets * mt_size(h)));
  if (h->prm.exact_tangent && !h->prm.linear) FB_TRY(h->kcorr.alloc((size_t)144 * P.n_tets * mt_size(h)));
  else h->kcorr.release();
  if (!h->masks_ready) FB_TRY(upload_masks(h));
  if (!P.send_local.empty()) FB_TRY(h->send_local.upload(P.send_local, s));
  FB_TRY(h->sendbuf.alloc(std::max<size_t>(1, (size_t)12 * P.send_local.size())));
  ZeroBatch zb(s);
  FB_TRY(h->vals.alloc((size_t)P.n_slots * 9 * 64 * mt_size(h)));
  zb.add(h->vals);
  FB_TRY(h->dlo.alloc((size_t)P.n_slices * 9 * 64 * mt_size(h)));
  zb.add(h->dlo);
  FB_TRY(h->mblk.alloc((size_t)P.n_slots * 64));
  zb.add(h->mblk);
  {
    // Element-major assembly where the accumulators of the widest slice fit the LDS of a CU (5 KB per slot: up to 32 slots);
    // FEMBRAIN_ASM_KERNEL=rows keeps the slot-major kernel (same result bit for bit, tests/test_fem_gpu.py)
    // (the widest slice the element-major kernels take, and how many are wider: those few go to the slot-major kernel -- one hub node
    // used to send the whole mesh there)
    int mw = 0, n_wide = 0, widest = 0;
    std::vector<int> wide;
    for (int sl = 0; sl < P.n_slices; sl++) {
      const int wsl = P.slice_off[sl + 1] - P.slice_off[sl];
      if (wsl > kIncMaxWidth) { n_wide++; widest = std::max(widest, wsl); wide.push_back(sl); } else mw = std::max(mw, wsl);
    }
    h->asm_wide = n_wide;
    h->asm_wide_slots = widest;
    h->asm_wide_grid = 0;
    if (n_wide > 0) {  // k_assemble_wide: one workgroup per wide slice while their scratch areas stay under 64 MB, fewer (each taking several) beyond
      const size_t area = (size_t)widest * kWideTerms * 64;  // doubles
      h->asm_wide_grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_wide, ((size_t)64 << 20) / (area * sizeof(double))));
      if (const char* wg = getenv("FEMBRAIN_ASM_WIDE_GRID")) h->asm_wide_grid = std::max(1, std::min(h->asm_wide_grid, atoi(wg)));  // (tests: several slices per workgroup)
      FB_TRY(h->wide_list.upload(wide, s));
      FB_TRY(h->wide_scratch.alloc(area * (size_t)h->asm_wide_grid));
    }
    hipDeviceProp_t prop;
    FB_HIP(hipGetDeviceProperties(&prop, h->prm.device));
    const int lds_cu = (int)std::min<size_t>(std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlock), 160 * 1024);  // gfx950: 160 KB per CU
    const char* e = getenv("FEMBRAIN_ASM_KERNEL");
    h->asm_max_width = mw;
    h->asm_lds = (mw * 10 + kAsmExtra) * 64 * (int)sizeof(double);
    h->asm_tets = mw >= 2 && mw <= kIncMaxWidth && h->asm_lds <= lds_cu && h->asm_lds <= (int)prop.sharedMemPerBlock && !(e && !strcmp(e, "rows"));
    if (h->asm_tets) {
      FB_TRY(build_incidence_device(s, P.n_slices, P.n_owned, h->slice_off.p, h->colidx.p, h->slot_coff.p, h->slot_ccnt.p, h->contrib.p, h->tets.p, h->inc_off,
                                    h->inc, h->inc_slot, h->plan_ws, P.n_ranks == 1));
      int per_cu = std::max(1, lds_cu / h->asm_lds);
      if (const char* pc = getenv("FEMBRAIN_ASM_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(pc)));  // development aid
      const int cus = std::max(8, (prop.multiProcessorCount / 8) * 8);
      const int chunk = ceil_div(P.n_slices, 8);
      h->asm_grid = 8 * std::max(1, std::min(chunk, (cus / 8) * per_cu));
      if (getenv("FEMBRAIN_TIMING")) fprintf(stderr, "[fembrain] element-major assembly: %d workgroups, %d B of LDS each (%d per CU; device reports %zu / %zu)\n", h->asm_grid, h->asm_lds, per_cu, (size_t)prop.maxSharedMemoryPerMultiProcessor, (size_t)prop.sharedMemPerBlock);
      const bool tangent = h->prm.exact_tangent && !h->prm.linear;
      const void* kerns[2][2][2] = {{{(const void*)k_assemble_tets<float, 2, false, false>, (const void*)k_assemble_tets<float, 2, false, true>},
                                     {(const void*)k_assemble_tets<float, 2, true, false>, (const void*)k_assemble_tets<float, 2, true, true>}},
                                    {{(const void*)k_assemble_tets<double, 2, false, false>, (const void*)k_assemble_tets<double, 2, false, true>},
                                     {(const void*)k_assemble_tets<double, 2, true, false>, (const void*)k_assemble_tets<double, 2, true, true>}}};
      for (int nm = 0; nm < 2; nm++) {  // (a Newmark handle assembles without qacc too: fb_fem_assemble)
        const void* kern = kerns[h->f64 ? 1 : 0][tangent ? 1 : 0][nm];
        FB_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, h->asm_lds));
      }
      // fp32 records and the plain tangent: the staged form (one record fetch per row and element, mass entries precomputed);
      // FEMBRAIN_ASM_KERNEL=tets1 keeps the unstaged element-major kernel (same bits)
      h->asm_lds_st = (mw * 9 + kAsmExtra) * 64 * (int)sizeof(double) + kAsmStageDoubles * (int)sizeof(double);
      h->asm_staged = !h->f64 && !tangent && h->asm_lds_st <= lds_cu && h->asm_lds_st <= (int)prop.sharedMemPerBlock && !(e && !strcmp(e, "tets1"));
      if (h->asm_staged) {
        int per_cu_st = std::max(1, lds_cu / h->asm_lds_st);
        if (const char* pc = getenv("FEMBRAIN_ASM_PER_CU")) per_cu_st = std::max(1, std::min(per_cu_st, atoi(pc)));
        h->asm_grid_st = 8 * std::max(1, std::min(chunk, (cus / 8) * per_cu_st));
        FB_HIP(hipFuncSetAttribute((const void*)k_assemble_tets_st<false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->asm_lds_st));
        FB_HIP(hipFuncSetAttribute((const void*)k_assemble_tets_st<true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->asm_lds_st));
        FB_HIP(hipFuncSetAttribute((const void*)k_mass_blocks<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (kBlock / 64) * mw * 64 * (int)sizeof(double)));
        if (getenv("FEMBRAIN_TIMING")) fprintf(stderr, "[fembrain] staged element-major assembly: %d workgroups, %d B of LDS each (%d per CU)\n", h->asm_grid_st, h->asm_lds_st, per_cu_st);
      }
    } else {
      h->asm_staged = false;
      h->inc_off.release(); h->inc.release(); h->inc_slot.release();
    }
    h->mass_valid = false;
  }
  if (h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI) {
    if (P.n_ranks > 1) return fail(FB_EINVAL, "FB_PCG_BLOCK_JACOBI is for unsharded handles");
    FB_TRY(h->invblk.alloc((size_t)9 * P.n_local));
  }
  const size_t nv = (size_t)3 * P.n_local + 2;  // spare doubles: the vector kernels walk owned rows in 16-byte pairs
  DevBuf<double>* vecs[] = {&h->q, &h->qvel, &h->fext, &h->fint, &h->rhs, &h->x, &h->r, &h->d, &h->Ad, &h->invdiag, &h->tmp};
  for (auto* v : vecs) {
    FB_TRY(v->alloc(nv));
    zb.add(*v);
  }
  DevBuf<double>* nvecs[] = {&h->qacc, &h->q1, &h->qvel1, &h->qacc1};
  for (auto* v : nvecs) {
    if (h->prm.integrator == FB_INTEGRATOR_NEWMARK) { FB_TRY(v->alloc(nv)); zb.add(*v); }
    else v->release();
  }
  FB_TRY(zb.flush());
  h->pcg_warm = false;
  const int chunk = ceil_div(P.n_slices, 8);
  const int per = std::max(1, std::min(kMaxPartials / 8, ceil_div(chunk, kWavesPerBlock)));
  h->grid = 8 * per;
  const int want = h->prm.spmv_kernel;
  if (want == FB_SPMV_SPLIT && 8 * ceil_div(chunk, 2) > kMaxPartials)
    return fail(FB_EINVAL, "split SpMV needs a mesh (or shard) of <= %d slices, this one has %d", 2 * kMaxPartials, P.n_slices);
  h->split = 0;
  if (want == FB_SPMV_SPLIT || want == 0) {
    if (8 * chunk <= kMaxPartials) h->split = 4;
    // two wavefronts per slice still pay up to ~1,700 slices (us per iteration, split vs rows: 17.5 / 18.1 at 1,158 slices,
    // 19.1 / 19.5 at 1,521, 23.1 / 22.1 at 1,954); an explicit FB_SPMV_SPLIT is honoured up to the partial-sum limit
    else if (8 * ceil_div(chunk, 2) <= kMaxPartials && (want == FB_SPMV_SPLIT || P.n_slices <= 1700)) h->split = 2;
  }
  h->sgrid = h->split == 4 ? 8 * chunk : (h->split == 2 ? 8 * ceil_div(chunk, 2) : h->grid);
  {
    // bytes one PCG iteration moves: matrix values + indices + the 12 vector streams of the two kernels
    const double iter_bytes = (double)P.n_slots * 64 * (9.0 * mt_size(h) + 4.0) + 12.0 * 24.0 * P.n_local;
    const char* e = getenv("FEMBRAIN_SPMV_NT");
    h->spmv_nt = e ? atoi(e) != 0 : iter_bytes > 384.0 * 1024 * 1024;  // measured: -2 % at 341 MB, +11 % at 469 MB, +15 % at 1.1 GB
  }
  // small meshes: one 16-byte pair per thread in the merged vector pass (8.9 vs 9.3 us per iteration at 105k tets; on the 1M-tet
  // mesh the extra blocks cost more in the partial-sum prologue than they save: 29.9 vs 29.1)
  h->vgrid = h->split == 4 ? 8 * std::max(1, ceil_div(chunk * 96, kBlock)) : h->grid;
  // Persistent solver: unsharded handles whose slices fit one wavefront each on the CUs of the device.  Opt-in by
  // fb_fem_params.pcg_variant = FB_PCG_PERSISTENT or FEMBRAIN_PCG_PERSIST=1 (=0 forces it off)
  FB_TRY(setup_persist(h));
  FB_TRY(h->part_a.alloc(3 * kMaxPartials));
  FB_TRY(h->part_b.alloc(kMaxPartials));
  FB_TRY(h->part_c.alloc(3 * kMaxPartials));
  FB_TRY(h->scal.alloc(8));
  FB_TRY(h->st.alloc(1));
  FB_TRY(h->st.zero(s));
  FB_TRY(h->counter.alloc(1));
  if (!h->device_plan) FB_HIP(hipStreamSynchronize(s));  // (a device-built plan: the rest-state check that follows waits for all of it)
  h->system_valid = false;
  return FB_OK;
}

int launch_rest(fb_fem_s* h, int* first_flat = nullptr) {
  h->mass_valid = false;  // (the rest volumes may change)
  const int nt = h->plan.n_tets;
  hipLaunchKernelGGL(k_tet_rest, dim3(ceil_div(nt, kBlock)), dim3(kBlock), 0, h->stream, nt, h->tets.p, h->x0.p, h->rest.p, first_flat, h->volf.p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// halo refresh of a per-node array (`width` doubles per node: 3 for vectors, 12 for PCG records); no-op when unsharded
__global__ __launch_bounds__(kBlock) void k_pack_nodes(int n, int width, const int* __restrict__ ids, const double* __restrict__ v,
                                                       double* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * width) return;
  const int node = i / width, c = i - node * width;
  out[i] = v[(size_t)width * ids[node] + c];
}

int halo_exchange(fb_fem_s* h, double* vec, int width = 3) {
  if (!h->comm || h->comm->n_ranks == 1) return FB_OK;
  const FemPlan& P = h->plan;
  const int ns = (int)P.send_local.size();
  if (h->xch_mode >= FB_XCH_P2P)
    return p2p_halo(h->p2p, width, h->send_local.p, h->send_off_dev.p, P.n_local - P.n_owned, h->halo_off_dev.p, P.n_owned, vec, h->stream);
  if (ns > 0) {
    hipLaunchKernelGGL(k_pack_nodes, dim3(ceil_div(ns * width, kBlock)), dim3(kBlock), 0, h->stream, ns, width, h->send_local.p, vec, h->sendbuf.p);
    FB_HIP(hipGetLastError());
  }
  return comm_exchange_nodes(h->comm, h->sendbuf.p, P.send_off.data(), vec + (size_t)width * P.n_owned, P.halo_off.data(), width, h->stream);
}

template <typename MT>
int launch_warp(fb_fem_s* h, const double* u, double* rot) {
  const int nt = h->plan.n_tets;
  if (h->kcorr.p)
    hipLaunchKernelGGL((k_tet_warp<MT, true>), dim3(ceil_div(nt, kBlock)), dim3(kBlock), 0, h->stream, nt, h->tets.p, h->x0.p, u, h->rest.p, (MT*)h->rec.p, h->fe.p,
                       rot, h->lambda, h->mu, h->prm.linear != 0 ? 1 : 0, (MT*)h->kcorr.p);
  else
    hipLaunchKernelGGL((k_tet_warp<MT, false>), dim3(ceil_div(nt, kBlock)), dim3(kBlock), 0, h->stream, nt, h->tets.p, h->x0.p, u, h->rest.p, (MT*)h->rec.p, h->fe.p,
                       rot, h->lambda, h->mu, h->prm.linear != 0 ? 1 : 0, (MT*)nullptr);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

template <typename MT>
int launch_rows(fb_fem_s* h, const AsmParams& ap, const double* qvel, const double* fext, double* mblk_out, double* fint_out,
                double* rhs, double* invdiag, const double* qacc = nullptr) {
  AsmOut<MT> o;
  o.dofmask = h->dofmask.p; o.nodemask = h->nodemask.p; o.qvel = qvel; o.fext = fext; o.qacc = qacc; o.vals = (MT*)h->vals.p; o.dlo = (MT*)h->dlo.p; o.mblk_out = mblk_out;
  o.fint_out = fint_out; o.rhs = rhs; o.invdiag = invdiag;
  o.res_all = nullptr;
  if (rhs && h->prm.integrator == FB_INTEGRATOR_NEWMARK && h->nm_max_newton > 1) {  // (kept between calls: alloc is a no-op while the size fits)
    FB_TRY(h->res_all.alloc((size_t)3 * h->plan.n_local + 2));
    o.res_all = h->res_all.p;
  }
  o.invblk = invdiag && h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI ? h->invblk.p : nullptr;
  o.mblk_in = nullptr;
  // The few slices too wide for the element-major kernel (hub nodes, hull nodes of a Delaunay mesh): the slot-major kernel on those only,
  // on a stream of its own BESIDE the element-major launch -- one wavefront takes ~0.8 ms for a 59-slot slice (one latency-bound slot after
  // the other) while the element-major kernel does the other 1,300 slices of the 606k-tet probe in 0.19 ms; the two write disjoint slices.
  // (A CU-masked handle has no second stream with the same mask: the pass then follows in order.)
  const bool wide_pass = h->asm_tets && !mblk_out && h->asm_wide > 0;
  bool wide_beside = false;
  if (wide_pass) {
    if (!h->side && h->cu_limit == 0) {
      if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->side = nullptr; }
      for (auto& e : h->ev_side) if (h->side && !e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); }
    }
    wide_beside = h->side && h->ev_side[0] && h->ev_side[1] && !getenv("FEMBRAIN_ASM_WIDE_ROWS");
    if (wide_beside) {
      FB_HIP(hipEventRecord(h->ev_side[0], h->stream));  // (the records and element forces of this assembly are complete)
      FB_HIP(hipStreamWaitEvent(h->side, h->ev_side[0], 0));
      hipLaunchKernelGGL(k_assemble_wide<MT>, dim3(h->asm_wide_grid), dim3(kWideBlock), 0, h->side, sell_view(h), h->wide_list.p, h->asm_wide, h->asm_wide_slots,
                         h->wide_scratch.p, h->slot_coff.p, h->slot_ccnt.p, h->contrib.p, (const MT*)h->rec.p, h->fe.p, o, ap, (const MT*)h->kcorr.p);
      FB_HIP(hipGetLastError());
      FB_HIP(hipEventRecord(h->ev_side[1], h->side));
    }
  }
  if (h->asm_tets && h->asm_staged && std::is_same<MT, float>::value && !h->kcorr.p && !mblk_out) {
    if (!h->mass_valid) {  // once per rebuild of the rest data
      hipLaunchKernelGGL(k_mass_blocks<float>, dim3(ceil_div(h->plan.n_slices, kBlock / 64)), dim3(kBlock), (size_t)(kBlock / 64) * h->asm_max_width * 64 * sizeof(double),
                         h->stream, sell_view(h), h->inc_off.p, h->inc.p, h->inc_slot.p, h->volf.p, ap.rho20, h->asm_max_width, h->mblk.p);
      FB_HIP(hipGetLastError());
      h->mass_valid = true;
    }
    o.mblk_in = h->mblk.p;
    unsigned long long* prof = nullptr;
    if (getenv("FEMBRAIN_ASM_PROFILE")) {
      FB_HIP(hipMalloc((void**)&prof, 16 * sizeof(unsigned long long)));
      FB_HIP(hipMemset(prof, 0, 16 * sizeof(unsigned long long)));
    }
    const AsmOut<float>& of = reinterpret_cast<const AsmOut<float>&>(o);
    if (qacc) hipLaunchKernelGGL(k_assemble_tets_st<true>, dim3(h->asm_grid_st), dim3(kBlock), (size_t)h->asm_lds_st, h->stream, sell_view(h), h->inc_off.p, h->inc.p, h->inc_slot.p,
                                 (const float*)h->rec.p, h->fe.p, of, ap, h->asm_max_width, prof);
    else hipLaunchKernelGGL(k_assemble_tets_st<false>, dim3(h->asm_grid_st), dim3(kBlock), (size_t)h->asm_lds_st, h->stream, sell_view(h), h->inc_off.p, h->inc.p, h->inc_slot.p,
                            (const float*)h->rec.p, h->fe.p, of, ap, h->asm_max_width, prof);
    if (prof) {
      FB_HIP(hipStreamSynchronize(h->stream));
      unsigned long long t[16];
      FB_HIP(hipMemcpy(t, prof, sizeof t, hipMemcpyDeviceToHost));
      for (int w = 0; w < 4; w++)
        fprintf(stderr, "[fembrain] k_assemble_tets_st wavefront %d: elements %.1f us, wait %.1f, algebra %.1f, wait %.1f (mean per workgroup)\n", w,
                t[4 * w] * 0.01 / h->asm_grid_st, t[4 * w + 1] * 0.01 / h->asm_grid_st, t[4 * w + 2] * 0.01 / h->asm_grid_st, t[4 * w + 3] * 0.01 / h->asm_grid_st);
      (void)hipFree(prof);
    }
  } else if (h->asm_tets && !mblk_out) {  // (the per-block mass read-back, fb_fem_mass, goes through the slot-major kernel)
    constexpr int G = 2;  // list rows whose records are in flight per lane (measured at 1M tets, fp32: 224 us with 4, 209 with 2, 211 with 1)
    unsigned long long* prof = nullptr;
    if (getenv("FEMBRAIN_ASM_PROFILE")) {
      FB_HIP(hipMalloc((void**)&prof, 16 * sizeof(unsigned long long)));
      FB_HIP(hipMemset(prof, 0, 16 * sizeof(unsigned long long)));
    }
#define FB_ASM_TETS(TANGENT, NEWMARK)                                                                                                                        \
  hipLaunchKernelGGL((k_assemble_tets<MT, G, TANGENT, NEWMARK>), dim3(h->asm_grid), dim3(kBlock), (size_t)h->asm_lds, h->stream, sell_view(h), h->inc_off.p, \
                     h->inc.p, h->inc_slot.p, (const MT*)h->rec.p, h->fe.p, o, ap, (const MT*)h->kcorr.p, h->asm_max_width, prof)
    if (h->kcorr.p) { if (qacc) FB_ASM_TETS(true, true); else FB_ASM_TETS(true, false); }
    else { if (qacc) FB_ASM_TETS(false, true); else FB_ASM_TETS(false, false); }
#undef FB_ASM_TETS
    if (prof) {  // FEMBRAIN_ASM_PROFILE=1: where the wavefronts of k_assemble_tets spend their time (100 MHz ticks summed over the workgroups)
      FB_HIP(hipStreamSynchronize(h->stream));
      unsigned long long t[16];
      FB_HIP(hipMemcpy(t, prof, sizeof t, hipMemcpyDeviceToHost));
      for (int w = 0; w < 4; w++)
        fprintf(stderr, "[fembrain] k_assemble_tets wavefront %d: elements %.1f us, wait %.1f, algebra %.1f, wait %.1f (mean per workgroup)\n", w,
                t[4 * w] * 0.01 / h->asm_grid, t[4 * w + 1] * 0.01 / h->asm_grid, t[4 * w + 2] * 0.01 / h->asm_grid, t[4 * w + 3] * 0.01 / h->asm_grid);
      (void)hipFree(prof);
    }
  } else {
    hipLaunchKernelGGL(k_assemble_rows<MT>, dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), h->slot_coff.p, h->slot_ccnt.p,
                       h->contrib.p, (const MT*)h->rec.p, h->fe.p, o, ap, (const MT*)h->kcorr.p, 0);
  }
  if (wide_pass && wide_beside) {
    FB_HIP(hipStreamWaitEvent(h->stream, h->ev_side[1], 0));
  } else if (wide_pass) {
    AsmOut<MT> ow = o;
    ow.mblk_in = nullptr;
    if (getenv("FEMBRAIN_ASM_WIDE_ROWS"))  // development aid: the one-wavefront-per-slice form (same bits)
      hipLaunchKernelGGL(k_assemble_rows<MT>, dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), h->slot_coff.p, h->slot_ccnt.p,
                         h->contrib.p, (const MT*)h->rec.p, h->fe.p, ow, ap, (const MT*)h->kcorr.p, h->asm_max_width);
    else
      hipLaunchKernelGGL(k_assemble_wide<MT>, dim3(h->asm_wide_grid), dim3(kWideBlock), 0, h->stream, sell_view(h), h->wide_list.p, h->asm_wide, h->asm_wide_slots,
                         h->wide_scratch.p, h->slot_coff.p, h->slot_ccnt.p, h->contrib.p, (const MT*)h->rec.p, h->fe.p, ow, ap, (const MT*)h->kcorr.p);
  }
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// pass 1 + pass 2 of the system of the current state (Keff, rhs, invdiag)
int assemble_system(fb_fem_s* h) {
  const double hh = h->prm.timestep, cM = h->prm.damping_mass, cK = h->prm.damping_stiffness;
  FB_TRY(halo_exchange(h, h->q.p));
  FB_TRY(halo_exchange(h, h->qvel.p));
  AsmParams ap;
  ap.lambda = h->lambda; ap.mu = h->mu; ap.rho20 = h->prm.rho / 20.0;
  const double* qacc = nullptr;
  if (h->prm.integrator == FB_INTEGRATOR_NEWMARK) {
    // implicitNewmarkSparse.cpp:218-236: K_eff = K + alpha4 (cK K + cM M) + alpha1 M;  residual = -(M qaccel + (cK K + cM M) qvel + f_int - f_ext)
    const double a1 = 1.0 / (h->nm_beta * hh * hh), a4 = h->nm_gamma / (h->nm_beta * hh);
    FB_TRY(halo_exchange(h, h->qacc.p));
    ap.s_k = 1.0 + a4 * cK; ap.s_m = a4 * cM + a1;
    ap.g_k = cK; ap.g_m = cM; ap.g_a = 1.0; ap.rhs_scale = -1.0;
    qacc = h->qacc.p;
  } else {
    ap.s_k = hh * (hh + cK); ap.s_m = 1.0 + hh * cM;   // Keff = M + h D + h^2 K, D = cK K + cM M
    ap.g_k = hh + cK; ap.g_m = cM;                     // (h K + D) qvel
    ap.g_a = 0.0; ap.rhs_scale = -hh;
  }
  ap.apply_mask = 1;
  if (h->f64) {
    FB_TRY(launch_warp<double>(h, h->q.p, nullptr));
    FB_TRY(launch_rows<double>(h, ap, h->qvel.p, h->fext.p, nullptr, h->fint.p, h->rhs.p, h->invdiag.p, qacc));
  } else {
    FB_TRY(launch_warp<float>(h, h->q.p, nullptr));
    FB_TRY(launch_rows<float>(h, ap, h->qvel.p, h->fext.p, nullptr, h->fint.p, h->rhs.p, h->invdiag.p, qacc));
  }
  h->system_valid = true;
  return FB_OK;
}

template <typename MT, int MODE>
int launch_spmv(fb_fem_s* h, const double* x, double* y, const double* b, double* partial, int parity) {
  if (h->split == 4) {
    hipLaunchKernelGGL((k_spmv_split<MT, MODE, 4>), dim3(h->sgrid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  if (h->split == 2) {
    hipLaunchKernelGGL((k_spmv_split<MT, MODE, 2>), dim3(h->sgrid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  if (h->c16 == 2 && h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, MODE, 0, true, 2>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  else if (h->c16 == 2)
    hipLaunchKernelGGL((k_spmv<MT, MODE, 0, false, 2>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  else if (h->c16 && h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, MODE, 0, true, 1>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  else if (h->c16)
    hipLaunchKernelGGL((k_spmv<MT, MODE, 0, false, 1>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  else if (h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, MODE, 0, true>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  else
    hipLaunchKernelGGL((k_spmv<MT, MODE>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y, b,
                       h->invdiag.p, partial, h->st.p, parity, P2PArgs());
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// the merged-iteration SpMV of a sharded handle on the peer-to-peer transport: its last block posts the three sums;
// XCH = 2 also gathers the halo columns from the inbox (sent by the neighbours' previous vector pass)
template <typename MT, int XCH>
int launch_spmv_xch(fb_fem_s* h, const double* x, double* y, const double* b, double* partial, int parity, const P2PArgs& pa) {
  if (h->split == 4) {
    hipLaunchKernelGGL((k_spmv_split<MT, 3, 4, XCH>), dim3(h->sgrid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, pa);
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  if (h->split == 2) {
    hipLaunchKernelGGL((k_spmv_split<MT, 3, 2, XCH>), dim3(h->sgrid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x,
                       y, b, h->invdiag.p, partial, h->st.p, parity, pa);
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  if (h->c16 == 2 && h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH, true, 2>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, pa);
  else if (h->c16 == 2)
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH, false, 2>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, pa);
  else if (h->c16 && h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH, true, 1>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, pa);
  else if (h->c16)
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH, false, 1>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, pa);
  else if (h->spmv_nt)
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH, true>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y,
                       b, h->invdiag.p, partial, h->st.p, parity, pa);
  else
    hipLaunchKernelGGL((k_spmv<MT, 3, XCH>), dim3(h->grid), dim3(kBlock), 0, h->stream, sell_view(h), (const MT*)h->vals.p, (const MT*)h->dlo.p, x, y, b,
                       h->invdiag.p, partial, h->st.p, parity, pa);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

template <int MODE>
int spmv(fb_fem_s* h, const double* x, double* y, const double* b, double* partial, int parity) {
  return h->f64 ? launch_spmv<double, MODE>(h, x, y, b, partial, parity) : launch_spmv<float, MODE>(h, x, y, b, partial, parity);
}

// multi-GPU: fold the per-block partials into one scalar and all-reduce it; returns the device scalar pointer
// the consumer kernels should read, or nullptr (single GPU: consumers sum the partials themselves)
__global__ __launch_bounds__(kBlock) void k_fold_partials(const double* partial, int n, int count, double* out, const CGState* st) {
  __shared__ double lds[4];
  if (st && st->done) return;
  for (int c = 0; c < count; c++) {
    const double s = sum_partials(partial + (size_t)c * n, n, lds);
    if (threadIdx.x == 0) out[c] = s;
  }
}

// n = number of per-block partials per sum: h->grid for the vector kernels, h->sgrid for the SpMV launches (they differ when
// the split SpMV runs)
int global_scalar(fb_fem_s* h, const double* partial, double** out, bool check_done, int count = 1, int slot = 0, int n = -1) {
  *out = nullptr;
  if (!h->comm || (!h->comm->nccl && !h->comm->local)) return FB_OK;
  if (n < 0) n = h->grid;
  if (h->xch_mode >= FB_XCH_P2P) {  // fold + exchange + rank-ordered sum in one single-block kernel
    FB_TRY(p2p_reduce(h->p2p, partial, n, count, h->scal.p + slot, h->stream));
    *out = h->scal.p;
    return FB_OK;
  }
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(kBlock), 0, h->stream, partial, n, count, h->scal.p + slot, check_done ? h->st.p : nullptr);
  FB_HIP(hipGetLastError());
  // a converged solve leaves the previous (identical on every rank) values in place; the all-reduce still runs on
  // every rank so the collective sequence stays matched, and its result is ignored by the done-checking consumers
  FB_TRY(comm_allreduce_sum(h->comm, h->scal.p + slot, count, h->stream));
  *out = h->scal.p;  // consumers index the scalar block themselves ([0..2] pending sums, [3] exact rho)
  return FB_OK;
}

int pcg_iteration(fb_fem_s* h, int it, const double* b) {
  const FemPlan& P = h->plan;
  const int variant = h->literal_now ? FB_PCG_REFERENCE : h->prm.pcg_variant;  // (a tight tolerance runs the literal sequence, see pcg_solve)
  const int parity = (it - 1) & 1;
  const bool refresh = (it % 30 == 0);
  double* sc = nullptr;
  if (variant == FB_PCG_BLOCK_JACOBI) {  // literal sequence, z = B^-1 r (unsharded: the partials are summed by the consumers)
    FB_TRY(spmv<1>(h, h->d.p, h->Ad.p, nullptr, h->part_a.p, parity));
    if (!refresh) {
      hipLaunchKernelGGL(k_bj_update<false>, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_a.p, h->sgrid, h->d.p,
                         h->Ad.p, h->invblk.p, h->x.p, h->r.p, h->part_b.p);
    } else {
      hipLaunchKernelGGL(k_bj_update<true>, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_a.p, h->sgrid, h->d.p,
                         h->Ad.p, h->invblk.p, h->x.p, h->r.p, h->part_b.p);
      FB_TRY(spmv<2>(h, h->x.p, h->r.p, b, h->part_c.p, parity));  // r = b - A x (its Jacobi-weighted sum is not used)
      hipLaunchKernelGGL(k_bj_rho, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, h->r.p, h->invblk.p, h->part_b.p);
    }
    hipLaunchKernelGGL(k_bj_direction, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_b.p, h->grid, h->r.p,
                       h->invblk.p, h->d.p);
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  if (!refresh && (variant == FB_PCG_MERGED || variant == FB_PCG_PERSISTENT) && h->xch_mode >= FB_XCH_P2P_SUMS) {
    // Peer-to-peer transport with the exchanges inside the iteration's own kernels: the vector pass posts (block 0) and
    // awaits the three sums in its prologue; in FB_XCH_P2P_FUSED the SpMV also refreshes the halo in its prologue (a block
    // per neighbour sends, every block waits and gathers halo columns from the inbox) -- two launches, as on one GPU.
    P2PArgs pa = p2p_next_sum(h->p2p);
    pa.send_ids = h->send_local.p; pa.send_off = h->send_off_dev.p; pa.halo_off = h->halo_off_dev.p; pa.slice_halo = h->slice_halo.p;
    if (h->xch_mode == FB_XCH_P2P_FUSED) {
      pa.halo_seq = p2p_next_halo(h->p2p);
      FB_TRY(h->f64 ? (launch_spmv_xch<double, 2>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, parity, pa))
                    : (launch_spmv_xch<float, 2>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, parity, pa)));
    } else {
      FB_TRY(halo_exchange(h, h->d.p));
      FB_TRY(spmv<3>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, parity));
    }
    hipLaunchKernelGGL((k_cg_fused<true>), dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_a.p, h->sgrid,
                       (const double*)nullptr, h->Ad.p, h->invdiag.p, h->x.p, h->r.p, h->d.p, pa);
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  FB_TRY(halo_exchange(h, h->d.p));
  if (!refresh && (variant == FB_PCG_MERGED || variant == FB_PCG_PERSISTENT)) {
    // merged-reduction iteration: SpMV with the three sums, then one fused vector pass (one reduction / all-reduce)
    FB_TRY(spmv<3>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, parity));
    FB_TRY(global_scalar(h, h->part_a.p, &sc, true, 3, 0, h->sgrid));
    hipLaunchKernelGGL((k_cg_fused<false>), dim3(h->vgrid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_a.p, h->sgrid,
                       sc, h->Ad.p, h->invdiag.p, h->x.p, h->r.p, h->d.p, P2PArgs());
    FB_HIP(hipGetLastError());
    return FB_OK;
  }
  FB_TRY(spmv<1>(h, h->d.p, h->Ad.p, nullptr, h->part_a.p, parity));
  FB_TRY(global_scalar(h, h->part_a.p, &sc, true, 1, 0, h->sgrid));
  if (!refresh) {
    hipLaunchKernelGGL(k_cg_update<false>, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity,
                       h->part_a.p, h->sgrid, sc, h->d.p, h->Ad.p, h->invdiag.p, h->x.p, h->r.p, h->part_b.p);
    FB_HIP(hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_cg_update<true>, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity,
                       h->part_a.p, h->sgrid, sc, h->d.p, h->Ad.p, h->invdiag.p, h->x.p, h->r.p, h->part_b.p);
    FB_HIP(hipGetLastError());
    FB_TRY(halo_exchange(h, h->x.p));
    FB_TRY(spmv<2>(h, h->x.p, h->r.p, b, h->part_b.p, parity));
  }
  FB_TRY(global_scalar(h, h->part_b.p, &sc, true, 1, 0, refresh ? h->sgrid : h->grid));
  // part_b comes from the exact-residual SpMV on refresh iterations, from the vector kernel otherwise
  hipLaunchKernelGGL(k_cg_direction, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->st.p, parity, h->part_b.p,
                     refresh ? h->sgrid : h->grid, sc, h->r.p, h->invdiag.p, h->d.p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

void drop_graph(fb_fem_s* h) {
  if (h->batch_graph) (void)hipGraphExecDestroy(h->batch_graph);
  h->batch_graph = nullptr;
  h->graph_rhs = nullptr;
}

int pcg_iteration(fb_fem_s* h, int it, const double* b);

// captures iterations 1..30 (parity and the position of the exact-residual iteration repeat with period 30)
int ensure_batch_graph(fb_fem_s* h, const double* b, int batch) {
  if (h->batch_graph && h->graph_rhs == b && h->graph_literal == h->literal_now) return FB_OK;
  drop_graph(h);
  h->graph_literal = h->literal_now;
  hipGraph_t g = nullptr;
  FB_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = FB_OK;
  for (int k = 1; k <= batch && rc == FB_OK; k++) rc = pcg_iteration(h, k, b);
  const hipError_t e = hipStreamEndCapture(h->stream, &g);
  if (rc != FB_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess || !g) return fail(FB_EDEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
  const hipError_t ei = hipGraphInstantiate(&h->batch_graph, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ei != hipSuccess) { h->batch_graph = nullptr; return fail(FB_EDEVICE, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
  h->graph_rhs = b;
  return FB_OK;
}

bool host_finished(const CGState& s) {
  if (s.done) return true;
  const double rho = s.rho[s.iter & 1];
  return !(rho > s.eps2 * s.rho0) || s.iter >= s.max_iter;
}

// Jacobi-PCG on the assembled system, rhs b -> h->x.  iters_out: + converged / - not (CGSolver.cpp:189).

// One launch of the pipelined persistent solver (pcg_pipe.hip.h): `start` 1 = new solve from x = 0, 2 = new solve from the x in
// memory, 0 = continue; at most n_iters iterations
// 9 or 10 slices per CU in the plain 12-wavefront kernel: the LDS has room for 7 slots of a slice (65 / 9), one more than the (12, 6)
// instantiation's unroll bound takes -- the (12, 7) instantiation is the same kernel with that bound (unsharded Jacobi handles without the
// task table; FEMBRAIN_PIPE_KLT7=0 keeps (12, 6))
static bool pipe_klt7(const fb_fem_s* h) {
  const char* e = getenv("FEMBRAIN_PIPE_KLT7");
  return h->persist && !h->shard_persist && h->pipe_rows == 1 && h->pipe_wmax == 12 && !h->pipe_tasks.p && h->prm.pcg_variant != FB_PCG_BLOCK_JACOBI &&
         (h->persist_waves == 9 || h->persist_waves == 10) && !getenv("FEMBRAIN_PERSIST_TIMING") && !(e && atoi(e) == 0);
}
int launch_pipe(fb_fem_s* h, const double* b, int start, int n_iters, double eps, int max_iter) {
  PipeArgs pa;
  pa.post = h->pipe_post.p; pa.flags = h->pipe_flags.p; pa.error = h->pipe_flags.p + h->persist_blocks + h->pipe_flag_extra + 4; pa.seqs = h->pipe_flags.p + h->persist_blocks + h->pipe_flag_extra + 8;
  pa.producers = h->pipe_prod.p; pa.prod_count = h->pipe_prod_count.p; pa.prod_xcd = h->pipe_prod_xcd.p; pa.plain_local = h->pipe_plain_local; pa.xcc = h->pipe_xcc.p;
  pa.start = start; pa.n_iters = n_iters; pa.eps2 = eps * eps; pa.max_iter = max_iter;
  pa.timeout_ticks = h->persist_timeout_ticks;
  pa.timing = h->persist_timing.p;
  pa.planes = h->pipe_planes.p; pa.n_pad = h->shard_persist ? (size_t)ceil_div(h->plan.n_local, 64) * 64 : (size_t)h->plan.n_slices * 64;
  pa.pstate = h->pipe_state.p;
  pa.tasks = h->pipe_tasks.p; pa.n_help = h->pipe_n_help;
  pa.wg_first = h->pipe_wg_first.p;
  // LDS: the sync buffers, then KLT slots of every slice; the request is the whole 160 KB of a CU, so exactly one workgroup lands on each
  const size_t lds = 160 * 1024;
  // one wavefront more than slices where the instantiation has room: it collects the sums while the others multiply
  const bool want_service = !(getenv("FEMBRAIN_PIPE_SERVICE_WAVE") && atoi(getenv("FEMBRAIN_PIPE_SERVICE_WAVE")) == 0);
  const int cwaves = h->pipe_rows == 2 ? ceil_div(h->persist_waves, 2) : h->persist_waves;  // wavefronts that own slices
  pa.service = (h->shard_persist || want_service) && cwaves + h->pipe_help_waves < h->pipe_wmax ? 1 : 0;
  // values of the first streamed slots pulled into L2 during the neighbour wait: pays where the product is bandwidth-bound (9 and more
  // slices per CU: -6 % per iteration at 1M tets; neutral at 1,000 slices).  FEMBRAIN_PIPE_PREFETCH=0..4 overrides.
  const int prefetch = getenv("FEMBRAIN_PIPE_PREFETCH") ? std::max(0, std::min(4, atoi(getenv("FEMBRAIN_PIPE_PREFETCH")))) : -1;
  pa.prefetch_slots = prefetch >= 0 ? prefetch : (h->persist_waves >= 9 ? (h->pipe_rows == 2 ? 3 : 4) : 0);
  const dim3 grid(h->persist_blocks), block(64 * (cwaves + h->pipe_help_waves + pa.service));  // slices | helpers | the service wavefront
  FB_HIP(hipEventRecord(h->ev_p[0], h->stream));
  ShardArgs sa;
  memset(&sa, 0, sizeof sa);
  if (h->shard_persist) {
    sa.rank = h->plan.rank; sa.n_ranks = h->plan.n_ranks; sa.n_owned = h->plan.n_owned; sa.n_halo = h->plan.n_local - h->plan.n_owned;
    sa.box = h->sbox; sa.peer_box = h->sbox_peers.p; sa.peer_seg = h->sh_peer_seg.p; sa.halo_cap = h->sbox_halo_cap;
    sa.halo_off = h->sh_halo_off.p; sa.row_send_off = h->sh_row_send_off.p; sa.row_send_rank = h->sh_row_send_rank.p; sa.row_send_pos = h->sh_row_send_pos.p;
    sa.wg_send_mask = h->sh_wg_send_mask.p; sa.n_senders = h->sh_n_senders.p; sa.proxy_wg = h->sh_proxy_wg.p; sa.n_proxy = h->sh_n_proxy; sa.wg_duty = h->sh_wg_duty.p; sa.wg_range = h->sh_wg_range.p;
    sa.delay_ticks = remote_delay_ticks();
  }
  // (the attribute is per device and cheap to set: set at every launch)
#define FB_PIPE(C16, WMAX, KLT, TIMING, SHARD)                                                                                                        \
  do {                                                                                                                                                \
    FB_HIP(hipFuncSetAttribute((const void*)k_pcg_pipe<float, C16, WMAX, KLT, TIMING, SHARD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_pcg_pipe<float, C16, WMAX, KLT, TIMING, SHARD>), grid, block, lds, h->stream, sell_view(h), (const float*)h->vals.p,       \
                       (const float*)h->dlo.p, h->invdiag.p, b, h->x.p, h->r.p, h->Ad.p, h->pipe_z.p, h->pipe_s.p, h->d.p, h->st.p, pa, sa);          \
  } while (0)
#define FB_PIPE_HELP(C16, TIMING, XYZ)                                                                                                                \
  do {                                                                                                                                                \
    FB_HIP(hipFuncSetAttribute((const void*)k_pcg_pipe<float, C16, 12, 6, TIMING, false, false, true, XYZ>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)lds));                                                                                                           \
    hipLaunchKernelGGL((k_pcg_pipe<float, C16, 12, 6, TIMING, false, false, true, XYZ>), grid, block, lds, h->stream, sell_view(h),                  \
                       (const float*)h->vals.p, (const float*)h->dlo.p, h->invdiag.p, b, h->x.p, h->r.p, h->Ad.p, h->pipe_z.p, h->pipe_s.p, h->d.p, \
                       h->st.p, pa, sa);                                                                                                             \
  } while (0)
#define FB_PIPE_BJ(C16, WMAX, KLT)                                                                                                                     \
  do {                                                                                                                                                \
    FB_HIP(hipFuncSetAttribute((const void*)k_pcg_pipe<float, C16, WMAX, KLT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                               (int)lds));                                                                                                           \
    hipLaunchKernelGGL((k_pcg_pipe<float, C16, WMAX, KLT, false, false, true>), grid, block, lds, h->stream, sell_view(h), (const float*)h->vals.p,  \
                       (const float*)h->dlo.p, h->invblk.p, b, h->x.p, h->r.p, h->Ad.p, h->pipe_z.p, h->pipe_s.p, h->d.p, h->st.p, pa, sa);           \
  } while (0)
#define FB_PIPE2(C16, SHARD, XYZ)                                                                                                                     \
  do {                                                                                                                                                \
    FB_HIP(hipFuncSetAttribute((const void*)k_pcg_pipe2<C16, SHARD, XYZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
    hipLaunchKernelGGL((k_pcg_pipe2<C16, SHARD, XYZ>), grid, block, lds, h->stream, sell_view(h), (const float*)h->vals.p, (const float*)h->dlo.p,   \
                       h->invdiag.p, b, h->x.p, h->r.p, h->Ad.p, h->pipe_z.p, h->pipe_s.p, h->d.p, h->st.p, pa, sa);                                  \
  } while (0)
  static_assert(sizeof(double) * kPipeSyncDoubles + (size_t)kPipeMaxWaves * 2 * kPipe2LdsWordsPerRow * 64 * 4 <= 160 * 1024, "LDS budget of k_pcg_pipe2");
  // the instantiations: (wavefronts, most LDS slots per wavefront) = (8, 8) up to 8 slices per CU, (12, 6) up to 12; 16- or 32-bit column words;
  // two rows per lane (k_pcg_pipe2) for 13..24 slices per CU; SHARD = true on a sharded handle (32-bit local column ids)
  if (h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI) {  // (setup_persist: unsharded, one row per lane)
    if (pa.timing) return fail(FB_EINVAL, "FEMBRAIN_PERSIST_TIMING is built for the Jacobi kernel");
    if (h->pipe_wmax == 8) { if (h->c16) FB_PIPE_BJ(true, 8, 8); else FB_PIPE_BJ(false, 8, 8); }
    else { if (h->c16) FB_PIPE_BJ(true, 12, 6); else FB_PIPE_BJ(false, 12, 6); }
  } else if (h->shard_persist) {
    if (h->pipe_rows == 2) FB_PIPE2(false, true, false);
    else if (h->pipe_wmax == 8) FB_PIPE(false, 8, 8, false, true);
    else FB_PIPE(false, 12, 6, false, true);
  } else if (h->pipe_rows == 2) {
    if (h->pipe_xyz) { if (h->c16) FB_PIPE2(true, false, true); else FB_PIPE2(false, false, true); }
    else { if (h->c16) FB_PIPE2(true, false, false); else FB_PIPE2(false, false, false); }
  } else if (h->pipe_tasks.p) {  // helper wavefronts (setup_persist): the (12, 6) kernel compiled with them
    if (h->pipe_xyz) {  // (the published vector node by node)
      if (pa.timing) { if (h->c16) FB_PIPE_HELP(true, true, true); else FB_PIPE_HELP(false, true, true); }
      else { if (h->c16) FB_PIPE_HELP(true, false, true); else FB_PIPE_HELP(false, false, true); }
    } else {
      if (pa.timing) { if (h->c16) FB_PIPE_HELP(true, true, false); else FB_PIPE_HELP(false, true, false); }
      else { if (h->c16) FB_PIPE_HELP(true, false, false); else FB_PIPE_HELP(false, false, false); }
    }
  } else if (pa.timing) {  // development build with the phase clocks: the 1M-tet configuration and the small one
    if (h->pipe_wmax == 12 && h->c16) FB_PIPE(true, 12, 6, true, false);
    else if (h->pipe_wmax == 12) FB_PIPE(false, 12, 6, true, false);
    else if (h->pipe_wmax == 5 && h->c16) FB_PIPE(true, 5, 16, true, false);
    else if (h->pipe_wmax == 8 && h->c16) FB_PIPE(true, 8, 8, true, false);
    else if (h->pipe_wmax == 8) FB_PIPE(false, 8, 8, true, false);
    else return fail(FB_EINVAL, "FEMBRAIN_PERSIST_TIMING is built for one row per lane: (12, 6) and (5, 16) with 16-bit column words, (8, 8)");
  } else if (h->pipe_wmax == 5) { if (h->c16) FB_PIPE(true, 5, 16, false, false); else FB_PIPE(false, 5, 16, false, false); }
  else if (h->pipe_wmax == 8) { if (h->c16) FB_PIPE(true, 8, 8, false, false); else FB_PIPE(false, 8, 8, false, false); }
  else if (pipe_klt7(h)) { if (h->c16) FB_PIPE(true, 12, 7, false, false); else FB_PIPE(false, 12, 7, false, false); }
  else { if (h->c16) FB_PIPE(true, 12, 6, false, false); else FB_PIPE(false, 12, 6, false, false); }
#undef FB_PIPE2
#undef FB_PIPE_BJ
#undef FB_PIPE_HELP
#undef FB_PIPE
  FB_HIP(hipGetLastError());
  FB_HIP(hipEventRecord(h->ev_p[1], h->stream));
  return FB_OK;
}

void print_pipe_timing(fb_fem_s* h) {
  std::vector<long long> tm((size_t)h->persist_blocks * kPipeMaxWaves * 6);
  if (h->persist_timing.download(tm.data(), tm.size(), h->stream) != FB_OK) return;
  (void)h->persist_timing.zero(h->stream);
  const char* names[5] = {"publish (drained)", "flag+wait+acquire", "product", "sums sweep", "recurrences(+refresh)"};
  for (int k = 0; k < 5; k++) {
    double mn = 1e30, mx = 0, av = 0;
    int cnt = 0;
    for (int b = 0; b < h->persist_blocks; b++)
      for (int w = 0; w < h->persist_waves; w++) {
        const long long* t = &tm[((size_t)b * kPipeMaxWaves + w) * 6];
        if (t[5] <= 0) continue;
        const double us = (double)t[k] / (double)t[5] * 0.01;
        mn = std::min(mn, us); mx = std::max(mx, us); av += us; cnt++;
      }
    fprintf(stderr, "[fembrain] pipelined PCG %-22s per iteration: avg %.2f us  min %.2f  max %.2f (over %d waves)\n", names[k], av / std::max(cnt, 1), mn, mx, cnt);
  }
  if (getenv("FEMBRAIN_PERSIST_TIMING") && atoi(getenv("FEMBRAIN_PERSIST_TIMING")) >= 2) {
    // where the slow products are: by wavefront index, by XCD (workgroup & 7), and the slowest workgroups
    for (int k = 1; k <= 2; k++) {
      fprintf(stderr, "[fembrain] %s by wavefront:", names[k]);
      for (int w = 0; w < h->persist_waves; w++) {
        double av = 0; int cnt = 0;
        for (int b = 0; b < h->persist_blocks; b++) { const long long* t = &tm[((size_t)b * kPipeMaxWaves + w) * 6]; if (t[5] > 0) { av += (double)t[k] / (double)t[5] * 0.01; cnt++; } }
        fprintf(stderr, " %.2f", av / std::max(cnt, 1));
      }
      fprintf(stderr, "\n[fembrain] %s by XCD (avg / max over its workgroups' slowest wavefront):", names[k]);
      for (int x = 0; x < 8; x++) {
        double av = 0, mx = 0; int cnt = 0;
        for (int b = x; b < h->persist_blocks; b += 8) {
          double wmx = 0;
          for (int w = 0; w < h->persist_waves; w++) { const long long* t = &tm[((size_t)b * kPipeMaxWaves + w) * 6]; if (t[5] > 0) wmx = std::max(wmx, (double)t[k] / (double)t[5] * 0.01); }
          av += wmx; mx = std::max(mx, wmx); cnt++;
        }
        fprintf(stderr, " %.2f/%.2f", av / std::max(cnt, 1), mx);
      }
      fprintf(stderr, "\n");
    }
    std::vector<std::pair<double, int>> slow;
    for (int b = 0; b < h->persist_blocks; b++) {
      double wmx = 0;
      for (int w = 0; w < h->persist_waves; w++) { const long long* t = &tm[((size_t)b * kPipeMaxWaves + w) * 6]; if (t[5] > 0) wmx = std::max(wmx, (double)t[2] / (double)t[5] * 0.01); }
      slow.push_back({wmx, b});
    }
    std::sort(slow.begin(), slow.end());
    fprintf(stderr, "[fembrain] slowest product of a workgroup: median %.2f us, fastest", slow[slow.size() / 2].first);
    for (int k = 0; k < 6 && k < (int)slow.size(); k++) fprintf(stderr, " wg%d %.2f", slow[k].second, slow[k].first);
    fprintf(stderr, "; slowest");
    for (int k = 0; k < 12 && k < (int)slow.size(); k++) fprintf(stderr, " wg%d %.2f", slow[slow.size() - 1 - k].second, slow[slow.size() - 1 - k].first);
    fprintf(stderr, "\n");
  }
}

int pcg_solve(fb_fem_s* h, const double* b, double eps, int max_iter, int* iters_out, CGState* final_state, bool allow_persist = true);

// Jacobi-PCG inside persistent launches (normally ONE): CGSolver.cpp:129-190 in its pipelined form
int pcg_solve_pipe(fb_fem_s* h, const double* b, double eps, int max_iter, int* iters_out, CGState* final_state) {
  hipStream_t s = h->stream;
  const bool warm = h->pcg_warm;
  h->pcg_warm = false;
  int cut = max_iter + 1;  // iterations per launch: all of them (the kernel stops at convergence or max_iter)
  if (const char* e = getenv("FEMBRAIN_PERSIST_MAX_RUN")) cut = std::max(1, atoi(e));  // test knob: cut the solve into shorter launches
  int start = warm ? 2 : 1;
  CGState fin;
  memset(&fin, 0, sizeof fin);
  for (;;) {
    FB_TRY(launch_pipe(h, b, start, cut, eps, max_iter));
    h->persist_launches++;
    unsigned int err = 0;
    FB_HIP(hipMemcpyAsync(&h->st_host[0], h->st.p, sizeof(CGState), hipMemcpyDeviceToHost, s));
    FB_HIP(hipMemcpyAsync(&err, h->pipe_flags.p + h->persist_blocks + h->pipe_flag_extra + 4, sizeof err, hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
    {
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->ev_p[0], h->ev_p[1]) == hipSuccess) h->persist_seconds += ms * 1e-3;
    }
    if (h->shard_persist) {  // a time-out on any rank sends every rank to the two-launch solver, together
      std::vector<unsigned int> errs((size_t)h->plan.n_ranks);
      FB_TRY(comm_allgather_bytes(h->comm, &err, errs.data(), sizeof err, s));
      for (unsigned int e2 : errs) err |= e2;
    }
    if (err) {
      // A wait inside the launch gave up: the workgroups were not all resident (the device is shared with another process's
      // kernels?).  The launch wrote nothing back, so the solve is repeated from the same start with a launch per phase, and
      // this handle stays with that; fb_step_info.pcg_path / persist_fallbacks tell the host.
      FB_TRY(h->pipe_flags.zero(s));
      FB_TRY(h->pipe_post.zero(s));
      h->persist = false;
      h->persist_broken = true;
      h->persist_fallbacks++;
      h->clean_solves = 0;
      if (h->persist_fallbacks > 1) h->rearm_after = std::min(h->rearm_after * 2, 1 << 20);
      if (getenv("FEMBRAIN_PERSIST_STRICT") && atoi(getenv("FEMBRAIN_PERSIST_STRICT")) != 0)
        return fail(FB_EDEVICE, "persistent PCG: a wait inside the launch timed out after %.1f ms (the workgroups were not all resident?)", h->persist_timeout_ticks * 1e-5);
      fprintf(stderr, "[fembrain] persistent PCG: a wait timed out after %.1f ms; this handle falls back to the two-launch iteration\n", h->persist_timeout_ticks * 1e-5);
      h->pcg_warm = warm;
      const int rc = pcg_solve(h, b, eps, max_iter, iters_out, final_state, false);
      h->last_pcg_path = FB_PCG_PATH_FALLBACK;
      return rc;
    }
    h->persist_iterations += h->st_host[0].iter - (start == 0 ? fin.iter : 0);
    fin = h->st_host[0];
    if (fin.done) break;
    if (fin.iter > max_iter) return fail(FB_EDEVICE, "internal: persistent PCG ran past max_iter (iter %d)", fin.iter);
    start = 0;
  }
  if (h->persist_timing.p && h->pipe_rows == 1) print_pipe_timing(h);
  h->last_pcg_path = FB_PCG_PATH_PERSISTENT;
  const double rho = fin.rho[fin.iter & 1];
  const bool converged = !(rho > fin.eps2 * fin.rho0);
  if (!converged) {
    // The iteration cap was reached (CGSolver.cpp:189 returns -max_iter with the iterate it has).  The launch left its iterate in d and
    // the start vector in x.  One exact residual tells whether the pipelined recurrences were honest to the end: if the true
    // r . r / diag of the iterate is within a factor of four of what they carried, the system simply needs more iterations and the
    // iterate stands (a system that does not converge cost 2 x max_iter iterations).  Otherwise -- the recurrences
    // stalled or drifted -- the literal recurrences get the last word: the two-launch solver repeats the solve.  ("From the same start"
    // holds for a solve in ONE launch, the product's case; with the test knob FEMBRAIN_PERSIST_MAX_RUN the earlier launches have written
    // their iterate to x already, and the repeat continues from there as a warm start -- same answer to the tolerance, not the same bits.)
    const bool cap_check = !(getenv("FEMBRAIN_PERSIST_CAP_CHECK") && atoi(getenv("FEMBRAIN_PERSIST_CAP_CHECK")) == 0);  // (=0: always repeat; tests)
    if (cap_check && h->prm.pcg_variant != FB_PCG_BLOCK_JACOBI) {  // (block-Jacobi carries r . B^-1 r: not what the product kernel sums)
      FB_TRY(h->st.zero(s));  // done = 0: the product below is not a no-op
      FB_TRY(halo_exchange(h, h->d.p));
      FB_TRY(spmv<2>(h, h->d.p, h->r.p, b, h->part_b.p, 0));
      double* sc = nullptr;
      FB_TRY(global_scalar(h, h->part_b.p, &sc, false, 1, 0, h->sgrid));
      hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, s, h->st.p, h->part_b.p, h->sgrid, sc, eps, max_iter);
      FB_HIP(hipGetLastError());
      CGState chk;
      FB_HIP(hipMemcpyAsync(&chk, h->st.p, sizeof(CGState), hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
      if (std::isfinite(chk.rho0) && chk.rho0 <= 4.0 * rho) {
        FB_HIP(hipMemcpyAsync(h->x.p, h->d.p, sizeof(double) * 3 * (size_t)h->plan.n_local, hipMemcpyDeviceToDevice, s));
        fin.rho[fin.iter & 1] = chk.rho0;  // (the exact one)
        // the device copy of the solver state says what the solve ended with, not what the check's k_cg_begin left there
        h->st_host[1] = fin;
        h->st_host[1].done = 1;
        FB_HIP(hipMemcpyAsync(h->st.p, &h->st_host[1], sizeof(CGState), hipMemcpyHostToDevice, s));
        FB_HIP(hipStreamSynchronize(s));
        if (iters_out) *iters_out = -fin.iter;
        if (final_state) *final_state = fin;
        return FB_OK;
      }
    }
    h->pcg_warm = warm;
    const int rc = pcg_solve(h, b, eps, max_iter, iters_out, final_state, false);
    h->last_pcg_path = FB_PCG_PATH_RESOLVED;
    return rc;
  }
  if (iters_out) *iters_out = converged ? fin.iter : -fin.iter;
  if (final_state) *final_state = fin;
  return FB_OK;
}

// The pipelined recurrences of the persistent solver stop making progress at a relative residual of ~1e-11 on these systems
// (measured, tools/pipelined_pcg_numerics.py and DESIGN.md; the literal recurrences go on below 1e-12): a solve asked for a
// tolerance within three orders of that goes to the two-launch solver, whatever the handle runs otherwise.
constexpr double kPersistMinEps = 1e-8;

int pcg_solve(fb_fem_s* h, const double* b, double eps, int max_iter, int* iters_out, CGState* final_state, bool allow_persist) {
  if (allow_persist && !h->persist && h->persist_broken && !h->shard_persist && h->plan.n_ranks == 1 && h->rearm_after > 0 && h->clean_solves >= h->rearm_after) {
    h->persist = true;  // the set-up of the persistent solver is still in place (flags and sums were cleared when it fell back)
    h->persist_broken = false;
    h->persist_rearms++;
    read_persist_timeout(h);
    fprintf(stderr, "[fembrain] persistent PCG: re-armed after %d two-launch solves\n", h->clean_solves);
  }
  if (allow_persist && h->persist && eps >= kPersistMinEps &&
      (h->prm.pcg_variant == FB_PCG_MERGED || h->prm.pcg_variant == FB_PCG_PERSISTENT || h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI))
    return pcg_solve_pipe(h, b, eps, max_iter, iters_out, final_state);
  if (allow_persist && h->persist_broken) h->clean_solves++;
  h->last_pcg_path = FB_PCG_PATH_TWO_LAUNCH;
  // The merged recurrence for rho (rho' = rho - 2 alpha S1 + alpha^2 S2) loses digits the literal sum r.r/diag keeps: measured on
  // the 17,576-node cube it stalls above 1e-12 where the literal sequence converges.  Tolerances below kPersistMinEps therefore run
  // the literal sequence of FB_PCG_REFERENCE, whatever variant the handle was made with (block-Jacobi is literal already).
  struct LiteralScope { fb_fem_s* h; ~LiteralScope() { h->literal_now = false; } } literal_scope{h};
  h->literal_now = eps < kPersistMinEps && (h->prm.pcg_variant == FB_PCG_MERGED || h->prm.pcg_variant == FB_PCG_PERSISTENT);
  const FemPlan& P = h->plan;
  hipStream_t s = h->stream;
  double* sc = nullptr;
  if (h->pcg_warm) {
    // start from the x the previous solve left (CGSolver.cpp:131-141 with a non-zero x): r = b - A x, d = r / diag
    h->pcg_warm = false;
    FB_TRY(h->st.zero(s));  // done = 0: the SpMV below is not a no-op
    FB_TRY(halo_exchange(h, h->x.p));
    FB_TRY(spmv<2>(h, h->x.p, h->r.p, b, h->part_b.p, 0));
    if (h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI) {  // d = B^-1 r and rho = r . B^-1 r (the SpMV's Jacobi-weighted sum is not used)
      hipLaunchKernelGGL(k_bj_init_warm, dim3(h->grid), dim3(kBlock), 0, s, P.n_slices, P.n_owned, h->r.p, h->invblk.p, h->d.p, h->part_b.p);
      FB_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, s, h->st.p, h->part_b.p, h->grid, (const double*)nullptr, eps, max_iter);
    } else {
      hipLaunchKernelGGL(k_cg_init_warm, dim3(h->grid), dim3(kBlock), 0, s, P.n_slices, P.n_owned, h->r.p, h->invdiag.p, h->d.p);
      FB_HIP(hipGetLastError());
      FB_TRY(global_scalar(h, h->part_b.p, &sc, false, 1, 0, h->sgrid));
      hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, s, h->st.p, h->part_b.p, h->sgrid, sc, eps, max_iter);
    }
  } else if (h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI) {
    hipLaunchKernelGGL(k_bj_init, dim3(h->grid), dim3(kBlock), 0, s, P.n_slices, P.n_owned, b, h->invblk.p, h->x.p, h->r.p, h->d.p, h->part_b.p);
    FB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, s, h->st.p, h->part_b.p, h->grid, (const double*)nullptr, eps, max_iter);
  } else {
    hipLaunchKernelGGL(k_cg_init, dim3(h->grid), dim3(kBlock), 0, s, P.n_slices, P.n_owned, b, h->invdiag.p, h->x.p, h->r.p, h->d.p,
                       h->part_b.p);
    FB_HIP(hipGetLastError());
    FB_TRY(global_scalar(h, h->part_b.p, &sc, false));
    hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, s, h->st.p, h->part_b.p, h->grid, sc, eps, max_iter);
  }
  FB_HIP(hipGetLastError());
  const int kBatch = 30;
  int it = 1, slot = 0;
  bool pending[2] = {false, false};
  CGState fin;
  memset(&fin, 0, sizeof fin);
  bool finished = false;
  // replay pays on small meshes only (us per iteration, replay vs launches: 8.2 vs 9.6 at 22k tets, 9.4 vs 9.8 at 105k,
  // 13.1 vs 12.9 at 257k, 29.4 vs 29.1 at 1M); sharded kernels carry sequence numbers and are launched one by one
  const bool graphable = h->use_graph && h->prm.pcg_variant != FB_PCG_BLOCK_JACOBI && P.n_slices <= 512 && (!h->comm || h->comm->n_ranks == 1);
  while (!finished) {
    const int n = std::min(kBatch, max_iter - it + 1);
    bool replayed = false;
    if (graphable && h->use_graph && n == kBatch && (it - 1) % kBatch == 0) {
      if (ensure_batch_graph(h, b, kBatch) == FB_OK && hipGraphLaunch(h->batch_graph, s) == hipSuccess) {
        replayed = true;
        it += kBatch;
      } else {  // capture or replay unavailable: plain launches from now on
        (void)hipGetLastError();
        drop_graph(h);
        h->use_graph = false;
      }
    }
    if (!replayed) {
      for (int k = 0; k < n; k++, it++) FB_TRY(pcg_iteration(h, it, b));
    }
    FB_HIP(hipMemcpyAsync(&h->st_host[slot], h->st.p, sizeof(CGState), hipMemcpyDeviceToHost, s));
    FB_HIP(hipEventRecord(h->ev_batch[slot], s));
    pending[slot] = true;
    const int prev = slot ^ 1;
    if (pending[prev]) {  // look one batch behind so the queue never drains
      FB_HIP(hipEventSynchronize(h->ev_batch[prev]));
      pending[prev] = false;
      if (host_finished(h->st_host[prev])) finished = true;
    }
    if (it > max_iter) finished = true;
    slot ^= 1;
  }
  FB_HIP(hipStreamSynchronize(s));
  if (h->p2p) FB_TRY(p2p_check(h->p2p, s));
  // the newest snapshot is in the slot written last
  fin = h->st_host[slot ^ 1];
  if (!host_finished(fin)) return fail(FB_EDEVICE, "internal: PCG batches ended without a terminal state (iter %d)", fin.iter);
  const double rho = fin.rho[fin.iter & 1];
  const bool converged = !(rho > fin.eps2 * fin.rho0);
  if (iters_out) *iters_out = converged ? fin.iter : -fin.iter;
  if (final_state) *final_state = fin;
  return FB_OK;
}

// The plan of an unsharded handle, built on the device (plan_device.hip).  The host keeps the scalars, the slice offsets,
// the identity numbering and the constraint mask; the pattern arrays stay on the device until an inspection entry point
// asks for them (ensure_host_pattern).
// tets: host node ids, or -- d_tets non-null -- ids already on this device (the polygonizer's own output: in range by construction)
int build_plan_on_device(fb_fem_s* h, int n_nodes, int n_tets, const int* tets, int n_fixed, const int* fixed, const double* xyz, const uint4* d_tets = nullptr,
                         const float* d_xyz = nullptr, const double* d_xyz64 = nullptr) {
  if (n_nodes <= 0 || n_tets <= 0 || (!tets && !d_tets)) return fail(FB_EINVAL, "empty mesh (%d nodes, %d tets)", n_nodes, n_tets);
  if ((long long)n_tets >= (1LL << 28)) return fail(FB_EINVAL, "too many tets for the packed contribution word");

  static const bool timing = getenv("FEMBRAIN_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[fembrain] device plan: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  FemPlan& P = h->plan;
  P = FemPlan();
  P.n_global = n_nodes; P.n_ranks = 1; P.rank = 0;
  P.splits = {0, n_nodes};
  P.node_lo = 0; P.node_hi = n_nodes;
  P.n_owned = P.n_local = n_nodes; P.n_halo = 0;
  P.local2global.resize(n_nodes);
  for (int l = 0; l < n_nodes; l++) P.local2global[l] = l;
  P.halo_off.assign(2, 0);
  P.send_off.assign(2, 0);
  P.n_tets = n_tets;
  if (d_tets) {
    FB_TRY(h->tets.alloc((size_t)n_tets));
    FB_HIP(hipMemcpyAsync(h->tets.p, d_tets, sizeof(int4) * (size_t)n_tets, hipMemcpyDeviceToDevice, h->stream));
  } else {
    FB_TRY(h->tets.upload((const int4*)tets, (size_t)n_tets, h->stream));
  }
  lap("elements uploaded");
  {
    // the internal node order (renumber.h): decided from the widest element, built from the rest positions
    bool want = false;
    FB_TRY(renumber_decide(h->stream, renumber_mode(h), n_nodes, n_tets, h->tets.p, h->plan_ws, h->ren, &want));
    if (want) {
      FB_TRY(h->xyz_in.reserve((size_t)3 * n_nodes));
      if (d_xyz64) {
        FB_HIP(hipMemcpyAsync(h->xyz_in.p, d_xyz64, sizeof(double) * 3 * (size_t)n_nodes, hipMemcpyDeviceToDevice, h->stream));
      } else if (d_xyz) {
        const long long n3 = 3LL * n_nodes;
        hipLaunchKernelGGL(k_widen_positions, dim3((unsigned)((n3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, n3, d_xyz, h->xyz_in.p);
        FB_HIP(hipGetLastError());
      } else {
        FB_HIP(hipMemcpyAsync(h->xyz_in.p, xyz, sizeof(double) * 3 * (size_t)n_nodes, hipMemcpyHostToDevice, h->stream));
      }
      FB_TRY(renumber_build(h->stream, renumber_mode(h), n_nodes, n_tets, h->tets.p, h->xyz_in.p, h->plan_ws, h->ren));
      if (h->ren.active) {
        FB_TRY(relabel_tets(h->stream, n_tets, h->tets.p, n_nodes, h->ren.d_new_of_old.p));
        FB_TRY(h->x0.alloc((size_t)3 * n_nodes));
        FB_TRY(gather_nodes(h->stream, n_nodes, 3, h->xyz_in.p, h->ren.d_old_of_new.p, h->x0.p));
        h->x0_ready = true;  // (P.local2global -- internal id -> the caller's -- is fetched when an inspection entry point asks: ensure_host_order)
        h->ren_nodes_at_build = n_nodes;
      }
    }
    if (getenv("FEMBRAIN_TIMING"))
      fprintf(stderr, "[fembrain] node order: widest element %d -> %d (%s%s)\n", h->ren.span_before, h->ren.span_after, h->ren.active ? "renumbered" : "caller's order kept",
              h->ren.active && h->ren.sigma ? ", rows sorted by element count inside windows" : "");
  }
  lap("node order");
  // constraint masks on the device, in the internal order (host: 0.23 ms of loops and two uploads at 1M tets)
  FB_TRY(device_constraint_masks(h->stream, n_nodes, n_fixed, fixed, h->ren.active ? h->ren.d_new_of_old.p : nullptr, h->fixed_stage, h->dofmask, h->nodemask));
  P.n_fixed_owned = n_fixed;
  h->masks_ready = true;
  lap("constraints");
  DevicePlan D;
  D.slice_off = &h->slice_off; D.colidx = &h->colidx; D.slot_coff = &h->slot_coff; D.slot_ccnt = &h->slot_ccnt; D.contrib = &h->contrib;
  D.bptr = &h->d_bptr; D.bcol = &h->d_bcol; D.blk_slot = &h->d_blk_slot; D.coldelta = &h->coldelta; D.ucnt_keep = &h->d_ucnt;
  // (the widest element was measured for the node order: FB_RENUMBER_OFF skips that pass and leaves the sort its 64-bit keys)
  const int span = renumber_mode(h) == FB_RENUMBER_OFF ? -1 : (h->ren.active ? h->ren.span_after : h->ren.span_before);
  const int rc = build_plan_device(h->stream, n_nodes, n_tets, h->tets.p, D, h->plan_ws, nullptr, span);
  if (h->plan_ws.bytes() > ((size_t)2 << 30)) h->plan_ws.release();  // kept for the next re-sync only while it is small change (0.8 GB at 1M tets)
  if (rc != FB_OK && D.first_bad_tet >= 0 && tets) {  // say which node, as the host builder does
    for (int k = 0; k < 4; k++) {
      const int id = tets[4 * (size_t)D.first_bad_tet + k];
      if (id < 0 || id >= n_nodes) return fail(FB_EINVAL, "tet %d references node %d outside [0,%d)", D.first_bad_tet, id, n_nodes);
    }
  }
  FB_TRY(rc);
  P.n_blocks = D.n_blocks; P.n_slices = D.n_slices; P.n_slots = D.n_slots; P.n_crows = D.n_crows;
  h->c16 = (D.deltas_fit16 && !(getenv("FEMBRAIN_SPMV_C16") && atoi(getenv("FEMBRAIN_SPMV_C16")) == 0)) ? 1 : 0;
  P.slice_off = D.slice_off_host;
  h->csr_ready = true;
  return FB_OK;
}

// global node ids of a rank's elements -> local ids: owned nodes first, then the halo in ascending global order
__global__ void __launch_bounds__(kBlock) k_tets_to_local(int n_tets, int4* __restrict__ tets, int node_lo, int n_owned, const int* __restrict__ halo, int n_halo) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n_tets) return;
  int4 t = tets[e];
  int* v = &t.x;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int g = v[i];
    if (g >= node_lo && g < node_lo + n_owned) { v[i] = g - node_lo; continue; }
    int lo = 0, hi = n_halo;  // lower bound; g is in the list by construction
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (halo[mid] < g) lo = mid + 1; else hi = mid;
    }
    v[i] = n_owned + lo;
  }
  tets[e] = t;
}

// One rank's share of a sharded system: partition, local numbering, halo and send lists on the host (build_fem_partition: one
// pass over the element list), then pattern / SELL-64 / contribution lists of the owned rows on the device -- the part that
// took 68-75 ms per rank at 1M tets on the host.
int build_shard_plan_on_device(fb_fem_s* h, int n_nodes, int n_tets, const int* tets, int n_fixed, const int* fixed, int n_ranks, int rank,
                               const int* splits, const double* xyz) {
  FemPlan& P = h->plan;
  static const bool timing = getenv("FEMBRAIN_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[fembrain] shard plan: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  // the partition (elements with an owned node, halo, send lists, local numbering) is computed on the device from the uploaded
  // list, be it the rank's own elements or the whole mesh; on the host only for more than 64 ranks or FEMBRAIN_PARTITION_DEVICE=0
  const bool on_device = n_ranks <= 64 && !(getenv("FEMBRAIN_PARTITION_DEVICE") && atoi(getenv("FEMBRAIN_PARTITION_DEVICE")) == 0);
  DevBuf<int> d_halo;
  if (on_device) {
    FB_TRY(begin_fem_partition(P, n_nodes, n_tets, n_ranks, rank, splits));
    FB_TRY(h->tets.upload((const int4*)tets, (size_t)n_tets, h->stream));
    if (renumber_mode(h) == FB_RENUMBER_ON) {
      // Opt-in on a sharded handle (include/fembrain_hip.h, "Node numbering"): every rank derives the SAME internal order from the whole
      // mesh, relabels its copy of the element list, and the ranks then own contiguous ranges of THAT order -- slabs of the body with
      // two neighbours each, whatever the caller's numbering.  Everything below works on internal ids; l2c maps back at the ABI.
      bool want = false;
      FB_TRY(renumber_decide(h->stream, FB_RENUMBER_ON, n_nodes, n_tets, h->tets.p, h->plan_ws, h->ren, &want));
      if (want) {
        FB_TRY(h->xyz_in.reserve((size_t)3 * n_nodes));
        FB_HIP(hipMemcpyAsync(h->xyz_in.p, xyz, sizeof(double) * 3 * (size_t)n_nodes, hipMemcpyHostToDevice, h->stream));
        FB_TRY(renumber_build(h->stream, FB_RENUMBER_ON, n_nodes, n_tets, h->tets.p, h->xyz_in.p, h->plan_ws, h->ren));
      }
      if (h->ren.active) {
        FB_TRY(relabel_tets(h->stream, n_tets, h->tets.p, n_nodes, h->ren.d_new_of_old.p));
        FB_TRY(h->ren.host_maps(h->stream));
        h->order_sum = 0;
        for (int l = 0; l < n_nodes; l++) h->order_sum += (unsigned long long)(l + 1) * (unsigned long long)(h->ren.old_of_new[l] + 7);
      }
      lap("node order");
    }
    DevicePartition dp;
    const int rc = device_partition(h->stream, n_tets, h->tets, n_nodes, n_ranks, rank, P.splits, dp, h->plan_ws);
    if (rc != FB_OK && dp.first_bad_tet >= 0) {  // say which node, as the host builder does
      for (int k = 0; k < 4; k++) {
        const int id = tets[4 * (size_t)dp.first_bad_tet + k];
        if (id < 0 || id >= n_nodes) return fail(FB_EINVAL, "tet %d references node %d outside [0,%d)", dp.first_bad_tet, id, n_nodes);
      }
    }
    FB_TRY(rc);
    P.n_tets = dp.n_kept;
    P.tet_global = dp.tet_global;
    P.n_owned_corners = dp.owned_corners;
    set_partition_halo(P, dp.halo);
    P.send_off = dp.send_off;
    P.send_local = dp.send_local;
  } else {
    FB_TRY(build_fem_partition(P, n_nodes, n_tets, tets, n_ranks, rank, splits, false));
  }
  lap(on_device ? "partition (device)" : "partition (host)");
  if (h->ren.active) {
    // caller ids of the local nodes, and the constrained DOFs in internal ids (ascending again)
    h->l2c.resize((size_t)P.n_local);
    for (int l = 0; l < P.n_local; l++) h->l2c[l] = h->ren.old_of_new[P.local2global[l]];
    for (int i = 0; i < n_fixed; i++) {
      if (fixed[i] < 0 || fixed[i] >= 3 * n_nodes) return fail(FB_EINVAL, "constrained DOF %d out of range [0,%d)", fixed[i], 3 * n_nodes);
      if (i && fixed[i] <= fixed[i - 1]) return fail(FB_EINVAL, "constrained DOFs must be strictly ascending (index %d)", i);
    }
    std::vector<int> mapped((size_t)n_fixed);
    for (int i = 0; i < n_fixed; i++) mapped[i] = 3 * h->ren.new_of_old[fixed[i] / 3] + fixed[i] % 3;
    std::sort(mapped.begin(), mapped.end());
    FB_TRY(plan_set_constraints(P, n_fixed, mapped.data()));
  } else {
    FB_TRY(plan_set_constraints(P, n_fixed, fixed));
  }
  lap("constraints");
  if (P.n_halo > 0) FB_TRY(d_halo.upload(P.local2global.data() + P.n_owned, (size_t)P.n_halo, h->stream));
  if (on_device) {
    // numbered locally already
  } else if (P.tets.empty()) {  // host partition of the caller's own list: local numbering on the device
    FB_TRY(h->tets.upload((const int4*)tets, (size_t)P.n_tets, h->stream));
    hipLaunchKernelGGL(k_tets_to_local, dim3(ceil_div(P.n_tets, kBlock)), dim3(kBlock), 0, h->stream, P.n_tets, h->tets.p, P.node_lo, P.n_owned, d_halo.p, P.n_halo);
    FB_HIP(hipGetLastError());
  } else {
    FB_TRY(h->tets.upload((const int4*)P.tets.data(), (size_t)P.n_tets, h->stream));
  }
  lap("element upload");
  PlanShard sh;
  sh.n_rows = P.n_owned; sh.node_lo = P.node_lo; sh.n_global = P.n_global; sh.d_halo = d_halo.p; sh.n_halo = P.n_halo;
  sh.n_pairs = 4 * P.n_owned_corners + P.n_owned;
  DevicePlan D;
  D.slice_off = &h->slice_off; D.colidx = &h->colidx; D.slot_coff = &h->slot_coff; D.slot_ccnt = &h->slot_ccnt; D.contrib = &h->contrib;
  D.bptr = &h->d_bptr; D.bcol = &h->d_bcol; D.blk_slot = &h->d_blk_slot; D.coldelta = &h->coldelta; D.halo_base = &h->halo_base;
  const int rc = build_plan_device(h->stream, P.n_local, P.n_tets, h->tets.p, D, h->plan_ws, &sh);
  if (h->plan_ws.bytes() > ((size_t)2 << 30)) h->plan_ws.release();
  FB_TRY(rc);
  P.n_blocks = D.n_blocks; P.n_slices = D.n_slices; P.n_slots = D.n_slots; P.n_crows = D.n_crows;
  h->c16 = (D.deltas_fit16 && !(getenv("FEMBRAIN_SPMV_C16") && atoi(getenv("FEMBRAIN_SPMV_C16")) == 0)) ? 2 : 0;  // the halo form
  P.slice_off = D.slice_off_host;
  return FB_OK;
}

// inspection entry points (pattern, block values, mass) index the CSR pattern on the host
int ensure_host_pattern(fb_fem_s* h) {
  if (h->host_pattern) return FB_OK;
  FemPlan& P = h->plan;
  P.bptr.resize((size_t)P.n_owned + 1);
  P.bcol.resize((size_t)P.n_blocks);
  P.blk_slot.resize((size_t)P.n_blocks);
  FB_TRY(h->d_bptr.download(P.bptr.data(), P.bptr.size(), h->stream));
  FB_TRY(h->d_bcol.download(P.bcol.data(), P.bcol.size(), h->stream));
  FB_TRY(h->d_blk_slot.download(P.blk_slot.data(), P.blk_slot.size(), h->stream));
  h->host_pattern = true;
  return FB_OK;
}

// rest state of a device-built plan; the kernel reports the first flat element (see build)
int rest_state_checked(fb_fem_s* h) {
  const int none = 0x7fffffff;
  FB_TRY(h->flat_flag.alloc(1));
  FB_HIP(hipMemcpyAsync(h->flat_flag.p, &none, sizeof(int), hipMemcpyHostToDevice, h->stream));
  FB_TRY(launch_rest(h, h->flat_flag.p));
  int first = none;
  FB_TRY(h->flat_flag.download(&first, 1, h->stream));
  if (first != none)
    return fail(FB_EINVAL, "element %d has zero (or non-finite) rest volume", h->plan.tet_global.empty() ? first : h->plan.tet_global[first]);
  return FB_OK;
}

int build(fb_fem_s* h, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed, const int* fixed, int n_ranks,
          int rank, const int* splits, const DeviceTetMesh* dm = nullptr) {
  drop_graph(h);  // the buffers it refers to are about to be replaced
  SlackScope slack(handle_slack(h, n_nodes, n_tets));
  if (h->prm.matrix_precision == FB_MATRIX_AUTO) h->f64 = auto_matrix_f64(h, n_nodes, n_ranks);
  h->last_resync_path = FB_RESYNC_FULL;
  h->csr_ready = false;
  h->span_stale = false;
  h->ren.clear();
  h->l2c.clear();
  h->order_sum = 0;
  h->x0_ready = false;
  h->masks_ready = false;
  h->caller_pattern = false;
  static const bool timing = getenv("FEMBRAIN_TIMING") != nullptr;  // development aid: where a (re)build spends its time
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[fembrain] build: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  const bool want_device = !(getenv("FEMBRAIN_PLAN_DEVICE") && atoi(getenv("FEMBRAIN_PLAN_DEVICE")) == 0);
  h->device_plan = (want_device || dm) && (n_ranks == 1 || !dm);
  h->host_pattern = !h->device_plan;
  if (h->device_plan) {
    const int rc = n_ranks > 1 ? build_shard_plan_on_device(h, n_nodes, n_tets, tets, n_fixed, fixed, n_ranks, rank, splits, xyz)
                               : build_plan_on_device(h, n_nodes, n_tets, tets, n_fixed, fixed, xyz, dm ? dm->tets : nullptr, dm ? dm->xyz : nullptr, dm ? dm->xyz64 : nullptr);
    if (dm && rc != FB_OK) return rc;
    if (rc == FB_ENOMEM) {  // no room for the sort's temporaries: the host builder needs none on the device
      (void)hipGetLastError();
      h->device_plan = false;
      h->host_pattern = true;
      h->ren.clear();  // (the host builder works in the caller's order)
      h->l2c.clear();
      h->order_sum = 0;
      h->x0_ready = false;
      h->masks_ready = false;
    } else if (rc != FB_OK) {
      return rc;
    }
    lap("device plan");
  }
  if (!h->device_plan) {
    h->c16 = 0;
    FB_TRY(build_fem_plan(h->plan, n_nodes, n_tets, tets, n_fixed, fixed, n_ranks, rank, splits));
    lap("host plan");
  }
  // A flat element makes inverse4x4 (corotationalLinearFEM.cpp:529-572) divide by zero; the reference then carries
  // inf/NaN into the step silently.  Refuse it here instead (checked on this rank's elements, rest geometry; a handle
  // whose plan was built on the device lets the rest-state kernel look, below).
  for (int e = 0; e < (h->device_plan ? 0 : h->plan.n_tets); e++) {
    const double* p[4];
    for (int k = 0; k < 4; k++)
      p[k] = xyz + 3 * (size_t)(h->device_plan ? tets[4 * (size_t)e + k] : h->plan.local2global[h->plan.tets[4 * (size_t)e + k]]);
    double a[3], b[3], c[3];
    for (int k = 0; k < 3; k++) { a[k] = p[1][k] - p[0][k]; b[k] = p[2][k] - p[0][k]; c[k] = p[3][k] - p[0][k]; }
    const double det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    if (!(det != 0.0) || !std::isfinite(det))
      return fail(FB_EINVAL, "element %d has zero (or non-finite) rest volume", h->plan.tet_global.empty() ? e : h->plan.tet_global[e]);
  }
  lap("volume check");
  FB_TRY(upload_plan(h, xyz, dm ? dm->xyz : nullptr, dm ? dm->xyz64 : nullptr));
  lap("upload");
  if (h->device_plan) {
    FB_TRY(rest_state_checked(h));
  } else {
    FB_TRY(launch_rest(h));
    FB_HIP(hipStreamSynchronize(h->stream));
  }
  lap("rest state");
  return FB_OK;
}

// The collective half: every rank allocates its box, the ranks exchange the IPC handles, the halo sizes and how many of their
// workgroups send to whom, map each other's boxes, and agree that all of it worked -- otherwise every rank alike runs the
// two-launch iteration.
int attach_pipe_shard(fb_fem_s* h) {
  const FemPlan& P = h->plan;
  const int R = P.n_ranks, me = P.rank;
  if (R < 2 || !h->comm) return FB_OK;
  if (!(getenv("FEMBRAIN_SHARDED_PERSIST") && atoi(getenv("FEMBRAIN_SHARDED_PERSIST")) != 0)) return FB_OK;  // (the same on every rank)
  {  // a re-sync: the previous box and mappings go first
    const bool keep = h->shard_persist;
    release_shard_persist(h);
    h->shard_persist = keep;
  }
  struct Meta { hipIpcMemHandle_t handle; long long n_halo; int halo_off[kP2PMaxRanks + 1]; int senders_to[kP2PMaxRanks]; int ok; };
  Meta mine;
  memset(&mine, 0, sizeof mine);
  bool ok = h->shard_persist && R <= kP2PMaxRanks;
  // the box is sized by the largest halo of all ranks, so a first round agrees on that
  long long nh = ok ? P.n_local - P.n_owned : -1;
  std::vector<long long> nhs((size_t)R);
  FB_TRY(comm_allgather_bytes(h->comm, &nh, nhs.data(), sizeof nh, h->stream));
  long long cap = 1;
  for (int q = 0; q < R; q++) { if (nhs[q] < 0) ok = false; cap = std::max(cap, nhs[q]); }
  const ShardBoxLayout BL = shard_box_layout(cap);
  if (ok) {
    ok = hipExtMallocWithFlags((void**)&h->sbox, BL.bytes, hipDeviceMallocFinegrained) == hipSuccess;
    ok = ok && hipMemsetAsync(h->sbox, 0, BL.bytes, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
    ok = ok && hipIpcGetMemHandle(&mine.handle, h->sbox) == hipSuccess;
    (void)hipGetLastError();
    mine.n_halo = P.n_local - P.n_owned;
    for (int q = 0; q <= R; q++) mine.halo_off[q] = P.halo_off[q];
    std::vector<unsigned int> wg_mask((size_t)h->persist_blocks);
    ok = ok && h->sh_wg_send_mask.download(wg_mask.data(), wg_mask.size(), h->stream) == FB_OK;
    for (unsigned int m : wg_mask)
      for (int q = 0; q < R; q++) mine.senders_to[q] += (m >> q) & 1u;
  }
  mine.ok = ok ? 1 : 0;
  std::vector<Meta> all((size_t)R);
  FB_TRY(comm_allgather_bytes(h->comm, &mine, all.data(), sizeof(Meta), h->stream));
  bool all_ok = true;
  for (int q = 0; q < R; q++) all_ok = all_ok && all[q].ok;
  int opened_ok = 1;
  std::vector<char*> peers((size_t)R, nullptr);
  std::vector<int> seg((size_t)R, 0), senders((size_t)R, 0);
  if (all_ok) {
    for (int q = 0; q < R; q++) {
      seg[q] = all[q].halo_off[me];
      senders[q] = all[q].senders_to[me];
      if (q == me) { peers[q] = h->sbox; continue; }
      void* ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, all[q].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened_ok = 0; break; }
      h->sbox_opened[q] = ptr;
      peers[q] = (char*)ptr;
    }
  }
  std::vector<int> oks((size_t)R);
  FB_TRY(comm_allgather_bytes(h->comm, &opened_ok, oks.data(), sizeof(int), h->stream));
  for (int q = 0; q < R; q++) all_ok = all_ok && oks[q];
  if (!all_ok) { release_shard_persist(h); h->persist = false; return FB_OK; }
  h->sbox_halo_cap = cap;
  FB_TRY(h->sbox_peers.upload(peers.data(), peers.size(), h->stream));
  FB_TRY(h->sh_peer_seg.upload(seg, h->stream));
  FB_TRY(h->sh_n_senders.upload(senders, h->stream));
  h->persist = true;
  if (getenv("FEMBRAIN_TIMING")) fprintf(stderr, "[fembrain] rank %d: sharded persistent solver attached (%d workgroups, up to %d slices per CU, %d fewer where halo rows are gathered, %d proxies per source rank, halo %lld of cap %lld)\n", me, h->persist_blocks, h->persist_waves, h->sh_relief, h->sh_n_proxy, (long long)(P.n_local - P.n_owned), cap);
  return FB_OK;
}

// Collective, BEFORE the rank-local build: FB_RENUMBER_AUTO on a sharded handle ("the same contiguous-range rule after an
// RCM / space-filling-curve renumbering").  Equal ranges of the caller's ids are slabs of the body only while the caller numbers plane
// by plane; after cuts (nodes appended at the end: VolMesh.cpp:1086-1091) or on a TetGen mesh every rank is a neighbour of every other
// and half the mesh is halo (profiles/r04_numbering_probe_after.json: 7 neighbours and 128k halo nodes per rank at 8 ranks, against 2 and
// 6k).  Every rank counts the ranks its own elements couple it to under the caller's numbering and sums a checksum of the element list
// it was handed; the ranks all-gather (neighbours, sizes, checksum, mode) and switch the internal order on -- all of them or none --
// when some rank would have more than two neighbours (or, from 8,192 nodes, most of its elements reaching into another rank) AND every
// rank holds the same whole mesh (the internal order is derived from all
// nodes and all elements; a rank that was handed its own elements only keeps the caller's numbering, as before).  One pass over the
// element list on the host (2 ms per million tets) and one all-gather of 40 bytes.
struct ShardOrderVote { int neighbours, n_nodes, n_tets, mode; unsigned long long sum; int splits_sum, own_elements, boundary_elements, pad; };
int vote_shard_order(fb_fem_s* h, int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* splits) {
  h->shard_auto_on = false;
  h->shard_vote_neighbours = 0;
  if (!h->comm || n_ranks < 2) return FB_OK;
  ShardOrderVote mine;
  memset(&mine, 0, sizeof mine);
  int mode = h->prm.renumber > 0 ? FB_RENUMBER_ON : (h->prm.renumber < 0 ? FB_RENUMBER_OFF : FB_RENUMBER_AUTO);
  if (const char* e = getenv("FEMBRAIN_RENUMBER")) mode = atoi(e) != 0 ? FB_RENUMBER_ON : FB_RENUMBER_OFF;
  mine.mode = mode; mine.n_nodes = n_nodes; mine.n_tets = n_tets; mine.neighbours = -1;
  if (mode == FB_RENUMBER_AUTO && tets && n_tets > 0 && n_nodes > 0 && n_ranks <= 64) {
    // (bad ranges, bad ids: the build refuses them with its own message; this rank then votes "unknown")
    shard_neighbour_count(n_nodes, n_tets, tets, n_ranks, rank, splits, &mine.neighbours, &mine.sum, &mine.splits_sum, &mine.own_elements, &mine.boundary_elements);
  }
  std::vector<ShardOrderVote> all((size_t)n_ranks);
  FB_TRY(comm_allgather_bytes(h->comm, &mine, all.data(), sizeof mine, h->stream));
  bool same = true, any_wide = false;
  int most = 0;
  for (int q = 0; q < n_ranks; q++) {
    same = same && all[q].mode == FB_RENUMBER_AUTO && all[q].neighbours >= 0 && all[q].n_nodes == all[0].n_nodes && all[q].n_tets == all[0].n_tets &&
           all[q].sum == all[0].sum && all[q].splits_sum == all[0].splits_sum;
    // more than two neighbour ranks; or (meshes of a size AUTO renumbers at all) most of a rank's elements reach into another rank -- the
    // two-rank form of the same disorder, where "every rank a neighbour" still means one
    any_wide = any_wide || all[q].neighbours > 2 || (n_nodes >= kRenumberMinNodes && 2LL * all[q].boundary_elements > all[q].own_elements);
    most = std::max(most, all[q].neighbours);
  }
  h->shard_vote_neighbours = most;
  h->shard_auto_on = same && any_wide;
  if (getenv("FEMBRAIN_TIMING"))
    fprintf(stderr, "[fembrain] rank %d: node-order vote: %d neighbour ranks here, %d at most, same mesh on every rank: %s -> %s\n", rank, mine.neighbours, most, same ? "yes" : "no",
            h->shard_auto_on ? "internal slab order" : "caller's numbering");
  return FB_OK;
}

// collective: a renumbered sharded handle works only if every rank derived the same internal order (from the same whole mesh)
int agree_on_node_order(fb_fem_s* h) {
  if (!h->comm || h->comm->n_ranks < 2) return FB_OK;
  const int R = h->comm->n_ranks;
  std::vector<unsigned long long> sums((size_t)R);
  FB_TRY(comm_allgather_bytes(h->comm, &h->order_sum, sums.data(), sizeof(unsigned long long), h->stream));
  for (int q = 0; q < R; q++)
    if (sums[q] != sums[0])
      return fail(FB_EINVAL, "the ranks derived different node orders (FB_RENUMBER_ON on a sharded handle needs the WHOLE mesh -- all nodes, all elements -- on every rank, and the same setting)");
  return FB_OK;
}

// collective: every rank of the communicator creates its handle at the same point of its program
int attach_p2p(fb_fem_s* h) {
  const FemPlan& P = h->plan;
  FB_TRY(p2p_attach(h->comm, P.n_local - P.n_owned, P.halo_off.data(), h->stream, &h->p2p));
  if (!h->p2p) return FB_OK;
  h->xch_mode = FB_XCH_P2P_FUSED;
  if (const char* e = getenv("FEMBRAIN_XCH_MODE")) {  // 2 / 3 / 4: how much of the exchange rides inside the PCG kernels
    const int m = atoi(e);
    if (m >= FB_XCH_COLLECTIVE && m <= FB_XCH_P2P_FUSED) h->xch_mode = m;
  }
  FB_TRY(h->send_off_dev.upload(P.send_off, h->stream));
  FB_TRY(h->halo_off_dev.upload(P.halo_off, h->stream));
  // slices with a halo column are done after the halo wait
  if (h->device_plan) {
    FB_TRY(h->slice_halo.alloc((size_t)std::max(1, P.n_slices)));
    hipLaunchKernelGGL(k_slice_halo, dim3(ceil_div(std::max(1, P.n_slices), kWavesPerBlock)), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->slice_off.p,
                       h->colidx.p, h->slice_halo.p);
    FB_HIP(hipGetLastError());
  } else {
    std::vector<unsigned char> sh((size_t)std::max(1, P.n_slices), 0);
    for (int sl = 0; sl < P.n_slices; sl++)
      for (size_t k = (size_t)P.slice_off[sl] * kSliceRows; k < (size_t)P.slice_off[sl + 1] * kSliceRows && !sh[sl]; k++) sh[sl] = P.colidx[k] >= P.n_owned;
    FB_TRY(h->slice_halo.upload(sh, h->stream));
  }
  if (P.send_local.empty()) FB_TRY(h->send_local.alloc(1));
  return FB_OK;
}

int prewarm_delta(fb_fem_s* h, int n_fixed, const int* fixed);

int create_common(fb_fem_t* out, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed, const int* fixed,
                  const fb_fem_params* params, int n_ranks, int rank, const int* splits, fb_comm_t comm, const DeviceTetMesh* dm = nullptr) {
  if (!out || (!dm && (!xyz || !tets)) || !params) return fail(FB_EINVAL, "null argument");
  if (dm && dm->device != params->device) return fail(FB_EINVAL, "the polygonizer lives on device %d, the FEM handle is asked for device %d", dm->device, params->device);
  if (n_fixed < 0 || (n_fixed > 0 && !fixed)) return fail(FB_EINVAL, "bad constrained DOF list");
  if (params->integrator != FB_INTEGRATOR_VOLUME_CONSERVING && params->integrator != FB_INTEGRATOR_NEWMARK) return fail(FB_EINVAL, "unknown integrator %d", params->integrator);
  if (!(params->timestep > 0) || !(params->E > 0) || !(params->rho > 0) || !(params->nu > -1.0 && params->nu < 0.5))
    return fail(FB_EINVAL, "bad material / timestep parameters");
  if (params->pcg_variant != FB_PCG_MERGED && params->pcg_variant != FB_PCG_REFERENCE && params->pcg_variant != FB_PCG_PERSISTENT &&
      params->pcg_variant != FB_PCG_BLOCK_JACOBI)
    return fail(FB_EINVAL, "unknown pcg_variant %d", params->pcg_variant);
  if (n_ranks > 1 && !comm) return fail(FB_EINVAL, "sharded handle needs a communicator");
  if (comm && (comm->n_ranks != n_ranks || comm->rank != rank)) return fail(FB_EINVAL, "communicator rank/size does not match the handle");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FB_EDEVICE, "no HIP device visible");
  if (params->device < 0 || params->device >= ndev) return fail(FB_EINVAL, "device %d out of range (%d visible)", params->device, ndev);
  FB_HIP(hipSetDevice(params->device));
  fb_fem_s* h = new fb_fem_s;
  h->prm = *params;
  h->comm = comm;  // a one-rank communicator with a live RCCL handle still runs the collectives (plumbing test)
  if (params->matrix_precision != FB_MATRIX_F32 && params->matrix_precision != FB_MATRIX_F64 && params->matrix_precision != FB_MATRIX_AUTO) {
    delete h;
    return fail(FB_EINVAL, "unknown matrix_precision %d", params->matrix_precision);
  }
  h->f64 = params->matrix_precision == FB_MATRIX_F64;  // (FB_MATRIX_AUTO: build() decides by size, at every re-sync again)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, params->device) == hipSuccess) h->n_cu_device = prop.multiProcessorCount;
  }
  h->lambda = (params->nu * params->E) / ((1 + params->nu) * (1 - 2 * params->nu));
  h->mu = params->E / (2 * (1 + params->nu));
  int rc = FB_OK;
  do {
   // collective, before anything rank-local can fail: the node order of a sharded handle under FB_RENUMBER_AUTO
   if (comm && comm->n_ranks > 1 && (rc = vote_shard_order(h, n_nodes, n_tets, tets, n_ranks, rank, splits)) != FB_OK) break;
   // rank-local set-up; a failure here (stream, events, pinned memory, the build) must still reach the agreement below -- the other ranks
   // are on their way into that all-gather
   do {
    // FEMBRAIN_CU_MASK=first:count -- the handle's stream runs on `count` CUs from bit `first` of the CU mask only (development and test
    // aid: two processes on one GPU, each with a persistent kernel on its own half of the CUs); the persistent grid follows
    if (const char* cm = getenv("FEMBRAIN_CU_MASK")) {
      int first = 0, count = 0;
      if (sscanf(cm, "%d:%d", &first, &count) != 2 || first < 0 || count < 8 || first + count > 512) { rc = fail(FB_EINVAL, "FEMBRAIN_CU_MASK=first:count"); break; }
      uint32_t mask[16];
      memset(mask, 0, sizeof mask);
      for (int b = first; b < first + count; b++) mask[b >> 5] |= 1u << (b & 31);
      if (hipExtStreamCreateWithCUMask(&h->stream, 16, mask) != hipSuccess) { rc = fail(FB_EDEVICE, "hipExtStreamCreateWithCUMask failed"); break; }
      h->cu_limit = count;
    } else
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(FB_EDEVICE, "hipStreamCreate failed"); break; }
    for (auto& e : h->ev) if (hipEventCreate(&e) != hipSuccess) rc = fail(FB_EDEVICE, "hipEventCreate failed");
    for (auto& e : h->ev_batch) if (hipEventCreate(&e) != hipSuccess) rc = fail(FB_EDEVICE, "hipEventCreate failed");
    for (auto& e : h->ev_p) if (hipEventCreate(&e) != hipSuccess) rc = fail(FB_EDEVICE, "hipEventCreate failed");
    if (rc != FB_OK) break;
    if (hipHostMalloc((void**)&h->st_host, 2 * sizeof(CGState), hipHostMallocDefault) != hipSuccess) { rc = fail(FB_ENOMEM, "hipHostMalloc failed"); break; }
    if (const char* e = getenv("FEMBRAIN_GRAPH")) h->use_graph = atoi(e) != 0;
    rc = build(h, n_nodes, xyz, n_tets, tets, n_fixed, fixed, n_ranks, rank, splits, dm);
    // a mesh that will be cut: the workspace of fb_fem_resync_delta now, not inside the first cut
    if (rc == FB_OK && h->prm.expect_cuts && n_ranks == 1 && h->device_plan) rc = prewarm_delta(h, n_fixed, fixed);
   } while (0);
    if (comm && comm->n_ranks > 1) {  // creation is collective: the ranks agree on the outcome so far before the collective attach
      const std::string why = rc == FB_OK ? std::string() : last_error();
      std::vector<int> rcs((size_t)n_ranks, 0);
      const int mine = rc;
      const int rc_x = comm_allgather_bytes(comm, &mine, rcs.data(), sizeof(int), h->stream);  // (host-staged or on the null stream if ours could not be made)
      if (rc_x != FB_OK) { rc = rc_x; break; }
      for (int q = 0; q < n_ranks && rc == FB_OK; q++)
        if (rcs[q] != FB_OK) rc = fail(rcs[q], "handle creation failed on rank %d (code %d)", q, rcs[q]);
      if (mine != FB_OK) { last_error() = why; rc = mine; }
      if (rc == FB_OK) rc = agree_on_node_order(h);
    }
    if (rc == FB_OK && comm && comm->n_ranks > 1) rc = attach_p2p(h);
    if (rc == FB_OK && comm && comm->n_ranks > 1) rc = attach_pipe_shard(h);
  } while (0);
  if (rc != FB_OK) {
    std::string keep = last_error();
    fb_fem_destroy(h);
    last_error() = keep;
    return rc;
  }
  *out = h;
  return FB_OK;
}

#define CHECK_HANDLE(h)                                                                                                        \
  if (!(h)) return fail(FB_EINVAL, "null FEM handle");                                                                        \
  if ((h)->poisoned) return fail(FB_EINVAL, "handle unusable after a failed fb_fem_resync: re-sync it with a valid mesh or destroy it"); \
  FB_HIP(hipSetDevice((h)->prm.device))

// global-length host vector -> local device vector (owned + halo)
int upload_global_vec(fb_fem_s* h, const double* g, DevBuf<double>& dst) {
  const FemPlan& P = h->plan;
  if (P.n_ranks == 1 && h->ren.active) {  // caller order -> internal order on the device
    FB_TRY(h->io.reserve((size_t)3 * P.n_local));
    FB_HIP(hipMemcpyAsync(h->io.p, g, sizeof(double) * 3 * (size_t)P.n_local, hipMemcpyHostToDevice, h->stream));
    FB_TRY(gather_nodes(h->stream, P.n_local, 3, h->io.p, h->ren.d_old_of_new.p, dst.p));
    FB_HIP(hipStreamSynchronize(h->stream));
    return FB_OK;
  }
  if (P.n_ranks == 1) {
    FB_HIP(hipMemcpyAsync(dst.p, g, sizeof(double) * 3 * (size_t)P.n_local, hipMemcpyHostToDevice, h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
    return FB_OK;
  }
  std::vector<double> loc((size_t)3 * P.n_local);
  const int* ids = h->l2c.empty() ? P.local2global.data() : h->l2c.data();
  for (int l = 0; l < P.n_local; l++)
    for (int k = 0; k < 3; k++) loc[3 * (size_t)l + k] = g[3 * (size_t)ids[l] + k];
  FB_HIP(hipMemcpyAsync(dst.p, loc.data(), sizeof(double) * loc.size(), hipMemcpyHostToDevice, h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  return FB_OK;
}

// owned part of a device vector -> its range of a global-length host vector
int download_owned(fb_fem_s* h, const DevBuf<double>& src, double* g) {
  const FemPlan& P = h->plan;
  if (P.n_ranks == 1 && h->ren.active) {  // internal order -> caller order on the device, one contiguous copy out
    FB_TRY(h->io.reserve((size_t)3 * P.n_local));
    FB_TRY(gather_nodes(h->stream, P.n_local, 3, src.p, h->ren.d_new_of_old.p, h->io.p));
    return h->io.download(g, (size_t)3 * P.n_local, h->stream);
  }
  if (!h->l2c.empty()) {  // a renumbered shard: its owned nodes lie anywhere in the caller's order
    std::vector<double> loc((size_t)3 * P.n_owned);
    FB_TRY(src.download(loc.data(), loc.size(), h->stream));
    for (int l = 0; l < P.n_owned; l++)
      for (int k = 0; k < 3; k++) g[3 * (size_t)h->l2c[l] + k] = loc[3 * (size_t)l + k];
    return FB_OK;
  }
  return src.download(g + 3 * (size_t)P.node_lo, (size_t)3 * P.n_owned, h->stream);
}

// host copies of the node maps of a renumbered handle (inspection entry points only)
int ensure_host_order(fb_fem_s* h) {
  if (!h->ren.active) return FB_OK;
  FB_TRY(h->ren.host_maps(h->stream));
  if (h->plan.n_ranks == 1) h->plan.local2global = h->ren.old_of_new;  // (a shard keeps its local -> internal map there, and l2c beside it)
  return FB_OK;
}

// The pattern in the caller's numbering (renumbered handles): row g of the caller = internal row new_of_old[g], its columns mapped
// back and sorted ascending -- fb_fem_pattern's contract -- with the internal block each of them is.
int ensure_caller_pattern(fb_fem_s* h) {
  FB_TRY(ensure_host_pattern(h));
  if (h->caller_pattern) return FB_OK;
  FB_TRY(ensure_host_order(h));
  const FemPlan& P = h->plan;
  h->c_bptr.assign((size_t)P.n_owned + 1, 0);
  h->c_bcol.resize((size_t)P.n_blocks);
  h->c_src.resize((size_t)P.n_blocks);
  std::vector<std::pair<int, int>> row;
  int at = 0;
  const bool shard = !h->l2c.empty();  // (a renumbered shard: rows in the order of fb_fem_owned_nodes, columns in the caller's ids)
  for (int g = 0; g < P.n_owned; g++) {
    const int a = shard ? g : h->ren.new_of_old[g];
    row.clear();
    for (int p = P.bptr[a]; p < P.bptr[a + 1]; p++) row.emplace_back(shard ? h->l2c[P.bcol[p]] : P.local2global[P.bcol[p]], p);
    std::sort(row.begin(), row.end());
    for (const auto& e : row) { h->c_bcol[at] = e.first; h->c_src[at] = e.second; at++; }
    h->c_bptr[(size_t)g + 1] = at;
  }
  h->caller_pattern = true;
  return FB_OK;
}


// SELL device values -> 9 doubles per block in CSR (fb_fem_pattern) order (diagonal blocks: hi + lo)
int download_blocks(fb_fem_s* h, double* out) {
  FB_TRY(ensure_host_pattern(h));
  const FemPlan& P = h->plan;
  const size_t n = (size_t)P.n_slots * 9 * 64, nl = (size_t)P.n_slices * 9 * 64;
  std::vector<double> host(n), lo(nl);
  if (h->f64) {
    FB_HIP(hipMemcpyAsync(host.data(), h->vals.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FB_HIP(hipMemcpyAsync(lo.data(), h->dlo.p, nl * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
  } else {
    std::vector<float> hf(n), lf(nl);
    FB_HIP(hipMemcpyAsync(hf.data(), h->vals.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    FB_HIP(hipMemcpyAsync(lf.data(), h->dlo.p, nl * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < n; i++) host[i] = hf[i];
    for (size_t i = 0; i < nl; i++) lo[i] = lf[i];
  }
  const bool mapped = h->ren.active;
  std::vector<double> internal;
  if (mapped) {
    FB_TRY(ensure_caller_pattern(h));
    internal.resize((size_t)9 * P.n_blocks);
  }
  double* dst = mapped ? internal.data() : out;
  for (int a = 0; a < P.n_owned; a++)
    for (int p = P.bptr[a]; p < P.bptr[a + 1]; p++)
      for (int v = 0; v < 9; v++) {
        double x = host[((size_t)P.blk_slot[p] * 9 + v) * 64 + (a & 63)];
        if (P.bcol[p] == a) x += lo[((size_t)(a >> 6) * 9 + v) * 64 + (a & 63)];
        dst[9 * (size_t)p + v] = x;
      }
  if (mapped)
    for (int pc = 0; pc < P.n_blocks; pc++) memcpy(out + 9 * (size_t)pc, internal.data() + 9 * (size_t)h->c_src[pc], 9 * sizeof(double));
  return FB_OK;
}

}  // namespace

extern "C" {

const char* fb_last_error(void) { return last_error().c_str(); }

int fb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fb_device_info(int dev, char* name, int name_len, char* arch, int arch_len, int* n_cu) {
  hipDeviceProp_t p;
  FB_HIP(hipGetDeviceProperties(&p, dev));
  if (name && name_len > 0) snprintf(name, name_len, "%s", p.name);
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
  if (n_cu) *n_cu = p.multiProcessorCount;
  return FB_OK;
}

int fb_host_register(void* p, unsigned long long bytes) {
  if (!p || bytes == 0) return fail(FB_EINVAL, "null argument");
  const hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(FB_EDEVICE, "hipHostRegister(%llu bytes): %s", bytes, hipGetErrorString(e)); }
  return FB_OK;
}

int fb_host_unregister(void* p) {
  if (!p) return FB_OK;
  const hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(FB_EDEVICE, "hipHostUnregister: %s", hipGetErrorString(e)); }
  return FB_OK;
}

void fb_fem_default_params(fb_fem_params* p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->E = 1e7; p->nu = 0.46; p->rho = 1000.0;
  p->timestep = 0.0333; p->damping_mass = 0.0; p->damping_stiffness = 0.01;
  p->cg_eps = 1e-6; p->cg_max_iter = 10000;
  p->matrix_precision = FB_MATRIX_AUTO; p->device = 0;
}

int fb_fem_create(fb_fem_t* out, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed_dofs,
                  const int* fixed_dofs, const fb_fem_params* params) {
  return create_common(out, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, params, 1, 0, nullptr, nullptr);
}

int fb_fem_create_from_poly(fb_fem_t* out, fb_poly_t poly, int n_fixed_dofs, const int* fixed_dofs, const fb_fem_params* params) {
  DeviceTetMesh dm;
  FB_TRY(poly_device_tetmesh(poly, &dm));
  if (dm.n_tets < 1) return fail(FB_EINVAL, "the polygonizer holds no tets");
  return create_common(out, dm.n_vertices, nullptr, dm.n_tets, nullptr, n_fixed_dofs, fixed_dofs, params, 1, 0, nullptr, nullptr, &dm);
}

int fb_fem_create_sharded(fb_fem_t* out, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed_dofs,
                          const int* fixed_dofs, const fb_fem_params* params, int n_ranks, int rank, const int* node_splits,
                          fb_comm_t comm) {
  return create_common(out, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, params, n_ranks, rank, node_splits, comm);
}

int fb_fem_destroy(fb_fem_t h) {
  if (!h) return FB_OK;
  (void)hipSetDevice(h->prm.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->ev_batch) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->ev_p) if (e) (void)hipEventDestroy(e);
  if (h->st_host) (void)hipHostFree(h->st_host);
  if (h->p2p) p2p_detach(h->p2p);
  release_shard_persist(h);
  drop_graph(h);
  DevBuf<double>* vecs[] = {&h->q, &h->qvel, &h->fext, &h->fint, &h->rhs, &h->x, &h->r, &h->d, &h->Ad, &h->invdiag, &h->tmp};
  for (auto* v : vecs) v->release();
  hipStream_t s = h->stream;
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  for (auto& e : h->ev_side) if (e) (void)hipEventDestroy(e);
  delete h;  // frees the remaining device buffers
  if (s) (void)hipStreamDestroy(s);
  return FB_OK;
}

int fb_fem_transport(fb_fem_t h) {
  if (!h) return -1;
  if (!h->comm || h->comm->n_ranks == 1) return 0;
  return h->xch_mode;
}

int fb_fem_set_exchange_mode(fb_fem_t h, int mode) {
  CHECK_HANDLE(h);
  if (!h->comm || h->comm->n_ranks == 1) return fail(FB_EINVAL, "an unsharded handle has no exchange");
  if (mode < FB_XCH_COLLECTIVE || mode > FB_XCH_P2P_FUSED) return fail(FB_EINVAL, "exchange mode %d", mode);
  if (mode >= FB_XCH_P2P && !h->p2p) return fail(FB_ECOMM, "the peer-to-peer transport is not attached (mapping failed or FEMBRAIN_P2P=0)");
  if (mode == FB_XCH_COLLECTIVE && !h->comm->nccl && !h->comm->local) return fail(FB_ECOMM, "the communicator has no collective library");
  FB_HIP(hipStreamSynchronize(h->stream));
  h->xch_mode = mode;
  return FB_OK;
}

int fb_fem_sharded_persist(fb_fem_t h) {
  if (!h) return 0;
  return h->shard_persist && h->persist ? 1 : 0;
}

int fb_fem_set_sharded_persist(fb_fem_t h, int on) {
  CHECK_HANDLE(h);
  if (!h->shard_persist || !h->sbox) return fail(FB_EINVAL, "the sharded persistent solver is not attached to this handle (FEMBRAIN_SHARDED_PERSIST=1 at creation, <= %d slices per CU)", 2 * (kPipeMaxWaves - 1));
  if (on && h->persist_broken) return fail(FB_EDEVICE, "the sharded persistent solver of this handle timed out before; it stays with the two-launch iteration");
  FB_HIP(hipStreamSynchronize(h->stream));
  h->persist = on != 0;
  return FB_OK;
}

// collective re-sync of a sharded handle; node_splits NULL keeps the handle's ranges when the node count is unchanged, else equal ranges
static int resync_sharded(fb_fem_t h, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed_dofs, const int* fixed_dofs,
                          const int* node_splits) {
  // every rank re-syncs at the same point of its program; the peer-to-peer inboxes are sized by the halo and are attached again
  h->poisoned = true;
  h->system_valid = false;
  const int n_ranks = h->plan.n_ranks, rank = h->plan.rank;
  const std::vector<int> kept = h->plan.splits;
  if (!node_splits && n_nodes == h->plan.n_global) node_splits = kept.data();
  // build() is rank-local (a bad node id or a flat element after a cut, no memory ...); what follows is collective.  The ranks
  // agree on the outcome first: if any of them failed, none enters the collective attach -- all stay poisoned and return an
  // error, instead of the healthy ones waiting in an all-gather for a rank that has left.
  FB_TRY(vote_shard_order(h, n_nodes, n_tets, tets, n_ranks, rank, node_splits));  // (collective; before the rank-local build)
  const int rc_mine = build(h, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, n_ranks, rank, node_splits);
  const std::string why_mine = rc_mine == FB_OK ? std::string() : last_error();
  if (h->comm && h->comm->n_ranks > 1) {
    std::vector<int> rcs((size_t)n_ranks, 0);
    const int rc_x = comm_allgather_bytes(h->comm, &rc_mine, rcs.data(), sizeof(int), h->stream);
    if (rc_x != FB_OK) return rc_x;
    for (int q = 0; q < n_ranks; q++)
      if (rcs[q] != FB_OK && rc_mine == FB_OK)
        return fail(rcs[q], "re-sync failed on rank %d (code %d); this rank's handle is unusable until a re-sync succeeds on every rank", q, rcs[q]);
  }
  if (rc_mine != FB_OK) { last_error() = why_mine; return rc_mine; }
  FB_TRY(agree_on_node_order(h));
  const int mode = h->xch_mode;
  if (h->p2p) { p2p_detach(h->p2p); h->p2p = nullptr; }
  if (h->comm && h->comm->n_ranks > 1) FB_TRY(attach_p2p(h));
  if (h->comm && h->comm->n_ranks > 1) FB_TRY(attach_pipe_shard(h));
  if (h->p2p && mode >= FB_XCH_P2P) h->xch_mode = mode;   // the form chosen before the re-sync stays
  else if (mode == FB_XCH_COLLECTIVE) h->xch_mode = FB_XCH_COLLECTIVE;
  h->poisoned = false;
  return FB_OK;
}

int fb_fem_resync_sharded(fb_fem_t h, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed_dofs, const int* fixed_dofs,
                          const int* node_splits) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  FB_HIP(hipSetDevice(h->prm.device));
  if (!xyz || !tets) return fail(FB_EINVAL, "null mesh");
  FB_HIP(hipStreamSynchronize(h->stream));
  if (h->plan.n_ranks == 1) {
    if (node_splits && (node_splits[0] != 0 || node_splits[1] != n_nodes)) return fail(FB_EINVAL, "node splits must cover [0,%d)", n_nodes);
    return fb_fem_resync(h, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs);
  }
  return resync_sharded(h, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, node_splits);
}

int fb_fem_resync(fb_fem_t h, int n_nodes, const double* xyz, int n_tets, const int* tets, int n_fixed_dofs, const int* fixed_dofs) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  FB_HIP(hipSetDevice(h->prm.device));
  if (!xyz || !tets) return fail(FB_EINVAL, "null mesh");
  FB_HIP(hipStreamSynchronize(h->stream));
  if (h->plan.n_ranks > 1) return resync_sharded(h, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, nullptr);
  // build() replaces the plan and the buffers in place; if it fails half way (a node id out of range after a bad
  // subdivision, a flat element, no memory) the handle holds pieces of two meshes and must not step
  h->poisoned = true;
  h->system_valid = false;
  FB_TRY(build(h, n_nodes, xyz, n_tets, tets, n_fixed_dofs, fixed_dofs, 1, 0, nullptr));
  h->poisoned = false;
  return FB_OK;
}

// ---- fb_fem_resync_delta (delta.h) ----
namespace {
// The full builder from the device copy of the new mesh in the caller's numbering (no sorted list to update, or the new mesh is to
// get a new node order): what fb_fem_resync does, without the host hop.
int resync_delta_rebuild(fb_fem_s* h, int n_old, int n_new, int n_fixed, const int* fixed) {
  hipStream_t s = h->stream;
  MeshDelta& D = h->delta;
  const int nt_old = D.n_tets_old, nt_new = D.n_tets_new();
  // the old element list and the rest positions in the caller's numbering
  FB_TRY(h->tets_caller.alloc((size_t)nt_old));
  FB_HIP(hipMemcpyAsync(h->tets_caller.p, h->tets.p, sizeof(int4) * (size_t)nt_old, hipMemcpyDeviceToDevice, s));
  if (h->ren.active) FB_TRY(relabel_tets(s, nt_old, h->tets_caller.p, n_old, h->ren.d_old_of_new.p));
  FB_TRY(h->tets_next.alloc((size_t)nt_new));
  FB_TRY(delta_tets(s, D, h->tets_caller.p, nullptr, h->tets_next.p));
  FB_TRY(h->x0_next.alloc((size_t)3 * n_new));
  if (h->ren.active) FB_TRY(scatter_nodes(s, n_old, 3, h->x0.p, h->ren.d_old_of_new.p, h->x0_next.p));
  else FB_HIP(hipMemcpyAsync(h->x0_next.p, h->x0.p, sizeof(double) * 3 * (size_t)n_old, hipMemcpyDeviceToDevice, s));
  if (D.n_new_nodes) FB_HIP(hipMemcpyAsync(h->x0_next.p + 3 * (size_t)n_old, D.new_xyz.p, sizeof(double) * 3 * (size_t)D.n_new_nodes, hipMemcpyDeviceToDevice, s));
  DeviceTetMesh dm;
  dm.device = h->prm.device; dm.n_vertices = n_new; dm.n_tets = nt_new; dm.xyz = nullptr;
  dm.tets = reinterpret_cast<const uint4*>(h->tets_next.p); dm.xyz64 = h->x0_next.p;
  FB_TRY(build(h, n_new, nullptr, nt_new, nullptr, n_fixed, fixed, 1, 0, nullptr, &dm));
  h->last_resync_path = FB_RESYNC_DELTA_REBUILT;
  return FB_OK;
}

int resync_delta(fb_fem_s* h, int n_removed, const int* removed, int n_changed, const int* changed_ids, const int* changed_nodes, int n_added, const int* added,
                 int n_new_nodes, const double* new_xyz, int n_fixed, const int* fixed) {
  static const bool timing = getenv("FEMBRAIN_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[fembrain] delta re-sync: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  hipStream_t s = h->stream;
  MeshDelta& D = h->delta;
  PlanWorkspace& W = h->plan_ws;
  const int n_old = h->plan.n_global, nt_old = h->plan.n_tets, n_new = n_old + n_new_nodes;
  SlackScope slack(handle_slack(h, n_new, nt_old - n_removed + n_added));
  drop_graph(h);
  FB_TRY(delta_upload(s, nt_old, n_removed, removed, n_changed, changed_ids, changed_nodes, n_added, added, n_new_nodes, new_xyz, D, W));
  const int nt_new = D.n_tets_new();
  lap("change uploaded");
  const char* env = getenv("FEMBRAIN_RESYNC_DELTA");
  bool merge = h->csr_ready && !(env && !strcmp(env, "rebuild"));
  // A renumbered handle keeps the order it has while nodes are added -- new nodes are merged in, old ones stay where they are although
  // the cut has changed how many elements sit on them, so rows of unlike length come to share a slice and the matrix pads.  Measured at
  // 1.1M tets (tools/probe_resync_delta.py): 9 % more nodes merged in, 42.5 us per PCG iteration against 30.6 in a fresh order -- 26 ms per
  // step -- where the full builder from the device copy of the mesh costs 2.9 ms once.  So once kFreshOrderPercent more nodes have come
  // than the order was built for, the change gets a fresh order (FEMBRAIN_FRESH_ORDER_PERCENT overrides: the tests keep the merged
  // path busy with larger changes)
  if (merge && h->ren.active && (long long)n_new * 100 > (long long)h->ren_nodes_at_build * (100 + fresh_order_percent())) merge = false;
  // FB_MATRIX_AUTO is decided again at every re-sync: a mesh that has grown across the size rule gets its values in the other width,
  // which the full builder allocates
  if (merge && h->prm.matrix_precision == FB_MATRIX_AUTO && auto_matrix_f64(h, n_new, 1) != h->f64) merge = false;
  const int mode = renumber_mode(h);
  int span = -1;
  double mean = 0.0;
  if (merge && !h->ren.active && mode != FB_RENUMBER_OFF) {
    // A handle in the caller's order: would the full builder try a new node order for the new mesh (renumber_decide)?  Then it is to
    // decide.  The widest element of the new list: the kept elements are no wider than before, so the new ones settle it unless
    // the old widest element was removed -- measured over the whole new list, as the builder would.
    FB_TRY(h->tets_next.alloc((size_t)nt_new));
    FB_TRY(delta_tets(s, D, h->tets.p, nullptr, h->tets_next.p));
    FB_TRY(tet_span_device(s, nt_new, h->tets_next.p, n_new, nullptr, W, &span, &mean));
    if (mode == FB_RENUMBER_ON || (n_new >= kRenumberMinNodes && span > renumber_span_limit(n_new))) merge = false;
  }
  if (!merge) return resync_delta_rebuild(h, n_old, n_new, n_fixed, fixed);

  // ---- the node order ----
  if (h->ren.active) {
    if (n_new_nodes) {
      FB_TRY(delta_node_order(s, D, n_old, h->ren.geom, h->ren.d_keys.p, h->ren.d_old_of_new.p, h->map_next_a, h->map_next_b, W, h->ren.sigma ? h->ren.n_windows : 0,
                              h->ren.sigma ? h->ren.d_win_keys.p : nullptr));
      h->ren.d_old_of_new.swap(h->map_next_a);
      h->ren.d_new_of_old.swap(h->map_next_b);
      h->ren.d_keys.swap(D.node_keys);
      h->ren.n = n_new;
      h->ren.old_of_new.clear();
      h->ren.new_of_old.clear();
    }
    FB_TRY(delta_relabel_nodes(s, D, n_new, h->ren.d_new_of_old.p));
    FB_TRY(h->tets_next.alloc((size_t)nt_new));
    FB_TRY(delta_tets(s, D, h->tets.p, D.mapped ? D.imap.p : nullptr, h->tets_next.p));
    h->span_stale = true;  // (the widest element under the merged order is measured when fb_fem_renumbering asks: a pass and a wait saved here)
  } else {
    h->ren.clear();
    if (mode != FB_RENUMBER_OFF) { h->ren.span_before = h->ren.span_after = span; h->ren.mean_before = h->ren.mean_after = mean; }
    else {
      FB_TRY(h->tets_next.alloc((size_t)nt_new));
      FB_TRY(delta_tets(s, D, h->tets.p, nullptr, h->tets_next.p));
    }
  }
  h->tets.swap(h->tets_next);
  FB_TRY(h->x0_next.alloc((size_t)3 * n_new));
  FB_TRY(delta_positions(s, D, n_old, h->x0.p, h->x0_next.p));
  h->x0.swap(h->x0_next);
  h->l2c.clear();
  h->order_sum = 0;
  h->x0_ready = true;
  h->caller_pattern = false;
  h->last_resync_path = FB_RESYNC_DELTA_MERGED;
  lap("node order and elements");
  FB_TRY(device_constraint_masks(s, n_new, n_fixed, fixed, h->ren.active ? h->ren.d_new_of_old.p : nullptr, h->fixed_stage, h->dofmask, h->nodemask));
  h->masks_ready = true;
  // ---- the plan, from the plan (delta.hip) ----
  OldPlanArrays old_plan = {h->d_bptr.p, h->d_bcol.p, h->d_ucnt.p, h->slice_off.p, h->slot_coff.p, h->contrib.p, n_old, h->plan.n_blocks};
  FemPlan& P = h->plan;
  P = FemPlan();
  P.n_global = n_new; P.n_ranks = 1; P.rank = 0;
  P.splits = {0, n_new};
  P.node_lo = 0; P.node_hi = n_new;
  P.n_owned = P.n_local = n_new; P.n_halo = 0;
  P.local2global.resize(n_new);
  for (int l = 0; l < n_new; l++) P.local2global[l] = l;
  P.halo_off.assign(2, 0);
  P.send_off.assign(2, 0);
  P.n_tets = nt_new;
  P.n_fixed_owned = n_fixed;
  DevicePlan Dp;
  // (built next to the arrays it is built from; colidx, the list heights and the 16-bit column words are read by no stage of it)
  Dp.slice_off = &D.slice_off2; Dp.colidx = &h->colidx; Dp.slot_coff = &D.slot_coff2; Dp.slot_ccnt = &h->slot_ccnt; Dp.contrib = &D.contrib2;
  Dp.bptr = &D.bptr2; Dp.bcol = &D.bcol2; Dp.blk_slot = &D.blk_slot2; Dp.coldelta = &h->coldelta; Dp.ucnt_keep = &D.ucnt2;
  h->csr_ready = false;
  FB_TRY(delta_plan(s, D, old_plan, h->tets_next.p, h->tets.p, n_new, Dp, W));  // (tets_next: the old list, swapped out above)
  h->slice_off.swap(D.slice_off2); h->slot_coff.swap(D.slot_coff2); h->contrib.swap(D.contrib2);
  h->d_bptr.swap(D.bptr2); h->d_bcol.swap(D.bcol2); h->d_blk_slot.swap(D.blk_slot2); h->d_ucnt.swap(D.ucnt2);
  h->csr_ready = true;
  P.n_blocks = Dp.n_blocks; P.n_slices = Dp.n_slices; P.n_slots = Dp.n_slots; P.n_crows = Dp.n_crows;
  h->c16 = (Dp.deltas_fit16 && !(getenv("FEMBRAIN_SPMV_C16") && atoi(getenv("FEMBRAIN_SPMV_C16")) == 0)) ? 1 : 0;
  P.slice_off = Dp.slice_off_host;
  h->device_plan = true;
  h->host_pattern = false;
  lap("plan");
  FB_TRY(upload_plan(h, nullptr));
  lap("per-step arrays");
  FB_TRY(rest_state_checked(h));
  lap("rest state");
  return FB_OK;
}

// fb_fem_params.expect_cuts: everything a fb_fem_resync_delta allocates -- the second set of plan arrays it builds into, the element and
// node maps, the sort's temporaries for a change of a twentieth of the elements -- is allocated at creation by running the empty change
int prewarm_delta(fb_fem_s* h, int n_fixed, const int* fixed) {
  MeshDelta& D = h->delta;
  FB_TRY(delta_reserve(h->stream, D, h->plan_ws, (long long)16 * (h->plan.n_tets / 20 + 64)));
  if (h->ren.active) {
    // ... and the node order's second stage on this mesh, into a scratch order: a cut mesh needs it (renumber.h) where the uncut one did
    // not, and the first launch of its kernels in a process costs milliseconds (7 of the 9.7 ms of a first cut, tools/probe_resync_delta.py)
    Renumbering scratch;
    FB_TRY(renumber_build(h->stream, FB_RENUMBER_ON, h->plan.n_global, h->plan.n_tets, h->tets.p, h->x0.p, h->plan_ws, scratch, true));
  }
  FB_TRY(resync_delta(h, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, n_fixed, fixed));
  h->last_resync_path = FB_RESYNC_FULL;
  return FB_OK;
}

}  // namespace

int fb_fem_resync_delta(fb_fem_t h, int n_removed, const int* removed, int n_changed, const int* changed_ids, const int* changed_nodes, int n_added,
                        const int* added_tets, int n_new_nodes, const double* new_xyz, int n_fixed_dofs, const int* fixed_dofs) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  FB_HIP(hipSetDevice(h->prm.device));
  if (h->poisoned) return fail(FB_EINVAL, "the handle is unusable after a failed re-sync: a full fb_fem_resync first");
  if (h->plan.n_ranks > 1 || !h->device_plan) return fail(FB_EINVAL, "fb_fem_resync_delta is for unsharded handles whose plan was built on the device");
  if (n_removed < 0 || n_changed < 0 || n_added < 0 || n_new_nodes < 0 || n_fixed_dofs < 0) return fail(FB_EINVAL, "negative count");
  if ((n_removed && !removed) || (n_changed && (!changed_ids || !changed_nodes)) || (n_added && !added_tets) || (n_new_nodes && !new_xyz) || (n_fixed_dofs && !fixed_dofs))
    return fail(FB_EINVAL, "null array with a non-zero count");
  const int nt_old = h->plan.n_tets;
  const long long n_new = (long long)h->plan.n_global + n_new_nodes, nt_new = (long long)nt_old - n_removed + n_added;
  if (n_new >= (1LL << 31) - 2 || nt_new >= (1LL << 28)) return fail(FB_EINVAL, "mesh too large");
  if (nt_new < 1) return fail(FB_EINVAL, "the change leaves no element");
  for (int k = 0; k < n_removed; k++)
    if (removed[k] < 0 || removed[k] >= nt_old || (k && removed[k] <= removed[k - 1])) return fail(FB_EINVAL, "removed[%d] = %d: ids must ascend inside [0,%d)", k, removed[k], nt_old);
  for (int k = 0, r = 0; k < n_changed; k++) {
    if (changed_ids[k] < 0 || changed_ids[k] >= nt_old || (k && changed_ids[k] <= changed_ids[k - 1]))
      return fail(FB_EINVAL, "changed_ids[%d] = %d: ids must ascend inside [0,%d)", k, changed_ids[k], nt_old);
    while (r < n_removed && removed[r] < changed_ids[k]) r++;
    if (r < n_removed && removed[r] == changed_ids[k]) return fail(FB_EINVAL, "element %d is both removed and changed", changed_ids[k]);
  }
  for (long long k = 0; k < 4LL * n_changed; k++)
    if (changed_nodes[k] < 0 || changed_nodes[k] >= n_new) return fail(FB_EINVAL, "changed element %d references node %d outside [0,%lld)", changed_ids[k / 4], changed_nodes[k], n_new);
  for (long long k = 0; k < 4LL * n_added; k++)
    if (added_tets[k] < 0 || added_tets[k] >= n_new) return fail(FB_EINVAL, "added element %lld references node %d outside [0,%lld)", k / 4, added_tets[k], n_new);
  for (int k = 0; k < n_fixed_dofs; k++) {
    if (fixed_dofs[k] < 0 || fixed_dofs[k] >= 3 * n_new) return fail(FB_EINVAL, "constrained DOF %d out of range [0,%lld)", fixed_dofs[k], 3 * n_new);
    if (k && fixed_dofs[k] <= fixed_dofs[k - 1]) return fail(FB_EINVAL, "constrained DOFs must be strictly ascending (index %d)", k);
  }
  FB_HIP(hipStreamSynchronize(h->stream));
  h->poisoned = true;
  h->system_valid = false;
  FB_TRY(resync_delta(h, n_removed, removed, n_changed, changed_ids, changed_nodes, n_added, added_tets, n_new_nodes, new_xyz, n_fixed_dofs, fixed_dofs));
  h->poisoned = false;
  return FB_OK;
}

int fb_fem_resync_path(fb_fem_t h) { return h ? h->last_resync_path : FB_RESYNC_FULL; }

int fb_fem_rebuild_elements(fb_fem_t h) {
  CHECK_HANDLE(h);
  h->system_valid = false;
  return launch_rest(h);
}

int fb_fem_set_external_forces(fb_fem_t h, const double* f) {
  CHECK_HANDLE(h);
  if (!f) return fail(FB_EINVAL, "null force vector");
  return upload_global_vec(h, f, h->fext);
}

int fb_fem_add_external_forces(fb_fem_t h, const double* f) {
  CHECK_HANDLE(h);
  if (!f) return fail(FB_EINVAL, "null force vector");
  FB_TRY(upload_global_vec(h, f, h->tmp));
  const int n = 3 * h->plan.n_local;
  hipLaunchKernelGGL(k_axpy, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, h->stream, n, 1.0, h->tmp.p, h->fext.p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int fb_fem_set_external_forces_zero(fb_fem_t h) {
  CHECK_HANDLE(h);
  return h->fext.zero(h->stream);
}

int fb_fem_set_uniform_force(fb_fem_t h, int axis, double value) {
  CHECK_HANDLE(h);
  if (axis < 0 || axis > 2) return fail(FB_EINVAL, "axis %d", axis);
  const int n = h->plan.n_local;
  hipLaunchKernelGGL(k_fill_axis, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, h->stream, n, axis, value, h->fext.p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

namespace {
// ImplicitNewmarkSparse::DoTimestep (implicitNewmarkSparse.cpp:183-379), PCG solver
int newmark_step(fb_fem_s* h, fb_step_info* info) {
  hipStream_t s = h->stream;
  const int n = 3 * h->plan.n_owned;
  const double hh = h->prm.timestep, beta = h->nm_beta, gamma = h->nm_gamma;
  NewmarkAlphas al;
  al.a1 = 1.0 / (beta * hh * hh); al.a2 = 1.0 / (beta * hh); al.a3 = (1.0 - 2.0 * beta) / (2.0 * beta);
  al.a4 = gamma / (beta * hh); al.a5 = 1 - gamma / beta; al.a6 = (1.0 - gamma / (2.0 * beta)) * hh;
  FB_HIP(hipEventRecord(h->ev[0], s));
  const size_t bytes = sizeof(double) * (size_t)n;
  FB_HIP(hipMemcpyAsync(h->q1.p, h->q.p, bytes, hipMemcpyDeviceToDevice, s));
  FB_HIP(hipMemcpyAsync(h->qvel1.p, h->qvel.p, bytes, hipMemcpyDeviceToDevice, s));
  FB_HIP(hipMemcpyAsync(h->qacc1.p, h->qacc.p, bytes, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_newmark_update, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, s, n, (const double*)nullptr, h->dofmask.p, al, h->q1.p, h->qvel1.p, h->qacc1.p,
                     h->q.p, h->qvel.p, h->qacc.p);
  FB_HIP(hipGetLastError());
  int total = 0, newton = 0;
  double error0 = 0.0, asm_s = 0.0, solve_s = 0.0;
  CGState fin;
  memset(&fin, 0, sizeof fin);
  bool ok = true;
  do {
    FB_HIP(hipEventRecord(h->ev[0], s));
    FB_TRY(assemble_system(h));
    // Newton error test on the residual, summed over ALL DOFs as implicitNewmarkSparse.cpp:258-262 does before RemoveRows -- the reaction
    // forces at the clamped DOFs included: the assembly leaves the unmasked residual in res_all beside the right-hand side (whose clamped
    // rows are 0, identity rows here).  (Rounds 2-4 summed the right-hand side, i.e. the free DOFs only, and could stop an iteration early.)
    // FemBrain itself runs one Newton iteration (Deformable.cpp:205-214).
    if (h->nm_max_newton > 1) {
      hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(kBlock), 0, s, (size_t)n, h->res_all.p, h->scal.p + 4);
      FB_HIP(hipGetLastError());
      double err = 0.0;
      FB_HIP(hipMemcpyAsync(&err, h->scal.p + 4, sizeof err, hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
      if (h->comm && h->comm->n_ranks > 1) return fail(FB_EINVAL, "Newmark with more than one Newton iteration is not built for sharded handles");
      if (newton == 0) error0 = err;
      else if (err / error0 < h->nm_eps * h->nm_eps) break;
    }
    FB_HIP(hipEventRecord(h->ev[1], s));
    int iters = 0;
    h->pcg_warm = true;  // `buffer` is not cleared between solves (implicitNewmarkSparse.cpp:317-320)
    FB_TRY(pcg_solve(h, h->rhs.p, h->prm.cg_eps, h->prm.cg_max_iter, &iters, &fin));
    FB_HIP(hipEventRecord(h->ev[2], s));
    total += std::abs(iters);
    if (iters < 0) { ok = false; break; }
    hipLaunchKernelGGL(k_newmark_update, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, s, n, h->x.p, h->dofmask.p, al, h->q1.p, h->qvel1.p, h->qacc1.p, h->q.p,
                       h->qvel.p, h->qacc.p);
    FB_HIP(hipGetLastError());
    FB_HIP(hipStreamSynchronize(s));
    float ms_a = 0, ms_s = 0;
    FB_HIP(hipEventElapsedTime(&ms_a, h->ev[0], h->ev[1]));
    FB_HIP(hipEventElapsedTime(&ms_s, h->ev[1], h->ev[2]));
    asm_s += ms_a * 1e-3; solve_s += ms_s * 1e-3;
    newton++;
  } while (newton < h->nm_max_newton);
  FB_HIP(hipStreamSynchronize(s));
  h->system_valid = false;
  h->last_assembly_s = asm_s; h->last_solve_s = solve_s;
  if (info) {
    info->cg_iterations = total;
    info->converged = ok ? 1 : 0;
    info->assembly_seconds = asm_s;
    info->solve_seconds = solve_s;
    info->rho0 = fin.rho0;
    info->rho = fin.rho[fin.iter & 1];
    info->pcg_path = h->last_pcg_path;
    info->persist_fallbacks = h->persist_fallbacks;
    info->newton_iterations = newton;
  }
  if (!ok) return fail(FB_ESOLVER, "PCG sparse solver returned non-zero exit status %d", -total);
  return FB_OK;
}
}  // namespace

int fb_fem_set_newmark(fb_fem_t h, double beta, double gamma, int max_newton_iterations, double epsilon) {
  CHECK_HANDLE(h);
  if (!(beta > 0) || !(gamma > 0) || max_newton_iterations < 1 || !(epsilon >= 0)) return fail(FB_EINVAL, "bad Newmark parameters");
  if (h->prm.integrator != FB_INTEGRATOR_NEWMARK) return fail(FB_EINVAL, "the handle was not created with fb_fem_params.integrator = FB_INTEGRATOR_NEWMARK");
  if (max_newton_iterations > 1 && h->comm && h->comm->n_ranks > 1)
    return fail(FB_EINVAL, "Newmark with more than one Newton iteration is not built for sharded handles (the error quotient is a global sum)");
  h->nm_beta = beta; h->nm_gamma = gamma; h->nm_max_newton = max_newton_iterations; h->nm_eps = epsilon;
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_step(fb_fem_t h, fb_step_info* info) {
  CHECK_HANDLE(h);
  if (h->prm.integrator == FB_INTEGRATOR_NEWMARK) return newmark_step(h, info);
  hipStream_t s = h->stream;
  FB_HIP(hipEventRecord(h->ev[0], s));
  FB_TRY(assemble_system(h));
  FB_HIP(hipEventRecord(h->ev[1], s));
  int iters = 0;
  CGState fin;
  FB_TRY(pcg_solve(h, h->rhs.p, h->prm.cg_eps, h->prm.cg_max_iter, &iters, &fin));
  FB_HIP(hipEventRecord(h->ev[2], s));
  const bool ok = iters >= 0;
  if (ok) {
    const int n = 3 * h->plan.n_owned;
    hipLaunchKernelGGL(k_state_update, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, s, n, h->x.p, h->dofmask.p, h->prm.timestep, h->q.p,
                       h->qvel.p);
    FB_HIP(hipGetLastError());
  }
  FB_HIP(hipStreamSynchronize(s));
  float ms_a = 0, ms_s = 0;
  FB_HIP(hipEventElapsedTime(&ms_a, h->ev[0], h->ev[1]));
  FB_HIP(hipEventElapsedTime(&ms_s, h->ev[1], h->ev[2]));
  h->last_assembly_s = ms_a * 1e-3; h->last_solve_s = ms_s * 1e-3;
  if (info) {
    info->cg_iterations = std::abs(iters);
    info->converged = ok ? 1 : 0;
    info->assembly_seconds = h->last_assembly_s;
    info->solve_seconds = h->last_solve_s;
    info->rho0 = fin.rho0;
    info->rho = fin.rho[fin.iter & 1];
    info->pcg_path = h->last_pcg_path;
    info->persist_fallbacks = h->persist_fallbacks;
    info->newton_iterations = 1;
  }
  if (!ok) return fail(FB_ESOLVER, "PCG sparse solver returned non-zero exit status %d", iters);
  return FB_OK;
}

int fb_fem_get_state(fb_fem_t h, double* q, double* qvel, double* qaccel) {
  CHECK_HANDLE(h);
  if (q) FB_TRY(download_owned(h, h->q, q));
  if (qvel) FB_TRY(download_owned(h, h->qvel, qvel));
  if (qaccel && h->prm.integrator == FB_INTEGRATOR_NEWMARK) FB_TRY(download_owned(h, h->qacc, qaccel));
  else if (qaccel && !h->l2c.empty()) {
    for (int l = 0; l < h->plan.n_owned; l++) memset(qaccel + 3 * (size_t)h->l2c[l], 0, 3 * sizeof(double));
  } else if (qaccel && h->ren.active) memset(qaccel, 0, sizeof(double) * 3 * (size_t)h->plan.n_global);
  else if (qaccel) memset(qaccel + 3 * (size_t)h->plan.node_lo, 0, sizeof(double) * 3 * (size_t)h->plan.n_owned);  // forced 0, PS_VolumeConservingIntegrator.cpp:55
  return FB_OK;
}

int fb_fem_set_state(fb_fem_t h, const double* q, const double* qvel, const double* qaccel) {
  CHECK_HANDLE(h);
  if (!q) return fail(FB_EINVAL, "q must not be null (IntegratorBase::SetqState)");
  FB_TRY(upload_global_vec(h, q, h->q));
  if (qvel) FB_TRY(upload_global_vec(h, qvel, h->qvel));
  if (qaccel && h->prm.integrator == FB_INTEGRATOR_NEWMARK) FB_TRY(upload_global_vec(h, qaccel, h->qacc));
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_reset(fb_fem_t h) {
  CHECK_HANDLE(h);
  FB_TRY(h->q.zero(h->stream));
  FB_TRY(h->qvel.zero(h->stream));
  if (h->prm.integrator == FB_INTEGRATOR_NEWMARK) {
    FB_TRY(h->qacc.zero(h->stream));
    FB_TRY(h->x.zero(h->stream));  // the solver's start vector (IntegratorBase::ResetToRest clears its buffers)
  }
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_set_timestep(fb_fem_t h, double timestep) {
  CHECK_HANDLE(h);
  if (!(timestep > 0)) return fail(FB_EINVAL, "timestep must be positive");
  h->prm.timestep = timestep;
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_set_damping(fb_fem_t h, double damping_mass, double damping_stiffness) {
  CHECK_HANDLE(h);
  h->prm.damping_mass = damping_mass; h->prm.damping_stiffness = damping_stiffness;
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_set_internal_force_scaling(fb_fem_t h, double factor) {
  CHECK_HANDLE(h);
  if (!(factor > 0) || !std::isfinite(factor)) return fail(FB_EINVAL, "internal force scaling factor must be positive");
  // f_int and K are linear in Young's modulus, so scaling both (integratorBase.cpp:46, implicitNewmarkSparse.cpp:200-204;
  // PS_VolumeConservingIntegrator.cpp:84-90) is scaling the Lame parameters the element kernels use
  const double E = h->prm.E * factor, nu = h->prm.nu;
  h->lambda = (nu * E) / ((1 + nu) * (1 - 2 * nu));
  h->mu = E / (2 * (1 + nu));
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_set_cg(fb_fem_t h, double eps, int max_iter) {
  CHECK_HANDLE(h);
  if (!(eps > 0) || max_iter < 0) return fail(FB_EINVAL, "bad PCG parameters");
  h->prm.cg_eps = eps; h->prm.cg_max_iter = max_iter;
  return FB_OK;
}

int fb_fem_set_constrained_dofs(fb_fem_t h, int n_fixed_dofs, const int* fixed_dofs) {
  CHECK_HANDLE(h);
  if (n_fixed_dofs < 0 || (n_fixed_dofs > 0 && !fixed_dofs)) return fail(FB_EINVAL, "bad constrained DOF list");
  if (h->device_plan && h->plan.n_ranks == 1) {
    FB_TRY(device_constraint_masks(h->stream, h->plan.n_global, n_fixed_dofs, fixed_dofs, h->ren.active ? h->ren.d_new_of_old.p : nullptr, h->fixed_stage, h->dofmask, h->nodemask));
    h->plan.n_fixed_owned = n_fixed_dofs;
  } else if (!h->l2c.empty()) {  // a renumbered shard: the list in internal ids, ascending again
    const int r = 3 * h->plan.n_global;
    for (int i = 0; i < n_fixed_dofs; i++) {
      if (fixed_dofs[i] < 0 || fixed_dofs[i] >= r) return fail(FB_EINVAL, "constrained DOF %d out of range [0,%d)", fixed_dofs[i], r);
      if (i && fixed_dofs[i] <= fixed_dofs[i - 1]) return fail(FB_EINVAL, "constrained DOFs must be strictly ascending (index %d)", i);
    }
    std::vector<int> mapped((size_t)n_fixed_dofs);
    for (int i = 0; i < n_fixed_dofs; i++) mapped[i] = 3 * h->ren.new_of_old[fixed_dofs[i] / 3] + fixed_dofs[i] % 3;
    std::sort(mapped.begin(), mapped.end());
    FB_TRY(plan_set_constraints(h->plan, n_fixed_dofs, mapped.data()));
    FB_TRY(upload_masks(h));
  } else {
    FB_TRY(plan_set_constraints(h->plan, n_fixed_dofs, fixed_dofs));
    FB_TRY(upload_masks(h));
  }
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_floor_collision(fb_fem_t h, double floor_y, double restitution, int* n_collided) {
  CHECK_HANDLE(h);
  FB_TRY(h->counter.zero(h->stream));
  const int n = h->plan.n_owned;
  hipLaunchKernelGGL(k_floor, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, h->stream, n, h->x0.p, floor_y, restitution, h->q.p,
                     h->qvel.p, h->counter.p);
  FB_HIP(hipGetLastError());
  int c = 0;
  FB_TRY(h->counter.download(&c, 1, h->stream));
  if (n_collided) *n_collided = c;
  h->system_valid = false;
  return FB_OK;
}

int fb_fem_plan_on_device(fb_fem_t h) { return h && h->device_plan ? 1 : 0; }
int fb_fem_assembly_kernel(fb_fem_t h) { return h && h->asm_tets ? (h->asm_staged ? 2 : 1) : 0; }
int fb_fem_assembly_wide_slices(fb_fem_t h) { return h && h->asm_tets ? h->asm_wide : 0; }

long long fb_fem_device_plan_get(fb_fem_t h, const char* name, int* out, long long capacity) {
  CHECK_HANDLE(h);
  if (!name) return fail(FB_EINVAL, "null name");
  const FemPlan& P = h->plan;
  const std::string n(name);
  const void* src = nullptr;
  long long cnt = 0;
  if (n == "slice_off") { src = h->slice_off.p; cnt = P.n_slices + 1; }
  else if (n == "colidx") { src = h->colidx.p; cnt = (long long)P.n_slots * kSliceRows; }
  else if (n == "slot_coff") { src = h->slot_coff.p; cnt = P.n_slots; }
  else if (n == "slot_ccnt") { src = h->slot_ccnt.p; cnt = P.n_slots; }
  else if (n == "contrib") { src = h->contrib.p; cnt = (long long)P.n_crows * kSliceRows; }
  else if (h->device_plan && n == "bptr") { src = h->d_bptr.p; cnt = P.n_owned + 1; }
  else if (h->device_plan && n == "bcol") { src = h->d_bcol.p; cnt = P.n_blocks; }
  else if (h->device_plan && n == "blk_slot") { src = h->d_blk_slot.p; cnt = P.n_blocks; }
  else return fail(FB_EINVAL, "no device plan array '%s'", name);
  if (out && capacity > 0 && cnt > 0) {
    FB_HIP(hipMemcpyAsync(out, src, sizeof(int) * (size_t)std::min(cnt, capacity), hipMemcpyDeviceToHost, h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
  }
  return cnt;
}

int fb_fem_num_nodes(fb_fem_t h) { return h ? h->plan.n_global : 0; }
int fb_fem_num_tets(fb_fem_t h) { return h ? h->plan.n_tets : 0; }
int fb_fem_num_blocks(fb_fem_t h) { return h ? h->plan.n_blocks : 0; }
int fb_fem_matrix_precision(fb_fem_t h) { return h && h->f64 ? FB_MATRIX_F64 : FB_MATRIX_F32; }
int fb_fem_owned_range(fb_fem_t h, int lo_hi[2]) {
  if (!h || !lo_hi) return fail(FB_EINVAL, "null argument");
  lo_hi[0] = h->plan.node_lo; lo_hi[1] = h->plan.node_hi;
  return FB_OK;
}

int fb_fem_pattern(fb_fem_t h, int* bptr, int* bcol) {
  if (!h || !bptr || !bcol) return fail(FB_EINVAL, "null argument");
  FB_HIP(hipSetDevice(h->prm.device));
  FB_TRY(ensure_host_pattern(h));
  const FemPlan& P = h->plan;
  if (h->ren.active) {
    FB_TRY(ensure_caller_pattern(h));
    memcpy(bptr, h->c_bptr.data(), sizeof(int) * (P.n_owned + 1));
    memcpy(bcol, h->c_bcol.data(), sizeof(int) * (size_t)P.n_blocks);
    return FB_OK;
  }
  memcpy(bptr, P.bptr.data(), sizeof(int) * (P.n_owned + 1));
  for (int p = 0; p < P.n_blocks; p++) bcol[p] = P.local2global[P.bcol[p]];
  return FB_OK;
}

int fb_fem_halo_info(fb_fem_t h, int* n_halo_nodes, int* n_neighbour_ranks) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  const FemPlan& P = h->plan;
  if (n_halo_nodes) *n_halo_nodes = P.n_local - P.n_owned;
  int nb = 0;
  for (int q = 0; q + 1 < (int)P.halo_off.size(); q++) nb += P.halo_off[q + 1] > P.halo_off[q] ? 1 : 0;
  if (n_neighbour_ranks) *n_neighbour_ranks = nb;
  return FB_OK;
}

int fb_fem_renumbering(fb_fem_t h, int* span_caller, int* span_internal) {
  if (!h) return 0;
  if (h->span_stale && !h->poisoned && hipSetDevice(h->prm.device) == hipSuccess) {  // (a merged fb_fem_resync_delta left the measurement for now)
    int span = 0;
    double mean = 0.0;
    if (tet_span_device(h->stream, h->plan.n_tets, h->tets.p, h->plan.n_global, nullptr, h->plan_ws, &span, &mean) == FB_OK) {
      h->ren.span_after = span; h->ren.mean_after = mean;
      h->span_stale = false;
    }
  }
  if (span_caller) *span_caller = h->ren.span_before;
  if (span_internal) *span_internal = h->ren.span_after;
  return h->ren.active ? 1 : 0;
}

int fb_fem_owned_nodes(fb_fem_t h, int* ids) {
  if (!h || !ids) return fail(FB_EINVAL, "null argument");
  FB_HIP(hipSetDevice(h->prm.device));
  FB_TRY(ensure_host_order(h));
  const FemPlan& P = h->plan;
  for (int l = 0; l < P.n_owned; l++) ids[l] = h->l2c.empty() ? P.local2global[l] : h->l2c[l];
  return FB_OK;
}

int fb_fem_element_stiffness(fb_fem_t h, int first, int count, double* K0, double* Minv) {
  CHECK_HANDLE(h);
  if (first < 0 || count < 0 || first + count > h->plan.n_tets || !K0) return fail(FB_EINVAL, "element range [%d,%d) outside [0,%d)", first, first + count, h->plan.n_tets);
  const int kChunk = 1 << 16;
  DevBuf<double> dK, dM;
  FB_TRY(dK.alloc((size_t)144 * std::min(count, kChunk)));
  if (Minv) FB_TRY(dM.alloc((size_t)16 * std::min(count, kChunk)));
  for (int done = 0; done < count; done += kChunk) {
    const int n = std::min(kChunk, count - done);
    hipLaunchKernelGGL(k_element_K0_mfma, dim3(ceil_div(n, kWavesPerBlock)), dim3(kBlock), 0, h->stream, first + done, n, h->rest.p,
                       h->lambda, h->mu, dK.p, Minv ? dM.p : nullptr, h->x0.p, h->tets.p);
    FB_HIP(hipGetLastError());
    FB_TRY(dK.download(K0 + (size_t)144 * done, (size_t)144 * n, h->stream));
    if (Minv) FB_TRY(dM.download(Minv + (size_t)16 * done, (size_t)16 * n, h->stream));
  }
  return FB_OK;
}

int fb_fem_time_element_stiffness(fb_fem_t h, int reps, double* seconds_per_pass) {
  CHECK_HANDLE(h);
  if (reps < 1 || !seconds_per_pass) return fail(FB_EINVAL, "bad arguments");
  const int kChunk = 1 << 16, nt = h->plan.n_tets;
  DevBuf<double> dK;
  FB_TRY(dK.alloc((size_t)144 * std::min(nt, kChunk)));  // the 1.15 GB of a 1M-tet K0 set are produced chunk by chunk into the same scratch
  auto pass = [&]() {
    for (int done = 0; done < nt; done += kChunk) {
      const int n = std::min(kChunk, nt - done);
      hipLaunchKernelGGL(k_element_K0_mfma, dim3(ceil_div(n, kWavesPerBlock)), dim3(kBlock), 0, h->stream, done, n, h->rest.p, h->lambda, h->mu, dK.p,
                         (double*)nullptr, h->x0.p, h->tets.p);
    }
  };
  pass();
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; r++) pass();
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  FB_HIP(hipGetLastError());
  float ms = 0;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *seconds_per_pass = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_fem_assemble(fb_fem_t h, const double* u, double* f, double* K_blocks) {
  CHECK_HANDLE(h);
  if (!u) return fail(FB_EINVAL, "null displacement");
  FB_TRY(upload_global_vec(h, u, h->tmp));
  FB_TRY(h->Ad.zero(h->stream));  // stands in for qvel and fext (both unused in raw mode)
  AsmParams ap;
  ap.lambda = h->lambda; ap.mu = h->mu; ap.rho20 = h->prm.rho / 20.0;
  ap.s_k = 1.0; ap.s_m = 0.0; ap.g_k = 0.0; ap.g_m = 0.0; ap.g_a = 0.0; ap.rhs_scale = 0.0; ap.apply_mask = 0;
  if (h->f64) {
    FB_TRY(launch_warp<double>(h, h->tmp.p, nullptr));
    FB_TRY(launch_rows<double>(h, ap, h->Ad.p, h->Ad.p, h->mblk.p, h->r.p, nullptr, nullptr));
  } else {
    FB_TRY(launch_warp<float>(h, h->tmp.p, nullptr));
    FB_TRY(launch_rows<float>(h, ap, h->Ad.p, h->Ad.p, h->mblk.p, h->r.p, nullptr, nullptr));
  }
  h->system_valid = false;
  if (f) FB_TRY(download_owned(h, h->r, f));
  if (K_blocks) FB_TRY(download_blocks(h, K_blocks));
  return FB_OK;
}

int fb_fem_system(fb_fem_t h, double* Keff_blocks, double* rhs) {
  CHECK_HANDLE(h);
  FB_TRY(assemble_system(h));
  if (rhs) FB_TRY(download_owned(h, h->rhs, rhs));
  if (Keff_blocks) FB_TRY(download_blocks(h, Keff_blocks));
  return FB_OK;
}

int fb_fem_mass(fb_fem_t h, double* m_blocks) {
  CHECK_HANDLE(h);
  if (!m_blocks) return fail(FB_EINVAL, "null output");
  std::vector<double> u((size_t)3 * h->plan.n_global, 0.0);
  FB_TRY(fb_fem_assemble(h, u.data(), nullptr, nullptr));
  FB_TRY(ensure_host_pattern(h));
  const FemPlan& P = h->plan;
  std::vector<double> host((size_t)P.n_slots * 64);
  FB_TRY(h->mblk.download(host.data(), host.size(), h->stream));
  const bool mapped = h->ren.active;
  std::vector<double> internal;
  if (mapped) {
    FB_TRY(ensure_caller_pattern(h));
    internal.resize((size_t)P.n_blocks);
  }
  double* dst = mapped ? internal.data() : m_blocks;
  for (int a = 0; a < P.n_owned; a++)
    for (int p = P.bptr[a]; p < P.bptr[a + 1]; p++) dst[p] = host[(size_t)P.blk_slot[p] * 64 + (a & 63)];
  if (mapped)
    for (int pc = 0; pc < P.n_blocks; pc++) m_blocks[pc] = internal[(size_t)h->c_src[pc]];
  return FB_OK;
}

int fb_fem_spmv(fb_fem_t h, const double* x, double* y) {
  CHECK_HANDLE(h);
  if (!x || !y) return fail(FB_EINVAL, "null vector");
  if (!h->system_valid) FB_TRY(assemble_system(h));
  FB_TRY(upload_global_vec(h, x, h->tmp));
  FB_TRY(halo_exchange(h, h->tmp.p));
  FB_TRY(spmv<0>(h, h->tmp.p, h->Ad.p, nullptr, nullptr, 0));
  return download_owned(h, h->Ad, y);
}

int fb_fem_pcg(fb_fem_t h, const double* rhs, double* x, double eps, int max_iter, int* iterations_out) {
  CHECK_HANDLE(h);
  if (!rhs || !x) return fail(FB_EINVAL, "null vector");
  if (!(eps > 0) || max_iter < 0) return fail(FB_EINVAL, "bad PCG parameters");
  if (!h->system_valid) FB_TRY(assemble_system(h));
  FB_TRY(upload_global_vec(h, rhs, h->tmp));
  int iters = 0;
  FB_TRY(pcg_solve(h, h->tmp.p, eps, max_iter, &iters, nullptr));
  if (iterations_out) *iterations_out = iters;
  return download_owned(h, h->x, x);
}

int fb_fem_time_spmv(fb_fem_t h, int reps, double* seconds_per_spmv) {
  CHECK_HANDLE(h);
  if (reps < 1 || !seconds_per_spmv) return fail(FB_EINVAL, "bad arguments");
  if (!h->system_valid) FB_TRY(assemble_system(h));
  // the kernel timed is the one the PCG loop launches every iteration: k_spmv<MT,3> (q = A d + the three merged sums),
  // on the state a solve of the current right-hand side starts from (so its convergence test does not exit early)
  const FemPlan& P = h->plan;
  hipLaunchKernelGGL(k_cg_init, dim3(h->grid), dim3(kBlock), 0, h->stream, P.n_slices, P.n_owned, h->rhs.p, h->invdiag.p, h->x.p, h->r.p, h->d.p,
                     h->part_b.p);
  hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(kBlock), 0, h->stream, h->st.p, h->part_b.p, h->grid, (const double*)nullptr, 1e-30, 1 << 30);
  FB_HIP(hipGetLastError());
  FB_TRY(spmv<3>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, 0));  // warm
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int i = 0; i < reps; i++) FB_TRY(spmv<3>(h, h->d.p, h->Ad.p, h->r.p, h->part_a.p, 0));
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  float ms = 0;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *seconds_per_spmv = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_fem_time_exchange(fb_fem_t h, int reps, double* seconds_per_halo, double* seconds_per_sum) {
  CHECK_HANDLE(h);
  if (reps < 1) return fail(FB_EINVAL, "bad arguments");
  if (seconds_per_halo) *seconds_per_halo = 0.0;
  if (seconds_per_sum) *seconds_per_sum = 0.0;
  if (!h->comm || h->comm->n_ranks == 1) return FB_OK;
  hipStream_t s = h->stream;
  double* sc = nullptr;
  FB_TRY(halo_exchange(h, h->tmp.p));  // warm, and lines the ranks up
  FB_TRY(global_scalar(h, h->part_a.p, &sc, false, 3));
  float ms = 0;
  FB_HIP(hipEventRecord(h->ev[0], s));
  for (int i = 0; i < reps; i++) FB_TRY(halo_exchange(h, h->tmp.p));
  FB_HIP(hipEventRecord(h->ev[1], s));
  for (int i = 0; i < reps; i++) FB_TRY(global_scalar(h, h->part_a.p, &sc, false, 3));
  FB_HIP(hipEventRecord(h->ev[2], s));
  FB_HIP(hipStreamSynchronize(s));
  if (h->p2p) FB_TRY(p2p_check(h->p2p, s));
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  if (seconds_per_halo) *seconds_per_halo = ms * 1e-3 / reps;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[1], h->ev[2]));
  if (seconds_per_sum) *seconds_per_sum = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_fem_time_assembly(fb_fem_t h, int reps, double* seconds_per_assembly) {
  CHECK_HANDLE(h);
  if (reps < 1 || !seconds_per_assembly) return fail(FB_EINVAL, "bad arguments");
  FB_TRY(assemble_system(h));
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int i = 0; i < reps; i++) FB_TRY(assemble_system(h));
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  float ms = 0;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *seconds_per_assembly = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_fem_persist_info(fb_fem_t h, int* waves_per_cu, int* workgroups, int* lds_slots) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  if (waves_per_cu) *waves_per_cu = h->persist ? h->persist_waves : 0;
  if (workgroups) *workgroups = h->persist ? h->persist_blocks : 0;
  if (lds_slots) {
    *lds_slots = 0;
    if (h->persist) *lds_slots = h->pipe_klt;
  }
  return h->persist ? 1 : 0;
}

int fb_fem_pcg_path(fb_fem_t h, char* name, int name_len, int* persist_launches, int* persist_fallbacks, int* max_producers) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  if (name && name_len > 0) {
    if (h->persist && h->shard_persist && h->pipe_rows == 2) snprintf(name, name_len, "k_pcg_pipe2_shard");
    else if (h->persist && h->shard_persist) snprintf(name, name_len, "k_pcg_pipe_shard<%d,%d>", h->pipe_wmax, h->pipe_wmax == 8 ? 8 : 6);
    else if (h->persist && h->pipe_rows == 2) snprintf(name, name_len, "k_pcg_pipe2<%s>", h->c16 ? "c16" : "c32");
    else if (h->persist)
      snprintf(name, name_len, "k_pcg_pipe<float,%s,%d,%d%s>", h->c16 ? "c16" : "c32", h->pipe_wmax, h->pipe_wmax == 5 ? 16 : (h->pipe_wmax == 8 ? 8 : (pipe_klt7(h) ? 7 : 6)),
               h->prm.pcg_variant == FB_PCG_BLOCK_JACOBI ? ",bj" : "");
    else name[0] = 0;
  }
  if (persist_launches) *persist_launches = h->persist_launches;
  if (persist_fallbacks) *persist_fallbacks = h->persist_fallbacks;
  if (max_producers && h->persist && h->pipe_stats_pending) {
    int st[2] = {0, 0};
    FB_TRY(h->pipe_stats.download(st, 2, h->stream));
    h->pipe_max_producers = st[1] ? -1 : st[0];
    h->pipe_stats_pending = false;
  }
  if (max_producers) *max_producers = h->persist ? h->pipe_max_producers : 0;
  return h->last_pcg_path;
}

int fb_fem_persist_rearms(fb_fem_t h) { return h ? h->persist_rearms : 0; }
int fb_fem_persist_helpers(fb_fem_t h) { return h && h->persist ? h->pipe_help_tasks : 0; }
int fb_fem_persist_gather(fb_fem_t h, double* lines_planes, double* lines_records) {
  if (lines_planes) *lines_planes = h && h->persist ? h->pipe_gather_lines[0] : 0.0;
  if (lines_records) *lines_records = h && h->persist ? h->pipe_gather_lines[1] : 0.0;
  return h && h->persist && h->pipe_xyz ? 1 : 0;
}

int fb_fem_persist_stats(fb_fem_t h, int* launches, double* seconds, long long* iterations) {
  if (!h) return fail(FB_EINVAL, "null FEM handle");
  if (launches) *launches = h->persist_launches;
  if (seconds) *seconds = h->persist_seconds;
  if (iterations) *iterations = h->persist_iterations;
  return FB_OK;
}

int fb_fem_time_persist(fb_fem_t h, int reps, int n_iters, double* seconds_per_launch) {
  CHECK_HANDLE(h);
  if (reps < 1 || n_iters < 1 || n_iters > 100000 || !seconds_per_launch) return fail(FB_EINVAL, "bad arguments");
  if (!h->persist) return fail(FB_EINVAL, "this handle does not run the persistent PCG iterations");
  if (!h->system_valid) FB_TRY(assemble_system(h));
  double total = 0.0;
  for (int r = -1; r < reps; r++) {  // r = -1: warm-up
    float ms = 0;
    // a solve of the current right-hand side with a tolerance it cannot reach, cut after n_iters iterations
    FB_HIP(hipEventRecord(h->ev[0], h->stream));
    FB_TRY(launch_pipe(h, h->rhs.p, 1, n_iters, 1e-30, 1 << 30));
    FB_HIP(hipEventRecord(h->ev[1], h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
    FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    if (r >= 0) total += ms * 1e-3;
  }
  unsigned int err = 0;
  FB_HIP(hipMemcpy(&err, h->pipe_flags.p + h->persist_blocks + h->pipe_flag_extra + 4, sizeof err, hipMemcpyDeviceToHost));
  if (err) {
    FB_TRY(h->pipe_flags.zero(h->stream));
    FB_TRY(h->pipe_post.zero(h->stream));
    if (h->shard_persist) {
      // the peers' cumulative box counters are ahead of the flags just cleared: this handle must not launch the sharded kernel again
      // (ADVICE r3); its solves run the two-launch iteration until a re-sync attaches fresh boxes
      h->persist = false;
      h->persist_broken = true;
    }
    return fail(FB_EDEVICE, "persistent PCG: a wait inside the launch timed out");
  }
  h->system_valid = false;
  *seconds_per_launch = total / reps;
  return FB_OK;
}

int fb_fem_iteration_bytes(fb_fem_t h, double* bytes) {
  if (!h || !bytes) return fail(FB_EINVAL, "null argument");
  const FemPlan& P = h->plan;
  // SURVEY.md 8(d), one Jacobi-PCG iteration: the BSR SpMV (nnzb 3x3 blocks with a 2- or 4-byte column id, the row pointers,
  // the low part of the diagonal blocks) + the fused lower bound of the vector traffic, 9 vector streams (x, r, d read and
  // written, q, 1/diag and d read by the product), here in fp64
  const double idx = h->c16 ? 2.0 : 4.0;
  *bytes = (double)P.n_blocks * (9.0 * mt_size(h) + idx) + (P.n_owned + 1) * 4.0 + 6.0 * mt_size(h) * P.n_owned + 9.0 * 3.0 * P.n_owned * 8.0;
  return FB_OK;
}

int fb_fem_spmv_bytes(fb_fem_t h, double* bytes) {
  if (!h || !bytes) return fail(FB_EINVAL, "null argument");
  const FemPlan& P = h->plan;
  // SURVEY.md 8d BSR figure, for the launch the PCG loop makes (k_spmv<MT,3>): nnzb*(9 values + 4 B index) + (rows+1)*4
  // + the low part of each row's diagonal block (9 values per row) + fp64 vectors: d read once, q written once, and the
  // own-row r and 1/diag the merged sums need
  // (the index is 2 bytes where the row kernel reads 16-bit column differences)
  const double idx = (h->c16 && h->split == 0) ? 2.0 : 4.0;
  // (6 of the 9 planes of the symmetric low part are read)
  *bytes = (double)P.n_blocks * (9.0 * mt_size(h) + idx) + (P.n_owned + 1) * 4.0 + 6.0 * mt_size(h) * P.n_owned + 3.0 * P.n_owned * 8.0 * 4.0;
  return FB_OK;
}

int fb_fem_assembly_bytes(fb_fem_t h, double* bytes) {
  if (!h || !bytes) return fail(FB_EINVAL, "null argument");
  const FemPlan& P = h->plan;
  // pass 1: tet ids 16 + rest record 128 + 4 nodes * (x0 + u) 48 + rotated record 16*MT + element force 96
  // pass 2: contribution words 16*4 per tet + Keff values 9*MT per block + index 4 per block + 5 node vectors
  *bytes = (double)P.n_tets * (16.0 + 128.0 + 192.0 + 16.0 * mt_size(h) + 96.0 + 64.0) + (double)P.n_blocks * (9.0 * mt_size(h) + 4.0) +
           3.0 * P.n_owned * 8.0 * 5.0;
  return FB_OK;
}

}  // extern "C"
