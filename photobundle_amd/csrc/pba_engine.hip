// pba_engine.hip -- C-ABI (include/pba.h) of the MI355X photometric bundle-adjustment engine.
//
// Owns all device memory, one HIP stream per engine; every pass is stream-ordered and the host synchronises
// once per step attempt (pba_step) to read a 32-double scalar block.  There is no CPU fallback: pba_create
// fails with PBA_ERR_NO_DEVICE when no GPU is visible.
#include "../../include/pba.h"
#include "pba_comm.h"
#include "pba_internal.h"
#include "pba_kernels.h"
#include "pba_frontend.h"
#include "pba_resident.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <string>
#include <vector>

using namespace pba;

struct pba_engine {
  pba_config cfg{};
  hipStream_t stream = nullptr;
  std::string err;

  // frames
  uint32_t* d_frames = nullptr;     // [max_frames][rows*cols] packed texels
  float* d_frames_mc = nullptr;     // channels > 1: [max_frames][channels][rows*cols] channel VALUES (gradients are formed at use)
  int channels = 1;
  uint8_t* d_img_stage = nullptr;   // [rows*cols]
  bool img_stage_valid = false;     // d_img_stage holds the u8 image of the frame uploaded last (set by the u8 / descriptor / pyr_down
                                    // setters, cleared by the float setters): what the device front-end's ZNCC reads
  uint8_t* h_img_stage = nullptr;   // pinned host copy of the frame being uploaded
  double* h_state_stage = nullptr;  // pinned, host-mapped landing buffer of pba_get_state (grown on demand)
  double* h_state_dev = nullptr;    // its device address
  size_t h_state_cap = 0;           // doubles
  hipEvent_t ev_img_stage = nullptr;
  bool img_stage_busy = false;
  hipEvent_t ev_xdep = nullptr;     // orders another engine's stream against this one (pba_set_frame_pyr_down)
  uint8_t* d_u8_work[2] = {nullptr, nullptr};   // device-side descriptor producers: smoothed frame, census image (on first use)
  // device front-end (pba_frontend_*), everything on first use
  uint8_t* d_fe_mask = nullptr;     // [rows*cols] selection mask (1 = free)
  uint8_t* d_fe_flag = nullptr;     // [rows*cols] candidate flags
  float* d_fe_smap = nullptr;       // [rows*cols] saliency
  float* d_fe_depth = nullptr;      // [rows*cols]
  int32_t* d_fe_rows = nullptr;     // [2 rows + 1] per-row counts | offsets
  pba_candidate* d_fe_cand = nullptr;   // [rows*cols]
  char* d_fe_io = nullptr;          // grow-only device scratch of the visibility / descriptor calls
  char* h_fe_io = nullptr;          // pinned staging (grow-only)
  size_t h_fe_cap = 0;
  bool fe_mask_valid = false;       // pba_frontend_visibility ran for the current frame
  int fe_n_cand = 0;
  std::vector<uint8_t> frame_set;
  uint32_t slot_mask = 0;           // window slots referenced by the observation list

  // problem
  int n_points = 0, n_obs = 0, n_frames = 0, fixed_slot = -1, n_free = 0;
  bool have_problem = false, have_cams = false, have_lin = false;
  bool lin_valid[2] = {false, false};   // rec[k] holds a Jacobian pass of point k
  bool speculate = true;            // candidate pass = Jacobian pass (skips the re-linearisation after an accept)
  int cur = 0;                      // ping-pong index of the current point
  double* d_xyz[2] = {nullptr, nullptr};
  double* d_cams[2] = {nullptr, nullptr};
  CamGeom* d_geom[2] = {nullptr, nullptr};
  float* d_desc = nullptr;
  double* d_w2 = nullptr;
  double* d_rays = nullptr;         // inverse-depth variant: [n_points][6] fixed world rays (null pointer passed otherwise)
  bool inverse_depth = false;
  std::vector<double> h_rays;       // host copy for pba_get_points_world
  int32_t* d_obs_point = nullptr;
  uint8_t* d_obs_slot = nullptr;
  int32_t* d_pt_begin = nullptr;
  int4* d_tile_info = nullptr;
  int2* d_lane_rec = nullptr;       // [n_tiles][kTile] per tile lane: point, slot | first lane << 8 | observations of the point << 16
  int n_tiles = 0;
  // linearisation + solve scratch
  int64_t rec_stride = 0;
  double* d_rec[2] = {nullptr, nullptr};   // [6][rec_stride] per point parity
  double* d_sp = nullptr;           // [n_points][3]
  double* d_ptrec = nullptr;        // [n_points][12]
  double* d_sc = nullptr;           // [2][6 * kMaxFrames]: Jacobi scales | live flags (the [1] part starts at 6 n_free)
  double* d_delta_c = nullptr;      // [kMaxFrames][6]
  double* d_partial = nullptr;      // [schur_grid][part_stride]
  double* d_red = nullptr;          // [kChunks][part_stride]
  double* d_packed = nullptr;       // [part_stride] (the tri layout of pba_solve.h needs packed_stride(n) < part_stride of them)
  uint32_t* d_solve_tab = nullptr;  // window-shape index tables of the reduced solve (solve_tables), rebuilt when n_free changes
  int solve_tab_nf = -1;
  double* d_S = nullptr;            // [n*n] debug copy
  double* d_rhs = nullptr;          // [n]
  double* d_block_cost[2] = {nullptr, nullptr};   // [sample_grid] block costs of the last pass at point parity k
  int32_t* d_block_fail[2] = {nullptr, nullptr};
  int cost_blocks[2] = {0, 0};      // valid entries in d_block_cost[k] (grid of the pass that wrote them)
  double* d_bs_out = nullptr;       // [backsub_grid][3]
  double* d_scal = nullptr;         // [kNumScal]
  double* d_xchg = nullptr;         // [kSumBCount + kMaxCount * world] multi-rank exchange buffer
  double* h_scal = nullptr;         // pinned + mapped: [kNumScal] doubles then one u64 sequence number
  double* h_scal_dev = nullptr;     // device view of h_scal
  unsigned long long seq = 0;
  int64_t jac_passes = 0, cost_passes = 0;
  int schur_grid = 0, sample_grid = 0, backsub_grid = 0, sample_waves = 4, fused_grid = 0;
  bool unit_weights = false;        // all patch weights are exactly 1 (MakePatchWeights without the Gaussian)
  bool fuse = true;                 // back-substitution + finalisation fused into the candidate pass (radius <= 3)
  unsigned int* d_ticket = nullptr;
  unsigned int* d_ticket_solve = nullptr;   // arrival counter of k_reduce_solve
  unsigned long long* d_stamp = nullptr;    // [kStampMaxIters + 1][kStampRecord] device time stamps of the pipelined iterations (pba_set_profiling(e, 2))
  bool stamps = false;
  int stamp_iter = 0;                       // record of the iteration being enqueued (0: the first linearisation)
  int solve_kind = 0;               // PBA_SOLVE: 0 blocked workgroup L D L^T (fused with the reduction at one rank), 1 k_solve_generic (diagnostics)
  // asynchronous driver (device-side trust-region decisions)
  LmState* d_lm = nullptr;          // device state
  LmState* h_lm = nullptr;          // host-mapped mirror, written at every publish
  LmState* h_lm_dev = nullptr;
  pba_iteration_summary* h_log = nullptr;      // host-mapped iteration log
  pba_iteration_summary* h_log_dev = nullptr;
  pba_iteration_summary* d_log = nullptr;      // device iteration log of the asynchronous driver (flushed at the end)
  static constexpr int kMaxLog = 1024;
  int async_cur = 0;                // parity assumed at enqueue time
  bool use_async = true;            // PBA_ASYNC=0 disables
  // resident solve (pba_resident.h): one cooperative launch per pba_solve when the window fits one resident round of workgroups
  bool use_resident = true;         // PBA_RESIDENT=0 disables
  unsigned* d_res_sync = nullptr;   // flag block (kResSyncBytes), zeroed once: epochs grow from launch to launch
  static constexpr unsigned kResEpochLimit = 0xF0000000u;      // the epochs restart (flag block zeroed) before a launch would count beyond this
  unsigned res_epoch = 1;           // first epoch of the next resident launch
  unsigned res_epoch_launch = 1;    // ... of the one in flight / last finished
  int res_groups_max[2][2] = {{-1, -1}, {-1, -1}};   // [radius - 1][unit weights]: co-resident workgroups of k_resident (-1: not asked yet, 0: none)
  int res_groups_n[2][2] = {{-1, -1}, {-1, -1}};     // ... the reduced-system size that answer was given for
  int n_cus = 0, coop_launch = 0;
  int64_t res_launches = 0;
  int last_driver = 0;              // 0 none, 1 resident, 2 pipelined, 3 host-stepped (pba_solve_driver)
  double wait_timeout_s = 120.0;    // PBA_WAIT_TIMEOUT_S: watchdog of the publication waits
  double tick_hz = 1e8;             // rate of s_memrealtime (hipDeviceAttributeWallClockRate; 100 MHz on gfx950): device-side time-outs
  bool poisoned = false;            // a publication wait timed out: the stream still holds the stalled work, every later
                                    // call fails fast and pba_destroy neither waits for the stream nor for the collective
  unsigned long long* d_dbg = nullptr;   // PBA_SCHUR_TIMING diagnostics
  int dbg_left = 0;
  int n_pairs = 0, part_stride = 0;
  static constexpr int kChunks = 32;

  Comm comm;
  unsigned int* h_comm_err = nullptr;      // host-mapped: a peer-exchange wait timed out (k_peer_allreduce)
  unsigned int* h_comm_err_dev = nullptr;

  // counters
  bool profile = false;
  static constexpr int kEvPairs = 7;     // Jacobian pass, cost pass, Schur, reduction, solve, exchange A, exchange B
  hipEvent_t ev[2 * kEvPairs] = {};
  bool ev_used[kEvPairs] = {};
  pba_counters ctr{};
  std::unordered_map<void*, size_t> dev_cap;   // capacity in bytes of every dev_alloc'ed buffer, keyed by its owner field
};

namespace {

double wall_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(pba_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

#define HIP_TRY(e, call)                                                                         \
  do {                                                                                           \
    hipError_t _r = (call);                                                                      \
    if (_r != hipSuccess) return fail((e), PBA_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
  } while (0)

// A timed-out publication wait leaves the stream stalled: every later call returns at once instead of hanging on it.
#define PBA_NOT_POISONED(e) \
  do { if ((e)->poisoned) return fail((e), PBA_ERR_STATE, "the engine is unusable after a timed-out step (stalled stream / collective); destroy it"); } while (0)

// A peer-exchange wait that timed out on the device (k_peer_allreduce / the solve's prologue) reports through a host-mapped
// word.  The report is STICKY: the sums of that step were poisoned (NaN) on the device, the exchange sequence numbers may no
// longer match the peers', so the engine is unusable from then on -- every later call fails with PBA_ERR_COMM / PBA_ERR_STATE.
int comm_failed(pba_engine* e) {
  if (!e->h_comm_err) return PBA_OK;
  const unsigned int w = *reinterpret_cast<volatile unsigned int*>(e->h_comm_err);
  if (!w) return PBA_OK;
  e->poisoned = true;
  return fail(e, PBA_ERR_COMM, "peer exchange timed out after %.0f s waiting for rank %u", 0.5 * e->wait_timeout_s, w - 1);
}

// Grow-only device buffers: a sliding-window caller re-submits a problem of similar size for every frame, and a
// hipFree + hipMalloc pair per buffer per frame is pure overhead.  Contents are undefined after the call (every user
// overwrites or uploads the whole buffer).
template <class T>
int dev_alloc(pba_engine* e, T** p, size_t n) {
  if (n == 0) n = 1;
  const size_t bytes = n * sizeof(T);
  auto it = e->dev_cap.find(static_cast<void*>(p));
  if (*p && it != e->dev_cap.end() && it->second >= bytes) return PBA_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  const size_t want = bytes + bytes / 8;            // a little headroom against frame-to-frame jitter
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(p), want));
  e->dev_cap[static_cast<void*>(p)] = want;
  return PBA_OK;
}
template <class T>
void dev_free(T** p) { if (*p) { (void)hipFree(*p); *p = nullptr; } }

// Call-order / consistency check before any pass touches the device (include/pba.h: PBA_ERR_STATE).
int check_ready(pba_engine* e, const char* who) {
  if (e->poisoned) return fail(e, PBA_ERR_STATE, "%s: the engine is unusable after a timed-out step (stalled stream / collective); destroy it", who);
  if (!e->have_problem || !e->have_cams)
    return fail(e, PBA_ERR_STATE, "call order violated: %s before set_problem/set_cameras", who);
  for (int s = 0; s < kMaxFrames; ++s) {
    if (!((e->slot_mask >> s) & 1u)) continue;
    if (s >= e->n_frames) return fail(e, PBA_ERR_STATE, "call order violated: an observation uses slot %d but only %d cameras are set", s, e->n_frames);
    if (!e->frame_set[s]) return fail(e, PBA_ERR_STATE, "call order violated: an observation uses slot %d but no frame was uploaded to it", s);
  }
  return PBA_OK;
}

constexpr int kSampleWaves = 4;    // 256-thread workgroups at every patch radius (two 128-observation tiles when fused)

template <int R, bool JAC, bool FUSED>
void launch_sample_r(pba_engine* e, const SampleParams& sp) {
  const int grid = FUSED ? e->fused_grid : e->sample_grid;
  const dim3 block(kSampleWaves * 64);
  if constexpr (FUSED) {
    // the reduced-precision sweep modes never take the fused path (fused_capable)
    if (e->unit_weights) hipLaunchKernelGGL((k_sample<R, JAC, kSampleWaves, true, true, false>), dim3(grid), block, 0, e->stream, sp);
    else hipLaunchKernelGGL((k_sample<R, JAC, kSampleWaves, true, false, false>), dim3(grid), block, 0, e->stream, sp);
  } else {
    if (e->unit_weights && sp.prec != 0) hipLaunchKernelGGL((k_sample<R, JAC, kSampleWaves, false, true, true>), dim3(grid), block, 0, e->stream, sp);
    else if (e->unit_weights) hipLaunchKernelGGL((k_sample<R, JAC, kSampleWaves, false, true, false>), dim3(grid), block, 0, e->stream, sp);
    else hipLaunchKernelGGL((k_sample<R, JAC, kSampleWaves, false, false, false>), dim3(grid), block, 0, e->stream, sp);
  }
}
template <int R, bool JAC, bool FUSED>
void launch_sample_mc_r(pba_engine* e, const SampleParams& sp) {
  const dim3 grid(FUSED ? e->fused_grid : e->sample_grid), block(kSampleWaves * 64);
  if (e->unit_weights) hipLaunchKernelGGL((k_sample_mc<R, JAC, kSampleWaves, FUSED, true>), grid, block, 0, e->stream, sp, (const float*)e->d_frames_mc, e->channels);
  else hipLaunchKernelGGL((k_sample_mc<R, JAC, kSampleWaves, FUSED, false>), grid, block, 0, e->stream, sp, (const float*)e->d_frames_mc, e->channels);
}
template <bool JAC, bool FUSED = false>
void launch_sample(pba_engine* e, const SampleParams& sp) {
  if (e->channels > 1) {
    switch (e->cfg.radius) {
      case 1: launch_sample_mc_r<1, JAC, FUSED>(e, sp); break;
      case 2: launch_sample_mc_r<2, JAC, FUSED>(e, sp); break;
      case 3: launch_sample_mc_r<3, JAC, FUSED>(e, sp); break;
      case 4: launch_sample_mc_r<4, JAC, FUSED>(e, sp); break;
      default: launch_sample_mc_r<5, JAC, FUSED>(e, sp); break;
    }
    return;
  }
  switch (e->cfg.radius) {
    case 1: launch_sample_r<1, JAC, FUSED>(e, sp); break;
    case 2: launch_sample_r<2, JAC, FUSED>(e, sp); break;
    case 3: launch_sample_r<3, JAC, FUSED>(e, sp); break;
    case 4: launch_sample_r<4, JAC, FUSED>(e, sp); break;
    default: launch_sample_r<5, JAC, FUSED>(e, sp); break;
  }
}
// one kernel for back-substitution + candidate pass + step finalisation, at every patch radius; the opt-in
// reduced-precision sampler modes (pba_config.flags bits 1-2) keep the unfused kernels
bool fused_capable(const pba_engine* e) { return e->fuse && ((e->cfg.flags >> 1) & 3) == 0; }
int sample_waves_for_radius(int) { return kSampleWaves; }

__global__ void k_noop() {}
// initial trust-region state of a solve, passed by value (a device-to-device hipMemcpyAsync of the host-mapped mirror goes through
// the copy engine: its hand-over sat in front of the first kernel of every solve)
#ifndef PBA_LM_INIT_KERNEL
#define PBA_LM_INIT_KERNEL 0
#endif
__global__ void k_lm_init(LmState* dst, LmState st) { if (threadIdx.x == 0) *dst = st; }
// device -> host-mapped pinned memory (8-byte words, grid-stride); visible to the host once the stream has drained
__global__ void k_to_host(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// A timing bracket that starts on an idle stream would charge the kernel with the host's launch latency (the begin
// event is stamped at once, the kernel arrives microseconds later): a no-op kernel in front absorbs it.
void ev_begin(pba_engine* e, int k) {
  if (e->profile) {
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(1), 0, e->stream);
    (void)hipEventRecord(e->ev[2 * k], e->stream);
  }
}
void ev_end(pba_engine* e, int k) { if (e->profile) { (void)hipEventRecord(e->ev[2 * k + 1], e->stream); e->ev_used[k] = true; } }
void ev_collect(pba_engine* e) {
  if (!e->profile) return;
  double* acc[pba_engine::kEvPairs] = {&e->ctr.linearize_ms, &e->ctr.cost_ms, &e->ctr.schur_ms, &e->ctr.solve_ms, &e->ctr.solve_ms,
                                       &e->ctr.exchange_ms, &e->ctr.exchange_ms};
  int64_t* cnt[pba_engine::kEvPairs] = {&e->ctr.linearize_launches, &e->ctr.cost_launches, &e->ctr.schur_launches, &e->ctr.solve_launches,
                                        nullptr, &e->ctr.exchange_launches, nullptr};
  for (int k = 0; k < pba_engine::kEvPairs; ++k) {
    if (!e->ev_used[k]) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[2 * k], e->ev[2 * k + 1]) == hipSuccess) { *acc[k] += ms; if (cnt[k]) *cnt[k] += 1; }
    e->ev_used[k] = false;
  }
}

void launch_solve(pba_engine* e, const SolveParams& so, int n) {
  if (e->solve_kind == 0) {
    if (e->n_free > kSolveNarrowFree) hipLaunchKernelGGL((k_solve_blocked<kSolveWideThreads>), dim3(1), dim3(kSolveWideThreads), solve_blocked_smem_bytes(n), e->stream, so);
    else hipLaunchKernelGGL((k_solve_blocked<kSolveBlockedThreads>), dim3(1), dim3(kSolveBlockedThreads), solve_blocked_smem_bytes(n), e->stream, so);
    return;
  }
  const size_t solve_smem = sizeof(double) * ((size_t)n * (n + 1) + 5 * n);
  hipLaunchKernelGGL(k_solve_generic, dim3(1), dim3(kSolveThreads), solve_smem, e->stream, so);
}

// record of the iteration being enqueued in the device time-stamp log (null: stamps off / beyond the log)
unsigned long long* stamp_record(pba_engine* e) {
  if (!e->stamps || !e->d_stamp || e->stamp_iter > kStampMaxIters) return nullptr;
  return e->d_stamp + (size_t)e->stamp_iter * kStampRecord;
}

// ---- peer exchange (pba_comm.h): kind 0 = packed reduced system, 1 = step scalars -------------------------------------
PeerParams peer_params(const pba_engine* e) {
  PeerParams pp{};
  for (int q = 0; q < Comm::kMaxPeers; ++q) pp.mb[q] = e->comm.mb_peer[q];
  pp.own = e->comm.mb_own; pp.world = e->comm.world; pp.rank = e->comm.rank;
  return pp;
}
// where the producer of the NEXT exchange of `kind` writes this rank's contribution
double* peer_slot(pba_engine* e, int kind) {
  const unsigned long long x = (kind == 0 ? e->comm.seq_a : e->comm.seq_b) + 1;
  return e->comm.mb_own + Comm::data_offset(kind, x);
}
// the exchange itself: flags + rank-ordered sum of n doubles into `out` (device memory of this rank)
int peer_allreduce(pba_engine* e, int kind, int n, double* out) {
  const unsigned long long x = ++(kind == 0 ? e->comm.seq_a : e->comm.seq_b);
  // device-side wait: half the host watchdog (100 MHz s_memrealtime), so that the recoverable PBA_ERR_COMM wins the race
  const unsigned long long ticks = (unsigned long long)(0.5 * e->wait_timeout_s * e->tick_hz);
  const int grid = std::max(1, std::min(32, (n + 255) / 256));
  hipLaunchKernelGGL(k_peer_allreduce, dim3(grid), dim3(256), 0, e->stream, peer_params(e), (int)Comm::flag_index(kind, x),
                     (unsigned long long)Comm::data_offset(kind, x), n, x, out, ticks, e->h_comm_err_dev);
  HIP_TRY(e, hipGetLastError());
  return PBA_OK;
}

// Reduction of the Schur partials + reduced solve.  One launch (k_reduce_solve) at a single rank; with more ranks the
// exchange of the packed sums sits between the two, so they stay separate kernels.  With the peer exchange there is no
// exchange kernel: the reduction's last workgroup raises this rank's mailbox flag, the solve's prologue waits for every
// rank's flag and sums the mailbox slots in rank order (pba_solve.h).
int launch_reduce_and_solve(pba_engine* e, SolveParams so, int n, int cur, int cand, const LmState* lm, int final_pass, int n_cost_blocks,
                            const ReduceSolveParams::Fin* fin = nullptr) {
  const bool multi = e->comm.multi();
  // The gradient-only pass neither forms nor reduces the pair blocks and the right-hand side: the reduced-system test hook
  // (pba_get_reduced_system) keeps the system of the last FULL step instead of being overwritten with entries nobody computed
  if (final_pass && !so.init_scale) { so.S_dbg = nullptr; so.rhs_dbg = nullptr; }
  const int grid = (e->part_stride + kReduceEntries - 1) / kReduceEntries + 1;
  const int pstride = packed_stride(n);
  ReduceParams rp{};
  rp.partial = e->d_partial; rp.n_blocks = e->schur_grid; rp.stride = e->part_stride; rp.n_free = e->n_free; rp.n_pairs = e->n_pairs;
  rp.block_cost = e->d_block_cost[cur]; rp.block_fail = e->d_block_fail[cur]; rp.n_cost_blocks = n_cost_blocks;
  rp.packed = e->d_packed; rp.scal = e->d_scal;
  if (!multi && e->solve_kind == 0 && !(PBA_PHASE_TIMING && getenv("PBA_SPLIT_SOLVE"))) {
    ReduceSolveParams rsp{};
    rsp.rp = rp;
    rsp.block_cost_alt = e->d_block_cost[cand]; rsp.block_fail_alt = e->d_block_fail[cand];
    rsp.ticket = e->d_ticket_solve; rsp.so = so;
    if (fin) {
      rsp.fin = *fin;
      // gradient-only: the pair blocks and the right-hand side of the partials are not consumed
      if (final_pass && !so.init_scale) rsp.rp.first_entry = 36 * e->n_pairs + n;
    }
    rsp.stamp = (lm && !final_pass) ? stamp_record(e) : nullptr;
    ev_begin(e, 3);
    hipLaunchKernelGGL(k_reduce_solve, dim3(grid), dim3(kReduceThreads), solve_blocked_smem_bytes(n), e->stream, rsp);
    ev_end(e, 3);
    HIP_TRY(e, hipGetLastError());
    return PBA_OK;
  }
  const bool peer = multi && e->comm.peer;
  if (peer && (size_t)pstride > Comm::kCapA) return fail(e, PBA_ERR_COMM, "reduced system of %d doubles exceeds the peer mailbox", pstride);
  ReduceFinalParams fp{};
  fp.rp = rp; fp.lm = lm; fp.enq_cur = cur; fp.final_pass = final_pass;
  fp.block_cost_alt = e->d_block_cost[cand]; fp.block_fail_alt = e->d_block_fail[cand];
  fp.sys_stores = peer ? 1 : 0; fp.ticket = e->d_ticket_solve; fp.peer_flag = -1;
  if (peer) {
    const unsigned long long x = ++e->comm.seq_a;
    fp.rp.packed = e->comm.mb_own + Comm::data_offset(0, x);
    fp.peer_own = e->comm.mb_own; fp.peer_flag = (int)Comm::flag_index(0, x); fp.peer_seq = x;
    so.peer = peer_params(e); so.peer_world = e->comm.world; so.peer_flag = (int)Comm::flag_index(0, x);
    so.peer_off = (unsigned long long)Comm::data_offset(0, x); so.peer_seq = x;
    so.peer_timeout = (unsigned long long)(0.5 * e->wait_timeout_s * e->tick_hz);      // device-side wait: half the host watchdog, so that the recoverable error wins
    so.peer_err = e->h_comm_err_dev;
  }
  ev_begin(e, 3);
  hipLaunchKernelGGL(k_reduce_final, dim3(grid), dim3(kReduceThreads), 0, e->stream, fp);
  ev_end(e, 3);
  HIP_TRY(e, hipGetLastError());
  if (multi && !peer) {
    ev_begin(e, 5);
    if (e->comm.allreduce_device(e->d_packed, (size_t)pstride, 0, e->stream))
      return fail(e, PBA_ERR_COMM, "allreduce(reduced system) failed: %s", e->comm.err.c_str());
    ev_end(e, 5);
  }
  ev_begin(e, 4);
  launch_solve(e, so, n);
  ev_end(e, 4);
  HIP_TRY(e, hipGetLastError());
  return PBA_OK;
}

void launch_schur(pba_engine* e, const SchurParams& sp) {
  hipLaunchKernelGGL(k_schur, dim3(e->schur_grid), dim3(kTile), 0, e->stream, sp);
}

SampleParams make_sample_params(pba_engine* e, int which_point) {
  const int which_out = which_point;
  SampleParams sp{};
  sp.prec = (e->cfg.flags >> 1) & 3;
  sp.frames = e->d_frames;
  sp.geom = e->d_geom[which_point];
  sp.xyz = e->d_xyz[which_point];
  sp.rays = e->inverse_depth ? e->d_rays : nullptr;
  sp.desc = e->d_desc;
  sp.w2 = e->d_w2;
  sp.obs_point = e->d_obs_point;
  sp.obs_slot = e->d_obs_slot;
  sp.rec = e->d_rec[which_point];
  sp.block_cost = e->d_block_cost[which_out];
  sp.block_fail = e->d_block_fail[which_out];
  sp.rec_stride = e->rec_stride;
  sp.n_obs = e->n_obs;
  sp.n_frames = e->n_frames;
  sp.rows = e->cfg.rows;
  sp.cols = e->cfg.cols;
  sp.fx = e->cfg.fx; sp.fy = e->cfg.fy; sp.cx = e->cfg.cx; sp.cy = e->cfg.cy;
  sp.huber = e->cfg.huber;
  return sp;
}

// ONE sum all-reduce for the step scalars of all ranks (sum group + rank-slotted max group, see k_xchg_pack)
int ensure_xchg(pba_engine* e) {
  // grow-only and cheap when large enough: a transport re-initialised with a larger world must not overrun the buffer
  return dev_alloc(e, &e->d_xchg, (size_t)kSumBCount + (size_t)kMaxCount * e->comm.world);
}

// packed == true: the fused sampling kernel's last workgroup already filled the exchange buffer
int exchange_step_scalars(pba_engine* e, bool packed) {
  const int world = e->comm.world;
  const size_t n = (size_t)kSumBCount + (size_t)kMaxCount * world;
  int rc = ensure_xchg(e);
  if (rc) return rc;
  if (!packed) {
    hipLaunchKernelGGL(k_xchg_pack, dim3(1), dim3(64), 0, e->stream, e->d_scal, e->comm.peer ? peer_slot(e, 1) : e->d_xchg, e->comm.rank, world,
                       e->comm.peer ? 1 : 0);
    HIP_TRY(e, hipGetLastError());
  }
  ev_begin(e, 6);
  if (e->comm.peer) {
    const int rcp = peer_allreduce(e, 1, (int)n, e->d_xchg);
    if (rcp) return rcp;
  } else if (e->comm.allreduce_device(e->d_xchg, n, 0, e->stream)) {
    return fail(e, PBA_ERR_COMM, "allreduce(step scalars) failed: %s", e->comm.err.c_str());
  }
  ev_end(e, 6);
  return PBA_OK;
}

}  // namespace

extern "C" {

const char* pba_status_string(int s) {
  switch (s) {
    case PBA_OK: return "ok";
    case PBA_ERR_INVALID: return "invalid argument";
    case PBA_ERR_HIP: return "HIP runtime error";
    case PBA_ERR_NO_DEVICE: return "no HIP device (the engine has no CPU fallback)";
    case PBA_ERR_STATE: return "call order violated";
    case PBA_ERR_COMM: return "collective transport error";
    case PBA_ERR_NUMERIC: return "non-finite evaluation";
    default: return "unknown";
  }
}

const char* pba_last_error(const pba_engine* e) { return e ? e->err.c_str() : ""; }

void pba_default_solver_options(pba_solver_options* o) {
  o->max_num_iterations = 500;              // reference photobundle.cc:751
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;             // :756
  o->gradient_tolerance = 1e-6;             // :757
  o->parameter_tolerance = 1e-6;            // :758
  o->initial_trust_region_radius = 1e4;     // Ceres defaults
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1;
  o->verbose = 0;
}

static int ensure_state_stage(pba_engine* e, size_t doubles) {
  if (e->h_state_cap >= doubles) return PBA_OK;
  if (e->h_state_stage) { (void)hipHostFree(e->h_state_stage); e->h_state_stage = nullptr; e->h_state_cap = 0; }
  const size_t want = doubles + doubles / 8;
  HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_state_stage), sizeof(double) * want, hipHostMallocMapped));
  HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_state_dev), e->h_state_stage, 0));
  e->h_state_cap = want;
  return PBA_OK;
}

int pba_create(const pba_config* cfg, pba_engine** out) {
  if (!cfg || !out) return PBA_ERR_INVALID;
  *out = nullptr;
  if (cfg->rows < 8 || cfg->cols < 8 || cfg->max_frames < 2 || cfg->max_frames > kMaxFrames || cfg->radius < 1 ||
      cfg->radius > kMaxRadius || (int64_t)cfg->rows * cfg->cols * cfg->max_frames >= (1ll << 31))
    return PBA_ERR_INVALID;
  if (cfg->channels < 0 || cfg->channels > PBA_MAX_CHANNELS ||
      (int64_t)cfg->rows * cfg->cols * cfg->max_frames * std::max(1, cfg->channels) >= (1ll << 31))
    return PBA_ERR_INVALID;
  if (cfg->channels > 1 && ((cfg->flags >> 1) & 3) != 0) return PBA_ERR_INVALID;   // the sweep modes are single-channel
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || cfg->device < 0 || cfg->device >= n_dev) {
    (void)hipGetLastError();
    return PBA_ERR_NO_DEVICE;
  }
  pba_engine* e = new pba_engine();
  e->cfg = *cfg;
  int rc = PBA_OK;
  auto bail = [&](int code) { pba_destroy(e); return code; };
  if (hipSetDevice(cfg->device) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(PBA_ERR_HIP);
  e->comm.stream = e->stream;
  const size_t npix = (size_t)cfg->rows * cfg->cols;
  if ((rc = dev_alloc(e, &e->d_frames, npix * cfg->max_frames))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_img_stage, npix))) return bail(rc);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_img_stage), npix, hipHostMallocDefault) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipEventCreateWithFlags(&e->ev_img_stage, hipEventDisableTiming) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipMemsetAsync(e->d_frames, 0, npix * cfg->max_frames * sizeof(uint32_t), e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  e->channels = std::max(1, cfg->channels);
  if (e->channels > 1) {
    if ((rc = dev_alloc(e, &e->d_frames_mc, npix * cfg->max_frames * e->channels))) return bail(rc);
    if (hipMemsetAsync(e->d_frames_mc, 0, npix * cfg->max_frames * e->channels * sizeof(float), e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  }
  e->frame_set.assign(cfg->max_frames, 0);
  for (int k = 0; k < 2; ++k) {
    if ((rc = dev_alloc(e, &e->d_cams[k], 6 * kMaxFrames))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_geom[k], kMaxFrames))) return bail(rc);
  }
  if ((rc = dev_alloc(e, &e->d_sc, 2 * 6 * kMaxFrames))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_delta_c, 6 * kMaxFrames))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_scal, (size_t)kNumScal))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_S, (size_t)36 * kMaxFrames * kMaxFrames))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_rhs, (size_t)6 * kMaxFrames))) return bail(rc);
  if (hipMemsetAsync(e->d_scal, 0, kNumScal * sizeof(double), e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_scal), (kNumScal + 1) * sizeof(double), hipHostMallocMapped) != hipSuccess) return bail(PBA_ERR_HIP);
  std::memset(e->h_scal, 0, (kNumScal + 1) * sizeof(double));
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_scal_dev), e->h_scal, 0) != hipSuccess) return bail(PBA_ERR_HIP);
  if (const char* sv = getenv("PBA_SPECULATE")) e->speculate = atoi(sv) != 0;
  if (const char* sv = getenv("PBA_FUSE")) e->fuse = atoi(sv) != 0;
  if (const char* sv = getenv("PBA_ASYNC")) e->use_async = atoi(sv) != 0;
  if (const char* sv = getenv("PBA_RESIDENT")) e->use_resident = atoi(sv) != 0;
  if (const char* sv = getenv("PBA_RES_EPOCH0")) e->res_epoch = e->res_epoch_launch = (unsigned)strtoul(sv, nullptr, 0);      // test aid: start next to the restart of the epochs
  (void)hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
  (void)hipDeviceGetAttribute(&e->coop_launch, hipDeviceAttributeCooperativeLaunch, cfg->device);
  if ((rc = dev_alloc(e, &e->d_res_sync, kResSyncBytes / sizeof(unsigned)))) return bail(rc);
  if (hipMemsetAsync(e->d_res_sync, 0, kResSyncBytes, e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  if (const char* sv = getenv("PBA_WAIT_TIMEOUT_S")) { const double v = atof(sv); if (v > 0.0) e->wait_timeout_s = v; }
  { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg->device) == hipSuccess && khz > 0) e->tick_hz = 1e3 * khz; }
  if ((rc = dev_alloc(e, &e->d_lm, (size_t)1))) return bail(rc);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_lm), sizeof(LmState), hipHostMallocMapped) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_lm_dev), e->h_lm, 0) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_log), sizeof(pba_iteration_summary) * pba_engine::kMaxLog, hipHostMallocMapped) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_log_dev), e->h_log, 0) != hipSuccess) return bail(PBA_ERR_HIP);
  if ((rc = dev_alloc(e, &e->d_log, (size_t)pba_engine::kMaxLog))) return bail(rc);
  if (const char* sv = getenv("PBA_SCHUR_TIMING")) {
    if (PBA_PHASE_TIMING) e->dbg_left = atoi(sv);
    else std::fprintf(stderr, "PBA_SCHUR_TIMING ignored: libpba_hip.so was built without the phase stamps (make TIMING=1)\n");
  }
  if ((rc = dev_alloc(e, &e->d_ticket, (size_t)1))) return bail(rc);
  if (hipMemsetAsync(e->d_ticket, 0, sizeof(unsigned int), e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_comm_err), sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) return bail(PBA_ERR_HIP);
  *e->h_comm_err = 0;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_comm_err_dev), e->h_comm_err, 0) != hipSuccess) return bail(PBA_ERR_HIP);
  if ((rc = dev_alloc(e, &e->d_ticket_solve, (size_t)1))) return bail(rc);
  if (hipMemsetAsync(e->d_ticket_solve, 0, sizeof(unsigned int), e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  if (const char* sv = getenv("PBA_SOLVE")) e->solve_kind = atoi(sv);
  for (int k = 0; k < 2 * pba_engine::kEvPairs; ++k)
    if (hipEventCreate(&e->ev[k]) != hipSuccess) return bail(PBA_ERR_HIP);
  if (hipEventCreateWithFlags(&e->ev_xdep, hipEventDisableTiming) != hipSuccess) return bail(PBA_ERR_HIP);
  e->sample_waves = sample_waves_for_radius(cfg->radius);
  if ((rc = ensure_state_stage(e, (size_t)6 * kMaxFrames + 3 * 65536))) return bail(rc);   // grown on demand beyond 64k points
  // The first frame-sized host -> device DMA of a process costs ~8 ms (seen in the drop-in class: first
  // pba_set_frame_u8): paid here, from the engine's own pinned buffer
  if (hipMemcpyAsync(e->d_img_stage, e->h_img_stage, npix, hipMemcpyHostToDevice, e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  // loads this library's code object now rather than at the first frame (10+ ms in a process that has not touched it yet)
  hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, e->stream);
  if (hipGetLastError() != hipSuccess) return bail(PBA_ERR_HIP);
  // (The first COOPERATIVE launch of a process -- the resident solve's kind -- sets up a queue of its own, ~10 ms.  It is NOT paid here:
  // a cooperative queue is exclusive on the device, so two processes that share one GPU -- two ranks on one device in the tests -- then
  // time-slice at ~20 ms per LM step (measured: 139 us -> 22.5 ms), and rocprofv3 crashes in its exit handlers once a process has made
  // a cooperative launch (its output files are complete by then).  The first resident solve pays it; multi-rank engines never make one.)
  if (hipStreamSynchronize(e->stream) != hipSuccess) return bail(PBA_ERR_HIP);
  *out = e;
  return PBA_OK;
}

void pba_destroy(pba_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  if (e->poisoned) {
    // the stream holds work that will never finish (pba_step / pba_solve timed out): waiting for it, freeing memory it
    // may still touch or destroying its stream would hang or fault; abort the communicator and leak the device side
    e->comm.shutdown(true);
    delete e;
    return;
  }
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  e->comm.shutdown();
  dev_free(&e->d_frames); dev_free(&e->d_img_stage); dev_free(&e->d_frames_mc);
  for (int k = 0; k < 2; ++k) { dev_free(&e->d_xyz[k]); dev_free(&e->d_cams[k]); dev_free(&e->d_geom[k]); dev_free(&e->d_block_cost[k]); dev_free(&e->d_block_fail[k]); }
  dev_free(&e->d_rays);
  dev_free(&e->d_desc); dev_free(&e->d_w2); dev_free(&e->d_obs_point); dev_free(&e->d_obs_slot); dev_free(&e->d_pt_begin);
  dev_free(&e->d_tile_info); dev_free(&e->d_lane_rec); dev_free(&e->d_rec[0]); dev_free(&e->d_rec[1]); dev_free(&e->d_sp); dev_free(&e->d_ptrec); dev_free(&e->d_sc);
  dev_free(&e->d_delta_c); dev_free(&e->d_partial); dev_free(&e->d_red); dev_free(&e->d_packed); dev_free(&e->d_solve_tab); dev_free(&e->d_S);
  dev_free(&e->d_rhs); dev_free(&e->d_bs_out); dev_free(&e->d_scal); dev_free(&e->d_xchg); dev_free(&e->d_ticket); dev_free(&e->d_ticket_solve); dev_free(&e->d_stamp);
  if (e->h_scal) (void)hipHostFree(e->h_scal);
  if (e->h_lm) (void)hipHostFree(e->h_lm);
  if (e->h_comm_err) (void)hipHostFree(e->h_comm_err);
  if (e->h_img_stage) (void)hipHostFree(e->h_img_stage);
  if (e->h_state_stage) (void)hipHostFree(e->h_state_stage);
  if (e->ev_img_stage) (void)hipEventDestroy(e->ev_img_stage);
  if (e->ev_xdep) (void)hipEventDestroy(e->ev_xdep);
  dev_free(&e->d_u8_work[0]); dev_free(&e->d_u8_work[1]);
  dev_free(&e->d_fe_mask); dev_free(&e->d_fe_flag); dev_free(&e->d_fe_smap); dev_free(&e->d_fe_depth); dev_free(&e->d_fe_rows);
  dev_free(&e->d_fe_cand); dev_free(&e->d_fe_io);
  if (e->h_fe_io) (void)hipHostFree(e->h_fe_io);
  if (e->h_log) (void)hipHostFree(e->h_log);
  dev_free(&e->d_lm);
  dev_free(&e->d_log);
  dev_free(&e->d_res_sync);
  for (int k = 0; k < 2 * pba_engine::kEvPairs; ++k) if (e->ev[k]) (void)hipEventDestroy(e->ev[k]);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int pba_set_frame_channels_f32(pba_engine* e, int slot, int32_t n_channels, const float* channels) {
  if (!e || !channels || slot < 0 || slot >= e->cfg.max_frames) return PBA_ERR_INVALID;
  if (e->channels <= 1 || n_channels != e->channels)
    return fail(e, PBA_ERR_INVALID, "pba_set_frame_channels_f32: the engine was created for %d channel(s), got %d", e->channels, n_channels);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t npix = (size_t)e->cfg.rows * e->cfg.cols;
  // the slot keeps the channel VALUES as they come ([C][rows*cols], the layout of the argument): one copy, no kernel; the
  // caller's buffer is only borrowed for the call
  HIP_TRY(e, hipMemcpyAsync(e->d_frames_mc + (size_t)slot * n_channels * npix, channels, (size_t)n_channels * npix * sizeof(float),
                            hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->img_stage_valid = false;       // no u8 image behind this frame: the device front-end's ZNCC has nothing to read
  e->frame_set[slot] = 1;
  return PBA_OK;
}

int pba_set_frame_u8(pba_engine* e, int slot, const uint8_t* image) {
  if (!e || !image || slot < 0 || slot >= e->cfg.max_frames) return PBA_ERR_INVALID;
  if (e->channels > 1) return fail(e, PBA_ERR_INVALID, "pba_set_frame_u8: the engine was created for %d channels (pba_set_frame_channels_f32)", e->channels);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t npix = (size_t)e->cfg.rows * e->cfg.cols;
  // The caller's buffer is only borrowed for the call: it is copied into a pinned staging buffer here, and the upload +
  // packing run asynchronously behind the return (a pageable hipMemcpy plus a stream sync cost ~1.2 ms per frame).
  // The previous use of the staging buffers is awaited first; it normally finished long ago.
  if (e->img_stage_busy) { HIP_TRY(e, hipEventSynchronize(e->ev_img_stage)); e->img_stage_busy = false; }
  std::memcpy(e->h_img_stage, image, npix);
  HIP_TRY(e, hipMemcpyAsync(e->d_img_stage, e->h_img_stage, npix, hipMemcpyHostToDevice, e->stream));
  dim3 grid((e->cfg.cols + 255) / 256, e->cfg.rows);
  hipLaunchKernelGGL(k_pack_frame, grid, dim3(256), 0, e->stream, e->d_img_stage, e->d_frames + npix * slot,
                     e->cfg.rows, e->cfg.cols);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipEventRecord(e->ev_img_stage, e->stream));
  e->img_stage_busy = true;
  e->img_stage_valid = true;
  e->frame_set[slot] = 1;
  return PBA_OK;
}

int pba_get_frame_planes(pba_engine* e, int slot, float* I, float* Gx, float* Gy) {
  if (!e || slot < 0 || slot >= e->cfg.max_frames || !I || !Gx || !Gy) return PBA_ERR_INVALID;
  if (e->channels > 1) return fail(e, PBA_ERR_INVALID, "pba_get_frame_planes reads the single-channel planes");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t npix = (size_t)e->cfg.rows * e->cfg.cols;
  float* d = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d), 3 * npix * sizeof(float)));
  hipLaunchKernelGGL(k_unpack_frame, dim3((npix + 255) / 256), dim3(256), 0, e->stream, e->d_frames + npix * slot, d,
                     d + npix, d + 2 * npix, (int)npix);
  hipError_t r = hipMemcpyAsync(I, d, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipMemcpyAsync(Gx, d + npix, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipMemcpyAsync(Gy, d + 2 * npix, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (r != hipSuccess) return fail(e, PBA_ERR_HIP, "pba_get_frame_planes: %s", hipGetErrorString(r));
  return PBA_OK;
}

int pba_get_frame_channel(pba_engine* e, int slot, int32_t channel, float* I, float* Gx, float* Gy) {
  if (!e || slot < 0 || slot >= e->cfg.max_frames || !I || !Gx || !Gy) return PBA_ERR_INVALID;
  if (e->channels <= 1 || channel < 0 || channel >= e->channels)
    return fail(e, PBA_ERR_INVALID, "pba_get_frame_channel: channel %d of an engine with %d channel(s)", channel, e->channels);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t npix = (size_t)e->cfg.rows * e->cfg.cols;
  float* d = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d), 3 * npix * sizeof(float)));
  hipLaunchKernelGGL(k_unpack_channel, dim3((e->cfg.cols + 255) / 256, e->cfg.rows), dim3(256), 0, e->stream,
                     (const float*)e->d_frames_mc + ((size_t)slot * e->channels + channel) * npix, e->cfg.rows, e->cfg.cols, d, d + npix, d + 2 * npix);
  hipError_t r = hipMemcpyAsync(I, d, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipMemcpyAsync(Gx, d + npix, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipMemcpyAsync(Gy, d + 2 * npix, npix * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (r != hipSuccess) return fail(e, PBA_ERR_HIP, "pba_get_frame_channel: %s", hipGetErrorString(r));
  return PBA_OK;
}

int pba_get_frame_channels_f32(pba_engine* e, int slot, float* channels) {
  if (!e || slot < 0 || slot >= e->cfg.max_frames || !channels) return PBA_ERR_INVALID;
  if (e->channels <= 1) return fail(e, PBA_ERR_INVALID, "pba_get_frame_channels_f32: single-channel engine");
  if (!e->frame_set[slot]) return fail(e, PBA_ERR_STATE, "pba_get_frame_channels_f32: slot %d holds no frame", slot);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t n = (size_t)e->channels * e->cfg.rows * e->cfg.cols;
  HIP_TRY(e, hipMemcpyAsync(channels, e->d_frames_mc + (size_t)slot * n, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return PBA_OK;
}

int pba_sample_frame(pba_engine* e, int slot, int32_t channel, int32_t n, const float* y, const float* x, float* out3) {
  if (!e || slot < 0 || slot >= e->cfg.max_frames || n <= 0 || !y || !x || !out3 || channel < 0 || channel >= e->channels) return PBA_ERR_INVALID;
  if (!e->frame_set[slot]) return fail(e, PBA_ERR_STATE, "pba_sample_frame: slot %d holds no frame", slot);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t npix = (size_t)e->cfg.rows * e->cfg.cols;
  float* d = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d), (size_t)5 * n * sizeof(float)));
  hipError_t r = hipMemcpyAsync(d, y, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream);
  if (r == hipSuccess) r = hipMemcpyAsync(d + n, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream);
  if (r == hipSuccess) {
    if (e->channels > 1)
      hipLaunchKernelGGL(k_sample_probe_mc, dim3((n + 255) / 256), dim3(256), 0, e->stream,
                         (const float*)e->d_frames_mc + ((size_t)slot * e->channels + channel) * npix, e->cfg.rows, e->cfg.cols, n,
                         (const float*)d, (const float*)(d + n), d + 2 * (size_t)n);
    else
      hipLaunchKernelGGL(k_sample_probe, dim3((n + 255) / 256), dim3(256), 0, e->stream, (const uint32_t*)(e->d_frames + npix * slot),
                         e->cfg.rows, e->cfg.cols, n, (const float*)d, (const float*)(d + n), d + 2 * (size_t)n);
    r = hipGetLastError();
  }
  if (r == hipSuccess) r = hipMemcpyAsync(out3, d + 2 * (size_t)n, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (r != hipSuccess) return fail(e, PBA_ERR_HIP, "pba_sample_frame: %s", hipGetErrorString(r));
  return PBA_OK;
}

// cv::getGaussianKernel(n, sigma > 0, CV_32F) as host/imgproc.h gaussianKernel restates it
static void gaussian_kernel_f32(int n, double sigma, float* k) {
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2x * x * x);
    sum += k[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

int pba_set_frame_descriptor_u8(pba_engine* e, int slot, const uint8_t* image, int32_t descriptor, float sigma_ct, float sigma_bp) {
  if (!e || !image || slot < 0 || slot >= e->cfg.max_frames) return PBA_ERR_INVALID;
  if (descriptor == PBA_DESCRIPTOR_INTENSITY) return pba_set_frame_u8(e, slot, image);
  const int want = descriptor == PBA_DESCRIPTOR_INTENSITY_AND_GRADIENT ? 3 : descriptor == PBA_DESCRIPTOR_BITPLANES ? 8 : 0;
  if (!want) return fail(e, PBA_ERR_INVALID, "pba_set_frame_descriptor_u8: unknown descriptor %d", descriptor);
  if (e->channels != want)
    return fail(e, PBA_ERR_INVALID, "pba_set_frame_descriptor_u8: descriptor %d has %d channels, the engine was created for %d", descriptor, want, e->channels);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int rows = e->cfg.rows, cols = e->cfg.cols;
  const size_t npix = (size_t)rows * cols;
  int rc;
  if (descriptor == PBA_DESCRIPTOR_BITPLANES)
    for (int k = 0; k < 2; ++k)
      if (!e->d_u8_work[k] && (rc = dev_alloc(e, &e->d_u8_work[k], npix))) return rc;
  // same borrowed-buffer protocol as pba_set_frame_u8: pinned copy, everything else asynchronous behind the return
  if (e->img_stage_busy) { HIP_TRY(e, hipEventSynchronize(e->ev_img_stage)); e->img_stage_busy = false; }
  std::memcpy(e->h_img_stage, image, npix);
  HIP_TRY(e, hipMemcpyAsync(e->d_img_stage, e->h_img_stage, npix, hipMemcpyHostToDevice, e->stream));
  const dim3 grid((cols + 255) / 256, rows), block(256);
  float* planes = e->d_frames_mc + (size_t)slot * want * npix;      // the slot's channel planes: produced in place
  if (descriptor == PBA_DESCRIPTOR_INTENSITY_AND_GRADIENT) {
    hipLaunchKernelGGL(k_channels_intensity_gradient, grid, block, 0, e->stream, (const uint8_t*)e->d_img_stage, planes, rows, cols);
  } else {
    const uint8_t* src = e->d_img_stage;
    if (sigma_ct > 0.0f) {
      float kf[3];
      gaussian_kernel_f32(3, (double)sigma_ct, kf);
      int ki[3];
      for (int i = 0; i < 3; ++i) ki[i] = (int)std::nearbyint((double)kf[i] * 256.0);
      hipLaunchKernelGGL(k_blur3_u8, grid, block, 0, e->stream, src, e->d_u8_work[0], rows, cols, ki[0], ki[1], ki[2]);
      src = e->d_u8_work[0];
    }
    hipLaunchKernelGGL(k_census, grid, block, 0, e->stream, src, e->d_u8_work[1], rows, cols);
    float k5[5] = {0.f, 0.f, 1.f, 0.f, 0.f};
    if (sigma_bp > 0.0f) gaussian_kernel_f32(5, (double)sigma_bp, k5);
    hipLaunchKernelGGL(k_bitplanes, grid, block, 0, e->stream, (const uint8_t*)e->d_u8_work[1], planes, rows, cols,
                       sigma_bp > 0.0f ? 1 : 0, k5[0], k5[1], k5[2]);
  }
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipEventRecord(e->ev_img_stage, e->stream));
  e->img_stage_busy = true;
  e->img_stage_valid = true;
  e->frame_set[slot] = 1;
  return PBA_OK;
}

int pba_set_frame_pyr_down(pba_engine* e, int slot, pba_engine* finer, int finer_slot, uint8_t* image_out) {
  if (!e || !finer || e == finer || slot < 0 || slot >= e->cfg.max_frames || finer_slot < 0 || finer_slot >= finer->cfg.max_frames)
    return PBA_ERR_INVALID;
  if (e->channels > 1 || finer->channels > 1) return fail(e, PBA_ERR_INVALID, "pba_set_frame_pyr_down: single-channel engines only");
  if (e->cfg.device != finer->cfg.device) return fail(e, PBA_ERR_INVALID, "pba_set_frame_pyr_down: the two engines live on devices %d and %d", e->cfg.device, finer->cfg.device);
  const int rows = finer->cfg.rows, cols = finer->cfg.cols, drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
  if (e->cfg.rows != drows || e->cfg.cols != dcols)
    return fail(e, PBA_ERR_INVALID, "pba_set_frame_pyr_down: %dx%d is not the next level of %dx%d (%dx%d)", e->cfg.rows, e->cfg.cols, rows, cols, drows, dcols);
  if (!finer->frame_set[finer_slot]) return fail(e, PBA_ERR_STATE, "pba_set_frame_pyr_down: slot %d of the finer level holds no frame", finer_slot);
  PBA_NOT_POISONED(e);
  PBA_NOT_POISONED(finer);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  // the finer frame is produced on the other engine's stream
  HIP_TRY(e, hipEventRecord(finer->ev_xdep, finer->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, finer->ev_xdep, 0));
  const dim3 grid((dcols + 255) / 256, drows), block(256);
  hipLaunchKernelGGL(k_pyr_down<uint32_t>, grid, block, 0, e->stream, (const uint32_t*)(finer->d_frames + (size_t)rows * cols * finer_slot),
                     e->d_img_stage, rows, cols, drows, dcols);
  // ... and must not be overwritten there before it has been read here
  HIP_TRY(e, hipEventRecord(e->ev_xdep, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(finer->stream, e->ev_xdep, 0));
  hipLaunchKernelGGL(k_pack_frame, grid, block, 0, e->stream, (const uint8_t*)e->d_img_stage, e->d_frames + (size_t)drows * dcols * slot, drows, dcols);
  HIP_TRY(e, hipGetLastError());
  if (image_out) {
    HIP_TRY(e, hipMemcpyAsync(image_out, e->d_img_stage, (size_t)drows * dcols, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  e->img_stage_valid = true;
  e->frame_set[slot] = 1;
  return PBA_OK;
}

// ---- device front-end ---------------------------------------------------------------------------------------------------
static int fe_stage(pba_engine* e, size_t bytes) {      // pinned staging buffer of the front-end calls, grow-only
  if (e->h_fe_cap >= bytes) return PBA_OK;
  if (e->h_fe_io) { (void)hipHostFree(e->h_fe_io); e->h_fe_io = nullptr; e->h_fe_cap = 0; }
  const size_t want = bytes + bytes / 4 + 4096;
  HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_fe_io), want, hipHostMallocDefault));
  e->h_fe_cap = want;
  return PBA_OK;
}

int pba_frontend_visibility(pba_engine* e, int32_t n, const double* uv, const int32_t* rc, const float* patches26, double min_score,
                            int32_t mask_radius, uint8_t* hit) {
  if (!e || n < 0 || mask_radius < 0 || (n > 0 && (!uv || !rc || !patches26 || !hit))) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  if (n > 0 && !e->img_stage_valid)
    return fail(e, PBA_ERR_STATE, "pba_frontend_visibility: no u8 frame behind the ZNCC (upload one with pba_set_frame_u8 / _descriptor_u8 / _pyr_down first)");
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int rows = e->cfg.rows, cols = e->cfg.cols;
  const size_t npix = (size_t)rows * cols;
  int rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_mask, npix))) return rcode;
  HIP_TRY(e, hipMemsetAsync(e->d_fe_mask, 1, npix, e->stream));
  e->fe_mask_valid = true;
  if (n == 0) return PBA_OK;
  for (int i = 0; i < n; ++i)
    if (rc[2 * i] < mask_radius || rc[2 * i] >= rows - mask_radius || rc[2 * i + 1] < mask_radius || rc[2 * i + 1] >= cols - mask_radius)
      return fail(e, PBA_ERR_INVALID, "pba_frontend_visibility: point %d at (row %d, col %d) is closer than the mask radius to the border", i, rc[2 * i], rc[2 * i + 1]);
  // one upload: uv (16 B) | patches (104 B) | rc (8 B) per point; hits come back through the same pinned buffer
  const size_t o_uv = 0, o_pt = o_uv + (size_t)n * 16, o_rc = o_pt + (size_t)n * 104, o_hit = o_rc + (size_t)n * 8, total = o_hit + (size_t)n;
  if ((rcode = fe_stage(e, total))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_io, total + total / 4))) return rcode;
  std::memcpy(e->h_fe_io + o_uv, uv, (size_t)n * 16);
  std::memcpy(e->h_fe_io + o_pt, patches26, (size_t)n * 104);
  std::memcpy(e->h_fe_io + o_rc, rc, (size_t)n * 8);
  HIP_TRY(e, hipMemcpyAsync(e->d_fe_io, e->h_fe_io, o_hit, hipMemcpyHostToDevice, e->stream));
  uint8_t* d_hit = reinterpret_cast<uint8_t*>(e->d_fe_io + o_hit);
  hipLaunchKernelGGL(k_fe_visibility, dim3((n + 127) / 128), dim3(128), 0, e->stream, (const uint8_t*)e->d_img_stage, rows, cols, n,
                     reinterpret_cast<const double*>(e->d_fe_io + o_uv), reinterpret_cast<const float*>(e->d_fe_io + o_pt), min_score, d_hit, (float*)nullptr);
  hipLaunchKernelGGL(k_fe_mask_stamp, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, reinterpret_cast<const int2*>(e->d_fe_io + o_rc),
                     (const uint8_t*)d_hit, mask_radius, e->d_fe_mask, cols);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(e->h_fe_io + o_hit, d_hit, (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  std::memcpy(hit, e->h_fe_io + o_hit, (size_t)n);
  return PBA_OK;
}

int pba_frontend_zncc_probe(pba_engine* e, int32_t n, const double* uv, const float* patches26, float* out27) {
  if (!e || n <= 0 || !uv || !patches26 || !out27) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  if (!e->img_stage_valid)
    return fail(e, PBA_ERR_STATE, "pba_frontend_zncc_probe: no u8 frame behind the ZNCC (upload one with pba_set_frame_u8 / _descriptor_u8 / _pyr_down first)");
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const size_t o_pt = (size_t)n * 16, o_out = o_pt + (size_t)n * 104, total = o_out + (size_t)n * 108;
  int rcode;
  if ((rcode = fe_stage(e, total))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_io, total + total / 4))) return rcode;
  std::memcpy(e->h_fe_io, uv, (size_t)n * 16);
  std::memcpy(e->h_fe_io + o_pt, patches26, (size_t)n * 104);
  HIP_TRY(e, hipMemcpyAsync(e->d_fe_io, e->h_fe_io, o_out, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_fe_visibility, dim3((n + 127) / 128), dim3(128), 0, e->stream, (const uint8_t*)e->d_img_stage, e->cfg.rows, e->cfg.cols, n,
                     reinterpret_cast<const double*>(e->d_fe_io), reinterpret_cast<const float*>(e->d_fe_io + o_pt), 0.0, (uint8_t*)nullptr,
                     reinterpret_cast<float*>(e->d_fe_io + o_out));
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(e->h_fe_io + o_out, e->d_fe_io + o_out, (size_t)n * 108, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  std::memcpy(out27, e->h_fe_io + o_out, (size_t)n * 108);
  return PBA_OK;
}

int pba_frontend_candidates(pba_engine* e, int32_t slot, const float* depth, double min_depth, double max_depth, int32_t nms_radius,
                            int32_t border, int32_t* n_out) {
  if (!e || !depth || !n_out || slot < 0 || slot >= e->cfg.max_frames || border < 0) return PBA_ERR_INVALID;
  if (!e->frame_set[slot]) return fail(e, PBA_ERR_STATE, "pba_frontend_candidates: slot %d holds no frame", slot);
  if (nms_radius > border) return fail(e, PBA_ERR_INVALID, "pba_frontend_candidates: nms radius %d reaches over the border %d", nms_radius, border);
  if (2 * border >= e->cfg.rows || 2 * border >= e->cfg.cols)
    return fail(e, PBA_ERR_INVALID, "pba_frontend_candidates: border %d leaves no interior in a %d x %d image", border, e->cfg.rows, e->cfg.cols);
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int rows = e->cfg.rows, cols = e->cfg.cols;
  const size_t npix = (size_t)rows * cols;
  int rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_mask, npix))) return rcode;
  if (!e->fe_mask_valid) HIP_TRY(e, hipMemsetAsync(e->d_fe_mask, 1, npix, e->stream));      // no visibility call for this frame: nothing is masked
  e->fe_mask_valid = false;       // (consumed: the next frame starts from a fresh mask)
  if ((rcode = dev_alloc(e, &e->d_fe_flag, npix))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_smap, npix))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_depth, npix))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_rows, (size_t)2 * rows + 1))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_cand, npix))) return rcode;
  if ((rcode = fe_stage(e, npix * sizeof(float)))) return rcode;
  std::memcpy(e->h_fe_io, depth, npix * sizeof(float));
  HIP_TRY(e, hipMemcpyAsync(e->d_fe_depth, e->h_fe_io, npix * sizeof(float), hipMemcpyHostToDevice, e->stream));
  const dim3 grid((cols + 255) / 256, rows), block(256);
  const bool mc = e->channels > 1;
  hipLaunchKernelGGL(k_fe_saliency, grid, block, 0, e->stream, (const uint32_t*)(e->d_frames + npix * slot),
                     mc ? (const float*)(e->d_frames_mc + (size_t)slot * e->channels * npix) : (const float*)nullptr, e->channels, rows, cols, e->d_fe_smap);
  CandParams cp{};
  cp.smap = e->d_fe_smap; cp.depth = e->d_fe_depth; cp.mask = e->d_fe_mask; cp.rows = rows; cp.cols = cols; cp.border = border; cp.nms = nms_radius;
  cp.min_depth = min_depth; cp.max_depth = max_depth; cp.flag = e->d_fe_flag; cp.row_count = e->d_fe_rows; cp.row_offset = e->d_fe_rows + rows;
  cp.out = e->d_fe_cand;
  hipLaunchKernelGGL(k_fe_cand_flags, dim3(rows), dim3(256), 0, e->stream, cp);
  hipLaunchKernelGGL(k_fe_scan_rows, dim3(1), dim3(256), 0, e->stream, (const int32_t*)cp.row_count, cp.row_offset, rows);
  hipLaunchKernelGGL(k_fe_cand_write, dim3(rows), dim3(256), 0, e->stream, cp);
  HIP_TRY(e, hipGetLastError());
  int32_t* h_n = reinterpret_cast<int32_t*>(e->h_fe_io);
  HIP_TRY(e, hipMemcpyAsync(h_n, cp.row_offset + rows, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->fe_n_cand = *h_n;
  *n_out = e->fe_n_cand;
  return PBA_OK;
}

int pba_frontend_get_candidates(pba_engine* e, pba_candidate* out, int32_t n) {
  if (!e || n < 0 || (n > 0 && !out)) return PBA_ERR_INVALID;
  if (n > e->fe_n_cand) return fail(e, PBA_ERR_STATE, "pba_frontend_get_candidates: %d requested, the last scan found %d", n, e->fe_n_cand);
  if (n == 0) return PBA_OK;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  { const int rcode = fe_stage(e, (size_t)n * sizeof(pba_candidate)); if (rcode) return rcode; }
  HIP_TRY(e, hipMemcpyAsync(e->h_fe_io, e->d_fe_cand, (size_t)n * sizeof(pba_candidate), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  std::memcpy(out, e->h_fe_io, (size_t)n * sizeof(pba_candidate));
  return PBA_OK;
}

int pba_frontend_descriptors(pba_engine* e, int32_t slot, int32_t n, const int32_t* xy, float* desc) {
  if (!e || n < 0 || slot < 0 || slot >= e->cfg.max_frames || (n > 0 && (!xy || !desc))) return PBA_ERR_INVALID;
  if (!e->frame_set[slot]) return fail(e, PBA_ERR_STATE, "pba_frontend_descriptors: slot %d holds no frame", slot);
  if (n == 0) return PBA_OK;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int rows = e->cfg.rows, cols = e->cfg.cols, R = e->cfg.radius, P = (2 * R + 1) * (2 * R + 1);
  const size_t npix = (size_t)rows * cols;
  const size_t b_xy = (size_t)n * 8, b_out = (size_t)n * e->channels * P * sizeof(float);
  int rcode;
  if ((rcode = fe_stage(e, b_xy + b_out))) return rcode;
  if ((rcode = dev_alloc(e, &e->d_fe_io, b_xy + b_out + (b_xy + b_out) / 4))) return rcode;
  std::memcpy(e->h_fe_io, xy, b_xy);
  HIP_TRY(e, hipMemcpyAsync(e->d_fe_io, e->h_fe_io, b_xy, hipMemcpyHostToDevice, e->stream));
  const bool mc = e->channels > 1;
  const size_t total = (size_t)n * e->channels * P;
  hipLaunchKernelGGL(k_fe_descriptors, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, e->stream,
                     (const uint32_t*)(e->d_frames + npix * slot), mc ? (const float*)(e->d_frames_mc + (size_t)slot * e->channels * npix) : (const float*)nullptr,
                     e->channels, rows, cols, R, n, reinterpret_cast<const int2*>(e->d_fe_io), reinterpret_cast<float*>(e->d_fe_io + b_xy));
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(e->h_fe_io + b_xy, e->d_fe_io + b_xy, b_out, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  std::memcpy(desc, e->h_fe_io + b_xy, b_out);
  return PBA_OK;
}

int pba_set_problem(pba_engine* e, int32_t n_points, const double* xyz, const double* desc, int32_t n_obs,
                    const int32_t* obs_point, const int32_t* obs_slot, const double* weights) {
  if (!e || n_points <= 0 || n_obs <= 0 || !xyz || !desc || !obs_point || !obs_slot || !weights) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int P = (2 * e->cfg.radius + 1) * (2 * e->cfg.radius + 1);     // pixels of one patch (weights)
  const int PD = P * e->channels;                                       // descriptor entries per point
  // validate + CSR + tiles (whole points per tile of <= kTile observations)
  std::vector<int32_t> pt_begin(n_points + 1, 0);
  std::vector<uint8_t> slot8(n_obs);
  for (int o = 0; o < n_obs; ++o) {
    const int p = obs_point[o], s = obs_slot[o];
    if (p < 0 || p >= n_points || s < 0 || s >= e->cfg.max_frames) return fail(e, PBA_ERR_INVALID, "observation %d out of range", o);
    if (o > 0 && p < obs_point[o - 1]) return fail(e, PBA_ERR_INVALID, "observations must be grouped by point");
    if (o > 0 && p == obs_point[o - 1] && s <= obs_slot[o - 1]) return fail(e, PBA_ERR_INVALID, "duplicate / unsorted slot for point %d", p);
    pt_begin[p + 1]++;
    slot8[o] = (uint8_t)s;
  }
  for (int p = 0; p < n_points; ++p) {
    if (pt_begin[p + 1] == 0) return fail(e, PBA_ERR_INVALID, "point %d has no observation", p);
    pt_begin[p + 1] += pt_begin[p];
  }
  std::vector<int32_t> tiles(1, 0);
  {
    int begin = 0;
    for (int p = 0; p < n_points; ++p) {
      if (pt_begin[p + 1] - begin > kTile) { tiles.push_back(pt_begin[p]); begin = pt_begin[p]; }
    }
    tiles.push_back(n_obs);
  }
  e->n_tiles = (int)tiles.size() - 1;
  std::vector<int4> tinfo(e->n_tiles);
  std::vector<int2> lane_rec((size_t)e->n_tiles * kTile, make_int2(0, 0));
  for (int t = 0; t < e->n_tiles; ++t) {
    const int o0 = tiles[t], o1 = tiles[t + 1];
    tinfo[t] = make_int4(o0, o1 - o0, obs_point[o0], obs_point[o1 - 1] - obs_point[o0] + 1);
    for (int o = o0; o < o1; ++o) {
      const int p = obs_point[o];
      lane_rec[(size_t)t * kTile + (o - o0)] = make_int2(p, (int)slot8[o] | ((pt_begin[p] - o0) << 8) | ((pt_begin[p + 1] - pt_begin[p]) << 16));
    }
  }
  e->n_points = n_points;
  e->n_obs = n_obs;
  e->ctr.n_obs = n_obs; e->ctr.n_points = n_points;

  std::vector<float> descf((size_t)n_points * PD);
  for (size_t i = 0; i < descf.size(); ++i) descf[i] = (float)desc[i];
  std::vector<double> w2(P);
  e->unit_weights = true;
  for (int i = 0; i < P; ++i) { w2[i] = weights[i] * weights[i]; if (weights[i] != 1.0) e->unit_weights = false; }

  int rc;
  for (int k = 0; k < 2; ++k) if ((rc = dev_alloc(e, &e->d_xyz[k], (size_t)3 * n_points))) return rc;
  if ((rc = dev_alloc(e, &e->d_desc, descf.size()))) return rc;
  if ((rc = dev_alloc(e, &e->d_w2, (size_t)P))) return rc;
  if ((rc = dev_alloc(e, &e->d_obs_point, (size_t)n_obs))) return rc;
  if ((rc = dev_alloc(e, &e->d_obs_slot, (size_t)n_obs))) return rc;
  if ((rc = dev_alloc(e, &e->d_pt_begin, (size_t)n_points + 1))) return rc;
  if ((rc = dev_alloc(e, &e->d_tile_info, tinfo.size()))) return rc;
  if ((rc = dev_alloc(e, &e->d_lane_rec, lane_rec.size()))) return rc;
  e->rec_stride = ((int64_t)n_obs + 255) / 256 * 256;
  for (int k = 0; k < 2; ++k) if ((rc = dev_alloc(e, &e->d_rec[k], (size_t)6 * e->rec_stride))) return rc;
  if ((rc = dev_alloc(e, &e->d_sp, (size_t)3 * n_points))) return rc;
  if ((rc = dev_alloc(e, &e->d_ptrec, (size_t)12 * n_points))) return rc;
  e->sample_grid = (n_obs + e->sample_waves * 64 - 1) / (e->sample_waves * 64);
  {
    const int tiles_per_block = std::max(1, e->sample_waves * 64 / kTile);
    e->fused_grid = (e->n_tiles + tiles_per_block - 1) / tiles_per_block;
  }
  const int cost_blocks = std::max(e->sample_grid, e->fused_grid);
  for (int k = 0; k < 2; ++k) {
    if ((rc = dev_alloc(e, &e->d_block_cost[k], (size_t)cost_blocks))) return rc;
    if ((rc = dev_alloc(e, &e->d_block_fail[k], (size_t)cost_blocks))) return rc;
    e->cost_blocks[k] = 0;
  }
  e->backsub_grid = (n_points + 255) / 256;
  if ((rc = dev_alloc(e, &e->d_bs_out, (size_t)3 * std::max(e->backsub_grid, e->fused_grid)))) return rc;
  e->schur_grid = std::min(e->n_tiles, 256 * 4);
  // (experiment switch: workgroups of the persistent k_schur launch; the partial buffers are sized for <= 1024)
  if (const char* sv = getenv("PBA_SCHUR_GRID")) { const int g = atoi(sv); if (g >= 1 && g <= 1024) e->schur_grid = std::min(e->n_tiles, g); }

  // the points go to the CURRENT parity: the cameras of an earlier pba_set_cameras live there too, so the two calls may
  // come in either order
  HIP_TRY(e, hipMemcpyAsync(e->d_xyz[e->cur], xyz, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_desc, descf.data(), sizeof(float) * descf.size(), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_w2, w2.data(), sizeof(double) * P, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_obs_point, obs_point, sizeof(int32_t) * n_obs, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_obs_slot, slot8.data(), n_obs, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_pt_begin, pt_begin.data(), sizeof(int32_t) * (n_points + 1), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_tile_info, tinfo.data(), sizeof(int4) * tinfo.size(), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_lane_rec, lane_rec.data(), sizeof(int2) * lane_rec.size(), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->slot_mask = 0;
  for (int o = 0; o < n_obs; ++o) e->slot_mask |= 1u << slot8[o];
  e->have_problem = true;
  e->inverse_depth = false;         // back to the reference's free world points until pba_set_inverse_depth says otherwise
  e->have_lin = false;
  e->lin_valid[0] = e->lin_valid[1] = false;
  return PBA_OK;
}

int pba_set_cameras(pba_engine* e, const double* cams6, int32_t n_frames, int32_t fixed_slot) {
  if (!e || !cams6 || n_frames < 2 || n_frames > e->cfg.max_frames || fixed_slot >= n_frames) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  e->n_frames = n_frames;
  e->fixed_slot = fixed_slot < 0 ? -1 : fixed_slot;
  e->n_free = n_frames - (e->fixed_slot >= 0 ? 1 : 0);
  e->n_pairs = e->n_free * (e->n_free + 1) / 2;
  e->part_stride = 36 * e->n_pairs + 3 * 6 * e->n_free + 3;
  if (e->n_pairs > kTile) return fail(e, PBA_ERR_INVALID, "too many free cameras for the Schur tile (%d pairs)", e->n_pairs);
  {
    // the reduced solve keeps the whole augmented matrix in LDS (dynamic) next to ~21 KB of static LDS: fits the 160 KB of a
    // gfx950 CU at every supported window; a device with less (gfx90a / gfx942: 64 KB) gets a clear error instead of a failed launch
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, e->cfg.device) == hipSuccess && lds_max > 0) {
      const size_t need = solve_blocked_smem_bytes(6 * e->n_free) + 22 * 1024;
      if (need > (size_t)lds_max)
        return fail(e, PBA_ERR_INVALID, "%d free cameras need %zu bytes of LDS for the reduced solve, the device offers %d per workgroup", e->n_free, need, lds_max);
    }
  }
  int rc;
  if ((rc = dev_alloc(e, &e->d_partial, (size_t)(256 * 4) * (((size_t)e->part_stride + 15) / 16 * 16)))) return rc;      // [entry / 16][workgroup][16]
  if ((rc = dev_alloc(e, &e->d_red, (size_t)pba_engine::kChunks * e->part_stride))) return rc;
  if ((rc = dev_alloc(e, &e->d_packed, (size_t)e->part_stride))) return rc;
  if (e->solve_tab_nf != e->n_free) {
    std::vector<uint32_t> tab((size_t)solve_table_words(e->n_free));
    solve_tables(e->n_free, tab.data());
    if ((rc = dev_alloc(e, &e->d_solve_tab, tab.size()))) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_solve_tab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));      // (the host vector goes out of scope)
    e->solve_tab_nf = e->n_free;
  }
  HIP_TRY(e, hipMemcpyAsync(e->d_cams[e->cur], cams6, sizeof(double) * 6 * n_frames, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_cam_geom, dim3(1), dim3(64), 0, e->stream, e->d_cams[e->cur], e->d_geom[e->cur], n_frames, e->fixed_slot);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->have_cams = true;
  e->have_lin = false;
  e->lin_valid[0] = e->lin_valid[1] = false;
  return PBA_OK;
}

static int fetch_state(pba_engine* e, double* cams6, double* xyz) {
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  // through the engine's host-mapped pinned buffer, written by a kernel: the runtime's copy call blocks for 8 ms the
  // first time a process moves a few hundred KB device -> host (seen in the drop-in class, pinned or pageable
  // destination alike); a kernel that stores over PCIe plus a host copy is ~50 us every time
  const size_t nc = (size_t)6 * e->n_frames, nx = (size_t)3 * e->n_points;
  { const int rc = ensure_state_stage(e, nc + nx); if (rc) return rc; }
  if (cams6) k_to_host<<<dim3(1), dim3(256), 0, e->stream>>>(e->d_cams[e->cur], e->h_state_dev, nc);
  if (xyz) k_to_host<<<dim3((unsigned)std::min<size_t>((nx + 1023) / 1024, 256)), dim3(256), 0, e->stream>>>(e->d_xyz[e->cur], e->h_state_dev + nc, nx);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (cams6) std::memcpy(cams6, e->h_state_stage, sizeof(double) * nc);
  if (xyz) std::memcpy(xyz, e->h_state_stage + nc, sizeof(double) * nx);
  return PBA_OK;
}

int pba_get_state(pba_engine* e, double* cams6, double* xyz) {
  if (!e) return PBA_ERR_INVALID;
  if (e->poisoned) return fail(e, PBA_ERR_STATE, "pba_get_state: the engine is unusable after a timed-out step; destroy it");
  if (!e->have_problem || !e->have_cams) return fail(e, PBA_ERR_STATE, "call order violated: pba_get_state before set_problem/set_cameras");
  return fetch_state(e, cams6, xyz);
}

int pba_set_inverse_depth(pba_engine* e, const double* rays6, const double* rho) {
  if (!e || !rays6 || !rho) return PBA_ERR_INVALID;
  if (!e->have_problem) return fail(e, PBA_ERR_STATE, "call order violated: pba_set_inverse_depth before pba_set_problem");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int n = e->n_points;
  std::vector<double> prm((size_t)3 * n, 0.0);
  for (int i = 0; i < n; ++i) {
    if (!(rho[i] > 0.0) || !std::isfinite(rho[i])) return fail(e, PBA_ERR_INVALID, "inverse depth of point %d is not positive", i);
    prm[3 * (size_t)i] = rho[i];
  }
  int rc = dev_alloc(e, &e->d_rays, (size_t)6 * n);
  if (rc) return rc;
  e->h_rays.assign(rays6, rays6 + (size_t)6 * n);
  HIP_TRY(e, hipMemcpyAsync(e->d_rays, rays6, sizeof(double) * 6 * n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_xyz[e->cur], prm.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->inverse_depth = true;
  e->have_lin = false;
  e->lin_valid[0] = e->lin_valid[1] = false;
  return PBA_OK;
}

int pba_get_points_world(pba_engine* e, double* xyz) {
  if (!e || !xyz) return PBA_ERR_INVALID;
  if (!e->have_problem) return fail(e, PBA_ERR_STATE, "call order violated: pba_get_points_world before pba_set_problem");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  { const int rc = fetch_state(e, nullptr, xyz); if (rc) return rc; }
  if (e->inverse_depth) {
    for (int i = 0; i < e->n_points; ++i) {
      const double inv = 1.0 / xyz[3 * (size_t)i];
      const double* r = e->h_rays.data() + 6 * (size_t)i;
      for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = r[k] + r[3 + k] * inv;
    }
  }
  return PBA_OK;
}

int pba_linearize(pba_engine* e, double* cost) {
  if (!e) return PBA_ERR_INVALID;
  { const int rc0 = check_ready(e, "pba_linearize"); if (rc0) return rc0; }
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  if (!e->lin_valid[e->cur]) {
    SampleParams sp = make_sample_params(e, e->cur);
    ev_begin(e, 0);
    launch_sample<true>(e, sp);
    ev_end(e, 0);
    HIP_TRY(e, hipGetLastError());
    e->cost_blocks[e->cur] = e->sample_grid;
    e->lin_valid[e->cur] = true;
    e->jac_passes++;
  }
  e->have_lin = true;
  if (cost) {
    std::vector<double> bc(e->cost_blocks[e->cur]);
    HIP_TRY(e, hipMemcpyAsync(bc.data(), e->d_block_cost[e->cur], sizeof(double) * bc.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    ev_collect(e);
    double c = 0.0;
    for (double v : bc) c += v;
    if (e->comm.multi()) {
      if (e->comm.allreduce_host(&c, 1, 0)) return fail(e, PBA_ERR_COMM, "allreduce(cost) failed: %s", e->comm.err.c_str());
    }
    *cost = c;
  }
  return PBA_OK;
}

int pba_step(pba_engine* e, double radius, int32_t init_scale, const pba_solver_options* o, pba_step_info* out) {
  return pba_internal_step(e, radius, init_scale, o, out, 0);
}

// grad_only != 0: stop after the reduced solve (cost / gradient norms of the linearisation point only).
int pba_internal_step(pba_engine* e, double radius, int32_t init_scale, const pba_solver_options* o, pba_step_info* out,
                      int grad_only) {
  if (!e || !o || !out || !(radius > 0.0)) return PBA_ERR_INVALID;
  e->last_driver = 3;
  if (!e->have_lin) return fail(e, PBA_ERR_STATE, "call order violated: pba_step before pba_linearize");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int cur = e->cur, cand = 1 - e->cur;
  const int n = 6 * e->n_free;
  const bool multi = e->comm.multi();

  SchurParams sc{};
  sc.xyz = e->d_xyz[cur]; sc.rays = e->inverse_depth ? e->d_rays : nullptr; sc.geom = e->d_geom[cur]; sc.rec = e->d_rec[cur]; sc.obs_point = e->d_obs_point;
  sc.obs_slot = e->d_obs_slot; sc.tile_info = e->d_tile_info; sc.lane_rec = e->d_lane_rec; sc.sp = e->d_sp;
  sc.ptrec = e->d_ptrec; sc.partial = e->d_partial; sc.rec_stride = e->rec_stride; sc.n_tiles = e->n_tiles; sc.n_frames = e->n_frames;
  sc.n_free = e->n_free; sc.n_pairs = e->n_pairs; sc.part_stride = e->part_stride; sc.init_scale = init_scale;
  sc.jacobi = o->jacobi_scaling; sc.fx = e->cfg.fx; sc.fy = e->cfg.fy; sc.radius = radius; sc.inv_radius = 1.0 / radius;
  sc.min_diag = o->min_lm_diagonal; sc.max_diag = o->max_lm_diagonal;
  sc.dbg = nullptr; sc.lm = nullptr;
  if (e->dbg_left > 0) {
    if (!e->d_dbg) { (void)hipMalloc(reinterpret_cast<void**>(&e->d_dbg), sizeof(unsigned long long) * 8 * (1024 + 4096)); }
    sc.dbg = e->d_dbg;
  }
  ev_begin(e, 2);
  launch_schur(e, sc);
  ev_end(e, 2);
  if (sc.dbg) {
    std::vector<unsigned long long> h(8 * (size_t)e->schur_grid);
    (void)hipMemcpyAsync(h.data(), e->d_dbg, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, e->stream);
    (void)hipStreamSynchronize(e->stream);
    double avg[8] = {0};
    for (int b = 0; b < e->schur_grid; ++b) for (int k = 0; k < 8; ++k) avg[k] += (double)h[8 * b + k] / e->schur_grid;
    std::fprintf(stderr, "k_schur phase cycles/block (stage, P1, point totals, P + camera record, camera sums, W|Y write, pair blocks): %.0f %.0f %.0f %.0f %.0f %.0f %.0f  tiles/block %.2f\n",
                 avg[0], avg[1], avg[2], avg[3], avg[4], avg[5], avg[6], (double)e->n_tiles / e->schur_grid);
    {
      const int rem = e->n_tiles % e->schur_grid;   // blocks [0, rem) run one tile more than the others
      double d_long = 0, d_short = 0, d_max = 0; int n_long = 0, n_short = 0;
      for (int b = 0; b < e->schur_grid; ++b) {
        const double d = 0.01 * (double)h[8 * b + 7];
        d_max = std::max(d_max, d);
        if (b < rem) { d_long += d; ++n_long; } else { d_short += d; ++n_short; }
      }
      std::fprintf(stderr, "  k_schur tile loop: %d blocks with the extra tile %.2f us, %d others %.2f us, max %.2f us\n", n_long,
                   n_long ? d_long / n_long : 0.0, n_short, n_short ? d_short / n_short : 0.0, d_max);
    }
    --e->dbg_left;
  }
  SolveParams so{};
  so.packed = e->d_packed;
  so.cams = e->d_cams[cur]; so.cams_cand = e->d_cams[cand]; so.delta_c = e->d_delta_c;
  so.sc = e->d_sc; so.S_dbg = (e->cfg.flags & 1) ? e->d_S : nullptr; so.rhs_dbg = e->d_rhs; so.scal = e->d_scal; so.geom = e->d_geom[cur];
  so.n_frames = e->n_frames; so.n_free = e->n_free; so.n_pairs = e->n_pairs; so.stride = e->part_stride;
  so.fixed_slot = e->fixed_slot; so.tab = e->d_solve_tab;
  so.geom_cand = (!grad_only && fused_capable(e)) ? e->d_geom[cand] : nullptr;
  so.init_scale = init_scale; so.jacobi = o->jacobi_scaling; so.radius = radius; so.min_diag = o->min_lm_diagonal;
  so.max_diag = o->max_lm_diagonal;
  so.lm = nullptr;
  so.dbg = (e->dbg_left > 0) ? 1 : 0;
  { const int rcs = launch_reduce_and_solve(e, so, n, cur, cand, nullptr, 0, e->cost_blocks[cur]); if (rcs) return rcs; }
  const unsigned long long seq = ++e->seq;
  unsigned long long* h_seq_dev = reinterpret_cast<unsigned long long*>(e->h_scal_dev + kNumScal);
  bool xchg_packed = false;
  if (!grad_only && fused_capable(e)) {
    // one kernel: back-substitution -> candidate pass (Jacobian pass when speculating) -> step finalisation
    SampleParams sp = make_sample_params(e, cand);
    sp.tile_info = e->d_tile_info; sp.lane_rec = e->d_lane_rec; sp.geom_prev = e->d_geom[cur];
    sp.xyz_prev = e->d_xyz[cur]; sp.rec_prev = e->d_rec[cur]; sp.sp = e->d_sp; sp.ptrec = e->d_ptrec;
    sp.delta_c = e->d_delta_c; sp.block_bs = e->d_bs_out; sp.ticket = e->d_ticket; sp.scal = e->d_scal;
    sp.host_scal = multi ? nullptr : e->h_scal_dev; sp.host_seq = h_seq_dev; sp.seq = seq; sp.n_tiles = e->n_tiles;
    sp.dbg = (e->dbg_left > 0 && e->d_dbg) ? e->d_dbg + 8 * 1024 : nullptr;
    if (multi) {
      int rcx = ensure_xchg(e);
      if (rcx) return rcx;
      sp.xchg = e->comm.peer ? peer_slot(e, 1) : e->d_xchg; sp.xchg_sys = e->comm.peer ? 1 : 0;
      sp.xchg_rank = e->comm.rank; sp.xchg_world = e->comm.world; xchg_packed = true;
    }
    if (e->speculate) {
      ev_begin(e, 0);
      launch_sample<true, true>(e, sp);
      ev_end(e, 0);
      e->jac_passes++;
    } else {
      ev_begin(e, 1);
      launch_sample<false, true>(e, sp);
      ev_end(e, 1);
      e->cost_passes++;
    }
    e->cost_blocks[cand] = e->fused_grid;
    e->lin_valid[cand] = e->speculate;
    if (sp.dbg) {
      const int fl = e->fused_grid;
      std::vector<unsigned long long> h(8 * (size_t)fl + 8);
      (void)hipMemcpyAsync(h.data(), sp.dbg, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, e->stream);
      (void)hipStreamSynchronize(e->stream);
      double avg[8] = {0};
      for (int b = 0; b < fl; ++b) for (int k = 0; k < 6; ++k) avg[k] += (double)(h[8 * b + k] & 0xffffffffffffffull) / fl;
      // per-XCD timeline (the counters of different XCDs are not assumed to be synchronised)
      {
        unsigned long long g0 = ~0ull, g1 = 0;
        for (int b = 0; b < fl; ++b) { g0 = std::min(g0, h[8 * b + 6]); g1 = std::max(g1, h[8 * b + 7]); }
        double mean_dur = 0.0;
        for (int b = 0; b < fl; ++b) mean_dur += 0.01 * (double)(h[8 * b + 7] - h[8 * b + 6]) / fl;
        std::fprintf(stderr, "k_sample(fused) timeline: first start -> last end %.2f us, mean block duration %.2f us\n",
                     0.01 * (double)(g1 - g0), mean_dur);
        int hist[16] = {0};
        for (int b = 0; b < fl; ++b) { int k = (int)((h[8 * b + 6] - g0) / 400); hist[k > 15 ? 15 : k]++; }
        const unsigned long long* hf = &h[8 * (size_t)fl];
        std::fprintf(stderr, "  last workgroup: own work %.2f us, finalisation %.2f us (loads %.2f, reduce %.2f, decide %.2f)\n",
                     0.01 * (double)hf[1], 0.01 * (double)hf[0], 0.01 * (double)hf[2], 0.01 * (double)hf[3], 0.01 * (double)hf[4]);
        std::fprintf(stderr, "  block starts per 4 us bin:");
        for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d", hist[k]);
        std::fprintf(stderr, "\n");
      }
      for (int x = 0; x < 8; ++x) {
        unsigned long long t0 = ~0ull, t1 = 0; int n = 0; int late = 0;
        int mism = 0;
        for (int b = 0; b < fl; ++b) {
          if ((int)(h[8 * b] >> 56) != x) continue;
          t0 = std::min(t0, h[8 * b + 6]); t1 = std::max(t1, h[8 * b + 7]); ++n;
          if ((b & 7) != x) ++mism;
        }
        unsigned long long first_end = ~0ull;
        for (int b = 0; b < fl; ++b) if ((int)(h[8 * b] >> 56) == x) first_end = std::min(first_end, h[8 * b + 7]);
        for (int b = 0; b < fl; ++b) if ((int)(h[8 * b] >> 56) == x && h[8 * b + 6] >= first_end) ++late;
        (void)mism;
        std::fprintf(stderr, "  xcd %d: %d blocks, span %.2f us, first block done after %.2f us, %d blocks started later than that\n",
                     x, n, 0.01 * (double)(t1 - t0), 0.01 * (double)(first_end - t0), late);
      }
      std::fprintf(stderr, "k_sample(fused) phase cycles/block: geom-stage %.0f  backsub %.0f  geometry+base %.0f  staging %.0f  walk %.0f  loss+reduce %.0f\n",
                   avg[0], avg[1], avg[2], avg[3], avg[4], avg[5]);
    }
  } else if (!grad_only) {
    BacksubParams bs{};
    bs.xyz = e->d_xyz[cur]; bs.rays = e->inverse_depth ? e->d_rays : nullptr; bs.xyz_cand = e->d_xyz[cand]; bs.geom = e->d_geom[cur]; bs.rec = e->d_rec[cur];
    bs.pt_begin = e->d_pt_begin; bs.obs_slot = e->d_obs_slot; bs.sp = e->d_sp; bs.ptrec = e->d_ptrec;
    bs.delta_c = e->d_delta_c; bs.block_out = e->d_bs_out; bs.rec_stride = e->rec_stride; bs.n_points = e->n_points;
    bs.fx = e->cfg.fx; bs.fy = e->cfg.fy;
    bs.cams_cand = e->d_cams[cand]; bs.geom_cand = e->d_geom[cand]; bs.n_frames = e->n_frames; bs.fixed_slot = e->fixed_slot;
    hipLaunchKernelGGL(k_backsub, dim3(e->backsub_grid), dim3(256), 0, e->stream, bs);
    // candidate point: Jacobian pass when speculating on acceptance, else cost pass
    SampleParams sp = make_sample_params(e, cand);
    if (e->speculate) {
      ev_begin(e, 0);
      launch_sample<true>(e, sp);
      ev_end(e, 0);
      e->jac_passes++;
    } else {
      ev_begin(e, 1);
      launch_sample<false>(e, sp);
      ev_end(e, 1);
      e->cost_passes++;
    }
    e->cost_blocks[cand] = e->sample_grid;
    e->lin_valid[cand] = e->speculate;
    hipLaunchKernelGGL(k_finalize_step, dim3(1), dim3(256), 0, e->stream, e->d_bs_out, e->backsub_grid, e->d_block_cost[cand],
                       e->d_block_fail[cand], e->sample_grid, e->d_scal, multi ? nullptr : e->h_scal_dev, h_seq_dev, seq);
  }
  HIP_TRY(e, hipGetLastError());
  if (multi) {
    int rc2 = exchange_step_scalars(e, xchg_packed);
    if (rc2) return rc2;
  }
  if (multi || grad_only) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, e->stream, e->d_scal, e->h_scal_dev, h_seq_dev, seq,
                       (const double*)(multi ? e->d_xchg : nullptr), e->comm.world);
    HIP_TRY(e, hipGetLastError());
  }
  // wait for the device to publish this step's scalar block (host-mapped memory, no driver round trip)
  {
    volatile unsigned long long* h_seq = reinterpret_cast<volatile unsigned long long*>(e->h_scal + kNumScal);
    unsigned long spins = 0;
    double t_first = -1.0;
    { const int rcc = comm_failed(e); if (rcc) return rcc; }
    while (*h_seq != seq) {
      { const int rcc = comm_failed(e); if (rcc) return rcc; }
      // hipStreamQuery is not free for the device (it showed up as a ~6 us bubble in front of the next kernel), so it only
      // serves as a watchdog here: roughly every 50 ms of spinning
      if ((++spins & 0x3ffffff) == 0) {
        const hipError_t q = hipStreamQuery(e->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(e, PBA_ERR_HIP, "stream error while waiting: %s", hipGetErrorString(q));
        if (q == hipSuccess && *h_seq != seq) return fail(e, PBA_ERR_HIP, "step finished without publishing its scalars");
        const double t = wall_seconds();
        if (t_first < 0.0) t_first = t;
        else if (t - t_first > e->wait_timeout_s) {
          e->poisoned = true;
          return fail(e, multi ? PBA_ERR_COMM : PBA_ERR_HIP, "timed out after %.0f s waiting for step %llu", e->wait_timeout_s, seq);
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // the sequence number may already have been there when the wait was entered: the error word is checked on the way out too
    { const int rcc = comm_failed(e); if (rcc) return rcc; }
  }
  if (e->profile) { HIP_TRY(e, hipStreamSynchronize(e->stream)); ev_collect(e); }
  double s[kNumScal];
  for (int k = 0; k < kNumScal; ++k) s[k] = reinterpret_cast<volatile double*>(e->h_scal)[k];
  out->cost = s[kCostLin];
  out->gradient_max_norm = std::max(s[kGmaxPts], s[kGmaxCams]);
  out->gradient_norm = std::sqrt(s[kGnorm2Pts] + s[kGnorm2Cams]);
  out->model_cost_change = s[kMccPts] + s[kMccCams];
  out->step_norm = std::sqrt(s[kStep2Pts] + s[kStep2Cams]);
  out->x_norm = std::sqrt(s[kX2Pts] + s[kX2Cams]);
  out->candidate_cost = s[kCandCost];
  out->linear_solver_ok = (s[kSolveOk] > 0.5 && s[kSchurFail] < 0.5) ? 1 : 0;
  out->eval_ok = (s[kEvalFailCand] < 0.5 && std::isfinite(s[kCandCost])) ? 1 : 0;
  if (s[kEvalFailLin] > 0.5) return fail(e, PBA_ERR_NUMERIC, "non-finite residual block at the linearisation point");
  return PBA_OK;
}

int pba_accept(pba_engine* e) {
  if (!e) return PBA_ERR_INVALID;
  if (!e->have_lin) return fail(e, PBA_ERR_STATE, "call order violated: pba_accept before pba_step");
  e->lin_valid[e->cur] = false;
  e->cur = 1 - e->cur;
  e->have_lin = e->lin_valid[e->cur];
  return PBA_OK;
}

int pba_get_reduced_system(pba_engine* e, double* S, double* rhs, int32_t* n_out) {
  if (!e || !n_out) return PBA_ERR_INVALID;
  if (!(e->cfg.flags & 1)) return fail(e, PBA_ERR_STATE, "call order violated: pba_get_reduced_system needs pba_config.flags bit 0 (keep reduced system)");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int n = 6 * e->n_free;
  *n_out = n;
  if (S) HIP_TRY(e, hipMemcpyAsync(S, e->d_S, sizeof(double) * n * n, hipMemcpyDeviceToHost, e->stream));
  if (rhs) HIP_TRY(e, hipMemcpyAsync(rhs, e->d_rhs, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return PBA_OK;
}

int pba_get_obs_records(pba_engine* e, double* rec6) {
  if (!e || !rec6) return PBA_ERR_INVALID;
  if (!e->have_problem) return fail(e, PBA_ERR_STATE, "no problem");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  std::vector<double> tmp((size_t)6 * e->rec_stride);
  HIP_TRY(e, hipMemcpyAsync(tmp.data(), e->d_rec[e->cur], sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  for (int o = 0; o < e->n_obs; ++o)
    for (int k = 0; k < 6; ++k) rec6[(size_t)o * 6 + k] = tmp[(size_t)k * e->rec_stride + o];
  return PBA_OK;
}

int pba_comm_unique_id(void* id128) { return Comm::unique_id(id128) ? PBA_ERR_COMM : PBA_OK; }

int pba_comm_init_rccl(pba_engine* e, const void* id128, int32_t rank, int32_t world) {
  if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  if (e->comm.init_rccl(id128, rank, world)) return fail(e, PBA_ERR_COMM, "%s", e->comm.err.c_str());
  return PBA_OK;
}

int pba_comm_init_callback(pba_engine* e, pba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world) {
  if (!e || !fn || world < 1 || rank < 0 || rank >= world) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  if (e->comm.init_callback(fn, ctx, rank, world)) return fail(e, PBA_ERR_COMM, "%s", e->comm.err.c_str());
  return PBA_OK;
}

int pba_comm_enable_peer_exchange(pba_engine* e) {
  if (!e) return PBA_ERR_INVALID;
  if (e->comm.kind == 0) return fail(e, PBA_ERR_STATE, "call order violated: pba_comm_enable_peer_exchange before pba_comm_init_*");
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->comm.enable_peer()) e->err = e->comm.err;      // not an error: the base transport keeps serving (pba_comm_transport tells)
  return PBA_OK;
}

int pba_comm_rank_count(const pba_engine* e) { return e ? e->comm.rank_count() : 0; }

const char* pba_comm_transport(const pba_engine* e) {
  if (!e || e->comm.kind == 0) return "none";
  if (e->comm.kind == 1) return e->comm.peer ? "rccl+peer" : "rccl";
  return e->comm.peer ? "callback+peer" : "callback";
}

const char* pba_solve_driver(const pba_engine* e) {
  static const char* names[] = {"none", "resident", "pipelined", "host-stepped"};
  return names[(e && e->last_driver >= 0 && e->last_driver <= 3) ? e->last_driver : 0];
}

int pba_get_counters(pba_engine* e, pba_counters* c) {
  if (!e || !c) return PBA_ERR_INVALID;
  if (e->stamps && e->d_stamp && e->last_driver == 1) {
    // resident solve: phase stamps of the serial workgroup.  The three counters keep their meaning as SHARES OF THE ITERATION: elimination
    // (k_schur's work) | reduction of the partials + reduced solve (k_reduce_solve's) | back-substitution + sampling + decision (k_sample's)
    PBA_NOT_POISONED(e);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int n_rec = std::min(std::max(e->stamp_iter, 0) + 2, (int)kStampMaxIters);
    std::vector<unsigned long long> st((size_t)n_rec * kResStampRecord);
    HIP_TRY(e, hipMemcpyAsync(st.data(), e->d_stamp, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    double d_sample = 0.0, d_schur = 0.0, d_solve = 0.0;
    int64_t n = 0;
    for (int it = 2; it < n_rec; ++it) {
      const unsigned long long* r = &st[(size_t)it * kResStampRecord];
      const unsigned long long prev = st[(size_t)(it - 1) * kResStampRecord + kResStampDecided];
      if (!prev || r[kResStampSchur] <= prev || r[kResStampSolved] <= r[kResStampSchur] || r[kResStampDecided] <= r[kResStampSolved]) continue;   // a step that did not run
      d_schur += (double)(r[kResStampSchur] - prev); d_solve += (double)(r[kResStampSolved] - r[kResStampSchur]);
      d_sample += (double)(r[kResStampDecided] - r[kResStampSolved]);
      ++n;
    }
    e->ctr.linearize_ms = 1e-5 * d_sample; e->ctr.linearize_launches = n;
    e->ctr.schur_ms = 1e-5 * d_schur; e->ctr.schur_launches = n;
    e->ctr.solve_ms = 1e-5 * d_solve; e->ctr.solve_launches = n;
  } else if (e->stamps && e->d_stamp) {
    // device time stamps of the LAST asynchronous solve (100 MHz ticks): intervals between consecutive kernel ends
    PBA_NOT_POISONED(e);
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const int n_rec = std::min(e->stamp_iter, (int)kStampMaxIters) + 1;
    std::vector<unsigned long long> st((size_t)n_rec * kStampRecord);
    HIP_TRY(e, hipMemcpyAsync(st.data(), e->d_stamp, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    double d_sample = 0.0, d_schur = 0.0, d_solve = 0.0;
    int64_t n = 0;
    for (int it = 1; it < n_rec; ++it) {
      const unsigned long long* r = &st[(size_t)it * kStampRecord];
      const unsigned long long prev = st[(size_t)(it - 1) * kStampRecord + kStampEndSample];
      unsigned long long e_schur = 0;
      for (int b = 0; b < kStampSchurBlocks; ++b) e_schur = std::max(e_schur, r[kStampSchur0 + b]);
      if (!prev || e_schur <= prev || r[kStampEndSolve] <= e_schur || r[kStampEndSample] <= r[kStampEndSolve]) continue;   // a step that did not run
      d_schur += (double)(e_schur - prev); d_solve += (double)(r[kStampEndSolve] - e_schur); d_sample += (double)(r[kStampEndSample] - r[kStampEndSolve]);
      ++n;
    }
    e->ctr.linearize_ms = 1e-5 * d_sample; e->ctr.linearize_launches = n;
    e->ctr.schur_ms = 1e-5 * d_schur; e->ctr.schur_launches = n;
    e->ctr.solve_ms = 1e-5 * d_solve; e->ctr.solve_launches = n;
  }
  *c = e->ctr;
  return PBA_OK;
}

int pba_reset_counters(pba_engine* e) {
  if (!e) return PBA_ERR_INVALID;
  const int64_t no = e->ctr.n_obs, np = e->ctr.n_points;
  e->ctr = pba_counters{};
  e->ctr.n_obs = no; e->ctr.n_points = np;
  e->profile = true;   // counters are only collected once asked for
  e->stamps = false;
  return PBA_OK;
}

int pba_set_profiling(pba_engine* e, int32_t mode) {
  if (!e || mode < 0 || mode > 2) return PBA_ERR_INVALID;
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  const int64_t no = e->ctr.n_obs, np = e->ctr.n_points;
  e->ctr = pba_counters{};
  e->ctr.n_obs = no; e->ctr.n_points = np;
  e->profile = (mode == 1);
  e->stamps = (mode == 2);
  e->stamp_iter = 0;
  if (e->stamps) {
    const int rc = dev_alloc(e, &e->d_stamp, (size_t)(kStampMaxIters + 1) * kStampRecord);
    if (rc) return rc;
    HIP_TRY(e, hipMemsetAsync(e->d_stamp, 0, sizeof(unsigned long long) * (kStampMaxIters + 1) * kStampRecord, e->stream));
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return PBA_OK;
}

}  // extern "C"

// resident solve (pba_resident.h): kernel and LDS pool of the engine's patch radius / weights
namespace {
template <int R, bool UNITW>
const void* resident_kernel() { return reinterpret_cast<const void*>(&k_resident<R, UNITW>); }
const void* resident_kernel_for(const pba_engine* e) {
  if (e->cfg.radius == 1) return e->unit_weights ? resident_kernel<1, true>() : resident_kernel<1, false>();
  return e->unit_weights ? resident_kernel<2, true>() : resident_kernel<2, false>();
}
size_t resident_pool_for(const pba_engine* e) {
  const int n = 6 * e->n_free;
  return e->cfg.radius == 1 ? resident_pool_bytes<1>(n) : resident_pool_bytes<2>(n);
}
}  // namespace

// engine internals needed by the LM driver (pba_lm.cpp)
extern "C" {
int pba_internal_world(const pba_engine* e) { return e->comm.world; }
int pba_internal_rank(const pba_engine* e) { return e->comm.rank; }
int pba_internal_is_multi(const pba_engine* e) { return e->comm.multi() ? 1 : 0; }
int64_t pba_internal_local_blocks(const pba_engine* e) { return e->n_obs; }
int pba_internal_patch_len(const pba_engine* e) { return e->channels * (2 * e->cfg.radius + 1) * (2 * e->cfg.radius + 1); }
// ---- asynchronous driver ----------------------------------------------------------------------------------------
int pba_internal_async_capable(const pba_engine* e, const pba_solver_options* o) {
  return e->use_async && fused_capable(e) && (e->comm.kind != 2 || e->comm.peer) && o->max_num_iterations < pba_engine::kMaxLog - 2 && !e->profile;
}

int pba_internal_ready(pba_engine* e) { return check_ready(e, "pba_solve"); }

// 1: the final (gradient-only) enqueue, kind 2, also flushes to the host mirror -- no kind-3 enqueue behind it (single rank, the
// reduction + solve as one launch)
int pba_internal_final_flushes(const pba_engine* e) {
  static const bool off = [] { const char* sv = getenv("PBA_FUSE_FINAL"); return sv && atoi(sv) == 0; }();
  return (!off && !e->comm.multi() && e->solve_kind == 0 && !PBA_PHASE_TIMING) ? 1 : 0;
}

int pba_internal_async_begin(pba_engine* e, const pba_solver_options* o) {
  { const int rc0 = check_ready(e, "pba_solve"); if (rc0) return rc0; }
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  LmState st;
  std::memset(&st, 0, sizeof(st));
  st.radius = o->initial_trust_region_radius; st.decrease_factor = 2.0;
  st.cur = e->cur; st.pending_grad = -1; st.first = 1;
  st.function_tolerance = o->function_tolerance; st.gradient_tolerance = o->gradient_tolerance;
  st.parameter_tolerance = o->parameter_tolerance; st.max_radius = o->max_trust_region_radius;
  st.min_radius = o->min_trust_region_radius; st.min_relative_decrease = o->min_relative_decrease;
  st.max_num_iterations = o->max_num_iterations; st.max_invalid = o->max_num_consecutive_invalid_steps;
  *e->h_lm = st;
  // The device copy of the initial state is made by the first kernel of the solve (workgroup 0 of the kind-0 sampling pass reads the
  // host-mapped mirror: SampleParams::lm_init_*); PBA_LM_INIT_KERNEL=1 keeps the separate launch of rounds 3-4.
  if (PBA_LM_INIT_KERNEL) {
    hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, e->stream, e->d_lm, st);
    HIP_TRY(e, hipGetLastError());
  }
  e->async_cur = e->cur;
  e->last_driver = 2;
  return PBA_OK;
}

// kind 0: first linearisation; 1: full iteration; 2: gradient norms of the final point; 3: flush (log + state to the
// host).  Returns the sequence number that marks the enqueued work as complete (0 for kind 0).  Single rank: a kind-1
// sequence number is published by the NEXT kind-1 / kind-3 enqueue, so callers end every solve with a flush.
int pba_internal_async_enqueue(pba_engine* e, int kind, int init_scale, const pba_solver_options* o, unsigned long long* seq_out) {
  const int cur = e->async_cur, cand = 1 - cur;
  const int n = 6 * e->n_free;
  const bool multi = e->comm.multi();
  unsigned long long* h_seq_dev = reinterpret_cast<unsigned long long*>(e->h_scal_dev + kNumScal);
  *seq_out = 0;
  auto sample_params = [&](bool skip) {
    SampleParams sp = make_sample_params(e, skip ? cur : cand);
    sp.tile_info = e->d_tile_info; sp.lane_rec = e->d_lane_rec;
    sp.geom_prev = e->d_geom[skip ? cand : cur]; sp.xyz_prev = e->d_xyz[skip ? cand : cur]; sp.rec_prev = e->d_rec[skip ? cand : cur];
    sp.sp = e->d_sp; sp.ptrec = e->d_ptrec; sp.delta_c = e->d_delta_c; sp.block_bs = e->d_bs_out; sp.ticket = e->d_ticket;
    sp.scal = e->d_scal; sp.n_tiles = e->n_tiles; sp.skip_backsub = skip ? 1 : 0;
    sp.block_cost_alt = e->d_block_cost[skip ? cand : cur]; sp.block_fail_alt = e->d_block_fail[skip ? cand : cur];
    sp.host_state = e->h_lm_dev; sp.log = e->d_log; sp.max_log = pba_engine::kMaxLog;
    return sp;
  };
  if (kind == 0) {
    // plain Jacobian pass at the current point, on the fused (tile) grid so that both parities share one block count
    SampleParams sp = sample_params(true);
    sp.lm = nullptr; sp.host_scal = nullptr; sp.host_seq = h_seq_dev; sp.seq = 0; sp.decide = 0; sp.enq_cur = cur;
    if (!PBA_LM_INIT_KERNEL) { sp.lm_init_dst = e->d_lm; sp.lm_init_src = e->h_lm_dev; }
    e->stamp_iter = 0;
    sp.stamp = stamp_record(e);
    launch_sample<true, true>(e, sp);
    e->jac_passes++;
    e->cost_blocks[0] = e->cost_blocks[1] = e->fused_grid;
    HIP_TRY(e, hipGetLastError());
    return PBA_OK;
  }
  if (kind == 3) {
    // end of the solve: iteration log, final state and sequence number to the host mirror
    const unsigned long long seq = ++e->seq;
    hipLaunchKernelGGL(k_flush, dim3(1), dim3(256), 0, e->stream, (const LmState*)e->d_lm, e->h_lm_dev, (const double*)e->d_scal,
                       e->h_scal_dev, (const pba_iteration_summary*)e->d_log, e->h_log_dev, (int)pba_engine::kMaxLog, h_seq_dev, seq);
    HIP_TRY(e, hipGetLastError());
    *seq_out = seq;
    return PBA_OK;
  }
  SchurParams sc{};
  sc.xyz = e->d_xyz[cur]; sc.rays = e->inverse_depth ? e->d_rays : nullptr; sc.geom = e->d_geom[cur]; sc.rec = e->d_rec[cur]; sc.obs_point = e->d_obs_point;
  sc.obs_slot = e->d_obs_slot; sc.tile_info = e->d_tile_info; sc.lane_rec = e->d_lane_rec; sc.sp = e->d_sp;
  sc.ptrec = e->d_ptrec; sc.partial = e->d_partial; sc.rec_stride = e->rec_stride; sc.n_tiles = e->n_tiles; sc.n_frames = e->n_frames;
  sc.n_free = e->n_free; sc.n_pairs = e->n_pairs; sc.part_stride = e->part_stride; sc.init_scale = init_scale;
  // the previous enqueue decided on the device (last workgroup of the fused sampling kernel, or k_decide after the
  // multi-rank exchange) without publishing; this kernel does
  sc.pub_state = e->h_lm_dev; sc.pub_scal = e->d_scal; sc.pub_host_scal = e->h_scal_dev; sc.pub_host_seq = h_seq_dev; sc.pub_seq = e->seq;
  sc.jacobi = o->jacobi_scaling; sc.fx = e->cfg.fx; sc.fy = e->cfg.fy; sc.radius = 1.0; sc.inv_radius = 1.0;
  sc.min_diag = o->min_lm_diagonal; sc.max_diag = o->max_lm_diagonal; sc.dbg = nullptr;
  if (kind == 1 && e->stamps) e->stamp_iter++;
  sc.stamp = (kind == 1) ? stamp_record(e) : nullptr;
  sc.lm = e->d_lm; sc.enq_cur = cur; sc.final_pass = (kind == 2) ? 1 : 0; sc.xyz_alt = e->d_xyz[cand]; sc.geom_alt = e->d_geom[cand]; sc.rec_alt = e->d_rec[cand];
  launch_schur(e, sc);
  SolveParams so{};
  so.packed = e->d_packed; so.cams = e->d_cams[cur]; so.cams_cand = e->d_cams[cand]; so.delta_c = e->d_delta_c;
  so.sc = e->d_sc; so.S_dbg = (e->cfg.flags & 1) ? e->d_S : nullptr; so.rhs_dbg = e->d_rhs; so.scal = e->d_scal; so.geom = e->d_geom[cur];
  so.n_frames = e->n_frames; so.n_free = e->n_free; so.n_pairs = e->n_pairs; so.stride = e->part_stride; so.fixed_slot = e->fixed_slot; so.tab = e->d_solve_tab;
  so.geom_cand = (kind == 1) ? e->d_geom[cand] : nullptr;
  so.init_scale = init_scale; so.jacobi = o->jacobi_scaling; so.radius = 1.0; so.min_diag = o->min_lm_diagonal; so.max_diag = o->max_lm_diagonal;
  so.lm = e->d_lm; so.enq_cur = cur; so.final_pass = (kind == 2) ? 1 : 0; so.cams_alt = e->d_cams[cand]; so.cams_cand_alt = e->d_cams[cur]; so.geom_alt = e->d_geom[cand];
  so.geom_cand_alt = e->d_geom[cur];
  // single rank: the final pass decides and flushes in its own last workgroup (no k_decide, no k_flush behind it)
  const bool fin_fused = kind == 2 && pba_internal_final_flushes(e);
  const unsigned long long seq = ++e->seq;
  ReduceSolveParams::Fin fin{};
  if (fin_fused) {
    fin.lm = e->d_lm; fin.log = e->d_log; fin.host_log = e->h_log_dev; fin.max_log = (int)pba_engine::kMaxLog; fin.host_state = e->h_lm_dev;
    fin.host_scal = e->h_scal_dev; fin.host_seq = h_seq_dev; fin.seq = seq;
  }
  { const int rcs = launch_reduce_and_solve(e, so, n, cur, cand, e->d_lm, kind == 2 ? 1 : 0, e->fused_grid, fin_fused ? &fin : nullptr); if (rcs) return rcs; }
  if (fin_fused) { HIP_TRY(e, hipGetLastError()); *seq_out = seq; return PBA_OK; }
  unsigned long long fused_x = 0;      // exchange number of the step scalars when k_decide does the exchange itself
  if (kind == 1) {
    SampleParams sp = sample_params(false);
    sp.lm = e->d_lm; sp.enq_cur = cur; sp.decide = multi ? 0 : 1;
    sp.stamp = stamp_record(e);
    sp.host_scal = nullptr;     // published by the next k_schur (or k_flush): see SchurParams::pub_*
    sp.host_seq = h_seq_dev; sp.seq = seq;
    if (multi) {
      int rcx = ensure_xchg(e);
      if (rcx) return rcx;
      sp.xchg = e->comm.peer ? peer_slot(e, 1) : e->d_xchg; sp.xchg_sys = e->comm.peer ? 1 : 0;
      sp.xchg_rank = e->comm.rank; sp.xchg_world = e->comm.world;
      if (e->comm.peer) {
        // the exchange of the step scalars has no kernel of its own: the sampling kernel's last workgroup raises the flag,
        // k_decide waits for every rank's and sums the slots
        fused_x = ++e->comm.seq_b;
        sp.xchg = e->comm.mb_own + Comm::data_offset(1, fused_x);
        sp.xchg_flag = reinterpret_cast<unsigned long long*>(e->comm.mb_own) + Comm::flag_index(1, fused_x);
        sp.xchg_seq = fused_x;
      }
    }
    launch_sample<true, true>(e, sp);
    e->jac_passes++;
    HIP_TRY(e, hipGetLastError());
  }
  if (multi && !fused_x) {
    int rc2 = exchange_step_scalars(e, kind == 1);
    if (rc2) return rc2;
  }
  if (multi || kind == 2) {
    DecideParams dp{};
    if (fused_x) {
      dp.peer = peer_params(e); dp.peer_world = e->comm.world; dp.peer_flag = (int)Comm::flag_index(1, fused_x);
      dp.peer_off = (unsigned long long)Comm::data_offset(1, fused_x); dp.peer_seq = fused_x;
      dp.peer_timeout = (unsigned long long)(0.5 * e->wait_timeout_s * e->tick_hz); dp.peer_err = e->h_comm_err_dev;
    }
    dp.lm = e->d_lm; dp.host_state = e->h_lm_dev; dp.scal = e->d_scal; dp.host_scal = e->h_scal_dev; dp.log = e->d_log;
    dp.xchg = (multi && !fused_x) ? e->d_xchg : nullptr; dp.world = e->comm.world;
    dp.max_log = pba_engine::kMaxLog; dp.grad_only = (kind == 2) ? 1 : 0; dp.seq = seq;
    dp.host_seq = (kind == 1) ? nullptr : h_seq_dev;      // a full step is published by the next k_schur / k_flush
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, e->stream, dp);
  }
  HIP_TRY(e, hipGetLastError());
  *seq_out = seq;
  return PBA_OK;
}

int pba_internal_async_wait(pba_engine* e, unsigned long long seq) {
  volatile unsigned long long* h_seq = reinterpret_cast<volatile unsigned long long*>(e->h_scal + kNumScal);
  unsigned long spins = 0;
  double t_first = -1.0;
  { const int rcc = comm_failed(e); if (rcc) return rcc; }
  while (*h_seq < seq) {
    { const int rcc = comm_failed(e); if (rcc) return rcc; }
    // hipStreamQuery is not free for the device (it showed up as a ~6 us bubble in front of the next kernel), so it only
    // serves as a watchdog here: roughly every 50 ms of spinning
    if ((++spins & 0x3ffffff) == 0) {
      const hipError_t q = hipStreamQuery(e->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(e, PBA_ERR_HIP, "stream error while waiting: %s", hipGetErrorString(q));
      if (q == hipSuccess && *h_seq < seq) {
        // a terminated solve turns the remaining kernels into no-ops that publish nothing
        if (reinterpret_cast<volatile LmState*>(e->h_lm)->done) return PBA_OK;
        return fail(e, PBA_ERR_HIP, "step finished without publishing");
      }
      // a collective that never completes (a peer died, or the ranks disagree on the number of enqueued steps) would
      // otherwise spin forever: bounded by PBA_WAIT_TIMEOUT_S (default 120 s) of wall-clock per awaited step
      const double t = wall_seconds();
      if (t_first < 0.0) t_first = t;
      else if (t - t_first > e->wait_timeout_s) {
        e->poisoned = true;
        return fail(e, e->comm.multi() ? PBA_ERR_COMM : PBA_ERR_HIP, "timed out after %.0f s waiting for step %llu (last published %llu)",
                    e->wait_timeout_s, seq, (unsigned long long)*h_seq);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return comm_failed(e);      // (normal with the pipelined driver: the awaited number was already visible on entry)
}

// ---- resident solve ------------------------------------------------------------------------------------------------

// A resident launch that left without publishing: one of its device-side waits timed out (a lost hand-over) and raised the abort word, which
// stays raised -- like every timed-out wait, this poisons the engine (later calls fail fast, pba_destroy does not wait for the stream).
void pba_internal_resident_failed(pba_engine* e) {
  e->poisoned = true;
  e->err += " -- the resident solve ended without publishing (a device-side wait timed out); the engine is unusable, destroy it";
}

void pba_internal_resident_done(pba_engine* e, int iterations) {
  e->res_epoch = e->res_epoch_launch + (unsigned)std::max(0, iterations) + 8u;
}

// PBA_RES_TRACE: phase intervals of the serial workgroup (100 MHz stamps), averaged over the steps of the last resident solve
void pba_internal_resident_trace(pba_engine* e, int iterations) {
  static const bool res_trace = getenv("PBA_RES_TRACE") != nullptr;
  if (!res_trace || !e->d_stamp || iterations < 1 || iterations + 2 > kStampMaxIters) return;
  std::vector<unsigned long long> st((size_t)kResStampRecord * (iterations + 2));
  if (hipMemcpyAsync(st.data(), e->d_stamp, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost, e->stream) != hipSuccess) return;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return;
  double d[5] = {0, 0, 0, 0, 0}, x[5] = {0, 0, 0, 0, 0};
  int n = 0;
  for (int r = 1; r <= iterations; ++r) {
    const unsigned long long* c = &st[(size_t)r * kResStampRecord];
    const unsigned long long prev = (r == 1) ? 0ull : st[(size_t)(r - 1) * kResStampRecord + kResStampDecided];
    if (!c[kResStampDecided] || !c[kResStampSchur]) continue;
    if (prev) d[0] += (double)(c[kResStampSchur] - prev);
    d[1] += (double)(c[kResStampReduced] - c[kResStampSchur]); d[2] += (double)(c[kResStampSolved] - c[kResStampReduced]);
    d[3] += (double)(c[kResStampSampled] - c[kResStampSolved]); d[4] += (double)(c[kResStampDecided] - c[kResStampSampled]);
    if (prev) x[0] += (double)(c[kResStampSchurBegin] - prev);
    x[1] += (double)(c[kResStampSchurBody] - c[kResStampSchurBegin]); x[2] += (double)(c[kResStampSchur] - c[kResStampSchurBody]);
    x[3] += (double)(c[kResStampGathered] - c[kResStampSampled]); x[4] += (double)(c[kResStampSampled] - c[kResStampSampleBegin]);
    ++n;
  }
  if (n > 1)
    std::fprintf(stderr, "resident solve, serial workgroup, us per step over %d steps: elimination %.2f | wait partials + reduce + wait reducers %.2f | solve %.2f | "
                         "back-substitution + sampling %.2f | wait cost partials + sum + decide %.2f | first linearisation + launch %.2f\n",
                 n, 0.01 * d[0] / (n - 1), 0.01 * d[1] / n, 0.01 * d[2] / n, 0.01 * d[3] / n, 0.01 * d[4] / n,
                 0.01 * (double)(st[kResStampRecord + kResStampSchur] - st[kResStampStart]));
  if (atoi(getenv("PBA_RES_TRACE")) >= 2 && e->d_dbg) {
    std::vector<unsigned long long> h(16 * (size_t)e->n_tiles);
    (void)hipMemcpyAsync(h.data(), e->d_dbg, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, e->stream);
    (void)hipStreamSynchronize(e->stream);
    double avg[12] = {0};
    for (int b = 0; b < e->n_tiles; ++b) for (int k = 0; k < 12; ++k) avg[k] += (double)h[16 * b + k] / e->n_tiles;
    std::fprintf(stderr, "   schur_body epilogue cycles: combine groups %.0f, camera sums out %.0f, pair blocks out %.0f, statistics %.0f\n", avg[8], avg[9], avg[10], avg[11]);
    std::fprintf(stderr, "   schur_body phase cycles per tile (100 MHz x ?: s_memtime): stage %.0f, P1 %.0f, point totals %.0f, P + camera record %.0f, camera sums %.0f, factors %.0f, pair blocks %.0f; tile loop %.2f us\n",
                 avg[0], avg[1], avg[2], avg[3], avg[4], avg[5], avg[6], 0.01 * avg[7]);
  }
  if (n > 1)
    std::fprintf(stderr, "   detail: decision -> elimination starts %.2f, schur_body %.2f, partial stores drained + flag %.2f | cost partials gathered %.2f | "
                         "sample_wg alone %.2f\n", 0.01 * x[0] / (n - 1), 0.01 * x[1] / n, 0.01 * x[2] / n, 0.01 * x[3] / n, 0.01 * x[4] / n);
}

// The window fits ONE resident round of 256-thread workgroups (two whole-point tiles each) and the solve is one the resident kernel
// covers: single rank, single channel, reference-exact sampler, free world points, patch radius <= 2, <= 8 free cameras.
int pba_internal_resident_capable(pba_engine* e, const pba_solver_options* o) {
  if (!e->use_resident || !e->use_async || !e->coop_launch || e->comm.multi() || e->channels != 1 || !fused_capable(e) || e->inverse_depth) return 0;
  if (e->cfg.radius > 2 || e->n_free < 1 || e->n_free > kSolveNarrowFree || (e->cfg.flags & 1) || e->solve_kind != 0 || e->profile || PBA_PHASE_TIMING) return 0;
  if (o->max_num_iterations >= pba_engine::kMaxLog - 2) return 0;
  const int groups = (e->n_tiles + 1) / 2;
  if (groups > kResMaxGroups || (e->part_stride + kReduceEntries - 1) / kReduceEntries + 1 > kResMaxReduce) return 0;
  int& gmax = e->res_groups_max[e->cfg.radius - 1][e->unit_weights ? 1 : 0];
  int& gn = e->res_groups_n[e->cfg.radius - 1][e->unit_weights ? 1 : 0];
  if (gmax < 0 || gn != e->n_free) {
    gmax = 0; gn = e->n_free;
    if (hipSetDevice(e->cfg.device) != hipSuccess) return 0;
    const void* fn = resident_kernel_for(e);
    const size_t pool = resident_pool_for(e);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kResThreads, pool) == hipSuccess && per_cu > 0) gmax = per_cu * e->n_cus;
    (void)hipGetLastError();
  }
  return groups <= gmax ? 1 : 0;
}

int pba_internal_resident_launch(pba_engine* e, const pba_solver_options* o, unsigned long long* seq_out) {
  { const int rc0 = check_ready(e, "pba_solve"); if (rc0) return rc0; }
  PBA_NOT_POISONED(e);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  LmState st;
  std::memset(&st, 0, sizeof(st));
  st.radius = o->initial_trust_region_radius; st.decrease_factor = 2.0;
  st.cur = e->cur; st.pending_grad = -1; st.first = 1;
  st.function_tolerance = o->function_tolerance; st.gradient_tolerance = o->gradient_tolerance;
  st.parameter_tolerance = o->parameter_tolerance; st.max_radius = o->max_trust_region_radius;
  st.min_radius = o->min_trust_region_radius; st.min_relative_decrease = o->min_relative_decrease;
  st.max_num_iterations = o->max_num_iterations; st.max_invalid = o->max_num_consecutive_invalid_steps;
  *e->h_lm = st;
  __atomic_thread_fence(__ATOMIC_RELEASE);
  ResidentParams P{};
  P.frames = e->d_frames; P.desc = e->d_desc; P.w2 = e->d_w2; P.tile_info = e->d_tile_info; P.lane_rec = e->d_lane_rec;
  for (int k = 0; k < 2; ++k) {
    P.xyz[k] = e->d_xyz[k]; P.rec[k] = e->d_rec[k]; P.cams[k] = e->d_cams[k]; P.geom[k] = e->d_geom[k];
    P.block_cost[k] = e->d_block_cost[k]; P.block_fail[k] = e->d_block_fail[k];
  }
  P.sp = e->d_sp; P.ptrec = e->d_ptrec; P.delta_c = e->d_delta_c; P.sc = e->d_sc; P.packed = e->d_packed; P.partial = e->d_partial;
  P.scal = e->d_scal; P.block_bs = e->d_bs_out; P.tab = e->d_solve_tab; P.rec_stride = e->rec_stride;
  P.n_tiles = e->n_tiles; P.n_obs = e->n_obs; P.n_frames = e->n_frames; P.n_free = e->n_free; P.n_pairs = e->n_pairs;
  P.part_stride = e->part_stride; P.fixed_slot = e->fixed_slot; P.rows = e->cfg.rows; P.cols = e->cfg.cols; P.jacobi = o->jacobi_scaling;
  P.cur0 = e->cur; P.max_num_iterations = o->max_num_iterations;
  P.fx = e->cfg.fx; P.fy = e->cfg.fy; P.cx = e->cfg.cx; P.cy = e->cfg.cy; P.huber = e->cfg.huber;
  P.min_diag = o->min_lm_diagonal; P.max_diag = o->max_lm_diagonal; P.radius0 = o->initial_trust_region_radius;
  // epochs: one per step trip; the launch reserves max_num_iterations + 8 of them, pba_internal_resident_done hands back what the solve did not use
  // (the reference's 500-iteration limit against ~30 iterations actually taken: the 32-bit epochs then last ten times longer)
  // (32-bit epochs: a zeroed word would validate at epoch 0 and a stale one 2^32 epochs -- about a day of back-to-back solves -- later, so long
  // before the count wraps the block is zeroed again, stream-ordered behind the previous solve, and the count restarts)
  const unsigned need = (unsigned)std::max(0, o->max_num_iterations) + 8u;
  if (e->res_epoch > pba_engine::kResEpochLimit - need) {
    HIP_TRY(e, hipMemsetAsync(e->d_res_sync, 0, kResSyncBytes, e->stream));
    e->res_epoch = 1;
  }
  P.sync = e->d_res_sync; P.epoch0 = e->res_epoch;
  e->res_epoch_launch = e->res_epoch;
  e->res_epoch += need;
  // every device-side wait is bounded (a lost flag must not hang the GPU): well below the host watchdog and the compute-queue's own
  P.timeout_ticks = (unsigned long long)(std::min(2.0, 0.25 * e->wait_timeout_s) * e->tick_hz);
  P.lm_init = e->h_lm_dev; P.host_state = e->h_lm_dev; P.log = e->d_log; P.host_log = e->h_log_dev; P.max_log = (int)pba_engine::kMaxLog;
  P.host_scal = e->h_scal_dev; P.host_seq = reinterpret_cast<unsigned long long*>(e->h_scal_dev + kNumScal);
  const unsigned long long seq = ++e->seq;
  P.seq = seq;
  P.stamp = nullptr;
  if (const char* sv = getenv("PBA_RES_STOP")) P.debug_stop = atoi(sv);
  static const bool res_trace = getenv("PBA_RES_TRACE") != nullptr;      // dev aid: per-phase stamps of the serial workgroup, printed after the solve
  if (res_trace && o->max_num_iterations + 2 <= kStampMaxIters) {
    const int rcs = dev_alloc(e, &e->d_stamp, (size_t)(kStampMaxIters + 1) * kStampRecord);
    if (rcs) return rcs;
    HIP_TRY(e, hipMemsetAsync(e->d_stamp, 0, sizeof(unsigned long long) * kResStampRecord * (o->max_num_iterations + 2), e->stream));
    P.stamp = e->d_stamp;
    if (atoi(getenv("PBA_RES_TRACE")) >= 2) {
      if (!e->d_dbg) { (void)hipMalloc(reinterpret_cast<void**>(&e->d_dbg), sizeof(unsigned long long) * 8 * (1024 + 4096)); }
      P.schur_dbg = e->d_dbg;
    }
  }
  if (!P.stamp && e->stamps && e->d_stamp && o->max_num_iterations + 2 <= kStampMaxIters) {
    // pba_set_profiling(e, 2): the serial workgroup's phase stamps (the block holds kStampMaxIters + 1 records of kStampRecord words)
    static_assert((int)kResStampRecord <= (int)kStampRecord, "resident stamp records fit the block");
    HIP_TRY(e, hipMemsetAsync(e->d_stamp, 0, sizeof(unsigned long long) * kResStampRecord * (o->max_num_iterations + 2), e->stream));
    P.stamp = e->d_stamp;
  }
  const int groups = (e->n_tiles + 1) / 2;
  void* args[] = {&P};
  // A cooperative launch can be REFUSED by the runtime before anything runs (the grid does not fit the co-residency this process is
  // granted on this device: CU masks, partition modes, a queue limit).  That is not an error of the solve: the epochs and the sequence
  // number go back, this engine stays on the pipelined driver from here on and the caller runs the same solve there (same device
  // functions, same bits).  PBA_RES_REFUSE=1 injects the refusal (tests/test_gpu_resident.py).
  const char* inject = getenv("PBA_RES_REFUSE");
  const hipError_t le = (inject && atoi(inject) != 0) ? hipErrorCooperativeLaunchTooLarge
      : hipLaunchCooperativeKernel(resident_kernel_for(e), dim3(groups), dim3(kResThreads), args, (unsigned)resident_pool_for(e), e->stream);
  if (le != hipSuccess) {
    (void)hipGetLastError();
    e->res_epoch = e->res_epoch_launch;
    e->seq = seq - 1;
    e->use_resident = 0;
    std::fprintf(stderr, "[pba] cooperative launch of %d workgroups refused (%s): this engine solves on the pipelined driver from here on\n", groups,
                 hipGetErrorString(le));
    return PBA_INTERNAL_RESIDENT_REFUSED;
  }
  e->res_launches++;
  e->last_driver = 1;
  e->stamp_iter = o->max_num_iterations;
  e->cost_blocks[0] = e->cost_blocks[1] = groups;
  *seq_out = seq;
  return PBA_OK;
}

const void* pba_internal_async_state(const pba_engine* e) { return e->h_lm; }
const pba_iteration_summary* pba_internal_async_log(const pba_engine* e) { return e->h_log; }

// Leaves the engine in the state a synchronous solve would: current parity, valid linearisation.
int pba_internal_async_end(pba_engine* e) {
  // the host mirror was written by the last publishing kernel the caller waited for; kernels enqueued after a
  // termination are no-ops, so nothing behind it touches the state
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  { const int rcc = comm_failed(e); if (rcc) return rcc; }
  const LmState st = *e->h_lm;
  e->cur = st.cur;
  e->lin_valid[e->cur] = true; e->lin_valid[1 - e->cur] = false;
  e->have_lin = true;
  return PBA_OK;
}

void pba_internal_set_speculate(pba_engine* e, int on) {
  static const bool forced_off = [] { const char* sv = getenv("PBA_SPECULATE"); return sv && atoi(sv) == 0; }();
  e->speculate = on != 0 && !forced_off;
}
void pba_internal_pass_counts(const pba_engine* e, int64_t* jac, int64_t* cost) { *jac = e->jac_passes; *cost = e->cost_passes; }
void pba_internal_reset_pass_counts(pba_engine* e) { e->jac_passes = 0; e->cost_passes = 0; }
int pba_internal_allreduce_host(pba_engine* e, double* v, int n, int op) {
  if (!e->comm.multi()) return 0;
  return e->comm.allreduce_host(v, n, op);
}
}  // extern "C"
