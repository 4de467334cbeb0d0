// pba_resident.h -- the RESIDENT Levenberg-Marquardt solve: ONE cooperative launch per pba_solve (gfx950 only).
//
// The three-kernel path (k_schur -> k_reduce_solve -> k_sample, pba_kernels.h / pba_solve.h) re-reads, in every kernel of every
// iteration, state that did not change since the last one: tile descriptor -> observation indices -> point, Jacobi scale, damped
// point inverse -> Jacobian-pass records -> descriptors.  At the reference's own operating point (config/kitti_stereo.cfg: window 5,
// 3x3 patches, ~20-25 k residual blocks, 29-41 iterations per optimize(), reference src/photobundle.cc:764-876) those dependent
// round trips ARE the iteration: 45.8 us, and 44.4 us at half the size (profiles/r05/small_window_kernel_stats.csv).
//
// Here every 256-thread workgroup owns a FIXED pair of whole-point tiles (the same pair a fused k_sample workgroup works on, the
// same two tiles two k_schur workgroups would) for the whole solve and keeps, per lane = observation, in REGISTERS (ResLane): the
// indices, the point (current | candidate), the Jacobian-pass record (current | candidate), the Jacobi scale, P | g_p | D^2 and the
// descriptor; the camera tables of the current and the candidate point stay in LDS.  Per iteration only these cross HBM / L2:
//   texels of the footprints                                   (sampling phase)
//   ONE Schur partial per tile, write-through                   -> reducing workgroups -> packed reduced system
//   camera step + candidate camera table, 2-7 KB                <- the serial workgroup (reduced solve)
//   4 doubles of block partials per workgroup                   -> the serial workgroup (trust-region decision)
//   the decision (parity, radius, termination), 32 bytes        <- the serial workgroup
// Phases are separated by flag words instead of kernel boundaries: every producer stores ITS OWN epoch word (64 bytes apart, no
// read-modify-write: 256 same-address atomics cost 3.7 us on this part, a gather of 256 private flags + one broadcast flag 1.8 us --
// tools/probes/grid_sync_probe.hip, profiles/r06/grid_sync_probe.txt), consumers poll with agent-scope loads.  Workgroup 0 is the
// SERIAL workgroup: it runs the reduced solve and the trust-region decision, so the LM state, the iteration log, the Jacobi scales of
// the cameras and the cameras themselves are private to one CU and need no cross-XCD coherence at all.
//
// The arithmetic is that of the three-kernel path, literally: the phases call the same device functions (sample_wg, schur_body,
// reduce_partials, solve_blocked, fused_sum_partials, lm_decide) on the same tiles with the same per-tile partials reduced in the same
// fixed order, so a resident solve is bit-identical to an asynchronous one (tests/test_gpu_resident.py).
#pragma once

namespace pba {

constexpr int kResThreads = 256;
constexpr int kResFlagStride = 16;            // u32 words between two flag words (64 bytes)
constexpr int kResMaxGroups = 512;            // workgroups of one resident launch (flag arrays are sized for it)
constexpr int kResMaxReduce = 128;            // virtual reduction blocks (92 + 1 at eight free cameras)
// flag block (u32 words): [arrive1: kResMaxGroups][arrive2: kResMaxReduce][arrive4: kResMaxGroups][go3][go5][abort] each kResFlagStride apart,
// then the decision block (8 doubles)
constexpr int kResFlagArrive1 = 0;
constexpr int kResFlagArrive2 = kResFlagArrive1 + kResMaxGroups;
constexpr int kResFlagArrive4 = kResFlagArrive2 + kResMaxReduce;
constexpr int kResFlagGo3 = kResFlagArrive4 + kResMaxGroups;
constexpr int kResFlagGo5 = kResFlagGo3 + 1;
constexpr int kResFlagAbort = kResFlagGo5 + 1;
constexpr int kResFlagCount = kResFlagAbort + 1;
// ... behind the flags: the decision as three self-validating words (epoch << 32 | radius high | radius low | parity, termination, final
// pass), 64 bytes apart, then one 64-byte line of eight such words per workgroup for its four block partials of the sampling pass
constexpr size_t kResDecisionOffset = sizeof(unsigned) * kResFlagStride * kResFlagCount;      // bytes
constexpr size_t kResTagOffset = kResDecisionOffset + 3 * 64;
// ... then the solve's output -- camera step (6 doubles per frame) and candidate camera table (CamGeom as 8-byte words) -- as tagged word pairs
constexpr size_t kResGeoOffset = kResTagOffset + (size_t)kResMaxGroups * 64;
constexpr int kResGeoDoubles = 6 * kMaxFrames + kMaxFrames * (int)(sizeof(CamGeom) / 8);
constexpr size_t kResSyncBytes = kResGeoOffset + (size_t)kResGeoDoubles * 16;

struct ResidentParams {
  // ---- problem (device memory) ----
  const uint32_t* frames; const float* desc; const double* w2;
  const int4* tile_info; const int2* lane_rec;
  double* xyz[2];                 // [parity]: xyz[cur0] holds the initial points; the final ones are written to xyz[final parity]
  double* rec[2];                 // final Jacobian-pass records -> rec[final parity] (pba_get_obs_records, a later pba_step)
  double* sp; double* ptrec;      // final Jacobi scales / point records (a later pba_step)
  double* cams[2]; CamGeom* geom[2];
  double* delta_c; double* sc; double* packed; double* partial; double* scal;
  double* block_cost[2]; int32_t* block_fail[2]; double* block_bs;
  const uint32_t* tab;
  int64_t rec_stride;
  int32_t n_tiles, n_obs, n_frames, n_free, n_pairs, part_stride, fixed_slot, rows, cols, jacobi, cur0, max_num_iterations;
  double fx, fy, cx, cy, huber, min_diag, max_diag, radius0;
  // ---- hand-over ----
  unsigned* sync;                 // kResSyncBytes of device memory; never reset: epochs grow from launch to launch
  unsigned epoch0;                // first epoch of this launch
  unsigned long long timeout_ticks;   // bound of every wait (100 MHz ticks): a workgroup that times out raises the abort word, everybody leaves
  // ---- trust region (serial workgroup) ----
  const LmState* lm_init;         // host-mapped initial state
  LmState* host_state; pba_iteration_summary* log; pba_iteration_summary* host_log; int32_t max_log;
  double* host_scal; unsigned long long* host_seq; unsigned long long seq;
  unsigned long long* stamp;      // null, or [kResStampRecord * (iterations + 2)] 100 MHz stamps of the serial workgroup
  unsigned long long* schur_dbg;  // development aid (PBA_RES_TRACE=2): [n_tiles][8] per-phase cycle sums of the elimination (thread 0 of every tile)
  int32_t debug_stop;             // development aid (PBA_RES_STOP): leave the loop behind phase k of the first step (0: never)
};
enum ResStamp { kResStampStart = 0, kResStampSchur, kResStampReduced, kResStampSolved, kResStampSampled, kResStampDecided, kResStampSchurBegin, kResStampSchurBody,
                kResStampGathered, kResStampSampleBegin, kResStampRecord = 12 };

__device__ __forceinline__ void res_flag_set(unsigned* sync, int idx, unsigned ep) {
  __hip_atomic_store(sync + (size_t)idx * kResFlagStride, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All threads of the workgroup: wait until the n flags first, first + 1, ... show epoch `ep` (or later).  Returns false when the wait
// timed out or another workgroup raised the abort word (uniform over the workgroup).
// (One poll per round trip.  Keeping four polls of every flag in flight, issued a fraction of a microsecond apart so that a flag is seen
// sooner after it lands, was measured and is SLOWER: the gather of 200 cost partials 2.8 -> 4.8 us, partials + reduction 6.2 -> 9.0 us --
// the extra loads queue in front of the write-through stores they are waiting for.  profiles/r06/resident_phase_trace.txt.)
__device__ __forceinline__ bool res_wait(unsigned* sync, int first, int n, unsigned ep, unsigned long long timeout_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  for (;;) {
    int vote = 0;
    for (int g = threadIdx.x; g < n; g += kResThreads) {
      const unsigned f = __hip_atomic_load(sync + (size_t)(first + g) * kResFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(f - ep) < 0) vote |= 1;
    }
    if (threadIdx.x == kResThreads - 1 && (++spins & 63u) == 0) {
      if (__hip_atomic_load(sync + (size_t)kResFlagAbort * kResFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) vote |= 2;
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
#ifdef PBA_RES_DEBUG_PRINT
        printf("wg %d: wait timed out\n", (int)blockIdx.x);
#endif
        res_flag_set(sync, kResFlagAbort, 1u);
        vote |= 2;
      }
    }
    // (__syncthreads_or returns a PREDICATE, not the OR of the values: two collectives -- the second only while somebody is still behind)
    if (!__syncthreads_or(vote)) return true;
    if (__syncthreads_or(vote & 2)) return false;
  }
}

// The same wait on self-validating words instead of flags: not_ready() = this thread's words do not carry the epoch yet (it loads them
// itself and keeps what it read); uniform result, false on time-out / abort.
template <class F>
__device__ __forceinline__ bool res_poll(unsigned* sync, unsigned long long timeout_ticks, F&& not_ready) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  for (;;) {
    int vote = not_ready() ? 1 : 0;
    if (threadIdx.x == kResThreads - 1 && (++spins & 63u) == 0) {
      if (__hip_atomic_load(sync + (size_t)kResFlagAbort * kResFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) vote |= 2;
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
#ifdef PBA_RES_DEBUG_PRINT
        printf("wg %d: wait timed out\n", (int)blockIdx.x);
#endif
        res_flag_set(sync, kResFlagAbort, 1u);
        vote |= 2;
      }
    }
    // (__syncthreads_or returns a PREDICATE, not the OR of the values: two collectives -- the second only while somebody is still behind)
    if (!__syncthreads_or(vote)) return true;
    if (__syncthreads_or(vote & 2)) return false;
  }
}

// LDS the phases share (they never overlap in time inside a workgroup): sampling | two Schur tiles | reduction + reduced solve
template <int R>
constexpr size_t resident_pool_bytes(int n) {
  size_t b = sizeof(SampleSmem<R, 4>);
  if (b < 2 * kSchurSmemBytes) b = 2 * kSchurSmemBytes;
  if (b < solve_blocked_smem_bytes(n)) b = solve_blocked_smem_bytes(n);
  return (b + 15) / 16 * 16;
}

template <int R, bool UNITW>
__global__ __launch_bounds__(kResThreads) void k_resident(ResidentParams P) {
  constexpr int WAVES = kResThreads / 64;
  constexpr int W = 2 * R + 1;
  extern __shared__ __attribute__((aligned(16))) char pool[];
  __shared__ CamGeom s_geomL[2][kMaxFrames];             // camera tables of the two parities, persistent
  __shared__ CamGeom s_gc[kMaxFrames];                   // serial workgroup: candidate table as the solve's epilogue writes it
  __shared__ double s_dc[6 * kMaxFrames];                // camera step of the step being taken (serial workgroup: written by the solve's epilogue)
  __shared__ double s_red[kReduceThreads / kReduceEntries][kReduceEntries + 1];
  __shared__ int s_f[16];
  __shared__ double s_r4[4 * WAVES];
  __shared__ LmState s_lm;                               // serial workgroup: THE trust-region state of the solve
  __shared__ double s_dec[4];                            // decision as every workgroup reads it: parity, termination, radius, final pass needed
  __shared__ unsigned s_decw[3];                         // ... its three payload words as polled

  const int tid0 = threadIdx.x;
  const int G = gridDim.x;
  const int w = xcd_logical_block(blockIdx.x, G);        // tile pair (2 w, 2 w + 1): an XCD works on one contiguous eighth of the points
  const bool serial = (w == 0);
  const int tile = 2 * w + (tid0 >> 7);
  const bool has_tile = tile < P.n_tiles;
  const int n = 6 * P.n_free;
  const int nred = (P.part_stride + kReduceEntries - 1) / kReduceEntries + 1;
  unsigned* sync = P.sync;
  unsigned long long* dec_g = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sync) + kResDecisionOffset);      // decision words (64 bytes apart)
  unsigned long long* tag_g = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sync) + kResTagOffset);          // [G][8] tagged block partials
  unsigned long long* geo_g = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sync) + kResGeoOffset);          // tagged camera step | candidate table
  unsigned ep = P.epoch0;
  unsigned long long* stamp = (serial && tid0 == 0) ? P.stamp : nullptr;
  if (stamp) stamp[kResStampStart] = __builtin_amdgcn_s_memrealtime();

  // ---- resident state of the lane's observation ------------------------------------------------------------------------------
  ResLane<R> rl;
  rl.ti = make_int4(0, 0, 0, 0); rl.pt = 0; rl.slot = 0; rl.l0 = 0; rl.cnt = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { rl.X[k] = 0.0; rl.Xc[k] = 0.0; rl.sp[k] = 1.0; }
#pragma unroll
  for (int k = 0; k < 6; ++k) { rl.rec[k] = 0.0; rl.recc[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < 12; ++k) rl.pr[k] = 0.0;
#pragma unroll
  for (int k = 0; k < W * W; ++k) rl.desc[k] = 0.f;
  if (has_tile) {
    rl.ti = P.tile_info[tile];
    const int2 r = P.lane_rec[(size_t)tile * kTile + (tid0 & 127)];      // (lanes beyond the tile's observations hold zeros)
    rl.pt = r.x; rl.slot = r.y & 0xff; rl.l0 = (r.y >> 8) & 0xff; rl.cnt = (r.y >> 16) & 0xff;
  }
  const bool active = (tid0 & 127) < rl.ti.y;
  int cur = P.cur0;
  if (active) {
    const double* x0 = P.xyz[cur] + 3 * (size_t)rl.pt;
    rl.X[0] = x0[0]; rl.X[1] = x0[1]; rl.X[2] = x0[2];
    const float* d0 = P.desc + (size_t)rl.pt * (W * W);
#pragma unroll
    for (int k = 0; k < W * W; ++k) rl.desc[k] = d0[k];
  }
  if (serial) {
    static_assert(sizeof(LmState) % 4 == 0, "word copy");
    if (tid0 < (int)(sizeof(LmState) / 4))
      reinterpret_cast<unsigned*>(&s_lm)[tid0] = __hip_atomic_load(reinterpret_cast<const unsigned*>(P.lm_init) + tid0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }

  SampleSmem<R, WAVES>& sm = *reinterpret_cast<SampleSmem<R, WAVES>*>(pool);
  // every store of the phase has left the CU, then the workgroup's own flag
  auto arrive = [&](int flag_idx) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) res_flag_set(sync, flag_idx, ep);
  };

  // ONE loop, one call site per phase (the kernel is ~100 KB of code as it is):
  //   trip 0        sampling at the initial point (first linearisation: a Jacobian pass whose records become the current ones)
  //   step trips    elimination -> reduction -> reduced solve -> back-substitution + sampling at the candidate -> decision
  //   final trip    gradient-only elimination + solve epilogue + decision (iteration limit right behind an accepted step; iteration
  //                 zero of a zero-iteration solve), the step kernels of the three-kernel path with final_pass = 1
  double radius = P.radius0;
  int done = 0, need_final = 1, it = 0, first = 1;
  int which = cur, skip = 1;
  bool ok = true;
  __syncthreads();
  for (;;) {
    // The thread index through an opaque copy per trip: with the plain builtin every per-lane address, mask and index of EVERY phase is
    // loop-invariant, and the compiler hoists them all in front of the loop -- hundreds of registers live across the whole iteration
    // (the first build of this kernel spilled 340 of them to scratch memory).
    int tz = 0;
    asm volatile("" : "+v"(tz));
    const int tid = (int)threadIdx.x + tz, lane = tid & 63, wave = tid >> 6, half = tid >> 7, lt = tid & 127;
    // ---- sampling phase (back-substitution first unless `skip`) ----------------------------------------------------------------
    {
      SampleParams sp{};
      sp.frames = P.frames; sp.desc = P.desc; sp.w2 = P.w2; sp.rec_stride = P.rec_stride; sp.n_obs = P.n_obs; sp.n_frames = P.n_frames;
      sp.rows = P.rows; sp.cols = P.cols; sp.fx = P.fx; sp.fy = P.fy; sp.cx = P.cx; sp.cy = P.cy; sp.huber = P.huber;
      sp.delta_c = s_dc; sp.block_bs = P.block_bs; sp.n_tiles = P.n_tiles;
      sp.geom = skip ? P.geom[which] : nullptr; sp.geom_prev = s_geomL[cur];      // (step trips: both tables and the camera step are in LDS)
      sp.block_cost = P.block_cost[which]; sp.block_fail = P.block_fail[which];
      sp.skip_backsub = skip;
      if (!skip) { sp.res_tag = tag_g; sp.res_epoch = ep; }
      if (stamp) stamp[kResStampSampleBegin] = __builtin_amdgcn_s_memrealtime();
      sample_wg<R, true, WAVES, true, UNITW, false, true>(sp, sm, rl, w, G, tid, s_geomL[which]);
    }
    if (P.debug_stop == (skip ? 1 : 5)) break;
    if (skip) {
#pragma unroll
      for (int k = 0; k < 6; ++k) rl.rec[k] = rl.recc[k];
    } else {
      if (stamp) stamp[kResStampSampled] = __builtin_amdgcn_s_memrealtime();
      // ---- trust-region decision (serial workgroup), read by everybody.  Both hand-overs are SELF-VALIDATING words (epoch << 32 | payload,
      // 64-bit atomic stores, fire and forget): the consumer polls the data itself, so "drain the stores, raise a flag, see the flag, load the
      // data" -- three dependent round trips of 1-2 us -- is one. ----
      if (serial) {
        double* s_bsL = reinterpret_cast<double*>(pool);            // [3 G] | [G]: the gathered partials, laid out like the global arrays
        double* s_costL = s_bsL + 3 * kResMaxGroups;
        auto gather = [&]() -> bool {
          bool behind = false;
          for (int g = tid; g < G; g += kResThreads) {
            double v[4];
            bool okl = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) okl = load_tagged(tag_g + 8 * (size_t)g + 2 * q, ep, v[q]) && okl;
            s_costL[g] = v[0]; s_bsL[3 * g] = v[1]; s_bsL[3 * g + 1] = v[2]; s_bsL[3 * g + 2] = v[3];
            behind = behind || !okl;
          }
          return behind;
        };
        __syncthreads();                                  // (every thread is done with the sampling phase's LDS)
        if (!res_poll(sync, P.timeout_ticks, gather)) { ok = false; break; }
        if (stamp) stamp[kResStampGathered] = __builtin_amdgcn_s_memrealtime();
        fused_sum_partials<WAVES, false>(s_bsL, s_costL, G, lane, wave, s_f, s_r4, nullptr, tid);
        // the serial workgroup's decision for the step and its publication
        if (tid == 0) {
          double sl[kNumScal];
#pragma unroll
          for (int k = 0; k < kNumScal; ++k) sl[k] = load_agent(P.scal + k);      // (max |g_p| and the failure flags came from other workgroups)
          sl[kMccPts] = s_r4[0]; sl[kStep2Pts] = s_r4[WAVES]; sl[kX2Pts] = s_r4[2 * WAVES];
          sl[kCandCost] = s_r4[3 * WAVES]; sl[kEvalFailCand] = (double)s_f[0];
          P.scal[kMccPts] = sl[kMccPts]; P.scal[kStep2Pts] = sl[kStep2Pts]; P.scal[kX2Pts] = sl[kX2Pts];
          P.scal[kCandCost] = sl[kCandCost]; P.scal[kEvalFailCand] = sl[kEvalFailCand];
          lm_decide(&s_lm, sl, P.log, P.max_log, 0);
          if (s_lm.done && s_lm.done_seq == 0) s_lm.done_seq = P.seq;
          s_decw[0] = (unsigned)__double2hiint(s_lm.radius); s_decw[1] = (unsigned)__double2loint(s_lm.radius);
          s_decw[2] = (unsigned)(s_lm.cur & 1) | ((unsigned)s_lm.done << 4) | (lm_final_pass_needed(&s_lm) ? 1u << 12 : 0u);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            __hip_atomic_store(dec_g + 8 * k, ((unsigned long long)ep << 32) | s_decw[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stamp) stamp[kResStampDecided] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
      } else {
        auto decision = [&]() -> bool {
          bool behind = false;
          if (tid < 3) {
            const unsigned long long wv = __hip_atomic_load(dec_g + 8 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            behind = (unsigned)(wv >> 32) != ep;
            s_decw[tid] = (unsigned)wv;
          }
          return behind;
        };
        if (!res_poll(sync, P.timeout_ticks, decision)) { ok = false; break; }
      }
      if (tid == 0) {
        s_dec[0] = (double)(s_decw[2] & 1u); s_dec[1] = (double)((s_decw[2] >> 4) & 0xffu); s_dec[3] = (double)((s_decw[2] >> 12) & 1u);
        s_dec[2] = __hiloint2double((int)s_decw[0], (int)s_decw[1]);
      }
      __syncthreads();
    }
    int final_pass = 0;
    if (!skip && P.debug_stop == 6) break;
    if (!skip) {
      const int new_cur = (int)s_dec[0];
      done = (int)s_dec[1]; radius = s_dec[2]; need_final = (int)s_dec[3];
      if (new_cur != cur) {      // accepted: the candidate becomes the current point, its (Jacobian-pass) records the linearisation
#pragma unroll
        for (int k = 0; k < 3; ++k) rl.X[k] = rl.Xc[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) rl.rec[k] = rl.recc[k];
        cur = new_cur;
      }
      ++it;
    }
    // ---- what comes next ------------------------------------------------------------------------------------------------------
    if (done || it >= P.max_num_iterations) {
      if (!need_final) break;
      final_pass = 1;
    }
    const int init_scale = final_pass ? (P.max_num_iterations <= 0 ? 1 : 0) : first;
    const bool grad_only = final_pass && !init_scale;
    first = 0;
    ++ep;
    if (stamp) stamp += kResStampRecord;
    // ---- Schur elimination of the workgroup's two tiles at the current point ----------------------------------------------------
    {
      SchurParams sc{};
      sc.partial = P.partial; sc.rec_stride = P.rec_stride; sc.n_tiles = P.n_tiles; sc.n_frames = P.n_frames; sc.n_free = P.n_free;
      sc.n_pairs = P.n_pairs; sc.part_stride = P.part_stride; sc.init_scale = init_scale; sc.jacobi = P.jacobi; sc.fx = P.fx; sc.fy = P.fy;
      sc.radius = radius; sc.inv_radius = 1.0 / radius; sc.min_diag = P.min_diag; sc.max_diag = P.max_diag; sc.final_pass = final_pass; sc.dbg = final_pass ? nullptr : P.schur_dbg;
      __syncthreads();                                    // the pool changes hands: sampling -> Schur tiles (s_dec is free again, too)
      if (stamp) stamp[kResStampSchurBegin] = __builtin_amdgcn_s_memrealtime();
      schur_body<ResLane<R>>(sc, pool + (size_t)half * kSchurSmemBytes, lt, tile, P.n_tiles, &rl, s_geomL[cur], has_tile);
      if (stamp) stamp[kResStampSchurBody] = __builtin_amdgcn_s_memrealtime();
    }
    if (P.debug_stop == 2) break;
    // (fault injection, PBA_RES_STOP=100, tests/test_gpu_resident.py: in the second step the last workgroup keeps its flag to itself -- every
    // wait that depends on it must end in the time-out / abort path, the kernel must leave, the host must report an error instead of hanging)
    if (!(P.debug_stop == 100 && it == 1 && w == G - 1 && !serial))
    arrive(kResFlagArrive1 + w);
    if (stamp) stamp[kResStampSchur] = __builtin_amdgcn_s_memrealtime();
    // ---- reduction of the per-tile partials: virtual blocks w, w + G, ... of the reduction grid -----------------------------------
    if (w < nred) {
      if (!res_wait(sync, kResFlagArrive1, G, ep, P.timeout_ticks)) { ok = false; break; }
      ReduceParams rp{};
      rp.partial = P.partial; rp.n_blocks = P.n_tiles; rp.stride = P.part_stride; rp.n_free = P.n_free; rp.n_pairs = P.n_pairs;
      rp.block_cost = P.block_cost[cur]; rp.block_fail = P.block_fail[cur]; rp.n_cost_blocks = G; rp.packed = P.packed; rp.scal = P.scal;
      rp.first_entry = grad_only ? 36 * P.n_pairs + n : 0;
      for (int v = w; v < nred; v += G) {
        reduce_partials<1, kResThreads, true, 16>(rp, s_red, s_f, v, nred, tid);
        arrive(kResFlagArrive2 + v);
      }
    }
    if (P.debug_stop == 3) break;
    // ---- reduced camera solve (serial workgroup) -----------------------------------------------------------------------------------
    if (serial) {
      if (!res_wait(sync, kResFlagArrive2, nred, ep, P.timeout_ticks)) { ok = false; break; }
      if (stamp) stamp[kResStampReduced] = __builtin_amdgcn_s_memrealtime();
      SolveParams so{};
      so.packed = P.packed; so.cams = P.cams[cur]; so.cams_cand = P.cams[1 - cur]; so.delta_c = s_dc; so.sc = P.sc; so.scal = P.scal;
      so.geom = P.geom[cur]; so.geom_cand = final_pass ? nullptr : s_gc; so.tab = P.tab;
      so.n_frames = P.n_frames; so.n_free = P.n_free; so.n_pairs = P.n_pairs; so.stride = P.part_stride; so.fixed_slot = P.fixed_slot;
      so.init_scale = init_scale; so.jacobi = P.jacobi; so.radius = radius; so.min_diag = P.min_diag; so.max_diag = P.max_diag;
      so.final_pass = final_pass;
      solve_blocked<true, kSolveBlockedThreads>(so, reinterpret_cast<double*>(pool), tid);
      __syncthreads();
      if (final_pass) {
        // gradient norms of the final point: the gradient-only decision (k_decide of the three-kernel path)
        if (tid == 0) {
          double sl[kNumScal];
#pragma unroll
          for (int k = 0; k < kNumScal; ++k) sl[k] = load_agent(P.scal + k);
          lm_decide(&s_lm, sl, P.log, P.max_log, 1);
          if (s_lm.done && s_lm.done_seq == 0) s_lm.done_seq = P.seq;
        }
      } else {
        // camera step and candidate table -> every workgroup, as self-validating word pairs (epoch << 32 | half a double: fire and forget, the
        // consumers poll the data itself); the serial workgroup's own copies go LDS -> LDS; the table in global memory (plain stores) is for
        // the calls behind pba_solve
        const int words = P.n_frames * (int)(sizeof(CamGeom) / 8);
        const double* src = reinterpret_cast<const double*>(s_gc);
        double* mine = reinterpret_cast<double*>(s_geomL[1 - cur]);
        double* glob = reinterpret_cast<double*>(P.geom[1 - cur]);
        for (int k = tid; k < 6 * P.n_frames; k += kResThreads) store_tagged(geo_g + 2 * k, ep, s_dc[k]);
        for (int k = tid; k < words; k += kResThreads) {
          const double v = src[k];
          store_tagged(geo_g + 2 * (6 * kMaxFrames + k), ep, v);
          mine[k] = v; glob[k] = v;
        }
        if (stamp) stamp[kResStampSolved] = __builtin_amdgcn_s_memrealtime();
      }
    }
    if (final_pass || P.debug_stop == 4) break;
    // ---- everybody: the camera step is there ------------------------------------------------------------------------------------------
    if (!serial) {
      const int words = P.n_frames * (int)(sizeof(CamGeom) / 8);
      double* tab = reinterpret_cast<double*>(s_geomL[1 - cur]);
      auto tables = [&]() -> bool {
        bool behind = false;
        for (int k = tid; k < 6 * P.n_frames + words; k += kResThreads) {
          const bool is_dc = k < 6 * P.n_frames;
          const int idx = is_dc ? k : 6 * kMaxFrames + (k - 6 * P.n_frames);
          double v;
          if (!load_tagged(geo_g + 2 * idx, ep, v)) behind = true;
          if (is_dc) s_dc[k] = v; else tab[k - 6 * P.n_frames] = v;
        }
        return behind;
      };
      if (!res_poll(sync, P.timeout_ticks, tables)) { ok = false; break; }
    }
    __syncthreads();                                    // the pool changes hands: Schur tiles / solve -> sampling; the tables are in LDS
    which = 1 - cur; skip = 0;
  }
#ifdef PBA_RES_DEBUG_PRINT
  if (threadIdx.x == 0 && !ok) printf("wg %d (w %d) leaves the loop, ok %d it %d\n", (int)blockIdx.x, w, (int)ok, it);
#endif
  // ---- state back to global memory for the calls behind pba_solve (pba_get_state, pba_get_obs_records, pba_step) -----------------
  if (active) {
    if ((tid0 & 127) == rl.l0) {
      double* xo = P.xyz[cur] + 3 * (size_t)rl.pt;
      xo[0] = rl.X[0]; xo[1] = rl.X[1]; xo[2] = rl.X[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) P.sp[3 * (size_t)rl.pt + k] = rl.sp[k];
#pragma unroll
      for (int k = 0; k < 12; ++k) P.ptrec[12 * (size_t)rl.pt + k] = rl.pr[k];
    }
    const int obs = rl.ti.x + (tid0 & 127);
#pragma unroll
    for (int k = 0; k < 6; ++k) P.rec[cur][k * P.rec_stride + obs] = rl.rec[k];
  }
  if (serial) {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // thread 0's log / scalar stores are read by the whole workgroup
    if (ok) flush_to_host(&s_lm, P.host_state, P.scal, P.host_scal, P.log, P.host_log, P.max_log, P.host_seq, P.seq, tid0, kResThreads);
  }
}

}  // namespace pba
