// pba_solve.h -- the serial stretch of an LM iteration: reduction of the per-workgroup Schur partials, the reduced
// camera system (scaling, damping, dense L D L^T, camera step, candidate cameras and their geometry).
// Replaces Ceres' SchurComplementSolver / dense Cholesky of the reduced system (reference src/photobundle.cc:743, :829).
// Included by pba_kernels.h (needs its helpers); gfx950 only.
#pragma once

namespace pba {

// =====================================================================================================
// Packed reduced system, "tri" layout (doubles), n = 6 n_free, N1 = n + 1:
//   [0, TRI)            lower triangle of the AUGMENTED matrix [S y; y^T .], row-major packed: (r, c <= r) at r (r + 1) / 2 + c;
//                       row n holds the right-hand side (entry (n, n) unused).  TRI = N1 (N1 + 1) / 2
//   [TRI, +n) g_c   [.., +n) diag(U)   [TRI + 2 n] cost at the linearisation point   [TRI + 2 n + 1] sum g_p^2
// Every entry is a plain SUM over ranks (the multi-rank exchange reduces all packed_stride(n) of them).  The destination of
// a k_schur partial entry is a pure function of the window shape, so the SCATTER happens on the write side of the
// reduction (72+ workgroups in parallel, index math under the loads) and the solving workgroup's prologue is a straight
// copy: round 3 spent 6.8 k of its 37.7 k cycles decoding pair indices in the one workgroup everybody waits for.
// =====================================================================================================
__host__ __device__ inline int tri_index(int r) { return r * (r + 1) / 2; }
__host__ __device__ inline int packed_stride(int n) { return tri_index(n + 1) + 2 * n + 2; }

// Update work items of the blocked solve: (row r, block column j), ordered by block column j = 1 .. nf - 1, rows 6 j .. n within one
// (the items of panel k are the suffix j >= k + 2); first item of block column j:
__host__ __device__ inline int solve_item_base(int j, int N1) { return (j - 1) * (N1 - 3 * j); }
// Host: the two index tables of the solve for n_free free cameras (pure functions of the window shape; round 3 decoded them in
// the solving workgroup, ~6 k cycles on the serial stretch of every LM iteration).  tab must hold tri_index(n + 1) + items words.
inline int solve_table_words(int nf) { const int n = 6 * nf; return tri_index(n + 1) + (nf > 1 ? solve_item_base(nf, n + 1) : 0); }
inline void solve_tables(int nf, uint32_t* tab) {
  const int n = 6 * nf, N1 = n + 1, TRI = tri_index(N1);
  for (int r = 0; r < N1; ++r)
    for (int c = 0; c <= r; ++c) tab[tri_index(r) + c] = ((uint32_t)r << 16) | (uint32_t)c;
  for (int j = 1; j < nf; ++j)
    for (int r = 6 * j; r <= n; ++r) tab[TRI + solve_item_base(j, N1) + (r - 6 * j)] = (uint32_t)r | ((uint32_t)j << 16);
}

// k_schur partial entry e (T pair blocks | rhs | g_c | diag U | 3 tail values) -> tri-layout index; -1: dropped (lower
// triangle of a diagonal block: its mirror image is the one that is kept, as in rounds 1-3); -2: one of the tail values.
__device__ __forceinline__ int packed_dest(int e, int nf, int n_pairs) {
  const int n = 6 * nf, nT = 36 * n_pairs;
  if (e < nT) {
    const int pair = PBA_PARTIAL_T ? e % n_pairs : e / 36, k = PBA_PARTIAL_T ? e / n_pairs : e - 36 * pair, i = k / 6, j = k - 6 * i;
    int a = 0, rem = pair;
    while (rem >= nf - a) { rem -= nf - a; ++a; }      // pairs enumerated row by row: (a, a..nf-1)
    const int b = a + rem;
    if (a == b && j < i) return -1;
    return tri_index(6 * b + j) + 6 * a + i;           // entry (6a+i, 6b+j) of the upper triangle = (row 6b+j, col 6a+i) below
  }
  const int v = e - nT;
  if (v < n) return tri_index(n) + v;                  // rhs = row n of the augmented matrix
  if (v < 3 * n) return tri_index(n + 1) + (v - n);    // g_c | diag U
  return -2;
}

template <int STORE> __device__ __forceinline__ void packed_store(double* p, double v) {
  if (STORE == 2) store_system_f64(p, v);      // this rank's peer-exchange mailbox (read by other devices)
  else if (STORE == 1) store_agent(p, v);      // consumed by another workgroup of the same launch
  else *p = v;
}

// Fixed-order reduction of the per-block partials: workgroup = kReduceEntries entries x 32 sub-chunks of blocks; thread
// (ex, sub) sums blocks sub, sub + 32, ... (32 loads in flight), the four sub-chunks of a wave combine by two cross-lane
// steps, the 8 waves through LDS in wave order.  The LAST workgroup of the grid sums the Jacobian-pass block costs.
// 512 threads (r3: 1024): the workgroup that goes on to SOLVE may then use 256 registers per lane -- the panel wave of the
// look-ahead factorisation holds two 6x6 triangles, the six block rows of L and its own rows (~180): at 1024 threads the
// cap is 128 and it spilled 166 of them.
constexpr int kReduceEntries = 16;      // packed entries per workgroup (x 32 sub-chunks of blocks)
constexpr int kReduceThreads = 512;
constexpr int kReduceInFlight = 32;
struct ReduceParams {
  const double* partial; int32_t n_blocks, stride;      // k_schur partials: `stride` entries per workgroup, [entry / 16][n_blocks][16]
  int32_t n_free, n_pairs;
  const double* block_cost; const int32_t* block_fail; int32_t n_cost_blocks;
  double* packed; double* scal;
  int32_t first_entry;      // entries below it are neither read nor stored (gradient-only final pass: the pair blocks and the rhs)
};

// T = threads of the calling workgroup: kReduceThreads (the reduction kernels), or 256 in the resident solve, where every thread
// plays 512 / T of the 512 (entry, sub-chunk) roles -- the same loads, the same sums in the same order.  (vb, nvb) = this workgroup's
// index in / size of the reduction grid (blockIdx.x, gridDim.x in the kernels; virtual blocks in the resident solve).  AGL: the
// partials and block costs were written by other workgroups of the SAME launch (resident solve): agent-scope loads.
// NF = partials in flight per (entry, sub-chunk) role: the same sums in the same order for every NF (the resident solve, <= 512 partials, uses 16).
template <int STORE, int T = kReduceThreads, bool AGL = false, int NF = kReduceInFlight>
__device__ __forceinline__ void reduce_partials(const ReduceParams& rp, double (*s_red)[kReduceEntries + 1], int* s_f, const int vb, const int nvb,
                                                const int tid = (int)threadIdx.x) {
  constexpr int EX = kReduceEntries, SUB = kReduceThreads / EX;
  constexpr int ROLES = kReduceThreads / T;
  static_assert(T * ROLES == kReduceThreads && T % 64 == 0, "whole waves of roles");
  const int stride = rp.stride, n_blocks = rp.n_blocks;
  const int n = 6 * rp.n_free, TRI = tri_index(n + 1);
  auto ldp = [&](const double* q) { return AGL ? load_agent(q) : *q; };
  if (vb < nvb - 1) {
    const int ex = tid % EX;                       // (T is a multiple of EX: the same for every role of the thread)
    const int e = vb * EX + ex;
    const bool valid = e < stride && e >= rp.first_entry;
    const bool is_max = (e == stride - 3) || (e == stride - 1);
    double acc[ROLES];
    double v[ROLES][NF];
    if (valid) {
#pragma unroll
      for (int r = 0; r < ROLES; ++r) {
        const int sub = (tid + r * T) / EX;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          const int bb = sub + SUB * k;
          v[r][k] = (bb < n_blocks) ? ldp(rp.partial + ((size_t)vb * n_blocks + bb) * EX + ex) : 0.0;
        }
      }
    }
    // destination of this entry: integer work under the loads
    const int dest = (valid && tid < EX) ? packed_dest(e, rp.n_free, rp.n_pairs) : -1;
#pragma unroll
    for (int r = 0; r < ROLES; ++r) {
      const int sub = (tid + r * T) / EX;
      acc[r] = 0.0;
      if (valid) {
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[r] = is_max ? fmax(acc[r], v[r][k]) : acc[r] + v[r][k];
        for (int b = sub + NF * SUB; b < n_blocks; b += NF * SUB) {      // (grids beyond 1024 partials: not used today)
#pragma unroll
          for (int k = 0; k < NF; ++k) {
            const int bb = b + SUB * k;
            v[r][k] = (bb < n_blocks) ? ldp(rp.partial + ((size_t)vb * n_blocks + bb) * EX + ex) : 0.0;
          }
#pragma unroll
          for (int k = 0; k < NF; ++k) acc[r] = is_max ? fmax(acc[r], v[r][k]) : acc[r] + v[r][k];
        }
      }
      static_assert(EX == 16, "lane = 16 (sub % 4) + ex");
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const double o = __shfl_xor(acc[r], off);
        acc[r] = is_max ? fmax(acc[r], o) : acc[r] + o;
      }
      if ((tid & 63) < EX) s_red[(tid + r * T) >> 6][ex] = acc[r];
    }
    __syncthreads();
    if (tid < EX && valid) {
      double s = s_red[0][ex];
#pragma unroll
      for (int w = 1; w < kReduceThreads / 64; ++w) s = is_max ? fmax(s, s_red[w][ex]) : s + s_red[w][ex];
      if (dest >= 0) packed_store<STORE>(rp.packed + dest, s);
      else if (e == stride - 3) packed_store<STORE>(rp.scal + kGmaxPts, s);      // (STORE 1: read by the last workgroup of this launch when it decides)
      else if (e == stride - 2) packed_store<STORE>(rp.packed + TRI + 2 * n + 1, s);
      else if (e == stride - 1) packed_store<STORE>(rp.scal + kSchurFail, s);
    }
  } else {
    // last workgroup: cost of the linearisation point = fixed-order sum of the Jacobian-pass block partials (strided per
    // thread, butterfly per wave, the 8 waves in order: one barrier instead of a ten-level LDS tree)
#pragma unroll
    for (int r = 0; r < ROLES; ++r) {
      double acc = 0.0; int f = 0;
      for (int b = tid + r * T; b < rp.n_cost_blocks; b += kReduceThreads) {
        acc += ldp(rp.block_cost + b);
        f |= AGL ? __hip_atomic_load(rp.block_fail + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rp.block_fail[b];
      }
      acc = wave_sum(acc);
      f = __any(f) ? 1 : 0;
      if ((tid & 63) == 0) { s_red[(tid + r * T) >> 6][0] = acc; s_f[(tid + r * T) >> 6] = f; }
    }
    __syncthreads();
    if (tid == 0) {
      double s = s_red[0][0]; int ff = s_f[0];
      for (int w = 1; w < kReduceThreads / 64; ++w) { s += s_red[w][0]; ff |= s_f[w]; }
      packed_store<STORE>(rp.packed + TRI + 2 * n, s);
      packed_store<STORE>(rp.scal + kEvalFailLin, (double)ff);      // (read by the LAST workgroup of this launch when it decides: not a plain store)
      if (n >= 0) packed_store<STORE>(rp.packed + tri_index(n) + n, 0.0);      // the unused corner (n, n) of the augmented matrix
    }
  }
}

// =====================================================================================================
// reduced camera system
// =====================================================================================================
struct SolveParams {
  const double* packed;     // reduced (and, multi-rank, all-reduced) packed sums, tri layout
  const double* cams;       // current cameras [n_frames][6]
  double* cams_cand;        // candidate cameras
  double* delta_c;          // [n_frames][6] unscaled camera step (0 for the constant camera)
  double* sc;               // [2][6 n_free] Jacobi scale of the camera columns | column-is-live flags (written when init_scale)
  double* S_dbg;            // [n*n] scaled + damped reduced matrix (test hook), may be null
  double* rhs_dbg;          // [n]
  double* scal;
  const CamGeom* geom;      // current geometry (free_index of every slot)
  CamGeom* geom_cand;       // candidate geometry output (null: produced elsewhere)
  const uint32_t* tab;      // window-shape tables built by the host at pba_set_cameras (solve_tables): [TRI] row << 16 | column of the
                            // packed triangle entries, then the update work items row | block column << 16
  int32_t n_frames, n_free, n_pairs, stride, fixed_slot;
  int32_t init_scale, jacobi;
  double radius, min_diag, max_diag;
  // asynchronous driver
  const LmState* lm;
  int32_t enq_cur, final_pass;
  int32_t dbg;
  const double* cams_alt; double* cams_cand_alt; const CamGeom* geom_alt; CamGeom* geom_cand_alt;
  // multi-rank with the peer exchange: the packed sums are the rank-ordered sum of every rank's mailbox slot, formed in the
  // prologue of the solve itself (no separate exchange kernel): peer_world > 0, mailboxes in `peer`
  PeerParams peer;
  int32_t peer_world, peer_flag;
  unsigned long long peer_off, peer_seq, peer_timeout;
  unsigned int* peer_err;
};

__device__ __forceinline__ bool solve_resolve(SolveParams& p) {
  if (!p.lm) return true;
  if (p.lm->done && !p.final_pass) return false;
  if (p.final_pass && !lm_final_pass_needed(p.lm)) return false;
  if (p.lm->cur != p.enq_cur) {
    p.cams = p.cams_alt; p.cams_cand = p.cams_cand_alt; p.geom = p.geom_alt;
    if (p.geom_cand) p.geom_cand = p.geom_cand_alt;
  }
  p.radius = p.lm->radius;
  return true;
}

// flat index of a packed lower-triangular entry -> (row, column)
__device__ __forceinline__ void tri_row_col(int t, int& r, int& c) {
  r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);      // exact to +-1 for t < 2^20
  while (tri_index(r + 1) <= t) ++r;
  while (tri_index(r) > t) --r;
  c = t - tri_index(r);
}

// LOADER of the packed sums: plain loads (an earlier kernel / the all-reduce produced them), agent-scope loads (other
// workgroups of THIS launch did, k_reduce_solve), or the rank-ordered sum over the peer mailboxes.
template <bool AGENT>
__device__ __forceinline__ double packed_load(const SolveParams& p, int idx) {
  if (AGENT) return load_agent(p.packed + idx);
  if (p.peer_world > 0) return peer_sum(p.peer, p.peer_world, p.peer_off, idx);
  return p.packed[idx];
}

// Generic path (any n <= 96; diagnostics, PBA_SOLVE=1): matrix in LDS, 256 threads, 2-D trailing update, one barrier pair
// per column, plain Cholesky.  Independent of the blocked solve below (the A/B reference for it).
constexpr int kSolveThreads = 256;

template <int T>
__device__ __forceinline__ void solve_prologue(const SolveParams& p, int n, int ld, double* S, double* y, double* sc,
                                               double* D2, double* gcs, double* gc, int tid) {
  const int TRI = tri_index(n + 1);
  for (int i = tid; i < n; i += T) {
    const double du = packed_load<false>(p, TRI + n + i);
    const double g = packed_load<false>(p, TRI + i);
    double s;
    // sc[n + i]: the column has a nonzero norm at the initial point.  A free camera whose six flags are all zero has no
    // residual block anywhere (the sums are global at world > 1): not a parameter block of the Ceres program
    if (p.init_scale) { s = p.jacobi ? 1.0 / (1.0 + sqrt(du)) : 1.0; p.sc[i] = s; p.sc[n + i] = du > 0.0 ? 1.0 : 0.0; }
    else s = p.sc[i];
    sc[i] = s;
    D2[i] = fmin(fmax(s * s * du, p.min_diag), p.max_diag) / p.radius;
    gc[i] = g;
    gcs[i] = s * g;
  }
  __syncthreads();
  for (int t = tid; t < TRI - 1; t += T) {
    int r, c;
    tri_row_col(t, r, c);
    const double val = packed_load<false>(p, t);
    if (r == n) { y[c] = sc[c] * val; continue; }
    double v = sc[c] * val * sc[r];
    if (r == c) v += D2[r];
    S[r * ld + c] = v;
    S[c * ld + r] = v;
  }
  __syncthreads();
  if (p.S_dbg) {
    for (int k = tid; k < n * n; k += T) p.S_dbg[k] = S[(k / n) * ld + (k % n)];
    for (int i = tid; i < n; i += T) p.rhs_dbg[i] = y[i];
  }
  __syncthreads();
}

template <int T>
__device__ __forceinline__ void solve_epilogue(const SolveParams& p, int n, const double* y, const double* sc,
                                               const double* D2, const double* gcs, const double* gc, bool chol_ok,
                                               int tid) {
  if (tid < 6 * p.n_frames && !(p.final_pass && !p.init_scale)) {
    const int slot = tid / 6, k = tid % 6;
    const int fa = p.geom[slot].free_index;
    double d = 0.0;
    if (fa >= 0) d = -sc[6 * fa + k] * y[6 * fa + k];
    p.delta_c[tid] = d;
    p.cams_cand[tid] = p.cams[tid] + d;
  }
  if (p.geom_cand) {
    // candidate camera geometry by the last wave (reads cams + delta directly: no dependency on the stores above)
    const int c = tid - (T - 64);
    if (c >= 0 && c < p.n_frames) {
      double cam6[6];
      const int fa = p.geom[c].free_index;
      for (int k = 0; k < 6; ++k) cam6[k] = p.cams[6 * c + k] + (fa >= 0 ? -sc[6 * fa + k] * y[6 * fa + k] : 0.0);
      cam_geom_one(cam6 - 6 * c, p.geom_cand, c, p.fixed_slot);
    }
  }
  if (tid < 64) {
    double mcc = 0.0, st2 = 0.0, x2 = 0.0, gmax = 0.0, gn2 = 0.0, bad = 0.0;
    for (int i = tid; i < n; i += 64) {
      mcc += 0.5 * y[i] * gcs[i] + 0.5 * D2[i] * y[i] * y[i];
      const double d = sc[i] * y[i];
      st2 += d * d;
      gmax = fmax(gmax, fabs(gc[i]));
      gn2 += gc[i] * gc[i];
      if (!isfinite(y[i])) bad = 1.0;
    }
    for (int i = tid; i < 6 * p.n_frames; i += 64) {
      const int fa = p.geom[i / 6].free_index;
      if (fa < 0) continue;
      const double* live = p.sc + n + 6 * fa;     // written by this kernel at the first linearisation (or just above)
      if (live[0] + live[1] + live[2] + live[3] + live[4] + live[5] > 0.0) x2 += p.cams[i] * p.cams[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mcc += __shfl_xor(mcc, off); st2 += __shfl_xor(st2, off); x2 += __shfl_xor(x2, off);
      gn2 += __shfl_xor(gn2, off); gmax = fmax(gmax, __shfl_xor(gmax, off)); bad = fmax(bad, __shfl_xor(bad, off));
    }
    if (tid == 0) {
      const int TRI = tri_index(n + 1);
      p.scal[kMccCams] = mcc; p.scal[kStep2Cams] = st2; p.scal[kX2Cams] = x2;
      p.scal[kGmaxCams] = gmax; p.scal[kGnorm2Cams] = gn2;
      p.scal[kSolveOk] = (chol_ok && bad == 0.0) ? 1.0 : 0.0;
      p.scal[kCostLin] = packed_load<false>(p, TRI + 2 * n);
      p.scal[kGnorm2Pts] = packed_load<false>(p, TRI + 2 * n + 1);
    }
  }
}

__global__ __launch_bounds__(kSolveThreads) void k_solve_generic(SolveParams p_in) {
  SolveParams p = p_in;
  if (!solve_resolve(p)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = 6 * p.n_free;
  const int ld = n + 1;
  double* S = reinterpret_cast<double*>(smem);   // [n][ld]
  double* y = S + n * ld;
  double* sc = y + n;
  double* D2 = sc + n;
  double* gcs = D2 + n;
  double* gc = gcs + n;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid == 0) s_ok = 1;
  if (p.peer_world > 0 && !peer_wait_all(p.peer, p.peer_world, p.peer_flag, p.peer_seq, p.peer_timeout, p.peer_err, tid)) s_ok = 0;
  solve_prologue<kSolveThreads>(p, n, ld, S, y, sc, D2, gcs, gc, tid);
  const int tx = tid & 15, ty = tid >> 4;
  for (int j = 0; j < n; ++j) {
    const double d = S[j * ld + j];
    const double dj = (d > 0.0) ? sqrt(d) : 1.0;
    if (tid == 0 && (!(d > 0.0) || !isfinite(d))) s_ok = 0;
    __syncthreads();
    for (int r = j + tid; r < n; r += kSolveThreads) S[r * ld + j] = (r == j) ? dj : S[r * ld + j] / dj;
    __syncthreads();
    for (int r = j + 1 + ty; r < n; r += 16) {
      const double lrj = S[r * ld + j];
      for (int c = j + 1 + tx; c <= r; c += 16) S[r * ld + c] -= lrj * S[c * ld + j];
    }
    __syncthreads();
  }
  for (int j = 0; j < n; ++j) {
    if (tid == 0) y[j] /= S[j * ld + j];
    __syncthreads();
    const double yj = y[j];
    for (int r = j + 1 + tid; r < n; r += kSolveThreads) y[r] -= S[r * ld + j] * yj;
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {
    if (tid == 0) y[j] /= S[j * ld + j];
    __syncthreads();
    const double yj = y[j];
    for (int r = tid; r < j; r += kSolveThreads) y[r] -= S[j * ld + r] * yj;
    __syncthreads();
  }
  solve_epilogue<kSolveThreads>(p, n, y, sc, D2, gcs, gc, s_ok != 0, tid);
}

// =====================================================================================================
// Blocked dense factorisation S = L D L^T of the reduced camera system with LOOK-AHEAD (r4), one workgroup.
//
// Round 3 walked the n_free panels of six columns in lock step -- phase A (every row thread factorises the 6x6 diagonal
// block and solves its own row of the panel), barrier, phase B (rank-6 update of the whole trailing matrix), barrier -- and
// measured (profiles/r03/solve_phase_cycles.txt) 2.5 k cycles per panel at n = 42 and 2.9 k at n = 90, of which the
// trailing update is NOT on the dependency chain of the next panel except for that panel's own six columns.  Now:
//   * wave 0 (the PANEL wave, raised priority) owns the whole chain: it factorises panel k, and right after the barrier
//     that publishes L_k it applies panel k's update to the six columns of panel k + 1 itself (its threads keep their own
//     unscaled rows w = l d in registers; the six block rows of L_k come back through LDS), shares the updated diagonal
//     block inside the wave, and factorises panel k + 1 -- one thread per row (two rows per thread beyond 64 rows);
//   * waves 1.. (the UPDATE waves) meanwhile apply panel k to the rest of the trailing matrix (columns >= 6 (k + 2)): one
//     thread per (row, 6-column block), 36 FMAs, the same work items as round 3's phase B;
//   * ONE workgroup barrier per panel: behind it L_k / w_k are visible to the update waves and update k - 1 is complete
//     for the panel wave.  The unscaled rows are double-buffered (update k reads w_k while panel k + 1 writes w_{k+1}).
// The right-hand side rides along as row n of the augmented matrix [S y; y^T .] (its factor row is D^-1 L^-1 y: forward
// substitution and diagonal scaling for free).  Backward substitution L^T x = z is ONE wave with the solution vector in
// registers: per column one v_readlane pair + one FMA per lane, the rows of L prefetched six at a time (round 3: a
// barrier per panel, 6.0 k cycles at n = 42; this form is a chain of ~30 cycles per column).
// The candidate camera geometry is split: one lane per camera runs the scalar chain (angle, sine / cosine, reciprocal,
// R) for ALL cameras at once, 32 lanes per camera then form B and dR from LDS (round 3: every 32-lane group repeated the
// scalar chain and picked its entries with select chains, ~6 k cycles).
// =====================================================================================================
constexpr int kSolveBlockedThreads = 256;     // four waves (one panel + three update waves) up to kSolveNarrowFree free cameras,
constexpr int kSolveWideThreads = 512;        // eight beyond (the update of a 90 x 90 system has up to 559 work items per panel)
constexpr int kSolveNarrowFree = 8;
__host__ __device__ inline size_t solve_blocked_smem_bytes(int n) {
  // augmented matrix (n + 1)^2 | two buffers of unscaled panel rows (n + 1) x 6 | xs, sc, D2, gcs, gc, live (n + 1 each) | items
  const size_t N1 = (size_t)n + 1;
  const size_t items = (size_t)(n / 6) * N1;       // upper bound of the update work items
  return sizeof(double) * (N1 * N1 + 2 * 6 * N1 + 6 * N1 + 2) + sizeof(uint32_t) * items;
}

// One lane per camera: the scalar chain of cam_geom_one.  cg: [32] doubles of LDS: cam6 (6) | w (3) | ct, st | R (9) | theta2 | rodrigues
__device__ inline void cam_geom_scalar(const double cam6[6], double* cg) {
  const double wx = cam6[0], wy = cam6[1], wz = cam6[2];
  const double theta2 = wx * wx + wy * wy + wz * wz;
  const bool rod = theta2 > DBL_EPSILON;
#pragma unroll
  for (int k = 0; k < 6; ++k) cg[k] = cam6[k];
  double R[9], w3[3], ct, st;
  if (rod) {
    const double theta = sqrt(theta2);
    sincos_angle(theta, st, ct);
    const double ti = 1.0 / theta;
    const double ax = wx * ti, ay = wy * ti, az = wz * ti;
    const double oc = 1.0 - ct;
    w3[0] = ax; w3[1] = ay; w3[2] = az;
    R[0] = ct + ax * ax * oc;       R[1] = ax * ay * oc - az * st;  R[2] = ay * st + ax * az * oc;
    R[3] = az * st + ax * ay * oc;  R[4] = ct + ay * ay * oc;       R[5] = -ax * st + ay * az * oc;
    R[6] = -ay * st + ax * az * oc; R[7] = ax * st + ay * az * oc;  R[8] = ct + az * az * oc;
  } else {
    w3[0] = w3[1] = w3[2] = 0.0; ct = 1.0; st = 0.0;
    R[0] = 1; R[1] = -wz; R[2] = wy; R[3] = wz; R[4] = 1; R[5] = -wx; R[6] = -wy; R[7] = wx; R[8] = 1;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) cg[6 + k] = w3[k];
  cg[9] = ct; cg[10] = st;
#pragma unroll
  for (int k = 0; k < 9; ++k) cg[11 + k] = R[k];
  cg[20] = theta2;
  cg[21] = rod ? 1.0 : 0.0;
}

// 32 lanes per camera (sub = lane within the group, the group aligned to 32 lanes of a wave): B = (w w^T + (R^T - I) [w]x) /
// theta^2 by lanes 0..8, dR_k = R [B_k]x by lanes 0..26 (three cross-lane reads of B's column k), everything stored.
// Same formulas and operand order as cam_geom_one.
__device__ inline void cam_geom_finish(const double* cg, CamGeom* __restrict__ out, int c, int fixed_slot, int sub) {
  CamGeom& g = out[c];
  const bool rod = cg[21] != 0.0;
  if (sub < 3) { g.aa[sub] = cg[sub]; g.t[sub] = cg[3 + sub]; g.w[sub] = cg[6 + sub]; }
  if (sub == 3) {
    g.rodrigues = rod; g.is_free = (c != fixed_slot);
    g.free_index = (c == fixed_slot) ? -1 : (fixed_slot >= 0 && c > fixed_slot ? c - 1 : c);
    g.pad = 0;
  }
  if (sub == 4) { g.ct = cg[9]; g.st = cg[10]; }
  const double* R = cg + 11;
  const int e9 = sub < 9 ? sub : 8;
  if (sub < 9) g.R[sub] = R[e9];
  const int s27 = sub < 27 ? sub : 26;
  double dRv;
  if (rod) {
    const int ei = e9 / 3, ej = e9 - 3 * ei;
    // [w]x entry (q, ej) of 0 -wz wy | wz 0 -wx | -wy wx 0: zero on the diagonal, -w[other] one right of it (cyclically),
    // +w[other] two right of it; other = 3 - q - ej.  (cg[0..2] = the angle-axis vector; runtime indices: LDS reads)
    double acc = cg[ei] * cg[ej];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int d = (ej - q + 3) % 3;
      const double wo = cg[(d == 0) ? 0 : 3 - q - ej];
      const double wxqj = (d == 0) ? 0.0 : (d == 1 ? -wo : wo);
      acc += (R[3 * q + ei] - (ei == q ? 1.0 : 0.0)) * wxqj;
    }
    const double Bv = acc / cg[20];
    const int k = s27 / 9, ij = s27 - 9 * k, i = ij / 3, j = ij - 3 * i;
    const int lane0 = threadIdx.x & 32;                     // first lane of this camera's 32-lane group within the wave
    const double b0 = __shfl(Bv, lane0 + k), b1 = __shfl(Bv, lane0 + 3 + k), b2 = __shfl(Bv, lane0 + 6 + k);
    // column j of Bx = [0 -b2 b1; b2 0 -b0; -b1 b0 0]
    const double x0 = (j == 0) ? 0.0 : (j == 1 ? -b2 : b1);
    const double x1 = (j == 0) ? b2 : (j == 1 ? 0.0 : -b0);
    const double x2 = (j == 0) ? -b1 : (j == 1 ? b0 : 0.0);
    double a3 = 0.0;
    a3 += R[3 * i] * x0; a3 += R[3 * i + 1] * x1; a3 += R[3 * i + 2] * x2;
    dRv = a3;
  } else {
    const int k = s27 / 9, m = s27 - 9 * k;
    // [e_k]x, row-major: (k == 0) 0 0 0 | 0 0 -1 | 0 1 0;  (k == 1) 0 0 1 | 0 0 0 | -1 0 0;  (k == 2) 0 -1 0 | 1 0 0 | 0 0 0
    const int i = m / 3, j = m - 3 * i;
    dRv = 0.0;
    if (i != j && i != k && j != k) dRv = ((j - i + 3) % 3 == 1) ? -1.0 : 1.0;
  }
  if (sub < 27) g.dR[sub] = dRv;
}

// T threads: 256 (four waves) up to eight free cameras, 512 beyond
template <bool AGENT, int T>
__device__ __forceinline__ void solve_blocked(SolveParams& p, double* smem, int tid) {
  const int nf = p.n_free;
  const int n = 6 * nf;
  const int N1 = n + 1;
  const int ld = N1;                          // odd: consecutive rows start in different banks
  double* A = smem;                           // [N1][ld] lower triangle, rhs = row n
  double* Wp = A + (size_t)N1 * ld;           // [2][N1][6] the panels' rows, unscaled (w = l d), double-buffered
  double* xs = Wp + 2 * (size_t)N1 * 6;       // [N1] solution of the scaled system
  double* sc = xs + N1;                       // [N1] Jacobi scales, sc[n] = 1
  double* D2 = sc + N1;
  double* gcs = D2 + N1;
  double* gc = gcs + N1;
  double* live = gc + N1;                     // column-is-live flags
  uint32_t* items = reinterpret_cast<uint32_t*>(live + N1 + 1);   // update work items: row | block column << 16
  __shared__ int s_ok;
  __shared__ double s_cams[6 * kMaxFrames];     // current cameras and their free indices: requested with the first round trip
  __shared__ int s_free[kMaxFrames];
  __shared__ double s_cg[kMaxFrames][32];       // candidate geometry, scalar part per camera
  __shared__ double s_stat[3];                  // gradient max norm, squared gradient norm, |x|^2 of the cameras (known before the solve)
  __shared__ double s_dyn[3];                   // model cost change, squared step norm, non-finite flag (known after it)
  const int TRI = tri_index(N1);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the panel / update split is a uniform branch)
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PBA_TS(k) do { if (PBA_PHASE_TIMING) ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
  // sections of the panel loop (cycles summed over the panels): 0 loop top -> 1 diagonal block factorised -> 2 own rows solved and
  // stored -> 3 block rows of L requested -> 4 barrier passed -> 5 next panel's columns updated + published -> 6 diagonal block reloaded
  unsigned long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sec_t = 0;
#define PBA_SEC(k) do { if (PBA_PHASE_TIMING) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); if (k) sec[k] += tn - sec_t; sec_t = tn; } } while (0)
  PBA_TS(0);
  bool peer_ok = true;
  if (!AGENT && p.peer_world > 0) peer_ok = peer_wait_all(p.peer, p.peer_world, p.peer_flag, p.peer_seq, p.peer_timeout, p.peer_err, tid);
  // ---- prologue: every global load is issued before the first one is consumed (ONE round trip) ---------------------
  constexpr int kPer = (T > 256) ? 9 : 5;       // packed triangle entries per thread: TRI <= 1225 (T = 256), <= 4186 (T = 512)
  double val[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) { const int t = tid + u * T; val[u] = (t < TRI) ? packed_load<AGENT>(p, t) : 0.0; }
  const double cam_v = (tid < 6 * p.n_frames) ? p.cams[tid] : 0.0;
  const int free_v = (tid < p.n_frames) ? p.geom[tid].free_index : -1;
  static_assert(T >= 6 * kMaxFrames, "one thread per reduced-system row");
  double du = 0.0, g = 0.0, s_old = 1.0, live_old = 0.0, cost_lin = 0.0, gn2_pts = 0.0;
  if (tid < n) {
    du = packed_load<AGENT>(p, TRI + n + tid);
    g = packed_load<AGENT>(p, TRI + tid);
    if (!p.init_scale) { s_old = p.sc[tid]; live_old = p.sc[n + tid]; }
  }
  if (tid == 0) { cost_lin = packed_load<AGENT>(p, TRI + 2 * n); gn2_pts = packed_load<AGENT>(p, TRI + 2 * n + 1); }
  // (row, column) of the packed entries and the update work items: window-shape tables from the host, same round trip
  uint32_t rc[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) { const int t = tid + u * T; rc[u] = (t < TRI) ? p.tab[t] : 0u; }
  const int n_items = (nf > 1) ? solve_item_base(nf, N1) : 0;
  constexpr int kItemsPer = (T > 256) ? 2 : 1;       // n_items <= 185 at eight free cameras, <= 644 at fifteen
  uint32_t itv[kItemsPer];
#pragma unroll
  for (int u = 0; u < kItemsPer; ++u) { const int t = tid + u * T; itv[u] = (t < n_items) ? p.tab[TRI + t] : 0u; }
#pragma unroll
  for (int u = 0; u < kItemsPer; ++u) { const int t = tid + u * T; if (t < n_items) items[t] = itv[u]; }
  if (tid < 6 * p.n_frames) s_cams[tid] = cam_v;
  if (tid < p.n_frames) s_free[tid] = free_v;
  if (tid < n) {
    const int i = tid;
    double s, lv;
    if (p.init_scale) { s = p.jacobi ? 1.0 / (1.0 + sqrt(du)) : 1.0; lv = du > 0.0 ? 1.0 : 0.0; p.sc[i] = s; p.sc[n + i] = lv; }
    else { s = s_old; lv = live_old; }
    sc[i] = s;
    live[i] = lv;
    D2[i] = fmin(fmax(s * s * du, p.min_diag), p.max_diag) / p.radius;
    gc[i] = g;
    gcs[i] = s * g;
  }
  if (tid == n) sc[n] = 1.0;
  if (tid == 0) s_ok = peer_ok ? 1 : 0;
  __syncthreads();
  PBA_TS(1);
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int t = tid + u * T;
    if (t >= TRI) continue;
    const int r = (int)(rc[u] >> 16), c = (int)(rc[u] & 0xffffu);
    double v = sc[c] * val[u] * sc[r];          // (rounds 1-3 scaled the mirrored entry: column scale first)
    if (r == c) v = (r < n) ? v + D2[r] : 0.0;
    A[(size_t)r * ld + c] = v;
    if (r != c) A[(size_t)c * ld + r] = 0.0;      // strict upper triangle: zeros (the back-substitution reads whole rows unmasked)
  }
  __syncthreads();
  if (p.S_dbg) {
    for (int k = tid; k < n * n; k += T) { const int r = k / n, c = k - r * n; p.S_dbg[k] = (r >= c) ? A[(size_t)r * ld + c] : A[(size_t)c * ld + r]; }
    for (int i = tid; i < n; i += T) p.rhs_dbg[i] = A[(size_t)n * ld + i];
    __syncthreads();
  }
  PBA_TS(2);
  constexpr int ROWS = (T > 256) ? 2 : 1;       // rows per thread of the panel wave
  constexpr int NB = T - 64;                    // update threads
  // scalars that do not depend on the solution (gradient norms of the camera part, |x|^2 of the live cameras): the last
  // update wave, off the dependency chain (round 3 reduced six values through a six-level butterfly after the substitution:
  // 4 k cycles of the epilogue)
  if (wave == T / 64 - 1) {
    double q2[2] = {0.0, 0.0};
    double gmax = 0.0;
    for (int i = lane; i < n; i += 64) { gmax = fmax(gmax, fabs(gc[i])); q2[0] += gc[i] * gc[i]; }
    for (int i = lane; i < 6 * p.n_frames; i += 64) {
      const int fa = s_free[i / 6];
      if (fa < 0) continue;
      const double* lv = live + 6 * fa;
      if (lv[0] + lv[1] + lv[2] + lv[3] + lv[4] + lv[5] > 0.0) q2[1] += s_cams[i] * s_cams[i];
    }
    wave_sum_n<2>(q2);
    gmax = wave_max(gmax);
    if (lane == 0) { s_stat[0] = gmax; s_stat[1] = q2[0]; s_stat[2] = q2[1]; }
  }
  if (!(p.final_pass && !p.init_scale)) {
    // ---- factorisation S = L D L^T (unit lower L in A, the rhs row ends up as D^-1 L^-1 y) ---------------------------
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);
      double a[ROWS][6], w[ROWS][6];
      double Wd[21];                            // diagonal block of the current panel, lower triangle packed i (i + 1) / 2 + m
#pragma unroll
      for (int q = 0; q < ROWS; ++q) {
        const int r = lane + 64 * q;
#pragma unroll
        for (int m = 0; m < 6; ++m) { a[q][m] = A[(size_t)(r <= n ? r : n) * ld + m]; w[q][m] = 0.0; }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int m = 0; m <= i; ++m) Wd[i * (i + 1) / 2 + m] = A[(size_t)i * ld + m];
      for (int k = 0; k < nf; ++k) {
        const int c0 = 6 * k;
        double Lt[21], rd[6];
        bool pd = true;
        PBA_SEC(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          // column j of the block: w_ij = a_ij - sum_{m<j} w_im l_jm (i >= j), d_j = w_jj, l_ij = w_ij / d_j
#pragma unroll
          for (int i = j; i < 6; ++i) {
            double v = Wd[i * (i + 1) / 2 + j];
#pragma unroll
            for (int m = 0; m < j; ++m) v = fma(-Wd[i * (i + 1) / 2 + m], Lt[j * (j + 1) / 2 + m], v);
            Wd[i * (i + 1) / 2 + j] = v;
          }
          const double d = Wd[j * (j + 1) / 2 + j];
          const bool ok = (d > 0.0) && isfinite(d);
          pd = pd && ok;
          rd[j] = fast_rcp(ok ? d : 1.0);
#pragma unroll
          for (int i = j + 1; i < 6; ++i) Lt[i * (i + 1) / 2 + j] = Wd[i * (i + 1) / 2 + j] * rd[j];
        }
        PBA_SEC(1);
        // own rows of the panel: w_j = a_j - sum_{m<j} w_m l_jm, l_j = w_j / d_j (for the six rows of the diagonal block this
        // repeats the recurrence above: l = 1 on the diagonal; entries right of it are never read)
        double* Wk = Wp + (size_t)(k & 1) * N1 * 6;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
          const int r = lane + 64 * q;
          double l[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            double v = a[q][j];
#pragma unroll
            for (int m = 0; m < j; ++m) v = fma(-w[q][m], Lt[j * (j + 1) / 2 + m], v);
            w[q][j] = v;
            l[j] = v * rd[j];
          }
          if (r >= c0 && r <= n) {
#pragma unroll
            for (int m = 0; m < 6; ++m) { A[(size_t)r * ld + c0 + m] = l[m]; Wk[(size_t)r * 6 + m] = w[q][m]; }
            if (r == c0 && !pd) s_ok = 0;
          }
        }
        PBA_SEC(2);
        // the six block rows of L_k that panel k + 1 needs were written by THIS wave: requested ahead of the barrier
        double Lb[36];
        const bool more = k + 1 < nf;
        if (more) {
          wave_lds_sync();
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) Lb[6 * e + m] = A[(size_t)(c0 + 6 + e) * ld + c0 + m];
        }
        PBA_SEC(3);
        __syncthreads();           // L_k, w_k visible to the update waves; update k - 1 complete
        PBA_SEC(4);
        if (!more) break;
        // panel k's update of the six columns of panel k + 1, own rows
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
          const int r = lane + 64 * q;
          const bool act = r >= c0 + 6 && r <= n;
          // (rows outside the trailing matrix read a valid address and carry on with finite garbage that is never stored: no
          // exec-mask branch per load on the chain)
          const double* arow = A + (size_t)(r <= n ? r : n) * ld + c0 + 6;
          double an[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) an[e] = arow[e];
#pragma unroll
          for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int e = 0; e < 6; ++e) an[e] = fma(-w[q][m], Lb[6 * e + m], an[e]);
          if (act && r < c0 + 12) {      // the six rows of the next diagonal block publish theirs (entries right of the diagonal: unused)
#pragma unroll
            for (int e = 0; e < 6; ++e) A[(size_t)r * ld + c0 + 6 + e] = an[e];
          }
#pragma unroll
          for (int e = 0; e < 6; ++e) a[q][e] = an[e];
        }
        PBA_SEC(5);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int m = 0; m <= i; ++m) Wd[i * (i + 1) / 2 + m] = A[(size_t)(c0 + 6 + i) * ld + c0 + 6 + m];
        if (PBA_PHASE_TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        PBA_SEC(6);
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
      const int qb = tid - 64;
      for (int k = 0; k < nf; ++k) {
        __syncthreads();
        if (k + 2 >= nf) continue;
        const int c0 = 6 * k;
        const double* Wk = Wp + (size_t)(k & 1) * N1 * 6;
        for (int it = solve_item_base(k + 2, N1) + qb; it < n_items; it += NB) {
          const uint32_t wd = items[it];
          const int r = (int)(wd & 0xffffu), j = (int)(wd >> 16);
          double wr[6], lc[36], acc[6];
#pragma unroll
          for (int m = 0; m < 6; ++m) wr[m] = Wk[(size_t)r * 6 + m];
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) lc[6 * e + m] = A[(size_t)(6 * j + e) * ld + c0 + m];
#pragma unroll
          for (int e = 0; e < 6; ++e) acc[e] = A[(size_t)r * ld + 6 * j + e];
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int m = 0; m < 6; ++m) acc[e] = fma(-wr[m], lc[6 * e + m], acc[e]);
#pragma unroll
          for (int e = 0; e < 6; ++e) A[(size_t)r * ld + 6 * j + e] = acc[e];
        }
      }
    }
    // the panel wave stored its six-column rows whole: inside a diagonal block the entries on and right of the diagonal (l_jj = 1
    // and unused values) become zeros, so that row j of A is L[j][i] for i < j and 0 beyond -- the masks of the substitution's loads
    // (a compare and two selects per value on the one wave everybody waits for) are paid here, once, by all threads
    for (int t = tid; t < n; t += T) {
      const int c0 = 6 * (t / 6);
#pragma unroll
      for (int m = 0; m < 6; ++m) if (c0 + m >= t) A[(size_t)t * ld + c0 + m] = 0.0;
    }
    __syncthreads();
    PBA_TS(3);
    // ---- backward substitution L^T x = z (z = row n = D^-1 L^-1 y, L unit lower): the panel wave alone ----------------
    // Solution vector in registers (lane i <-> unknown i, two per lane beyond 64); column j: x_j = z_j of lane j (one
    // v_readlane pair), z_i -= L[j][i] x_j for i < j.  The rows of L come in chunks of six straight out of A (zero for i >= j, see
    // above: nothing but the FMA on the dependency chain) and are double-buffered in two register sets (no copies, so the wait
    // for a chunk sits one chunk after its loads).
    if (wave == 0) {
      double z[ROWS];
#pragma unroll
      for (int q = 0; q < ROWS; ++q) { const int i = lane + 64 * q; z[q] = (i < n) ? A[(size_t)n * ld + i] : 0.0; }
      const double* Arow = A + (lane < n ? lane : 0);      // (lanes beyond the system -- and 64 + lane > n -- read in-bounds junk into unknowns nobody uses)
      auto load_chunk = [&](int jb, double (&L6)[6][ROWS]) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int q = 0; q < ROWS; ++q) L6[u][q] = Arow[(size_t)(jb + u) * ld + 64 * q];      // (zero on and right of the diagonal)
      };
      auto apply_chunk = [&](int jb, const double (&L6)[6][ROWS]) {
#pragma unroll
        for (int u = 5; u >= 0; --u) {
          const int j = jb + u;
          const double zsrc = (ROWS == 2 && j >= 64) ? z[ROWS - 1] : z[0];
          const double xj = readlane_f64(zsrc, j & 63);
#pragma unroll
          for (int q = 0; q < ROWS; ++q) z[q] = fma(-L6[u][q], xj, z[q]);
        }
      };
      double LA[6][ROWS], LB[6][ROWS];
      int jb = n - 6;
      load_chunk(jb, LA);
      for (;;) {
        if (jb >= 6) load_chunk(jb - 6, LB);
        apply_chunk(jb, LA);
        jb -= 6;
        if (jb < 0) break;
        if (jb >= 6) load_chunk(jb - 6, LA);
        apply_chunk(jb, LB);
        jb -= 6;
        if (jb < 0) break;
      }
      // the step's own scalars straight from the registers: model cost change and squared norm of the camera step
      double q2[2] = {0.0, 0.0};
      bool bad = false;
#pragma unroll
      for (int q = 0; q < ROWS; ++q) {
        const int i = lane + 64 * q;
        if (i < n) {
          xs[i] = z[q];
          q2[0] += 0.5 * z[q] * gcs[i] + 0.5 * D2[i] * z[q] * z[q];
          const double d = sc[i] * z[q];
          q2[1] += d * d;
          bad = bad || !isfinite(z[q]);
        }
      }
      wave_sum_n<2>(q2);
      const bool any_bad = __any(bad ? 1 : 0) != 0;
      if (lane == 0) { s_dyn[0] = q2[0]; s_dyn[1] = q2[1]; s_dyn[2] = any_bad ? 1.0 : 0.0; }
    }
    __syncthreads();
  } else {
    for (int i = tid; i < n; i += T) xs[i] = 0.0;
    if (tid == 0) { s_dyn[0] = 0.0; s_dyn[1] = 0.0; s_dyn[2] = 0.0; }
    __syncthreads();
  }
  PBA_TS(4);
  // ---- epilogue (camera step, candidate cameras + geometry, replicated scalars) -----------------------------------
  const bool chol_ok = s_ok != 0;
  const bool poisoned = !AGENT && p.peer_world > 0 && !peer_ok;      // a peer never showed up: nothing below may look like a step
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  if (tid < 6 * p.n_frames && !(p.final_pass && !p.init_scale)) {
    const int slot = tid / 6, k = tid % 6;
    const int fa = s_free[slot];
    double d = 0.0;
    if (fa >= 0) d = -sc[6 * fa + k] * xs[6 * fa + k];
    if (poisoned) d = qnan;
    p.delta_c[tid] = d;
    p.cams_cand[tid] = s_cams[tid] + d;
  }
  if (p.geom_cand) {
    // scalar chain of every camera at once, one lane each (wave 1; wave 0 reduces the scalars meanwhile)
    if (wave == 1 && lane < p.n_frames) {
      const int c = lane;
      double cam6[6];
      const int fa = s_free[c];
#pragma unroll
      for (int k = 0; k < 6; ++k) cam6[k] = s_cams[6 * c + k] + (fa >= 0 ? -sc[6 * fa + k] * xs[6 * fa + k] : 0.0);
      cam_geom_scalar(cam6, s_cg[c]);
    }
  }
  if (tid == 0) {
    p.scal[kMccCams] = s_dyn[0]; p.scal[kStep2Cams] = s_dyn[1]; p.scal[kX2Cams] = s_stat[2];
    p.scal[kGmaxCams] = s_stat[0]; p.scal[kGnorm2Cams] = s_stat[1];
    p.scal[kSolveOk] = (chol_ok && s_dyn[2] == 0.0 && !poisoned) ? 1.0 : 0.0;
    p.scal[kCostLin] = poisoned ? qnan : cost_lin;
    p.scal[kGnorm2Pts] = gn2_pts;
  }
  PBA_TS(6);
  if (p.geom_cand) {
    __syncthreads();
    for (int c = tid / 32; c < p.n_frames; c += T / 32) cam_geom_finish(s_cg[c], p.geom_cand, c, p.fixed_slot, tid & 31);
  }
  PBA_TS(5);
  if (PBA_PHASE_TIMING && p.dbg && tid == 0)
    printf("solve_blocked n %d cycles: prologue %llu (loads + tables %llu, scatter %llu) factorisation %llu [block %llu rows %llu Lb %llu barrier %llu update %llu reload %llu] substitution %llu epilogue %llu (scalars %llu)\n",
           n, ts[2] - ts[0], ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], sec[1], sec[2], sec[3], sec[4], sec[5], sec[6], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[4]);
#undef PBA_TS
#undef PBA_SEC
}

// The reduced solve as its own launch: multi-rank steps (the exchange of the packed sums sits between the reduction
// and the solve) and the PBA_FUSE_SOLVE=0 diagnostics path.
template <int T>
__global__ __launch_bounds__(T) void k_solve_blocked(SolveParams p_in) {
  SolveParams p = p_in;
  if (!solve_resolve(p)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  solve_blocked<false, T>(p, reinterpret_cast<double*>(smem), threadIdx.x);
}

// Reduction of the per-workgroup Schur partials as its own launch (multi-rank steps).  peer_flag >= 0: `packed` is this
// rank's peer-exchange mailbox slot (system-scope write-through stores) and the LAST workgroup to finish raises the rank's
// flag for exchange `peer_seq` once every workgroup's stores have left (ticket): the consumer (k_solve_blocked's prologue)
// waits for all ranks' flags, so no exchange kernel sits in between.
struct ReduceFinalParams {
  ReduceParams rp;
  const LmState* lm; int32_t enq_cur, final_pass;
  const double* block_cost_alt; const int32_t* block_fail_alt;
  int32_t sys_stores;
  unsigned int* ticket;            // zero between launches (peer_flag >= 0 only)
  double* peer_own; int32_t peer_flag; unsigned long long peer_seq;
};
__global__ __launch_bounds__(kReduceThreads) void k_reduce_final(ReduceFinalParams fp) {
  ReduceParams rp = fp.rp;
  const bool skip = fp.lm && ((fp.lm->done && !fp.final_pass) || (fp.final_pass && !lm_final_pass_needed(fp.lm)));
  __shared__ double s_red[kReduceThreads / kReduceEntries][kReduceEntries + 1];
  __shared__ int s_f[16];
  if (!skip) {
    if (fp.lm && fp.lm->cur != fp.enq_cur) { rp.block_cost = fp.block_cost_alt; rp.block_fail = fp.block_fail_alt; }
    if (fp.sys_stores) reduce_partials<2>(rp, s_red, s_f, (int)blockIdx.x, (int)gridDim.x);
    else reduce_partials<0>(rp, s_red, s_f, (int)blockIdx.x, (int)gridDim.x);
  }
  if (fp.peer_flag >= 0) {
    // (a terminated solve still raises the flag: the peers' consumers are no-ops too, but the sequence stays in step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      // the other workgroups' mailbox-slot stores are ordered before the flag by the ticket itself (release on the way in, acquire
      // in the workgroup that raises the flag with a system-scope release), not only by the system-scope write-through stores +
      // s_waitcnt above: the consumer may be another device
      const unsigned t = __hip_atomic_fetch_add(fp.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x - 1) {
        *fp.ticket = 0;
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fp.peer_own) + fp.peer_flag, fp.peer_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Reduction of the per-workgroup Schur partials AND the reduced solve in ONE launch (single-rank steps): the workgroups
// reduce kReduceEntries packed entries each, publish them with agent-scope (write-through) stores and take a ticket; the
// LAST workgroup to arrive keeps its first four waves (all eight beyond eight free cameras) and runs the blocked solve
// on the packed sums it reads back with agent-scope loads.
struct ReduceSolveParams {
  ReduceParams rp;
  const double* block_cost_alt; const int32_t* block_fail_alt;
  unsigned int* ticket;          // zero between launches
  unsigned long long* stamp;     // null, or the device time-stamp block (kStamp*)
  SolveParams so;                // lm / enq_cur / final_pass of the step live here
  // End of a single-rank solve folded into the final (gradient-only) pass: host_seq != null makes the last workgroup take the
  // gradient-only decision (k_decide) and flush log, state, scalars and sequence number to the host mirror (k_flush) itself --
  // two launches less at the end of every solve.  A pass that turns out not to be needed (lm_final_pass_needed) only flushes.
  struct Fin {
    LmState* lm; pba_iteration_summary* log; pba_iteration_summary* host_log; int32_t max_log;
    LmState* host_state; double* host_scal; unsigned long long* host_seq; unsigned long long seq;
  } fin;
};

// log -> host log, then state, scalars and the sequence number (k_flush's body; nthreads threads of one workgroup)
__device__ inline void flush_to_host(const LmState* lm, LmState* host_state, const double* scal, double* host_scal, const pba_iteration_summary* log,
                                     pba_iteration_summary* host_log, int max_log, unsigned long long* host_seq, unsigned long long seq, int tid, int nthreads) {
  const int n_words = (lm->n_log < max_log ? lm->n_log : max_log) * (int)(sizeof(pba_iteration_summary) / 4);
  static_assert(sizeof(pba_iteration_summary) % 4 == 0, "word copy");
  const unsigned* src = reinterpret_cast<const unsigned*>(log);
  for (int k = tid; k < n_words; k += nthreads) store_system_u32(reinterpret_cast<unsigned*>(host_log) + k, src[k]);
  lm_publish(lm, host_state, scal, host_scal, host_seq, seq, tid, nthreads);
}

__global__ __launch_bounds__(kReduceThreads) void k_reduce_solve(ReduceSolveParams rsp) {
  const LmState* lm = rsp.so.lm;
  ReduceParams rp = rsp.rp;
  const bool fin_mode = rsp.fin.host_seq != nullptr;
  if (lm) {
    if (lm->done && !rsp.so.final_pass) return;
    if (rsp.so.final_pass && !lm_final_pass_needed(lm)) {
      if (fin_mode && blockIdx.x == 0)
        flush_to_host(lm, rsp.fin.host_state, rsp.so.scal, rsp.fin.host_scal, rsp.fin.log, rsp.fin.host_log, rsp.fin.max_log, rsp.fin.host_seq,
                      rsp.fin.seq, threadIdx.x, kReduceThreads);
      return;
    }
    if (lm->cur != rsp.so.enq_cur) { rp.block_cost = rsp.block_cost_alt; rp.block_fail = rsp.block_fail_alt; }
  }
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];      // the solve's matrix (last workgroup only)
  const unsigned long long t_k0 = PBA_PHASE_TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;
  __shared__ double s_red[kReduceThreads / kReduceEntries][kReduceEntries + 1];
  __shared__ int s_f[16];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  reduce_partials<1>(rp, s_red, s_f, (int)blockIdx.x, (int)gridDim.x);
  // ---- ticket: the last workgroup to arrive solves ----------------------------------------------------------------
  // every storing thread waits until its write-through stores have left the CU, then the workgroup takes its ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(rsp.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  const bool wide = rsp.so.n_free > kSolveNarrowFree;
  static_assert(kSolveWideThreads == kReduceThreads, "the wide solve uses the whole workgroup");
  if (!wide && tid >= kSolveBlockedThreads) return;   // four of the eight waves leave; barriers below count the remaining four
  if (tid == 0) *rsp.ticket = 0;
  const unsigned long long t_k1 = PBA_PHASE_TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;
  SolveParams so = rsp.so;
  if (so.lm) {
    if (so.lm->cur != so.enq_cur) {
      so.cams = so.cams_alt; so.cams_cand = so.cams_cand_alt; so.geom = so.geom_alt;
      if (so.geom_cand) so.geom_cand = so.geom_cand_alt;
    }
    so.radius = so.lm->radius;
  }
  so.packed = rp.packed;
  if (wide) solve_blocked<true, kSolveWideThreads>(so, reinterpret_cast<double*>(dyn_smem), tid);
  else solve_blocked<true, kSolveBlockedThreads>(so, reinterpret_cast<double*>(dyn_smem), tid);
  if (fin_mode) {
    // gradient-only decision + flush by this (last) workgroup: what k_decide and k_flush did as two more launches
    const int nth = wide ? kSolveWideThreads : kSolveBlockedThreads;
    __syncthreads();                                  // the epilogue's scalars have left (the barrier drains the stores)
    if (tid == 0) {
      double sl[kNumScal];
      for (int k = 0; k < kNumScal; ++k) sl[k] = load_agent(so.scal + k);     // (max |g_p| came from another workgroup of this launch)
      lm_decide(rsp.fin.lm, sl, rsp.fin.log, rsp.fin.max_log, 1);
      if (rsp.fin.lm->done && rsp.fin.lm->done_seq == 0) rsp.fin.lm->done_seq = rsp.fin.seq;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // thread 0's state / log stores (write-through) are read by the whole workgroup
    flush_to_host(rsp.fin.lm, rsp.fin.host_state, so.scal, rsp.fin.host_scal, rsp.fin.log, rsp.fin.host_log, rsp.fin.max_log, rsp.fin.host_seq,
                  rsp.fin.seq, tid, nth);
  }
  if (rsp.stamp && tid == 0) rsp.stamp[kStampEndSolve] = __builtin_amdgcn_s_memrealtime();
  if (PBA_PHASE_TIMING && rsp.so.dbg && tid == 0)
    printf("k_reduce_solve: last workgroup %d of %d reached the solve %.2f us after its own start, finished it %.2f us later\n", (int)blockIdx.x,
           (int)gridDim.x, 0.01 * (double)(t_k1 - t_k0), 0.01 * (double)(__builtin_amdgcn_s_memrealtime() - t_k1));
}

}  // namespace pba
