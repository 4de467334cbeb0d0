// pba_comm.h -- collective transports for the point-sharded multi-GPU path.
//   RCCL     : device-direct ncclAllReduce over xGMI (one process per GPU).  librccl is dlopen'ed so that the
//              engine loads on boxes / processes that never go multi-rank, and so that a process that already
//              carries a librccl (torch) shares that copy.
//   callback : host-staged all-reduce supplied by the caller (gloo / MPI / tests).
//   peer     : (on top of either) the two per-step exchanges as flag-and-slot reads of peer-mapped mailboxes
//              (hipIpcGetMemHandle / hipIpcOpenMemHandle, fine-grained device memory) inside small kernels on the
//              engine's stream: no collective launch per step.  The base transport bootstraps it (the IPC handles
//              travel through its host all-reduce) and stays the fallback.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/pba.h"

namespace pba {

struct Comm {
  int world = 1, rank = 0;
  int kind = 0;   // 0 none, 1 rccl, 2 callback
  bool force = false;   // PBA_FORCE_MULTI=1: run the multi-rank code path (and the collectives) even at world == 1
  std::string err;
  hipStream_t stream = nullptr;

  // rccl
  void* nccl_comm = nullptr;
  // callback
  pba_allreduce_fn fn = nullptr;
  void* ctx = nullptr;
  double* h_stage = nullptr;   // pinned
  size_t stage_cap = 0;
  double* d_small = nullptr;   // 64 doubles, for host-scalar reductions over RCCL

  // ---- peer exchange (optional) ------------------------------------------------------------------------------------
  // Mailbox of every rank (doubles): [0, kFlagDoubles) flags (u64: A slot 0, A slot 1, B slot 0, B slot 1) |
  // A: 2 x kCapA  (the packed reduced camera system) | B: 2 x kCapB (the step scalars).  A rank writes only its OWN
  // mailbox and reads everybody's; slot = sequence number & 1 (a slot is rewritten two exchanges later, when every peer
  // has provably finished reading it: it raised its flag for the exchange in between).
  static constexpr int kMaxPeers = 8;
  static constexpr size_t kFlagDoubles = 16, kCapA = 4608, kCapB = 64;
  static constexpr size_t kMailboxDoubles = kFlagDoubles + 2 * kCapA + 2 * kCapB;
  bool peer = false;
  double* mb_own = nullptr;
  double* mb_peer[kMaxPeers] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long seq_a = 0, seq_b = 0;      // exchanges enqueued so far (identical on every rank)
  // all-gathers the IPC handles through allreduce_host, maps the peers; every rank ends with the same answer
  int enable_peer();
  int rank_count() const;     // what the transport itself reports (ncclCommCount), else `world`
  void close_peer();
  static size_t flag_index(int kind, unsigned long long seq) { return (size_t)(2 * kind + (int)(seq & 1)); }
  static size_t data_offset(int kind, unsigned long long seq) { return kFlagDoubles + (kind == 0 ? (seq & 1) * kCapA : 2 * kCapA + (seq & 1) * kCapB); }

  static int unique_id(void* id128);
  int init_rccl(const void* id128, int rank, int world);
  int init_callback(pba_allreduce_fn fn, void* ctx, int rank, int world);
  // in-place all-reduce of n doubles in device memory, ordered on `s`; op 0 = sum, 1 = max
  int allreduce_device(double* d, size_t n, int op, hipStream_t s);
  // in-place all-reduce of n (<= 64) host doubles
  int allreduce_host(double* h, int n, int op);
  bool multi() const { return world > 1 || (force && kind != 0); }
  // abort: the communicator holds a collective that never completed (ncclCommAbort instead of ncclCommDestroy)
  void shutdown(bool abort = false);
};

}  // namespace pba
