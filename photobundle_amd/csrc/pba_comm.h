// pba_comm.h -- collective transports for the point-sharded multi-GPU path.
//   RCCL     : device-direct ncclAllReduce over xGMI (one process per GPU).  librccl is dlopen'ed so that the
//              engine loads on boxes / processes that never go multi-rank, and so that a process that already
//              carries a librccl (torch) shares that copy.
//   callback : host-staged all-reduce supplied by the caller (gloo / MPI / tests).
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
