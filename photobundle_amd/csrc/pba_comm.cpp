// pba_comm.cpp -- see pba_comm.h
#include "pba_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

namespace pba {

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));   // optional
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));   // optional
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString) { err = "librccl misses a required symbol"; return false; }
    return true;
  }
};
Rccl& rccl() { static Rccl r; return r; }
}  // namespace

int Comm::unique_id(void* id128) {
  Rccl& r = rccl();
  if (!r.load()) return 1;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return 1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, 128);
  return 0;
}

int Comm::init_rccl(const void* id128, int rank_, int world_) {
  shutdown();
  Rccl& r = rccl();
  if (!r.load()) { err = r.err; return 1; }
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclComm_t c = nullptr;
  const ncclResult_t rc = r.CommInitRank(&c, world_, id, rank_);
  if (rc != ncclSuccess) { err = std::string("ncclCommInitRank: ") + r.GetErrorString(rc); return 1; }
  if (hipMalloc(reinterpret_cast<void**>(&d_small), 64 * sizeof(double)) != hipSuccess) { err = "hipMalloc(d_small)"; return 1; }
  nccl_comm = c; world = world_; rank = rank_; kind = 1;
  if (const char* f = getenv("PBA_FORCE_MULTI")) force = atoi(f) != 0;
  return 0;
}

int Comm::init_callback(pba_allreduce_fn f, void* c, int rank_, int world_) {
  shutdown();
  fn = f; ctx = c; world = world_; rank = rank_; kind = 2;
  if (const char* fm = getenv("PBA_FORCE_MULTI")) force = atoi(fm) != 0;
  return 0;
}

int Comm::allreduce_device(double* d, size_t n, int op, hipStream_t s) {
  if (!multi() || n == 0) return 0;
  if (kind == 1) {
    Rccl& r = rccl();
    const ncclResult_t rc = r.AllReduce(d, d, n, ncclDouble, op == 0 ? ncclSum : ncclMax, static_cast<ncclComm_t>(nccl_comm), s);
    if (rc != ncclSuccess) { err = std::string("ncclAllReduce: ") + r.GetErrorString(rc); return 1; }
    return 0;
  }
  if (kind == 2) {
    if (n > stage_cap) {
      if (h_stage) (void)hipHostFree(h_stage);
      h_stage = nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&h_stage), n * sizeof(double)) != hipSuccess) { err = "hipHostMalloc(stage)"; return 1; }
      stage_cap = n;
    }
    if (hipMemcpyAsync(h_stage, d, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { err = "stage D2H"; return 1; }
    if (fn(h_stage, (int64_t)n, op, ctx) != 0) { err = "all-reduce callback failed"; return 1; }
    if (hipMemcpyAsync(d, h_stage, n * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) { err = "stage H2D"; return 1; }
    return 0;
  }
  err = "no transport initialised";
  return 1;
}

int Comm::allreduce_host(double* h, int n, int op) {
  if (!multi() || n <= 0) return 0;
  if (n > 64) { err = "allreduce_host: n > 64"; return 1; }
  if (kind == 2) {
    if (fn(h, n, op, ctx) != 0) { err = "all-reduce callback failed"; return 1; }
    return 0;
  }
  if (kind == 1) {
    if (hipMemcpyAsync(d_small, h, n * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) { err = "H2D"; return 1; }
    if (allreduce_device(d_small, (size_t)n, op, stream)) return 1;
    if (hipMemcpyAsync(h, d_small, n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { err = "D2H"; return 1; }
    return 0;
  }
  err = "no transport initialised";
  return 1;
}

int Comm::rank_count() const {
  if (kind == 1 && nccl_comm && rccl().CommCount) {
    int n = 0;
    if (rccl().CommCount(static_cast<ncclComm_t>(nccl_comm), &n) == ncclSuccess) return n;
  }
  return world;
}

int Comm::enable_peer() {
  if (kind == 0 || world < 1 || world > kMaxPeers) { err = "peer exchange needs an initialised transport and <= 8 ranks"; return 1; }
  if (peer) return 0;
  // 1. own mailbox: fine-grained device memory (stores reach memory without waiting for the end of a kernel, remote
  //    reads never see a stale L2 line), zeroed, exported
  int ok = 1;
  hipIpcMemHandle_t mine;
  std::memset(&mine, 0, sizeof(mine));
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&mb_own), kMailboxDoubles * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError(); mb_own = nullptr; ok = 0;
  }
  if (ok && (hipMemset(mb_own, 0, kMailboxDoubles * sizeof(double)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) ok = 0;
  if (ok && hipIpcGetMemHandle(&mine, mb_own) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
  // 2. all-gather of the handles: one byte per double (exact under summation), 64 doubles = one rank's handle per call
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "one handle per allreduce_host call");
  std::vector<hipIpcMemHandle_t> all(world);
  for (int q = 0; q < world; ++q) {
    double buf[64];
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&mine);
    for (int i = 0; i < 64; ++i) buf[i] = (q == rank && ok) ? (double)b[i] : 0.0;
    if (allreduce_host(buf, 64, 0)) { ok = 0; break; }
    unsigned char* o = reinterpret_cast<unsigned char*>(&all[q]);
    for (int i = 0; i < 64; ++i) o[i] = (unsigned char)buf[i];
  }
  // 3. map the peers
  if (ok) {
    for (int q = 0; q < world && ok; ++q) {
      if (q == rank) { mb_peer[q] = mb_own; continue; }
      void* ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, all[q], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = 0; break; }
      mb_peer[q] = static_cast<double*>(ptr);
    }
  }
  // 4. unanimous or not at all
  double flag = ok ? 0.0 : 1.0;
  if (allreduce_host(&flag, 1, 1)) flag = 1.0;
  if (flag != 0.0) {
    close_peer();
    err = "peer exchange unavailable (fine-grained allocation or IPC mapping failed on some rank): staying on the base transport";
    return 1;
  }
  peer = true; seq_a = seq_b = 0;
  return 0;
}

void Comm::close_peer() {
  for (int q = 0; q < kMaxPeers; ++q) {
    if (mb_peer[q] && mb_peer[q] != mb_own) (void)hipIpcCloseMemHandle(mb_peer[q]);
    mb_peer[q] = nullptr;
  }
  if (mb_own) { (void)hipFree(mb_own); mb_own = nullptr; }
  peer = false;
}

void Comm::shutdown(bool abort) {
  if (abort) {
    // the stream holds work that will never finish: abort the communicator FIRST and leak everything a free would have to
    // synchronise for (hipFree / hipIpcCloseMemHandle wait for the device, and peers may still be polling this mailbox)
    if (kind == 1 && nccl_comm && rccl().CommAbort) rccl().CommAbort(static_cast<ncclComm_t>(nccl_comm));
    nccl_comm = nullptr;
    for (int q = 0; q < kMaxPeers; ++q) mb_peer[q] = nullptr;
    mb_own = nullptr; peer = false; h_stage = nullptr; stage_cap = 0; d_small = nullptr;
    kind = 0; world = 1; rank = 0; fn = nullptr; ctx = nullptr; force = false;
    return;
  }
  close_peer();
  if (kind == 1 && nccl_comm) {
    // a communicator with a collective that will never complete must be aborted: ncclCommDestroy would wait for it
    if (abort && rccl().CommAbort) rccl().CommAbort(static_cast<ncclComm_t>(nccl_comm));
    else if (!abort) rccl().CommDestroy(static_cast<ncclComm_t>(nccl_comm));
    nccl_comm = nullptr;
  }
  if (h_stage) { (void)hipHostFree(h_stage); h_stage = nullptr; stage_cap = 0; }
  if (d_small) { (void)hipFree(d_small); d_small = nullptr; }
  kind = 0; world = 1; rank = 0; fn = nullptr; ctx = nullptr; force = false;
}

}  // namespace pba
