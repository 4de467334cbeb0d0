// comm_probe.cpp -- test harness: the product's collective transport (photobundle_amd/csrc/pba_comm.cpp, compiled into
// this library unchanged) behind a few C entry points, so that its host-staged callback path can be exercised on a box
// without a GPU by world-size-2 gloo processes (tests/test_comm_cpu.py).  Not part of the product.
#include "../../photobundle_amd/csrc/pba_comm.h"

extern "C" {
void* probe_comm_create() { return new pba::Comm(); }
void probe_comm_destroy(void* c) { auto* p = static_cast<pba::Comm*>(c); p->shutdown(); delete p; }
int probe_comm_init_callback(void* c, pba_allreduce_fn fn, void* ctx, int rank, int world) {
  return static_cast<pba::Comm*>(c)->init_callback(fn, ctx, rank, world);
}
int probe_comm_allreduce_host(void* c, double* v, int n, int op) { return static_cast<pba::Comm*>(c)->allreduce_host(v, n, op); }
int probe_comm_multi(void* c) { return static_cast<pba::Comm*>(c)->multi() ? 1 : 0; }
int probe_comm_world(void* c) { return static_cast<pba::Comm*>(c)->world; }
const char* probe_comm_error(void* c) { return static_cast<pba::Comm*>(c)->err.c_str(); }
}
