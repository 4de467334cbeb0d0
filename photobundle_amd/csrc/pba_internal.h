// pba_internal.h -- engine entry points shared by the engine and the host LM driver (not part of the public ABI).
#pragma once
#include "../../include/pba.h"

extern "C" {
// grad_only != 0: stop after the reduced solve (cost / gradient norms of the linearisation point only).
int pba_internal_step(pba_engine* e, double radius, int32_t init_scale, const pba_solver_options* o, pba_step_info* out,
                      int grad_only);
int pba_internal_ready(pba_engine* e);   /* PBA_OK, or PBA_ERR_STATE with the reason in pba_last_error */
int pba_internal_world(const pba_engine* e);
int pba_internal_rank(const pba_engine* e);
int pba_internal_is_multi(const pba_engine* e);   /* collectives are enqueued with every step */
int64_t pba_internal_local_blocks(const pba_engine* e);
int pba_internal_patch_len(const pba_engine* e);
int pba_internal_allreduce_host(pba_engine* e, double* v, int n, int op);
// candidate pass = Jacobian pass (speculative linearisation) on/off
void pba_internal_set_speculate(pba_engine* e, int on);
void pba_internal_pass_counts(const pba_engine* e, int64_t* jac, int64_t* cost);
void pba_internal_reset_pass_counts(pba_engine* e);
// asynchronous driver (device-side trust-region decisions)
int pba_internal_async_capable(const pba_engine* e, const pba_solver_options* o);
int pba_internal_async_begin(pba_engine* e, const pba_solver_options* o);
int pba_internal_async_enqueue(pba_engine* e, int kind, int init_scale, const pba_solver_options* o, unsigned long long* seq_out);
int pba_internal_async_wait(pba_engine* e, unsigned long long seq);
const void* pba_internal_async_state(const pba_engine* e);
const pba_iteration_summary* pba_internal_async_log(const pba_engine* e);
int pba_internal_async_end(pba_engine* e);
// resident solve (pba_resident.h): the whole pba_solve as ONE cooperative launch
int pba_internal_resident_capable(pba_engine* e, const pba_solver_options* o);
/* PBA_INTERNAL_RESIDENT_REFUSED: the runtime refused the cooperative launch before anything ran -- nothing changed, the engine has left the
   resident driver for good and the caller runs the solve on the pipelined one */
#define PBA_INTERNAL_RESIDENT_REFUSED (-1000)
int pba_internal_resident_launch(pba_engine* e, const pba_solver_options* o, unsigned long long* seq_out);
void pba_internal_resident_done(pba_engine* e, int iterations);   /* the solve took `iterations` step trips: unused epochs go back */
void pba_internal_resident_failed(pba_engine* e);   /* a resident launch ended without publishing (device-side wait timed out): the engine is unusable */
void pba_internal_resident_trace(pba_engine* e, int iterations);   /* PBA_RES_TRACE: phase intervals of the last resident solve to stderr */
int pba_internal_final_flushes(const pba_engine* e);   /* 1: the kind-2 enqueue also flushes (no kind 3 behind it) */
}
