/*
 * oracle/port/vracer_port.h -- CPU restatement of the smarties V-RACER learner
 * update.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the CHECKER; never by the product path.
 *
 * Pinned against golden fixtures generated from the compiled reference
 * (oracle/_ref/ref_driver, tests/golden/, script tests/golden/make_golden.sh).
 *
 * The exported C functions are the hl_* entry points of include/smarties_hip.h
 * with the prefix ol_ (same structs, same call sequence).
 */
#ifndef SMARTIES_AMD_ORACLE_PORT_H
#define SMARTIES_AMD_ORACLE_PORT_H
#include "../../include/smarties_hip.h"
#include "../synth.h"

#ifdef __cplusplus
extern "C" {
#endif
typedef struct ol_learner ol_learner;

HL_API int ol_create(const hl_config* cfg, ol_learner** out);
HL_API int ol_destroy(ol_learner* h);
HL_API const char* ol_last_error(const ol_learner* h);
HL_API int64_t ol_num_params(const ol_learner* h);
HL_API int32_t ol_num_outputs(const ol_learner* h);
HL_API int32_t ol_num_layers(const ol_learner* h);
HL_API int ol_param_layout(const ol_learner* h, int64_t* indW, int64_t* nW, int64_t* indB, int64_t* nB);
HL_API int ol_init_weights(ol_learner* h);
HL_API int ol_set_params(ol_learner* h, const float* w, const float* m1, const float* m2);
HL_API int ol_get_params(ol_learner* h, float* w, float* m1, float* m2);
HL_API int ol_set_rng_state(ol_learner* h, const uint32_t state[625]);
HL_API int ol_get_rng_state(ol_learner* h, uint32_t state[625]);
HL_API int ol_append_episode(ol_learner* h, int32_t nsteps, const float* states, const double* actions,
                             const double* mu, const double* rewards, const float* values,
                             const float* advantages, int32_t terminated, int64_t tag);
HL_API int ol_get_scaling(ol_learner* h, float* stateMean, float* stateScale, float* rew3);
HL_API int ol_set_scaling(ol_learner* h, const float* stateMean, const float* stateScale, const float* rew3);
HL_API int ol_get_episode_field(ol_learner* h, int64_t pos, int32_t field, float* dst, int32_t cap);
HL_API int ol_get_episode_info(ol_learner* h, int64_t pos, int64_t* tag, int32_t* nsteps, int32_t* terminated);
HL_API int ol_get_episode_stats(ol_learner* h, int64_t pos, float* dst9);
HL_API int ol_initialize(ol_learner* h);
HL_API int ol_initialize_begin(ol_learner* h);
HL_API int ol_initialize_end(ol_learner* h);
HL_API int ol_step(ol_learner* h, int32_t n_steps, const int64_t* flat_indices);
HL_API int ol_step_begin(ol_learner* h, const int64_t* flat_indices);
HL_API int ol_grad_exchange(ol_learner* h, float* grad_io, int32_t write_back);
HL_API int ol_counters_exchange(ol_learner* h, int64_t counters_io[4], int32_t write_back);
HL_API int ol_moments_exchange(ol_learner* h, double* io, int32_t write_back);
HL_API int ol_step_end(ol_learner* h);
HL_API int ol_sync(ol_learner* h);
HL_API int ol_prepare_steps(ol_learner* h, int32_t n_steps);
HL_API int64_t ol_packed_episode_size(const ol_learner* h, int32_t nsteps);
HL_API int ol_append_packed_episode(ol_learner* h, const float* data, int64_t n_floats);
HL_API int ol_pack_episode(ol_learner* h, int64_t episode_pos, float* dst, int64_t cap_floats);
HL_API int ol_save(ol_learner* h, const char* base);
HL_API int ol_restart(ol_learner* h, const char* base);
HL_API int ol_forward(ol_learner* h, int32_t n, const float* states, double* outputs);
HL_API int ol_grad_stats(ol_learner* h, double* mean, double* rms);
HL_API int ol_set_log_base(ol_learner* h, const char* base);
HL_API int ol_set_episode_log(ol_learner* h, const char* path);
HL_API int ol_forward_sequence(ol_learner* h, int32_t n_steps, const float* states, double* outputs);
HL_API int ol_set_tap(ol_learner* h, int32_t enable);
HL_API int ol_readback(ol_learner* h, int32_t what, void* dst, int64_t dst_bytes);
HL_API int ol_get_scalars(ol_learner* h, hl_scalars* out);
HL_API int ol_get_stats(ol_learner* h, hl_stats* out);
HL_API int ol_get_initial_data(ol_learner* h, int64_t* nInitialData);
HL_API int ol_metrics(ol_learner* h, char* header, int32_t header_cap, char* line, int32_t line_cap);
HL_API int ol_impweight_histogram(ol_learner* h, char* text, int32_t text_cap, int64_t counts[HL_IMPW_BINS]);
HL_API int ol_get_counts(ol_learner* h, int64_t* nStoredSteps, int64_t* nStoredEps, int64_t* nGradSteps, int64_t* nSeenSteps, int64_t* nSeenEps);

/* synthetic replay (oracle/synth.h) exposed for python */
HL_API int ol_synth_episode_len(const synth_cfg* c, uint64_t e, int* terminated);
HL_API void ol_synth_episode(const synth_cfg* c, uint64_t e, float* states, double* actions,
                             double* mu, double* rewards, float* values);
/* single-function probes used by unit tests of the head math */
HL_API void ol_head_vracer(int dA, const uint8_t* bounded, const double* O, const double* act,
                           const double* mu, double Qret, double beta, double Cmax, double Cinv,
                           double* grad /*1+2dA*/, double* rho, double* dkl, double* deltaQ, int* isFar,
                           double* Vval);
HL_API void ol_head_racer(int dA, int nAdv, const uint8_t* bounded, const double* O, const double* act,
                          const double* mu, double Qret, double beta, double Cmax, double Cinv,
                          double* grad /*1+nAdv+2dA*/, double* rho, double* dkl, double* deltaQ, int* isFar,
                          double* Vval, double* Qval);
HL_API void ol_head_discrete(int nOpt, const double* O, double actMsg, const double* mu, double Qret, double beta,
                             double Cmax, double Cinv, double* grad /*1+2nOpt*/, double* rho, double* dkl, double* deltaQ,
                             int* isFar, double* Vval, double* Qval);
HL_API void ol_synth_episode_discrete(const synth_cfg* c, int nOpt, uint64_t e, float* states, double* actions,
                                      double* mu, double* rewards, float* values);
#ifdef __cplusplus
}
#endif
#endif
