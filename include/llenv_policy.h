/*
 * llenv_policy.h -- C ABI of the on-device PMC policy: the caller-side neighbour of the hot path.
 *
 * Replaces the actor's forward pass of the trained primitive-level policy
 *   networks/legged_robot/pmc_net/pmc_net.py:117-178 (pmc_net), :99-114 (llc), :41-46 (vq_encoder), :148-157 (nearest code)
 * with ONE fused kernel: observation normalisation, encoder 207 -> 256 -> 256 -> 32, nearest code of the (32, 256) codebook,
 * low-level controller [relu(prop 135 -> 64) | relu(code 32 -> 32)] -> 256 -> 256 -> 12, mean action.  The matrix products run
 * on the matrix cores (v_mfma_f32_16x16x4_f32: exact float32), 16 environments per workgroup, activations in LDS.
 *
 * Same conventions as llenv.h.  d_obs / d_actions are DEVICE pointers -- normally the engine's own buffers (ll_device_ptrs),
 * so that a rollout is  ll_policy_act ; ll_step  on one stream with no host involvement.
 */
#ifndef LLENV_POLICY_H
#define LLENV_POLICY_H

#include <stdint.h>

#include "llenv.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LLP_N_ARRAYS 28      /* the model's arrays in the order of the reference's checkpoint (tools/extract_policy.py) */
#define LLP_N_FLOATS 358647  /* their total size: rms 135+135+72+72, vf head, encoder, codebook, llc, decoder, logstd */
#define LLP_OBS_DIM 207
#define LLP_ACT_DIM 12

typedef struct ll_policy ll_policy;

/* h_weights: the 28 arrays, float32, each row-major [in][out], concatenated in checkpoint order. */
int ll_policy_create(const float* h_weights, int n_floats, int device, ll_policy** out);
int ll_policy_destroy(ll_policy* p);
/* mean action of every env: d_obs [n_envs][207] -> d_actions [n_envs][12]; d_code (nullable) [n_envs] int32 receives the index
 * of the chosen code.  Asynchronous on hip_stream (NULL: the default stream). */
int ll_policy_act(ll_policy* p, const float* d_obs, float* d_actions, int32_t* d_code, int n_envs, void* hip_stream);
/*
 * The same forward pass as the ACTOR of a policy-gradient learner needs it (learner inputs pmc_net.py:61-96):
 *   sample != 0   a ~ DiagGaussian(mean, exp(logstd)) (pmc_net.py:109-113) drawn with Philox4x32-10 keyed on (seed; env, step)
 *                 -- the same (seed, step) gives the same actions -- instead of the mean action;
 *   d_neglogp     (nullable) [n_envs]  -log p(a | obs) of the emitted action, DiagGaussianPd.neglogp:
 *                 0.5 sum ((a - mean) / std)^2 + 6 log(2 pi) + sum logstd;
 *   d_value       (nullable) [n_envs]  the value head tanh(207 -> 256) -> tanh(256) -> 1 (pmc_net.py:141-146), arrays w04..w09.
 * Written next to the actions, typically into the engine's own buffers (ll_pg_ptrs), from where the step kernel copies them into
 * the unroll it records (ll_enable_unrolls).
 */
int ll_policy_act_pg(ll_policy* p, const float* d_obs, float* d_actions, int32_t* d_code, float* d_neglogp, float* d_value, int n_envs, uint64_t seed,
                     uint64_t step, int sample, void* hip_stream);
/* HIP-event time of the ll_policy_act launches since the last call (like ll_kernel_time_ms). */
int ll_policy_enable_timing(ll_policy* p, int on);
int ll_policy_time_ms(ll_policy* p, double* avg_ms, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif
