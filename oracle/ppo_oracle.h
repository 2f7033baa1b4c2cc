/* oracle/ppo_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's PPO hot path (AidanShipperley/PPO-LibTorch), written from
 * the expressions cited next to each function (paths relative to /root/reference).  It is the CHECKER
 * for the HIP path in ppo-libtorch_amd/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  It is pinned against golden vectors produced by the compiled reference
 * itself (oracle/ref_harness.cpp -> the .pgld files under tests/golden; tests/test_oracle_vs_golden.py).
 *
 * Number model: IEEE binary32 with separately rounded operations (built with -ffp-contract=off), the
 * association of every expression as written in the reference; reductions (means, norms) accumulate in
 * binary64 (the reference's LibTorch cascade sums differ from any fixed order by ~1e-7 relative, inside
 * the 1e-5 loss tolerance north_star states).
 */
#ifndef PPO_ORACLE_H
#define PPO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_HEADS 8

enum { ORC_DIST_CATEGORICAL = 0, ORC_DIST_MASKED = 1 };
enum { ORC_ENV_CARTPOLE = 0, ORC_ENV_MOUNTAINCAR = 1 };
/* ORC_DTYPE_F32: the reference's arithmetic.  ORC_DTYPE_BF16: "bf16 with MFMA GEMMs" of BASELINE.json configs[4] (no reference
 * counterpart: the reference has no reduced-precision mode) -- the mixed-precision scheme the HIP path implements with
 * ppo_config.compute_dtype = PPO_DTYPE_BF16, restated in scalar C so that it can be checked:
 *   every operand of a Linear product (weights, layer inputs, back-propagated d(pre-activation)) is rounded to bf16, round to nearest
 *   even; products are exact and accumulated in f32; biases, tanh, the loss, bias gradients and the optimizer stay f32; a hidden
 *   activation is STORED as bf16, so tanh' in the backward pass sees the rounded value. */
enum { ORC_DTYPE_F32 = 0, ORC_DTYPE_BF16 = 1 };

typedef struct orc_net {
    int32_t obs_size;
    int32_t n_heads;
    int32_t head_dims[ORC_MAX_HEADS];
    int32_t hidden;    /* width of every hidden layer (reference: 64, Agent.cpp:25-32) */
    int32_t n_hidden;  /* number of tanh hidden layers (reference: 2) */
    int32_t dist_kind; /* ORC_DIST_* */
    int32_t dtype;     /* ORC_DTYPE_*: arithmetic of the Linear layers */
} orc_net;

typedef struct orc_hparams {
    float gamma, gae_lambda, clip_coef, ent_coef, vf_coef, max_grad_norm;
    int32_t norm_adv, clip_vloss;
} orc_hparams;

/* ---- libm restatement (glibc 2.35 sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, FMA ifunc variant) ---- */
float orc_sinf(float x);
float orc_cosf(float x);

/* ---- RNG ---- */
/* std::mt19937(seed) + std::uniform_real_distribution<float>(-0.05f, 0.05f): CartPole.h:28-29, CartPole.cpp:3-4,96-100.
 * Writes the first n_resets reset states ([n_resets,4]) of the stream every env shares (PPO_Discrete.cpp:84-86). */
void orc_cartpole_reset_stream(int64_t seed, int64_t n_resets, float* out);
/* Philox4x32-10 (the build's own counter-based generator; no reference counterpart). */
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);

/* ---- environments ---- */
/* CartPole::step, CartPole.cpp:47-94.  state[4] updated in place; returns reward; *terminated set. */
float orc_cartpole_step(float* state, int64_t action, int32_t* terminated);
/* MountainCar::step, MountainCar.cpp:29-57. */
float orc_mountaincar_step(float* state, int64_t action, int32_t* terminated);
/* n independent transitions through the two functions above (kind: 0 CartPole, 1 MountainCar), in place on state [n, obs]. */
void orc_step_many(int32_t kind, float* state, const int64_t* action, int64_t n, float* reward, int32_t* terminated);

/* Synthetic environment of BASELINE configs[4] (obs ~ N(0,1), reward ~ U(-1,1), done ~ Bernoulli(0.01), random masks
 * with at least one valid action per head).  No reference counterpart: it is the build's own definition, memoryless and
 * counter-based (Philox keyed by seed; global env, global step, block, tag), restated here with integer arithmetic so that the device
 * kernels (ppo-libtorch_amd/csrc/kernels_generic.hip) can be checked bit for bit. */
void orc_synthetic_obs(int64_t seed, int64_t env, int64_t step, int32_t obs_size, float* out);
void orc_synthetic_mask(int64_t seed, int64_t env, int64_t step, int32_t n_heads, const int32_t* head_dims, uint8_t* out);
void orc_synthetic_transition(int64_t seed, int64_t env, int64_t step, float* reward, int32_t* done);

/* Vectorised env with the semantics of PPO_Discrete::initEnvs/stepEnvs (PPO_Discrete.cpp:365-483). */
typedef struct orc_vecenv orc_vecenv;
orc_vecenv* orc_vecenv_create(int32_t kind, int64_t num_envs, int64_t seed, int64_t max_episode_steps, int64_t env_offset);
void orc_vecenv_destroy(orc_vecenv*);
void orc_vecenv_init(orc_vecenv*, float* obs_out);                       /* initEnvs */
void orc_vecenv_step(orc_vecenv*, const int64_t* action, float* obs_out, /* stepEnvs */
                     float* reward_out, int32_t* done_out);
void orc_vecenv_set_state(orc_vecenv*, const float* state, const int64_t* ep_len, const float* ep_rew, const int64_t* reset_count);
void orc_vecenv_get_state(const orc_vecenv*, float* state, int64_t* ep_len, float* ep_rew, int64_t* reset_count);
/* CircularBuffer(100) of finished episodes (Utils.h:30-79): out = {avgLength, avgReward, size}. */
void orc_vecenv_episode_stats(const orc_vecenv*, double out[3]);

/* ---- network ---- */
int64_t orc_param_count(const orc_net*);
/* Flat parameter order = Agent::parameters(): critic {W,b} per layer, then actor {W,b} per layer (Agent.cpp:65-66). */
void orc_param_shapes(const orc_net*, int64_t* shapes /* [2*(n_hidden+1)*2][2] */);
/* Critic forward, Agent::getValue (Agent.cpp:107-109). */
void orc_get_value(const orc_net*, const float* params, const float* x, int64_t n, float* value);
/* Actor logits (m_Actor->forward, Agent.cpp:119). */
void orc_actor_logits(const orc_net*, const float* params, const float* x, int64_t n, float* logits);
/* Categorical / CategoricalMasked on given logits (Categorical.cpp:28-39,92-119; CategoricalMasked.cpp:31-46,107-144).
 * One head of width A; mask may be NULL for ORC_DIST_CATEGORICAL. */
void orc_categorical(int32_t dist_kind, const float* logits, const uint8_t* mask, const int64_t* value, int64_t n, int32_t A,
                     float* m_logits, float* m_probs, float* log_prob, float* entropy);
/* Agent::getActionAndValueDiscrete / getActionAndValueMasked with a given action (teacher-forced), Agent.cpp:117-170.
 * action is [n, n_heads] (the transposed layout the reference returns); mask is [n, sum(head_dims)] or NULL. */
void orc_evaluate(const orc_net*, const float* params, const float* x, const uint8_t* mask, const int64_t* action, int64_t n,
                  float* logprob, float* entropy, float* value);
/* Sampling with the build's own generator: inverse-CDF on softmax probabilities with
 * u = word (step % 4) of philox(seed; env, step / 4, head, 0), >> 8, scaled by 2^-24.  (The reference samples with torch::multinomial on
 * LibTorch's global CPU generator, Categorical.cpp:73-79 -- not reproducible off that generator.) */
void orc_act(const orc_net*, const float* params, const float* x, const uint8_t* mask, int64_t n, int64_t seed,
             int64_t env_offset, int64_t step, int64_t* action, float* logprob, float* entropy, float* value);

/* ---- advantages ---- */
/* PPO_Discrete::calcAdvantage GAE branch, PPO_Discrete.cpp:283-306.  Buffers are time-major [T,N]. */
void orc_gae(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
             int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages, float* returns);
/* n-step branch, PPO_Discrete.cpp:309-329. */
void orc_nstep(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
               int64_t T, int64_t N, float gamma, float* advantages, float* returns);

/* ---- update ---- */
/* One minibatch of PPO_Discrete.cpp:576-638: gather by idx, forward, losses, backward.  grads[P] (unclipped).
 * stats = {pg_loss, v_loss, entropy_loss, approx_kl, clipfrac, loss}.  b_actions is the float buffer the reference stores
 * ([B] for Discrete, [B, act_cols] for MultiDiscrete with act_cols = action_size), mask [B, A] or NULL. */
void orc_minibatch_grads(const orc_net*, const orc_hparams*, const float* params, const float* b_obs, const float* b_actions,
                         int32_t act_cols, const uint8_t* b_mask, const float* b_logprobs, const float* b_advantages,
                         const float* b_returns, const float* b_values, const int64_t* idx, int64_t M, float* grads,
                         double stats[6]);
/* One shard of a data-parallel minibatch step (see ppo_oracle.c): global advantage sums in, 1/global_M-scaled gradients out. */
void orc_minibatch_grads_shard(const orc_net*, const orc_hparams*, const float* params, const float* b_obs, const float* b_actions,
                               int32_t act_cols, const uint8_t* b_mask, const float* b_logprobs, const float* b_advantages,
                               const float* b_returns, const float* b_values, const int64_t* idx, int64_t M, const double* adv_sums,
                               int64_t global_M, float* grads, double stats[6], double local_adv_sums[2]);
/* torch::nn::utils::clip_grad_norm_ (LibTorch clip_grad.h:22-85, called at PPO_Discrete.cpp:640). Returns total_norm. */
double orc_clip_grad_norm(const orc_net*, float* grads, float max_norm);
/* torch::optim::AdamW::step (PPO_Discrete.cpp:76-78,641; eps 1e-5f, betas .9/.999, weight_decay 1e-2). step_t is 1-based. */
void orc_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t P, double lr, int64_t step_t);
/* Explained variance, PPO_Discrete.cpp:647-648. */
double orc_explained_variance(const float* returns, const float* values, int64_t B);

#ifdef __cplusplus
}
#endif
#endif
