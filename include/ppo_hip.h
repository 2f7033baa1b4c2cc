/* include/ppo_hip.h -- C-ABI of libppo_hip.so: the MI355X (gfx950) PPO rollout-buffer hot path.
 *
 * The reference (AidanShipperley/PPO-LibTorch) has no plugin/FFI boundary: its "API" is a set of C++ classes
 * whose methods exchange torch::Tensor (PPO/PPO_Discrete.h:24-108, PPO/Agent.h:22-54, the Distributions and
 * Environments headers).  This header is the boundary a maintainer would bind instead of LibTorch for that path:
 * plain pointers and sizes, int32 status codes, no exceptions, no torch types.  Every entry point cites the
 * reference interface (file:line, relative to the reference root) it replaces.  The C++ classes with the
 * reference's names (ppo-libtorch_amd/host/) and the Python ctypes binding (ppo-libtorch_amd/binding.py) sit on
 * top of exactly these symbols; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - All bulk pointers are DEVICE pointers unless the parameter name ends in _h (host).  Callers without a HIP
 *    runtime of their own use ppo_device_alloc / ppo_memcpy_h2d / ppo_memcpy_d2h.
 *  - Work is enqueued on the context's stream (ppo_stream); functions return after enqueueing unless stated.
 *  - Rollout buffers are TIME-MAJOR [T, N, ...] exactly like the reference's m_obs/m_rewards/... tensors
 *    (PPO_Discrete.cpp:90-95), env index contiguous.
 *  - Status 0 = PPO_OK; on failure ppo_last_error() describes it.  Nothing throws across this ABI.
 */
#ifndef PPO_HIP_H
#define PPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_MAX_HEADS 8
#define PPO_API __attribute__((visibility("default")))
#define PPO_ABI_VERSION 5

typedef int32_t ppo_status;
enum { PPO_OK = 0, PPO_ERR_INVALID = 1, PPO_ERR_HIP = 2, PPO_ERR_STATE = 3, PPO_ERR_COMM = 4, PPO_ERR_UNSUPPORTED = 5 };

/* PPO_ENV_SYNTHETIC: the synthetic env of BASELINE configs[4] (obs ~ N(0,1) of any width, reward ~ U(-1,1), done ~ Bernoulli(0.01), random
 * action masks); it is the env with which networks other than the reference's 2 x 64 (hidden / n_hidden) are accepted. */
enum { PPO_ENV_CARTPOLE = 0, PPO_ENV_MOUNTAINCAR = 1, PPO_ENV_SYNTHETIC = 2 };
/* PPO_DIST_CATEGORICAL reproduces Distributions/Categorical.cpp including its entropy clamp (:112-119);
 * PPO_DIST_MASKED reproduces Distributions/CategoricalMasked.cpp (true entropy, -1e8 masking). */
enum { PPO_DIST_CATEGORICAL = 0, PPO_DIST_MASKED = 1 };
/* PPO_DTYPE_F32: every product carries f32 accuracy (on the matrix cores: f32 operands as three exact bf16 terms, f32 accumulation).
 * PPO_DTYPE_BF16: "bf16 with MFMA GEMMs" of BASELINE configs[4] -- operands and stored activations rounded to bf16 (nearest even),
 * f32 accumulation, f32 master weights, gradients and optimizer state. */
enum { PPO_DTYPE_F32 = 0, PPO_DTYPE_BF16 = 1 };

/* Hyper-parameters: the m_* fields of PPO_Discrete (PPO_Discrete.h:52-85) / the TOML keys (PPO_Discrete.cpp:107-255). */
typedef struct ppo_config {
    int32_t struct_size;        /* = sizeof(ppo_config) */
    int32_t device;             /* HIP device ordinal (reference: use_cuda -> torch::kCUDA, PPO_Discrete.cpp:65) */
    int32_t env_kind;           /* PPO_ENV_* : PPO_Discrete owns CartPole, PPO_MultiDiscrete owns MountainCar */
    int32_t dist_kind;          /* PPO_DIST_* */
    int32_t obs_size;           /* [environment] obs_size */
    int32_t n_heads;            /* Agent::m_actionSpace.size() (Agent.cpp:21) */
    int32_t head_dims[PPO_MAX_HEADS]; /* Agent::m_actionSpace; reference: { action_size } */
    int32_t hidden;             /* 64 (Agent.cpp:25-32) */
    int32_t n_hidden;           /* 2 tanh layers */
    int32_t num_envs;           /* envs owned by THIS context (a shard when global_num_envs > num_envs) */
    int32_t num_steps;
    int32_t num_minibatches;
    int32_t update_epochs;
    int32_t max_episode_steps;
    int32_t use_gae, norm_adv, clip_vloss, anneal_lr;
    int64_t seed;
    int64_t total_timesteps;    /* global; drives the LR anneal (PPO_Discrete.cpp:496,514-518) */
    int64_t env_offset;         /* global index of this shard's env 0 */
    int64_t global_num_envs;    /* 0 or num_envs when not sharded */
    float learning_rate, gamma, gae_lambda, clip_coef, ent_coef, vf_coef, max_grad_norm;
    int32_t compute_dtype;      /* PPO_DTYPE_*: arithmetic of the layer GEMMs of a PPO_ENV_SYNTHETIC network; the 2 x 64 paths are always f32 */
    int32_t kernel_flags;       /* PPO_KERNEL_* (0 = the defaults): which hand-written kernel runs a stage, for A/B runs and for bit-exact replays */
} ppo_config;

/* ppo_config.kernel_flags.  The defaults put the reference's two shapes on the matrix cores; every alternative computes the same function.
 *   PPO_KERNEL_ROLLOUT_VECTOR   the fused rollout AND the stand-alone policy (ppo_policy_act) on the vector ALU (rollout2_kernel, policy_act_kernel): logits
 *                               formed by plain fp32 multiply-adds.  The default rollout16_kernel forms layer 2 and the logits as two-term fp16 products on
 *                               the matrix cores, ~1e-7 away.  ppo_policy_act ALWAYS runs the arithmetic the context's rollout runs (default:
 *                               policy_act16_kernel, the same products in the same order; under this flag, or for a launch whose output-layer weights do
 *                               not fit fp16, the vector form), so a free-running rollout and the stand-alone policy on the same observations, weights and
 *                               step index agree in every log-prob bit and every sampled action (tests/test_gpu_parity.py:
 *                               test_fused_rollout_equals_stepwise_api).  Between the two arithmetics an action sampled where the uniform sits within ~1e-6
 *                               of a bin edge can differ (measured: <= 2 of 8 192): use this flag to replay recordings made with it, or with round <= 5
 *                               builds' stand-alone policy.
 *   fp16 ranges                 The matrix-core kernels carry some operands as fp16 terms: rollout16_kernel 2^8 W3 (|W3| < 255), the update kernels c W2 and
 *                               the products through its columns (sum |W2[:, k]| < ~350), c W1 / c b1 / c b2 (< 22 700) and the observation (< 65 504).  The
 *                               reference has none of these limits and a caller never meets the weight limits: the maxima of |parameter| per class are taken
 *                               once per update (and after every host write of the parameters), the host reads a pinned mirror of them without synchronising (thresholds at half the
 *                               limits: max |W3| >= 128, max |W2| >= 4, anything else >= 8192), and a launch whose weights do not fit takes the vector
 *                               kernel (plain fp32, the same function) -- for that launch only, with the default flags.  An OBSERVATION beyond fp16 written
 *                               into PPO_BUF_OBS cannot be foreseen: the update's record packing raises the context's error word (only when the matrix-core
 *                               kernel will read the records) and ppo_read_stats / ppo_stats_snapshot_read fail with PPO_ERR_STATE; PPO_KERNEL_UPDATE_VECTOR
 *                               has no such limit.  The same holds for a hand-over wait of the update kernel, or of the time-pipelined advantage scan, that
 *                               runs out (protocol errors, never observed in 1.6 M launches).  ABI 5: the optimizer kernels read the error word and do NOT
 *                               APPLY a step behind any of these: parameters and AdamW moments keep the values they had before the error (the
 *                               learning-rate schedule and the step counters run on), the error is sticky.  A gradient that is merely not finite (the
 *                               reference's 1-row minibatch) is applied as the reference applies it.
 *   PPO_KERNEL_UPDATE_VECTOR    the update's forward / backward on the vector ALU (fwd_bwd_kernel) for every shape: plain fp32 arithmetic; also for
 *                               rehearsals of more than two ranks on ONE GPU (tests/test_gpu_exchange.py).
 *   PPO_KERNEL_UPDATE_ONE_WAVE  the one-wave-per-tile matrix-core kernel (fwd_bwd_mfma_kernel) instead of the wave-specialised one
 *                               (fwd_bwd_mfma_ws_kernel) for the reference's two shapes: the A/B partner of the default.
 *   PPO_KERNEL_COMM_SELFTEST    ppo_comm_init(..., rank 0, nranks 1) really creates a ONE-rank RCCL communicator and every collective of the multi-rank
 *                               path is really issued (sums over one rank = identity): the only way to drive the RCCL calls -- library lookup,
 *                               datatype / op enums, stream ordering, the three-kernel optimizer path -- on a box with a single GPU.
 *   PPO_KERNEL_GENERIC_CLASSIC  generic networks with bf16 storage (PPO_ENV_SYNTHETIC, e.g. BASELINE configs[4]): a minibatch step in its first form -- the
 *                               minibatch gathered into dense copies, one net per launch with the critic's passes on a second stream, loss sums / gradient
 *                               norm / AdamW / weight planes as four launches -- instead of rows read in place through the index list, both nets in every
 *                               launch on one stream and one optimizer launch.  Same kernels for the products, same partial-sum partitions: the two forms
 *                               agree to the last bits of a float (only the order in which the gradient norm's partial sums are added differs).  For A/B
 *                               runs and tests.
 *   PPO_KERNEL_GENERIC_SPLIT_HEAD  generic networks with bf16 storage: heads, masked categorical, PPO loss and the head layers' backward as launches of their own
 *                               (loss_lanes_kernel, bwd_layer_kernel<1, ...>) behind a forward launch that writes logits, values and the top hidden activation
 *                               to memory -- round 5's step -- instead of in the forward launch's epilogue, on the tile still in LDS (ABI 5 default where the
 *                               shape allows: <= 4 heads of <= 4 logits).  Same arithmetic (the same bf16 roundings, f32 sums); partial sums are formed per
 *                               forward workgroup instead of per row range, so the two agree to f32 summation order.  For A/B runs and tests. */
enum { PPO_KERNEL_ROLLOUT_VECTOR = 1, PPO_KERNEL_UPDATE_VECTOR = 2, PPO_KERNEL_UPDATE_ONE_WAVE = 4, PPO_KERNEL_COMM_SELFTEST = 8, PPO_KERNEL_GENERIC_CLASSIC = 16,
       PPO_KERNEL_GENERIC_SPLIT_HEAD = 32 };

/* Scalars the reference prints per update (PPO_Discrete.cpp:700-774) plus per-step diagnostics. */
typedef struct ppo_stats {
    double pg_loss, v_loss, entropy_loss, approx_kl, loss; /* last minibatch of the update (:588,:599,:619,:628,:631) */
    double clipfrac_last, clipfrac_mean;                   /* m_clipfracs.back() / PPOUtils::getVectorMean (:349,:755) */
    double total_norm;                                     /* clip_grad_norm_ return value (:640) */
    double explained_variance;                             /* :647-648 */
    double learning_rate;                                  /* :515-517 */
    double ep_len_mean, ep_rew_mean;                       /* CircularBuffer (Utils.h:72-78) */
    int64_t ep_count;                                      /* CircularBuffer::size() */
    int64_t global_step;                                   /* :526 */
    int64_t optimizer_steps;                               /* AdamW state step */
    int64_t updates;                                       /* iterations done */
} ppo_stats;

typedef struct ppo_ctx ppo_ctx;

/* Internal device buffers (non-owning views).  Element types in brackets. */
enum {
    PPO_BUF_OBS = 0,      /* m_obs        f32 [T,N,O]                       PPO_Discrete.cpp:90  */
    PPO_BUF_ACTIONS,      /* m_actions    i32 [T,N,H] (reference stores f32 [T,N,1], :91,:537)   */
    PPO_BUF_LOGPROBS,     /* m_logprobs   f32 [T,N]                         :92  */
    PPO_BUF_REWARDS,      /* m_rewards    f32 [T,N]                         :93  */
    PPO_BUF_DONES,        /* m_dones      f32 [T,N]                         :94  */
    PPO_BUF_VALUES,       /* m_values     f32 [T,N]                         :95  */
    PPO_BUF_MASKS,        /* m_action_masks u8 [T,N,A]        PPO_MultiDiscrete.cpp:98 */
    PPO_BUF_ADVANTAGES,   /* f32 [T,N]  calcAdvantage()[1]                  :305 */
    PPO_BUF_RETURNS,      /* f32 [T,N]  calcAdvantage()[0] */
    PPO_BUF_NEXT_OBS,     /* f32 [N,O]                                      :494 */
    PPO_BUF_NEXT_DONE,    /* i32 [N]   (stepEnvs' done tensor is kInt32, :418) */
    PPO_BUF_NEXT_VALUE,   /* f32 [N]                                        :280 */
    PPO_BUF_PARAMS,       /* f32 [P]  Agent::parameters() order: critic then actor (Agent.cpp:65-66) */
    PPO_BUF_GRADS,        /* f32 [P] */
    PPO_BUF_EXP_AVG,      /* f32 [P]  AdamW exp_avg */
    PPO_BUF_EXP_AVG_SQ,   /* f32 [P]  AdamW exp_avg_sq */
    PPO_BUF_ENV_STATE,    /* f32 [O,N] struct-of-arrays env state (CartPole::state, CartPole.h:38) */
    PPO_BUF_EP_LEN,       /* i32 [N]  episode_length  (CartPole.h:44) */
    PPO_BUF_EP_REW,       /* f32 [N]  episode_reward  (CartPole.h:45) */
    PPO_BUF_RESET_COUNT,  /* i32 [N]  resets drawn so far from the env's private generator (CartPole.h:28-29) */
    PPO_BUF_PERM,         /* i32 [E,B] minibatch permutations of the current update (torch::randperm, :569) */
    PPO_BUF_FIN_LEN,      /* i32 [T,N] length of the episode that finished at (t,n), else 0 (:455) */
    PPO_BUF_FIN_REW,      /* f32 [T,N] reward of that episode (:456) */
    PPO_BUF_COUNT_
};

/* ---------------------------------------------------------------------------------------------------------
 * Lifecycle / plumbing
 * ------------------------------------------------------------------------------------------------------- */
PPO_API int32_t ppo_abi_version(void);
/* PPO_Discrete::PPO_Discrete() (PPO_Discrete.cpp:4-100): allocates every device buffer once (rollout [T,N,*],
 * parameters, AdamW state, env SoA, reset-stream table); no allocation happens afterwards. */
PPO_API ppo_status ppo_ctx_create(const ppo_config* cfg, ppo_ctx** out);
PPO_API void ppo_ctx_destroy(ppo_ctx* ctx);
/* Error text of the last failing call on ctx (ctx == NULL: of the last failing ppo_ctx_create in this thread).
 * The C++ facade rethrows it as std::runtime_error like the reference's obs-size check does (:370-375). */
PPO_API const char* ppo_last_error(const ppo_ctx* ctx);
PPO_API ppo_status ppo_sync(ppo_ctx* ctx);              /* block until the context's stream is idle */
PPO_API void* ppo_stream(ppo_ctx* ctx);                 /* hipStream_t */
PPO_API ppo_status ppo_get_config(const ppo_ctx* ctx, ppo_config* out);
PPO_API ppo_status ppo_buffer(ppo_ctx* ctx, int32_t which, void** dev_ptr, size_t* bytes);
PPO_API ppo_status ppo_device_alloc(ppo_ctx* ctx, size_t bytes, void** dev_ptr);
PPO_API ppo_status ppo_device_free(ppo_ctx* ctx, void* dev_ptr);
PPO_API ppo_status ppo_memcpy_h2d(ppo_ctx* ctx, void* dst_dev, const void* src_h, size_t bytes); /* synchronous */
PPO_API ppo_status ppo_memcpy_d2h(ppo_ctx* ctx, void* dst_h, const void* src_dev, size_t bytes); /* synchronous */

/* ---------------------------------------------------------------------------------------------------------
 * Agent (PPO/Agent.h:22-54)
 * ------------------------------------------------------------------------------------------------------- */
PPO_API int64_t ppo_param_count(const ppo_ctx* ctx);
/* 2*(n_hidden+1) tensors per net, critic first: rows {out,in} for weights and {out,1} for biases. */
PPO_API ppo_status ppo_param_shapes(const ppo_ctx* ctx, int64_t* shapes_h, int32_t* n_tensors);
/* Agent::ppoLayerInit (Agent.cpp:91-99): orthogonal_(W, gain) with gain sqrt(2) / 1.0 (critic head) / 0.01 (actor
 * head), bias 0.  Own Householder-QR of a Philox Gaussian (LibTorch's LAPACK+mt19937 draw is not reproducible). */
PPO_API ppo_status ppo_params_init_orthogonal(ppo_ctx* ctx, int64_t seed);
PPO_API ppo_status ppo_params_set_h(ppo_ctx* ctx, const float* params_h, int64_t count); /* also resets nothing else */
PPO_API ppo_status ppo_params_get_h(ppo_ctx* ctx, float* params_h, int64_t count);
PPO_API ppo_status ppo_optimizer_set_h(ppo_ctx* ctx, const float* exp_avg_h, const float* exp_avg_sq_h, int64_t count, int64_t step);
PPO_API ppo_status ppo_optimizer_get_h(ppo_ctx* ctx, float* exp_avg_h, float* exp_avg_sq_h, int64_t count, int64_t* step);

/* Agent::getValue (Agent.cpp:107-109): value[n] = Critic(obs[n,O]). */
PPO_API ppo_status ppo_get_value(ppo_ctx* ctx, const float* obs, int64_t n, float* value);
/* Agent::getActionAndValueDiscrete (Agent.cpp:117-128) / getActionAndValueMasked (:137-170).
 *   obs [n,O]; mask u8 [n,A] or NULL; forced_action i64 [n,H] or NULL (NULL -> sample, Categorical.cpp:73-79, with the
 *   context's counter-based generator keyed by (seed, env_offset+row, step_index, head));
 *   outputs action i64 [n,H] (the transposed layout the reference returns, Agent.cpp:168), logprob/entropy/value f32 [n]
 *   (log-probs and entropies summed over heads, :165-168).  Any output may be NULL.
 *   Arithmetic: that of the kernel ppo_rollout would run in this context now (PPO_KERNEL_ROLLOUT_VECTOR above): bit for bit the rollout's log-probs / actions. */
PPO_API ppo_status ppo_policy_act(ppo_ctx* ctx, const float* obs, const uint8_t* mask, const int64_t* forced_action, int64_t n,
                          int64_t step_index, int64_t* action, float* logprob, float* entropy, float* value);

/* ---------------------------------------------------------------------------------------------------------
 * Distributions (Distributions/Categorical.h:11-22, CategoricalMasked.h:12-23), stateless
 * ------------------------------------------------------------------------------------------------------- */
/* Constructor + log_prob + entropy + mode on logits [n,A]: m_logits = logits - logsumexp, m_probs = softmax
 * (Categorical.cpp:28-39; CategoricalMasked.cpp:31-46), log_prob = gather (:92-101), entropy (:112-119 / :127-144),
 * mode = argmax (:139-141).  value i64 [n] may be NULL; outputs may be NULL. */
PPO_API ppo_status ppo_categorical(int32_t dist_kind, const float* logits, const uint8_t* mask, const int64_t* value, int64_t n,
                           int32_t A, float* m_logits, float* m_probs, float* log_prob, float* entropy, int64_t* mode,
                           void* stream);

/* Categorical::sample / CategoricalMasked::sample (Categorical.cpp:73-79: multinomial(probs, 1, replacement = true)) on given
 * m_probs [n,A]: inverse-CDF draw with the counter-based generator keyed by (seed; row_offset + row, step_index, head). */
PPO_API ppo_status ppo_categorical_sample(const float* m_probs, int64_t n, int32_t A, int64_t seed, int64_t row_offset, int64_t step_index,
                                  int32_t head, int64_t* sample, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Linear layers of networks wider than the reference's 2 x 64 (Agent.cpp:25-59 with the widths of BASELINE configs[4]), stateless
 * ------------------------------------------------------------------------------------------------------- */
/* c[M,N] = epilogue(sum_k A(m,k) B(n,k)) on the matrix cores, all operands f32 on the device:
 *   A(m,k) = trans_a ? a[k * lda + m] : a[m * lda + k],  B(n,k) = trans_b ? b[k * ldb + n] : b[n * ldb + k]
 *   forward  h = tanh(x W^T + b): (0, 0, rows, out, in, x, W, EPI_BIAS_TANH, aux = b)
 *   d(input) dz' = (dz W)(1 - h'^2): (0, 1, rows, in, out, dz, W, EPI_DTANH, aux = h' [M, ld_aux])
 *   d(weight) dW = dz^T x: (1, 1, out, in, rows, dz, x, EPI_NONE)
 * precision PPO_MM_F32X3: every f32 operand is carried as three bf16 terms (exact split) and every product as six bf16 MFMAs with
 * f32 accumulation -- f32 accuracy; PPO_MM_BF16: one round-to-nearest bf16 term per operand, f32 accumulation. */
enum { PPO_MM_EPI_NONE = 0, PPO_MM_EPI_BIAS = 1, PPO_MM_EPI_BIAS_TANH = 2, PPO_MM_EPI_DTANH = 3 };
enum { PPO_MM_F32X3 = 0, PPO_MM_BF16 = 1 };
PPO_API ppo_status ppo_matmul(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, const float* a, int64_t lda, const float* b,
                      int64_t ldb, float* c, int64_t ldc, int32_t epilogue, const float* aux, int64_t ld_aux, int32_t precision,
                      void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Environments (Environments/CartPole.h, MountainCar.h) and their vectorised driver
 * ------------------------------------------------------------------------------------------------------- */
/* Stateless batched CartPole::step (CartPole.cpp:47-94) / MountainCar::step (MountainCar.cpp:29-57) on injected
 * states: state_in [n,O] (array-of-structs like the reference's std::vector<float> state), action i64 [n];
 * outputs next_state [n,O], reward f32 [n], terminated i32 [n]. */
PPO_API ppo_status ppo_env_transition(int32_t env_kind, const float* state_in, const int64_t* action, int64_t n, float* next_state,
                              float* reward, int32_t* terminated, void* stream);
/* First n_resets reset states [n_resets,4] of std::mt19937(seed) + uniform_real_distribution<float>(-0.05,0.05)
 * (CartPole.h:28-29, CartPole.cpp:3-4,34-45): host-side table builder used by the context. */
PPO_API ppo_status ppo_cartpole_reset_stream_h(int64_t seed, int64_t n_resets, float* out_h);
/* PPO_Discrete::initEnvs (PPO_Discrete.cpp:365-402): resets every env (env 0 twice, :368+:389), fills NEXT_OBS,
 * zeroes NEXT_DONE; checks obs_size against the env (:370-375 -> PPO_ERR_INVALID with the reference's message). */
PPO_API ppo_status ppo_env_reset(ppo_ctx* ctx);
/* PPO_Discrete::stepEnvs (PPO_Discrete.cpp:413-483): action i64 [N,H] (column 0 drives the env);
 * outputs obs [N,O] (first obs of the new episode where done), reward f32 [N], done i32 [N]. */
PPO_API ppo_status ppo_env_step(ppo_ctx* ctx, const int64_t* action, float* obs, float* reward, int32_t* done);
/* Inject / read env state for teacher-forced parity: state_h [N,O] AoS, ep_len i32, ep_rew f32, reset_count i32 (NULL = skip). */
PPO_API ppo_status ppo_env_set_state_h(ppo_ctx* ctx, const float* state_h, const int32_t* ep_len_h, const float* ep_rew_h,
                               const int32_t* reset_count_h);
PPO_API ppo_status ppo_env_get_state_h(ppo_ctx* ctx, float* state_h, int32_t* ep_len_h, float* ep_rew_h, int32_t* reset_count_h);

/* ---------------------------------------------------------------------------------------------------------
 * Rollout and advantages
 * ------------------------------------------------------------------------------------------------------- */
/* The rollout loop of PPO_Discrete::train (PPO_Discrete.cpp:524-548; PPO_MultiDiscrete.cpp:547-571) as ONE launch:
 * T x { store obs/done, policy forward + sample, store value/action/logprob, env step + auto-reset, store reward }.
 * forced_actions i64 [T,N,H] or NULL (teacher-forcing for parity).  Leaves NEXT_OBS / NEXT_DONE for the bootstrap. */
PPO_API ppo_status ppo_rollout(ppo_ctx* ctx, const int64_t* forced_actions);
/* PPO_Discrete::calcAdvantage (PPO_Discrete.cpp:274-331) on the context's buffers: bootstrap NEXT_VALUE = Critic(NEXT_OBS)
 * (:280), then GAE (:283-306) or n-step returns (:309-329) by cfg.use_gae; fills ADVANTAGES and RETURNS. */
PPO_API ppo_status ppo_calc_advantage(ppo_ctx* ctx);
/* The same scan on caller buffers (all [T,N] time-major f32; next_value f32 [N]; next_done i32 [N]).  Exact mode:
 * the reference's association and evaluation order along t, no FMA contraction -> bit-identical advantages/returns. */
PPO_API ppo_status ppo_gae(const float* rewards, const float* values, const float* dones, const float* next_value,
                   const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages,
                   float* returns, void* stream);
/* The same scan in FAST mode -- north_star's "segmented prefix sum": the recurrence as a scan of affine maps (c, d) o (c', d') = (c c', d + c d') over
 * chunks of rows, all lanes busy, a done flag cuts the segment.  NOT bit-identical to PPO_Discrete.cpp:283-306 (the carry into a chunk is associated
 * differently: <= 6 ULP of the largest advantage the chain has carried, measured); training (ppo_calc_advantage) never uses it.  Kept to state, with a number, what giving up the
 * reference's association order would buy (profiles/NOTES.md). */
PPO_API ppo_status ppo_gae_fast(const float* rewards, const float* values, const float* dones, const float* next_value,
                        const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages,
                        float* returns, void* stream);
PPO_API ppo_status ppo_nstep_returns(const float* rewards, const float* values, const float* dones, const float* next_value,
                             const int32_t* next_done, int64_t T, int64_t N, float gamma, float* advantages, float* returns,
                             void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Update (PPO_Discrete.cpp:554-648)
 * ------------------------------------------------------------------------------------------------------- */
/* torch::randperm replacement (:569): fills PERM[epoch] for every epoch of the coming update with a keyed
 * cycle-walking Feistel permutation of [0,B) (seed, update, epoch). */
PPO_API ppo_status ppo_generate_permutations(ppo_ctx* ctx);
/* One minibatch, forward + losses + backward (:576-638) on batch rows idx i32 [M] (device).  Leaves the UNCLIPPED
 * gradient of the global-minibatch loss in GRADS (local contribution when sharded) and the loss scalars in the stats. */
PPO_API ppo_status ppo_minibatch_forward_backward(ppo_ctx* ctx, const int32_t* idx, int64_t M);
/* New in the build (no reference counterpart): sum GRADS over ranks with one all-reduce on the context's transport
 * (direct exchange, RCCL or in-process group). No-op unsharded. */
PPO_API ppo_status ppo_allreduce_grads(ppo_ctx* ctx);
/* clip_grad_norm_ (:640; LibTorch clip_grad.h:22-85) + AdamW::step (:641; eps 1e-5f, betas .9/.999, weight_decay 1e-2,
 * :76-78) as one fused launch. */
PPO_API ppo_status ppo_optimizer_step(ppo_ctx* ctx);
/* All epochs x minibatches of one update (:567-644) with the context's own permutations, then explained variance (:647-648). */
PPO_API ppo_status ppo_update(ppo_ctx* ctx);
/* One iteration of the training loop (:511-659 minus printing/checkpoints): LR anneal, rollout, advantages, update. */
PPO_API ppo_status ppo_train_iteration(ppo_ctx* ctx);
/* Synchronises and returns the scalars of the last update (printPPOResults' inputs, :700-774).
 * Sharded runs (2..8 ranks): every number is the JOB's and identical on every rank, as the reference prints one table (:700-774).  The loss scalars
 * (pg / value / entropy loss, approx-KL, clipfrac, grad norm) ride the gradient all-reduce; the explained-variance sums (:647-648) and every
 * rank's ring of finished episodes, each episode tagged with its position in the reference's push order (step, then global env index; :474-480),
 * ride the per-update all-reduce of the advantage sums, and the host rebuilds the job's CircularBuffer(100) from the union: ep_rew_mean /
 * ep_len_mean / ep_count are what ONE context over all envs would report.  They describe the state at the last ppo_update: episodes that end in
 * a rollout taken AFTER it are not in them yet -- unlike a single context (one rank), whose episode ring is read as it stands.  ppo_comm_init
 * refuses more than 8 ranks (the block holds 8 slots: one node).  While a snapshot is pending (ppo_stats_snapshot) ppo_read_stats fails: read it first.
 * Device-side error words (reset table exhausted, fp16 range of the matrix-core rollout, a bounded wait of the update kernel or of the scan) surface here and in
 * ppo_stats_snapshot_read as PPO_ERR_STATE; a host that runs an iteration ahead of its snapshots (the facade's train()) learns of them one
 * iteration late and must check the snapshot before it writes a checkpoint (it does). */
PPO_API ppo_status ppo_read_stats(ppo_ctx* ctx, ppo_stats* out);
/* The same read in two steps, for a host that prints a table per update (printPPOResults, :700-774) and must not drain the GPU to do it:
 * ppo_stats_snapshot enqueues, behind the work enqueued so far, asynchronous copies of everything the statistics are made of into one pinned block
 * and notes the host-side training state; ppo_stats_snapshot_read waits for that snapshot only -- iterations enqueued after it keep running -- and
 * decodes it.  ppo_read_stats = snapshot + read + ppo_sync.  At most two snapshots may be pending (a host that runs one iteration ahead takes the
 * next one before it reads the previous one); they are read oldest first. */
PPO_API ppo_status ppo_stats_snapshot(ppo_ctx* ctx);
PPO_API ppo_status ppo_stats_snapshot_read(ppo_ctx* ctx, ppo_stats* out);
/* LR anneal (:514-518) is applied by ppo_train_iteration; direct control for tests. */
PPO_API ppo_status ppo_set_learning_rate(ppo_ctx* ctx, double lr);

/* ---------------------------------------------------------------------------------------------------------
 * Measurement (new; the reference only has a wall clock around each update, PPO_Discrete.cpp:650-652)
 * ------------------------------------------------------------------------------------------------------- */
/* Per-kernel device time from HIP events recorded on the context's stream around the instrumented launches. */
typedef struct ppo_profile {
    int64_t fwd_bwd_launches, gae_launches, rollout_launches, optimizer_launches, reduce_launches;
    double fwd_bwd_ms, gae_ms, rollout_ms, optimizer_ms, reduce_ms;   /* summed over the launches since enable/read */
    double phase_cycles[24];  /* mode 3: [critic, actor][12 phases] shader cycles of one wave of the dominant kernel */
    int64_t allreduce_launches;   /* N > 1: gradient / statistics all-reduces bracketed (every one in mode 1; in modes 2 and 4 the one behind a bracketed */
    double allreduce_ms;          /* dominant-kernel launch): the collective's device time on THIS rank, waiting for the slowest peer included */
    int64_t vector_fallback_launches;   /* ABI 5: launches since ppo_ctx_create (NOT reset by a read) that took a vector kernel because a weight did not fit the fp16
                                         * operands of the matrix-core kernel of the same function (kernel_flags, "fp16 ranges") */
} ppo_profile;
/* on: 0 = off, 1 = every instrumented launch, 2 = only the dominant kernel (fused forward/backward; ONE launch in 8 is bracketed: an
 * event pair costs the stream ~3 us, 40 pairs per update were 8 % of the run) and the GAE scan,
 * 3 = in-kernel phase stamps of the dominant kernel (diagnostic kernel variant; read shares, not run time),
 * 4 = as 2 with ONE launch in 41 (about one per update, a different step each time): five pairs per 2.4 ms iteration still cost 1.7 % of it
 *     (219.4 against 215.5 M env-steps/s), one costs 0.3 % */
PPO_API ppo_status ppo_profile_enable(ppo_ctx* ctx, int32_t on);
PPO_API ppo_status ppo_profile_read(ppo_ctx* ctx, ppo_profile* out);  /* synchronises; resets the accumulators */

/* ---------------------------------------------------------------------------------------------------------
 * Multi-GPU (new): one context per GPU/process, envs sharded, one gradient all-reduce per optimizer step
 * ------------------------------------------------------------------------------------------------------- */
#define PPO_COMM_ID_BYTES 128
PPO_API ppo_status ppo_comm_unique_id(void* id_out_h /* PPO_COMM_ID_BYTES */);
PPO_API ppo_status ppo_comm_init(ppo_ctx* ctx, const void* id_h, int32_t rank, int32_t nranks);
/* Same contract without RCCL for contexts that live in ONE process (one host thread per context; up to 8): the ranks of
 * `group_id` rendezvous inside each all-reduce and the last to arrive sums every rank's buffer in rank order on its stream. */
PPO_API ppo_status ppo_comm_init_local(ppo_ctx* ctx, int64_t group_id, int32_t rank, int32_t nranks);
/* One-shot direct exchange (no reference counterpart): the latency-bound all-reduce (36.6 KB of gradient per optimizer step) without a
 * ring.  One process per GPU, up to the 8 GPUs of a node.  Every rank exports a small exchange buffer (fine-grained device memory, HIP IPC),
 * the handles are gathered by the host (e.g. torch.distributed), every rank maps its peers'; an all-reduce is then ONE kernel per rank that
 * pushes its payload into its slot of every peer's buffer over xGMI and adds every rank's, in rank order (bit-identical sums everywhere), out
 * of its own buffer.  Inside ppo_update the exchange rides in the gradient reduction: a sharded optimizer step is two launches, as on one GPU.
 *   1. ppo_comm_exchange_handle(ctx, handle)          on every rank
 *   2. gather the nranks handles in rank order
 *   3. ppo_comm_init_exchange(ctx, handles, rank, n)  on every rank
 * A kernel never spins forever: it waits for a peer's share at most the wait limit (default 30 s, ppo_comm_set_wait_limit -- keep it above any
 * host-side skew between ranks: a checkpoint write, a statistics read-back), then gives up with an incomplete sum, sets the timeout flag
 * (ppo_comm_exchange_timeouts: zero / non-zero) and marks the communicator dead (no later call waits again).  The failure is REPORTED: the next
 * ppo_sync, ppo_read_stats or ppo_profile_read of the context returns PPO_ERR_COMM (the replicas have diverged; the job must stop).
 * HIP IPC between processes needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every rank on hosts whose driver only supports dmabuf IPC
 * (set before the process's first HIP call; ppo-libtorch_amd/dist.py does it at import). */
#define PPO_COMM_HANDLE_BYTES 64
PPO_API ppo_status ppo_comm_exchange_handle(ppo_ctx* ctx, void* handle_out_h /* PPO_COMM_HANDLE_BYTES */);
PPO_API ppo_status ppo_comm_init_exchange(ppo_ctx* ctx, const void* handles_h /* nranks x PPO_COMM_HANDLE_BYTES */, int32_t rank, int32_t nranks);
PPO_API ppo_status ppo_comm_set_wait_limit(ppo_ctx* ctx, double seconds);
PPO_API ppo_status ppo_comm_exchange_timeouts(ppo_ctx* ctx, int32_t* nonzero_out);

#ifdef __cplusplus
}
#endif
#endif /* PPO_HIP_H */
