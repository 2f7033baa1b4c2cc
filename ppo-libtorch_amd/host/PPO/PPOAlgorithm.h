// Shared implementation behind the two algorithm classes of the reference, which are near-verbatim copies of each other
// (PPO/PPO_Discrete.{h,cpp} and PPO/PPO_MultiDiscrete.{h,cpp}; their diff is the env type, the masked distribution, the
// action-mask buffer and max_episode_steps).  Public members keep the reference's names (PPO_Discrete.h:24-108).
#pragma once
#include <array>
#include <chrono>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../Tensor.h"
#include "../Utils/ThreadPool.h"
#include "../Utils/Utils.h"
#include "Agent.h"

class PPOAlgorithm {
  public:
    virtual ~PPOAlgorithm();

    // Setup
    void getArgs();                       // ./PPOConfig.toml, sections [environment] [general] [ppo] (PPO_Discrete.cpp:107-255)
    void loadPolicyFromCheckpoint();      // newest file of ./ModelCheckpoints + ./OptimizerCheckpoints (:782-835)

    // ALGO LOGIC
    std::array<ppo::Tensor, 2> calcAdvantage(const ppo::Tensor& next_obs, const ppo::Tensor& next_done) const;  // {returns, advantages}, :274-331
    ppo::Tensor getApproxKLAndClippedObj(const ppo::Tensor& ratio, const ppo::Tensor& logratio);                 // :343-356
    void train();                                                                                              // :485-690

    // Controlling Environments
    std::tuple<ppo::Tensor, ppo::Tensor, ppo::Tensor> stepEnvs(const ppo::Tensor& action);                      // :413-483

    // Printing results to console
    void printPPOResults(int64_t update, int64_t global_step, std::chrono::milliseconds fps, std::chrono::milliseconds time_elapsed,
                         ppo::Tensor& approx_kl, ppo::Tensor& entropy_loss, ppo::Tensor& explained_var, ppo::Tensor& loss, ppo::Tensor& pg_loss,
                         ppo::Tensor& v_loss);
    template <typename T> void printElement(T t, const int& width) {
        std::cout << std::left << std::setw(width) << std::setfill(' ') << t << std::right << "|\n";
    }

    // Hyperparameters (defaults PPO_Discrete.cpp:7-29)
    int64_t m_obs_size;
    int64_t m_action_size;
    float m_action_high, m_action_low;   // PPO_MultiDiscrete only, unused (PPO_MultiDiscrete.cpp:136-144)
    float m_learning_rate;
    int64_t m_seed;
    int64_t m_total_timesteps;
    bool m_use_cuda;                      // true = run on the GPU (there is no CPU path; false is rejected at construction)
    bool m_torch_deterministic;           // accepted for config compatibility; the HIP path is deterministic by construction
    int64_t m_num_envs;
    int64_t m_num_steps;
    bool m_anneal_lr;
    bool m_use_gae;
    float m_gamma;
    float m_gae_lambda;
    int64_t m_num_minibatches;
    int64_t m_update_epochs;
    bool m_norm_adv;
    float m_clip_coef;
    bool m_clip_vloss;
    float m_ent_coef;
    float m_vf_coef;
    float m_max_grad_norm;
    int64_t m_checkpoint_updates;
    int64_t m_max_episode_steps;

    int64_t m_batch_size;
    int64_t m_minibatch_size;

    std::shared_ptr<ppo::Device> m_device;
    std::shared_ptr<Agent> m_agent;
    ppo_ctx* m_ctx = nullptr;             // owns parameters, AdamW state (m_optimizer in the reference), envs and rollout buffers

    // Rollout buffers: views of the context's device buffers (time-major, PPO_Discrete.cpp:90-95)
    ppo::Tensor m_obs, m_actions, m_logprobs, m_rewards, m_dones, m_values, m_action_masks;

    std::vector<float> m_clipfracs;
    ppo_stats m_last_stats{};             // the statistics snapshot train() is printing (printPPOResults reads its learning rate from here, not from
    bool m_last_stats_valid = false;      // the context, which may already be an iteration ahead); false outside train(): the context is asked
    std::unique_ptr<CircularBuffer> m_episode_stats;
    uint64_t m_global_step;
    std::shared_ptr<ThreadPool> m_threadPool;

  protected:
    PPOAlgorithm(int env_kind, int dist_kind, int64_t default_obs, int64_t default_max_episode_steps);
    void construct();                     // second half of the reference's constructor: needs the final hyper-parameters
    ppo::Tensor initEnvsImpl();
    AgentOutput actImpl(const ppo::Tensor& obs, const ppo::Tensor* mask, const ppo::Tensor& action) const;
    ppo::Tensor bufferView(int which, std::vector<int64_t> shape, ppo::DType dt) const;
    void saveCheckpoint(const std::string& agentFile, const std::string& optimizerFile);
    int m_env_kind, m_dist_kind;
    std::string m_tag;                    // "PPO_Agent_" / file naming
};
