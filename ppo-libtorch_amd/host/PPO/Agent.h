// Drop-in for the reference's PPO/Agent.h:22-54: two independent MLPs (critic obs->64->64->1, actor obs->64->64->sum(actions),
// tanh, Agent.cpp:25-59) whose parameters live in a ppo_ctx on the GPU.  Same method names and argument meaning;
// ppo::Tensor stands for torch::Tensor.
#pragma once
#include <memory>
#include <numeric>
#include <vector>

#include "../Distributions/Categorical.h"
#include "../Distributions/CategoricalMasked.h"
#include "../Tensor.h"

struct AgentOutput {
    ppo::Tensor action;    // i64 [n, heads]  (the transposed layout getActionAndValueMasked returns, Agent.cpp:168)
    ppo::Tensor logprob;   // f32 [n]
    ppo::Tensor entropy;   // f32 [n]
    ppo::Tensor value;     // f32 [n, 1]
};

class Agent {
  public:
    // Agent(obsSize, actionSize, device), Agent.cpp:19-72.  `actionSpace` generalises the reference's hard-wired
    // m_actionSpace = { actionSize } (Agent.cpp:21) to several heads; `masked` selects CategoricalMasked.
    Agent(int64_t obsSize, int64_t actionSize, std::shared_ptr<ppo::Device> device, std::vector<int64_t> actionSpace = {}, bool masked = false,
          int64_t initSeed = 1);
    // Adopts the context of an algorithm object (PPO_Discrete owns the buffers the agent's parameters live next to).
    Agent(ppo_ctx* ctx, std::shared_ptr<ppo::Device> device, std::vector<int64_t> actionSpace);
    ~Agent();
    Agent(const Agent&) = delete;
    Agent& operator=(const Agent&) = delete;

    ppo::Tensor getValue(const ppo::Tensor& x);                                                       // Agent.cpp:107-109
    AgentOutput getActionAndValueDiscrete(const ppo::Tensor& x, ppo::Tensor action = ppo::Tensor());  // :117-128
    AgentOutput getActionAndValueMasked(const ppo::Tensor& x, const ppo::Tensor& mask, ppo::Tensor action = ppo::Tensor());  // :137-170
    void printAgent();

    // flat parameter vector in Agent::parameters() order (critic first, Agent.cpp:65-66)
    std::vector<float> parameters() const;
    void setParameters(const std::vector<float>& flat);
    int64_t parameterCount() const { return ppo_param_count(m_ctx); }

    std::vector<int64_t> m_actionSpace;
    int64_t m_actionSpaceSum;
    int64_t m_actionSpaceSize;
    std::shared_ptr<ppo::Device> m_device;
    ppo_ctx* m_ctx = nullptr;
    int64_t m_sampleCalls = 0;   // position of the sampling stream (step index handed to the kernel)

  private:
    AgentOutput act(const ppo::Tensor& x, const ppo::Tensor* mask, const ppo::Tensor& action);
    bool m_owns = false;
};
