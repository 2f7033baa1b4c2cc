#include "Agent.h"

#include <iostream>

using ppo::DType;
using ppo::Tensor;

Agent::Agent(int64_t obsSize, int64_t actionSize, std::shared_ptr<ppo::Device> device, std::vector<int64_t> actionSpace, bool masked, int64_t initSeed)
    : m_device(std::move(device)) {
    m_actionSpace = actionSpace.empty() ? std::vector<int64_t>{ actionSize } : std::move(actionSpace);
    m_actionSpaceSum = std::accumulate(m_actionSpace.begin(), m_actionSpace.end(), int64_t{0});
    m_actionSpaceSize = static_cast<int64_t>(m_actionSpace.size());
    if (m_actionSpaceSum != actionSize) throw std::runtime_error("Agent: sum(actionSpace) must equal actionSize");
    ppo_config c{};
    c.struct_size = sizeof c;
    c.device = m_device->ordinal();
    c.env_kind = obsSize == 2 ? PPO_ENV_MOUNTAINCAR : PPO_ENV_CARTPOLE;
    c.dist_kind = masked ? PPO_DIST_MASKED : PPO_DIST_CATEGORICAL;
    c.obs_size = static_cast<int32_t>(obsSize);
    c.n_heads = static_cast<int32_t>(m_actionSpace.size());
    for (size_t i = 0; i < m_actionSpace.size() && i < PPO_MAX_HEADS; i++) c.head_dims[i] = static_cast<int32_t>(m_actionSpace[i]);
    c.hidden = 64; c.n_hidden = 2; c.num_envs = 1; c.num_steps = 1; c.num_minibatches = 1; c.update_epochs = 1; c.max_episode_steps = 500;
    c.use_gae = 1; c.seed = initSeed; c.learning_rate = 3e-4f; c.gamma = 0.99f; c.gae_lambda = 0.95f; c.clip_coef = 0.2f; c.vf_coef = 0.5f; c.max_grad_norm = 0.5f;
    ppo::check(ppo_ctx_create(&c, &m_ctx), nullptr, "Agent");
    m_owns = true;
    // ppoLayerInit: orthogonal weights (gain sqrt 2, 1.0 critic head, 0.01 actor head), zero biases (Agent.cpp:25-37, 91-99)
    ppo::check(ppo_params_init_orthogonal(m_ctx, initSeed), m_ctx, "Agent init");
}

Agent::Agent(ppo_ctx* ctx, std::shared_ptr<ppo::Device> device, std::vector<int64_t> actionSpace)
    : m_actionSpace(std::move(actionSpace)), m_device(std::move(device)), m_ctx(ctx) {
    m_actionSpaceSum = std::accumulate(m_actionSpace.begin(), m_actionSpace.end(), int64_t{0});
    m_actionSpaceSize = static_cast<int64_t>(m_actionSpace.size());
}

Agent::~Agent() { if (m_owns) ppo_ctx_destroy(m_ctx); }

Tensor Agent::getValue(const Tensor& x) {
    const int64_t n = x.size(0);
    Tensor v(m_device, { n, 1 }, DType::f32);
    ppo::check(ppo_get_value(m_ctx, x.data<float>(), n, v.data<float>()), m_ctx, "Agent::getValue");
    ppo::check(ppo_sync(m_ctx), m_ctx, "sync");
    return v;
}

AgentOutput Agent::act(const Tensor& x, const Tensor* mask, const Tensor& action) {
    const int64_t n = x.size(0), H = m_actionSpaceSize;
    AgentOutput o;
    o.action = Tensor(m_device, { n, H }, DType::i64);
    o.logprob = Tensor(m_device, { n }, DType::f32);
    o.entropy = Tensor(m_device, { n }, DType::f32);
    o.value = Tensor(m_device, { n, 1 }, DType::f32);
    const bool forced = action.defined() && action.numel() > 0;   // if (!action.numel()) sample (Agent.cpp:122-124)
    if (forced && action.numel() != n * H) throw std::runtime_error("Agent: action must hold n x heads entries");
    ppo::check(ppo_policy_act(m_ctx, x.data<float>(), mask ? mask->data<uint8_t>() : nullptr, forced ? action.data<int64_t>() : nullptr, n,
                              m_sampleCalls, o.action.data<int64_t>(), o.logprob.data<float>(), o.entropy.data<float>(), o.value.data<float>()),
               m_ctx, "Agent::getActionAndValue");
    if (!forced) m_sampleCalls++;
    ppo::check(ppo_sync(m_ctx), m_ctx, "sync");
    return o;
}

AgentOutput Agent::getActionAndValueDiscrete(const Tensor& x, Tensor action) { return act(x, nullptr, action); }

AgentOutput Agent::getActionAndValueMasked(const Tensor& x, const Tensor& mask, Tensor action) { return act(x, &mask, action); }

std::vector<float> Agent::parameters() const {
    std::vector<float> p(static_cast<size_t>(ppo_param_count(m_ctx)));
    ppo::check(ppo_params_get_h(m_ctx, p.data(), static_cast<int64_t>(p.size())), m_ctx, "Agent::parameters");
    return p;
}

void Agent::setParameters(const std::vector<float>& flat) {
    ppo::check(ppo_params_set_h(m_ctx, flat.data(), static_cast<int64_t>(flat.size())), m_ctx, "Agent::setParameters");
}

void Agent::printAgent() {
    static const char* names[] = { "m_Critic.criticInputLayer", "m_Critic.criticMiddleLayer", "m_Critic.criticOutputLayer",
                                   "m_Actor.actorInputLayer", "m_Actor.actorMiddleLayer", "m_Actor.actorOutputLayer" };
    int64_t shapes[24];
    int32_t nt = 0;
    ppo::check(ppo_param_shapes(m_ctx, shapes, &nt), m_ctx, "param shapes");
    const std::vector<float> p = parameters();
    size_t off = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t rows = shapes[2 * t], cols = shapes[2 * t + 1];
        std::cout << names[t / 2] << (t % 2 ? ".bias" : ".weight") << " [" << rows << (t % 2 ? "" : " x " + std::to_string(cols)) << "]\n";
        for (int64_t i = 0; i < rows * cols && i < 8; i++) std::cout << " " << p[off + static_cast<size_t>(i)];
        std::cout << (rows * cols > 8 ? " ...\n" : "\n");
        off += static_cast<size_t>(rows * cols);
    }
}
