#include "CartPole.h"

CartPole::CartPole(int64_t seed, std::shared_ptr<ppo::Device> device)
    : state{ 0, 0, 0, 0 }, terminated(false), episode_length(0), episode_reward(0.0f), m_device(std::move(device)), m_seed(seed) {
    if (!m_device) m_device = std::make_shared<ppo::Device>(0);
}

float CartPole::randomUniform() {
    if (static_cast<size_t>(m_draws) >= m_stream.size()) {
        const int64_t resets = std::max<int64_t>(64, 2 * (m_draws / 4 + 1));
        m_stream.resize(static_cast<size_t>(resets * 4));
        if (ppo_cartpole_reset_stream_h(m_seed, resets, m_stream.data()) != PPO_OK) throw std::runtime_error("CartPole: reset stream");
    }
    return m_stream[static_cast<size_t>(m_draws++)];
}

std::vector<float> CartPole::reset() {
    for (float& s : state) s = randomUniform();
    terminated = false;
    episode_length = 0;
    episode_reward = 0.0f;
    return state;
}

std::tuple<std::vector<float>, float, bool, bool> CartPole::step(const int64_t& action) {
    using ppo::Tensor;
    Tensor s = Tensor::from_host<float>(m_device, state, { 1, 4 });
    Tensor a = Tensor::from_host<int64_t>(m_device, { action }, { 1 });
    Tensor ns(m_device, { 1, 4 }, ppo::DType::f32), r(m_device, { 1 }, ppo::DType::f32), t(m_device, { 1 }, ppo::DType::i32);
    ppo::check(ppo_env_transition(PPO_ENV_CARTPOLE, s.data<float>(), a.data<int64_t>(), 1, ns.data<float>(), r.data<float>(), t.data<int32_t>(),
                                  m_device->stream()),
               m_device->util(), "CartPole::step");
    state = ns.cpu<float>();
    if (t.item<int32_t>() != 0) terminated = true;   // sticky until reset(), as the reference's member (CartPole.cpp:76-78)
    const float reward = terminated ? -1.0f : 1.0f;  // CartPole.cpp:80-88
    episode_length += 1;
    episode_reward += reward;
    return { state, reward, terminated, false };
}
