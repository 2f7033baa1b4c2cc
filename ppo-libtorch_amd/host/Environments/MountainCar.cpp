#include "MountainCar.h"

#include <cmath>
#include <random>

MountainCar::MountainCar(std::shared_ptr<ppo::Device> device, int64_t seed, int64_t env_index)
    : min_position(-1.2f), max_position(0.6f), max_speed(0.07f), goal_position(0.5f), goal_velocity(0.0f), force(0.001f), gravity(0.0025f),
      low{ -1.2f, -0.07f }, high{ 0.6f, 0.07f }, episode_length(0), episode_reward(0.0f), m_device(std::move(device)), m_seed(seed),
      m_env_index(env_index) {
    if (!m_device) m_device = std::make_shared<ppo::Device>(0);
}

std::tuple<std::vector<float>, float, bool, bool> MountainCar::step(const int64_t& action) {
    using ppo::Tensor;
    Tensor s = Tensor::from_host<float>(m_device, state, { 1, 2 });
    Tensor a = Tensor::from_host<int64_t>(m_device, { action }, { 1 });
    Tensor ns(m_device, { 1, 2 }, ppo::DType::f32), r(m_device, { 1 }, ppo::DType::f32), t(m_device, { 1 }, ppo::DType::i32);
    ppo::check(ppo_env_transition(PPO_ENV_MOUNTAINCAR, s.data<float>(), a.data<int64_t>(), 1, ns.data<float>(), r.data<float>(), t.data<int32_t>(),
                                  m_device->stream()),
               m_device->util(), "MountainCar::step");
    state = ns.cpu<float>();
    const float reward = r.item<float>();
    episode_length += 1;
    episode_reward += reward;
    return { state, reward, t.item<int32_t>() != 0, false };
}

// The reference seeds a fresh std::mt19937 from std::random_device on every reset (MountainCar.cpp:79-88): not reproducible.
// Here the draw is a deterministic function of (seed, env index, reset count) through the same libstdc++ uniform mapping.
std::vector<float> MountainCar::randomUniform(float lo, float hi) {
    std::seed_seq seq{ static_cast<uint32_t>(m_seed), static_cast<uint32_t>(m_env_index), static_cast<uint32_t>(m_resets++) };
    std::mt19937 gen(seq);
    std::uniform_real_distribution<float> dis(lo, hi);
    return { dis(gen), 0.0f };
}

std::vector<float> MountainCar::reset() {
    state = randomUniform(-0.6f, -0.4f);
    episode_length = 0;
    episode_reward = 0.0f;
    return state;
}

ppo::Tensor MountainCar::getActionMask() { return ppo::Tensor::from_host<uint8_t>(m_device, { 1, 1, 1 }, { 3 }); }
