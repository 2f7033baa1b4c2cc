// Drop-in for the reference's Environments/CartPole.h: one CartPole-v1 environment with the reference's duck-typed interface
// (reset() / step() / episode_length / episode_reward).  The transition runs on the GPU through the C-ABI
// (ppo_env_transition: bit-identical to CartPole::step, CartPole.cpp:47-94); the reset noise is the reference's
// std::mt19937(seed) + uniform_real_distribution<float>(-0.05, 0.05) stream (CartPole.h:28-29).  The vectorised path used by
// PPO_Discrete never goes through this per-object class: it steps all envs in one struct-of-arrays kernel.
#pragma once
#include <tuple>
#include <vector>

#include "../Tensor.h"

class CartPole {
  public:
    std::vector<float> state;   // [x, x_dot, theta, theta_dot]
    bool terminated;
    int64_t episode_length;
    float episode_reward;

    explicit CartPole(int64_t seed, std::shared_ptr<ppo::Device> device = nullptr);
    CartPole(const CartPole&) = delete;
    CartPole& operator=(const CartPole&) = delete;
    CartPole(CartPole&&) = default;
    CartPole& operator=(CartPole&&) = default;

    std::tuple<std::vector<float>, float, bool, bool> step(const int64_t& action);
    std::vector<float> reset();

  protected:
    float randomUniform();

  private:
    std::shared_ptr<ppo::Device> m_device;
    int64_t m_seed;
    int64_t m_draws = 0;              // position in the generator's stream
    std::vector<float> m_stream;      // first draws of mt19937(seed) mapped to U(-0.05, 0.05), grown on demand
};
