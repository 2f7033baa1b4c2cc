// Drop-in for the reference's Environments/MountainCar.h (MountainCar-v0 physics, 3 actions, all-ones action mask).
#pragma once
#include <tuple>
#include <vector>

#include "../Tensor.h"

class MountainCar {
  public:
    float min_position, max_position, max_speed, goal_position, goal_velocity, force, gravity;
    std::vector<float> low, high, state;
    std::vector<int64_t> actionSpace;
    int64_t episode_length;
    float episode_reward;

    explicit MountainCar(std::shared_ptr<ppo::Device> device = nullptr, int64_t seed = 1, int64_t env_index = 0);
    std::tuple<std::vector<float>, float, bool, bool> step(const int64_t& action);
    std::vector<float> reset();
    ppo::Tensor getActionMask();   // ones(3), MountainCar.cpp:69-77

  protected:
    std::vector<float> randomUniform(float low, float high);

  private:
    std::shared_ptr<ppo::Device> m_device;
    int64_t m_seed, m_env_index, m_resets = 0;
};
