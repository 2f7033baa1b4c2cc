// Drop-in for the reference's PPO/PPO_MultiDiscrete.h: masked (multi-)categorical heads on MountainCar envs.
#pragma once
#include "PPOAlgorithm.h"
#include "../Environments/MountainCar.h"

class PPO_MultiDiscrete : public PPOAlgorithm {
  public:
    PPO_MultiDiscrete();
    AgentOutput computeActionLogic(const ppo::Tensor& next_obs, const ppo::Tensor& action_mask, const ppo::Tensor& action = ppo::Tensor());  // PPO_MultiDiscrete.cpp:271-277
    ppo::Tensor initEnvs(const ppo::Tensor& action_mask);                // :380-423 (fills the mask with ones)
};
