// Drop-in for the reference's PPO/PPO_Discrete.h: PPO with a single categorical head on CartPole envs.
#pragma once
#include "PPOAlgorithm.h"
#include "../Environments/CartPole.h"

class PPO_Discrete : public PPOAlgorithm {
  public:
    PPO_Discrete();
    AgentOutput computeActionLogic(const ppo::Tensor& next_obs) const;   // PPO_Discrete.cpp:258-263
    ppo::Tensor initEnvs();                                              // :365-402
};
