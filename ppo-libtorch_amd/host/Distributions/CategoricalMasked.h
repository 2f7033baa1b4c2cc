// Drop-in for the reference's Distributions/CategoricalMasked.h:12-23: invalid actions get logit -1e8 (CategoricalMasked.cpp:34-35)
// and are excluded from the entropy (:140-142).
#pragma once
#include "Categorical.h"

class CategoricalMasked : public Categorical {
  public:
    CategoricalMasked() = default;
    CategoricalMasked(const ppo::Tensor& logits, const ppo::Tensor& masks, std::shared_ptr<ppo::Device> device)
        : Categorical(logits, &masks, std::move(device), PPO_DIST_MASKED), m_masks(masks) {}
    ppo::Tensor m_masks;
};
