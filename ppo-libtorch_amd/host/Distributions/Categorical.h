// Drop-in for the reference's Distributions/Categorical.h:11-22 on the HIP path (same method set, ppo::Tensor for torch::Tensor).
#pragma once
#include "../Tensor.h"

class Categorical {
  public:
    Categorical() : m_num_events(0) {}
    // Categorical.cpp:28-39: m_logits = logits - logsumexp(logits), m_probs = softmax(logits)
    Categorical(const ppo::Tensor& logits, std::shared_ptr<ppo::Device> device);
    ppo::Tensor logits_to_probs(ppo::Tensor logits, bool is_binary = false);
    ppo::Tensor sample();                       // :73-79 (own counter-based generator)
    ppo::Tensor log_prob(ppo::Tensor value);    // :92-101
    ppo::Tensor entropy();                      // :112-119, including the clamp(min = FLT_MIN) behaviour
    ppo::Tensor mean();                         // :126-131 NaN
    ppo::Tensor mode();                         // :139-141 argmax
    ppo::Tensor variance();                     // :148-153 NaN
    ppo::Tensor enumerate_support();            // :162-166

    ppo::Tensor m_logits, m_probs;
    int64_t m_num_events;
    std::shared_ptr<ppo::Device> m_device;
    int64_t m_seed = 1, m_draws = 0;            // sampling stream position

  protected:
    Categorical(const ppo::Tensor& logits, const ppo::Tensor* masks, std::shared_ptr<ppo::Device> device, int dist_kind);
    int m_kind = PPO_DIST_CATEGORICAL;
    ppo::Tensor m_entropy, m_mode, m_raw_logits, m_masks_u8;
};
