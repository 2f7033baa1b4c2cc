#include "Categorical.h"

#include <cmath>
#include <limits>

using ppo::DType;
using ppo::Tensor;

Categorical::Categorical(const Tensor& logits, std::shared_ptr<ppo::Device> device) : Categorical(logits, nullptr, std::move(device), PPO_DIST_CATEGORICAL) {}

Categorical::Categorical(const Tensor& logits, const Tensor* masks, std::shared_ptr<ppo::Device> device, int dist_kind) {
    if (logits.sizes().size() != 2 || logits.dtype() != DType::f32) throw std::runtime_error("Categorical: logits must be f32 [n, events]");
    m_device = std::move(device);
    m_kind = dist_kind;
    m_raw_logits = logits;
    const int64_t n = logits.size(0), A = logits.size(1);
    m_num_events = A;
    m_logits = Tensor(m_device, { n, A }, DType::f32);
    m_probs = Tensor(m_device, { n, A }, DType::f32);
    m_entropy = Tensor(m_device, { n }, DType::f32);
    m_mode = Tensor(m_device, { n }, DType::i64);
    if (masks) m_masks_u8 = *masks;  // u8 [n, A]
    ppo::check(ppo_categorical(m_kind, logits.data<float>(), masks ? masks->data<uint8_t>() : nullptr, nullptr, n, static_cast<int32_t>(A),
                               m_logits.data<float>(), m_probs.data<float>(), nullptr, m_entropy.data<float>(), m_mode.data<int64_t>(),
                               m_device->stream()),
               m_device->util(), "Categorical");
}

Tensor Categorical::logits_to_probs(Tensor logits, bool is_binary) {
    if (is_binary) throw std::runtime_error("logits_to_probs: the binary (sigmoid) form is never used on the PPO path");
    return Categorical(logits, m_device).m_probs;
}

Tensor Categorical::sample() {
    const int64_t n = m_probs.size(0);
    Tensor out(m_device, { n }, DType::i64);
    ppo::check(ppo_categorical_sample(m_probs.data<float>(), n, static_cast<int32_t>(m_num_events), m_seed, 0, m_draws++, 0, out.data<int64_t>(),
                                      m_device->stream()),
               m_device->util(), "Categorical::sample");
    return out;
}

Tensor Categorical::log_prob(Tensor value) {
    const int64_t n = m_probs.size(0);
    if (value.dtype() != DType::i64 || value.numel() != n) throw std::runtime_error("log_prob: value must be i64 [n]");
    Tensor out(m_device, { n }, DType::f32);
    ppo::check(ppo_categorical(m_kind, m_raw_logits.data<float>(), m_masks_u8.defined() ? m_masks_u8.data<uint8_t>() : nullptr, value.data<int64_t>(), n,
                               static_cast<int32_t>(m_num_events), nullptr, nullptr, out.data<float>(), nullptr, nullptr, m_device->stream()),
               m_device->util(), "Categorical::log_prob");
    return out;
}

Tensor Categorical::entropy() { return m_entropy; }
Tensor Categorical::mode() { return m_mode; }

Tensor Categorical::mean() {
    return Tensor::from_host<float>(m_device, std::vector<float>(static_cast<size_t>(m_probs.size(0)), std::numeric_limits<float>::quiet_NaN()), { m_probs.size(0) });
}
Tensor Categorical::variance() { return mean(); }

Tensor Categorical::enumerate_support() {
    const int64_t n = m_probs.size(0);
    std::vector<int64_t> v(static_cast<size_t>(m_num_events * n));
    for (int64_t e = 0; e < m_num_events; e++)
        for (int64_t i = 0; i < n; i++) v[static_cast<size_t>(e * n + i)] = e;
    return Tensor::from_host<int64_t>(m_device, v, { m_num_events, n });
}
