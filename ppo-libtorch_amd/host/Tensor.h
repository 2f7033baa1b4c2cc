// ppo-libtorch_amd/host/Tensor.h -- the device-buffer view type that stands where torch::Tensor stands in the reference's
// signatures (PPO/PPO_Discrete.h:35-48, PPO/Agent.h:36-38, Distributions/Categorical.h:12-22).  Not a tensor library: shape,
// dtype, a device pointer, and copies to/from host.  All device work goes through the C-ABI (include/ppo_hip.h).
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "ppo_hip.h"

namespace ppo {

enum class DType { f32, i64, i32, u8 };
inline size_t dtype_size(DType d) { return d == DType::f32 || d == DType::i32 ? 4 : (d == DType::i64 ? 8 : 1); }

// Throws std::runtime_error carrying ppo_last_error(), the way the reference surfaces failures (PPO_Discrete.cpp:370-375,
// caught in driver.cpp:15-17).
inline void check(ppo_status st, const ppo_ctx* ctx, const char* what) {
    if (st != PPO_OK) throw std::runtime_error(std::string(what) + ": " + ppo_last_error(ctx));
}

// Stands for torch::Device / m_device: one GPU, with a small utility context for allocations, copies and stateless kernels.
class Device {
  public:
    explicit Device(int ordinal = 0) : m_ordinal(ordinal) {
        ppo_config c{};
        c.struct_size = sizeof c;
        c.device = ordinal;
        c.env_kind = PPO_ENV_CARTPOLE; c.dist_kind = PPO_DIST_CATEGORICAL; c.obs_size = 4; c.n_heads = 1; c.head_dims[0] = 2;
        c.hidden = 64; c.n_hidden = 2; c.num_envs = 1; c.num_steps = 1; c.num_minibatches = 1; c.update_epochs = 1;
        c.max_episode_steps = 500; c.use_gae = 1; c.seed = 1; c.learning_rate = 1e-3f; c.gamma = 0.99f; c.gae_lambda = 0.95f;
        c.clip_coef = 0.2f; c.vf_coef = 0.5f; c.max_grad_norm = 0.5f;
        check(ppo_ctx_create(&c, &m_util), nullptr, "ppo::Device");
    }
    ~Device() { ppo_ctx_destroy(m_util); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    int ordinal() const { return m_ordinal; }
    ppo_ctx* util() const { return m_util; }
    void* stream() const { return ppo_stream(m_util); }
    void sync() const { check(ppo_sync(m_util), m_util, "sync"); }

  private:
    int m_ordinal;
    ppo_ctx* m_util = nullptr;
};

class Tensor {
  public:
    Tensor() = default;
    Tensor(std::shared_ptr<Device> dev, std::vector<int64_t> shape, DType dt) : m_dev(std::move(dev)), m_shape(std::move(shape)), m_dtype(dt), m_has_shape(true) {
        void* p = nullptr;
        check(ppo_device_alloc(m_dev->util(), nbytes(), &p), m_dev->util(), "Tensor alloc");
        std::shared_ptr<Device> keep = m_dev;
        m_data = std::shared_ptr<void>(p, [keep](void* q) { ppo_device_free(keep->util(), q); });
    }
    // non-owning view of a context buffer (e.g. m_obs = PPO_BUF_OBS)
    static Tensor view(std::shared_ptr<Device> dev, void* ptr, std::vector<int64_t> shape, DType dt) {
        Tensor t;
        t.m_dev = std::move(dev); t.m_shape = std::move(shape); t.m_dtype = dt; t.m_has_shape = true;
        t.m_data = std::shared_ptr<void>(ptr, [](void*) {});
        return t;
    }
    template <class T>
    static Tensor from_host(std::shared_ptr<Device> dev, const std::vector<T>& v, std::vector<int64_t> shape);
    // a float scalar that lives on the HOST (the loss scalars train() hands to printPPOResults were read back with the statistics already:
    // wrapping each in a device allocation + copy + read-back cost ~0.3 ms per update).  item<float>() / cpu<float>() return it without a device call.
    static Tensor host_scalar(float v) {
        Tensor t;
        t.m_shape = { 1 }; t.m_dtype = DType::f32; t.m_has_shape = true; t.m_on_host = true;
        t.m_data = std::shared_ptr<void>(new float(v), [](void* q) { delete static_cast<float*>(q); });
        return t;
    }

    bool defined() const { return static_cast<bool>(m_data); }
    int64_t numel() const { return m_has_shape ? std::accumulate(m_shape.begin(), m_shape.end(), int64_t{1}, std::multiplies<int64_t>()) : 0; }
    const std::vector<int64_t>& sizes() const { return m_shape; }
    int64_t size(int i) const { return m_shape.at(i < 0 ? m_shape.size() + i : i); }
    DType dtype() const { return m_dtype; }
    size_t nbytes() const { return static_cast<size_t>(numel()) * dtype_size(m_dtype); }
    void* data_ptr() const { return m_data.get(); }
    template <class T> T* data() const { return static_cast<T*>(m_data.get()); }
    const std::shared_ptr<Device>& device() const { return m_dev; }
    Tensor reshape(std::vector<int64_t> shape) const { Tensor t = *this; t.m_shape = std::move(shape); return t; }

    template <class T> std::vector<T> cpu() const {
        std::vector<T> out(static_cast<size_t>(numel()));
        if (sizeof(T) != dtype_size(m_dtype)) throw std::runtime_error("Tensor::cpu: element size mismatch");
        if (m_on_host) { std::memcpy(out.data(), m_data.get(), nbytes()); return out; }
        if (numel()) check(ppo_memcpy_d2h(m_dev->util(), out.data(), m_data.get(), nbytes()), m_dev->util(), "Tensor::cpu");
        return out;
    }
    template <class T> T item() const { return cpu<T>().at(0); }
    template <class T> void copy_from_host(const std::vector<T>& v) {
        if (v.size() * sizeof(T) != nbytes()) throw std::runtime_error("Tensor::copy_from_host: size mismatch");
        if (m_on_host) { std::memcpy(m_data.get(), v.data(), nbytes()); return; }
        if (numel()) check(ppo_memcpy_h2d(m_dev->util(), m_data.get(), v.data(), nbytes()), m_dev->util(), "Tensor::copy_from_host");
    }

  private:
    std::shared_ptr<Device> m_dev;
    std::shared_ptr<void> m_data;
    std::vector<int64_t> m_shape;
    DType m_dtype = DType::f32;
    bool m_on_host = false;     // host_scalar(): the bytes are host memory
    bool m_has_shape = false;   // a default-constructed Tensor is the reference's undefined torch::Tensor(): numel() == 0
};

template <class T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr DType value = DType::f32; };
template <> struct dtype_of<int64_t> { static constexpr DType value = DType::i64; };
template <> struct dtype_of<int32_t> { static constexpr DType value = DType::i32; };
template <> struct dtype_of<uint8_t> { static constexpr DType value = DType::u8; };

template <class T>
Tensor Tensor::from_host(std::shared_ptr<Device> dev, const std::vector<T>& v, std::vector<int64_t> shape) {
    Tensor t(std::move(dev), std::move(shape), dtype_of<T>::value);
    t.copy_from_host(v);
    return t;
}

}  // namespace ppo
