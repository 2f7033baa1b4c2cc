// ppo-libtorch_amd/csrc/ppo_internal.hpp -- shared host/device definitions of libppo_hip.so (gfx950 only).
//
// Number model of the parity-critical paths (env step, reset mapping, GAE, AdamW element-wise): IEEE binary32 with
// separately rounded operations.  The whole library is compiled with -ffp-contract=off; every fused multiply-add in
// the MLP paths is an explicit __builtin_fmaf.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "ppo_hip.h"

#define PPO_HIDDEN 64          // one hidden unit per lane of a 64-wide wavefront
#define PPO_MAX_OBS 8
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) acts on the CURRENT device's copy of a kernel: a call site remembers, per device ordinal, that it
// has been made (one bit each; `done` is the site's own static).  Thread-safe: the worst a race does is set the attribute twice.
#include <atomic>
inline hipError_t allow_dynamic_lds(std::atomic<unsigned long long>& done, const void* kernel, int bytes = 160 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

#define PPO_MAX_ACT 32         // sum of head widths
#define PPO_TILE 64            // samples per update tile
#define PPO_LDS_STRIDE 68      // padded row stride (floats) of [unit][sample] LDS tiles: 16-B aligned, bank-spreading

// ---------------------------------------------------------------------------------------------------------
// Flat parameter layout = Agent::parameters() order (reference PPO/Agent.cpp:65-66): critic {W1,b1,W2,b2,W3,b3}, actor {..}
// ---------------------------------------------------------------------------------------------------------
struct NetLayout {
    int obs, act, n_heads;
    int head_dims[PPO_MAX_HEADS];
    int w1[2], b1[2], w2[2], b2[2], w3[2], b3[2];  // offsets (floats) per net: 0 = critic, 1 = actor
    int net_off[2], net_size[2];
    int P;
    int n_tensors;
    int tensor_off[13];                             // 12 tensors + end
};

inline NetLayout make_layout(int obs, int n_heads, const int* head_dims) {
    NetLayout L{};
    L.obs = obs;
    L.n_heads = n_heads;
    L.act = 0;
    for (int h = 0; h < n_heads; h++) { L.head_dims[h] = head_dims[h]; L.act += head_dims[h]; }
    int o = 0, t = 0;
    for (int net = 0; net < 2; net++) {
        const int out = net == 0 ? 1 : L.act;
        L.net_off[net] = o;
        L.tensor_off[t++] = o; L.w1[net] = o; o += PPO_HIDDEN * obs;
        L.tensor_off[t++] = o; L.b1[net] = o; o += PPO_HIDDEN;
        L.tensor_off[t++] = o; L.w2[net] = o; o += PPO_HIDDEN * PPO_HIDDEN;
        L.tensor_off[t++] = o; L.b2[net] = o; o += PPO_HIDDEN;
        L.tensor_off[t++] = o; L.w3[net] = o; o += out * PPO_HIDDEN;
        L.tensor_off[t++] = o; L.b3[net] = o; o += out;
        L.net_size[net] = o - L.net_off[net];
    }
    L.tensor_off[t] = o;
    L.n_tensors = t;
    L.P = o;
    return L;
}

// Hyper-parameters as the kernels consume them.
struct LossParams {
    float clip_coef, ent_coef, vf_coef;
    int norm_adv, clip_vloss;
    int dist_kind;
};

// Per-minibatch advantage statistics (sum, sum of squares over the GLOBAL minibatch) and the scalars derived from them.
struct AdvStat { double s1, s2; };
#ifndef PPO_ADV_PARTS
#define PPO_ADV_PARTS 32  // partial sums per minibatch (one workgroup each), added in order by the consumer (A/B on one box, tools/ab.sh: 32 parts 56.7 us per update launch and 161.7 M env-steps/s, 8 parts 57.9 us and 158.5 M)
#endif
#define PPO_EV_BLOCKS 512 // partial rows of the explained-variance sums, added in order by the host
// Job-global statistics block of a sharded run (doubles; summed over ranks by the per-update all-reduce, every rank writes only its own slot, so the
// "sum" is a gather): the reference prints ONE table for the job (PPO_Discrete.cpp:474-480, 647-648, 700-774), so every rank must hold the same numbers.
//   [0..3]   explained-variance sums {sum y, sum y^2, sum d, sum d^2} of this rank's shard (added over ranks = the global sums)
//   [4..7]   reserved (zero)
//   rank r:  [8 + r * PPO_GSTAT_RANK + 0] finished episodes so far, [.. + 1] entries in this rank's ring, [.. + 4 + 3 i + {0, 1, 2}] = {key, length, reward}
//            of ring entry i; key = (absolute rollout step) * global_num_envs + global env index = the position of the episode in the reference's
//            push order, so the host rebuilds the job's CircularBuffer(100) exactly: the newest 100 keys of the union.
#define PPO_GSTAT_HEAD 8
#define PPO_GSTAT_RANK (4 + 3 * 100)
#define PPO_GSTAT_DOUBLES (PPO_GSTAT_HEAD + 8 * PPO_GSTAT_RANK)

// Device-side record of one optimizer step's scalars (doubles so the host reads them as-is).
struct StepStats {
    double pg_loss, v_loss, entropy_loss, approx_kl, clipfrac, loss, total_norm, pad;
};

// Host-computed AdamW scalars of one step (LibTorch optim/adamw.cpp narrows its double scalars to float this way).
struct AdamCoef { float decay, neg_step, sqrt_bc2, pad; };

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 (build's own counter-based generator; mirrored bit-for-bit by oracle/ppo_oracle.c:orc_philox4x32)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// libstdc++ uniform_real_distribution<float>(a,b) on one 32-bit word (reference Environments/CartPole.cpp:96-100).
__device__ __forceinline__ float canon_to_uniform(uint32_t u, float a, float b) {
    float r = (float)u / 4294967296.0f;
    if (r >= 1.0f) r = 0x1.fffffep-1f;
    return r * (b - a) + a;
}

// ---------------------------------------------------------------------------------------------------------
// glibc 2.35 sinf/cosf (x86-64 FMA ifunc variant), evaluated in binary64 with explicit fma -- what std::sin/std::cos
// on float resolve to in the reference (Environments/CartPole.cpp:59-60, MountainCar.cpp:34).  Valid for |x| < 120;
// the environments stay far inside.  Mirrors oracle/ppo_oracle.c:orc_sinf/orc_cosf, which is pinned exhaustively
// against the host libm.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sc_poly(double x, double x2, int n, bool neg) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, S3, S2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(x7, s1, s);
    } else {
        const double sg = neg ? -1.0 : 1.0;
        const double x4 = x2 * x2;
        const double c2 = __builtin_fma(x2, sg * C4, sg * C3);
        const double c1 = __builtin_fma(x2, sg * C1, sg * C0);
        const double x6 = x4 * x2;
        const double c = __builtin_fma(x4, sg * C2, c1);
        return (float)__builtin_fma(x6, c2, c);
    }
}
__device__ __forceinline__ uint32_t abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
__device__ __forceinline__ double sc_reduce_fast(double x, int& n) {
    const double r = x * 0x1.45F306DC9C883p+23;
    n = ((int32_t)r + 0x800000) >> 24;
    return __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
}
__device__ __forceinline__ float glibc_sinf(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sc_poly(x, x * x, 0, false);
    }
    int n;
    x = sc_reduce_fast(x, n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sc_poly(x * s, x * x, n, (n & 2) != 0);
}
__device__ __forceinline__ float glibc_cosf(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sc_poly(x, x * x, 1, false);
    }
    int n;
    x = sc_reduce_fast(x, n);
    const int m = (n + 1) & 3;
    const double s = (m == 1 || m == 2) ? -1.0 : 1.0;
    return sc_poly(x * s, x * x, n ^ 1, ((n + 1) & 2) != 0);
}

// ---------------------------------------------------------------------------------------------------------
// Environment transitions
// ---------------------------------------------------------------------------------------------------------
// CartPole::step, reference Environments/CartPole.cpp:47-94 (constants :6-17).  st = {x, x_dot, theta, theta_dot}.
// The part of CartPole::step behind the trigonometry: everything that depends on the action.  cartpole_step = sin / cos of the state + this; the
// fused rollout forms sin / cos ONCE per step and runs the tail for every possible action before the policy has chosen (kernels_rollout.hip).
__device__ __forceinline__ float cartpole_tail(float* st, int action, float cos_theta, float sin_theta, int& terminated) {
    const float gravity = 9.8f, mass_pole = 0.1f, total_mass = 0.1f + 1.0f, length = 0.5f;
    const float polemass_length = 0.1f * 0.5f, force_mag = 10.0f, tau = 0.02f;
    const float theta_thr = 0x1.aceeap-3f; /* (float)(12*2*M_PI/360) = bits 0x3e567750, CartPole.cpp:16 */
    const float x_thr = 2.4f;
    float x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
    float force = force_mag;
    if (action == 0) force = -force;
    const float temp = (force + polemass_length * theta_dot * theta_dot * sin_theta) / total_mass;
    const float theta_acc = (gravity * sin_theta - cos_theta * temp) /
                            (length * (4.0f / 3.0f - mass_pole * cos_theta * cos_theta / total_mass));
    const float x_acc = temp - polemass_length * theta_acc * cos_theta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * x_acc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * theta_acc;
    st[0] = x; st[1] = x_dot; st[2] = theta; st[3] = theta_dot;
    terminated = (x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr) ? 1 : 0;
    return terminated ? -1.0f : 1.0f;
}
// Position and angle of CartPole's next state, and with them the termination test, do NOT depend on the action: the explicit Euler step advances x and theta
// with the OLD velocities (CartPole.cpp:76-80; only x_dot and theta_dot see the force).  Same expressions as cartpole_tail, bit for bit; the fused rollout's
// trigonometry wave uses this to form sin / cos of the NEXT state's angle a whole step ahead (kernels_rollout.hip).
__device__ __forceinline__ void cartpole_next_pose(const float* st, float& x1, float& theta1, int& terminated) {
    const float tau = 0.02f, theta_thr = 0x1.aceeap-3f, x_thr = 2.4f;
    x1 = st[0] + tau * st[1];
    theta1 = st[2] + tau * st[3];
    terminated = (x1 < -x_thr || x1 > x_thr || theta1 < -theta_thr || theta1 > theta_thr) ? 1 : 0;
}
__device__ __forceinline__ float cartpole_step(float* st, int action, int& terminated) {
    const float cos_theta = glibc_cosf(st[2]), sin_theta = glibc_sinf(st[2]);
    return cartpole_tail(st, action, cos_theta, sin_theta, terminated);
}

// MountainCar::step, reference Environments/MountainCar.cpp:29-57 (constants :5-13).  st = {position, velocity}.
__device__ __forceinline__ float mountaincar_tail(float* st, int action, float cos3p, int& terminated) {
    const float min_position = -1.2f, max_position = 0.6f, max_speed = 0.07f, goal_position = 0.5f, goal_velocity = 0.0f;
    const float force = 0.001f, gravity = 0.0025f;
    float position = st[0], velocity = st[1];
    velocity += ((float)action - 1.0f) * force + cos3p * (-gravity);
    velocity = velocity < -max_speed ? -max_speed : (velocity > max_speed ? max_speed : velocity);
    position += velocity;
    position = position < min_position ? min_position : (position > max_position ? max_position : position);
    if (position == min_position && velocity < 0.0f) velocity = 0.0f;
    terminated = (position >= goal_position && velocity >= goal_velocity) ? 1 : 0;
    st[0] = position; st[1] = velocity;
    return -1.0f;
}
__device__ __forceinline__ float mountaincar_step(float* st, int action, int& terminated) {
    return mountaincar_tail(st, action, glibc_cosf(3.0f * st[0]), terminated);
}

template <int ENV>
__device__ __forceinline__ float env_step(float* st, int action, int& terminated) {
    if constexpr (ENV == PPO_ENV_CARTPOLE) return cartpole_step(st, action, terminated);
    else return mountaincar_step(st, action, terminated);
}
// env_step in two halves: what depends on the state only (tr[0], tr[1]) ...
template <int ENV>
__device__ __forceinline__ void env_step_pre(const float* st, float* tr) {
    if constexpr (ENV == PPO_ENV_CARTPOLE) { tr[0] = glibc_cosf(st[2]); tr[1] = glibc_sinf(st[2]); }
    else { tr[0] = glibc_cosf(3.0f * st[0]); tr[1] = 0.0f; }
}
// ... and what depends on the action: env_step(st, a, t) == env_step_pre(st, tr) then env_step_tail(st, a, tr, t), bit for bit
template <int ENV>
__device__ __forceinline__ float env_step_tail(float* st, int action, const float* tr, int& terminated) {
    if constexpr (ENV == PPO_ENV_CARTPOLE) return cartpole_tail(st, action, tr[0], tr[1], terminated);
    else return mountaincar_tail(st, action, tr[0], terminated);
}

// CartPole::reset (CartPole.cpp:34-45): the k-th reset of ANY env reads words 4k..4k+3 of the one stream all envs share
// (every env is constructed with the same seed, PPO_Discrete.cpp:84-86).  MountainCar::reset (MountainCar.cpp:59-66,79-88)
// draws from std::random_device in the reference; the build keys it: philox(seed; global env, reset#, 0, 1).x.
template <int ENV>
__device__ __forceinline__ void env_reset(float* st, const float* __restrict__ reset_table, int k, int64_t seed, int64_t env_global) {
    if constexpr (ENV == PPO_ENV_CARTPOLE) {
        const float4 r = reinterpret_cast<const float4*>(reset_table)[k];
        st[0] = r.x; st[1] = r.y; st[2] = r.z; st[3] = r.w;
    } else {
        const uint4 w = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env_global, (uint32_t)k, 0u, 1u);
        st[0] = canon_to_uniform(w.x, -0.6f, -0.4f);
        st[1] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Categorical / CategoricalMasked over one head of width A held in a small array (uniform across the wave or per-thread).
// reference Distributions/Categorical.cpp:28-39,92-119; CategoricalMasked.cpp:31-46,107-144.
//   z[a] in: raw logits; out: m_logits (log-probs).  p[a] out: m_probs.  returns entropy.
// ---------------------------------------------------------------------------------------------------------
template <int DIST>
__device__ __forceinline__ float categorical_head(float* z, float* p, const uint8_t* mask, int A) {
    float mx = -INFINITY;
    for (int a = 0; a < A; a++) {
        if (DIST == PPO_DIST_MASKED && mask && !mask[a]) z[a] = -1e8f;
        mx = z[a] > mx ? z[a] : mx;
    }
    float se = 0.0f;
    for (int a = 0; a < A; a++) { p[a] = expf(z[a] - mx); se += p[a]; }
    const float lse = logf(se) + mx;
    float ent = 0.0f;
    for (int a = 0; a < A; a++) {
        z[a] = z[a] - lse;
        p[a] = p[a] / se;
        if (DIST == PPO_DIST_CATEGORICAL) {
            const float l = z[a] > 1.17549435e-38f ? z[a] : 1.17549435e-38f;  // torch::clamp(m_logits, FLT_MIN), Categorical.cpp:115
            ent += l * p[a];
        } else {
            const float plp = z[a] * p[a];
            ent += (mask == nullptr || mask[a]) ? plp : 0.0f;
        }
    }
    return -ent;
}

// Inverse-CDF draw on m_probs with u in [0,1) (mirrors oracle/ppo_oracle.c:orc_act).
__device__ __forceinline__ int sample_head(const float* p, int A, float u) {
    int a = 0, last = 0;
    float acc = 0.0f;
    bool hit = false;
    for (int k = 0; k < A; k++) {
        if (p[k] > 0.0f) last = k;
        acc += p[k];
        if (!hit && u < acc) { a = k; hit = true; }
    }
    return hit ? a : last;
}

// tanh for the update kernels: branch-free, ~3 ULP.  |x| >= 0.12: (1 - t) / (1 + t) with t = exp(-2|x|) on the hardware
// exp2 / rcp (v_exp_f32, v_rcp_f32); below that the odd Taylor polynomial to x^7 (truncation < 2e-9 relative at 0.12) avoids the
// cancellation in 1 - t.  The clamp keeps the exp2 argument inside the range where v_exp_f32 needs no denormal pre-scaling.
// exp / log on the hardware exp2 / log2 units (v_exp_f32, v_log_f32; ~1 ULP of the base-2 result).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(fmaxf(x, -100.0f) * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// tanh on the transcendental pipe: 1 - 2 / (1 + e^(2x)), five instructions (v_mul, v_exp, v_add, v_rcp, v_fma), absolute error
// <= ~2e-7 (the library tanhf costs ~40 and a sign-symmetric form 14; 64 tanh per lane per tile made them half of the kernel's
// vector instructions).  Saturates correctly: e = inf -> 1, e = 0 -> -1.
__device__ __forceinline__ float tanh_mufu(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fminf(fabsf(x), 20.0f);
    const float t = __builtin_amdgcn_exp2f(ax * -2.885390081777927f);   // exp(-2|x|)
    const float big = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    const float x2 = x * x;
    const float poly = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    const float small = x * poly;
    const float bigs = __builtin_copysignf(big, x);
    return ax < 0.12f ? small : bigs;
}

// categorical_head on the hardware exp2 / log2 / rcp units (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ULP each; log-probs within ~3e-7 of the
// library version): for kernels where one wave evaluates the heads on a step's critical path.
template <int DIST>
__device__ __forceinline__ float categorical_head_fast(float* z, float* p, const uint8_t* mask, int A) {
    float mx = -INFINITY;
    for (int a = 0; a < A; a++) {
        if (DIST == PPO_DIST_MASKED && mask && !mask[a]) z[a] = -1e8f;
        mx = z[a] > mx ? z[a] : mx;
    }
    float se = 0.0f;
    for (int a = 0; a < A; a++) { p[a] = fast_exp(z[a] - mx); se += p[a]; }
    const float lse = fast_log(se) + mx;
    const float rse = __builtin_amdgcn_rcpf(se);
    float ent = 0.0f;
    for (int a = 0; a < A; a++) {
        z[a] = z[a] - lse;
        p[a] = p[a] * rse;
        if (DIST == PPO_DIST_CATEGORICAL) {
            const float l = z[a] > 1.17549435e-38f ? z[a] : 1.17549435e-38f;
            ent += l * p[a];
        } else {
            const float plp = z[a] * p[a];
            ent += (mask == nullptr || mask[a]) ? plp : 0.0f;
        }
    }
    return -ent;
}

// Wave-wide float sum, result in every lane, on the DPP cross-lane network (no LDS crossbar round trips, which cost ~100+
// cycles each and are on the rollout's per-step dependency chain): butterfly inside each row of 16 lanes with
// quad_perm / row_half_mirror / row_mirror, then the four row sums are read through scalar registers.  Fixed order.
#define PPO_DPP_ADD(v, CTRL) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xf, 0xf, false)))
__device__ __forceinline__ float wave_sum(float v) {
    v = PPO_DPP_ADD(v, 0xB1);    // quad_perm:[1,0,3,2]
    v = PPO_DPP_ADD(v, 0x4E);    // quad_perm:[2,3,0,1]
    v = PPO_DPP_ADD(v, 0x141);   // row_half_mirror
    v = PPO_DPP_ADD(v, 0x140);   // row_mirror: every lane of a row now holds the row's sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The same sum on the DPP network (no ds_bpermute round trips: a double butterfly through LDS costs twelve of them); result in
// every lane, fixed order (inside rows of 16 by butterfly, then the four row sums in order).
#define PPO_DPP_ADD_D(v, CTRL)                                                                                              \
    do {                                                                                                                    \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), (CTRL), 0xf, 0xf, false);                         \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), (CTRL), 0xf, 0xf, false);                         \
        (v) += __hiloint2double(hi_, lo_);                                                                                  \
    } while (0)
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
    PPO_DPP_ADD_D(v, 0xB1);    // quad_perm:[1,0,3,2]
    PPO_DPP_ADD_D(v, 0x4E);    // quad_perm:[2,3,0,1]
    PPO_DPP_ADD_D(v, 0x141);   // row_half_mirror
    PPO_DPP_ADD_D(v, 0x140);   // row_mirror: every lane of a row now holds the row's sum
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        r[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * i), __builtin_amdgcn_readlane(__double2loint(v), 16 * i));
    return ((r[0] + r[1]) + r[2]) + r[3];
}

// ---------------------------------------------------------------------------------------------------------
// clip_grad_norm_ + AdamW::step (PPO_Discrete.cpp:640-641) pieces shared by the stand-alone optimizer kernel and the update kernel's
// prologue (deferred step): ONE definition, so the two paths are bit-identical.
// ---------------------------------------------------------------------------------------------------------
// Total gradient norm from the per-workgroup sums of squares: wave w (< 4) adds the partials of the workgroups that touch tensors w, w + 4,
// w + 8 in a fixed order (all three loads in flight together, sums on the DPP network).  Every thread of the workgroup must call it
// (one barrier inside); n2s: 12 doubles of shared memory.
__device__ __forceinline__ float opt_total_norm(const NetLayout& L, const double* __restrict__ partial, double* n2s, int tid) {
    // The wave index as a SCALAR (it is wave-uniform; the compiler cannot know): L.tensor_off[w + 4 j] is then a scalar load out of the kernel arguments.  Indexed
    // by the per-lane value it was a VECTOR load from the argument segment with a wait behind it, and the partials' loads depended on it: three such pairs in a
    // row, six dependent memory round trips in a 5 us kernel (round 6, found in the assembly).  Now: the three tensors' block ranges first, then EVERY load of the
    // wave's three tensors (two per lane and tensor cover 128 workgroups of the slab reduction: more than any tensor of the reference's shapes spans) in one
    // batch, then the sums -- in the same order as before (a lane's blocks in index order), so the same bits.
    // Loads are UNCONDITIONAL with clamped addresses and the values selected afterwards: a load inside a divergent branch makes the compiler wait for it at
    // the join.  The thirteen offsets come as one batch of scalar loads and are picked by selects over a fixed bound, not by a dependent indexed load each.
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (w < 4) {
        int off[13];
#pragma unroll
        for (int t = 0; t < 13; t++) off[t] = L.tensor_off[t];
        int blo[3], bhi[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int t = w + 4 * j;
            int lo = 0, hi = 0;
#pragma unroll
            for (int tt = 0; tt < 12; tt++) { lo = tt == t ? off[tt] : lo; hi = tt == t ? off[tt + 1] : hi; }
            const bool have = t < L.n_tensors;
            blo[j] = have ? lo / 64 : 1;
            bhi[j] = have ? (hi - 1) / 64 : 0;
        }
        double v0[3], v1[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int t = w + 4 * j, b = blo[j] + lane;
            const int b0 = b <= bhi[j] ? b : 0, b1 = b + 64 <= bhi[j] ? b + 64 : 0;
            v0[j] = partial[(size_t)b0 * 12 + t];
            v1[j] = partial[(size_t)b1 * 12 + t];
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int t = w + 4 * j, b = blo[j] + lane;
            double v = 0.0;
            v += b <= bhi[j] ? v0[j] : 0.0;
            v += b + 64 <= bhi[j] ? v1[j] : 0.0;
            if (bhi[j] - blo[j] >= 128)   // wave-uniform; no tensor of the reference's shapes gets here
                for (int bb = b + 128; bb <= bhi[j]; bb += 64) v += partial[(size_t)bb * 12 + t];
            const double r = wave_sum_d_dpp(v);
            if (lane == 0) n2s[t] = r;
        }
    }
    __syncthreads();
    // total = || (||g_1||, ..., ||g_12||) ||_2 with float per-tensor norms (clip_grad.h:58-70).  Lane t forms tensor t's norm -- ONE binary64 square
    // root per wave instead of twelve in a row on every lane (a binary64 sqrt is a ~40-instruction sequence) -- and the squares are added in tensor
    // order through scalar registers: the same operations in the same order as before, every thread gets the same bits.
    double sq = 0.0;
    if (lane < 12) { const float nrm = (float)sqrt(n2s[lane]); sq = (double)nrm * nrm; }
    double tot = 0.0;
#pragma unroll
    for (int t = 0; t < 12; t++)
        tot += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sq), t), __builtin_amdgcn_readlane(__double2loint(sq), t));
    return (float)sqrt(tot);
}
__device__ __forceinline__ float opt_clip_coef(float total, float max_norm) {   // clip_grad.h:76-78
    const float c = max_norm / (total + 1e-6f);
    return c > 1.0f ? 1.0f : c;
}
__device__ __forceinline__ void adamw_apply(float g, float c, const AdamCoef& k, float& p, float& m, float& v) {
    const float b1 = 0.9f, b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-5f;
    const float gc = g * c;
    const float pi = p * k.decay;
    const float mi = __builtin_fmaf(gc, omb1, m * b1);
    const float vi = __builtin_fmaf(omb2 * gc, gc, v * b2);
    const float denom = sqrtf(vi) / k.sqrt_bc2 + eps;
    p = pi + (k.neg_step * mi) / denom;
    m = mi;
    v = vi;
}
// printPPOResults' inputs of one step (PPO_Discrete.cpp:700-774) from its loss sums; one thread
__device__ __forceinline__ void opt_write_stats(const double* ls, double cf0, double cf1, double global_M, const LossParams& hp, float total,
                                                StepStats* stats_out, double* clipfrac_accum) {
    const float pg = (float)(ls[0] / global_M);
    const float vl = 0.5f * (float)(ls[4] / global_M);
    const float el = (float)(ls[1] / global_M);
    StepStats o;
    o.pg_loss = pg;
    o.v_loss = vl;
    o.entropy_loss = el;
    o.approx_kl = (float)(ls[2] / global_M);
    o.clipfrac = (float)ls[3] / (float)global_M;
    o.loss = (pg - hp.ent_coef * el) + vl * hp.vf_coef;
    o.total_norm = total;
    o.pad = 0.0;
    *stats_out = o;
    if (clipfrac_accum) { clipfrac_accum[0] = cf0 + o.clipfrac; clipfrac_accum[1] = cf1 + 1.0; }
}
#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------------------
// Host-side launcher prototypes (each defined next to its kernels)
// ---------------------------------------------------------------------------------------------------------
struct RolloutArgs {
    // network
    const float* params;
    NetLayout L;
    int dist_kind;
    // env
    int env_kind;
    int N, T;
    int max_episode_steps;
    int64_t seed, env_offset, step_base;  // step_base: global index of rollout step 0 (sampling stream position)
    float* env_state;      // [O,N]
    int32_t* ep_len;       // [N]
    float* ep_rew;         // [N]
    int32_t* reset_count;  // [N]
    const float* reset_table;
    int reset_cap;
    int32_t* error_flag;
    // rollout buffers
    float* obs;            // [T,N,O]
    int32_t* actions;      // [T,N,H]
    float* logprobs;       // [T,N]
    float* rewards;        // [T,N]
    float* dones;          // [T,N]
    float* values;         // [T,N]
    uint8_t* masks;        // [T,N,A] or null
    int32_t* fin_len;      // [T,N]
    float* fin_rew;        // [T,N]
    float* next_obs;       // [N,O]
    int32_t* next_done;    // [N]
    float* next_value;     // [N]
    const int64_t* forced_actions;  // [T,N,H] or null
    int vector_kernel;     // PPO_KERNEL_ROLLOUT_VECTOR: rollout2_kernel for the single-head shapes
};

hipError_t launch_rollout(const RolloutArgs& a, hipStream_t s);
hipError_t launch_values(const float* params, const NetLayout& L, const float* obs0, int64_t n0, float* out0, const float* obs1, int64_t n1,
                         float* out1, hipStream_t s);
hipError_t launch_env_reset(int env_kind, int N, int64_t seed, int64_t env_offset, float* env_state, int32_t* ep_len, float* ep_rew,
                            int32_t* reset_count, const float* reset_table, int reset_cap, float* next_obs, int32_t* next_done,
                            int32_t* error_flag, hipStream_t s);
hipError_t launch_env_step(int env_kind, int N, int H, int max_episode_steps, int64_t seed, int64_t env_offset, float* env_state,
                           int32_t* ep_len, float* ep_rew, int32_t* reset_count, const float* reset_table, int reset_cap,
                           const int64_t* action, float* obs, float* reward, int32_t* done, int32_t* error_flag, hipStream_t s);
hipError_t launch_env_transition(int env_kind, const float* state_in, const int64_t* action, int64_t n, float* next_state,
                                 float* reward, int32_t* terminated, hipStream_t s);
hipError_t launch_aos_to_soa(const float* aos, float* soa, int N, int O, bool to_soa, hipStream_t s);

hipError_t launch_policy_act(const float* params, const NetLayout& L, int dist_kind, const float* obs, const uint8_t* mask,
                             const int64_t* forced_action, int64_t n, int64_t seed, int64_t env_offset, int64_t step_index,
                             int64_t* action, float* logprob, float* entropy, float* value, bool value_only, hipStream_t s, bool as_rollout16 = false,
                             int32_t* error_flag = nullptr);   // as_rollout16: rollout16_kernel's arithmetic where that kernel serves the shape (policy_act16_serves)
bool policy_act16_serves(const NetLayout& L);
hipError_t launch_categorical(int dist_kind, const float* logits, const uint8_t* mask, const int64_t* value, int64_t n, int A,
                              float* m_logits, float* m_probs, float* log_prob, float* entropy, int64_t* mode, hipStream_t s);

hipError_t launch_categorical_sample(const float* probs, int64_t n, int A, int64_t seed, int64_t row_offset, int64_t step_index, int head,
                                     int64_t* out, hipStream_t s);

// error_flag (may be null: the stateless entry points): device word that gets PPO_ERRFLAG_GAE_PROTOCOL when a bounded wait of the time-pipelined scan runs out
// (the affected strip's outputs are NaN either way)
hipError_t launch_gae(const float* rewards, const float* values, const float* dones, const float* next_value,
                      const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* adv, float* ret,
                      int32_t* error_flag, hipStream_t s);
hipError_t launch_gae_fast(const float* rewards, const float* values, const float* dones, const float* next_value,
                           const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* adv, float* ret,
                           hipStream_t s);
hipError_t launch_nstep(const float* rewards, const float* values, const float* dones, const float* next_value,
                        const int32_t* next_done, int64_t T, int64_t N, float gamma, float* adv, float* ret, int32_t* error_flag, hipStream_t s);

struct UpdateArgs {
    const float* params;
    NetLayout L;
    LossParams hp;
    // flattened batch views [B,...]
    const float* obs;
    const int32_t* actions;  // [B,H]
    const uint8_t* masks;    // [B,A] or null
    const float* logprobs;
    const float* advantages;
    const float* returns;
    const float* values;
    const float* rec_critic; // [B][8] packed sample records of the matrix-core kernel (launch_pack_records); the vector kernel reads the views above
    const float* rec_actor;  // [B][8]
    const int32_t* idx;      // [M]
    int M;                   // local minibatch rows
    double inv_global_M;     // 1 / (rows of the global minibatch)
    const AdvStat* adv_stat; // global sums for this minibatch (nullptr when !norm_adv)
    const float4* adv_norm;  // { mean, 1 / (std + 1e-8), std (Bessel), 0 } of this minibatch's advantages, formed ONCE per update from adv_stat (launch_adv_norm)
    double global_M;
    float* slab;             // [n_blocks, P_net_max] partial gradients
    double* stat_slab;       // [n_blocks, 8] partial loss sums
    int n_blocks[2];         // workgroups of the critic / of the actor; slab row = (net ? n_blocks[0] : 0) + workgroup
    unsigned long long* stamps;  // diagnostic build only: [2 nets][12 phases] cycle sums of wave 0 / workgroup 0
    int32_t* error_flag;     // the context's device error word (bit 0: reset table exhausted; PPO_ERRFLAG_UPDATE_PROTOCOL: a bounded wait of the wave-specialised update kernel ran out)
    int single_wave;         // 1: the one-wave-per-tile matrix-core kernel even for the reference's two shapes (A/B; ppo_config.kernel_flags)
};
constexpr int32_t PPO_ERRFLAG_UPDATE_PROTOCOL = 2;
constexpr int32_t PPO_ERRFLAG_GAE_PROTOCOL = 16;   // gae_pipe_kernel: a bounded wait between its mover waves and its walker ran out (that strip's advantages / returns are NaN)
constexpr int32_t PPO_ERRFLAG_UPDATE_RANGE = 8;    // a matrix-core update kernel met an observation that does not fit its fp16 operand (|obs| >= 65 504)
constexpr int32_t PPO_ERRFLAG_ROLLOUT_RANGE = 4;   // rollout16_kernel: |W3| does not fit the fp16 operand (pre-scaled by 2^8)
constexpr int32_t PPO_ERRFLAG_SKIP_STEP = PPO_ERRFLAG_UPDATE_PROTOCOL | PPO_ERRFLAG_UPDATE_RANGE | PPO_ERRFLAG_GAE_PROTOCOL;   // the optimizer does not apply a step behind these
int update_blocks_per_net(int M);
hipError_t launch_minibatch_fwd_bwd(const UpdateArgs& a, hipStream_t s);
// matrix-core version of the same kernel; sum(head_dims) <= 4, obs in {2, 4}: fp32 carried as two fp16 terms, three
// v_mfma_f32_32x32x16_f16 products per fp32 product (fp32 accuracy).  Gathers from the packed records.
void update_blocks_mfma(int M, int n_blocks[2]);
hipError_t launch_minibatch_fwd_bwd_mfma(const UpdateArgs& a, hipStream_t s);
hipError_t launch_pack_records(const NetLayout& L, const float* obs, const int32_t* actions, const uint8_t* masks, const float* logprobs,
                               const float* advantages, const float* returns, const float* values, int64_t B, float* rec_critic, float* rec_actor,
                               double* ev_sums, int32_t* error_flag, hipStream_t s);
// fp16 ranges of the matrix-core kernels of the 2 x 64 layout:
//   wr_dev[0..2] = the maxima of |parameter| by class -- [PPO_WR_W3] both nets' output layers (rollout16_kernel carries 2^8 W3 as fp16: |W3| < 255),
//   [PPO_WR_W2] both nets' hidden-to-hidden weights (the matrix-core update kernels carry c W2 and products through its columns as fp16 terms),
//   [PPO_WR_REST] everything else (c W1, c b1, ... < 65 504) -- as float bit patterns, recomputed from the parameters ONCE PER UPDATE by weight_range_kernel on a
//   stream of its own (api.hip: sweep_weight_range; no dependency on the update's stream, so it costs that stream nothing -- inside the AdamW kernel or the update
//   kernel's prologue the same few instructions cost 0.4 - 0.7 us per optimizer step, 1.5 % of the iteration, and as an extra workgroup of pack_records_kernel
//   7 us per update: the sweep's memory round trips and the write to the host were that kernel's tail) and, synchronously, after every host write of the parameters
//   (refresh_weight_range); wr_dev[4..6] mirrors what the pinned host words hold.  The HOST reads the mirror without synchronising, an update or two late --
//   AdamW moves a weight by about lr per step, hence thresholds at half the kernels' limits -- and takes the vector kernels for a launch whose weights do
//   not fit fp16 instead of failing.
// The optimizer kernels read the context's error word (OptGuard: one load that is in flight with the step's other loads) and do NOT apply a step the
// device already knows is garbage -- a bounded wait of the update kernel or of the scan ran out, or an observation left the fp16 range of the matrix-core
// update kernel (PPO_ERRFLAG_SKIP_STEP): parameters and moments stay as they were, the statistics are still written, the error stays sticky and
// ppo_read_stats reports it.  A gradient that is merely not finite is applied as the reference applies it (a 1-row minibatch gives NaN parameters there
// too, tests/test_gpu_parity.py): that is parity, not an error.
struct OptGuard {
    int32_t* error_flag = nullptr;
};
#ifndef PPO_OPT_GUARD
#define PPO_OPT_GUARD 1   /* -DPPO_OPT_GUARD=0: the optimizer kernels without the error-word load (A/B of its cost) */
#endif
#ifdef __HIPCC__
// The error word as a VECTOR load: the address passes through a vector register the compiler cannot see through, so the load is issued in the same batch as
// the step's other vector loads and waited for with them.  Left to itself hipcc made it two DEPENDENT scalar loads (the pointer out of the kernel arguments,
// then the word) with a wait behind each at the very top of the kernel, in front of everything else: + 0.15 us on a 5 us launch (trace A/B, NOTES).
// error_flag is never null here: the launchers refuse a null word (no branch in the kernel: a branch made the compiler wait for the word on the spot).
__device__ __forceinline__ int32_t opt_guard_word(const int32_t* error_flag) {
    if (!PPO_OPT_GUARD) return 0;
    uintptr_t p = reinterpret_cast<uintptr_t>(error_flag);
    asm volatile("" : "+v"(p));
    return *reinterpret_cast<const __attribute__((address_space(1))) int32_t*>(p);
}
#endif
__device__ __forceinline__ int wr_class(const NetLayout& L, int p) {
    const bool w3 = (p >= L.w3[0] && p < L.b3[0]) || (p >= L.w3[1] && p < L.b3[1]);
    const bool w2 = (p >= L.w2[0] && p < L.b2[0]) || (p >= L.w2[1] && p < L.b2[1]);
    return w3 ? 0 : (w2 ? 1 : 2);
}
constexpr int PPO_WR_W3 = 0, PPO_WR_W2 = 1, PPO_WR_REST = 2;
// grads[p] = sum over blocks (fixed order); loss sums -> sums_out[8]
hipError_t launch_reduce_grads(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                               double* sums_out, hipStream_t s);
hipError_t launch_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const NetLayout& L, float max_grad_norm,
                             const AdamCoef* coef, const double* loss_sums, double global_M, LossParams hp, int world, bool do_step,
                             StepStats* stats_out, double* clipfrac_accum, double* norm2_scratch, hipStream_t s, OptGuard guard);
// recomputes the three maxima of OptGuard from the parameters (after the host wrote them): wr_dev and wr_host both
hipError_t launch_weight_range(const float* params, const NetLayout& L, uint32_t* wr_dev, uint32_t* wr_host, hipStream_t s);
// batched critic on the matrix cores (obs in {2, 4}); same contract as launch_values
hipError_t launch_values_mfma(const float* params, const NetLayout& L, const float* obs0, int64_t n0, float* out0, const float* obs1, int64_t n1,
                              float* out1, hipStream_t s);
// single-rank optimizer step: (slab reduction + per-workgroup sums of squares) then (norm from the partials + clip + AdamW)
int fused_opt_blocks(const NetLayout& L);
hipError_t launch_reduce_clip_adamw(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                                    double* sums_out, float* params, float* exp_avg, float* exp_avg_sq, float max_grad_norm,
                                    const AdamCoef* coef, double global_M, LossParams hp, StepStats* stats_out, double* clipfrac_accum,
                                    double* partial, hipStream_t s, OptGuard guard);
hipError_t launch_reduce_exchange_clip_adamw(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                                             double* sums_out, float* params, float* exp_avg, float* exp_avg_sq, float max_grad_norm, const AdamCoef* coef,
                                             double global_M, const LossParams& hp, StepStats* stats_out, double* clipfrac_accum, double* partial,
                                             void* const* peers, int rank, int n_ranks, size_t slot_bytes, uint64_t seq, int32_t* timeout_flag,
                                             hipStream_t s, OptGuard guard);
// Device-resident CircularBuffer(100) of finished episodes (reference Utils/Utils.h:30-79).
struct EpisodeRing {
    float rew[100];
    int32_t len[100];
    int32_t size, head;
    int64_t total;
    int64_t key[100];   // (absolute rollout step) * global_num_envs + global env index: the episode's position in the reference's push order
};
hipError_t launch_episode_ring_update(const int32_t* fin_len, const float* fin_rew, int T, int N, int32_t* row_counts, uint64_t* group_bits,
                                      EpisodeRing* ring, int64_t step_base, int64_t global_num_envs, int64_t env_offset, hipStream_t s);
// this rank's contribution to the job-global statistics block (PPO_GSTAT_*): zeroes the block, then writes the head and the rank's own slot
hipError_t launch_gstats_pack(const double* ev_sums, const EpisodeRing* ring, int rank, double* gstats, hipStream_t s);
hipError_t launch_adv_stats(const float* advantages, const int32_t* perm, int64_t B, int64_t MB, int n_mb_total, AdvStat* out,
                            hipStream_t s);
hipError_t launch_permutations(int32_t* perm, int64_t B, int E, int64_t seed, int64_t update_index, int64_t rank_salt, hipStream_t s);
// the two above in one pass (ppo_update with norm_adv): perm[e][j] and the advantage sums of every minibatch of the update
// { mean, 1 / (std + 1e-8), std, 0 } of the advantages of minibatch slots [0, n) from their PPO_ADV_PARTS partial sums (PPO_Discrete.cpp:591-594); slot k is
// minibatch k % per_epoch of an epoch (rows min(MB, B - start) x world), or explicit_M x world rows when explicit_M > 0
// zero2 != nullptr: the launch also clears those two doubles (the update's clip-fraction accumulator)
hipError_t launch_adv_norm(const AdvStat* stats, int n, int per_epoch, int64_t B, int64_t MB, int64_t explicit_M, int world, float4* out, hipStream_t s, double* zero2 = nullptr);
hipError_t launch_permutations_adv_stats(const float* advantages, int32_t* perm, int64_t B, int E, int64_t MB, int64_t seed, int64_t update_index,
                                         int64_t rank_salt, AdvStat* out, hipStream_t s);
hipError_t launch_explained_variance(const float* returns, const float* values, int64_t B, double* sums4, hipStream_t s);
hipError_t launch_orthogonal_init(float* params, const NetLayout& L, int64_t seed, hipStream_t s);
