/* oracle/ppo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See ppo_oracle.h for the contract.
 * Every function cites the reference lines (relative to /root/reference) whose arithmetic it restates.
 * Build: gcc -O2 -std=c11 -ffp-contract=off (oracle/Makefile) -- no FMA contraction, as the reference's own
 * x86-64 build has none.
 */
#define _GNU_SOURCE
#include "ppo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================================
 * libm: glibc 2.35 sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c;
 * third-party, not under /root/reference -- CartPole.cpp:59-60 and MountainCar.cpp:34 call std::sin/std::cos
 * on float, i.e. these).  Evaluated in binary64; on x86-64 with FMA the ifunc picks the -mfma build, where
 * the compiler contracts a*b+c, which is what the explicit fma() calls below restate.  Pinned exhaustively
 * against the host libm (all floats with |x| in [2^-14, 4]) when written; tests/test_oracle_vs_golden.py re-checks a dense sample.
 * ==================================================================================================== */
static const double SC_C0 = 0x1p0, SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                    SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
static const double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7, SC_S3 = -0x1.994eb3774cf24p-13;
static const double SC_HPI_INV = 0x1.45F306DC9C883p+23, SC_HPI = 0x1.921FB54442D18p0;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

/* sinf_poly: n even -> sine polynomial, n odd -> cosine polynomial; neg selects table[1] (cos coefficients negated). */
static float sc_poly(double x, double x2, int n, int neg) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = fma(x2, SC_S3, SC_S2);
        double x7 = x3 * x2;
        double s = fma(x3, SC_S1, x);
        return (float)fma(x7, s1, s);
    } else {
        double sg = neg ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = fma(x2, sg * SC_C4, sg * SC_C3);
        double c1 = fma(x2, sg * SC_C1, sg * SC_C0);
        double x6 = x4 * x2;
        double c = fma(x4, sg * SC_C2, c1);
        return (float)fma(x6, c2, c);
    }
}

static double sc_reduce_fast(double x, int* np) {
    double r = x * SC_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, SC_HPI, x); /* x - n * hpi, contracted */
}

static const double SC_SIGN[4] = { 1.0, -1.0, -1.0, 1.0 };

float orc_sinf(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { /* |y| < pi/4 */
        double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sc_poly(x, s, 0, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = sc_reduce_fast(x, &n);
        double s = SC_SIGN[n & 3];
        return sc_poly(x * s, x * x, n, (n & 2) != 0);
    }
    return sinf(y); /* outside every state the environments can reach */
}

float orc_cosf(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sc_poly(x, x2, 1, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = sc_reduce_fast(x, &n);
        double s = SC_SIGN[(n + 1) & 3];
        return sc_poly(x * s, x * x, n ^ 1, ((n + 1) & 2) != 0);
    }
    return cosf(y);
}

/* ======================================================================================================
 * RNG
 * ==================================================================================================== */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
/* libstdc++ uniform_real_distribution<float>(a,b): generate_canonical<float,24> takes ONE 32-bit draw:
 * float(u)/2^32 (clamped below 1), then *(b-a)+a.  CartPole.cpp:96-100. */
static inline float canon_to_uniform(uint32_t u, float a, float b) {
    float r = (float)u / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r * (b - a) + a;
}

void orc_cartpole_reset_stream(int64_t seed, int64_t n_resets, float* out) {
    mt19937_t g;
    mt_seed(&g, (uint32_t)seed); /* std::mt19937 gen(seed), CartPole.cpp:3-4 */
    for (int64_t i = 0; i < n_resets * 4; i++) out[i] = canon_to_uniform(mt_next(&g), -0.05f, 0.05f);
}

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
    uint64_t p = (uint64_t)a * b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ======================================================================================================
 * Environments
 * ==================================================================================================== */
/* CartPole::step, CartPole.cpp:47-94; constants CartPole.cpp:6-17. */
float orc_cartpole_step(float* st, int64_t action, int32_t* terminated) {
    const float gravity = 9.8f, mass_pole = 0.1f, total_mass = 0.1f + 1.0f, length = 0.5f;
    const float polemass_length = 0.1f * 0.5f, force_mag = 10.0f, tau = 0.02f;
    const float theta_thr = (float)(12 * 2 * M_PI / 360), x_thr = 2.4f;
    float x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
    float force = force_mag;
    if (action == 0) force = -force;
    float cos_theta = orc_cosf(theta), sin_theta = orc_sinf(theta);
    float temp = (force + polemass_length * theta_dot * theta_dot * sin_theta) / total_mass;
    float theta_acc = (gravity * sin_theta - cos_theta * temp) /
                      (length * (4.0f / 3.0f - mass_pole * cos_theta * cos_theta / total_mass));
    float x_acc = temp - polemass_length * theta_acc * cos_theta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * x_acc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * theta_acc;
    st[0] = x; st[1] = x_dot; st[2] = theta; st[3] = theta_dot;
    int term = (x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr);
    *terminated = term;
    return term ? -1.0f : 1.0f;
}

/* MountainCar::step, MountainCar.cpp:29-57; constants :5-13. */
float orc_mountaincar_step(float* st, int64_t action, int32_t* terminated) {
    const float min_position = -1.2f, max_position = 0.6f, max_speed = 0.07f, goal_position = 0.5f, goal_velocity = 0.0f;
    const float force = 0.001f, gravity = 0.0025f;
    float position = st[0], velocity = st[1];
    velocity += ((float)action - 1.0f) * force + orc_cosf(3.0f * position) * (-gravity);
    velocity = velocity < -max_speed ? -max_speed : (velocity > max_speed ? max_speed : velocity);
    position += velocity;
    position = position < min_position ? min_position : (position > max_position ? max_position : position);
    if (position == min_position && velocity < 0.0f) velocity = 0.0f;
    *terminated = (position >= goal_position && velocity >= goal_velocity);
    st[0] = position; st[1] = velocity;
    return -1.0f;
}

void orc_synthetic_obs(int64_t seed, int64_t env, int64_t step, int32_t obs_size, float* out) {
    for (int j = 0; j < obs_size; j++) {
        uint32_t w[4];
        orc_philox4x32((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env, (uint32_t)step, (uint32_t)(j >> 2), 0x10u, w);
        const uint32_t x = w[j & 3];
        const int sum = (int)(x & 255u) + (int)((x >> 8) & 255u) + (int)((x >> 16) & 255u) + (int)(x >> 24);
        out[j] = (float)(sum - 510) * 0.0067658990621566772f;   /* four uniform bytes: variance 4 (256^2 - 1) / 12 */
    }
}
void orc_synthetic_mask(int64_t seed, int64_t env, int64_t step, int32_t n_heads, const int32_t* head_dims, uint8_t* out) {
    uint32_t w[4];
    orc_philox4x32((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env, (uint32_t)step, 0u, 0x12u, w);
    int off = 0;
    for (int h = 0; h < n_heads; h++) {
        int any = 0;
        for (int k = 0; k < head_dims[h]; k++) { const int v = (w[0] >> (off + k)) & 1u; any |= v; out[off + k] = (uint8_t)v; }
        if (!any) out[off] = 1;
        off += head_dims[h];
    }
}
void orc_synthetic_transition(int64_t seed, int64_t env, int64_t step, float* reward, int32_t* done) {
    uint32_t w[4];
    orc_philox4x32((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env, (uint32_t)step, 0u, 0x11u, w);
    *reward = (float)(w[0] >> 8) * 0x1p-23f - 1.0f;
    *done = (w[1] >> 8) < 167772u ? 1 : 0;
}

void orc_step_many(int32_t kind, float* state, const int64_t* action, int64_t n, float* reward, int32_t* terminated) {
    const int obs = kind == 0 ? 4 : 2;
    for (int64_t i = 0; i < n; i++)
        reward[i] = kind == 0 ? orc_cartpole_step(state + i * obs, action[i], terminated + i)
                              : orc_mountaincar_step(state + i * obs, action[i], terminated + i);
}

struct orc_vecenv {
    int32_t kind;
    int64_t n, seed, max_steps, env_offset;
    int obs;
    float* state;       /* [n, obs] */
    int64_t* ep_len;    /* episode_length, CartPole.h:44 */
    float* ep_rew;      /* episode_reward, CartPole.h:45 */
    int64_t* reset_cnt; /* resets consumed from the env's private (identically seeded) stream */
    float* stream;      /* shared CartPole reset stream [cap,4] */
    int64_t stream_cap;
    /* CircularBuffer(100), Utils.h:30-79 */
    float ring_rew[100];
    int64_t ring_len[100];
    size_t ring_size, ring_head;
    double ring_rew_sum, ring_len_sum;
};

static void ring_add(orc_vecenv* e, float reward, int64_t length) {
    if (e->ring_size == 100) {
        e->ring_rew_sum -= e->ring_rew[e->ring_head];
        e->ring_len_sum -= (double)e->ring_len[e->ring_head];
    } else {
        e->ring_size++;
    }
    e->ring_rew[e->ring_head] = reward;
    e->ring_len[e->ring_head] = length;
    e->ring_rew_sum += reward;
    e->ring_len_sum += (double)length;
    e->ring_head = (e->ring_head + 1) % 100;
}

static void stream_ensure(orc_vecenv* e, int64_t k) {
    if (k < e->stream_cap) return;
    int64_t cap = e->stream_cap ? e->stream_cap : 1024;
    while (cap <= k) cap *= 2;
    e->stream = (float*)realloc(e->stream, (size_t)cap * 4 * sizeof(float));
    orc_cartpole_reset_stream(e->seed, cap, e->stream);
    e->stream_cap = cap;
}

/* CartPole::reset (CartPole.cpp:34-45) / MountainCar::reset (MountainCar.cpp:59-66). */
static void env_reset(orc_vecenv* e, int64_t i) {
    int64_t k = e->reset_cnt[i]++;
    if (e->kind == ORC_ENV_CARTPOLE) {
        stream_ensure(e, k);
        memcpy(&e->state[i * 4], &e->stream[k * 4], 4 * sizeof(float));
    } else {
        /* Reference draws from std::random_device (MountainCar.cpp:79-88): unseedable.  The build's documented
         * deviation: U(-0.6,-0.4) through the same libstdc++ mapping, word = philox(seed; env, reset#, 0, 1).x */
        uint32_t w[4];
        orc_philox4x32((uint32_t)e->seed, (uint32_t)((uint64_t)e->seed >> 32), (uint32_t)(e->env_offset + i), (uint32_t)k, 0u, 1u, w);
        e->state[i * 2] = canon_to_uniform(w[0], -0.6f, -0.4f);
        e->state[i * 2 + 1] = 0.0f;
    }
    e->ep_len[i] = 0;
    e->ep_rew[i] = 0.0f;
}

orc_vecenv* orc_vecenv_create(int32_t kind, int64_t n, int64_t seed, int64_t max_steps, int64_t env_offset) {
    orc_vecenv* e = (orc_vecenv*)calloc(1, sizeof *e);
    e->kind = kind; e->n = n; e->seed = seed; e->max_steps = max_steps; e->env_offset = env_offset;
    e->obs = kind == ORC_ENV_CARTPOLE ? 4 : 2;
    e->state = (float*)calloc((size_t)n * e->obs, sizeof(float));
    e->ep_len = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    e->ep_rew = (float*)calloc((size_t)n, sizeof(float));
    e->reset_cnt = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    return e;
}
void orc_vecenv_destroy(orc_vecenv* e) {
    if (!e) return;
    free(e->state); free(e->ep_len); free(e->ep_rew); free(e->reset_cnt); free(e->stream); free(e);
}

/* PPO_Discrete::initEnvs, PPO_Discrete.cpp:365-402: env 0 is reset once for the obs-size probe (:368) and
 * again with everyone else (:389). */
void orc_vecenv_init(orc_vecenv* e, float* obs_out) {
    if (e->env_offset == 0 && e->n > 0) env_reset(e, 0);
    for (int64_t i = 0; i < e->n; i++) env_reset(e, i);
    memcpy(obs_out, e->state, (size_t)e->n * e->obs * sizeof(float));
}

/* PPO_Discrete::stepEnvs, PPO_Discrete.cpp:413-483. */
void orc_vecenv_step(orc_vecenv* e, const int64_t* action, float* obs_out, float* reward_out, int32_t* done_out) {
    for (int64_t i = 0; i < e->n; i++) {
        int32_t term;
        float* st = &e->state[i * e->obs];
        float r = e->kind == ORC_ENV_CARTPOLE ? orc_cartpole_step(st, action[i], &term) : orc_mountaincar_step(st, action[i], &term);
        e->ep_len[i] += 1;   /* CartPole.cpp:90-91 */
        e->ep_rew[i] += r;
        if (e->ep_len[i] == e->max_steps) term = 1; /* :443-445 */
        if (term) {                                   /* :453-458; stats added in env order (:474-480) */
            ring_add(e, e->ep_rew[i], e->ep_len[i]);
            env_reset(e, i);
        }
        memcpy(&obs_out[i * e->obs], st, (size_t)e->obs * sizeof(float));
        reward_out[i] = r;
        done_out[i] = term;
    }
}

void orc_vecenv_set_state(orc_vecenv* e, const float* state, const int64_t* ep_len, const float* ep_rew, const int64_t* reset_count) {
    if (state) memcpy(e->state, state, (size_t)e->n * e->obs * sizeof(float));
    if (ep_len) memcpy(e->ep_len, ep_len, (size_t)e->n * sizeof(int64_t));
    if (ep_rew) memcpy(e->ep_rew, ep_rew, (size_t)e->n * sizeof(float));
    if (reset_count) memcpy(e->reset_cnt, reset_count, (size_t)e->n * sizeof(int64_t));
}
void orc_vecenv_get_state(const orc_vecenv* e, float* state, int64_t* ep_len, float* ep_rew, int64_t* reset_count) {
    if (state) memcpy(state, e->state, (size_t)e->n * e->obs * sizeof(float));
    if (ep_len) memcpy(ep_len, e->ep_len, (size_t)e->n * sizeof(int64_t));
    if (ep_rew) memcpy(ep_rew, e->ep_rew, (size_t)e->n * sizeof(float));
    if (reset_count) memcpy(reset_count, e->reset_cnt, (size_t)e->n * sizeof(int64_t));
}
void orc_vecenv_episode_stats(const orc_vecenv* e, double out[3]) {
    out[0] = e->ring_size ? e->ring_len_sum / (double)e->ring_size : 0.0;               /* avgLength, Utils.h:76-78 */
    out[1] = e->ring_size ? (double)(float)(e->ring_rew_sum / (double)e->ring_size) : 0.0; /* avgReward returns float, :72-74 */
    out[2] = (double)e->ring_size;
}

/* ======================================================================================================
 * Network (Agent.cpp:19-72): critic then actor, each n_hidden x (Linear, Tanh) + Linear head.
 * ==================================================================================================== */
static int act_total(const orc_net* c) {
    int a = 0;
    for (int h = 0; h < c->n_heads; h++) a += c->head_dims[h];
    return a;
}
static int layer_in(const orc_net* c, int l) { return l == 0 ? c->obs_size : c->hidden; }
static int layer_out(const orc_net* c, int net, int l) { return l == c->n_hidden ? (net == 0 ? 1 : act_total(c)) : c->hidden; }
static int64_t net_size(const orc_net* c, int net) {
    int64_t s = 0;
    for (int l = 0; l <= c->n_hidden; l++) s += (int64_t)layer_out(c, net, l) * layer_in(c, l) + layer_out(c, net, l);
    return s;
}
int64_t orc_param_count(const orc_net* c) { return net_size(c, 0) + net_size(c, 1); }
void orc_param_shapes(const orc_net* c, int64_t* shapes) {
    int k = 0;
    for (int net = 0; net < 2; net++)
        for (int l = 0; l <= c->n_hidden; l++) {
            shapes[k++] = layer_out(c, net, l); shapes[k++] = layer_in(c, l); /* weight [out,in] */
            shapes[k++] = layer_out(c, net, l); shapes[k++] = 1;              /* bias [out] */
        }
}

/* Forward of one net for one sample.  acts (optional) receives the post-tanh activations of every hidden
 * layer, [n_hidden][hidden].  out[layer_out(last)]. */
/* round to nearest even to bf16, returned as float (NaN stays NaN) */
static inline float bf16r(float x) {
    uint32_t u = f2u(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    float r; memcpy(&r, &u, 4); return r;
}
static inline float opnd(const orc_net* c, float x) { return c->dtype == ORC_DTYPE_BF16 ? bf16r(x) : x; }

static void mlp_forward1(const orc_net* c, int net, const float* p, const float* x, float* acts, float* out) {
    float bufA[1024], bufB[1024], bufX[8192];
    const float* in = x;
    if (c->dtype == ORC_DTYPE_BF16) { for (int k = 0; k < c->obs_size; k++) bufX[k] = bf16r(x[k]); in = bufX; }
    float* cur = bufA;
    for (int l = 0; l <= c->n_hidden; l++) {
        int ni = layer_in(c, l), no = layer_out(c, net, l);
        const float* W = p;
        const float* b = p + (size_t)no * ni;
        float* dst = (l == c->n_hidden) ? out : cur;
        for (int j = 0; j < no; j++) {
            float acc = b[j];
            for (int k = 0; k < ni; k++) acc += in[k] * opnd(c, W[(size_t)j * ni + k]);
            dst[j] = (l == c->n_hidden) ? acc : opnd(c, tanhf(acc));   /* a hidden activation is stored in the compute dtype */
        }
        if (l < c->n_hidden) {
            if (acts) memcpy(acts + (size_t)l * c->hidden, dst, (size_t)no * sizeof(float));
            in = dst;
            cur = (cur == bufA) ? bufB : bufA;
        }
        p += (size_t)no * ni + no;
    }
}

void orc_get_value(const orc_net* c, const float* params, const float* x, int64_t n, float* value) {
    for (int64_t i = 0; i < n; i++) mlp_forward1(c, 0, params, x + i * c->obs_size, NULL, &value[i]);
}
void orc_actor_logits(const orc_net* c, const float* params, const float* x, int64_t n, float* logits) {
    int A = act_total(c);
    const float* pa = params + net_size(c, 0);
    for (int64_t i = 0; i < n; i++) mlp_forward1(c, 1, pa, x + i * c->obs_size, NULL, logits + i * A);
}

/* One head, one row.  Categorical.cpp:28-39 (m_logits = logits - logsumexp, m_probs = softmax), :92-101 (gather),
 * :112-119 (entropy with clamp(min = FLT_MIN): every log-prob <= 0 becomes +FLT_MIN);
 * CategoricalMasked.cpp:31-46 (where(mask, logits, -1e8f)), :127-144 (entropy = -sum where(mask, p*logp, 0)). */
static void categorical_row(int32_t kind, const float* logits, const uint8_t* mask, int A, float* m_logits, float* m_probs,
                            float* entropy) {
    float z[64];
    float mx = -INFINITY;
    for (int a = 0; a < A; a++) {
        z[a] = (kind == ORC_DIST_MASKED && mask && !mask[a]) ? -1e8f : logits[a];
        if (z[a] > mx) mx = z[a];
    }
    float se = 0.0f;
    float ex[64];
    for (int a = 0; a < A; a++) { ex[a] = expf(z[a] - mx); se += ex[a]; }
    float lse = logf(se) + mx;
    float ent = 0.0f;
    for (int a = 0; a < A; a++) {
        m_logits[a] = z[a] - lse;
        m_probs[a] = ex[a] / se;
        if (kind == ORC_DIST_CATEGORICAL) {
            float l = m_logits[a] > FLT_MIN ? m_logits[a] : FLT_MIN; /* torch::clamp(m_logits, min_real) */
            ent += l * m_probs[a];
        } else {
            float plp = m_logits[a] * m_probs[a];
            ent += (mask == NULL || mask[a]) ? plp : 0.0f;
        }
    }
    *entropy = -ent;
}

void orc_categorical(int32_t kind, const float* logits, const uint8_t* mask, const int64_t* value, int64_t n, int32_t A,
                     float* m_logits, float* m_probs, float* log_prob, float* entropy) {
    for (int64_t i = 0; i < n; i++) {
        float ml[64], mp[64], ent;
        categorical_row(kind, logits + i * A, mask ? mask + i * A : NULL, A, ml, mp, &ent);
        if (m_logits) memcpy(m_logits + i * A, ml, (size_t)A * sizeof(float));
        if (m_probs) memcpy(m_probs + i * A, mp, (size_t)A * sizeof(float));
        if (log_prob && value) log_prob[i] = ml[value[i]];
        if (entropy) entropy[i] = ent;
    }
}

/* Agent::getActionAndValueDiscrete (Agent.cpp:117-128) / getActionAndValueMasked (:137-170), teacher-forced. */
void orc_evaluate(const orc_net* c, const float* params, const float* x, const uint8_t* mask, const int64_t* action, int64_t n,
                  float* logprob, float* entropy, float* value) {
    int A = act_total(c);
    const float* pa = params + net_size(c, 0);
    for (int64_t i = 0; i < n; i++) {
        float logits[64];
        mlp_forward1(c, 1, pa, x + i * c->obs_size, NULL, logits);
        float lp = 0.0f, en = 0.0f;
        int off = 0;
        for (int h = 0; h < c->n_heads; h++) {
            float ml[64], mp[64], e1;
            int Ah = c->head_dims[h];
            categorical_row(c->dist_kind, logits + off, mask ? mask + i * A + off : NULL, Ah, ml, mp, &e1);
            float l1 = ml[action[i * c->n_heads + h]];
            if (h == 0) { lp = l1; en = e1; } else { lp += l1; en += e1; } /* stack(...).sum(0), Agent.cpp:165-168 */
            off += Ah;
        }
        if (logprob) logprob[i] = lp;
        if (entropy) entropy[i] = en;
        if (value) mlp_forward1(c, 0, params, x + i * c->obs_size, NULL, &value[i]);
    }
}

void orc_act(const orc_net* c, const float* params, const float* x, const uint8_t* mask, int64_t n, int64_t seed,
             int64_t env_offset, int64_t step, int64_t* action, float* logprob, float* entropy, float* value) {
    int A = act_total(c);
    const float* pa = params + net_size(c, 0);
    for (int64_t i = 0; i < n; i++) {
        float logits[64];
        mlp_forward1(c, 1, pa, x + i * c->obs_size, NULL, logits);
        float lp = 0.0f, en = 0.0f;
        int off = 0;
        for (int h = 0; h < c->n_heads; h++) {
            float ml[64], mp[64], e1;
            int Ah = c->head_dims[h];
            categorical_row(c->dist_kind, logits + off, mask ? mask + i * A + off : NULL, Ah, ml, mp, &e1);
            uint32_t w[4];
            /* one Philox call feeds four consecutive steps: counter (row, step / 4, head, 0), word step % 4 */
            orc_philox4x32((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)(env_offset + i), (uint32_t)((uint64_t)step >> 2), (uint32_t)h, 0u, w);
            float u = (float)(w[step & 3] >> 8) * 0x1p-24f;
            int a = 0, last = 0;
            float acc = 0.0f;
            int hit = 0;
            for (int k = 0; k < Ah; k++) {
                if (mp[k] > 0.0f) last = k;
                acc += mp[k];
                if (!hit && u < acc) { a = k; hit = 1; }
            }
            if (!hit) a = last;
            action[i * c->n_heads + h] = a;
            if (h == 0) { lp = ml[a]; en = e1; } else { lp += ml[a]; en += e1; }
            off += Ah;
        }
        if (logprob) logprob[i] = lp;
        if (entropy) entropy[i] = en;
        if (value) mlp_forward1(c, 0, params, x + i * c->obs_size, NULL, &value[i]);
    }
}

/* ======================================================================================================
 * Advantages
 * ==================================================================================================== */
/* PPO_Discrete::calcAdvantage, GAE branch (PPO_Discrete.cpp:283-306):
 *   nextnonterminal = 1.0 - dones[t+1]             (t == T-1: 1.0 - next_done, an int32 tensor, :292)
 *   delta       = rewards[t] + m_gamma * nextvalues * nextnonterminal - values[t]        (:300)
 *               = ((r + ((gamma * nv) * nnt)) - v)    left-to-right tensor expression
 *   lastgaelam  = delta + m_gamma * m_gae_lambda * nextnonterminal * lastgaelam          (:301)
 *               = delta + (((gamma * lambda) * nnt) * last)   with gamma*lambda a C++ float product
 *   returns     = advantages + values                                                     (:305)        */
void orc_gae(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
             int64_t T, int64_t N, float gamma, float gae_lambda, float* advantages, float* returns) {
    const float gl = gamma * gae_lambda;
    for (int64_t n = 0; n < N; n++) {
        float last = 0.0f;
        for (int64_t t = T - 1; t >= 0; t--) {
            float nnt, nv;
            if (t == T - 1) { nnt = (float)(1 - next_done[n]); nv = next_value[n]; }
            else { nnt = 1.0f - dones[(t + 1) * N + n]; nv = values[(t + 1) * N + n]; }
            float delta = (rewards[t * N + n] + (gamma * nv) * nnt) - values[t * N + n];
            last = delta + (gl * nnt) * last;
            advantages[t * N + n] = last;
            returns[t * N + n] = last + values[t * N + n];
        }
    }
}

/* n-step branch (PPO_Discrete.cpp:309-329): returns[t] = rewards[t] + m_gamma * nextnonterminal * next_return;
 * advantages = returns - values. */
void orc_nstep(const float* rewards, const float* values, const float* dones, const float* next_value, const int32_t* next_done,
               int64_t T, int64_t N, float gamma, float* advantages, float* returns) {
    for (int64_t n = 0; n < N; n++) {
        float nr = 0.0f;
        for (int64_t t = T - 1; t >= 0; t--) {
            float nnt;
            if (t == T - 1) { nnt = (float)(1 - next_done[n]); nr = next_value[n]; }
            else { nnt = 1.0f - dones[(t + 1) * N + n]; }
            nr = rewards[t * N + n] + (gamma * nnt) * nr;
            returns[t * N + n] = nr;
            advantages[t * N + n] = nr - values[t * N + n];
        }
    }
}

/* ======================================================================================================
 * Update
 * ==================================================================================================== */
/* Backward of one net for one sample: dout[layer_out(last)] -> accumulates into g (double, same layout as p). */
static void mlp_backward1(const orc_net* c, int net, const float* p, const float* x, const float* acts, const float* dout,
                          double* g) {
    /* layer offsets */
    size_t off[16];
    size_t o = 0;
    for (int l = 0; l <= c->n_hidden; l++) { off[l] = o; o += (size_t)layer_out(c, net, l) * layer_in(c, l) + layer_out(c, net, l); }
    float dz[1024], dh[1024], xq[8192];
    int no = layer_out(c, net, c->n_hidden);
    for (int j = 0; j < no; j++) dz[j] = dout[j];
    if (c->dtype == ORC_DTYPE_BF16) { for (int k = 0; k < c->obs_size; k++) xq[k] = bf16r(x[k]); x = xq; }
    for (int l = c->n_hidden; l >= 0; l--) {
        int ni = layer_in(c, l);
        no = layer_out(c, net, l);
        const float* W = p + off[l];
        double* gW = g + off[l];
        double* gb = gW + (size_t)no * ni;
        const float* in = (l == 0) ? x : acts + (size_t)(l - 1) * c->hidden;   /* stored (rounded) activations */
        for (int k = 0; k < ni; k++) dh[k] = 0.0f;
        for (int j = 0; j < no; j++) {
            gb[j] += dz[j];                        /* bias gradient: the f32 d(pre-activation) */
            float d = opnd(c, dz[j]);              /* as an operand of the two products: the compute dtype */
            for (int k = 0; k < ni; k++) {
                gW[(size_t)j * ni + k] += (double)(d * in[k]);
                dh[k] += d * opnd(c, W[(size_t)j * ni + k]);
            }
        }
        if (l > 0)
            for (int k = 0; k < ni; k++) dz[k] = dh[k] * (1.0f - in[k] * in[k]); /* tanh' */
    }
}

/* Data-parallel building block (no reference counterpart): the same minibatch step on ONE shard of a global
 * minibatch.  adv_sums = {sum, sum of squares} of the advantages over the GLOBAL minibatch (NULL: use this shard's own),
 * global_M = rows of the global minibatch.  Gradients and loss sums come out scaled by 1/global_M, so that a plain sum over
 * shards equals the single-process result on the concatenated minibatch.  stats[] then holds this shard's share of each mean. */
void orc_minibatch_grads_shard(const orc_net* c, const orc_hparams* hp, const float* params, const float* b_obs, const float* b_actions,
                               int32_t act_cols, const uint8_t* b_mask, const float* b_logprobs, const float* b_advantages,
                               const float* b_returns, const float* b_values, const int64_t* idx, int64_t M, const double* adv_sums,
                               int64_t global_M, float* grads, double stats[6], double local_adv_sums[2]);

void orc_minibatch_grads(const orc_net* c, const orc_hparams* hp, const float* params, const float* b_obs, const float* b_actions,
                         int32_t act_cols, const uint8_t* b_mask, const float* b_logprobs, const float* b_advantages,
                         const float* b_returns, const float* b_values, const int64_t* idx, int64_t M, float* grads,
                         double stats[6]) {
    orc_minibatch_grads_shard(c, hp, params, b_obs, b_actions, act_cols, b_mask, b_logprobs, b_advantages, b_returns, b_values, idx, M,
                              NULL, M, grads, stats, NULL);
}

void orc_minibatch_grads_shard(const orc_net* c, const orc_hparams* hp, const float* params, const float* b_obs, const float* b_actions,
                               int32_t act_cols, const uint8_t* b_mask, const float* b_logprobs, const float* b_advantages,
                               const float* b_returns, const float* b_values, const int64_t* idx, int64_t M, const double* adv_sums,
                               int64_t global_M, float* grads, double stats[6], double local_adv_sums[2]) {
    const int A = act_total(c);
    const int64_t P = orc_param_count(c);
    const int64_t Pc = net_size(c, 0);
    const float* pc = params;
    const float* pa = params + Pc;
    double* g = (double*)calloc((size_t)P, sizeof(double));
    const float clip = hp->clip_coef;
    const float lo = 1 - clip, hi = 1 + clip; /* int - float -> float, PPO_Discrete.cpp:598 */

    /* advantage normalisation statistics over the minibatch, PPO_Discrete.cpp:591-594 (std is Bessel-corrected) */
    float mean_f = 0.0f, std_f = 0.0f;
    {
        double s1 = 0.0, s2 = 0.0;
        for (int64_t j = 0; j < M; j++) { double a = b_advantages[idx[j]]; s1 += a; s2 += a * a; }
        if (local_adv_sums) { local_adv_sums[0] = s1; local_adv_sums[1] = s2; }
    }
    if (hp->norm_adv) {
        if (adv_sums) {  /* global statistics supplied by the caller (summed over shards) */
            double mean = adv_sums[0] / (double)global_M;
            double var = (adv_sums[1] - adv_sums[0] * mean) / (double)(global_M - 1);
            mean_f = (float)mean;
            std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
        } else {
            double s = 0.0;
            for (int64_t j = 0; j < M; j++) s += b_advantages[idx[j]];
            double mean = s / (double)M;
            double ss = 0.0;
            for (int64_t j = 0; j < M; j++) { double d = b_advantages[idx[j]] - mean; ss += d * d; }
            mean_f = (float)mean;
            std_f = (float)sqrt(ss / (double)(M - 1));
        }
    }

    double s_pg = 0, s_v = 0, s_ent = 0, s_kl = 0;
    int64_t n_clip = 0;
    const float invM = 1.0f / (float)global_M;
    float* actsA = (float*)malloc((size_t)c->n_hidden * c->hidden * sizeof(float));
    float* actsC = (float*)malloc((size_t)c->n_hidden * c->hidden * sizeof(float));
    for (int64_t j = 0; j < M; j++) {
        const int64_t i = idx[j];
        const float* x = b_obs + i * c->obs_size;
        const uint8_t* mask = b_mask ? b_mask + i * A : NULL;
        float logits[64], dlogits[64], nv;
        mlp_forward1(c, 1, pa, x, actsA, logits);
        mlp_forward1(c, 0, pc, x, actsC, &nv);

        /* distribution */
        float ml[64], mp[64], headH[ORC_MAX_HEADS];
        int acts_idx[ORC_MAX_HEADS];
        float nlp = 0.0f, ent = 0.0f;
        int off = 0;
        for (int h = 0; h < c->n_heads; h++) {
            int Ah = c->head_dims[h];
            categorical_row(c->dist_kind, logits + off, mask ? mask + off : NULL, Ah, ml + off, mp + off, &headH[h]);
            /* b_actions.to(kLong): PPO_Discrete.cpp:581; MultiDiscrete picks column h of the [B, action_size] buffer (:607-609) */
            int a = (int)(int64_t)b_actions[i * act_cols + h];
            acts_idx[h] = a;
            if (h == 0) { nlp = ml[off + a]; ent = headH[h]; } else { nlp += ml[off + a]; ent += headH[h]; }
            off += Ah;
        }

        /* losses, PPO_Discrete.cpp:585-631 */
        float logratio = nlp - b_logprobs[i];
        float ratio = expf(logratio);
        if (fabsf(ratio - 1.0f) > clip) n_clip++;           /* :348 */
        s_kl += (double)((ratio - 1.0f) - logratio);        /* :352 */
        float adv = b_advantages[i];
        if (hp->norm_adv) adv = (adv - mean_f) / (std_f + 1e-8f);
        float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
        float l1 = -adv * ratio, l2 = -adv * rc;
        s_pg += (double)(l1 > l2 ? l1 : l2);
        const int inside = (ratio >= lo && ratio <= hi);
        float d_ratio; /* d max(l1,l2) / d ratio; torch splits ties half/half */
        if (l1 > l2) d_ratio = -adv;
        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
        float g_nlp = invM * d_ratio * ratio;

        float R = b_returns[i], vold = b_values[i];
        float un = (nv - R) * (nv - R);
        float g_v;
        if (hp->clip_vloss) {
            float dv = nv - vold;
            float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
            float vc = vold + dvc;
            float cl = (vc - R) * (vc - R);
            s_v += (double)(un > cl ? un : cl);
            const int vin = (dv >= -clip && dv <= clip);
            float d_un = 2.0f * (nv - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
            float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
            g_v = hp->vf_coef * 0.5f * invM * d;
        } else {
            s_v += (double)un;
            g_v = hp->vf_coef * 0.5f * invM * 2.0f * (nv - R);
        }
        s_ent += (double)ent;
        float g_ent = -hp->ent_coef * invM; /* dL/dentropy_j */

        /* d logits */
        off = 0;
        for (int h = 0; h < c->n_heads; h++) {
            int Ah = c->head_dims[h];
            for (int a = 0; a < Ah; a++) {
                int valid = (c->dist_kind == ORC_DIST_CATEGORICAL) || mask == NULL || mask[off + a];
                float p = mp[off + a];
                float d = g_nlp * ((a == acts_idx[h] ? 1.0f : 0.0f) - p);
                if (c->dist_kind == ORC_DIST_MASKED) d += g_ent * (-p * (ml[off + a] + headH[h]));
                /* ORC_DIST_CATEGORICAL: the clamp makes entropy = -FLT_MIN * sum(p): its gradient is a denormal
                 * (~1e-45) times ent_coef/M, i.e. zero (Categorical.cpp:112-119). */
                dlogits[off + a] = valid ? d : 0.0f;
            }
            off += Ah;
        }
        mlp_backward1(c, 1, pa, x, actsA, dlogits, g + Pc);
        mlp_backward1(c, 0, pc, x, actsC, &g_v, g);
    }
    free(actsA); free(actsC);
    for (int64_t k = 0; k < P; k++) grads[k] = (float)g[k];
    free(g);
    float pg = (float)(s_pg / (double)global_M);
    float vl = 0.5f * (float)(s_v / (double)global_M);
    float el = (float)(s_ent / (double)global_M);
    stats[0] = pg; stats[1] = vl; stats[2] = el;
    stats[3] = (float)(s_kl / (double)global_M);
    stats[4] = (float)n_clip / (float)global_M; /* PPO_Discrete.cpp:349 */
    stats[5] = (pg - hp->ent_coef * el) + vl * hp->vf_coef; /* :631 */
}

/* LibTorch clip_grad_norm_ (clip_grad.h:22-85; call site PPO_Discrete.cpp:640): L2 norm of the per-tensor L2 norms,
 * coef = max_norm / (total + 1e-6) clamped to <= 1, every grad multiplied by it (also when it is 1). */
double orc_clip_grad_norm(const orc_net* c, float* grads, float max_norm) {
    double tot = 0.0;
    int64_t o = 0;
    for (int net = 0; net < 2; net++)
        for (int l = 0; l <= c->n_hidden; l++) {
            int64_t nw = (int64_t)layer_out(c, net, l) * layer_in(c, l), nb = layer_out(c, net, l);
            double sw = 0, sb = 0;
            for (int64_t k = 0; k < nw; k++) sw += (double)grads[o + k] * grads[o + k];
            for (int64_t k = 0; k < nb; k++) sb += (double)grads[o + nw + k] * grads[o + nw + k];
            float nrm_w = (float)sqrt(sw), nrm_b = (float)sqrt(sb);
            tot += (double)nrm_w * nrm_w + (double)nrm_b * nrm_b;
            o += nw + nb;
        }
    float total = (float)sqrt(tot);
    float coef = max_norm / (total + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
    for (int64_t k = 0; k < o; k++) grads[k] *= coef;
    return (double)total;
}

/* torch::optim::AdamW::step (LibTorch optim/adamw.cpp; constructed at PPO_Discrete.cpp:76-78 with eps = 1e-5f,
 * defaults betas (0.9, 0.999), weight_decay 1e-2, amsgrad false).  Double scalars are narrowed to float when
 * applied to float tensors; addcmul_/addcdiv_ evaluate (value * t1) * t2 and (value * t1) / t2. */
void orc_adamw_step(float* p, const float* grad, float* m, float* v, int64_t P, double lr, int64_t t) {
    const double beta1 = 0.9, beta2 = 0.999, wd = 1e-2;
    const float eps = 1e-5f;
    const float decay = (float)(1.0 - lr * wd);
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    const float b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const float sq_bc2 = (float)sqrt(bc2);
    const float neg_step = (float)(-(lr / bc1));
    for (int64_t i = 0; i < P; i++) {
        float pi = p[i] * decay;
        /* add_(grad, alpha) and addcmul_ run through Vectorized<float>::fmadd in ATen's AVX-512 kernels: one rounding
         * for alpha*grad + m*beta1 and for (value*grad)*grad + v*beta2 (verified bit-exact on the golden vectors). */
        float mi = fmaf(grad[i], omb1, m[i] * b1);
        float vi = fmaf(omb2 * grad[i], grad[i], v[i] * b2);
        float denom = sqrtf(vi) / sq_bc2 + eps;
        p[i] = pi + (neg_step * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}

/* PPO_Discrete.cpp:647-648: 1 - var(returns - values) / var(returns), both unbiased. */
double orc_explained_variance(const float* returns, const float* values, int64_t B) {
    double sy = 0, sd = 0;
    for (int64_t i = 0; i < B; i++) { sy += returns[i]; sd += (double)(returns[i] - values[i]); }
    double my = sy / (double)B, md = sd / (double)B, vy = 0, vd = 0;
    for (int64_t i = 0; i < B; i++) {
        double a = returns[i] - my, b = (double)(returns[i] - values[i]) - md;
        vy += a * a; vd += b * b;
    }
    vy /= (double)(B - 1); vd /= (double)(B - 1);
    return (double)(1.0f - (float)vd / (float)vy);
}
