"""ctypes/numpy front-end of oracle/_build/libppo_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Each wrapper mirrors one C function of ppo_oracle.h (which cites the reference lines it restates).
`read_pgld` reads the golden-vector container written by oracle/ref_harness.cpp.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libppo_oracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "ref_harness")

MAX_HEADS = 8
DIST_CATEGORICAL, DIST_MASKED = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
ENV_CARTPOLE, ENV_MOUNTAINCAR = 0, 1


def build(force=False):
    """Compile the C restatement (and, where /root/reference exists, the reference harness)."""
    src = [os.path.join(_HERE, f) for f in ("ppo_oracle.c", "ppo_oracle.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _SO


class Net(C.Structure):
    _fields_ = [("obs_size", C.c_int32), ("n_heads", C.c_int32), ("head_dims", C.c_int32 * MAX_HEADS),
                ("hidden", C.c_int32), ("n_hidden", C.c_int32), ("dist_kind", C.c_int32), ("dtype", C.c_int32)]

    @staticmethod
    def make(obs_size, head_dims, hidden=64, n_hidden=2, dist_kind=DIST_CATEGORICAL, dtype=0):
        n = Net()
        n.obs_size, n.n_heads, n.hidden, n.n_hidden, n.dist_kind = obs_size, len(head_dims), hidden, n_hidden, dist_kind
        n.dtype = dtype
        for i, d in enumerate(head_dims):
            n.head_dims[i] = d
        return n

    @property
    def act_total(self):
        return sum(self.head_dims[i] for i in range(self.n_heads))


class HParams(C.Structure):
    _fields_ = [("gamma", C.c_float), ("gae_lambda", C.c_float), ("clip_coef", C.c_float), ("ent_coef", C.c_float),
                ("vf_coef", C.c_float), ("max_grad_norm", C.c_float), ("norm_adv", C.c_int32), ("clip_vloss", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_sinf.restype = C.c_float
        _lib.orc_sinf.argtypes = [C.c_float]
        _lib.orc_cosf.restype = C.c_float
        _lib.orc_cosf.argtypes = [C.c_float]
        _lib.orc_cartpole_step.restype = C.c_float
        _lib.orc_mountaincar_step.restype = C.c_float
        _lib.orc_param_count.restype = C.c_int64
        _lib.orc_vecenv_create.restype = C.c_void_p
        _lib.orc_clip_grad_norm.restype = C.c_double
        _lib.orc_explained_variance.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------- libm / rng
def sinf(x):
    x = _f32(x).ravel()
    return np.array([lib().orc_sinf(float(v)) for v in x], dtype=np.float32)


def cosf(x):
    x = _f32(x).ravel()
    return np.array([lib().orc_cosf(float(v)) for v in x], dtype=np.float32)


def cartpole_reset_stream(seed, n_resets):
    out = np.empty((n_resets, 4), np.float32)
    lib().orc_cartpole_reset_stream(C.c_int64(seed), C.c_int64(n_resets), _p(out))
    return out


def philox4x32(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return list(out)


# ---------------------------------------------------------------------------------------------- environments
def _step_many(kind, obs_dim, state, action):
    state = _f32(state).reshape(-1, obs_dim).copy()
    action = _i64(action).ravel()
    n = state.shape[0]
    rew = np.empty(n, np.float32)
    term = np.empty(n, np.int32)
    lib().orc_step_many(C.c_int32(kind), _p(state), _p(action), C.c_int64(n), _p(rew), _p(term))
    return state, rew, term


def synthetic_obs(seed, envs, step, obs_size):
    """Observations of the synthetic env (BASELINE configs[4]) of the given global envs at global step `step`: [len(envs), obs_size]."""
    envs = np.asarray(envs, np.int64).ravel()
    out = np.empty((envs.size, obs_size), np.float32)
    for i, e in enumerate(envs):
        lib().orc_synthetic_obs(C.c_int64(seed), C.c_int64(int(e)), C.c_int64(step), C.c_int32(obs_size), _p(out[i]))
    return out


def synthetic_mask(seed, envs, step, head_dims):
    envs = np.asarray(envs, np.int64).ravel()
    hd = (C.c_int32 * len(head_dims))(*head_dims)
    out = np.empty((envs.size, sum(head_dims)), np.uint8)
    for i, e in enumerate(envs):
        lib().orc_synthetic_mask(C.c_int64(seed), C.c_int64(int(e)), C.c_int64(step), C.c_int32(len(head_dims)), hd, _p(out[i]))
    return out


def synthetic_transition(seed, envs, step):
    envs = np.asarray(envs, np.int64).ravel()
    rew, done = np.empty(envs.size, np.float32), np.empty(envs.size, np.int32)
    r, d = C.c_float(), C.c_int32()
    for i, e in enumerate(envs):
        lib().orc_synthetic_transition(C.c_int64(seed), C.c_int64(int(e)), C.c_int64(step), C.byref(r), C.byref(d))
        rew[i], done[i] = r.value, d.value
    return rew, done


def cartpole_step(state, action):
    return _step_many(0, 4, state, action)


def mountaincar_step(state, action):
    return _step_many(1, 2, state, action)


class VecEnv:
    """PPO_Discrete::initEnvs/stepEnvs semantics (PPO_Discrete.cpp:365-483)."""

    def __init__(self, kind, num_envs, seed, max_episode_steps, env_offset=0):
        self.kind, self.n = kind, num_envs
        self.obs = 4 if kind == ENV_CARTPOLE else 2
        self.h = C.c_void_p(lib().orc_vecenv_create(C.c_int32(kind), C.c_int64(num_envs), C.c_int64(seed),
                                                    C.c_int64(max_episode_steps), C.c_int64(env_offset)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_vecenv_destroy(self.h)
            self.h = None

    def init(self):
        obs = np.empty((self.n, self.obs), np.float32)
        lib().orc_vecenv_init(self.h, _p(obs))
        return obs

    def step(self, action):
        action = _i64(action).ravel()
        obs = np.empty((self.n, self.obs), np.float32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.int32)
        lib().orc_vecenv_step(self.h, _p(action), _p(obs), _p(rew), _p(done))
        return obs, rew, done

    def set_state(self, state=None, ep_len=None, ep_rew=None, reset_count=None):
        s = None if state is None else _f32(state)
        l = None if ep_len is None else _i64(ep_len)
        r = None if ep_rew is None else _f32(ep_rew)
        k = None if reset_count is None else _i64(reset_count)
        lib().orc_vecenv_set_state(self.h, _p(s), _p(l), _p(r), _p(k))

    def get_state(self):
        s = np.empty((self.n, self.obs), np.float32)
        l = np.empty(self.n, np.int64)
        r = np.empty(self.n, np.float32)
        k = np.empty(self.n, np.int64)
        lib().orc_vecenv_get_state(self.h, _p(s), _p(l), _p(r), _p(k))
        return s, l, r, k

    def episode_stats(self):
        out = (C.c_double * 3)()
        lib().orc_vecenv_episode_stats(self.h, out)
        return {"ep_len_mean": out[0], "ep_rew_mean": out[1], "count": int(out[2])}


# ---------------------------------------------------------------------------------------------- network
def param_count(net):
    return int(lib().orc_param_count(C.byref(net)))


def param_shapes(net):
    n = 2 * (net.n_hidden + 1) * 2
    out = np.empty((n, 2), np.int64)
    lib().orc_param_shapes(C.byref(net), _p(out))
    return out


def get_value(net, params, x):
    x = _f32(x).reshape(-1, net.obs_size)
    v = np.empty(x.shape[0], np.float32)
    lib().orc_get_value(C.byref(net), _p(_f32(params)), _p(x), C.c_int64(x.shape[0]), _p(v))
    return v


def actor_logits(net, params, x):
    x = _f32(x).reshape(-1, net.obs_size)
    out = np.empty((x.shape[0], net.act_total), np.float32)
    lib().orc_actor_logits(C.byref(net), _p(_f32(params)), _p(x), C.c_int64(x.shape[0]), _p(out))
    return out


def categorical(dist_kind, logits, mask=None, value=None):
    logits = _f32(logits)
    n, A = logits.shape
    mask = _u8(mask)
    value = None if value is None else _i64(value)
    ml, mp = np.empty((n, A), np.float32), np.empty((n, A), np.float32)
    lp, en = np.empty(n, np.float32), np.empty(n, np.float32)
    lib().orc_categorical(C.c_int32(dist_kind), _p(logits), _p(mask), _p(value), C.c_int64(n), C.c_int32(A),
                          _p(ml), _p(mp), _p(lp) if value is not None else None, _p(en))
    return {"m_logits": ml, "m_probs": mp, "log_prob": lp if value is not None else None, "entropy": en}


def evaluate(net, params, x, action, mask=None):
    x = _f32(x).reshape(-1, net.obs_size)
    n = x.shape[0]
    action = _i64(action).reshape(n, net.n_heads)
    mask = _u8(mask)
    lp, en, v = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    lib().orc_evaluate(C.byref(net), _p(_f32(params)), _p(x), _p(mask), _p(action), C.c_int64(n), _p(lp), _p(en), _p(v))
    return lp, en, v


def act(net, params, x, seed, step, env_offset=0, mask=None):
    x = _f32(x).reshape(-1, net.obs_size)
    n = x.shape[0]
    mask = _u8(mask)
    a = np.empty((n, net.n_heads), np.int64)
    lp, en, v = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    lib().orc_act(C.byref(net), _p(_f32(params)), _p(x), _p(mask), C.c_int64(n), C.c_int64(seed), C.c_int64(env_offset),
                  C.c_int64(step), _p(a), _p(lp), _p(en), _p(v))
    return a, lp, en, v


# ---------------------------------------------------------------------------------------------- advantages
def gae(rewards, values, dones, next_value, next_done, gamma, gae_lambda):
    rewards, values, dones = _f32(rewards), _f32(values), _f32(dones)
    T, N = rewards.shape
    nv = _f32(next_value).ravel()
    nd = np.ascontiguousarray(next_done, dtype=np.int32).ravel()
    adv, ret = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    lib().orc_gae(_p(rewards), _p(values), _p(dones), _p(nv), _p(nd), C.c_int64(T), C.c_int64(N), C.c_float(gamma),
                  C.c_float(gae_lambda), _p(adv), _p(ret))
    return adv, ret


def nstep(rewards, values, dones, next_value, next_done, gamma):
    rewards, values, dones = _f32(rewards), _f32(values), _f32(dones)
    T, N = rewards.shape
    nv = _f32(next_value).ravel()
    nd = np.ascontiguousarray(next_done, dtype=np.int32).ravel()
    adv, ret = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    lib().orc_nstep(_p(rewards), _p(values), _p(dones), _p(nv), _p(nd), C.c_int64(T), C.c_int64(N), C.c_float(gamma),
                    _p(adv), _p(ret))
    return adv, ret


# ---------------------------------------------------------------------------------------------- update
STAT_NAMES = ("pg_loss", "v_loss", "entropy_loss", "approx_kl", "clipfrac", "loss")


def minibatch_grads(net, hp, params, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, idx, b_mask=None):
    b_obs = _f32(b_obs).reshape(-1, net.obs_size)
    b_actions = _f32(b_actions)
    act_cols = 1 if b_actions.ndim == 1 else b_actions.shape[1]
    idx = _i64(idx).ravel()
    grads = np.empty(param_count(net), np.float32)
    stats = (C.c_double * 6)()
    lib().orc_minibatch_grads(C.byref(net), C.byref(hp), _p(_f32(params)), _p(b_obs), _p(b_actions), C.c_int32(act_cols),
                              _p(_u8(b_mask)), _p(_f32(b_logprobs)), _p(_f32(b_advantages)), _p(_f32(b_returns)),
                              _p(_f32(b_values)), _p(idx), C.c_int64(idx.size), _p(grads), stats)
    return grads, dict(zip(STAT_NAMES, list(stats)))


def minibatch_grads_shard(net, hp, params, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, idx, global_M, adv_sums=None,
                          b_mask=None):
    """One shard of a data-parallel minibatch step: returns (grads scaled by 1/global_M, stats share, local advantage sums)."""
    b_obs = _f32(b_obs).reshape(-1, net.obs_size)
    b_actions = _f32(b_actions)
    act_cols = 1 if b_actions.ndim == 1 else b_actions.shape[1]
    idx = _i64(idx).ravel()
    grads = np.empty(param_count(net), np.float32)
    stats = (C.c_double * 6)()
    local = (C.c_double * 2)()
    sums = None if adv_sums is None else (C.c_double * 2)(float(adv_sums[0]), float(adv_sums[1]))
    lib().orc_minibatch_grads_shard(C.byref(net), C.byref(hp), _p(_f32(params)), _p(b_obs), _p(b_actions), C.c_int32(act_cols),
                                    _p(_u8(b_mask)), _p(_f32(b_logprobs)), _p(_f32(b_advantages)), _p(_f32(b_returns)),
                                    _p(_f32(b_values)), _p(idx), C.c_int64(idx.size), sums, C.c_int64(global_M), _p(grads), stats, local)
    return grads, dict(zip(STAT_NAMES, list(stats))), (local[0], local[1])


def clip_grad_norm(net, grads, max_norm):
    g = _f32(grads).copy()
    total = lib().orc_clip_grad_norm(C.byref(net), _p(g), C.c_float(max_norm))
    return g, total


def adamw_step(params, grads, exp_avg, exp_avg_sq, lr, step_t):
    p, m, v = _f32(params).copy(), _f32(exp_avg).copy(), _f32(exp_avg_sq).copy()
    lib().orc_adamw_step(_p(p), _p(_f32(grads)), _p(m), _p(v), C.c_int64(p.size), C.c_double(lr), C.c_int64(step_t))
    return p, m, v


def explained_variance(returns, values):
    r, v = _f32(returns).ravel(), _f32(values).ravel()
    return lib().orc_explained_variance(_p(r), _p(v), C.c_int64(r.size))


# ---------------------------------------------------------------------------------------------- golden files
_DT = {0: np.float32, 1: np.int64, 2: np.int32, 3: np.uint8, 4: np.float64}


def read_pgld(path):
    """Golden-vector container written by oracle/ref_harness.cpp (format documented there)."""
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != b"PGLD1\0\0\0":
            raise ValueError("not a PGLD1 file: " + path)
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            dt, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dq" % nd, f.read(8 * nd)) if nd else ()
            dtype = np.dtype(_DT[dt])
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(n * dtype.itemsize), dtype=dtype).reshape(dims).copy()
    return out
