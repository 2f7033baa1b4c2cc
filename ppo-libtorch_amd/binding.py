"""ctypes binding of libppo_hip.so (include/ppo_hip.h) -- the thin host layer tests and bench.py drive.

Pure ctypes + numpy: device memory is allocated through the C-ABI itself (ppo_device_alloc / ppo_memcpy_*), so no
PyTorch is needed on the single-GPU path.  There is NO CPU fallback: if the HIP library is missing or the GPU
cannot be reached, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPO_HIP_LIBRARY: another build of the SAME C-ABI (tools/ab.sh times variant builds side by side without copying over the shipped file);
# it must export every symbol of include/ppo_hip.h like the shipped library or lib() raises
LIB_PATH = os.environ.get("PPO_HIP_LIBRARY") or os.path.join(_HERE, "libppo_hip.so")

MAX_HEADS = 8
ENV_CARTPOLE, ENV_MOUNTAINCAR, ENV_SYNTHETIC = 0, 1, 2
MM_EPI_NONE, MM_EPI_BIAS, MM_EPI_BIAS_TANH, MM_EPI_DTANH = 0, 1, 2, 3
MM_F32X3, MM_BF16 = 0, 1
DIST_CATEGORICAL, DIST_MASKED = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
KERNEL_ROLLOUT_VECTOR, KERNEL_UPDATE_VECTOR, KERNEL_UPDATE_ONE_WAVE, KERNEL_COMM_SELFTEST, KERNEL_GENERIC_CLASSIC, KERNEL_GENERIC_SPLIT_HEAD = 1, 2, 4, 8, 16, 32   # ppo_config.kernel_flags (include/ppo_hip.h PPO_KERNEL_*)
ABI_VERSION = 5
COMM_ID_BYTES = 128
COMM_HANDLE_BYTES = 64

BUF = dict(OBS=0, ACTIONS=1, LOGPROBS=2, REWARDS=3, DONES=4, VALUES=5, MASKS=6, ADVANTAGES=7, RETURNS=8, NEXT_OBS=9,
           NEXT_DONE=10, NEXT_VALUE=11, PARAMS=12, GRADS=13, EXP_AVG=14, EXP_AVG_SQ=15, ENV_STATE=16, EP_LEN=17, EP_REW=18,
           RESET_COUNT=19, PERM=20, FIN_LEN=21, FIN_REW=22)
_BUF_DTYPE = dict(OBS=np.float32, ACTIONS=np.int32, LOGPROBS=np.float32, REWARDS=np.float32, DONES=np.float32,
                  VALUES=np.float32, MASKS=np.uint8, ADVANTAGES=np.float32, RETURNS=np.float32, NEXT_OBS=np.float32,
                  NEXT_DONE=np.int32, NEXT_VALUE=np.float32, PARAMS=np.float32, GRADS=np.float32, EXP_AVG=np.float32,
                  EXP_AVG_SQ=np.float32, ENV_STATE=np.float32, EP_LEN=np.int32, EP_REW=np.float32, RESET_COUNT=np.int32,
                  PERM=np.int32, FIN_LEN=np.int32, FIN_REW=np.float32)

# every symbol include/ppo_hip.h declares (tests/test_abi_symbols.py checks the built library exports exactly these)
ABI_SYMBOLS = [
    "ppo_abi_version", "ppo_ctx_create", "ppo_ctx_destroy", "ppo_last_error", "ppo_sync", "ppo_stream", "ppo_get_config",
    "ppo_buffer", "ppo_device_alloc", "ppo_device_free", "ppo_memcpy_h2d", "ppo_memcpy_d2h", "ppo_param_count",
    "ppo_param_shapes", "ppo_params_init_orthogonal", "ppo_params_set_h", "ppo_params_get_h", "ppo_optimizer_set_h",
    "ppo_optimizer_get_h", "ppo_get_value", "ppo_policy_act", "ppo_categorical", "ppo_categorical_sample", "ppo_matmul", "ppo_env_transition",
    "ppo_cartpole_reset_stream_h", "ppo_env_reset", "ppo_env_step", "ppo_env_set_state_h", "ppo_env_get_state_h",
    "ppo_rollout", "ppo_calc_advantage", "ppo_gae", "ppo_gae_fast", "ppo_nstep_returns", "ppo_generate_permutations",
    "ppo_minibatch_forward_backward", "ppo_allreduce_grads", "ppo_optimizer_step", "ppo_update", "ppo_train_iteration",
    "ppo_read_stats", "ppo_set_learning_rate", "ppo_profile_enable", "ppo_profile_read", "ppo_comm_unique_id", "ppo_comm_init",
    "ppo_comm_init_local", "ppo_comm_exchange_handle", "ppo_comm_init_exchange", "ppo_comm_exchange_timeouts",
    "ppo_comm_set_wait_limit", "ppo_stats_snapshot", "ppo_stats_snapshot_read",
]


class Config(C.Structure):
    """ppo_config: the m_* hyper-parameters of PPO_Discrete (reference PPO/PPO_Discrete.h:52-85)."""
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("env_kind", C.c_int32), ("dist_kind", C.c_int32),
                ("obs_size", C.c_int32), ("n_heads", C.c_int32), ("head_dims", C.c_int32 * MAX_HEADS), ("hidden", C.c_int32),
                ("n_hidden", C.c_int32), ("num_envs", C.c_int32), ("num_steps", C.c_int32), ("num_minibatches", C.c_int32),
                ("update_epochs", C.c_int32), ("max_episode_steps", C.c_int32), ("use_gae", C.c_int32), ("norm_adv", C.c_int32),
                ("clip_vloss", C.c_int32), ("anneal_lr", C.c_int32), ("seed", C.c_int64), ("total_timesteps", C.c_int64),
                ("env_offset", C.c_int64), ("global_num_envs", C.c_int64), ("learning_rate", C.c_float), ("gamma", C.c_float),
                ("gae_lambda", C.c_float), ("clip_coef", C.c_float), ("ent_coef", C.c_float), ("vf_coef", C.c_float),
                ("max_grad_norm", C.c_float), ("compute_dtype", C.c_int32), ("kernel_flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("pg_loss", C.c_double), ("v_loss", C.c_double), ("entropy_loss", C.c_double), ("approx_kl", C.c_double),
                ("loss", C.c_double), ("clipfrac_last", C.c_double), ("clipfrac_mean", C.c_double), ("total_norm", C.c_double),
                ("explained_variance", C.c_double), ("learning_rate", C.c_double), ("ep_len_mean", C.c_double),
                ("ep_rew_mean", C.c_double), ("ep_count", C.c_int64), ("global_step", C.c_int64), ("optimizer_steps", C.c_int64),
                ("updates", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Profile(C.Structure):
    _fields_ = [("fwd_bwd_launches", C.c_int64), ("gae_launches", C.c_int64), ("rollout_launches", C.c_int64),
                ("optimizer_launches", C.c_int64), ("reduce_launches", C.c_int64), ("fwd_bwd_ms", C.c_double), ("gae_ms", C.c_double),
                ("rollout_ms", C.c_double), ("optimizer_ms", C.c_double), ("reduce_ms", C.c_double), ("phase_cycles", C.c_double * 24),
                ("allreduce_launches", C.c_int64), ("allreduce_ms", C.c_double), ("vector_fallback_launches", C.c_int64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "phase_cycles"}
        d["phase_cycles"] = list(self.phase_cycles)
        return d


def make_config(env_kind=ENV_CARTPOLE, dist_kind=DIST_CATEGORICAL, obs_size=4, head_dims=(2,), num_envs=8, num_steps=32,
                num_minibatches=4, update_epochs=10, max_episode_steps=500, use_gae=True, norm_adv=True, clip_vloss=True,
                anneal_lr=True, seed=2, total_timesteps=100000, env_offset=0, global_num_envs=0, learning_rate=1e-3, gamma=0.98,
                gae_lambda=0.95, clip_coef=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, device=0, hidden=64, n_hidden=2,
                compute_dtype=0, kernel_flags=0):
    """Defaults = Environments/CartPoleRecommendedSettings.toml of the reference with action_size = 2."""
    c = Config()
    c.struct_size = C.sizeof(Config)
    c.device, c.env_kind, c.dist_kind, c.obs_size = device, env_kind, dist_kind, obs_size
    c.n_heads = len(head_dims)
    for i, d in enumerate(head_dims):
        c.head_dims[i] = d
    c.hidden, c.n_hidden = hidden, n_hidden
    c.num_envs, c.num_steps, c.num_minibatches, c.update_epochs = num_envs, num_steps, num_minibatches, update_epochs
    c.max_episode_steps = max_episode_steps
    c.use_gae, c.norm_adv, c.clip_vloss, c.anneal_lr = int(use_gae), int(norm_adv), int(clip_vloss), int(anneal_lr)
    c.seed, c.total_timesteps, c.env_offset, c.global_num_envs = seed, total_timesteps, env_offset, global_num_envs
    c.learning_rate, c.gamma, c.gae_lambda, c.clip_coef = learning_rate, gamma, gae_lambda, clip_coef
    c.ent_coef, c.vf_coef, c.max_grad_norm = ent_coef, vf_coef, max_grad_norm
    c.compute_dtype = compute_dtype
    c.kernel_flags = kernel_flags   # KERNEL_*: include/ppo_hip.h PPO_KERNEL_*
    return c


_lib = None


def lib():
    """Loads the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libppo_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the hot path)")
        L = C.CDLL(LIB_PATH)
        L.ppo_last_error.restype = C.c_char_p
        L.ppo_last_error.argtypes = [C.c_void_p]
        L.ppo_stream.restype = C.c_void_p
        L.ppo_param_count.restype = C.c_int64
        L.ppo_ctx_destroy.restype = None
        for name in ABI_SYMBOLS:
            getattr(L, name)  # AttributeError if the build lacks a declared symbol
        if L.ppo_abi_version() != ABI_VERSION:
            raise RuntimeError("libppo_hip.so ABI version mismatch")
        _lib = L
    return _lib


class PPOError(RuntimeError):
    pass


def _check(status, ctx=None):
    if status != 0:
        msg = lib().ppo_last_error(ctx)
        raise PPOError("ppo_hip status %d: %s" % (status, msg.decode() if msg else "?"))


class DeviceArray:
    """A typed device allocation made through the C-ABI (freed with the context or explicitly)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(s) for s in np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(lib().ppo_device_alloc(ctx.h, C.c_size_t(self.nbytes), C.byref(p)), ctx.h)
        self.ptr = C.c_void_p(p.value)
        ctx._arrays.append(self)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        _check(lib().ppo_memcpy_h2d(self.ctx.h, self.ptr, host.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes)), self.ctx.h)
        return self

    def download(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _check(lib().ppo_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.nbytes)), self.ctx.h)
        return out

    def free(self):
        if self.ptr is not None and self.ctx.h:
            lib().ppo_device_free(self.ctx.h, self.ptr)
        self.ptr = None


class Context:
    """One ppo_ctx: a PPO_Discrete / PPO_MultiDiscrete instance living on one GPU."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._arrays = []
        h = C.c_void_p()
        st = lib().ppo_ctx_create(C.byref(cfg), C.byref(h))
        if st != 0:
            msg = lib().ppo_last_error(None)
            raise PPOError("ppo_ctx_create status %d: %s" % (st, msg.decode() if msg else "?"))
        self.h = h
        self.T, self.N, self.O, self.H = cfg.num_steps, cfg.num_envs, cfg.obs_size, cfg.n_heads
        self.A = sum(cfg.head_dims[i] for i in range(cfg.n_heads))
        self.B = self.T * self.N
        self.P = int(lib().ppo_param_count(self.h))

    def close(self):
        if getattr(self, "h", None):
            for a in self._arrays:
                a.free()
            lib().ppo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing
    def dev(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, host.dtype).upload(host)

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def sync(self):
        _check(lib().ppo_sync(self.h), self.h)

    def stream(self):
        return lib().ppo_stream(self.h)

    def buffer_ptr(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().ppo_buffer(self.h, BUF[name], C.byref(p), C.byref(n)), self.h)
        return p, n.value

    def read(self, name, shape=None):
        p, n = self.buffer_ptr(name)
        out = np.empty(n // np.dtype(_BUF_DTYPE[name]).itemsize, _BUF_DTYPE[name])
        _check(lib().ppo_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), p, C.c_size_t(n)), self.h)
        return out.reshape(shape) if shape is not None else out

    def write(self, name, host):
        p, n = self.buffer_ptr(name)
        host = np.ascontiguousarray(host, dtype=_BUF_DTYPE[name])
        assert host.nbytes == n, (name, host.nbytes, n)
        _check(lib().ppo_memcpy_h2d(self.h, p, host.ctypes.data_as(C.c_void_p), C.c_size_t(n)), self.h)

    # ---- Agent
    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        _check(lib().ppo_params_set_h(self.h, p.ctypes.data_as(C.c_void_p), C.c_int64(p.size)), self.h)

    def get_params(self):
        p = np.empty(self.P, np.float32)
        _check(lib().ppo_params_get_h(self.h, p.ctypes.data_as(C.c_void_p), C.c_int64(self.P)), self.h)
        return p

    def init_orthogonal(self, seed):
        _check(lib().ppo_params_init_orthogonal(self.h, C.c_int64(seed)), self.h)

    def param_shapes(self):
        shapes = np.empty((32, 2), np.int64)   # up to 2 nets x 8 layers x {weight, bias}
        n = C.c_int32()
        _check(lib().ppo_param_shapes(self.h, shapes.ctypes.data_as(C.c_void_p), C.byref(n)), self.h)
        return shapes[:n.value]

    def set_optimizer(self, m, v, step):
        m, v = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(v, np.float32)
        _check(lib().ppo_optimizer_set_h(self.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_int64(m.size),
                                         C.c_int64(step)), self.h)

    def get_optimizer(self):
        m, v = np.empty(self.P, np.float32), np.empty(self.P, np.float32)
        step = C.c_int64()
        _check(lib().ppo_optimizer_get_h(self.h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_int64(self.P),
                                         C.byref(step)), self.h)
        return m, v, step.value

    def get_value(self, obs):
        """Agent::getValue (reference Agent.cpp:107-109)."""
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.O)
        d_obs, d_v = self.dev(obs), self.empty(obs.shape[0], np.float32)
        _check(lib().ppo_get_value(self.h, d_obs.ptr, C.c_int64(obs.shape[0]), d_v.ptr), self.h)
        return d_v.download()

    def policy_act(self, obs, mask=None, action=None, step_index=0):
        """Agent::getActionAndValueDiscrete / getActionAndValueMasked (reference Agent.cpp:117-170)."""
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.O)
        n = obs.shape[0]
        d_obs = self.dev(obs)
        d_mask = self.dev(mask, np.uint8) if mask is not None else None
        d_forced = self.dev(np.asarray(action).reshape(n, self.H), np.int64) if action is not None else None
        d_a, d_lp, d_en, d_v = self.empty((n, self.H), np.int64), self.empty(n, np.float32), self.empty(n, np.float32), self.empty(n, np.float32)
        _check(lib().ppo_policy_act(self.h, d_obs.ptr, d_mask.ptr if d_mask else None, d_forced.ptr if d_forced else None, C.c_int64(n),
                                    C.c_int64(step_index), d_a.ptr, d_lp.ptr, d_en.ptr, d_v.ptr), self.h)
        return d_a.download(), d_lp.download(), d_en.download(), d_v.download()

    # ---- Environments
    def env_reset(self):
        """PPO_Discrete::initEnvs (reference PPO_Discrete.cpp:365-402)."""
        _check(lib().ppo_env_reset(self.h), self.h)
        return self.read("NEXT_OBS", (self.N, self.O))

    def env_step(self, action):
        """PPO_Discrete::stepEnvs (reference PPO_Discrete.cpp:413-483)."""
        d_a = self.dev(np.asarray(action).reshape(self.N, self.H), np.int64)
        d_o, d_r, d_d = self.empty((self.N, self.O), np.float32), self.empty(self.N, np.float32), self.empty(self.N, np.int32)
        _check(lib().ppo_env_step(self.h, d_a.ptr, d_o.ptr, d_r.ptr, d_d.ptr), self.h)
        return d_o.download(), d_r.download(), d_d.download()

    def env_set_state(self, state=None, ep_len=None, ep_rew=None, reset_count=None):
        def p(a, dt):
            return None if a is None else np.ascontiguousarray(a, dt)
        s, l, r, k = p(state, np.float32), p(ep_len, np.int32), p(ep_rew, np.float32), p(reset_count, np.int32)
        _check(lib().ppo_env_set_state_h(self.h, *(x.ctypes.data_as(C.c_void_p) if x is not None else None for x in (s, l, r, k))), self.h)

    def env_get_state(self):
        s, l = np.empty((self.N, self.O), np.float32), np.empty(self.N, np.int32)
        r, k = np.empty(self.N, np.float32), np.empty(self.N, np.int32)
        _check(lib().ppo_env_get_state_h(self.h, *(x.ctypes.data_as(C.c_void_p) for x in (s, l, r, k))), self.h)
        return s, l, r, k

    # ---- rollout / advantages / update
    def rollout(self, forced_actions=None):
        d = self.dev(np.asarray(forced_actions).reshape(self.T, self.N, self.H), np.int64) if forced_actions is not None else None
        _check(lib().ppo_rollout(self.h, d.ptr if d else None), self.h)

    def calc_advantage(self):
        _check(lib().ppo_calc_advantage(self.h), self.h)
        return self.read("ADVANTAGES", (self.T, self.N)), self.read("RETURNS", (self.T, self.N))

    def generate_permutations(self):
        _check(lib().ppo_generate_permutations(self.h), self.h)
        return self.read("PERM", (self.cfg.update_epochs, self.B))

    def minibatch_forward_backward(self, idx):
        d = self.dev(idx, np.int32)
        _check(lib().ppo_minibatch_forward_backward(self.h, d.ptr, C.c_int64(d.shape[0])), self.h)
        return self.read("GRADS")

    def optimizer_step(self):
        _check(lib().ppo_optimizer_step(self.h), self.h)

    def set_learning_rate(self, lr):
        _check(lib().ppo_set_learning_rate(self.h, C.c_double(lr)), self.h)

    def update(self):
        _check(lib().ppo_update(self.h), self.h)

    def train_iteration(self):
        _check(lib().ppo_train_iteration(self.h), self.h)

    def stats(self):
        s = Stats()
        _check(lib().ppo_read_stats(self.h, C.byref(s)), self.h)
        return s.as_dict()

    def stats_snapshot(self):
        """Enqueue a statistics snapshot behind the work enqueued so far (ppo_hip.h); read it later with stats_snapshot_read()."""
        _check(lib().ppo_stats_snapshot(self.h), self.h)

    def stats_snapshot_read(self):
        s = Stats()
        _check(lib().ppo_stats_snapshot_read(self.h, C.byref(s)), self.h)
        return s.as_dict()

    def profile_enable(self, on=1):
        """0 = off, 1 = every instrumented launch, 2 = only the dominant kernel (1 launch in 8) and the GAE scan, 4 = the same with 1 launch in 41."""
        _check(lib().ppo_profile_enable(self.h, C.c_int32(int(on))), self.h)

    def profile_read(self):
        p = Profile()
        _check(lib().ppo_profile_read(self.h, C.byref(p)), self.h)
        return p.as_dict()

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(lib().ppo_comm_init(self.h, buf, C.c_int32(rank), C.c_int32(nranks)), self.h)


def _comm_init_local(self, group_id, rank, nranks):
    _check(lib().ppo_comm_init_local(self.h, C.c_int64(group_id), C.c_int32(rank), C.c_int32(nranks)), self.h)


Context.comm_init_local = _comm_init_local


def _comm_exchange_handle(self):
    """IPC handle of this context's exchange buffer (one-shot direct exchange; ppo_hip.h)."""
    buf = (C.c_char * COMM_HANDLE_BYTES)()
    _check(lib().ppo_comm_exchange_handle(self.h, buf), self.h)
    return bytes(buf)


def _comm_init_exchange(self, handles, rank, nranks):
    blob = b"".join(handles)
    assert len(blob) == nranks * COMM_HANDLE_BYTES
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    _check(lib().ppo_comm_init_exchange(self.h, buf, C.c_int32(rank), C.c_int32(nranks)), self.h)


def _comm_exchange_timeouts(self):
    n = C.c_int32()
    _check(lib().ppo_comm_exchange_timeouts(self.h, C.byref(n)), self.h)
    return n.value


def _comm_set_wait_limit(self, seconds):
    """How long a direct-exchange kernel waits for a peer before it gives up (default 30 s; ppo_hip.h)."""
    _check(lib().ppo_comm_set_wait_limit(self.h, C.c_double(seconds)), self.h)


Context.comm_set_wait_limit = _comm_set_wait_limit
Context.comm_exchange_handle = _comm_exchange_handle
Context.comm_init_exchange = _comm_init_exchange
Context.comm_exchange_timeouts = _comm_exchange_timeouts


def comm_unique_id():
    buf = (C.c_char * COMM_ID_BYTES)()
    st = lib().ppo_comm_unique_id(buf)
    if st != 0:
        raise PPOError("ppo_comm_unique_id failed: %s" % lib().ppo_last_error(None).decode())
    return bytes(buf)


# ---- stateless entry points (need a context only for device memory plumbing)
def gae(ctx, rewards, values, dones, next_value, next_done, gamma, gae_lambda, nstep=False, fast=False):
    """PPO_Discrete::calcAdvantage on caller buffers (reference PPO_Discrete.cpp:274-331).  fast: the associative scan (ppo_gae_fast; not bit-identical)."""
    rewards = np.ascontiguousarray(rewards, np.float32)
    T, N = rewards.shape
    d = [ctx.dev(rewards), ctx.dev(values, np.float32), ctx.dev(dones, np.float32), ctx.dev(np.ravel(next_value), np.float32),
         ctx.dev(np.ravel(next_done), np.int32)]
    adv, ret = ctx.empty((T, N), np.float32), ctx.empty((T, N), np.float32)
    if nstep:
        st = lib().ppo_nstep_returns(*(x.ptr for x in d), C.c_int64(T), C.c_int64(N), C.c_float(gamma), adv.ptr, ret.ptr, C.c_void_p(ctx.stream()))
    else:
        fn = lib().ppo_gae_fast if fast else lib().ppo_gae
        st = fn(*(x.ptr for x in d), C.c_int64(T), C.c_int64(N), C.c_float(gamma), C.c_float(gae_lambda), adv.ptr, ret.ptr, C.c_void_p(ctx.stream()))
    _check(st, ctx.h)
    ctx.sync()
    out = adv.download(), ret.download()
    for x in d + [adv, ret]:
        x.free()
    return out


def gae_launch(ctx, d_rewards, d_values, d_dones, d_next_value, d_next_done, T, N, gamma, gae_lambda, d_adv, d_ret, fast=False):
    """Enqueues the scan on device arrays that already live in HBM (no copies, no synchronisation)."""
    _check((lib().ppo_gae_fast if fast else lib().ppo_gae)(d_rewards.ptr, d_values.ptr, d_dones.ptr, d_next_value.ptr, d_next_done.ptr, C.c_int64(T), C.c_int64(N), C.c_float(gamma),
                         C.c_float(gae_lambda), d_adv.ptr, d_ret.ptr, C.c_void_p(ctx.stream())), ctx.h)


def env_transition(ctx, env_kind, state, action):
    state = np.ascontiguousarray(state, np.float32)
    n, O = state.shape
    d_s, d_a = ctx.dev(state), ctx.dev(np.ravel(action), np.int64)
    d_ns, d_r, d_t = ctx.empty((n, O), np.float32), ctx.empty(n, np.float32), ctx.empty(n, np.int32)
    _check(lib().ppo_env_transition(C.c_int32(env_kind), d_s.ptr, d_a.ptr, C.c_int64(n), d_ns.ptr, d_r.ptr, d_t.ptr, C.c_void_p(ctx.stream())), ctx.h)
    ctx.sync()
    return d_ns.download(), d_r.download(), d_t.download()


def categorical(ctx, dist_kind, logits, mask=None, value=None):
    logits = np.ascontiguousarray(logits, np.float32)
    n, A = logits.shape
    d_l = ctx.dev(logits)
    d_m = ctx.dev(mask, np.uint8) if mask is not None else None
    d_v = ctx.dev(np.ravel(value), np.int64) if value is not None else None
    o = dict(m_logits=ctx.empty((n, A), np.float32), m_probs=ctx.empty((n, A), np.float32), log_prob=ctx.empty(n, np.float32),
             entropy=ctx.empty(n, np.float32), mode=ctx.empty(n, np.int64))
    _check(lib().ppo_categorical(C.c_int32(dist_kind), d_l.ptr, d_m.ptr if d_m else None, d_v.ptr if d_v else None, C.c_int64(n), C.c_int32(A),
                                 o["m_logits"].ptr, o["m_probs"].ptr, o["log_prob"].ptr, o["entropy"].ptr, o["mode"].ptr,
                                 C.c_void_p(ctx.stream())), ctx.h)
    ctx.sync()
    return {k: v.download() for k, v in o.items()}


def matmul_launch(ctx, trans_a, trans_b, M, N, K, d_a, lda, d_b, ldb, d_c, ldc, epilogue=MM_EPI_NONE, d_aux=None, ld_aux=0, precision=MM_F32X3):
    """Enqueues ppo_matmul on device arrays (no copies, no synchronisation)."""
    _check(lib().ppo_matmul(C.c_int32(int(trans_a)), C.c_int32(int(trans_b)), C.c_int64(M), C.c_int64(N), C.c_int64(K), d_a.ptr, C.c_int64(lda), d_b.ptr,
                            C.c_int64(ldb), d_c.ptr, C.c_int64(ldc), C.c_int32(epilogue), d_aux.ptr if d_aux is not None else None, C.c_int64(ld_aux),
                            C.c_int32(precision), C.c_void_p(ctx.stream())), ctx.h)


def matmul(ctx, a, b, trans_a=False, trans_b=False, epilogue=MM_EPI_NONE, aux=None, precision=MM_F32X3):
    """c[M, N] = epilogue(sum_k A(m, k) B(n, k)) with A = a or a^T, B = b or b^T as stored (see ppo_matmul in ppo_hip.h)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    if trans_a:
        K, M = a.shape
    else:
        M, K = a.shape
    N = b.shape[1] if trans_b else b.shape[0]
    assert (b.shape[0] if trans_b else b.shape[1]) == K
    d_a, d_b, d_c = ctx.dev(a), ctx.dev(b), ctx.empty((M, N), np.float32)
    d_x = ctx.dev(aux, np.float32) if aux is not None else None
    matmul_launch(ctx, trans_a, trans_b, M, N, K, d_a, a.shape[1], d_b, b.shape[1], d_c, N, epilogue, d_x, N if epilogue == MM_EPI_DTANH else 0, precision)
    ctx.sync()
    out = d_c.download()
    for x in (d_a, d_b, d_c) + ((d_x,) if d_x is not None else ()):
        x.free()
    return out


def cartpole_reset_stream(seed, n_resets):
    out = np.empty((n_resets, 4), np.float32)
    st = lib().ppo_cartpole_reset_stream_h(C.c_int64(seed), C.c_int64(n_resets), out.ctypes.data_as(C.c_void_p))
    if st != 0:
        raise PPOError("ppo_cartpole_reset_stream_h failed")
    return out
