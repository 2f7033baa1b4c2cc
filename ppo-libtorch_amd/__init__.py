"""ppo-libtorch_amd: MI355X-native PPO rollout-buffer hot path behind the reference's API surface.

Layout
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/ppo_hip.h) -> libppo_hip.so
  binding.py   ctypes binding of that C-ABI (what tests and bench.py call)
  host/        C++ classes with the reference's names (Agent, PPO_Discrete, PPO_MultiDiscrete, CartPole, ...) on top of the C-ABI
  dist.py      env sharding + RCCL bootstrap over torch.distributed for N > 1 GPUs

The directory name carries a hyphen (repo convention), so it is imported through `__graft_entry__.load_package()`
under the module name `ppo_libtorch_amd`.
"""
from . import binding, dist  # noqa: F401
from .binding import *  # noqa: F401,F403
