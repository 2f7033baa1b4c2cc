"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, and exports exactly
the symbols include/ppo_hip.h declares; the ctypes mirror of ppo_config / ppo_stats matches the header; the product path
fails loudly (no CPU fallback) when no GPU is reachable."""
import ctypes as C
import os
import re
import subprocess

import pytest

from __graft_entry__ import ROOT, load_package

HDR = os.path.join(ROOT, "include", "ppo_hip.h")


@pytest.fixture(scope="module")
def P():
    subprocess.check_call(["make", "-s", "-j", "4", "-C", os.path.join(ROOT, "ppo-libtorch_amd", "csrc")])
    return load_package()


def declared_symbols():
    src = open(HDR).read()
    return sorted(set(re.findall(r"^PPO_API\s+[\w\s\*]+?\b(ppo_\w+)\s*\(", src, flags=re.M)))


def test_library_exports_every_declared_symbol(P):
    decl = declared_symbols()
    assert len(decl) >= 40
    assert sorted(P.binding.ABI_SYMBOLS) == decl
    lib = P.binding.lib()
    for name in decl:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", P.binding.LIB_PATH]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert [e for e in exported if e.startswith("ppo_")] == decl
    assert all(e.startswith("ppo_") or e.startswith("_") for e in exported)  # nothing else leaks from the C-ABI
    assert lib.ppo_abi_version() == P.binding.ABI_VERSION == 5


def test_struct_mirrors_match_header(P):
    # compile a probe against the header and compare sizeof / offsetof with the ctypes mirrors
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "ppo_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ppo_config), offsetof(ppo_config, head_dims), offsetof(ppo_config, seed),
         offsetof(ppo_config, learning_rate), offsetof(ppo_config, max_grad_norm), sizeof(ppo_stats), offsetof(ppo_stats, ep_count),
         sizeof(ppo_profile), offsetof(ppo_profile, allreduce_ms), offsetof(ppo_profile, vector_fallback_launches));
  return 0;
}'''
    exe = "/tmp/ppo_abi_probe"
    subprocess.run(["gcc", "-x", "c", "-I", os.path.join(ROOT, "include"), "-o", exe, "-"], input=probe.encode(), check=True)
    got = [int(x) for x in subprocess.check_output([exe]).split()]
    Cfg, St, Pr = P.binding.Config, P.binding.Stats, P.binding.Profile
    assert got == [C.sizeof(Cfg), Cfg.head_dims.offset, Cfg.seed.offset, Cfg.learning_rate.offset, Cfg.max_grad_norm.offset,
                   C.sizeof(St), St.ep_count.offset, C.sizeof(Pr), Pr.allreduce_ms.offset, Pr.vector_fallback_launches.offset]


def test_header_cites_reference_lines():
    src = open(HDR).read()
    # every API block names the reference interface it replaces (file:line)
    assert len(re.findall(r"\.(?:cpp|h):\d+", src)) >= 40


def test_no_cpu_fallback_in_product_path():
    pkg = os.path.join(ROOT, "ppo-libtorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                # comments may NAME the oracle file a device function mirrors; nothing may include, import, link or load it
                for pat in (r"^\s*(import|from)\s+oracle\b", r"#\s*include\s*[\"<][^\n]*ppo_oracle", r"libppo_oracle", r"oracle/_build", r"oracle/_ref"):
                    assert not re.search(pat, text, flags=re.M), (f, pat)


def test_context_creation_fails_loudly_without_gpu(P):
    import shutil
    if os.path.exists("/dev/kfd") and shutil.which("rocminfo"):
        pytest.skip("a GPU is present")
    with pytest.raises(P.binding.PPOError):
        P.Context(P.make_config())


def test_no_environment_switches_in_product():
    """Which kernel runs a stage is ppo_config.kernel_flags (include/ppo_hip.h PPO_KERNEL_*): the library reads nothing from the environment, and the
    Python layer only PPO_HIP_LIBRARY (binding.py: which build of the same C-ABI to load) and the rendezvous variables of torch.distributed (dist.py)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "ppo-libtorch_amd", "csrc", "*")) + glob.glob(os.path.join(root, "ppo-libtorch_amd", "host", "**", "*.*"), recursive=True):
        if f.endswith((".hip", ".hpp", ".cpp", ".h")):
            assert "getenv" not in open(f).read(), f
    names = set()
    for f in ("binding.py", "dist.py", "__init__.py"):
        names |= set(re.findall(r"environ(?:\.get\(|\[)\s*[\"']([A-Z_0-9]+)", open(os.path.join(root, "ppo-libtorch_amd", f)).read()))
    assert names <= {"PPO_HIP_LIBRARY", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"}, names
