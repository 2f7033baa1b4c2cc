"""The checkpoint container of the host facade (ppo-libtorch_amd/host/Utils/TorchArchive.*): LibTorch module archives, as the reference writes
them with torch::save(m_agent, ...) / torch::save(*m_optimizer, ...) (PPO/PPO_Discrete.cpp:662-686) and reads them back (:782-835).

Fixtures (tests/golden/ref_*_agent.pt, ref_*_optimizer.pt, ref_*_checkpoint_values.pgld) were written by the compiled, unmodified reference
(`oracle/_ref/ref_harness ptgold`): the two files its train() left under ./Models/ and, beside them, the values its Agent and AdamW held.
  reader : every tensor, step count and option of the reference's files, bit for bit
  writer : files written by this build load (a) through this build's reader, (b) through torch.jit.load, (c) through the reference's own
           torch::load calls (`ref_harness ptload`) -- with the same values
  errors : a truncated file, a flipped byte and a file that is no archive are refused with a reason
No GPU: the container is host code."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ppo-libtorch_amd", "host")
G = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(HOST, "torch_archive_tool")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

CASES = {"discrete": (4, 2), "multidiscrete": (2, 3)}


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-s", "-C", HOST, TOOL])
    return TOOL


def _run(args, ok=True):
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    if ok:
        assert r.returncode == 0, r.stderr[-2000:]
    return r


def _bits(words):
    return np.array([int(w, 16) for w in words], dtype=np.uint32)


def _dump_agent(tool, path):
    out = []
    for line in _run([tool, "dump-agent", path]).stdout.splitlines():
        f = line.split()
        assert f[0] == "tensor"
        out.append((f[1], [int(x) for x in f[2].strip("[]").split(",") if x], _bits(f[3:])))
    return out


def _dump_optimizer(tool, path):
    opts, steps, m, v = {}, [], [], []
    for line in _run([tool, "dump-optimizer", path]).stdout.splitlines():
        f = line.split()
        if f[0] == "options":
            opts = {kv.split("=")[0]: kv.split("=")[1] for kv in f[1:]}
        elif f[0] == "step":
            steps.append(int(f[2]))
        elif f[0] == "exp_avg":
            m.append(_bits(f[3:]))
        elif f[0] == "exp_avg_sq":
            v.append(_bits(f[3:]))
    return opts, steps, m, v


def _shapes(obs, act):
    return [[64, obs], [64], [64, 64], [64], [1, 64], [1], [64, obs], [64], [64, 64], [64], [act, 64], [act]]


NAMES = ["m_Critic.criticInputLayer", "m_Critic.criticMiddleLayer", "m_Critic.criticOutputLayer",
         "m_Actor.actorInputLayer", "m_Actor.actorMiddleLayer", "m_Actor.actorOutputLayer"]


@pytest.mark.parametrize("case", sorted(CASES))
def test_reader_returns_what_the_reference_saved(tool, case):
    obs, act = CASES[case]
    want = O.read_pgld(os.path.join(G, "ref_%s_checkpoint_values.pgld" % case))
    tensors = _dump_agent(tool, os.path.join(G, "ref_%s_agent.pt" % case))
    assert [t[0] for t in tensors] == [n + s for n in NAMES for s in (".weight", ".bias")]     # Agent::parameters() order (Agent.cpp:65-66)
    assert [t[1] for t in tensors] == _shapes(obs, act)
    assert np.array_equal(np.concatenate([t[2] for t in tensors]), want["params"].view(np.uint32))
    opts, steps, m, v = _dump_optimizer(tool, os.path.join(G, "ref_%s_optimizer.pt" % case))
    assert steps == [int(s) for s in want["steps"]] and len(set(steps)) == 1 and steps[0] > 0
    assert np.array_equal(np.concatenate(m), want["exp_avg"].view(np.uint32))
    assert np.array_equal(np.concatenate(v), want["exp_avg_sq"].view(np.uint32))
    assert float.fromhex(opts["lr"]) == float(want["lr"][0])                       # the annealed rate of the last update
    assert float.fromhex(opts["eps"]) == float(np.float32(1e-5))                   # AdamWOptions(lr).eps(1e-5f), PPO_Discrete.cpp:76-78
    assert float.fromhex(opts["beta1"]) == 0.9 and float.fromhex(opts["beta2"]) == 0.999 and float.fromhex(opts["weight_decay"]) == 0.01
    assert opts["amsgrad"] == "0"


@pytest.mark.parametrize("case", sorted(CASES))
def test_written_archives_load_everywhere(tool, tmp_path, case):
    obs, act = CASES[case]
    src_a, src_o = os.path.join(G, "ref_%s_agent.pt" % case), os.path.join(G, "ref_%s_optimizer.pt" % case)
    out_a, out_o = str(tmp_path / "PPO_Agent_777_steps.pt"), str(tmp_path / "PPO_Optimizer_777_steps.pt")
    _run([tool, "rewrite", src_a, src_o, out_a, out_o, str(obs), str(act)])
    want = O.read_pgld(os.path.join(G, "ref_%s_checkpoint_values.pgld" % case))
    # (a) this build's reader
    assert [(n, s, b.tolist()) for n, s, b in _dump_agent(tool, out_a)] == [(n, s, b.tolist()) for n, s, b in _dump_agent(tool, src_a)]
    assert _run([tool, "dump-optimizer", out_o]).stdout == _run([tool, "dump-optimizer", src_o]).stdout
    # (b) PyTorch's own archive loader (the same code path LibTorch's torch::load takes)
    torch = pytest.importorskip("torch")
    mod = torch.jit.load(out_a, map_location="cpu")
    got = list(mod.named_parameters())
    assert [n for n, _ in got] == [n + s for n in NAMES for s in (".weight", ".bias")]
    assert [list(p.shape) for _, p in got] == _shapes(obs, act)
    flat = np.concatenate([p.detach().numpy().reshape(-1) for _, p in got])
    assert np.array_equal(flat.view(np.uint32), want["params"].view(np.uint32))
    assert all(p.requires_grad for _, p in got)
    opt = torch.jit.load(out_o, map_location="cpu")
    assert opt.pytorch_version == "1.5.0"
    group = getattr(opt.param_groups, "param_groups/0")
    keys = [getattr(group, "params/%d" % i) for i in range(12)]
    assert len(set(keys)) == 12
    m = np.concatenate([getattr(opt.state, k).exp_avg.numpy().reshape(-1) for k in keys])
    assert np.array_equal(m.view(np.uint32), want["exp_avg"].view(np.uint32))
    assert [getattr(opt.state, k).step for k in keys] == [int(s) for s in want["steps"]]
    assert group.options.lr == float(want["lr"][0]) and group.options.betas == (0.9, 0.999)


@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_loads_what_this_build_writes(tool, tmp_path, case):
    """(c) the reference's own torch::load(m_agent, ...) / torch::load(*m_optimizer, ...) on this build's files."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference; `make -C oracle ref`)")
    obs, act = CASES[case]
    out_a, out_o = str(tmp_path / "PPO_Agent_777_steps.pt"), str(tmp_path / "PPO_Optimizer_777_steps.pt")
    _run([tool, "rewrite", os.path.join(G, "ref_%s_agent.pt" % case), os.path.join(G, "ref_%s_optimizer.pt" % case), out_a, out_o, str(obs), str(act)])
    loaded = str(tmp_path / "loaded.pgld")
    _run([REF, "ptload", out_a, out_o, str(obs), str(act), loaded])
    got, want = O.read_pgld(loaded), O.read_pgld(os.path.join(G, "ref_%s_checkpoint_values.pgld" % case))
    for k in ("params", "exp_avg", "exp_avg_sq"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got["steps"], want["steps"])
    assert got["lr"][0] == want["lr"][0] and got["eps"][0] == float(np.float32(1e-5)) and got["weight_decay"][0] == 0.01


def test_damaged_files_are_refused_with_a_reason(tool, tmp_path):
    src = open(os.path.join(G, "ref_discrete_agent.pt"), "rb").read()
    cut = tmp_path / "cut.pt"
    cut.write_bytes(src[: len(src) // 2])
    r = _run([tool, "dump-agent", str(cut)], ok=False)
    assert r.returncode == 1 and "end-of-central-directory" in r.stderr
    flipped = bytearray(src)
    flipped[2000] ^= 0x10                      # inside storage data/2 (criticMiddleLayer.weight)
    bad = tmp_path / "flipped.pt"
    bad.write_bytes(bytes(flipped))
    r = _run([tool, "dump-agent", str(bad)], ok=False)
    assert r.returncode == 1 and "CRC" in r.stderr
    other = tmp_path / "other.pt"
    other.write_bytes(b"PPOHIP01" + bytes(64))
    r = _run([tool, "dump-agent", str(other)], ok=False)
    assert r.returncode == 1 and "checkpoint" in r.stderr
    # an agent archive is not an optimizer archive
    r = _run([tool, "dump-optimizer", os.path.join(G, "ref_discrete_agent.pt")], ok=False)
    assert r.returncode == 1 and "optimizer archive" in r.stderr


def test_hostile_sizes_are_refused_not_allocated(tool, tmp_path):
    """The reader trusts nothing the pickle says: a tensor whose sizes claim more elements than its storage record holds -- here
    (2^31 - 1) x (2^31 - 1), which would also overflow the element count -- is refused with a reason before anything is sized by it, and a size that
    is not an integer is refused too.  The archive is otherwise intact (re-packed with correct CRCs), so only the tensor record can be the reason."""
    import zipfile
    src = os.path.join(G, "ref_discrete_agent.pt")

    def repack(dst, edit):
        with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, "w", zipfile.ZIP_STORED) as zout:
            for info in zin.infolist():
                data = zin.read(info.filename)
                if info.filename.endswith("/data.pkl"):
                    data = edit(data)
                zout.writestr(info.filename, data)

    small = b"(K@K\x04t"                       # MARK, BININT1 64, BININT1 4, TUPLE: the sizes of criticInputLayer.weight
    assert small in zipfile.ZipFile(src).read(next(n for n in zipfile.ZipFile(src).namelist() if n.endswith("/data.pkl")))
    huge = tmp_path / "huge.pt"
    repack(str(huge), lambda d: d.replace(small, b"(J\xff\xff\xff\x7fJ\xff\xff\xff\x7ft", 1))
    r = _run([tool, "dump-agent", str(huge)], ok=False)
    assert r.returncode == 1 and "more elements than its storage record holds" in r.stderr, r.stderr
    odd = tmp_path / "odd.pt"
    repack(str(odd), lambda d: d.replace(small, b"(K@Nt", 1))       # a size that is None
    r = _run([tool, "dump-agent", str(odd)], ok=False)
    assert r.returncode == 1 and ("not an integer" in r.stderr or "unexpected" in r.stderr), r.stderr
