"""The C++ classes with the reference's names (Agent, PPO_Discrete, PPO_MultiDiscrete, CartPole, MountainCar, Categorical,
CategoricalMasked, ThreadPool, CircularBuffer, PPOUtils; ppo-libtorch_amd/host/) exercised on a real GPU through the C-ABI:
environment duck type, distributions, agent, TOML keys, the reference's obs-size error text, a short train() run with
checkpoints in the reference's directories / file names, resume by newest mtime, the MultiDiscrete (masked) variant, the ThreadPool,
use_cuda = false, foreign checkpoint files, and -- against fixtures written by the compiled reference -- printPPOResults byte for byte, PPOUtils,
and the constructor's console lines + resulting hyper-parameters for a full / partial / missing PPOConfig.toml."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ppo-libtorch_amd", "host")


def test_host_sources_keep_the_reference_api_surface():
    """CPU-side: every public method / field name of the reference's headers is present in the facade headers."""
    algo = open(os.path.join(HOST, "PPO", "PPOAlgorithm.h")).read() + open(os.path.join(HOST, "PPO", "PPO_Discrete.h")).read() + \
        open(os.path.join(HOST, "PPO", "PPO_MultiDiscrete.h")).read()
    for name in ["getArgs", "loadPolicyFromCheckpoint", "computeActionLogic", "calcAdvantage", "getApproxKLAndClippedObj", "train", "initEnvs",
                 "stepEnvs", "printPPOResults", "printElement", "m_obs_size", "m_action_size", "m_learning_rate", "m_seed", "m_total_timesteps",
                 "m_use_cuda", "m_torch_deterministic", "m_num_envs", "m_num_steps", "m_anneal_lr", "m_use_gae", "m_gamma", "m_gae_lambda",
                 "m_num_minibatches", "m_update_epochs", "m_norm_adv", "m_clip_coef", "m_clip_vloss", "m_ent_coef", "m_vf_coef", "m_max_grad_norm",
                 "m_checkpoint_updates", "m_max_episode_steps", "m_batch_size", "m_minibatch_size", "m_device", "m_agent", "m_obs", "m_actions",
                 "m_logprobs", "m_rewards", "m_dones", "m_values", "m_action_masks", "m_clipfracs", "m_episode_stats", "m_global_step", "m_threadPool"]:
        assert name in algo, name
    agent = open(os.path.join(HOST, "PPO", "Agent.h")).read()
    for name in ["getValue", "getActionAndValueDiscrete", "getActionAndValueMasked", "printAgent", "AgentOutput", "m_actionSpace", "m_actionSpaceSum"]:
        assert name in agent, name
    cat = open(os.path.join(HOST, "Distributions", "Categorical.h")).read()
    for name in ["sample", "log_prob", "entropy", "mean", "mode", "variance", "enumerate_support", "logits_to_probs", "m_logits", "m_probs", "m_num_events"]:
        assert name in cat, name
    toml = open(os.path.join(HOST, "PPO", "PPOAlgorithm.cpp")).read()
    for key in ["obs_size", "action_size", "max_episode_steps", "seed", "total_timesteps", "use_cuda", "torch_deterministic", "checkpoint_updates",
                "learning_rate", "num_envs", "num_steps", "anneal_lr", "use_gae", "gamma", "gae_lambda", "num_minibatches", "update_epochs", "norm_adv",
                "clip_coef", "clip_vloss", "ent_coef", "vf_coef", "max_grad_norm", "action_high", "action_low"]:
        assert '"%s"' % key in toml, key


@pytest.mark.gpu
def test_host_facade_on_gpu(tmp_path):
    exe = os.path.join(HOST, "host_facade_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-j", "4", "-C", HOST])
    # the fixtures the compiled reference wrote (oracle/ref_harness hostgold): its console table for four crafted calls, its PPOUtils answers
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HOST_FACADE_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    # the checkpoint pair its train() run on the GPU left under ./Models/, through the compiled REFERENCE's own torch::load(m_agent, ...) /
    # torch::load(*m_optimizer, ...) (oracle/ref_harness ptload): the reference ends up holding exactly the values this build saved
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    tool = os.path.join(HOST, "torch_archive_tool")
    if os.path.exists(ref):
        import sys
        import numpy as np
        sys.path.insert(0, ROOT)
        import oracle as O
        a, o = str(tmp_path / "PPO_Agent_32768_steps.pt"), str(tmp_path / "PPO_Optimizer_32768_steps.pt")
        lr = subprocess.run([ref, "ptload", a, o, "4", "2", str(tmp_path / "loaded.pgld")], capture_output=True, text=True, timeout=120)
        assert lr.returncode == 0, lr.stderr[-2000:]
        got = O.read_pgld(str(tmp_path / "loaded.pgld"))
        saved = np.array([int(w, 16) for line in subprocess.run([tool, "dump-agent", a], capture_output=True, text=True, check=True).stdout.splitlines()
                          for w in line.split()[3:]], dtype=np.uint32)
        assert saved.size == 9155 and np.array_equal(got["params"].view(np.uint32), saved)
        assert list(got["steps"]) == [128] * 12 and got["eps"][0] == float(np.float32(1e-5))
        assert np.isfinite(got["exp_avg"]).all() and (got["exp_avg_sq"] >= 0).all() and np.abs(got["exp_avg"]).max() > 0
    # the SB3-style table of the reference's printPPOResults
    assert "ep_len_mean" in r.stdout and "policy_gradient_loss" in r.stdout and "explained_variance" in r.stdout
