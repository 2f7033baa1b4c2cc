"""`python bench.py --gpus N` (no torchrun environment) must start N fresh rank processes by itself -- before anything touches HIP -- relay rank 0's ONE
JSON line and exit with the ranks' return code.  No GPU here: the launcher is pointed at a stand-in worker
(--launch-worker) that does what a rank does on the host side: gloo rendezvous from the torchrun environment, a barrier, rank 0 prints the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""
import argparse, json, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int); ap.add_argument("--steps", type=int); ap.add_argument("--warmup", type=int)
ap.add_argument("--transport", default="auto"); ap.add_argument("--fallback-reason", default=None); ap.add_argument("--launch-worker")
ap.add_argument("--fail-on", default="")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" and os.environ["MASTER_ADDR"] == "127.0.0.1"
import torch.distributed as dist
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
print("rank %d says hello on stdout" % rank, flush=True)     # noise: must not reach the launcher's stdout
if args.transport in args.fail_on.split(","):
    sys.exit(7)
if rank == 0:
    print(json.dumps({"metric": "env-steps/sec (rollout+update)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "transport": args.transport,
                      "transport_fallback_reason": args.fallback_reason}), flush=True)
try:                      # (as bench.py: a peer that has already left is not a failed run)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:
    sys.stderr.write("teardown: %s\n" % e)
"""


def _run(tmp_path, extra):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-worker", str(stub)] + extra,
                          capture_output=True, text=True, env=env, timeout=300)


def test_plain_command_starts_the_ranks_and_relays_one_line(tmp_path):
    r = _run(tmp_path, [])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["transport"] == "auto" and d["transport_fallback_reason"] is None
    assert "rank 1 says hello" in r.stderr      # the ranks' other output is relayed on stderr


def test_failed_run_is_retried_once_in_fresh_processes_on_the_other_transport(tmp_path):
    r = _run(tmp_path, ["--fail-on", "auto"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["transport"] == "exchange" and "exited with rc" in d["transport_fallback_reason"]


def test_failure_is_the_exit_code_and_no_line(tmp_path):
    r = _run(tmp_path, ["--fail-on", "auto,exchange"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = _run(tmp_path, ["--transport", "rccl", "--fail-on", "rccl"])     # an explicit transport is not retried on another one
    assert r.returncode != 0 and not r.stdout.strip()
