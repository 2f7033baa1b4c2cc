#!/bin/bash
# Times the drop-in itself: PPO_driver (the C++ facade with the reference's class names, host/driver.cpp) on BASELINE configs[1] -- CartPole-v1,
# 4096 envs x 128 steps, action_size = 2, the recommended hyper-parameters -- as the reference would be run: ./PPOConfig.toml in the CWD, the console
# table printed every update.  `fps` is defined as what printPPOResults prints (PPO_Discrete.cpp:650-652,718), but the reference casts an update's
# duration to WHOLE MILLISECONDS first, so at 2.4 ms per update the printed column reads 262144000 or 174762666; the wall clock per update is therefore
# taken from two runs of different length: (wall(1100 updates) - wall(100 updates)) / 1000 -- start-up, code-object load and the final checkpoint cancel.
#   tools/facade_fps.sh [OUT.txt]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
OUT=${1:-$ROOT/gpurun_out/facade_fps.txt}
run() {   # $1 = updates
    D=$(mktemp -d /tmp/facade_XXXXXX)
    cat > $D/PPOConfig.toml <<TOML
[environment]
obs_size = 4
action_size = 2
max_episode_steps = 500
[general]
seed = 2
total_timesteps = $(( $1 * 4096 * 128 ))
use_cuda = true
torch_deterministic = true
checkpoint_updates = 1000000
[ppo]
learning_rate = 0.001
num_envs = 4096
num_steps = 128
anneal_lr = true
use_gae = true
gamma = 0.98
gae_lambda = 0.95
num_minibatches = 4
update_epochs = 10
norm_adv = true
clip_coef = 0.2
clip_vloss = true
ent_coef = 0.0
vf_coef = 0.5
max_grad_norm = 0.5
TOML
    ( cd $D && s=$(date +%s.%N) && $ROOT/ppo-libtorch_amd/host/PPO_driver > out.txt 2> err.txt; e=$(date +%s.%N); echo "$s $e" > wall.txt )
    python3 - $D $1 <<'PY'
import sys, re
d, n = sys.argv[1], int(sys.argv[2])
s, e = map(float, open(d + "/wall.txt").read().split())
out = open(d + "/out.txt").read()
fps = [int(x) for x in re.findall(r"fps\s+\|\s+(\d+)", out)]
its = re.findall(r"iterations\s+\|\s+(\d+)", out)
ep = re.findall(r"ep_len_mean\s+\|\s+(\S+)", out)
print("updates %d wall %.4f s tables %d last_iteration %s fps_column_median %d fps_column_min %d fps_column_max %d last_ep_len_mean %s" % (
    n, e - s, len(fps), its[-1] if its else "-", sorted(fps)[len(fps) // 2], min(fps), max(fps), ep[-1] if ep else "-"))
PY
}
{
    echo "# PPO_driver (C++ facade, console table every update) at BASELINE configs[1]: 4096 envs x 128 steps, 4 minibatches x 10 epochs"
    run 100; run 1100; run 100; run 1100
} | tee $OUT.raw
python3 - $OUT.raw > $OUT <<'PY'
import sys, re
rows = [l for l in open(sys.argv[1]) if l.startswith("updates")]
print(open(sys.argv[1]).read().rstrip())
w = {}
for l in rows:
    m = re.match(r"updates (\d+) wall ([\d.]+)", l)
    w.setdefault(int(m.group(1)), []).append(float(m.group(2)))
per = (min(w[1100]) - min(w[100])) / 1000.0
print("wall per update (1100-update run minus 100-update run, / 1000): %.4f ms  ->  %.1f M env-steps/s through the facade" % (per * 1e3, 4096 * 128 / per / 1e6))
PY
cat $OUT
