"""The hash-made batch of `oracle/ref_harness.cpp config4` (namespace c4 + goldConfig4), regenerated in numpy: BASELINE.json configs[4]'s shape
(obs 376, heads [3, 3, 3, 2]) has no environment in the reference, so every input of the fixture is a counter hash both sides can make --
observations, masks (>= 1 valid action per head), actions (valid under the mask), rewards, done flags, the permutations.  The fixtures carry
CRC-32s of these arrays as the compiled reference saw them; `check_crcs` holds this file to them.  Test infrastructure, not product code.
"""
import zlib

import numpy as np

HEADS = (3, 3, 3, 2)
O, A, H = 376, 11, 4
SEED_OBS, SEED_MASK, SEED_KEEP, SEED_START, SEED_REW, SEED_DONE, SEED_NOBS, SEED_NDONE, SEED_PERM = (
    np.uint64(s << 32) for s in (0x5101, 0x5202, 0x5303, 0x5404, 0x5505, 0x5606, 0x5707, 0x5808, 0x5909))


def mix64(x):
    """splitmix64 finaliser on a uint64 array (wraps modulo 2^64), as oracle/ref_harness.cpp hl::mix64."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def unit24(h):
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(5.9604644775390625e-8)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def _obs(seed, count):
    out = np.empty(count, np.float32)
    for lo in range(0, count, 1 << 22):      # 4 M elements at a time: the uint64 temporaries stay at 32 MB
        hi = min(count, lo + (1 << 22))
        u = unit24(mix64(seed + np.arange(lo, hi, dtype=np.uint64)))
        out[lo:hi] = (u * np.float32(2.0) - np.float32(1.0)) * np.float32(1.7320508)
    return out


def make_batch(T, N):
    """-> dict(obs [B, O] f32, masks [B, A] u8, actions [B, H] i64, rewards [B] f32, dones [B] f32, next_obs [N, O] f32, next_done [N] i32); flat index i = t * N + n."""
    B = T * N
    i = np.arange(B, dtype=np.uint64)
    masks = np.zeros((B, A), np.uint8)
    actions = np.zeros((B, H), np.int64)
    off = 0
    for h, w in enumerate(HEADS):
        for a in range(w):
            masks[:, off + a] = (mix64(SEED_MASK + i * np.uint64(A) + np.uint64(off + a)) % np.uint64(100)) >= np.uint64(35)
        keep = (mix64(SEED_KEEP + i * np.uint64(H) + np.uint64(h)) % np.uint64(w)).astype(np.int64)
        masks[np.arange(B), off + keep] = 1
        a = (mix64(SEED_START + i * np.uint64(H) + np.uint64(h)) % np.uint64(w)).astype(np.int64)
        for _ in range(w):                   # the first valid action from the hashed start, cyclically
            ok = masks[np.arange(B), off + a] != 0
            a = np.where(ok, a, (a + 1) % w)
        actions[:, h] = a
        off += w
    return dict(obs=_obs(SEED_OBS, B * O).reshape(B, O), masks=masks, actions=actions,
                rewards=unit24(mix64(SEED_REW + i)) * np.float32(2.0) - np.float32(1.0),
                dones=((mix64(SEED_DONE + i) % np.uint64(100)) == 0).astype(np.float32),
                next_obs=_obs(SEED_NOBS, N * O).reshape(N, O),
                next_done=((mix64(SEED_NDONE + np.arange(N, dtype=np.uint64)) % np.uint64(100)) == 0).astype(np.int32))


def permutation(epoch, B):
    keys = mix64(SEED_PERM + np.uint64(epoch * B) + np.arange(B, dtype=np.uint64))
    return np.argsort(keys, kind="stable").astype(np.int32)      # = std::sort of (key, index) pairs


def check_crcs(g, batch):
    assert crc(batch["obs"]) == int(g["crc_obs"][0])
    assert crc(batch["masks"]) == int(g["crc_masks"][0])
    assert crc(batch["actions"].astype(np.int32)) == int(g["crc_actions"][0])
    assert crc(batch["rewards"]) == int(g["crc_rewards_dones"][0]) and crc(batch["dones"]) == int(g["crc_rewards_dones"][1])


def load_meta(g):
    m, h = g["meta"], g["hparams"]
    return dict(T=int(m[0]), N=int(m[1]), obs=int(m[2]), act=int(m[3]), nmb=int(m[4]), epochs=int(m[5]), max_steps=int(m[6]), seed=int(m[7]),
                anneal=int(m[9]), use_gae=int(m[10]), norm_adv=int(m[11]), clip_vloss=int(m[12]), masked=int(m[13]), hidden=int(m[14]), n_hidden=int(m[15]),
                lr=float(h[0]), gamma=float(h[1]), lam=float(h[2]), clip=float(h[3]), ent=float(h[4]), vf=float(h[5]), mgn=float(h[6]))
