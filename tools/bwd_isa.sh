#!/bin/bash
# Device assembly of kernels_generic_bwd.hip and what matters in it per instantiation: registers, spills, scratch, the waits on the vector-memory counter.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-slp-vectorize -Iinclude -Ippo-libtorch_amd/csrc $1 --cuda-device-only -S ppo-libtorch_amd/csrc/kernels_generic_bwd.hip -o /tmp/bwd.s 2>&1 | grep -v warning | tail -3
python3 - <<'PY'
import re, collections
s=open('/tmp/bwd.s').read()
for name in ('ILi8ELb1ELi1E','ILi8ELb0ELi2E','ILi4ELb1ELi1E','ILi1ELb1ELi1E'):
    i=s.index('_ZN12_GLOBAL__N_116bwd_layer_kernel'+name+'EEvNS_7BwdPairE:')
    lines=[l.strip() for l in s[i:s.index('.Lfunc_end', i)].splitlines()[1:] if l.strip() and not l.strip().startswith(';')]
    c=collections.Counter(l.split()[0] for l in lines)
    meta=s[s.index('.name:           _ZN12_GLOBAL__N_116bwd_layer_kernel'+name):][:700]
    print(name, {k:c.get(k,0) for k in ('v_mfma_f32_32x32x16_bf16','ds_read_b128','ds_read_b64_tr_b16','global_load_lds_dwordx4','s_barrier','scratch_load_dword','scratch_load_dwordx2')}, re.findall(r'\.(vgpr_count|vgpr_spill_count):\s+(\d+)', meta), [l for l in lines if l.startswith('s_waitcnt') and 'vmcnt' in l])
PY
