// ppo-libtorch_amd/csrc/kernels_generic_fused.hip -- the whole forward pass of a bf16-storage network (generic.hpp; BASELINE configs[4]: obs 376,
// 4 x 256, heads [3,3,3,2]) in ONE launch: a workgroup carries a tile of rows through every Linear + tanh layer of one net
// (Agent.cpp:25-59 generalised) with the activations never leaving LDS.
//
//   rows x K tile of the layer input in LDS as bf16 (A operand, 16-byte fragment reads) ; the layer's weights come straight from L2 as B
//   fragments, out of a copy kept in fragment order (a wave's load for one k step is 1 KiB of consecutive bytes; read row-major, the same
//   load touched 32 different memory lines and the kernel was bound by L1 line lookups: 53 us per rollout step) -- all of a column
//   block's loads are issued before its first MFMA ; eight waves, wave w owns output columns 32 (w + 8 j) ; v_mfma_f32_32x32x16_bf16, f32
//   accumulation.  The product is issued TRANSPOSED (weights as the A operand, rows as the B operand): the result then has lane = row
//   and registers = 4-groups of consecutive output units, so the epilogue (bias + tanh -> bf16) leaves as 8-byte stores into the OTHER LDS
//   tile, which is the next layer's operand (lane = column cost sixteen 2-byte stores per block and ~25 vector instructions per element of
//   address arithmetic: 42 vector instructions per MFMA, measured).  One barrier per layer, no HBM traffic between layers, no per-layer launch.
//
// Two kernels on that core:
//   generic_rollout_kernel   the T-step rollout of PPO_MultiDiscrete::train() (PPO_MultiDiscrete.cpp:547-575) for the synthetic env: per step
//                            { observation (counter-based env) -> actor -> per-head (masked) categorical, sample, log-prob -> stores -> env
//                            transition }, 32 envs per workgroup, the step loop INSIDE the kernel (envs never interact in a rollout).  Replaces
//                            128 x { 5 layer products, heads, stores, 2 env kernels } = 1150 launches by one.
//   generic_forward_kernel   a net over a batch of rows (critic over the rollout's T N + N observations; both nets of a minibatch step, which
//                            also keep every hidden activation in HBM for the backward pass), 64 rows per workgroup.
// Arithmetic = launch_matmul_bf16's (operands bf16 round-to-nearest-even, f32 accumulation, bias and tanh in f32); the order of the k sum
// differs from the tiled kernel's only in chunking.
#include "generic.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FU_WAVES = 8, FU_THREADS = 64 * FU_WAVES;
constexpr int FU_KSTEPS = 16;   // k steps (of 16) whose weight fragments are in flight at once: 64 registers (a 256-wide layer in one pass)

__device__ __forceinline__ uint32_t fu_f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float fu_u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint16_t fu_bf16(float x) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ uint32_t fu_pack(float x0, float x1) { const bf16x2 v = { (__bf16)x0, (__bf16)x1 }; return __builtin_bit_cast(uint32_t, v); }

#ifdef FU_DBG_STAMPS
// diagnostic build only: cycle stamps of wave 0 of one workgroup (tools/fused_fwd_probe.py); `dbg` is a local of the enclosing function
#define FU_STAMP(i) do { if (dbg && threadIdx.x == 0) { dbg[i] = (unsigned int)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FU_STAMP(i) do { } while (0)
#endif
struct FusedNet {
    GenLayout L;
    int net;
    const float* params;            // f32 master parameters (biases are read from here)
    const uint16_t* wfrags;         // bf16 weights of every layer in MFMA-fragment order (gen_weight_planes: [column block][k step][lane][8])
    int64_t wp_off[GEN_MAX_LAYERS]; // of THIS net
    int wp_kpad[GEN_MAX_LAYERS];
    int ldA;                        // LDS row pitch (bf16 elements): the widest padded contraction length (a multiple of 256) + 8
};

// Runs every layer of the net on the FM x 32 rows whose layer-0 input sits in tile0 ([rows][ldA] bf16, zero beyond obs up to a multiple of
// 16).  Hidden activations alternate between tile1 and tile0; the head's f32 outputs land in s_out [rows][32] (columns beyond the head's width
// are zero).  keep[l] != nullptr: hidden layer l's activation tile is also stored to keep[l] + row0 * ld_keep (rows < n_rows only).
// Ends with a barrier: s_out and the last tile are visible to every thread.
// The weight fragments of a column block: 16 bytes per lane and k step.  A layer's first FU_KSTEPS k steps travel while the layer above is
// still in its epilogue (`pre`); what a wider layer has beyond them comes in passes of FU_KTAIL steps into the registers of the first
// FU_KTAIL steps, as soon as those have been multiplied (no second register set: the kernels sit at the 256-register line).
constexpr int FU_KTAIL = 8;
struct BFrags { u32x4 q[FU_KSTEPS]; };
__device__ __forceinline__ void load_bfrags(BFrags& b, const FusedNet& f, int l, int cb, int k0, int lane) {
    const uint16_t* wblk = f.wfrags + f.wp_off[l] + ((int64_t)cb * (f.wp_kpad[l] / 16) * 64 + lane) * 8;   // a k step's fragments: 64 lanes x 16 consecutive bytes
#pragma unroll
    for (int j = 0; j < FU_KSTEPS; j++)   // a layer narrower than FU_KSTEPS k steps reads on into the next column block: in bounds, not used
        b.q[j] = *reinterpret_cast<const u32x4*>(wblk + (int64_t)(k0 + j) * 512);
}
__device__ __forceinline__ void load_btail(BFrags& b, const FusedNet& f, int l, int cb, int k0, int lane) {
    const uint16_t* wblk = f.wfrags + f.wp_off[l] + ((int64_t)cb * (f.wp_kpad[l] / 16) * 64 + lane) * 8;
#pragma unroll
    for (int j = 0; j < FU_KTAIL; j++) b.q[j] = *reinterpret_cast<const u32x4*>(wblk + (int64_t)(k0 + j) * 512);
}

// XB: k steps whose row fragments are read from LDS together, in front of their products (4; 2 where registers are short: the f32-input kernel holds 48 of them for the next tile)
template <int FM, int NW = FU_WAVES, int XB = 4>
__device__ __forceinline__ void fused_layers(const FusedNet& f, int64_t next_off0, uint16_t* tile0, uint16_t* tile1, float* s_out, uint16_t* const* keep, int64_t ld_keep,
                                             int64_t row0, int n_rows, BFrags& pre, unsigned int* dbg = nullptr) {
    // `pre` holds, on entry, the fragments of (layer 0, column block = wave, k steps 0 ..) -- requested by the caller, e.g. while the input tile
    // was still being written -- and on exit those of layer 0 of the net of the NEXT call, whose offset into the fragment array is next_off0
    // (the same net's, or the other net's when a workgroup runs both nets on every tile; a scalar, not a reference chosen at run time: that sends the argument struct to scratch): a layer's first fragments are always requested
    // before the epilogue and the barrier of the layer above, so their trip to L2 is never waited for with nothing else to do.
    const GenLayout& L = f.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kg = lane >> 5;
    uint16_t* src = tile0;
    uint16_t* dst = tile1;
    for (int l = 0; l < L.n_layers; l++) {
        const int N = L.out_dim[f.net][l];
        const bool last = l == L.n_layers - 1;
        const int ksteps = f.wp_kpad[l] / 16;   // a multiple of 8; the weights of k >= K are zero, the tile's columns there finite
        const int nblk = (N + 31) / 32;
        const float* bias = f.params + L.b_off[f.net][l];
        const int ln = last ? 0 : l + 1;   // the layer whose first fragments are requested during this one
        for (int cb = wave; cb < nblk || cb == wave; cb += NW) {   // every wave passes once (it may own no column block): it still prefetches
            const bool own = cb < nblk;
            f32x16 acc[FM];
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
            // bias of the lane's 16 output units (4 groups of 4 consecutive ones): requested before the products, used after them
            float bv[16];
            {
                const bool vec = (N & 3) == 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = 32 * cb + 8 * q + 4 * kg;
                    if (vec) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias + (own && n < N ? n : 0));
                        bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) bv[4 * q + e] = bias[own && n + e < N ? n + e : 0];
                    }
                }
            }
            if (own) {
                if (cb != wave) load_bfrags(pre, f, l, cb, 0, lane);   // a second column block of this wave (layers wider than 256): not prefetched
                // the rows' fragments come from LDS four k steps at a time, all requested before the first product of the batch (read one by
                // one in front of its MFMA, each product waits out an LDS round trip)
                const uint16_t* xr[FM];
#pragma unroll
                for (int i = 0; i < FM; i++) xr[i] = src + (32 * i + li) * f.ldA + 8 * kg;
                auto batch = [&](int kx0, int jq0) {   // k steps kx0 .. kx0 + 3 of the tile against fragments q[jq0 .. jq0 + 3], XB at a time
#pragma unroll
                    for (int h = 0; h < 4 / XB; h++) {
                        const int kx = kx0 + XB * h, jq = jq0 + XB * h;
                        u32x4 x[FM][XB];
#pragma unroll
                        for (int j = 0; j < XB; j++)
#pragma unroll
                            for (int i = 0; i < FM; i++) x[i][j] = *reinterpret_cast<const u32x4*>(xr[i] + 16 * (kx + j));
                        // ... and a scheduling barrier keeps them there: without it hipcc sinks every read to its product and reuses ONE register quad -- ds_read, s_waitcnt
                        // lgkmcnt(0), v_mfma, 32 times per layer and tile, an LDS round trip in front of every product (found in the assembly in round 6; a memory
                        // fence is not enough: the products have no memory semantics and climb up between the reads)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < XB; j++)
#pragma unroll
                            for (int i = 0; i < FM; i++)   // D[n][row]: weights are the A operand, the rows the B operand
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pre.q[jq + j]), __builtin_bit_cast(bf16x8, x[i][j]), acc[i], 0, 0, 0);
                    }
                };
                FU_STAMP(32 + 4 * l);
                batch(0, 0); batch(4, 4);
                if (ksteps > FU_KSTEPS) load_btail(pre, f, l, cb, FU_KSTEPS, lane);   // ksteps is a multiple of 8
                if (ksteps > FU_KTAIL) { batch(8, 8); batch(12, 12); }
                for (int k0 = FU_KSTEPS; k0 < ksteps; k0 += FU_KTAIL) {
                    batch(k0, 0); batch(k0 + 4, 4);
                    if (k0 + FU_KTAIL < ksteps) load_btail(pre, f, l, cb, k0 + FU_KTAIL, lane);
                }
            }
            FU_STAMP(33 + 4 * l);
            // the wave's last pass of this layer: the next layer's first fragments are requested now -- by a wave with an epilogue to do in four
            // pieces spread over it (sixteen loads in a row from all eight waves fill the CU's load queue: ~1 200 cycles of issue stall each)
            const bool pf_next = cb + NW >= nblk;
            const uint16_t* nblkp = f.wfrags + (last ? next_off0 : f.wp_off[ln]) + ((int64_t)wave * (f.wp_kpad[ln] / 16) * 64 + lane) * 8;   // (layer widths are the same in both nets)
            auto next_piece = [&](int pc) {
#pragma unroll
                for (int j = 0; j < 4; j++) pre.q[4 * pc + j] = *reinterpret_cast<const u32x4*>(nblkp + (int64_t)(4 * pc + j) * 512);
            };
            if (pf_next && !own) { next_piece(0); next_piece(1); next_piece(2); next_piece(3); }
            FU_STAMP(34 + 4 * l);   // the wave's last pass of this layer: next layer's first fragments
            if (own) {
                // epilogue: lane's row = 32 i + li, register r <-> unit 32 cb + (r & 3) + 8 (r >> 2) + 4 kg
                const bool full = !last && (N & 31) == 0;   // every unit of the block exists: no guards (wave-uniform)
#pragma unroll
                for (int i = 0; i < FM; i++) {
                    const int row = 32 * i + li;
                    uint16_t* drow = dst + row * f.ldA + 32 * cb + 4 * kg;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (pf_next && ((FM == 1) || (q & 1))) next_piece(FM == 1 ? q : 2 * i + (q >> 1));
                        const int n = 32 * cb + 8 * q + 4 * kg;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = acc[i][4 * q + e] + bv[4 * q + e];
                        if (full) {
                            *reinterpret_cast<uint2*>(drow + 8 * q) = make_uint2(fu_pack(tanh_mufu(v[0]), tanh_mufu(v[1])), fu_pack(tanh_mufu(v[2]), tanh_mufu(v[3])));
                        } else if (last) {   // heads are at most 32 logits wide (PPO_MAX_ACT): one column block
                            *reinterpret_cast<float4*>(s_out + row * 32 + 8 * q + 4 * kg) =
                                make_float4(n < N ? v[0] : 0.0f, n + 1 < N ? v[1] : 0.0f, n + 2 < N ? v[2] : 0.0f, n + 3 < N ? v[3] : 0.0f);
                        } else {
                            *reinterpret_cast<uint2*>(drow + 8 * q) =
                                make_uint2(fu_pack(n < N ? tanh_mufu(v[0]) : 0.0f, n + 1 < N ? tanh_mufu(v[1]) : 0.0f),
                                           fu_pack(n + 2 < N ? tanh_mufu(v[2]) : 0.0f, n + 3 < N ? tanh_mufu(v[3]) : 0.0f));
                        }
                    }
                }
            }
        }
        FU_STAMP(8 + 2 * l);
        __syncthreads();
        FU_STAMP(9 + 2 * l);
        if (!last && keep && keep[l]) {   // the activation tile leaves for the backward pass in 16-byte row pieces
            // four pieces per thread with their LDS reads in flight together (one after the other -- read, wait, store -- the copy of a 256-wide
            // tile held every wave ~1 500 cycles in front of the next layer: FU_DBG_STAMPS); the piece -> (row, column) split without a division when
            // the row holds a power of two of pieces
            const int p8 = (N + 7) / 8, total = 32 * FM * p8;
            const bool pow2 = (p8 & (p8 - 1)) == 0;
            const int sh = __builtin_ctz((unsigned)p8);
            for (int e0 = tid; e0 < total; e0 += 4 * 64 * NW) {
                u32x4 v[4];
                int row[4], c8[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = e0 + j * 64 * NW, ee = e < total ? e : 0;
                    row[j] = pow2 ? ee >> sh : ee / p8; c8[j] = pow2 ? ee & (p8 - 1) : ee % p8;
                    v[j] = *reinterpret_cast<const u32x4*>(dst + row[j] * f.ldA + 8 * c8[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (e0 + j * 64 * NW < total && row[j] < n_rows) {
                        // streamed past the L2 (nt): 268 MB per step that this kernel never reads again (forward launch 172 -> 168 us, A/B in one call)
                        __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(keep[l] + (row0 + row[j]) * ld_keep + 8 * c8[j]));
                    }
            }
        }
        uint16_t* t = src; src = dst; dst = t;
    }
}

// observation j .. j + 3 of (env, step): four values from one Philox word each (kernels_generic.hip: syn_obs, bit for bit)
__device__ __forceinline__ float4 syn_obs4(int64_t seed, int64_t env, int64_t step, int j4) {
    const uint4 w = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env, (uint32_t)step, (uint32_t)j4, 0x10u);
    auto one = [](uint32_t x) {
        const int sum = (int)(x & 255u) + (int)((x >> 8) & 255u) + (int)((x >> 16) & 255u) + (int)(x >> 24);
        return (float)(sum - 510) * 0.0067658990621566772f;
    };
    return make_float4(one(w.x), one(w.y), one(w.z), one(w.w));
}

struct FusedRolloutArgs {
    FusedNet f;
    int dist_kind;
    int N, T, max_episode_steps;
    int64_t seed, env_offset, step_base;
    int32_t* ep_len; float* ep_rew;
    float* obs; uint8_t* masks; int32_t* actions; float* logprobs; float* rewards; float* dones;
    int32_t* fin_len; float* fin_rew;
    float* next_obs; int32_t* next_done; uint8_t* cur_mask;
    const int64_t* forced;          // [T, N, H] or null
};

template <int DIST>
__global__ __launch_bounds__(FU_THREADS, 1) void generic_rollout_kernel(const FusedRolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t fu_lds[];
    const GenLayout& L = a.f.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t* tile0 = fu_lds;
    uint16_t* tile1 = fu_lds + 32 * a.f.ldA;
    float* s_out = reinterpret_cast<float*>(fu_lds + 2 * 32 * a.f.ldA);      // [32][32]
    uint8_t* s_mask = reinterpret_cast<uint8_t*>(s_out + 32 * 32);          // [2][32][PPO_MAX_ACT]: this step's and the next step's masks
    const int e0 = blockIdx.x * 32;
    const int O = L.obs, A = L.act, H = L.n_heads;
    const int o4 = (O + 3) / 4, opad = (O + 15) / 16 * 16;
    BFrags pre;
    load_bfrags(pre, a.f, 0, wave, 0, lane);   // layer 0's first weight fragments travel while the first observation is formed
    // per-env state lives in the registers of lanes 0..31 of wave 0 for the whole rollout
    const bool env_lane = tid < 32 && e0 + tid < a.N;
    const int64_t env_g = a.env_offset + e0 + tid;
    int len = 0, prev_done = 0;
    float rew = 0.0f;
    if (env_lane) { len = a.ep_len[e0 + tid]; rew = a.ep_rew[e0 + tid]; prev_done = a.next_done[e0 + tid]; }

    // Observation (and mask) of global step `step` for the workgroup's envs, formed by threads [t0, t0 + nt): f32 to obs_dst [N, O], bf16 to
    // tile0 (the actor's layer-0 input), the mask to s_mask[slot] and mask_dst [N, A].  The synthetic env is memoryless (kernels_generic.hip),
    // so step t + 1's observation does not wait for step t's action: waves 1..7 form it while wave 0 samples and steps the envs.
    auto observe = [&](int64_t step, float* obs_dst, uint8_t* mask_dst, int slot, int t0, int nt) {
        const int me = tid - t0;
        if (me < 0 || me >= nt) return;
        for (int e = me; e < 32 * o4; e += nt) {
            const int env = e / o4, j4 = e % o4;
            const float4 v = syn_obs4(a.seed, a.env_offset + e0 + env, step, j4);
            const bool ok = e0 + env < a.N;
            uint16_t* trow = tile0 + env * a.f.ldA + 4 * j4;
            float* grow = obs_dst + (size_t)(e0 + env) * O + 4 * j4;
            if (4 * j4 + 3 < O) {
                *reinterpret_cast<uint2*>(trow) = make_uint2(fu_pack(v.x, v.y), fu_pack(v.z, v.w));
                if (ok) {
                    if ((O & 3) == 0) *reinterpret_cast<float4*>(grow) = v;
                    else { grow[0] = v.x; grow[1] = v.y; grow[2] = v.z; grow[3] = v.w; }
                }
            } else {
                const float vv[4] = { v.x, v.y, v.z, v.w };
                for (int q = 0; q < 4; q++) if (4 * j4 + q < O) { trow[q] = fu_bf16(vv[q]); if (ok) grow[q] = vv[q]; }
            }
        }
        {   // zero the k padding and whatever a hidden layer's output left behind the observation (see generic_forward_kernel)
            const int hpad = (L.hidden + 31) / 32 * 32, zhi = opad > hpad ? opad : hpad;
            for (int e = me; e < 32 * (zhi - O); e += nt) tile0[(e / (zhi - O)) * a.f.ldA + O + e % (zhi - O)] = 0;
        }
        if (me < 32) {
            uint8_t* mrow = s_mask + (slot * 32 + me) * PPO_MAX_ACT;
            uint32_t bits = 0xffffffffu;
            if (DIST == PPO_DIST_MASKED)
                bits = philox4x32_10((uint32_t)a.seed, (uint32_t)((uint64_t)a.seed >> 32), (uint32_t)(a.env_offset + e0 + me), (uint32_t)step, 0u, 0x12u).x;
            int off = 0;
            for (int h = 0; h < H; h++) {
                bool any = false;
                for (int k = 0; k < L.head_dims[h]; k++) { const bool v = (bits >> (off + k)) & 1u; any |= v; mrow[off + k] = v ? 1 : 0; }
                if (!any) mrow[off] = 1;
                off += L.head_dims[h];
            }
            if (e0 + me < a.N && mask_dst) for (int k = 0; k < A; k++) mask_dst[(size_t)(e0 + me) * A + k] = mrow[k];
        }
    };

    for (int e = tid; e < 2 * 32 * a.f.ldA / 8; e += FU_THREADS) { const u32x4 z = { 0u, 0u, 0u, 0u }; reinterpret_cast<u32x4*>(fu_lds)[e] = z; }   // finite everywhere
    __syncthreads();
    observe(a.step_base, a.obs, a.masks, 0, 0, FU_THREADS);   // m_obs[0] = next_obs, m_action_masks[0] = next_mask (:553-555)
    __syncthreads();
    for (int t = 0; t < a.T; t++) {
        const int64_t step = a.step_base + t;
        const size_t tn = (size_t)t * a.N;
        fused_layers<1>(a.f, a.f.wp_off[0], tile0, tile1, s_out, nullptr, 0, 0, 32, pre);   // ends with a barrier: logits in s_out, tile0 free again
        if (wave != 0) {
            const bool more = t + 1 < a.T;   // the observation the agent sees next: row t + 1 of the buffers, or NEXT_OBS / the current mask after the last step
            observe(step + 1, more ? a.obs + (tn + a.N) * O : a.next_obs, more ? (a.masks ? a.masks + (tn + a.N) * A : nullptr) : a.cur_mask, (t + 1) & 1, 64, FU_THREADS - 64);
        } else if (env_lane) {
            // per-head (masked) categorical on the row's logits, sample or forced action, summed log-prob (Agent.cpp:137-170; heads_kernel)
            const uint8_t* mrow = s_mask + ((t & 1) * 32 + tid) * PPO_MAX_ACT;
            float lp_sum = 0.0f;
            int off = 0;
            for (int h = 0; h < H; h++) {
                const int Ah = L.head_dims[h];
                float z[PPO_MAX_ACT], p[PPO_MAX_ACT];
                for (int k = 0; k < Ah; k++) z[k] = s_out[tid * 32 + off + k];
                categorical_head_fast<DIST>(z, p, DIST == PPO_DIST_MASKED ? mrow + off : nullptr, Ah);
                int act;
                if (a.forced) {
                    act = (int)a.forced[(tn + e0 + tid) * H + h];
                } else {
                    const uint4 w = philox4x32_10((uint32_t)a.seed, (uint32_t)((uint64_t)a.seed >> 32), (uint32_t)env_g, (uint32_t)(step >> 2), (uint32_t)h, 0u);
                    const uint32_t ws = (step & 3) == 0 ? w.x : ((step & 3) == 1 ? w.y : ((step & 3) == 2 ? w.z : w.w));
                    act = sample_head(p, Ah, (float)(ws >> 8) * 0x1p-24f);
                }
                a.actions[(tn + e0 + tid) * H + h] = act;
                float lp = 0.0f;
                for (int k = 0; k < Ah; k++) if (k == act) lp = z[k];
                lp_sum = h == 0 ? lp : lp_sum + lp;
                off += Ah;
            }
            a.logprobs[tn + e0 + tid] = lp_sum;
            a.dones[tn + e0 + tid] = (float)prev_done;                       // m_dones[step] = next_done (:554)
            // env transition of this step (kernels_generic.hip: synthetic_transition_kernel, bit for bit)
            const uint4 w = philox4x32_10((uint32_t)a.seed, (uint32_t)((uint64_t)a.seed >> 32), (uint32_t)env_g, (uint32_t)step, 0u, 0x11u);
            const float r = (float)(w.x >> 8) * 0x1p-23f - 1.0f;
            int term = (w.y >> 8) < 167772u ? 1 : 0;
            len += 1;
            rew += r;
            if (len == a.max_episode_steps) term = 1;
            int fl = 0;
            float fr = 0.0f;
            if (term) { fl = len; fr = rew; len = 0; rew = 0.0f; }
            a.rewards[tn + e0 + tid] = r;
            a.fin_len[tn + e0 + tid] = fl;
            a.fin_rew[tn + e0 + tid] = fr;
            prev_done = term;
        }
        __syncthreads();   // tile0 holds the next observation; s_out and the older mask slot are free
    }
    if (env_lane) { a.ep_len[e0 + tid] = len; a.ep_rew[e0 + tid] = rew; a.next_done[e0 + tid] = prev_done; }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Heads + PPO loss + the HEAD LAYER'S BACKWARD inside the forward kernel (round 6; PPO_MultiDiscrete.cpp:593-668 on a 64-row tile).  When a pass of a tile
// ends, the tile's logits (or values) sit in s_out and the top hidden activation h sits in an LDS tile -- everything the loss and the head layer's backward
// need.  Instead of writing logits and h to HBM, reading them back in loss_lanes_kernel (heads, masked categorical, loss, d logits) and again in
// bwd_layer_kernel<1, true, 1> (dW_head, dZ_top) -- 12.5 + 35.8 us and 168 MB per 65 536-row step -- the workgroup does it on the spot:
//   E1  threads 0 .. 255, four lanes per row (lane = head, loss_lanes_kernel's arithmetic): d logits / d value as bf16 into two small LDS images, [row][n]
//       and [n][row]; the loss sums, the head's bias gradient (column sums of the unrounded gradient) reduced per wave and added to per-wave LDS accumulators;
//   E3  dZ_top[row][k] = (dL[row][:] . W_head[:][k]) (1 - h[row][k]^2): ONE k step (logits padded to 16) per 32 x 32 block, wave w the columns 32 w ..;
//       bf16 into the free LDS tile, f32 column sums (the top hidden layer's bias gradient) into an LDS accumulator;
//   E4  dW_head[n][k] += sum_rows dL[row][n] h[row][k]: four k steps (rows), h by transposing reads, accumulated in LDS by the wave that owns the columns;
//   E5  the dZ_top tile leaves in 16-byte row pieces for the next backward launch.
// Per WORKGROUP (not per row range): one partial dW_head, one column-sum row, one row of loss sums and of head bias sums -- slab_sum_layers_kernel and
// gen_opt_fused_kernel add them in workgroup order as they add the other layers' partials.  Every accumulator has ONE writer and a fixed order: same inputs,
// same bits.  The per-row records (32 bytes through the step's index list) are requested when the tile's first pass starts and are used two passes later.
// Arithmetic is the unfused path's: the same bf16 roundings of d logits, h and W_head, f32 accumulation, tanh' and every sum in f32.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct FusedLossArgs {
    LossParams hp; float invM; double global_M;
    const AdvStat* adv_stat;              // null: advantages are not normalised
    const int32_t* idx; const float4* rec;
    uint16_t* dz_top[2]; int64_t ld_dz;   // [net] d(pre-activation) of the top hidden layer, [rows + pad][ld_dz] bf16
    const uint16_t* whead[2]; int64_t ldw;// [net] bf16 plane of the head layer's weights [n_pad][ldw]
    float* head_slab[2]; int64_t head_slab_stride;   // [net] per-workgroup partial dW_head [n_real][hidden]
    float* cs_top[2]; int64_t ld_cs;      // [net] per-workgroup column sums of dZ_top
    float* head_db_part;                  // [GEN_LOSS_BLOCKS][act + 1]
    double* loss_part;                    // [GEN_LOSS_BLOCKS][8]
};
#ifndef FL_ABL
#define FL_ABL 0   /* timing-only ablation builds (wrong results): 1 no E1, 2 no E3, 4 no E4, 8 no E5, 16 no record prefetch */
#endif
constexpr int FL_NP = 16;                 // logits as staged: one k step of the matrix cores (the slot-wise loss serves <= 4 heads of <= 4 logits)
constexpr int FL_DLT_PITCH = 72;          // [n][row] image: 64 rows + 8
constexpr int FL_DW_ROWS = FL_NP + 4;     // LDS rows of the dW_head accumulator: the actor's 16, the critic's (1, padded to 4)
constexpr size_t fused_loss_lds_bytes(int ld_h) {
    return (size_t)64 * FL_NP * 2 + (size_t)32 * FL_DLT_PITCH * 2 + (size_t)2 * FU_WAVES * 64 * 16 + (size_t)FL_DW_ROWS * ld_h * 4 + (size_t)4 * ld_h * 4 +
           16 * 8 * sizeof(double) + 16 * 20 * sizeof(float) + 64 + (size_t)64 * 32;
}
__device__ __forceinline__ uint2 fu_read_tr16(const uint16_t* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}

struct FusedForwardArgs {
    FusedNet f;
    const float* x_f32;             // [rows][obs] f32 (rounded to bf16 on the way into LDS), or
    const uint16_t* x_bf;           // [rows][ld_x] bf16, zero padded
    const int32_t* idx;             // x_bf only, may be null: row r of the pass is row idx[r] of x_bf (the minibatch's rows read in place, PPO_Discrete.cpp:576-582)
    int64_t ld_x, rows;
    float* out;                     // [rows][out_dim(last)]
    uint16_t* keep[GEN_MAX_LAYERS]; // hidden activations to keep (null: none), pitch ld_keep
    int64_t ld_keep;
};

// Persistent: one workgroup per CU walks the 64-row tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The NEXT tile's input travels from
// HBM into registers (16-byte pieces, all requested together) while the current tile goes through its layers, so the only memory round
// trip a workgroup ever waits for is its first.  (Staged by a plain loop -- two 4-byte loads, convert, store, per iteration -- the input
// cost a tile 24 dependent round trips: 196 us for the critic over 65 536 rows, half of all wave cycles parked at s_waitcnt.)
// 16-byte input pieces per thread and tile (eight threads per row): bf16 rows of up to 384 elements, f32 rows of up to 384
template <bool BF> constexpr int fu_np() { return BF ? 6 : 12; }
typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
// Two passes in one launch (the two nets over the same rows): workgroups [0, split) run a[0], the rest a[1] -- the second pass's workgroups take the CUs
// the first pass's leave, with no kernel boundary and no second stream between them.  split == gridDim.x: one pass.
// dual: every workgroup runs BOTH passes on each of its tiles -- the input tile is fetched and staged from the same registers twice, read from memory once.
struct FusedForwardPair { FusedForwardArgs a[2]; int split; int dual; FusedLossArgs loss; };
// LOSS: 0 = forward only; PPO_DIST_CATEGORICAL + 1 / PPO_DIST_MASKED + 1 = the loss and the head layer's backward in the epilogue of every pass (dual launches only)
// NW / FMT: waves per workgroup and 32-row blocks per tile.  8 / 2 = one workgroup of 64-row tiles per CU; 4 / 1 (LOSS == 0 only, exploration: -DFU_HALF) = TWO
// workgroups of 32-row tiles per CU, whose phases (weights from L2, LDS + matrix cores, tanh epilogue) can overlap each other's instead of running in lockstep.
template <bool BF, int LOSS = 0, int NW = FU_WAVES, int FMT = 2>
__global__ __launch_bounds__(64 * NW, NW == FU_WAVES ? 1 : 2) void generic_forward_kernel(const FusedForwardPair pp) {
    static_assert(LOSS == 0 || (NW == FU_WAVES && FMT == 2), "the loss epilogue is written for eight waves and 64-row tiles");
    constexpr int FU_THREADS = 64 * NW;   // (shadows the file's constant inside this kernel)
    const int second = (int)blockIdx.x >= pp.split ? 1 : 0;
    const FusedForwardArgs& a = pp.a[second];
    const int bid = (int)blockIdx.x - (second ? pp.split : 0), nblk = second ? (int)gridDim.x - pp.split : pp.split;
    constexpr int FU_NP = fu_np<BF>();
    extern __shared__ __attribute__((aligned(16))) uint16_t fu_lds[];
    constexpr int FM = FMT, RB = 32 * FM;
    const GenLayout& L = a.f.L;
    const int tid = threadIdx.x;
    uint16_t* tile0 = fu_lds;
    uint16_t* tile1 = fu_lds + RB * a.f.ldA;
    float* s_out = reinterpret_cast<float*>(fu_lds + 2 * RB * a.f.ldA);      // [RB][32]
    const int O = L.obs, opad = (O + 15) / 16 * 16;
    const int64_t n_tiles = (a.rows + RB - 1) / RB;
    BFrags pre;
    load_bfrags(pre, a.f, 0, tid >> 6, 0, tid & 63);   // layer 0's first weight fragments travel while the input tile is loaded
    // ---- LOSS: the LDS behind s_out (and the rollout kernel's mask area): images of d logits, the head weights' fragments, the workgroup's accumulators ----
    const FusedLossArgs& fl = pp.loss;
    const int ld_h = (int)fl.ld_dz;
    uint16_t* const s_dl = reinterpret_cast<uint16_t*>(s_out + RB * 32) + 32 * PPO_MAX_ACT;   // [64][FL_NP]   d logits (d value in column 0), row-major
    uint16_t* const s_dlt = s_dl + 64 * FL_NP;                                                  // [32][FL_DLT_PITCH] the same, transposed; rows 16 .. 31 stay zero
    u32x4* const s_wh = reinterpret_cast<u32x4*>(s_dlt + 32 * FL_DLT_PITCH);                    // [2 nets][8 waves][64 lanes] A fragments of W_head^T
    float* const s_dw = reinterpret_cast<float*>(s_wh + 2 * FU_WAVES * 64);                     // [FL_DW_ROWS][ld_h]: actor rows 0 .. 15, critic row 16
    // Sums over rows are formed on the DPP network inside each 16-lane row of a wave and added to an accumulator of that (wave, row of 16 lanes) alone: one writer
    // per word, no cross-row shuffle (80 ds_bpermute per pass in the first version: 29 us of the launch), and the flush adds the partials in a fixed order.
    float* const s_cs = s_dw + FL_DW_ROWS * ld_h;                                               // [2 nets][2 row halves][ld_h] column sums of dZ_top
    double* const s_ls = reinterpret_cast<double*>(s_cs + 4 * ld_h);                            // [4 waves x 4 rows of 16 lanes][8] loss sums
    float* const s_db = reinterpret_cast<float*>(s_ls + 16 * 8);                                // [16][20] head bias sums: [0 .. 15] d logits, [16] d value
    float* const s_misc = s_db + 16 * 20;                                                       // advantage mean, 1 / (std + eps)
    float4* const s_rec = reinterpret_cast<float4*>(s_misc + 16);                               // [64 rows][2] the tile's row records
    if constexpr (LOSS != 0) {
        const int lane = tid & 63, wave = tid >> 6, li = lane & 31, kg = lane >> 5;
        for (int e = tid; e < (int)((fused_loss_lds_bytes(ld_h) - 64 - 64 * 32) / 4); e += FU_THREADS) reinterpret_cast<uint32_t*>(s_dl)[e] = 0u;   // accumulators, the padding of the images
        if (tid == 0) {
            float mean_f = 0.0f, std_f = 0.0f;
            if (fl.adv_stat) {
                double t1 = 0.0, t2 = 0.0;
                for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += fl.adv_stat[i].s1; t2 += fl.adv_stat[i].s2; }
                const double mean = t1 / fl.global_M;
                const double var = (t2 - t1 * mean) / (fl.global_M - 1.0);
                mean_f = (float)mean;
                std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
            }
            s_misc[0] = mean_f; s_misc[1] = 1.0f / (std_f + 1e-8f);
        }
        __syncthreads();   // (the zeros above are in place before the fragments below are written over them)
        // W_head^T as the A operand of D[k][row] = sum_n W_head[n][k] dL[row][n]: lane (i = column 32 wave + i, kg) holds n = 8 kg .. 8 kg + 7
#pragma unroll
        for (int net = 0; net < 2; net++) {
            if (32 * wave >= ld_h) break;
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t lo = fl.whead[net][(int64_t)(8 * kg + 2 * e) * fl.ldw + 32 * wave + li], hi = fl.whead[net][(int64_t)(8 * kg + 2 * e + 1) * fl.ldw + 32 * wave + li];
                w[e] = lo | (hi << 16);
            }
            const u32x4 v = { w[0], w[1], w[2], w[3] };
            s_wh[(net * FU_WAVES + wave) * 64 + lane] = v;
        }
    }
    // The row records (32 bytes through the index list) ride with the input rows: the thread that fetches pieces sub = 0 / 1 of a row also fetches the two halves
    // of the row's record, a tile ahead, and parks them in s_rec when it stages the tile -- no load whose address or value an epilogue has to wait for.
    u32x4 prec = { 0u, 0u, 0u, 0u };
    // eight threads per row: thread (row = tid >> 3, sub = tid & 7) owns the row's 16-byte pieces sub, sub + 8, ... (8 bf16 or 4 floats each)
    constexpr bool bf = BF;
    const int ppr = bf ? opad / 8 : (O + 3) / 4;   // pieces per row (f32: O is a multiple of 4, host-checked)
    const int prow = tid >> 3, sub = tid & 7;
    u32x4 pf[FU_NP];
    // gathered input: the source row of this thread's row of a tile is requested one tile before the row itself (a whole tile's layers lie between
    // the index load and its use; tile `first`'s is the one round trip in front of the first fetch)
    int64_t src_next = 0;
    auto fetch_index = [&](int64_t tile) {
        const int64_t row = tile * RB + prow;
        src_next = (bf && a.idx && tile < n_tiles && row < a.rows) ? (int64_t)a.idx[row] : row;
    };
    auto fetch = [&](int64_t tile) {   // unconditional loads: a piece that does not exist re-reads the operand's first bytes and is zeroed at use
        const bool rok = tile < n_tiles && tile * RB + prow < a.rows;
        const int64_t row = src_next;
#pragma unroll
        for (int p = 0; p < FU_NP; p++) {
            const int piece = sub + 8 * p;
            const bool ok = rok && piece < ppr;
            if (bf) pf[p] = *reinterpret_cast<const u32x4*>(a.x_bf + (ok ? row * a.ld_x + 8 * piece : 0));
            else pf[p] = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4a4*>(a.x_f32 + (ok ? row * O + 4 * piece : 0)));
        }
        if constexpr (LOSS != 0) {
            if (!(FL_ABL & 16)) prec = *reinterpret_cast<const u32x4*>(fl.rec + 2 * (rok ? row : 0) + (sub & 1));   // (every thread loads: no divergent branch around a load)
        }
    };
    auto stage = [&](int64_t tile) {
        const bool rok = tile * RB + prow < a.rows;
        if constexpr (LOSS != 0) { if (sub < 2) s_rec[2 * prow + sub] = __builtin_bit_cast(float4, prec); }
#pragma unroll
        for (int p = 0; p < FU_NP; p++) {
            const int piece = sub + 8 * p;
            if (piece >= ppr) continue;
            u32x4 v = pf[p];
            if (!rok) { v[0] = 0u; v[1] = 0u; v[2] = 0u; v[3] = 0u; }
            if (bf) *reinterpret_cast<u32x4*>(tile0 + prow * a.f.ldA + 8 * piece) = v;
            else *reinterpret_cast<uint2*>(tile0 + prow * a.f.ldA + 4 * piece) =
                     make_uint2(fu_pack(fu_u2f(v[0]), fu_u2f(v[1])), fu_pack(fu_u2f(v[2]), fu_u2f(v[3])));
        }
        // columns behind the input that a hidden layer's output (written to this tile two layers ago) may have left non-zero: layer 0 reads them
        // against zero weights, which is only safe while they are finite -- clear them rather than argue
        const int zlo = bf ? opad : O, hpad = (L.hidden + 31) / 32 * 32, zhi = opad > hpad ? opad : hpad;
        for (int e = tid; e < RB * (zhi - zlo); e += FU_THREADS) tile0[(e / (zhi - zlo)) * a.f.ldA + zlo + e % (zhi - zlo)] = 0;
    };
    int64_t tile = bid;
    fetch_index(tile);
    fetch(tile);
    fetch_index(tile + nblk);
    // both tiles start as zeros: columns no layer ever writes are read against zero weights and must hold finite numbers
    for (int e = tid; e < 2 * RB * a.f.ldA / 8; e += FU_THREADS) { const u32x4 z = { 0u, 0u, 0u, 0u }; reinterpret_cast<u32x4*>(fu_lds)[e] = z; }
    __syncthreads();
    // dual: both passes on every tile, as an inner loop over the argument sets (ONE inlined copy of the layers; two call sites spilled 73 registers)
    const int npass = pp.dual ? 2 : 1;
    for (; tile < n_tiles; tile += nblk) {
        const int64_t row0 = tile * RB;
        const int n_rows = a.rows - row0 < RB ? (int)(a.rows - row0) : RB;
        for (int pass = 0; pass < npass; pass++) {
            const FusedForwardArgs& x = pp.a[pp.dual ? pass : second];
            const int64_t next_off0 = pp.a[pp.dual ? (pass ^ 1) : second].f.wp_off[0];   // whose layer 0 comes next: the other net's (same tile, or the next tile's first pass), or this net's
            unsigned int* dbg = nullptr;
#ifdef FU_DBG_STAMPS
            if (bid == 0 && tile == nblk && pass == 0) dbg = reinterpret_cast<unsigned int*>(a.out + a.rows);   // the probe allocates 64 spare outputs
#endif
            FU_STAMP(0);
            // tile0 is free: the previous pass's last layer ended with a barrier.  The second pass stages the tile again from the registers it was fetched into (the
            // layers have overwritten it); only behind the last pass's staging do those registers take the next tile
            stage(tile);
            FU_STAMP(1);
            if (pass == npass - 1) { fetch(tile + nblk); fetch_index(tile + 2 * nblk); }
            FU_STAMP(2);
            __syncthreads();
            FU_STAMP(3);
            fused_layers<FM, NW, BF ? 4 : 1>(x.f, next_off0, tile0, tile1, s_out, x.keep, x.ld_keep, row0, n_rows, pre, dbg);
            const int N = L.out_dim[x.f.net][L.n_layers - 1];
            if constexpr (LOSS == 0) {
                for (int e = tid; e < n_rows * N; e += FU_THREADS) x.out[(row0 + e / N) * N + e % N] = s_out[(e / N) * 32 + e % N];   // (s_out is written again four barriers from here)
            } else {
                constexpr int DIST = LOSS - 1;
                const int net = x.f.net;
                // the thread index through a register the compiler cannot see through: everything the epilogue derives from it (rows, LDS addresses, head offsets) is
                // then formed HERE, per pass, instead of being hoisted out of the tile loop and kept alive across the layers (the kernel sat at 256 registers and spilled)
                int te = tid;
                asm volatile("" : "+v"(te));
                const int lane = te & 63, wave = __builtin_amdgcn_readfirstlane(te >> 6), li = lane & 31, kg = lane >> 5;
                // the top hidden activation sits in the tile the head layer read; the other tile is free
                uint16_t* const hT = (L.n_hidden & 1) ? tile1 : tile0;
                uint16_t* const oT = (L.n_hidden & 1) ? tile0 : tile1;
                const int ldA = x.f.ldA;
                // ---- E1: four lanes per row (lane = head): masked categorical, PPO loss, d logits -- or, in the critic's pass, the value loss and d value ----
                if (te < 256 && !(FL_ABL & 1)) {
                    const int row = te >> 2, h = te & 3;
                    const bool live = row < n_rows;
                    const float4 lr0 = s_rec[2 * row], lr1 = s_rec[2 * row + 1];   // { old log-prob, advantage, return, old value }, { action bytes, mask bits, -, - }
                    const int n_heads = L.n_heads, act = L.act;
                    int A = 0, off = 0;
#pragma unroll
                    for (int hh = 0; hh < 4; hh++) { const int w = hh < n_heads ? L.head_dims[hh] : 0; if (hh < h) off += w; if (hh == h) A = w; }
                    const float clip = fl.hp.clip_coef, lo = 1 - clip, hi_c = 1 + clip, invM = fl.invM;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    float dbs[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                    if (net == 1) {
                        const uint32_t ab = __builtin_bit_cast(uint32_t, lr1.x), mask_bits = DIST == PPO_DIST_MASKED ? __builtin_bit_cast(uint32_t, lr1.y) : 0xffffffffu;
                        float z[4], p[4];
                        bool ok[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int k = off + j;
                            const bool in = j < A;
                            z[j] = s_out[row * 32 + (k < 32 ? k : 31)];
                            ok[j] = in && ((mask_bits >> k) & 1u) != 0;
                            z[j] = in ? ((DIST == PPO_DIST_MASKED && !ok[j]) ? -1e8f : z[j]) : -INFINITY;   // a slot that does not exist: exp -> 0, never the maximum
                            p[j] = 0.0f;
                        }
                        float lp = 0.0f, hH = 0.0f;
                        const int aidx = (int)((ab >> (8 * h)) & 0xffu);
                        if (A > 0) {
                            float mx = -INFINITY;
#pragma unroll
                            for (int j = 0; j < 4; j++) if (j < A) mx = z[j] > mx ? z[j] : mx;
                            float se = 0.0f;
#pragma unroll
                            for (int j = 0; j < 4; j++) if (j < A) { p[j] = fast_exp(z[j] - mx); se += p[j]; }
                            const float lse = fast_log(se) + mx;
                            const float rse = __builtin_amdgcn_rcpf(se);
                            float e = 0.0f;
#pragma unroll
                            for (int j = 0; j < 4; j++) if (j < A) {
                                z[j] = z[j] - lse;
                                p[j] = p[j] * rse;
                                if (DIST == PPO_DIST_CATEGORICAL) {
                                    const float l = z[j] > 1.17549435e-38f ? z[j] : 1.17549435e-38f;   // the reference's clamp (Categorical.cpp:112-119)
                                    e += l * p[j];
                                } else {
                                    const float plp = z[j] * p[j];
                                    e += ok[j] ? plp : 0.0f;
                                }
                                if (j == aidx) lp = z[j];
                            }
                            hH = -e;
                        }
                        // the row's sums in head order on every lane of the row: ((h0 + h1) + h2) + h3 over the heads that exist
                        // (the row's four lanes are one DPP quad: quad_perm broadcasts, no LDS round trips)
#define FL_QUAD(v, K) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v)), (K) * 0x55, 0xf, 0xf, false))
                        float nlp = FL_QUAD(lp, 0), ent = FL_QUAD(hH, 0);
                        { const float l2 = FL_QUAD(lp, 1), e2 = FL_QUAD(hH, 1); if (1 < n_heads) { nlp += l2; ent += e2; } }
                        { const float l2 = FL_QUAD(lp, 2), e2 = FL_QUAD(hH, 2); if (2 < n_heads) { nlp += l2; ent += e2; } }
                        { const float l2 = FL_QUAD(lp, 3), e2 = FL_QUAD(hH, 3); if (3 < n_heads) { nlp += l2; ent += e2; } }
#undef FL_QUAD
                        const float logratio = nlp - lr0.x;
                        const float ratio = fast_exp(logratio);
                        float adv = lr0.y;
                        if (fl.hp.norm_adv) adv = (adv - s_misc[0]) * s_misc[1];
                        const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
                        const float l1 = -adv * ratio, l2 = -adv * rc;
                        const bool inside = (ratio >= lo && ratio <= hi_c);
                        float d_ratio;
                        if (l1 > l2) d_ratio = -adv;
                        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
                        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half / half
                        const float g_nlp = invM * d_ratio * ratio;
                        const float g_ent = -fl.hp.ent_coef * invM;
#pragma unroll
                        for (int j = 0; j < 4; j++) if (j < A) {
                            float d = g_nlp * ((j == aidx ? 1.0f : 0.0f) - p[j]);
                            if (DIST == PPO_DIST_MASKED) d += g_ent * (-p[j] * (z[j] + hH));
                            d = (live && (DIST != PPO_DIST_MASKED || ok[j])) ? d : 0.0f;
                            const uint16_t b = fu_bf16(d);
                            s_dl[row * FL_NP + off + j] = b;
                            s_dlt[(off + j) * FL_DLT_PITCH + row] = b;
                            dbs[j] = d;
                        }
                        if (h == 3) for (int k = act; k < FL_NP; k++) { s_dl[row * FL_NP + k] = 0; s_dlt[k * FL_DLT_PITCH + row] = 0; }   // (the critic's pass wrote column 0 .. of these images)
                        if (h == 0 && live) {
                            s0 = (double)(l1 > l2 ? l1 : l2);
                            s1 = (double)ent;
                            s2 = (double)((ratio - 1.0f) - logratio);
                            s3 = (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
                        }
                        // head bias gradient: the lanes of one head are 4 apart; two rotations inside the 16-lane row leave on every lane the sum over the row's four
                        // samples; lanes 0 .. 3 of the row add head 0 .. 3's sums to the row's own accumulator
                        const int acc16 = (4 * wave + (lane >> 4));
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            float v = dbs[j];
                            v = PPO_DPP_ADD(v, 0x128);   // row_ror:8
                            v = PPO_DPP_ADD(v, 0x124);   // row_ror:4
                            if ((lane & 15) < 4 && j < A) s_db[acc16 * 20 + off + j] += v;
                        }
                        PPO_DPP_ADD_D(s0, 0xB1); PPO_DPP_ADD_D(s1, 0xB1); PPO_DPP_ADD_D(s2, 0xB1); PPO_DPP_ADD_D(s3, 0xB1);
                        PPO_DPP_ADD_D(s0, 0x4E); PPO_DPP_ADD_D(s1, 0x4E); PPO_DPP_ADD_D(s2, 0x4E); PPO_DPP_ADD_D(s3, 0x4E);
                        PPO_DPP_ADD_D(s0, 0x141); PPO_DPP_ADD_D(s1, 0x141); PPO_DPP_ADD_D(s2, 0x141); PPO_DPP_ADD_D(s3, 0x141);
                        PPO_DPP_ADD_D(s0, 0x140); PPO_DPP_ADD_D(s1, 0x140); PPO_DPP_ADD_D(s2, 0x140); PPO_DPP_ADD_D(s3, 0x140);
                        if ((lane & 15) == 0) { s_ls[acc16 * 8 + 0] += s0; s_ls[acc16 * 8 + 1] += s1; s_ls[acc16 * 8 + 2] += s2; s_ls[acc16 * 8 + 3] += s3; }
                    } else {
                        // the value loss (:603-625), once per row on the row's first lane
                        float g_v = 0.0f, lossv = 0.0f;
                        if (h == 0 && live) {
                            const float v = s_out[row * 32], R = lr0.z, vold = lr0.w;
                            const float un = (v - R) * (v - R);
                            if (fl.hp.clip_vloss) {
                                const float dv = v - vold;
                                const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
                                const float vc = vold + dvc;
                                const float cl = (vc - R) * (vc - R);
                                lossv = un > cl ? un : cl;
                                const bool vin = (dv >= -clip && dv <= clip);
                                const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
                                const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
                                g_v = fl.hp.vf_coef * 0.5f * invM * d;
                            } else {
                                lossv = un;
                                g_v = fl.hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {   // lane h clears columns 4 h .. 4 h + 3 of the row (column 0: d value)
                            const uint16_t b = (h == 0 && j == 0) ? fu_bf16(g_v) : (uint16_t)0;
                            s_dl[row * FL_NP + 4 * h + j] = b;
                            s_dlt[(4 * h + j) * FL_DLT_PITCH + row] = b;
                        }
                        const int acc16 = (4 * wave + (lane >> 4));
                        float tv = g_v;
                        tv = PPO_DPP_ADD(tv, 0xB1); tv = PPO_DPP_ADD(tv, 0x4E); tv = PPO_DPP_ADD(tv, 0x141); tv = PPO_DPP_ADD(tv, 0x140);
                        double t4 = (double)lossv;
                        PPO_DPP_ADD_D(t4, 0xB1); PPO_DPP_ADD_D(t4, 0x4E); PPO_DPP_ADD_D(t4, 0x141); PPO_DPP_ADD_D(t4, 0x140);
                        if ((lane & 15) == 0) { s_db[acc16 * 20 + 16] += tv; s_ls[acc16 * 8 + 4] += t4; }
                    }
                }
                __syncthreads();
                // ---- E3: dZ_top block of wave w: rows 32 i + li, columns 32 w + (r & 3) + 8 (r >> 2) + 4 kg; tanh' from the h tile; column sums in f32 ----
                const bool own_cols = 32 * wave < ld_h;   // a hidden vector of 128 columns: waves 0 .. 3 (wave-uniform)
                if (own_cols && !(FL_ABL & 2)) {
                    // D[row][column] (the rows' d logits as the A operand, W_head^T as the B operand): lane (i, kg) then holds ONE column, 32 w + i, and register r
                    // the row 32 blk + (r & 3) + 8 (r >> 2) + 4 kg -- the column sum is an in-lane sum of registers (the other orientation, lane = row, needed
                    // four DPP steps per register and a read-modify-write per value: two thirds of this phase).  h comes by transposing reads: one ds_read_b64_tr_b16
                    // hands a lane four consecutive rows of its column -- a 16-lane group addresses rows R .. R + 3 (lane 4 q + p: row R + q, columns 4 p ..).
                    const u32x4 wf = s_wh[(net * FU_WAVES + wave) * 64 + lane];
                    const int col = 32 * wave + li;
                    const int tr_at = ((lane & 15) >> 2) * ldA + 32 * wave + 16 * ((lane & 31) >> 4) + 4 * (lane & 3);   // element offset of this lane's share of a group's read
                    float csum = 0.0f;
#pragma unroll
                    for (int i = 0; i < FM; i++) {
                        const u32x4 afr = *reinterpret_cast<const u32x4*>(s_dl + (32 * i + li) * FL_NP + 8 * kg);
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr), __builtin_bit_cast(bf16x8, wf), acc, 0, 0, 0);
                        uint2 hv[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) hv[q] = fu_read_tr16(hT + (32 * i + 8 * q + 4 * kg) * ldA + tr_at);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int row = 32 * i + 8 * q + 4 * kg;
                            const float h0 = fu_u2f(hv[q].x << 16), h1 = fu_u2f(hv[q].x & 0xffff0000u), h2 = fu_u2f(hv[q].y << 16), h3 = fu_u2f(hv[q].y & 0xffff0000u);
                            const float v0 = acc[4 * q] * (1.0f - h0 * h0), v1 = acc[4 * q + 1] * (1.0f - h1 * h1);
                            const float v2 = acc[4 * q + 2] * (1.0f - h2 * h2), v3 = acc[4 * q + 3] * (1.0f - h3 * h3);
                            csum += v0; csum += v1; csum += v2; csum += v3;
                            oT[(row + 0) * ldA + col] = fu_bf16(v0); oT[(row + 1) * ldA + col] = fu_bf16(v1);
                            oT[(row + 2) * ldA + col] = fu_bf16(v2); oT[(row + 3) * ldA + col] = fu_bf16(v3);
                        }
                    }
                    s_cs[(2 * net + kg) * ld_h + col] += csum;   // one accumulator per (net, kg half, column): this lane's alone
                }
                // ---- E4: dW_head[n][32 w + i] += sum over the tile's rows of dL[row][n] h[row][32 w + i]: four k steps, h by transposing reads ----
                if (own_cols && !(FL_ABL & 4)) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        const u32x4 af = *reinterpret_cast<const u32x4*>(s_dlt + li * FL_DLT_PITCH + 16 * ks + 8 * kg);
                        const int r = 16 * ks + 8 * kg + ((lane & 15) >> 2);
                        const int col = 32 * wave + 16 * ((lane & 31) >> 4) + 4 * (lane & 3);
                        const uint2 blo = fu_read_tr16(hT + r * ldA + col), bhi = fu_read_tr16(hT + (r + 4) * ldA + col);
                        const u32x4 bfr = { blo.x, blo.y, bhi.x, bhi.y };
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr), acc, 0, 0, 0);
                    }
                    // register r of lane (i, kg) <-> n = (r & 3) + 8 (r >> 2) + 4 kg (< 16: registers 0 .. 7), column 32 w + i
                    const int n_real = N;
                    float* const dw = s_dw + (net == 1 ? 0 : FL_NP) * ld_h + 32 * wave + li;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const int n = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (n < n_real) dw[n * ld_h] += acc[r];
                    }
                }
                __syncthreads();
                // ---- E5: the dZ_top tile leaves in 16-byte row pieces ----
                if (!(FL_ABL & 8)) {
                    const int p8 = ld_h / 8, total = RB * p8;
                    for (int e0 = te; e0 < total; e0 += 4 * FU_THREADS) {
                        u32x4 v[4];
                        int row[4], c8[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int e = e0 + j * FU_THREADS, ee = e < total ? e : 0;
                            row[j] = ee / p8; c8[j] = ee % p8;
                            v[j] = *reinterpret_cast<const u32x4*>(oT + row[j] * ldA + 8 * c8[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (e0 + j * FU_THREADS < total && row[j] < n_rows) {
#ifdef FL_E5_NT
                                __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(fl.dz_top[net] + (row0 + row[j]) * fl.ld_dz + 8 * c8[j]));
#else
                                *reinterpret_cast<u32x4*>(fl.dz_top[net] + (row0 + row[j]) * fl.ld_dz + 8 * c8[j]) = v[j];
#endif
                            }
                    }
                }
                // the next pass stages its input into tile0: with the dZ tile in tile0 (odd depth) the copy above must have read it first; with the h tile there (even
                // depth) the barrier in front of E5 already covers the reads (uniform branch)
                if (L.n_hidden & 1) __syncthreads();
            }
            FU_STAMP(30);
        }
    }
    if constexpr (LOSS != 0) {
        // ---- the workgroup's partials: loss sums, head bias sums, dW_head of both nets, the column sums of dZ_top (every accumulator's last add is behind the barrier above) ----
        const int wg = (int)blockIdx.x, nwg = (int)gridDim.x, act = L.act;
        if (tid < 8) {
            double t = 0.0;
            if (tid < 5) for (int w = 0; w < 16; w++) t += s_ls[w * 8 + tid];
            fl.loss_part[wg * 8 + tid] = t;
            for (int b = wg + nwg; b < GEN_LOSS_BLOCKS; b += nwg) fl.loss_part[b * 8 + tid] = 0.0;   // (a launch of fewer workgroups than partial rows: the consumers add all of them)
        }
        if (tid >= 64 && tid - 64 <= act) {
            const int k = tid - 64, kk = k < act ? k : 16;
            float t = 0.0f;
            for (int w = 0; w < 16; w++) t += s_db[w * 20 + kk];
            fl.head_db_part[wg * (act + 1) + k] = t;
            for (int b = wg + nwg; b < GEN_LOSS_BLOCKS; b += nwg) fl.head_db_part[b * (act + 1) + k] = 0.0f;
        }
        const int hid = L.hidden;
        for (int e = tid; e < act * hid; e += FU_THREADS) fl.head_slab[1][(int64_t)wg * fl.head_slab_stride + e] = s_dw[(e / hid) * ld_h + e % hid];
        for (int e = tid; e < hid; e += FU_THREADS) fl.head_slab[0][(int64_t)wg * fl.head_slab_stride + e] = s_dw[FL_NP * ld_h + e];
        for (int e = tid; e < 2 * (int)fl.ld_cs; e += FU_THREADS) {
            const int net = e / (int)fl.ld_cs, k = e % (int)fl.ld_cs;
            fl.cs_top[net][(int64_t)wg * fl.ld_cs + k] = k < ld_h ? s_cs[2 * net * ld_h + k] + s_cs[(2 * net + 1) * ld_h + k] : 0.0f;
        }
    }
}

FusedNet make_fused_net(const GenericCtx& g, const float* params, int net) {
    FusedNet f{};
    f.L = g.L; f.net = net; f.params = params; f.wfrags = g.wfrags;
    for (int l = 0; l < g.L.n_layers; l++) { f.wp_off[l] = g.wp_off[net][l]; f.wp_kpad[l] = g.wp_kpad[l]; }
    int kmax = 0;   // every layer reads its padded contraction length out of the tile
    for (int l = 0; l < g.L.n_layers; l++) kmax = g.wp_kpad[l] > kmax ? g.wp_kpad[l] : kmax;
    f.ldA = kmax + 8;
    return f;
}
size_t fused_lds_bytes(const FusedNet& f, int rows) { return (size_t)2 * rows * f.ldA * 2 + (size_t)rows * 32 * 4 + 2 * 32 * PPO_MAX_ACT; }

}  // namespace

// The fused kernels need both LDS tiles of the widest layer input to fit (64 rows for the batch kernel) and a head of at most 32 logits.
bool gen_fused_ok(const GenericCtx& g) {
    if (!g.bf16 || g.L.act > 32) return false;
    const FusedNet f = make_fused_net(g, nullptr, 0);
    return fused_lds_bytes(f, 64) <= 150 * 1024;
}

// generic_forward_kernel stages its input as 16-byte pieces held in registers: f32 observations need obs % 4 == 0, and a row at most 96 pieces
bool gen_fused_forward_ok(const GenericCtx& g) {
    if (!gen_fused_ok(g)) return false;
    const int O = g.L.obs, opad = (O + 15) / 16 * 16;
    return (O & 3) == 0 && O / 4 <= 8 * fu_np<false>() && opad / 8 <= 8 * fu_np<true>();
}

static FusedForwardArgs fused_forward_args(const GenericCtx& g, const float* params, int net, const float* x_f32, const uint16_t* x_bf, int64_t ld_x, int64_t rows,
                                           bool keep, float* out, const int32_t* idx) {
    FusedForwardArgs a{};
    a.f = make_fused_net(g, params, net);
    a.x_f32 = x_f32; a.x_bf = x_bf; a.idx = x_bf ? idx : nullptr; a.ld_x = ld_x; a.rows = rows; a.out = out;
    for (int l = 0; l < g.L.n_hidden; l++) a.keep[l] = keep ? g.acts_bf[net][l] : nullptr;
    a.ld_keep = g.ld_h;
    return a;
}
template <int LOSS>
static hipError_t fused_forward_loss_launch(const FusedForwardPair& pp, unsigned blocks, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> lds_ok{0};
    const hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void*>(&generic_forward_kernel<true, LOSS>), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((generic_forward_kernel<true, LOSS>), dim3(blocks), dim3(FU_THREADS), lds, s, pp);
    return hipGetLastError();
}
static hipError_t fused_forward_launch(const FusedForwardPair& pp, unsigned blocks, bool bf, hipStream_t s) {
    const size_t lds = fused_lds_bytes(pp.a[0].f, 64);
    static std::atomic<unsigned long long> lds_ok[2] = {};
    {
        const hipError_t e = bf ? allow_dynamic_lds(lds_ok[0], reinterpret_cast<const void*>(&generic_forward_kernel<true>))
                                : allow_dynamic_lds(lds_ok[1], reinterpret_cast<const void*>(&generic_forward_kernel<false>));
        if (e != hipSuccess) return e;
    }
#ifdef FU_HALF
    if (bf && pp.dual) {   // exploration: two workgroups of four waves and 32-row tiles per CU
        const int64_t t32 = (pp.a[0].rows + 31) / 32;
        const unsigned hb = (unsigned)(t32 < 512 ? t32 : 512);
        FusedForwardPair q = pp;
        q.split = (int)hb;
        hipLaunchKernelGGL((generic_forward_kernel<true, 0, 4, 1>), dim3(hb), dim3(256), fused_lds_bytes(pp.a[0].f, 32), s, q);
        return hipGetLastError();
    }
#endif
    if (bf) hipLaunchKernelGGL(generic_forward_kernel<true>, dim3(blocks), dim3(FU_THREADS), lds, s, pp);
    else hipLaunchKernelGGL(generic_forward_kernel<false>, dim3(blocks), dim3(FU_THREADS), lds, s, pp);
    return hipGetLastError();
}

hipError_t gen_fused_forward(const GenericCtx& g, const float* params, int net, const float* x_f32, const uint16_t* x_bf, int64_t ld_x, int64_t rows, bool keep,
                             float* out, hipStream_t s, const int32_t* idx) {
    if (rows <= 0) return hipSuccess;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    FusedForwardPair pp{};
    pp.a[0] = fused_forward_args(g, params, net, x_f32, x_bf, ld_x, rows, keep, out, idx);
    const int64_t n_tiles = (rows + 63) / 64;
    const unsigned blocks = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    pp.split = (int)blocks;
    return fused_forward_launch(pp, blocks, x_bf != nullptr, s);
}

// Both nets over the same bf16 rows, activations kept, in ONE launch: the actor's workgroups first (the loss kernel behind it needs both outputs)
hipError_t gen_fused_forward_both(const GenericCtx& g, const float* params, const uint16_t* x_bf, int64_t ld_x, int64_t rows, float* logits, float* val,
                                  hipStream_t s, const int32_t* idx) {
    if (rows <= 0) return hipSuccess;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    FusedForwardPair pp{};
    pp.a[0] = fused_forward_args(g, params, 1, nullptr, x_bf, ld_x, rows, true, logits, idx);
    pp.a[1] = fused_forward_args(g, params, 0, nullptr, x_bf, ld_x, rows, true, val, idx);
    const int64_t n_tiles = (rows + 63) / 64;
    const unsigned per = (unsigned)(n_tiles < 256 ? n_tiles : 256);
#ifdef FU_AB_TWO_HALVES   // A/B build: the actor's workgroups, then the critic's (each fetching its own copy of the rows)
    pp.split = (int)per;
    return fused_forward_launch(pp, 2 * per, true, s);
#else
    pp.split = (int)per; pp.dual = 1;   // every workgroup runs both nets on each of its tiles: the rows are fetched once
    return fused_forward_launch(pp, per, true, s);
#endif
}

// Heads, loss and the head layers' backward inside the forward launch (FusedLossArgs above).  Shapes: the slot-wise loss's (<= 4 heads of <= 4 logits, packed row
// records through the index list), hidden widths the backward kernels serve, and room for one partial per workgroup in the head layer's slab and column-sum blocks.
bool gen_fused_loss_ok(const GenericCtx& g, int64_t rows) {
    const GenLayout& L = g.L;
    if (!gen_fused_forward_ok(g) || !gen_fused_backward_ok(g) || !g.obs_bf || !g.row_rec || L.n_heads > 4 || L.act > FL_NP || L.hidden > g.ld_h) return false;
    for (int h = 0; h < L.n_heads; h++) if (L.head_dims[h] > 4) return false;
    const FusedNet f = make_fused_net(g, nullptr, 0);
    if (fused_lds_bytes(f, 64) + fused_loss_lds_bytes(g.ld_h) > 160 * 1024) return false;
    const int64_t n_tiles = (rows + 63) / 64, blocks = n_tiles < 256 ? n_tiles : 256;
    return blocks * (int64_t)L.act * L.hidden <= g.wslab_layer_stride && blocks * g.ld_h <= g.cs_layer_stride && blocks <= GEN_LOSS_BLOCKS;
}
hipError_t gen_fused_forward_loss(const GenericCtx& g, const float* params, const uint16_t* x_bf, int64_t ld_x, int64_t rows, const LossParams& hp, double inv_global_M,
                                  double global_M, const AdvStat* adv_stat, const int32_t* idx, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (!gen_fused_loss_ok(g, rows) || !idx || !g.rows_rec) return hipErrorInvalidValue;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    const GenLayout& L = g.L;
    const int top = L.n_layers - 1;
    FusedForwardPair pp{};
    pp.a[0] = fused_forward_args(g, params, 1, nullptr, x_bf, ld_x, rows, true, nullptr, idx);
    pp.a[1] = fused_forward_args(g, params, 0, nullptr, x_bf, ld_x, rows, true, nullptr, idx);
    pp.a[0].keep[L.n_hidden - 1] = nullptr;   // the top hidden activation is consumed in LDS: it never reaches HBM
    pp.a[1].keep[L.n_hidden - 1] = nullptr;
    const int64_t n_tiles = (rows + 63) / 64;
    const unsigned blocks = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    pp.split = (int)blocks; pp.dual = 1;
    FusedLossArgs& fl = pp.loss;
    fl.hp = hp; fl.invM = (float)inv_global_M; fl.global_M = global_M; fl.adv_stat = (adv_stat && hp.norm_adv) ? adv_stat : nullptr;
    fl.idx = idx; fl.rec = g.rows_rec;
    fl.ld_dz = g.ld_h; fl.ldw = g.wp_kpad[top]; fl.head_slab_stride = (int64_t)L.act * L.hidden; fl.ld_cs = g.ld_h;
    for (int net = 0; net < 2; net++) {
        fl.dz_top[net] = g.dz_bf[net][top & 1];
        fl.whead[net] = g.wplanes + g.wp_off[net][top];
        fl.head_slab[net] = (net == 1 ? g.wslab1 : g.wslab) + (size_t)top * g.wslab_layer_stride;
        fl.cs_top[net] = g.cs_part[net] + (size_t)top * g.cs_layer_stride;
    }
    fl.head_db_part = g.head_db_part; fl.loss_part = g.loss_part;
    const size_t lds = fused_lds_bytes(pp.a[0].f, 64) + fused_loss_lds_bytes(g.ld_h);
    g.head_fused = (int)blocks;
    return hp.dist_kind == PPO_DIST_MASKED ? fused_forward_loss_launch<PPO_DIST_MASKED + 1>(pp, blocks, lds, s) : fused_forward_loss_launch<PPO_DIST_CATEGORICAL + 1>(pp, blocks, lds, s);
}

hipError_t gen_fused_rollout(const GenericCtx& g, const float* params, int dist_kind, int N, int T, int max_episode_steps, int64_t seed, int64_t env_offset,
                             int64_t step_base, int32_t* ep_len, float* ep_rew, float* obs, uint8_t* masks, int32_t* actions, float* logprobs, float* rewards,
                             float* dones, int32_t* fin_len, float* fin_rew, float* next_obs, int32_t* next_done, uint8_t* cur_mask, const int64_t* forced,
                             hipStream_t s) {
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    FusedRolloutArgs a{};
    a.f = make_fused_net(g, params, 1);
    a.dist_kind = dist_kind; a.N = N; a.T = T; a.max_episode_steps = max_episode_steps;
    a.seed = seed; a.env_offset = env_offset; a.step_base = step_base;
    a.ep_len = ep_len; a.ep_rew = ep_rew; a.obs = obs; a.masks = masks; a.actions = actions; a.logprobs = logprobs; a.rewards = rewards; a.dones = dones;
    a.fin_len = fin_len; a.fin_rew = fin_rew; a.next_obs = next_obs; a.next_done = next_done; a.cur_mask = cur_mask; a.forced = forced;
    const size_t lds = fused_lds_bytes(a.f, 32);
    const dim3 grid((unsigned)((N + 31) / 32)), block(FU_THREADS);
    static std::atomic<unsigned long long> lds_ok[2] = {};
    {
        const hipError_t e = dist_kind == PPO_DIST_MASKED ? allow_dynamic_lds(lds_ok[0], reinterpret_cast<const void*>(&generic_rollout_kernel<PPO_DIST_MASKED>))
                                                          : allow_dynamic_lds(lds_ok[1], reinterpret_cast<const void*>(&generic_rollout_kernel<PPO_DIST_CATEGORICAL>));
        if (e != hipSuccess) return e;
    }
    if (dist_kind == PPO_DIST_MASKED) hipLaunchKernelGGL(generic_rollout_kernel<PPO_DIST_MASKED>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(generic_rollout_kernel<PPO_DIST_CATEGORICAL>, grid, block, lds, s, a);
    return hipGetLastError();
}
