// ppo-libtorch_amd/csrc/kernels_update_mfma.hip -- K5-K8 (gather + forward + PPO loss + backward) on the CDNA4 matrix cores.
//
// Same contract as fwd_bwd_kernel in kernels_update.hip (reference PPO_Discrete.cpp:576-638), different machine mapping:
// the three 64x64 contractions per sample and net -- layer-2 forward, d(hidden 1), and the weight gradient dW2 -- run on the
// matrix cores, either as exact-fp32 MFMAs (PREC_F32) or as bf16 MFMAs over exact three-term splits of the fp32 operands
// (PREC_BF16X3, the default; see the PREC comment below).  Matrix cores are used only here because only here is the minibatch
// (131 072 rows at BASELINE configs[1]) a real contraction.  values_mfma_kernel at the end of the file is the forward half of
// the same mapping for the batched critic evaluation of the rollout.
//
// One WAVE owns a 32-sample tile of one net from gather to weight gradient; a workgroup is four (PREC_F32) or eight
// (PREC_BF16X3) such waves of the same net sharing LDS copies of the weights.  Register layout of every hidden vector is the
// MFMA C/D layout
//     lane (s = lane & 31, hi = lane >> 5), element e = r + 16 t   <->   unit  U(r, hi, t) = (r & 3) + 8 (r >> 2) + 4 hi + 32 t
// i.e. lane = sample, registers = 32 of the 64 units.  Because the contraction index of an MFMA may be enumerated in any
// order as long as A and B agree, a D-layout vector is directly the B operand of the next product when the weight operand is
// fetched in that same order: forward and d(hidden) need NO data movement.
// Only the products that contract over SAMPLES (dW2, dW3, dW1, bias gradients) need lane = unit: the tile is bounced through
// a private 32 x 68-float LDS image (4 times per tile).
// Weight-gradient accumulators (64 registers for dW2) live in registers across all tiles of the wave; the waves of a
// workgroup are then added in a fixed order through LDS and leave as ONE partial slab (deterministic, no float atomics).
#include "ppo_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT = 32;              // samples per wave tile
constexpr int LS = 68;              // padded LDS row stride (floats)
constexpr int NS = 68;              // bf16 per row of the update kernel's weight image (136 B: 32 rows x 8 B land on 64 distinct banks)
constexpr int WS = 72;              // bf16 per padded row of a split weight operand (144 B: 16 rows x 16 B land on 16 distinct bank groups)

// PREC selects the arithmetic of the three 64-wide contractions:
//   PREC_F32    v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain in k order).  On gfx950 this instruction runs at the vector fp32
//               rate and, measured (tools/probes/mfma_coexec.hip), excludes every other vector instruction of the SIMD while it
//               executes: matrix and vector time ADD, so the kernel cannot pass MFMA / (MFMA + VALU) of the fp32 matrix peak.
//   PREC_BF16X3 fp32 operands are cut (by truncation, exactly: x = t1 + t2 + t3, 8 mantissa bits each) into three bf16 terms and
//               every fp32 product a.b is issued as the six bf16 products a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 on
//               v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped products are <= 3 x 2^-24 |a||b|, i.e. the result
//               carries fp32 accuracy (no range loss: bf16 has the fp32 exponent).  16x the k-depth per instruction at half the
//               cycles makes the matrix time ~2.7x smaller, and bf16 MFMAs do co-execute with vector instructions.
constexpr int PREC_F32 = 0, PREC_BF16X3 = 1;
__host__ __device__ constexpr int mf_waves(int prec) { return prec == PREC_BF16X3 ? 8 : 4; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
// (x0, x1) -> three dwords of packed bf16 pairs (low half = x0's term), exact: x = t1 + t2 + t3 when the exponent does not underflow
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t u0 = f2u(x0), u1 = f2u(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - u2f(u0 & 0xffff0000u), r1 = x1 - u2f(u1 & 0xffff0000u);
    const uint32_t v0 = f2u(r0), v1 = f2u(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - u2f(v0 & 0xffff0000u), q1 = r1 - u2f(v1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(f2u(q1), f2u(q0), 0x07060302u);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// the six products of one (A chunk, B chunk) pair, small terms first
__device__ __forceinline__ f32x16 mfma_x3(const u32x4 a1, const u32x4 a2, const u32x4 a3, const u32x4 b1, const u32x4 b2, const u32x4 b3, f32x16 acc) {
    acc = mfma_bf16(a3, b1, acc);
    acc = mfma_bf16(a1, b3, acc);
    acc = mfma_bf16(a2, b2, acc);
    acc = mfma_bf16(a2, b1, acc);
    acc = mfma_bf16(a1, b2, acc);
    acc = mfma_bf16(a1, b1, acc);
    return acc;
}
// ds_read_b64_tr_b16: per group of 16 lanes a 4-row x 16-column block of 16-bit elements is read and delivered column-major: lane
// 4q + p of the group supplies the address of row q, columns 4p .. 4p+3; lane i receives column i of the four rows (row q in its
// element q).  EXEC must be all ones.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const uint16_t* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}

// slot (chunk c, half h, element e) of hidden unit k inside a 64-wide operand row: unit U(r, h, t) sits at c = 2 t + (r >> 3), e = r & 7,
// which is where a D-layout register vector presents it (register 8 c + e of half h)
__host__ __device__ inline int slot_of_unit(int k) {
    const int t = k >> 5, kk = k & 31, h = (kk >> 2) & 1, r = (kk & 3) | ((kk >> 3) << 2);
    return (2 * t + (r >> 3)) * 16 + h * 8 + (r & 7);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int umap(int r, int hi, int t) { return (r & 3) + 8 * (r >> 2) + 4 * hi + 32 * t; }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct MfSmem {
    int w2, w2t, w1, b1, b2, w3, b3, wave0, wave_stride, s_img, s_x, s_do, total;  // offsets in floats
};
__host__ __device__ inline MfSmem mf_smem(int obs, int aout, int prec) {
    MfSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    if (prec == PREC_BF16X3) {
        // ONE image per term, natural [n][k] order: the forward product reads rows (two 8-byte pieces per fragment), the backward
        // product reads the same bytes column-wise with the transposing LDS read ds_read_b64_tr_b16
        m.w2 = take(3 * 64 * NS / 2);
        m.w2t = m.w2;
    } else {
        m.w2 = take(64 * LS);     // [n][k], padded rows
        m.w2t = take(64 * LS);    // [k][n]
    }
    m.w1 = take(64 * obs);
    m.b1 = take(64);
    m.b2 = take(64);
    m.w3 = take(aout * 64);
    m.b3 = take(aout);
    m.wave0 = o;
    int w = 0;
    auto takew = [&](int n) { int r = w; w += (n + 3) & ~3; return r; };
    m.s_img = takew(MT * LS);   // [sample][unit] bounce image
    m.s_x = takew(obs * MT);    // [o][sample]
    m.s_do = takew(aout * MT);  // [a][sample]
    m.wave_stride = w;
    m.total = o + mf_waves(prec) * w;
    return m;
}

// Stores a D-layout vector (32 regs) into the [sample][unit] image: 8 x 16-byte stores per lane.
__device__ __forceinline__ void store_dlayout(float* img, const float* v, int s, int hi) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            st4(&img[s * LS + 8 * q + 4 * hi + 32 * t], make_float4(v[16 * t + 4 * q], v[16 * t + 4 * q + 1], v[16 * t + 4 * q + 2], v[16 * t + 4 * q + 3]));
}

// In-kernel phase stamps (diagnostic variant only, STAMP = true): cycles per phase of wave 0 of workgroup 0 of each net,
// summed over its tiles, written to a debug buffer no other code reads.
#define MF_STAMP(i)                                                                            \
    do {                                                                                       \
        if constexpr (STAMP) {                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
            __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            ph[i] += now_ - t_prev;                                                            \
            t_prev = now_;                                                                     \
        }                                                                                      \
    } while (0)

// EXACT: the policy has ONE head of exactly AMAX actions (the reference's PPO_Discrete / MountainCar shapes): every head loop
// folds at compile time.  Otherwise head count and widths are run-time values bounded by AMAX.
// Scheduling pin between an operand prefetch and the MFMA group it must stay ahead of: vector ALU, scalar and transcendental
// instructions may still move across it, LDS reads and MFMAs may not.
#define MF_PIN() __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400)

template <int NET, int DIST, int OBS, int AMAX, bool EXACT, int PREC, bool STAMP>
__device__ __forceinline__ void mf_body(const UpdateArgs& a, float* smem, const int blk, const int nblk) {
    unsigned long long ph[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long t_prev = 0;
    if constexpr (STAMP) t_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long t_prev0 = t_prev;
    const NetLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, hi = lane >> 5;
    const int AOUT = NET == 0 ? 1 : (EXACT ? AMAX : L.act);
    const int n_heads = EXACT ? 1 : L.n_heads;
    constexpr int MF_WAVES = mf_waves(PREC), MF_THREADS = 64 * MF_WAVES;
    const MfSmem m = mf_smem(OBS, AOUT, PREC);
    const uint16_t* sW2p = reinterpret_cast<const uint16_t*>(smem + m.w2);    // PREC_BF16X3 views
    const uint16_t* sW2Tp = reinterpret_cast<const uint16_t*>(smem + m.w2t);
    float* sW2 = smem + m.w2;
    float* sW2T = smem + m.w2t;
    float* sW1 = smem + m.w1;
    float* sB1 = smem + m.b1;
    float* sB2 = smem + m.b2;
    float* sW3 = smem + m.w3;
    float* sB3 = smem + m.b3;
    float* wbase = smem + m.wave0 + wave * m.wave_stride;
    float* img = wbase + m.s_img;
    float* sX = wbase + m.s_x;
    float* sDo = wbase + m.s_do;
    const float* __restrict__ P = a.params;

    // ---- weights of this net -> LDS (once per launch) ----
    // With a deferred optimizer step (a.opt.pending; ppo_internal.hpp: DeferredOpt) the weights loaded here are those BEFORE that step:
    // every thread applies clip + AdamW to the elements it loads (gradient, both moments: three more loads per element, issued together
    // with the weight's), uses the result, and workgroup 0 of the net also writes the new state to the other set of buffers.  That folds
    // the optimizer launch (5.0 us, 40 times per update) into a prologue that was loading these weights anyway -- and measured, it LOSES:
    // 13 elements per thread of IEEE sqrt + two divisions, replicated in all 256 workgroups, plus the norm's barrier make this kernel
    // 5.8 us longer (A/B in one gpurun call: 162.2 against 164.3 M env-steps/s).  Bit-identical to the stand-alone step (tested); enabled
    // only by PPO_DEFER_OPT=1.
    const bool opt = a.opt.pending != 0;
    const bool opt_writer = opt && blk == 0;
    // Every element this thread brings in is a slot: 4096 / threads of W2, then its share of W1, W3, b1, b2, b3.  ALL loads of all slots
    // (weight, and with a pending step gradient + both moments) are issued before anything is used: one memory round trip for the prologue
    // (a load-use-load chain per tensor cost 6 us per launch, more than the optimizer launch it replaces).
    constexpr int NW2 = PREC == PREC_BF16X3 ? 4096 / MF_THREADS : 16;
    constexpr int NW1 = (64 * OBS + MF_THREADS - 1) / MF_THREADS, NW3 = (AMAX * 64 + MF_THREADS - 1) / MF_THREADS;
    constexpr int NS_ = NW2 + NW1 + NW3 + 3;
    int se[NS_];          // global parameter index, -1: no element in this slot
#pragma unroll
    for (int i = 0; i < NW2; i++) se[i] = L.w2[NET] + tid + i * MF_THREADS;
#pragma unroll
    for (int i = 0; i < NW1; i++) { const int e = tid + i * MF_THREADS; se[NW2 + i] = e < 64 * OBS ? L.w1[NET] + e : -1; }
#pragma unroll
    for (int i = 0; i < NW3; i++) { const int e = tid + i * MF_THREADS; se[NW2 + NW1 + i] = e < AOUT * 64 ? L.w3[NET] + e : -1; }
    se[NS_ - 3] = tid < 64 ? L.b1[NET] + tid : -1;
    se[NS_ - 2] = tid < 64 ? L.b2[NET] + tid : -1;
    se[NS_ - 1] = tid < AOUT ? L.b3[NET] + tid : -1;
    float wv[NS_], og[NS_], om[NS_], ov[NS_];
#pragma unroll
    for (int i = 0; i < NS_; i++) wv[i] = P[se[i] < 0 ? 0 : se[i]];
    if (opt) {
#pragma unroll
        for (int i = 0; i < NS_; i++) {
            const int e = se[i] < 0 ? 0 : se[i];
            og[i] = a.opt.grads[e]; om[i] = a.opt.m_src[e]; ov[i] = a.opt.v_src[e];
        }
        double* n2s = reinterpret_cast<double*>(smem + m.wave0);   // the wave regions are idle until the tile loop
        double ls[5] = { 0, 0, 0, 0, 0 }, cf0 = 0.0, cf1 = 0.0;
        const bool stat_thread = NET == 0 && blk == 0 && tid == 0;
        if (stat_thread) {
            for (int i = 0; i < 5; i++) ls[i] = a.opt.sums[i];
            if (a.opt.clipfrac_accum) { cf0 = a.opt.clipfrac_accum[0]; cf1 = a.opt.clipfrac_accum[1]; }
        }
        const AdamCoef kco = *a.opt.coef;
        const float total = opt_total_norm(L, a.opt.partial, n2s, tid);   // one barrier inside
        const float clipc = opt_clip_coef(total, a.opt.max_norm);
        if (stat_thread) opt_write_stats(ls, cf0, cf1, a.opt.global_M, a.opt.hp, total, a.opt.stats_out, a.opt.clipfrac_accum);
#pragma unroll
        for (int i = 0; i < NS_; i++) {
            adamw_apply(og[i], clipc, kco, wv[i], om[i], ov[i]);
            if (opt_writer && se[i] >= 0) { a.opt.p_dst[se[i]] = wv[i]; a.opt.m_dst[se[i]] = om[i]; a.opt.v_dst[se[i]] = ov[i]; }
        }
    }
    if constexpr (PREC == PREC_BF16X3) {
        // 4096 weights, 8 per thread: each is cut into its three bf16 terms once per launch and stored at its natural [n][k] place
        uint16_t* wn = reinterpret_cast<uint16_t*>(smem + m.w2);
#pragma unroll
        for (int i = 0; i < NW2; i++) {
            const int e = tid + i * MF_THREADS;
            const int n = e >> 6, k = e & 63;
            const uint32_t u0 = f2u(wv[i]);
            const float r1 = wv[i] - u2f(u0 & 0xffff0000u);
            const uint32_t u1 = f2u(r1);
            const float r2 = r1 - u2f(u1 & 0xffff0000u);
            const int pn = n * NS + k;
            wn[pn] = (uint16_t)(u0 >> 16); wn[64 * NS + pn] = (uint16_t)(u1 >> 16); wn[2 * 64 * NS + pn] = (uint16_t)(f2u(r2) >> 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NW2; i++) {
            const int e = tid + i * MF_THREADS;
            const int n = e >> 6, k = e & 63;
            sW2[n * LS + k] = wv[i];
            sW2T[k * LS + n] = wv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NW1; i++) { const int e = tid + i * MF_THREADS; if (e < 64 * OBS) sW1[e] = wv[NW2 + i]; }
#pragma unroll
    for (int i = 0; i < NW3; i++) { const int e = tid + i * MF_THREADS; if (e < AOUT * 64) sW3[e] = wv[NW2 + NW1 + i]; }
    if (tid < 64) { sB1[tid] = wv[NS_ - 3]; sB2[tid] = wv[NS_ - 2]; }
    if (tid < AOUT) sB3[tid] = wv[NS_ - 1];

    // ---- gradient accumulators of this wave ----
    f32x16 gW2[2][2];  // [tn][tk]: D[i = n%32][j = k%32]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) gW2[i][j][r] = 0.0f;
    float gW3[AMAX], gW1[OBS];     // lane = unit
#pragma unroll
    for (int k = 0; k < AMAX; k++) gW3[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < OBS; k++) gW1[k] = 0.0f;
    float gb1 = 0.0f, gb2[2] = { 0.0f, 0.0f }, gb3[AMAX];
#pragma unroll
    for (int k = 0; k < AMAX; k++) gb3[k] = 0.0f;
    double st0 = 0.0, st1 = 0.0, st2 = 0.0, st3 = 0.0;
    const float clip = a.hp.clip_coef;
    const float lo = 1 - clip, hi_c = 1 + clip;
    const float invM = (float)a.inv_global_M;
    float mean_f = 0.0f, std_f = 0.0f;
    if (NET == 1 && a.hp.norm_adv) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += a.adv_stat[i].s1; t2 += a.adv_stat[i].s2; }
        const double mean = t1 / a.global_M;
        const double var = (t2 - t1 * mean) / (a.global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var > 0.0 ? var : 0.0);
    }
    const float inv_std = 1.0f / (std_f + 1e-8f);
    __syncthreads();
    // layer-1 A operands: W1[u = s + 32 t][o = 2 st + hi] (zero beyond OBS); K = OBS is contracted in ceil(OBS/2) MFMA steps
    constexpr int L1S = (OBS + 1) / 2;
    float w1op[2][L1S];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int stp = 0; stp < L1S; stp++) {
            const int o = 2 * stp + hi;
            w1op[t][stp] = o < OBS ? sW1[(s + 32 * t) * OBS + o] : 0.0f;   // from LDS: the weights there include a deferred step
        }

    MF_STAMP(0);   // prologue: weights -> LDS

    const int n_tiles = (a.M + MT - 1) / MT;
    const int tile_step = nblk * MF_WAVES;
    // gather of the first tile; every later tile's batch row is fetched one tile ahead (its two dependent HBM round trips
    // hide behind the current tile's MFMAs)
    int tile = blk * MF_WAVES + wave;
    // software pipeline of the gather: batch-row indices are fetched TWO tiles ahead and the observation row ONE tile ahead, so
    // neither of the two dependent HBM round trips is ever waited for inside a tile
    // Every load of the pipeline is UNCONDITIONAL (clamped address, value selected afterwards): a load inside a branch makes the
    // compiler's wait-count bookkeeping give up at the join and emit s_waitcnt vmcnt(0) at the next use of ANY loaded value, which
    // stalled every tile for a full memory round trip on the prefetches just issued.
    auto fetch_row = [&](int tl) -> int {
        const int j = tl * MT + s;
        const bool ok = tl < n_tiles && j < a.M;
        const int v = a.idx[ok ? j : 0];
        return ok ? v : -1;
    };
    auto load_obs = [&](int rown, float* xo) {   // row 0 stands in for a missing row; the caller zeroes it at the point of use
        const size_t rn = rown < 0 ? 0 : rown;
        if constexpr (OBS == 4) {
            const float4 v = *reinterpret_cast<const float4*>(a.obs + rn * 4);
            xo[0] = v.x; xo[1] = v.y; xo[2] = v.z; xo[3] = v.w;
        } else if constexpr (OBS == 2) {
            const float2 v = *reinterpret_cast<const float2*>(a.obs + rn * 2);
            xo[0] = v.x; xo[1] = v.y;
        } else {
#pragma unroll
            for (int o = 0; o < OBS; o++) xo[o] = a.obs[rn * OBS + o];
        }
    };
    int row_n = fetch_row(tile);
    int row_nn = fetch_row(tile + tile_step);
    float x_n[OBS];
    load_obs(row_n, x_n);
    const int wave_half = __builtin_amdgcn_readfirstlane(wave >> 2) & 1;   // SIMD partners are waves w and w + 4
    for (int it = 0; tile < n_tiles; tile += tile_step, it++) {
        if constexpr (MF_WAVES == 8) {
            // issue priority alternates between the two waves of a SIMD tile by tile: with equal priorities the older wave wins every
            // arbitration and finishes its tiles ~25% earlier, leaving its partner to run the tail alone
            if ((it ^ wave_half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        // ---------------- gather (K5): lanes (s, 0) and (s, 1) read the same batch row ----------------
        const bool valid = row_n >= 0;
        const int row = valid ? row_n : 0;
        float x[OBS];
#pragma unroll
        for (int o = 0; o < OBS; o++) x[o] = valid ? x_n[o] : 0.0f;
        // per-sample scalars of this tile: issued now, consumed at the loss
        float s_oldlp = 0.0f, s_adv = 0.0f, s_ret = 0.0f, s_oldv = 0.0f;
        int act_s[AMAX];
        uint32_t s_maskbits = 0xffffffffu;
#pragma unroll
        for (int h = 0; h < AMAX; h++) act_s[h] = 0;
        if (NET == 1) {
            s_oldlp = a.logprobs[row];
            s_adv = a.advantages[row];
#pragma unroll
            for (int h = 0; h < AMAX; h++) if (h < n_heads) act_s[h] = a.actions[(size_t)row * n_heads + h];
            if (DIST == PPO_DIST_MASKED && a.masks) {
                s_maskbits = 0u;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k < AOUT) s_maskbits |= (a.masks[(size_t)row * AOUT + k] ? 1u : 0u) << k;
            }
        } else {
            s_ret = a.returns[row];
            s_oldv = a.values[row];
        }
        {   // next tile's observation row (its index arrived a tile ago); the index after that
            row_n = row_nn;
            load_obs(row_n, x_n);
            row_nn = fetch_row(tile + 2 * tile_step);
        }
        if (hi == 0) {
#pragma unroll
            for (int o = 0; o < OBS; o++) sX[o * MT + s] = x[o];
        }

        MF_STAMP(1);   // gather hand-over + prefetch issue
        // ---------------- layer 1 (MFMA, K = OBS): z1^T[u][s] = b1[u] + sum_o W1[u][o] x[s][o], D layout ----------------
        float h1[32];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 b = ld4(&sB1[8 * q + 4 * hi + 32 * t]);
                acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
#pragma unroll
            for (int stp = 0; stp < L1S; stp++) {
                const float xb = (2 * stp + 1 < OBS) ? (hi ? x[2 * stp + 1] : x[2 * stp]) : (hi ? 0.0f : x[2 * stp]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1op[t][stp], xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) h1[16 * t + r] = tanh_mufu(acc[r]);
        }

        MF_STAMP(2);   // layer 1 + tanh
        // ---------------- layer 2 forward (MFMA): z2^T[n][s] = b2[n] + sum_k W2[n][k] h1[s][k] ----------------
        float h2[32];
        if constexpr (PREC == PREC_BF16X3) {
            // B operand = h1 itself (D layout): chunk c of the contraction is registers 8c .. 8c+7, cut into bf16 terms in place
            uint32_t hp[3][16];
#pragma unroll
            for (int j = 0; j < 16; j++) split3(h1[2 * j], h1[2 * j + 1], hp[0][j], hp[1][j], hp[2][j]);
            // g = 4 t + c: row n = s + 32 t, contraction chunk c.  Fragment element e <-> k = 16 c + 8 (e >> 2) + 4 hi + (e & 3), the
            // unit register 8 c + e of a D-layout vector holds: two 8-byte pieces of the natural row
            auto afrag = [&](int g, int term) -> u32x4 {
                const uint16_t* q = sW2p + term * 64 * NS + (s + 32 * (g >> 2)) * NS + (g & 3) * 16 + hi * 4;
                const uint2 lo2 = *reinterpret_cast<const uint2*>(q), hi2 = *reinterpret_cast<const uint2*>(q + 8);
                const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                return r;
            };
            f32x16 acc;
            u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1), an3 = afrag(0, 2);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int t = g >> 2, c = g & 3;
                if (c == 0) {
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const float4 b = ld4(&sB2[8 * qq + 4 * hi + 32 * t]);
                        acc[4 * qq] = b.x; acc[4 * qq + 1] = b.y; acc[4 * qq + 2] = b.z; acc[4 * qq + 3] = b.w;
                    }
                }
                const u32x4 a1 = an1, a2 = an2, a3 = an3;
                if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); an3 = afrag(g + 1, 2); }
                MF_PIN();
                const u32x4 b1 = { hp[0][4 * c], hp[0][4 * c + 1], hp[0][4 * c + 2], hp[0][4 * c + 3] };
                const u32x4 b2 = { hp[1][4 * c], hp[1][4 * c + 1], hp[1][4 * c + 2], hp[1][4 * c + 3] };
                const u32x4 b3 = { hp[2][4 * c], hp[2][4 * c + 1], hp[2][4 * c + 2], hp[2][4 * c + 3] };
                acc = mfma_x3(a1, a2, a3, b1, b2, b3, acc);
                if (c == 3) {
#pragma unroll
                    for (int r = 0; r < 16; r++) h2[16 * t + r] = tanh_mufu(acc[r]);
                }
            }
        } else
        {
            // 16 groups of 4 MFMAs (g = 8 t + 4 tk + q); the 16-byte weight operand of group g + 1 is read from LDS while group g
            // runs (PIN keeps the scheduler from sinking the read back next to its use, where its ~120-cycle latency is exposed)
            auto wptr = [&](int g) { return &sW2[(s + 32 * (g >> 3)) * LS + 8 * (g & 3) + 4 * hi + 32 * ((g >> 2) & 1)]; };  // W2[n][U(4q.., hi, tk)]
            f32x16 acc;
            float4 wn = ld4(wptr(0));
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int t = g >> 3, tk = (g >> 2) & 1, q = g & 3;
                if ((g & 7) == 0) {
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const float4 b = ld4(&sB2[8 * qq + 4 * hi + 32 * t]);
                        acc[4 * qq] = b.x; acc[4 * qq + 1] = b.y; acc[4 * qq + 2] = b.z; acc[4 * qq + 3] = b.w;
                    }
                }
                const float4 w = wn;
                if (g + 1 < 16) wn = ld4(wptr(g + 1));
                MF_PIN();
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h1[16 * tk + 4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h1[16 * tk + 4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h1[16 * tk + 4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h1[16 * tk + 4 * q + 3], acc, 0, 0, 0);
                if ((g & 7) == 7) {
#pragma unroll
                    for (int r = 0; r < 16; r++) h2[16 * t + r] = tanh_mufu(acc[r]);
                }
            }
        }

        MF_STAMP(3);   // layer 2 MFMA + tanh
        // The per-sample scalars were requested at the top of the tile; nothing may touch them before this point (left alone, the
        // scheduler hoists the cheap `action == k` compare to the top of the tile, right behind the load, and the wave then sits
        // out a memory round trip there).  An empty asm that "redefines" them pins every use below this line.
        asm volatile("" : "+v"(s_oldlp), "+v"(s_adv), "+v"(s_ret), "+v"(s_oldv));
#pragma unroll
        for (int h = 0; h < AMAX; h++) asm volatile("" : "+v"(act_s[h]));
        asm volatile("" : "+v"(s_maskbits));
        // ---------------- head + loss (K6, K7): both half-lanes of a sample compute the same scalars ----------------
        float dOut[AMAX];
#pragma unroll
        for (int k = 0; k < AMAX; k++) dOut[k] = 0.0f;
        if (NET == 0) {
            float part = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; g++) {   // g = 4 t + q: units 8q + 4hi + 32t .. +3
                const float4 w = ld4(&sW3[8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                part = __builtin_fmaf(h2[4 * g], w.x, part); part = __builtin_fmaf(h2[4 * g + 1], w.y, part);
                part = __builtin_fmaf(h2[4 * g + 2], w.z, part); part = __builtin_fmaf(h2[4 * g + 3], w.w, part);
            }
            const float other = __shfl_xor(part, 32, 64);
            const float v = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[0];  // same association in both halves
            const float R = s_ret, vold = s_oldv;
            const float un = (v - R) * (v - R);
            float g_v, lossv;
            if (a.hp.clip_vloss) {   // PPO_Discrete.cpp:603-620
                const float dv = v - vold;
                const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
                const float vc = vold + dvc;
                const float cl = (vc - R) * (vc - R);
                lossv = un > cl ? un : cl;
                const bool vin = (dv >= -clip && dv <= clip);
                const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
                const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
                g_v = a.hp.vf_coef * 0.5f * invM * d;
            } else {                 // :622-625
                lossv = un;
                g_v = a.hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
            }
            if (valid && hi == 0) st0 += (double)lossv;
            dOut[0] = valid ? g_v : 0.0f;
        } else {
            const int A = AOUT;
            float z[AMAX], pr[AMAX];
            bool ok[AMAX];
#pragma unroll
            for (int k = 0; k < AMAX; k++) {
                z[k] = 0.0f; pr[k] = 0.0f; ok[k] = true;
                if (k < A) {
                    float part = 0.0f;
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        const float4 w = ld4(&sW3[k * 64 + 8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                        part = __builtin_fmaf(h2[4 * g], w.x, part); part = __builtin_fmaf(h2[4 * g + 1], w.y, part);
                        part = __builtin_fmaf(h2[4 * g + 2], w.z, part); part = __builtin_fmaf(h2[4 * g + 3], w.w, part);
                    }
                    const float other = __shfl_xor(part, 32, 64);
                    z[k] = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[k];
                    if (DIST == PPO_DIST_MASKED && a.masks) ok[k] = ((s_maskbits >> k) & 1u) != 0u;
                    if (DIST == PPO_DIST_MASKED && !ok[k]) z[k] = -1e8f;
                }
            }
            float nlp = 0.0f, ent = 0.0f;
            float headH[AMAX];
#pragma unroll
            for (int h = 0; h < AMAX; h++) headH[h] = 0.0f;
            int off = 0;
#pragma unroll
            for (int h = 0; h < AMAX; h++) {
                if (h >= n_heads) break;
                const int Ah = EXACT ? AMAX : L.head_dims[h];
                const int act_h = act_s[h];
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) mx = z[k] > mx ? z[k] : mx;
                float se = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) { pr[k] = fast_exp(z[k] - mx); se += pr[k]; }
                const float lse = fast_log(se) + mx;
                const float rse = __builtin_amdgcn_rcpf(se);
                float e1 = 0.0f, lp = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) {
                    z[k] = z[k] - lse;
                    pr[k] = pr[k] * rse;
                    if (DIST == PPO_DIST_CATEGORICAL) {
                        const float l = z[k] > 1.17549435e-38f ? z[k] : 1.17549435e-38f;
                        e1 += l * pr[k];
                    } else {
                        e1 += ok[k] ? z[k] * pr[k] : 0.0f;
                    }
                    if (k == off + act_h) lp = z[k];
                }
                headH[h] = -e1;
                if (h == 0) { nlp = lp; ent = headH[h]; } else { nlp += lp; ent += headH[h]; }
                off += Ah;
            }
            const float logratio = nlp - s_oldlp;           // :585
            const float ratio = fast_exp(logratio);         // :586
            float adv = s_adv;
            if (a.hp.norm_adv) adv = (adv - mean_f) * inv_std;           // :593 (reciprocal hoisted: (a - mean) / (std + 1e-8))
            const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
            const float l1 = -adv * ratio, l2 = -adv * rc;  // :597-598
            const bool inside = (ratio >= lo && ratio <= hi_c);
            float d_ratio;
            if (l1 > l2) d_ratio = -adv;
            else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
            else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half/half
            const float g_nlp = invM * d_ratio * ratio;
            const float g_ent = -a.hp.ent_coef * invM;
            if (valid && hi == 0) {
                st0 += (double)(l1 > l2 ? l1 : l2);
                st1 += (double)ent;
                st2 += (double)((ratio - 1.0f) - logratio);
                st3 += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
            }
            off = 0;
#pragma unroll
            for (int h = 0; h < AMAX; h++) {
                if (h >= n_heads) break;
                const int Ah = EXACT ? AMAX : L.head_dims[h];
                const int act_h = act_s[h];
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) {
                    float d = g_nlp * ((k == off + act_h ? 1.0f : 0.0f) - pr[k]);
                    if (DIST == PPO_DIST_MASKED) d += g_ent * (-pr[k] * (z[k] + headH[h]));
                    dOut[k] = (valid && ok[k]) ? d : 0.0f;
                }
                off += Ah;
            }
        }

        MF_STAMP(4);   // head + loss
        // ---------------- h2 -> image; dOut -> [a][s]; then dW3[a][u = lane], db3 ----------------
        store_dlayout(img, h2, s, hi);
        if (hi == 0) {
#pragma unroll
            for (int k = 0; k < AMAX; k++) if (k < AOUT) sDo[k * MT + s] = dOut[k];
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < AMAX; k++) {
            if (k < AOUT) {
                float acc = 0.0f;
#pragma unroll
                for (int c0 = 0; c0 < MT; c0 += 16) {   // 16 image rows + 4 x 16-byte dOut reads in flight together
                    float hv[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) hv[i] = img[(c0 + i) * LS + lane];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        const float4 d = ld4(&sDo[k * MT + c0 + i]);
                        acc = __builtin_fmaf(d.x, hv[i], acc); acc = __builtin_fmaf(d.y, hv[i + 1], acc);
                        acc = __builtin_fmaf(d.z, hv[i + 2], acc); acc = __builtin_fmaf(d.w, hv[i + 3], acc);
                    }
                }
                gW3[k] += acc;
                gb3[k] += hi == 0 ? dOut[k] : 0.0f;   // summed over lanes at the end
            }
        }

        MF_STAMP(5);   // h2 image + dW3
        // ---------------- dz2 = (sum_a dOut[a] W3[a][u]) (1 - h2^2), D layout ----------------
        float dz2[32];
#pragma unroll
        for (int e = 0; e < 32; e++) dz2[e] = 0.0f;
#pragma unroll
        for (int k = 0; k < AMAX; k++) {
            if (k < AOUT) {
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const float4 w = ld4(&sW3[k * 64 + 8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                    dz2[4 * g] = __builtin_fmaf(dOut[k], w.x, dz2[4 * g]); dz2[4 * g + 1] = __builtin_fmaf(dOut[k], w.y, dz2[4 * g + 1]);
                    dz2[4 * g + 2] = __builtin_fmaf(dOut[k], w.z, dz2[4 * g + 2]); dz2[4 * g + 3] = __builtin_fmaf(dOut[k], w.w, dz2[4 * g + 3]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 32; e++) dz2[e] = dz2[e] * (1.0f - h2[e] * h2[e]);
        float dz1[32];
        if constexpr (PREC == PREC_BF16X3) {
            // B operand of d(hidden 1): dz2 in place (D layout), cut into its bf16 terms before the registers are recycled
            uint32_t zp[3][16];
#pragma unroll
            for (int j = 0; j < 16; j++) split3(dz2[2 * j], dz2[2 * j + 1], zp[0][j], zp[1][j], zp[2][j]);
            wave_lds_fence();  // dW3 reads of the image are done
            store_dlayout(img, dz2, s, hi);
            wave_lds_fence();
            // A operand of dW2: dz2[sample 16 c + 8 hi + e][unit s + 32 tn]  (lane index s plays the unit here), fp32 for now
            float opA[2][16];
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int i = 0; i < 16; i++) opA[tn][i] = img[(16 * (i >> 3) + 8 * hi + (i & 7)) * LS + s + 32 * tn];
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {   // db2[n = s + 32 tn] = sum over samples
                float c = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; i++) c += opA[tn][i];
                c += __shfl_xor(c, 32, 64);
                gb2[tn] += c;
            }
            wave_lds_fence();
            store_dlayout(img, h1, s, hi);
            wave_lds_fence();
            MF_STAMP(6);   // dz2, images, opA, db2
            // ---------------- dW2[n][k] += sum_s dz2[s][n] h1[s][k]: two chunks of 16 samples, 24 MFMAs each ----------------
#pragma unroll
            for (int c = 0; c < 2; c++) {
                u32x4 A[2][3], B[2][3];
#pragma unroll
                for (int tk = 0; tk < 2; tk++) {
                    float hb[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) hb[e] = img[(16 * c + 8 * hi + e) * LS + s + 32 * tk];   // h1[sample][k = s + 32 tk]
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t p1, p2, p3;
                        split3(hb[2 * j], hb[2 * j + 1], p1, p2, p3);
                        B[tk][0][j] = p1; B[tk][1][j] = p2; B[tk][2][j] = p3;
                    }
                }
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t p1, p2, p3;
                        split3(opA[tn][8 * c + 2 * j], opA[tn][8 * c + 2 * j + 1], p1, p2, p3);
                        A[tn][0][j] = p1; A[tn][1][j] = p2; A[tn][2][j] = p3;
                    }
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
#pragma unroll
                    for (int tk = 0; tk < 2; tk++)
                        gW2[tn][tk] = mfma_x3(A[tn][0], A[tn][1], A[tn][2], B[tk][0], B[tk][1], B[tk][2], gW2[tn][tk]);
            }
            MF_STAMP(7);   // dW2 MFMA
            // ---------------- dh1^T[k][s] = sum_n W2[n][k] dz2[s][n], dz1 = dh1 (1 - h1^2) ----------------
            // g = 4 t + c: output row k = s + 32 t, contraction chunk c over n.  Fragment element e <-> n = 16 c + 8 (e >> 2) + 4 hi + (e & 3):
            // column k of rows 16 c + 4 hi .. +3 (first transposing read) and of rows 16 c + 8 + 4 hi .. +3 (second)
            const int tq = (lane & 15) >> 2, tp = lane & 3, tkb = 16 * ((lane >> 4) & 1);
            auto afrag = [&](int g, int term) -> u32x4 {
                const uint16_t* q = sW2Tp + term * 64 * NS + (16 * (g & 3) + 4 * hi + tq) * NS + 32 * (g >> 2) + tkb + 4 * tp;
                const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 8 * NS);
                const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                return r;
            };
            f32x16 acc;
            u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1), an3 = afrag(0, 2);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int t = g >> 2, c = g & 3;
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
                }
                const u32x4 a1 = an1, a2 = an2, a3 = an3;
                if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); an3 = afrag(g + 1, 2); }
                MF_PIN();
                const u32x4 b1 = { zp[0][4 * c], zp[0][4 * c + 1], zp[0][4 * c + 2], zp[0][4 * c + 3] };
                const u32x4 b2 = { zp[1][4 * c], zp[1][4 * c + 1], zp[1][4 * c + 2], zp[1][4 * c + 3] };
                const u32x4 b3 = { zp[2][4 * c], zp[2][4 * c + 1], zp[2][4 * c + 2], zp[2][4 * c + 3] };
                acc = mfma_x3(a1, a2, a3, b1, b2, b3, acc);
                if (c == 3) {   // h1 comes back from its [sample][unit] image (still intact: dW2 only read it)
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const float4 hb = ld4(&img[s * LS + 8 * qq + 4 * hi + 32 * t]);
                        dz1[16 * t + 4 * qq + 0] = acc[4 * qq + 0] * (1.0f - hb.x * hb.x);
                        dz1[16 * t + 4 * qq + 1] = acc[4 * qq + 1] * (1.0f - hb.y * hb.y);
                        dz1[16 * t + 4 * qq + 2] = acc[4 * qq + 2] * (1.0f - hb.z * hb.z);
                        dz1[16 * t + 4 * qq + 3] = acc[4 * qq + 3] * (1.0f - hb.w * hb.w);
                    }
                }
            }
        } else {
            wave_lds_fence();  // dW3 reads of the image are done
            store_dlayout(img, dz2, s, hi);
            wave_lds_fence();
            // A operands of dW2: dz2[sample 2 st + hi][unit s + 32 tn]  (lane index s plays the unit here)
            float opA[2][16];
    #pragma unroll
            for (int tn = 0; tn < 2; tn++)
    #pragma unroll
                for (int stp = 0; stp < 16; stp++) opA[tn][stp] = img[(2 * stp + hi) * LS + s + 32 * tn];
            {   // db2[n = s + 32 tn] = sum over samples
    #pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    float c = 0.0f;
    #pragma unroll
                    for (int stp = 0; stp < 16; stp++) c += opA[tn][stp];
                    c += __shfl_xor(c, 32, 64);
                    gb2[tn] += c;
                }
            }
            wave_lds_fence();
            store_dlayout(img, h1, s, hi);
            wave_lds_fence();
            MF_STAMP(6);   // dz2, images, opA, db2
            // ---------------- dW2[n][k] += sum_s dz2[s][n] h1[s][k] (MFMA, contraction over samples) ----------------
            {
                float b0n = img[hi * LS + s], b1n = img[hi * LS + s + 32];   // h1[sample 2 stp + hi][k = s], [k = s + 32]; read one step ahead
    #pragma unroll
                for (int stp = 0; stp < 16; stp++) {
                    const float b0 = b0n, b1 = b1n;
                    if (stp + 1 < 16) { b0n = img[(2 * stp + 2 + hi) * LS + s]; b1n = img[(2 * stp + 2 + hi) * LS + s + 32]; }
                    MF_PIN();
                    gW2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][stp], b0, gW2[0][0], 0, 0, 0);
                    gW2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][stp], b1, gW2[0][1], 0, 0, 0);
                    gW2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[1][stp], b0, gW2[1][0], 0, 0, 0);
                    gW2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[1][stp], b1, gW2[1][1], 0, 0, 0);
                }
            }

            MF_STAMP(7);   // dW2 MFMA
            // ---------------- dh1^T[k][s] = sum_n W2[n][k] dz2[s][n] (MFMA), dz1 = dh1 (1 - h1^2) ----------------
            {
                auto wptr = [&](int g) { return &sW2T[(s + 32 * (g >> 3)) * LS + 8 * (g & 3) + 4 * hi + 32 * ((g >> 2) & 1)]; };  // W2[U(4q.., hi, tn)][k = s + 32 t]
                f32x16 acc;
                float4 wn = ld4(wptr(0));
    #pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int t = g >> 3, tn = (g >> 2) & 1, q = g & 3;
                    if ((g & 7) == 0) {
    #pragma unroll
                        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
                    }
                    const float4 w = wn;
                    if (g + 1 < 16) wn = ld4(wptr(g + 1));
                    MF_PIN();
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, dz2[16 * tn + 4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, dz2[16 * tn + 4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, dz2[16 * tn + 4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, dz2[16 * tn + 4 * q + 3], acc, 0, 0, 0);
                    if ((g & 7) == 7) {
                        // h1 comes back from its [sample][unit] image (still intact: dW2 only read it), so its 32 registers are free
                        // during the dW2 / d(hidden) MFMA phases
    #pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            const float4 hb = ld4(&img[s * LS + 8 * qq + 4 * hi + 32 * t]);
                            dz1[16 * t + 4 * qq + 0] = acc[4 * qq + 0] * (1.0f - hb.x * hb.x);
                            dz1[16 * t + 4 * qq + 1] = acc[4 * qq + 1] * (1.0f - hb.y * hb.y);
                            dz1[16 * t + 4 * qq + 2] = acc[4 * qq + 2] * (1.0f - hb.z * hb.z);
                            dz1[16 * t + 4 * qq + 3] = acc[4 * qq + 3] * (1.0f - hb.w * hb.w);
                        }
                    }
                }
            }
        }
        MF_STAMP(8);   // dh1 MFMA + dz1
        wave_lds_fence();  // dW2's reads of the h1 image are done
        store_dlayout(img, dz1, s, hi);
        wave_lds_fence();
        // ---------------- dW1[u = lane][o] += sum_s dz1[s][u] x[s][o]; db1 ----------------
        {
            float accw[OBS], accb = 0.0f;
#pragma unroll
            for (int o = 0; o < OBS; o++) accw[o] = 0.0f;
#pragma unroll
            for (int c0 = 0; c0 < MT; c0 += 16) {
                float dv[16];
#pragma unroll
                for (int i = 0; i < 16; i++) dv[i] = img[(c0 + i) * LS + lane];
#pragma unroll
                for (int i = 0; i < 16; i++) accb += dv[i];
#pragma unroll
                for (int o = 0; o < OBS; o++)
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        const float4 xv = ld4(&sX[o * MT + c0 + i]);
                        accw[o] = __builtin_fmaf(dv[i], xv.x, accw[o]); accw[o] = __builtin_fmaf(dv[i + 1], xv.y, accw[o]);
                        accw[o] = __builtin_fmaf(dv[i + 2], xv.z, accw[o]); accw[o] = __builtin_fmaf(dv[i + 3], xv.w, accw[o]);
                    }
            }
            gb1 += accb;
#pragma unroll
            for (int o = 0; o < OBS; o++) gW1[o] += accw[o];
        }
        wave_lds_fence();  // image and sX are rewritten by the next tile
        MF_STAMP(9);   // dz1 image + dW1
    }

    if constexpr (STAMP) {   // slot 11: tile-loop cycles of wave 4 (wave 0's SIMD partner), to compare with wave 0's phases 1..9
        if (blk == 0 && tid == (MF_WAVES > 4 ? 256 : 64) && a.stamps) a.stamps[NET * 12 + 11] += __builtin_amdgcn_s_memtime() - t_prev0;
    }
    // ---------------- the waves park their accumulators in private LDS regions (plain stores), then every thread adds the
    //                  regions in a fixed order into the workgroup's slab ----------------
    // loss sums first: registers only, and their landing place lies beyond every live byte of LDS, so this needs no barrier
    st0 = wave_sum_d_dpp(st0); st1 = wave_sum_d_dpp(st1); st2 = wave_sum_d_dpp(st2); st3 = wave_sum_d_dpp(st3);
    const int base = L.net_off[NET];
    const int nsz = L.net_size[NET];
    const int rstride = (nsz + 3) & ~3;
    int dred_off = MF_WAVES * rstride + 4;
    if (dred_off < m.total) dred_off = m.total;
    dred_off = (dred_off + 1) & ~1;
    double* dred = reinterpret_cast<double*>(smem + dred_off);
    if (lane == 0) { dred[wave * 4 + 0] = st0; dred[wave * 4 + 1] = st1; dred[wave * 4 + 2] = st2; dred[wave * 4 + 3] = st3; }
    __syncthreads();   // weights and images are dead: the rest of the dynamic LDS block is reused
    float* red = smem + wave * rstride;
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
        for (int tk = 0; tk < 2; tk++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[L.w2[NET] - base + umap(r, hi, tn) * 64 + s + 32 * tk] = gW2[tn][tk][r];
#pragma unroll
    for (int k = 0; k < AMAX; k++) if (k < AOUT) red[L.w3[NET] - base + k * 64 + lane] = gW3[k];
#pragma unroll
    for (int o = 0; o < OBS; o++) red[L.w1[NET] - base + lane * OBS + o] = gW1[o];
    red[L.b1[NET] - base + lane] = gb1;
    if (hi == 0) { red[L.b2[NET] - base + s] = gb2[0]; red[L.b2[NET] - base + s + 32] = gb2[1]; }
#pragma unroll
    for (int k = 0; k < AMAX; k++) if (k < AOUT) {
        const float t = wave_sum(gb3[k]);
        if (lane == 0) red[L.b3[NET] - base + k] = t;
    }
    __syncthreads();
    const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
    float* slab = a.slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blk) * Pmax;
    for (int e = tid; e < nsz; e += MF_THREADS) {
        float t = smem[e];
#pragma unroll
        for (int w = 1; w < MF_WAVES; w++) t += smem[w * rstride + e];   // fixed order
        slab[e] = t;
    }
    if (tid < 4) {
        double* o = a.stat_slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blk) * 8;
        double t = dred[tid];
#pragma unroll
        for (int w = 1; w < MF_WAVES; w++) t += dred[4 * w + tid];
        o[tid] = t;
    }
    MF_STAMP(10);  // epilogue
    if constexpr (STAMP) {
        if (blk == 0 && tid == 0 && a.stamps) {
            for (int i = 0; i < 11; i++) a.stamps[NET * 12 + i] += ph[i];
        }
    }
}

template <int DIST, int OBS, int AMAX, bool EXACT, int PREC, bool STAMP>
__global__ __launch_bounds__(64 * mf_waves(PREC), PREC == PREC_BF16X3 ? 1 : 2) void fwd_bwd_mfma_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 1-D grid: the first n_blocks[0] workgroups run the critic, the rest the actor
    const int b = blockIdx.x;
    if (b < a.n_blocks[0]) mf_body<0, DIST, OBS, 1, true, PREC, STAMP>(a, smem, b, a.n_blocks[0]);
    else mf_body<1, DIST, OBS, AMAX, EXACT, PREC, STAMP>(a, smem, b - a.n_blocks[0], a.n_blocks[1]);
}

// ---------------------------------------------------------------------------------------------------------
// Critic over rows [0, n0) of obs0 and [0, n1) of obs1 on the matrix cores (m_values[step] = Critic(obs[step]),
// PPO_Discrete.cpp:534-536, and the bootstrap value, :280): the forward half of mf_body<0> -- one wave per 32-row tile, layer 1 as
// fp32 MFMA, layer 2 as three-term bf16 products, head as a 32-term dot product per half-lane.  Rows are contiguous, so the
// "gather" is one coalesced 16-byte load per row.
// ---------------------------------------------------------------------------------------------------------
template <int OBS>
__global__ __launch_bounds__(256, 2) void values_mfma_kernel(const float* __restrict__ P, NetLayout L, const float* __restrict__ obs0, int64_t n0,
                                                             float* __restrict__ out0, const float* __restrict__ obs1, int64_t n1,
                                                             float* __restrict__ out1) {
    __shared__ __attribute__((aligned(16))) uint16_t sW2p[3 * 64 * WS];
    __shared__ __attribute__((aligned(16))) float sB1[64], sB2[64], sW3[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, hi = lane >> 5;
    {
        float wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wv[i] = P[L.w2[0] + tid + i * 256];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + i * 256;
            const int n = e >> 6, k = e & 63;
            const uint32_t u0 = f2u(wv[i]);
            const float r1 = wv[i] - u2f(u0 & 0xffff0000u);
            const uint32_t u1 = f2u(r1);
            const float r2 = r1 - u2f(u1 & 0xffff0000u);
            const int pf = n * WS + slot_of_unit(k);
            sW2p[pf] = (uint16_t)(u0 >> 16); sW2p[64 * WS + pf] = (uint16_t)(u1 >> 16); sW2p[2 * 64 * WS + pf] = (uint16_t)(f2u(r2) >> 16);
        }
    }
    if (tid < 64) { sB1[tid] = P[L.b1[0] + tid]; sB2[tid] = P[L.b2[0] + tid]; sW3[tid] = P[L.w3[0] + tid]; }
    const float b3 = P[L.b3[0]];
    constexpr int L1S = (OBS + 1) / 2;
    float w1op[2][L1S];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int stp = 0; stp < L1S; stp++) {
            const int o = 2 * stp + hi;
            w1op[t][stp] = o < OBS ? P[L.w1[0] + (s + 32 * t) * OBS + o] : 0.0f;
        }
    __syncthreads();
    const int64_t n = n0 + n1;
    const int64_t n_tiles = (n + MT - 1) / MT;
    // the observation row of a tile is requested one tile ahead (unconditional loads: a missing row reads row 0 and is zeroed at use)
    auto load_x = [&](int64_t tl, float* xo) {
        const int64_t r = tl * MT + s;
        const bool ok = r < n;
        const float* src = !ok ? obs0 : (r < n0 ? obs0 + r * OBS : obs1 + (r - n0) * OBS);
        if constexpr (OBS == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            xo[0] = v.x; xo[1] = v.y; xo[2] = v.z; xo[3] = v.w;
        } else if constexpr (OBS == 2) {
            const float2 v = *reinterpret_cast<const float2*>(src);
            xo[0] = v.x; xo[1] = v.y;
        } else {
#pragma unroll
            for (int o = 0; o < OBS; o++) xo[o] = src[o];
        }
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    float x_n[OBS];
    load_x((int64_t)blockIdx.x * 4 + wave, x_n);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += tile_step) {
        const int64_t row = tile * MT + s;
        const bool valid = row < n;
        float x[OBS];
#pragma unroll
        for (int o = 0; o < OBS; o++) x[o] = valid ? x_n[o] : 0.0f;
        load_x(tile + tile_step, x_n);
        float h1[32];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 b = ld4(&sB1[8 * q + 4 * hi + 32 * t]);
                acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
#pragma unroll
            for (int stp = 0; stp < L1S; stp++) {
                const float xb = (2 * stp + 1 < OBS) ? (hi ? x[2 * stp + 1] : x[2 * stp]) : (hi ? 0.0f : x[2 * stp]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1op[t][stp], xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) h1[16 * t + r] = tanh_mufu(acc[r]);
        }
        uint32_t hp[3][16];
#pragma unroll
        for (int j = 0; j < 16; j++) split3(h1[2 * j], h1[2 * j + 1], hp[0][j], hp[1][j], hp[2][j]);
        auto aptr = [&](int g, int term) {
            return reinterpret_cast<const u32x4*>(sW2p + term * 64 * WS + (s + 32 * (g >> 2)) * WS + (g & 3) * 16 + hi * 8);
        };
        float part = 0.0f;
        f32x16 acc;
        u32x4 an1 = *aptr(0, 0), an2 = *aptr(0, 1), an3 = *aptr(0, 2);
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int t = g >> 2, c = g & 3;
            if (c == 0) {
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const float4 b = ld4(&sB2[8 * qq + 4 * hi + 32 * t]);
                    acc[4 * qq] = b.x; acc[4 * qq + 1] = b.y; acc[4 * qq + 2] = b.z; acc[4 * qq + 3] = b.w;
                }
            }
            const u32x4 a1 = an1, a2 = an2, a3 = an3;
            if (g + 1 < 8) { an1 = *aptr(g + 1, 0); an2 = *aptr(g + 1, 1); an3 = *aptr(g + 1, 2); }
            MF_PIN();
            const u32x4 b1 = { hp[0][4 * c], hp[0][4 * c + 1], hp[0][4 * c + 2], hp[0][4 * c + 3] };
            const u32x4 b2 = { hp[1][4 * c], hp[1][4 * c + 1], hp[1][4 * c + 2], hp[1][4 * c + 3] };
            const u32x4 b3v = { hp[2][4 * c], hp[2][4 * c + 1], hp[2][4 * c + 2], hp[2][4 * c + 3] };
            acc = mfma_x3(a1, a2, a3, b1, b2, b3v, acc);
            if (c == 3) {
#pragma unroll
                for (int q = 0; q < 4; q++) {   // units 8q + 4hi + 32t .. +3 are registers 4q .. 4q+3
                    const float4 w = ld4(&sW3[8 * q + 4 * hi + 32 * t]);
                    part = __builtin_fmaf(tanh_mufu(acc[4 * q]), w.x, part); part = __builtin_fmaf(tanh_mufu(acc[4 * q + 1]), w.y, part);
                    part = __builtin_fmaf(tanh_mufu(acc[4 * q + 2]), w.z, part); part = __builtin_fmaf(tanh_mufu(acc[4 * q + 3]), w.w, part);
                }
            }
        }
        const float other = __shfl_xor(part, 32, 64);
        const float v = (part + other) + b3;
        if (valid && hi == 0) { if (row < n0) out0[row] = v; else out1[row - n0] = v; }
    }
}

}  // namespace

// Both flavours keep two waves on every SIMD: PREC_F32 as two 4-wave workgroups per CU, PREC_BF16X3 (whose split weight images
// need 55 KB) as one 8-wave workgroup per CU.
void update_blocks_mfma(int M, double actor_share, int prec, int n_blocks[2]) {
    const int waves = mf_waves(prec);
    const int total = prec == PREC_BF16X3 ? 256 : 512;           // resident workgroups of the chip
    const int tiles = (M + MT - 1) / MT;
    const int per_net = (tiles + waves - 1) / waves;             // workgroups that still get a tile per wave
    if (per_net <= total / 2) { n_blocks[0] = n_blocks[1] = per_net > 0 ? per_net : 1; return; }
    int na = (int)(total * actor_share + 0.5);
    na = na < total / 8 ? total / 8 : (na > total - total / 8 ? total - total / 8 : na);
    n_blocks[1] = na;
    n_blocks[0] = total - na;
}

hipError_t launch_minibatch_fwd_bwd_mfma(const UpdateArgs& a, int prec, hipStream_t s) {
    if (a.M <= 0) return hipErrorInvalidValue;
    if (a.L.act > 4) return hipErrorNotSupported;  // wider heads run on the VALU kernel
    const int aout = a.L.act > 1 ? a.L.act : 1;
    const int waves = mf_waves(prec);
    size_t shmem = (size_t)mf_smem(a.L.obs, aout, prec).total * sizeof(float);
    {   // the epilogue parks one gradient image per wave (+ 4 doubles of loss sums each) in the same block
        const int nmax = a.L.net_size[0] > a.L.net_size[1] ? a.L.net_size[0] : a.L.net_size[1];
        size_t dred_off = (size_t)waves * ((nmax + 3) & ~3) + 4;
        if (dred_off < (size_t)mf_smem(a.L.obs, aout, prec).total) dred_off = (size_t)mf_smem(a.L.obs, aout, prec).total;
        const size_t need = (dred_off + 2 + 8 * waves + 8) * sizeof(float);
        if (need > shmem) shmem = need;
    }
    if (shmem > 160 * 1024) return hipErrorNotSupported;
    const dim3 grid((unsigned)(a.n_blocks[0] + a.n_blocks[1])), block(64 * waves);
#define PPO_LAUNCH_MF2(DIST, OBS, AMAX, EXACT, PREC)                                                                   \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, PREC, false>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
            if (e == hipSuccess)                                                                                       \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, PREC, true>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
            if (e != hipSuccess) return e;                                                                             \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        if (a.stamps) hipLaunchKernelGGL((fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, PREC, true>), grid, block, shmem, s, a); \
        else hipLaunchKernelGGL((fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, PREC, false>), grid, block, shmem, s, a); \
    } while (0)
#define PPO_LAUNCH_MF(DIST, OBS, AMAX, EXACT)                                                                          \
    do {                                                                                                               \
        if (prec == PREC_BF16X3) PPO_LAUNCH_MF2(DIST, OBS, AMAX, EXACT, PREC_BF16X3);                                  \
        else PPO_LAUNCH_MF2(DIST, OBS, AMAX, EXACT, PREC_F32);                                                         \
    } while (0)
    // the reference's two shapes get fully folded head code: CartPole (obs 4, one head of 2) and MountainCar (obs 2, one masked
    // head of 3); anything else with <= 4 logits runs the generic variant
    const bool one = a.L.n_heads == 1;
    if (a.L.obs == 4) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) {
            if (one && a.L.act == 2) PPO_LAUNCH_MF(PPO_DIST_CATEGORICAL, 4, 2, true); else PPO_LAUNCH_MF(PPO_DIST_CATEGORICAL, 4, 4, false);
        } else {
            PPO_LAUNCH_MF(PPO_DIST_MASKED, 4, 4, false);
        }
    } else if (a.L.obs == 2) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) {
            PPO_LAUNCH_MF(PPO_DIST_CATEGORICAL, 2, 4, false);
        } else {
            if (one && a.L.act == 3) PPO_LAUNCH_MF(PPO_DIST_MASKED, 2, 3, true); else PPO_LAUNCH_MF(PPO_DIST_MASKED, 2, 4, false);
        }
    } else {
        return hipErrorNotSupported;
    }
#undef PPO_LAUNCH_MF
#undef PPO_LAUNCH_MF2
    return hipGetLastError();
}

hipError_t launch_values_mfma(const float* params, const NetLayout& L, const float* obs0, int64_t n0, float* out0, const float* obs1, int64_t n1,
                              float* out1, hipStream_t s) {
    const int64_t n = n0 + n1;
    if (n <= 0) return hipSuccess;
    const int64_t wg_needed = ((n + MT - 1) / MT + 3) / 4;
    const unsigned grid = (unsigned)(wg_needed < 1024 ? wg_needed : 1024);
    if (L.obs == 4) hipLaunchKernelGGL((values_mfma_kernel<4>), dim3(grid), dim3(256), 0, s, params, L, obs0, n0, out0, obs1, n1, out1);
    else if (L.obs == 2) hipLaunchKernelGGL((values_mfma_kernel<2>), dim3(grid), dim3(256), 0, s, params, L, obs0, n0, out0, obs1, n1, out1);
    else return hipErrorNotSupported;
    return hipGetLastError();
}
