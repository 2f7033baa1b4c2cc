// ppo-libtorch_amd/csrc/kernels_update_mfma.hip -- K5-K8 (gather + forward + PPO loss + backward) on the CDNA4 matrix cores.
//
// Same contract as fwd_bwd_kernel in kernels_update.hip (reference PPO_Discrete.cpp:576-638), different machine mapping:
// the three 64x64 contractions per sample and net -- layer-2 forward, d(hidden 1), and the weight gradient dW2 -- run on the
// matrix cores as f16 MFMAs over two-term fp16 splits of the fp32 operands (see "arithmetic" below).  Matrix cores are used
// only here because only here is the minibatch (131 072 rows at BASELINE configs[1]) a real contraction.  values_mfma_kernel at
// the end of the file is the forward half of the same mapping for the batched critic evaluation of the rollout.
//
// One WAVE owns a 32-sample tile of one net from gather to weight gradient; a workgroup is eight such waves of the same net
// (two per SIMD: one wave's MFMAs run beside its partner's vector work) sharing LDS copies of the weights.  Register layout of
// every hidden vector is the MFMA C/D layout
//     lane (s = lane & 31, hi = lane >> 5), element e = r + 16 t   <->   unit  U(r, hi, t) = (r & 3) + 8 (r >> 2) + 4 hi + 32 t
// i.e. lane = sample, registers = 32 of the 64 units.  Because the contraction index of an MFMA may be enumerated in any
// order as long as A and B agree, a D-layout vector is directly the B operand of the next product when the weight operand is
// fetched in that same order: forward and d(hidden) need NO data movement.
// Only the products that contract over SAMPLES (dW2, dW3, dW1, bias gradients) need lane = unit: the tile goes through a
// private LDS region of the wave -- h2, h1 and dz1 as 32 x 68-float images, dz2 as the two fp16 TERM images its d(hidden)
// product has already formed (read back transposed by ds_read_b64_tr_b16: no second split of dz2).
// Weight-gradient accumulators (64 registers for dW2) live in registers across all tiles of the wave; the waves of a
// workgroup are then added in a fixed order through LDS and leave as ONE partial slab (deterministic, no float atomics).
//
// Arithmetic: every fp32 operand x is carried as TWO fp16 terms, t1 = rn16(x), t2 = rn16(x - t1) (round to nearest even; x - t1 is
// exact: one v_fma_mix_f32 with t1 read straight out of its packed pair), and every fp32 product a.b is issued as the three f16
// products a1b1 + a1b2 + a2b1 on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  |x - t1 - t2| <= 2^-23 |x| (the residual x - t1 has up to 13
// significant bits and fp16 keeps 11) and the dropped a2b2 is <= 2^-22 |a||b| (|t2| <= 2^-11 |x|), so a product is off by about 5 x 2^-24 |a||b|:
// a few fp32 roundings.  fp16 has 5 exponent bits
// where fp32 has 8; the range is handled per operand:
//   * h1 (|h| <= 1) and c W2 are used as they are: below 2^-2 the second term runs into fp16's denormals and the representation error
//     becomes absolute, <= 2^-25 -- a quarter of the rounding step of an fp32 number near 1, and h, W enter sums of O(1) terms.
//     |c W2| must stay below 65504 (|W2| < 22 700); beyond that the conversion yields inf and the update fails loudly (NaN).
//   * dz2 (~1 / minibatch size, times anything the advantages and returns carry) is formed already multiplied by 2^S, S a wave-uniform
//     integer chosen from the tile's own data so that no |dz2| 2^S can pass 2^14 (see the tile loop); everything downstream of dz2
//     carries the factor and the gradient accumulators lose it, exactly, in the epilogue.  Elements down to 2^-15 of the tile's bound
//     keep the full 24 bits, smaller ones an absolute error of 2^-39 of it.
// Per 32-sample tile and net: 72 f16 MFMAs (24 per contraction) + 4 fp32 ones for layer 1, and 4 vector instructions per pair of
// values for a split (two conversions that also pack, two residuals).  Round 1-2's bf16 form of the same idea (three truncated bf16
// terms, six products, 5.5 vector instructions per value) measured 52.7 us per launch against this form's 44.0, A/B in one call.
// The exact-fp32 instruction v_mfma_f32_32x32x2_f32 runs at the vector rate and excludes every other vector instruction of its SIMD
// while it executes (tools/probes/mfma_coexec.hip); it is used only for layer 1 (K = obs_size <= 4).
//
// tanh(z) = 1 - 2 / (1 + 2^(c z)), c = 2 log2(e).  c is folded into the LDS copies of W1, b1, W2, b2, so a pre-activation leaves the
// matrix cores already scaled and tanh is four instructions (v_exp, v_add, v_rcp, v_fma).  The backward product through the scaled W2
// yields c dh1, hence c dW1 and c db1: those two accumulators are multiplied by 1 / c once, in the epilogue.
#include "ppo_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 32;              // samples per wave tile
constexpr int LS = 68;              // padded row stride (floats) of an fp32 [sample][unit] image
constexpr int NS = 68;              // fp16 per row of a term image (136 B: 32 rows x 8 B land on 64 distinct banks)
constexpr int WS = 72;              // fp16 per padded row of values_mfma_kernel's weight operand (144 B: 16 rows x 16 B land on 16 distinct bank groups)
constexpr int MF_WAVES = 8, MF_THREADS = 64 * MF_WAVES;
constexpr float TANH_C = 2.885390081777927f;        // 2 log2(e)
constexpr float TANH_C_INV = 0.34657359027997264f;  // ln(2) / 2

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
// two floats -> one dword of fp16 (low half = x0), round to nearest even: v_cvt_pk_f16_f32, formed BY THE COMPILER from the vector conversion so
// that its hazard recognizer sees the producer of an MFMA operand: a VGPR written by a VALU instruction needs 2 wait states before an MFMA
// reads it as A/B, hipcc pads that pair only when both instructions are its own, and an MFMA that is not stalled on an accumulator chain reads
// the OLD register otherwise (seen in rollout16_kernel: logits off by 1e-3 with the conversion in an asm string; tools/check_asm_hazards.py
// scans the device assembly for the pair).  The residuals below stay asm (LLVM does not form v_fma_mix_f32 from x - (float)h): they never
// feed an MFMA directly.
typedef float f32x2_cv __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_f16(float x0, float x1) {
    const f32x2_cv x = { x0, x1 };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
}
// x - (float)half(p): the residual of a rounding to fp16 is a float, so this fused multiply-add with an fp16 source is exact
__device__ __forceinline__ float resid_lo(uint32_t p, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float resid_hi(uint32_t p, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
// (x0, x1) -> two dwords of packed fp16 pairs (low half = x0's term): x = t1 + t2 + d, |d| <= max(2^-23 |x|, 2^-25) for |x| < 65520
// (four instructions per pair: the conversion packs, the residual reads its half of the pair directly)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p1, uint32_t& p2) {
    p1 = pk_f16(x0, x1);
    p2 = pk_f16(resid_lo(p1, x0), resid_hi(p1, x1));
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4 a, const u32x4 b, const f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// the three products of one (A chunk, B chunk) pair, small terms first
__device__ __forceinline__ f32x16 mfma_x2(const u32x4 a1, const u32x4 a2, const u32x4 b1, const u32x4 b2, f32x16 acc) {
    acc = mfma_f16(a2, b1, acc);
    acc = mfma_f16(a1, b2, acc);
    acc = mfma_f16(a1, b1, acc);
    return acc;
}
// v_mfma_f32_16x16x32_f16: A lane (i = lane & 15, kg = lane >> 4) holds k = 8 kg .. 8 kg + 7 of row i, B the same for column j = lane & 15,
// D lane (j = lane & 15, kg) holds rows 4 kg .. 4 kg + 3 of column j
__device__ __forceinline__ f32x4 mfma16_f16(const u32x4 a, const u32x4 b, const f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
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
__device__ __forceinline__ float uniform_f(float v) { return u2f(__builtin_amdgcn_readfirstlane(f2u(v))); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct MfSmem {
    int w2, w1, b1, b2, w3, b3, wave0, wave_stride, s_reg, s_x, s_do, total;  // offsets in floats
};
// private region of a wave: one fp32 [sample][unit] image (h2, h1, dz1 in turn) or the two fp16 term images of dz2
constexpr int MF_REGION = (2 * MT * NS / 2) > (MT * LS) ? (2 * MT * NS / 2) : (MT * LS);
__host__ __device__ inline MfSmem mf_smem(int obs, int aout) {
    MfSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    // ONE image per term of c W2, natural [n][k] order: the forward product reads rows (two 8-byte pieces per fragment), the backward
    // product reads the same bytes column-wise with the transposing LDS read ds_read_b64_tr_b16
    m.w2 = take(2 * 64 * NS / 2);
    m.w1 = take(64 * obs);
    m.b1 = take(64);
    m.b2 = take(64);
    m.w3 = take(aout * 64);
    m.b3 = take(aout);
    m.wave0 = o;
    int w = 0;
    auto takew = [&](int n) { int r = w; w += (n + 3) & ~3; return r; };
    m.s_reg = takew(MF_REGION);
    m.s_x = takew(obs * MT);    // [o][sample]
    m.s_do = takew(aout * MT);  // [a][sample]
    m.wave_stride = w;
    m.total = o + MF_WAVES * w;
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
// tanh of a pre-activation that arrives multiplied by c = 2 log2(e): 1 - 2 / (1 + 2^(c z)); saturates correctly (2^x = inf -> 1, 0 -> -1)
__device__ __forceinline__ float tanh_scaled(float cz) {
    const float e = __builtin_amdgcn_exp2f(cz);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}
// (Packed fp32 arithmetic -- v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 for the additions, fmas and products around the two transcendentals and in
// the (1 - h^2) factors, 229 scalar instructions of a tile as 133 packed ones -- was built and measured in round 4: + 1.0 us per launch.  Not used.)
// acc + lo(p) + hi(p) for a packed fp16 pair (v_dot2c_f32_f16 against (1, 1))
__device__ __forceinline__ float add_pair(uint32_t p, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, p), __builtin_bit_cast(f16x2, 0x3c003c00u), acc, false);
}

// largest of the lanes' non-negative floats (compared as integers: NaN patterns rank above infinity and win), on the DPP network
#define MF_DPP_MAXU(v, CTRL) max((v), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (CTRL), 0xf, 0xf, false))
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
    v = MF_DPP_MAXU(v, 0xB1);    // quad_perm:[1,0,3,2]
    v = MF_DPP_MAXU(v, 0x4E);    // quad_perm:[2,3,0,1]
    v = MF_DPP_MAXU(v, 0x141);   // row_half_mirror
    v = MF_DPP_MAXU(v, 0x140);   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ float pow2i(int e) { return u2f((uint32_t)(127 + e) << 23); }   // 2^e, -126 <= e <= 127

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

// DW1M: the layer-1 weight gradient dW1[u][o] = sum_s dz1[s][u] x[s][o] (and db1, as the column of a constant 1) on the matrix cores too
// (v_mfma_f32_16x16x32_f16: K = the tile's 32 samples in ONE instruction per 16 units and term product).  As vector work it was the phase that
// read the most LDS of the tile -- 32 broadcast 16-byte reads of the observations beside the 32 reads of the dz1 image -- and 160 vector
// instructions; now dz1 is cut into fp16 terms (64 instructions), leaves as two [sample][unit] term images like dz2 and comes back through 16
// transposing reads.  dz1 carries dz2's factor 2^S times at most sum_n |c W2[n][k]|; a fixed 2^-7 keeps its terms inside fp16 (bound 2^14 x
// 64 x |c W2| < 2^21 for |W2| < 0.7; larger weights saturate to inf and the update fails loudly like an oversized c W2 does).  Used where the
// register file has room for the 16 accumulator registers (the reference's two shapes); the generic variants keep the vector form.
template <int NET, int DIST, int OBS, int AMAX, bool EXACT, bool STAMP, bool DW1M>
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
    const MfSmem m = mf_smem(OBS, AOUT);
    const uint16_t* sW2p = reinterpret_cast<const uint16_t*>(smem + m.w2);
    float* sW1 = smem + m.w1;
    float* sB1 = smem + m.b1;
    float* sB2 = smem + m.b2;
    float* sW3 = smem + m.w3;
    float* sB3 = smem + m.b3;
    float* wbase = smem + m.wave0 + wave * m.wave_stride;
    float* img = wbase + m.s_reg;
    uint16_t* zimg = reinterpret_cast<uint16_t*>(img);   // the same bytes as two fp16 term images [term][sample][NS]
    float* sX = wbase + m.s_x;
    float* sDo = wbase + m.s_do;
    const float* __restrict__ P = a.params;

    // ---- the first tile's batch rows: requested before anything else (see the gather pipeline below) ----
    const int n_tiles = (a.M + MT - 1) / MT;
    const int tile_step = nblk * MF_WAVES;
    int tile = blk * MF_WAVES + wave;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(NET == 0 ? a.rec_critic : a.rec_actor);
    auto fetch_row = [&](int tl) -> int {
        const int j = tl * MT + s;
        const bool ok = tl < n_tiles && j < a.M;
        const int v = a.idx[ok ? j : 0];
        return ok ? v : -1;
    };
    int row_n = fetch_row(tile);
    int row_nn = fetch_row(tile + tile_step);

    // ---- weights of this net -> LDS (once per launch) ----
    // Every element this thread brings in is a slot: 4096 / threads of W2, then its share of W1, W3, b1, b2, b3.  ALL loads are issued
    // before anything is used: one memory round trip for the prologue.  W1, b1, W2, b2 are stored multiplied by c (tanh_scaled).
    constexpr int NW2 = 4096 / MF_THREADS;
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
    float wv[NS_];
#pragma unroll
    for (int i = 0; i < NS_; i++) wv[i] = P[se[i] < 0 ? 0 : se[i]];
    float4 x_n = rec[4 * (size_t)(row_n < 0 ? 0 : row_n)];   // behind the weight loads: its wait for the index overlaps theirs.  Row 0 stands in for a missing row; zeroed at the point of use
    {
        // 4096 weights, 8 per thread: each is scaled, cut into its two fp16 terms once per launch and stored at its natural [n][k] place
        uint16_t* wn = reinterpret_cast<uint16_t*>(smem + m.w2);
#pragma unroll
        for (int i = 0; i < NW2; i++) {
            const int e = tid + i * MF_THREADS;
            const int n = e >> 6, k = e & 63;
            const float w = wv[i] * TANH_C;
            uint32_t p1, p2;
            split2(w, 0.0f, p1, p2);
            const int pn = n * NS + k;
            wn[pn] = (uint16_t)p1; wn[64 * NS + pn] = (uint16_t)p2;
        }
    }
#pragma unroll
    for (int i = 0; i < NW1; i++) { const int e = tid + i * MF_THREADS; if (e < 64 * OBS) sW1[e] = wv[NW2 + i] * TANH_C; }
#pragma unroll
    for (int i = 0; i < NW3; i++) { const int e = tid + i * MF_THREADS; if (e < AOUT * 64) sW3[e] = wv[NW2 + NW1 + i]; }
    if (tid < 64) { sB1[tid] = wv[NS_ - 3] * TANH_C; sB2[tid] = wv[NS_ - 2] * TANH_C; }
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
    f32x4 gW1m[4];                 // DW1M: [unit block b]: lane (j, kg) holds dW1[16 b + 4 kg + r][o = j] (j < OBS) and db1[16 b + 4 kg + r] (j == OBS)
#pragma unroll
    for (int k = 0; k < AMAX; k++) gW3[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < OBS; k++) gW1[k] = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) gW1m[b][r] = 0.0f;
    float gb1 = 0.0f, gb2[2] = { 0.0f, 0.0f }, gb3[AMAX];   // gb2: lane (n, hi) holds the sum over ITS half of the samples
#pragma unroll
    for (int k = 0; k < AMAX; k++) gb3[k] = 0.0f;
    // loss sums: a lane adds a handful of samples (tiles per wave) in fp32; lanes, waves and workgroups are then added in binary64
    float st0 = 0.0f, st1 = 0.0f, st2 = 0.0f, st3 = 0.0f;
    // wave-uniform scalars live in scalar registers (the vector file is the scarce resource of this kernel)
    const float clip = a.hp.clip_coef;
    const float lo = uniform_f(1 - clip), hi_c = uniform_f(1 + clip);
    const float invM = uniform_f((float)a.inv_global_M);
    // mean and 1 / (Bessel std + 1e-8) of the minibatch's advantages: two scalars formed once per update (adv_norm_kernel), one 16-byte scalar load here
    float mean_f = 0.0f, inv_std = 0.0f;
    if (NET == 1 && a.hp.norm_adv) {
        const float4 an = *a.adv_norm;
        mean_f = uniform_f(an.x);
        inv_std = uniform_f(an.y);
    }
    __syncthreads();
    // Range of the fp16 terms of dz2 (gradients are ~1 / M): dz2 is formed already multiplied by 2^S, S a wave-uniform integer kept
    // such that the tile's largest possible |dz2| 2^S -- (sum_a |dOut[a]|) max|W3|, since |1 - h^2| <= 1 -- stays below 2^14.  S only
    // ever moves down, and when it does the gradient accumulators that carry the factor (dW2, db2, dW1, db1) move with it; the
    // factor leaves in the epilogue.  Powers of two: every step is exact.
    float w3max;
    {
        uint32_t mb = 0u;
#pragma unroll
        for (int k = 0; k < AMAX; k++) if (k < AOUT) mb = max(mb, f2u(fabsf(sW3[k * 64 + lane])));
        w3max = u2f(wave_max_u(mb));
    }
    int S_w = 100;
    float scaleS = pow2i(S_w);
    constexpr int L1S = (OBS + 1) / 2;   // K = OBS of layer 1 is contracted in ceil(OBS / 2) fp32 MFMA steps

    MF_STAMP(0);   // prologue: weights -> LDS

    // Gather (K5) from the packed sample records (pack_records_kernel): ONE 32-byte record per sample and net, so a permuted row
    // costs one memory line instead of one per field.  Software pipeline: batch-row indices are fetched TWO tiles ahead and the
    // record's first half (the observation) ONE tile ahead, so neither of the two dependent round trips is ever waited for inside a tile;
    // the second half (loss scalars) is requested at the top of its own tile and first used after layer 2.
    // Every load of the pipeline is UNCONDITIONAL (clamped address, value selected afterwards): a load inside a branch makes the
    // compiler's wait-count bookkeeping give up at the join and emit s_waitcnt vmcnt(0) at the next use of ANY loaded value.
    // (The first two index fetches and the first record fetch are issued at the very TOP of the kernel, in front of the weight staging: the two dependent
    // round trips of the first tile then run under the prologue instead of in front of the first tile -- 42.5 -> 41.9 us per launch, A/B in one call.)
    const int wave_half = __builtin_amdgcn_readfirstlane(wave >> 2) & 1;   // SIMD partners are waves w and w + 4
    for (int it = 0; tile < n_tiles; tile += tile_step, it++) {
        // issue priority alternates between the two waves of a SIMD tile by tile: with equal priorities the older wave wins every
        // arbitration and finishes its tiles ~25% earlier, leaving its partner to run the tail alone
        // (measured again in round 2, A/B in one call: no alternation 55.5 us per launch, a fixed priority for waves 4-7 54.7, this 53.2;
        // starting waves 4-7 2k-8k cycles late changes nothing: the partners do not run in lockstep)
        if ((it ^ wave_half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        // ---------------- gather (K5): lanes (s, 0) and (s, 1) read the same batch row ----------------
        const bool valid = row_n >= 0;
        const int row = valid ? row_n : 0;
        float x[4] = { valid ? x_n.x : 0.0f, valid ? x_n.y : 0.0f, valid ? x_n.z : 0.0f, valid ? x_n.w : 0.0f };
        // per-sample scalars of this tile: issued now, consumed at the loss.  critic: {return, old value, -, -}; actor: {old log-prob,
        // advantage, actions (8 bits per head), mask bits}
        const float4 sc = rec[4 * (size_t)row + 1];
        float s_f0 = sc.x, s_f1 = sc.y;
        uint32_t s_actbits = f2u(sc.z), s_maskbits = f2u(sc.w);
        {   // next tile's observation (its index arrived a tile ago); the index after that
            row_n = row_nn;
            x_n = rec[4 * (size_t)(row_n < 0 ? 0 : row_n)];
            row_nn = fetch_row(tile + 2 * tile_step);
        }
        if (hi == 0) {
#pragma unroll
            for (int o = 0; o < OBS; o++) sX[o * MT + s] = x[o];
        }

        MF_STAMP(1);   // gather hand-over + prefetch issue
        // ---------------- layer 1 (MFMA, K = OBS): c z1^T[u][s] = c b1[u] + sum_o c W1[u][o] x[s][o], D layout ----------------
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
                // A operand: c W1[u = s + 32 t][o = 2 stp + hi] (zero beyond OBS), re-read from LDS every tile: four registers less to carry
                const int o = 2 * stp + hi;
                const float wa = o < OBS ? sW1[(s + 32 * t) * OBS + o] : 0.0f;
                const float xb = (2 * stp + 1 < OBS) ? (hi ? x[2 * stp + 1] : x[2 * stp]) : (hi ? 0.0f : x[2 * stp]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) h1[16 * t + r] = tanh_scaled(acc[r]);
        }

        MF_STAMP(2);   // layer 1 + tanh
        // ---------------- layer 2 forward (MFMA): c z2^T[n][s] = c b2[n] + sum_k c W2[n][k] h1[s][k] ----------------
        float h2[32];
        {
            // B operand = h1 itself (D layout): chunk c of the contraction is registers 8c .. 8c+7, cut into fp16 terms in place
            uint32_t hp[2][16];
#pragma unroll
            for (int j = 0; j < 16; j++) split2(h1[2 * j], h1[2 * j + 1], hp[0][j], hp[1][j]);
            // g = 4 t + c: row n = s + 32 t, contraction chunk c.  Fragment element e <-> k = 16 c + 8 (e >> 2) + 4 hi + (e & 3), the
            // unit register 8 c + e of a D-layout vector holds: two 8-byte pieces of the natural row
            auto afrag = [&](int g, int term) -> u32x4 {
                const uint16_t* q = sW2p + term * 64 * NS + (s + 32 * (g >> 2)) * NS + (g & 3) * 16 + hi * 4;
                const uint2 lo2 = *reinterpret_cast<const uint2*>(q), hi2 = *reinterpret_cast<const uint2*>(q + 8);
                const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                return r;
            };
            f32x16 acc;
            u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1);
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
                const u32x4 a1 = an1, a2 = an2;
                if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); }
                MF_PIN();
                const u32x4 b1 = { hp[0][4 * c], hp[0][4 * c + 1], hp[0][4 * c + 2], hp[0][4 * c + 3] };
                const u32x4 b2 = { hp[1][4 * c], hp[1][4 * c + 1], hp[1][4 * c + 2], hp[1][4 * c + 3] };
                acc = mfma_x2(a1, a2, b1, b2, acc);
                if (c == 3) {
#pragma unroll
                    for (int r = 0; r < 16; r++) h2[16 * t + r] = tanh_scaled(acc[r]);
                }
            }
        }

        MF_STAMP(3);   // layer 2 MFMA + tanh
        // The per-sample scalars were requested at the top of the tile; nothing may touch them before this point (left alone, the
        // scheduler hoists the cheap `action == k` compare to the top of the tile, right behind the load, and the wave then sits
        // out a memory round trip there).  An empty asm that "redefines" them pins every use below this line.
        asm volatile("" : "+v"(s_f0), "+v"(s_f1), "+v"(s_actbits), "+v"(s_maskbits));
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
            const float R = s_f0, vold = s_f1;
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
            if (valid && hi == 0) st0 += lossv;
            dOut[0] = valid ? g_v : 0.0f;
        } else {
            const int A = AOUT;
            const float s_oldlp = s_f0, s_adv = s_f1;
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
                    if (DIST == PPO_DIST_MASKED) ok[k] = ((s_maskbits >> k) & 1u) != 0u;
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
                const int act_h = (int)((s_actbits >> (8 * h)) & 0xffu);
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
                st0 += l1 > l2 ? l1 : l2;
                st1 += ent;
                st2 += (ratio - 1.0f) - logratio;
                st3 += (fabsf(ratio - 1.0f) > clip) ? 1.0f : 0.0f;
            }
            off = 0;
#pragma unroll
            for (int h = 0; h < AMAX; h++) {
                if (h >= n_heads) break;
                const int Ah = EXACT ? AMAX : L.head_dims[h];
                const int act_h = (int)((s_actbits >> (8 * h)) & 0xffu);
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) {
                    float d = g_nlp * ((k == off + act_h ? 1.0f : 0.0f) - pr[k]);
                    if (DIST == PPO_DIST_MASKED) d += g_ent * (-pr[k] * (z[k] + headH[h]));
                    dOut[k] = (valid && ok[k]) ? d : 0.0f;
                }
                off += Ah;
            }
        }

        {
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < AMAX; k++) if (k < AOUT) dsum += fabsf(dOut[k]);
            const float tmax = u2f(wave_max_u(f2u(dsum))) * w3max;
            const int e = (int)((f2u(tmax) >> 23) & 0xffu) - 127;     // floor(log2(tmax)) (128: inf / NaN, -127: zero / denormal)
            if (e + S_w > 13) {
                int S_new = 11 - e;
                S_new = S_new < -100 ? -100 : S_new;
                int d = S_new - S_w;
                d = d < -126 ? -126 : d;
                const float f = pow2i(d);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) gW2[i][j][r] *= f;
#pragma unroll
                for (int o = 0; o < OBS; o++) gW1[o] *= f;
                if constexpr (DW1M) {
#pragma unroll
                    for (int b = 0; b < 4; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) gW1m[b][r] *= f;
                }
                gb1 *= f; gb2[0] *= f; gb2[1] *= f;
                S_w = S_new;
                scaleS = pow2i(S_w);
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
        {
            float accw3[AMAX];
#pragma unroll
            for (int k = 0; k < AMAX; k++) accw3[k] = 0.0f;
#pragma unroll
            for (int c0 = 0; c0 < MT; c0 += 16) {   // 16 image rows, read ONCE for all logits, + 4 x 16-byte dOut reads per logit in flight together
                float hv[16];
#pragma unroll
                for (int i = 0; i < 16; i++) hv[i] = img[(c0 + i) * LS + lane];
#pragma unroll
                for (int k = 0; k < AMAX; k++) {
                    if (k < AOUT) {
#pragma unroll
                        for (int i = 0; i < 16; i += 4) {
                            const float4 d = ld4(&sDo[k * MT + c0 + i]);
                            accw3[k] = __builtin_fmaf(d.x, hv[i], accw3[k]); accw3[k] = __builtin_fmaf(d.y, hv[i + 1], accw3[k]);
                            accw3[k] = __builtin_fmaf(d.z, hv[i + 2], accw3[k]); accw3[k] = __builtin_fmaf(d.w, hv[i + 3], accw3[k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < AMAX; k++) {
                if (k < AOUT) {
                    gW3[k] += accw3[k];
                    gb3[k] += hi == 0 ? dOut[k] : 0.0f;   // summed over lanes at the end
                }
            }
        }

        MF_STAMP(5);   // h2 image + dW3
        // ---------------- dz2 = (sum_a dOut[a] W3[a][u]) (1 - h2^2), D layout ----------------
        float dz2[32];   // carries the factor 2^S from here on, and so do dh1, dz1 and the accumulators they feed
#pragma unroll
        for (int e = 0; e < 32; e++) dz2[e] = 0.0f;
#pragma unroll
        for (int k = 0; k < AMAX; k++) {
            if (k < AOUT) {
                const float dS = dOut[k] * scaleS;
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const float4 w = ld4(&sW3[k * 64 + 8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                    dz2[4 * g] = __builtin_fmaf(dS, w.x, dz2[4 * g]); dz2[4 * g + 1] = __builtin_fmaf(dS, w.y, dz2[4 * g + 1]);
                    dz2[4 * g + 2] = __builtin_fmaf(dS, w.z, dz2[4 * g + 2]); dz2[4 * g + 3] = __builtin_fmaf(dS, w.w, dz2[4 * g + 3]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 32; e++) dz2[e] = dz2[e] * __builtin_fmaf(-h2[e], h2[e], 1.0f);   // 1 - h^2 in one rounding (was mul, sub: 32 instructions more)
        // B operand of d(hidden 1): dz2 in place (D layout), cut into its fp16 terms -- the ONLY split of dz2: the same terms, written to
        // the wave's region as two [sample][unit] fp16 images and read back transposed, are the A operand of dW2
        uint32_t zp[2][16];
#pragma unroll
        for (int j = 0; j < 16; j++) split2(dz2[2 * j], dz2[2 * j + 1], zp[0][j], zp[1][j]);
        wave_lds_fence();  // dW3 reads of the image are done
#pragma unroll
        for (int term = 0; term < 2; term++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int q = 0; q < 4; q++)   // registers 16t + 4q .. +3 = units 8q + 4hi + 32t .. +3 = pairs zp[.][8t + 2q], zp[.][8t + 2q + 1]
                    *reinterpret_cast<uint2*>(zimg + term * MT * NS + s * NS + 8 * q + 4 * hi + 32 * t) = make_uint2(zp[term][8 * t + 2 * q], zp[term][8 * t + 2 * q + 1]);
        wave_lds_fence();
        // A operand of dW2, both 16-sample chunks: lane (n = lane & 31, hi) holds dz2 terms of unit n + 32 tn for samples 16 c + 8 hi + 0..7.
        // Transposing read: the 16-lane group (lane >> 4) reads a 4-sample x 16-unit block; lane 4q + p supplies the address of sample row q,
        // units 4p .. 4p+3, and receives unit (lane & 15) of the four samples.
        const int tq = (lane & 15) >> 2, tp = lane & 3, tb = (lane >> 4) & 1;
        u32x4 A[2][2][2];   // [chunk][tn][term]
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int term = 0; term < 2; term++) {
                    const uint16_t* q = zimg + term * MT * NS + (16 * c + 8 * hi + tq) * NS + 32 * tn + 16 * tb + 4 * tp;
                    const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 4 * NS);
                    const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                    A[c][tn][term] = r;
                }
        // db2[n] = sum over samples of dz2[s][n]: each lane adds the 16 samples it holds (the two halves meet in the epilogue)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            float cacc = 0.0f;
#pragma unroll
            for (int term = 1; term >= 0; term--)   // small terms first
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int d = 0; d < 4; d++) cacc = add_pair(A[c][tn][term][d], cacc);
            gb2[tn] += cacc;
        }
        wave_lds_fence();  // the term images are in registers: the region is free for h1
        store_dlayout(img, h1, s, hi);
        wave_lds_fence();
        MF_STAMP(6);   // dz2, term images, A fragments, db2
        // ---------------- dW2[n][k] += sum_s dz2[s][n] h1[s][k]: two chunks of 16 samples, 24 MFMAs each ----------------
#pragma unroll
        for (int c = 0; c < 2; c++) {
            u32x4 B[2][2];
#pragma unroll
            for (int tk = 0; tk < 2; tk++) {
                float hb[8];
#pragma unroll
                for (int e = 0; e < 8; e++) hb[e] = img[(16 * c + 8 * hi + e) * LS + s + 32 * tk];   // h1[sample][k = s + 32 tk]
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t p1, p2;
                    split2(hb[2 * j], hb[2 * j + 1], p1, p2);
                    B[tk][0][j] = p1; B[tk][1][j] = p2;
                }
            }
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int tk = 0; tk < 2; tk++)
                    gW2[tn][tk] = mfma_x2(A[c][tn][0], A[c][tn][1], B[tk][0], B[tk][1], gW2[tn][tk]);
        }
        MF_STAMP(7);   // dW2 MFMA
        // ---------------- c dh1^T[k][s] = sum_n c W2[n][k] dz2[s][n], c dz1 = c dh1 (1 - h1^2) ----------------
        // g = 4 t + c: output row k = s + 32 t, contraction chunk c over n.  Fragment element e <-> n = 16 c + 8 (e >> 2) + 4 hi + (e & 3):
        // column k of rows 16 c + 4 hi .. +3 (first transposing read) and of rows 16 c + 8 + 4 hi .. +3 (second)
        float dz1[32];
        {
            auto afrag = [&](int g, int term) -> u32x4 {
                const uint16_t* q = sW2p + term * 64 * NS + (16 * (g & 3) + 4 * hi + tq) * NS + 32 * (g >> 2) + 16 * tb + 4 * tp;
                const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 8 * NS);
                const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                return r;
            };
            f32x16 acc;
            u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int t = g >> 2, c = g & 3;
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
                }
                const u32x4 a1 = an1, a2 = an2;
                if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); }
                MF_PIN();
                const u32x4 b1 = { zp[0][4 * c], zp[0][4 * c + 1], zp[0][4 * c + 2], zp[0][4 * c + 3] };
                const u32x4 b2 = { zp[1][4 * c], zp[1][4 * c + 1], zp[1][4 * c + 2], zp[1][4 * c + 3] };
                acc = mfma_x2(a1, a2, b1, b2, acc);
                if (c == 3) {   // h1 comes back from its [sample][unit] image (still intact: dW2 only read it)
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const float4 hb = ld4(&img[s * LS + 8 * qq + 4 * hi + 32 * t]);
                        // (1 - h^2), DW1M: times 2^-7 (an exact scaling), as one multiply and one fused multiply-add
                        constexpr float K7 = DW1M ? 0x1p-7f : 1.0f;
                        dz1[16 * t + 4 * qq + 0] = acc[4 * qq + 0] * __builtin_fmaf(hb.x * -K7, hb.x, K7);
                        dz1[16 * t + 4 * qq + 1] = acc[4 * qq + 1] * __builtin_fmaf(hb.y * -K7, hb.y, K7);
                        dz1[16 * t + 4 * qq + 2] = acc[4 * qq + 2] * __builtin_fmaf(hb.z * -K7, hb.z, K7);
                        dz1[16 * t + 4 * qq + 3] = acc[4 * qq + 3] * __builtin_fmaf(hb.w * -K7, hb.w, K7);
                    }
                }
            }
        }
        MF_STAMP(8);   // dh1 MFMA + dz1
        wave_lds_fence();  // dW2's reads of the h1 image are done
        if constexpr (DW1M) {
            // ---------------- c 2^-7 dW1[u][o] += sum_s dz1[s][u] x[s][o], db1 as the column x[s][OBS] := 1: 12 MFMAs of 16 x 16 x 32 ----------------
            {
                uint32_t zq[2][16];
#pragma unroll
                for (int j = 0; j < 16; j++) split2(dz1[2 * j], dz1[2 * j + 1], zq[0][j], zq[1][j]);
#pragma unroll
                for (int term = 0; term < 2; term++)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            *reinterpret_cast<uint2*>(zimg + term * MT * NS + s * NS + 8 * q + 4 * hi + 32 * t) = make_uint2(zq[term][8 * t + 2 * q], zq[term][8 * t + 2 * q + 1]);
            }
            // B operand: lane (j = lane & 15, kg = lane >> 4) holds x[s = 8 kg + e][o = j], e = 0 .. 7; column OBS is the constant 1, the rest 0
            const int j16 = lane & 15, kg = lane >> 4;
            u32x4 b1, b2;
            {
                const int jo = j16 < OBS ? j16 : OBS - 1;
                const float4 xa = ld4(&sX[jo * MT + 8 * kg]), xb = ld4(&sX[jo * MT + 8 * kg + 4]);
                const float fill = j16 == OBS ? 1.0f : 0.0f;
                const bool use = j16 < OBS;
                float xv[8] = { use ? xa.x : fill, use ? xa.y : fill, use ? xa.z : fill, use ? xa.w : fill,
                                use ? xb.x : fill, use ? xb.y : fill, use ? xb.z : fill, use ? xb.w : fill };
#pragma unroll
                for (int e = 0; e < 4; e++) { uint32_t p1, p2; split2(xv[2 * e], xv[2 * e + 1], p1, p2); b1[e] = p1; b2[e] = p2; }
            }
            wave_lds_fence();
            // A operand of unit block b: the 16-lane group kg reads samples 8 kg .. + 3 (second read: + 4) x units 16 b .. + 15 and receives unit
            // 16 b + (lane & 15) of the four samples
#pragma unroll
            for (int b = 0; b < 4; b++) {
                u32x4 A1, A2;
                {
                    const uint16_t* q = zimg + (8 * kg + tq) * NS + 16 * b + 4 * tp;
                    const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 4 * NS);
                    const uint2 lo3 = lds_read_tr16(q + MT * NS), hi3 = lds_read_tr16(q + MT * NS + 4 * NS);
                    A1 = u32x4{ lo2.x, lo2.y, hi2.x, hi2.y };
                    A2 = u32x4{ lo3.x, lo3.y, hi3.x, hi3.y };
                }
                f32x4 acc4 = gW1m[b];
                acc4 = mfma16_f16(A2, b1, acc4);
                acc4 = mfma16_f16(A1, b2, acc4);
                acc4 = mfma16_f16(A1, b1, acc4);
                gW1m[b] = acc4;
            }
        } else {
        store_dlayout(img, dz1, s, hi);
        wave_lds_fence();
        // ---------------- c dW1[u = lane][o] += sum_s c dz1[s][u] x[s][o]; c db1 ----------------
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
        }
        wave_lds_fence();  // image and sX are rewritten by the next tile
        MF_STAMP(9);   // dz1 image + dW1
    }

    if constexpr (STAMP) {   // slot 11: tile-loop cycles of wave 4 (wave 0's SIMD partner), to compare with wave 0's phases 1..9
        if (blk == 0 && tid == 256 && a.stamps) a.stamps[NET * 12 + 11] += __builtin_amdgcn_s_memtime() - t_prev0;
    }
    // ---------------- the waves park their accumulators in private LDS regions (plain stores), then every thread adds the
    //                  regions in a fixed order into the workgroup's slab ----------------
    // loss sums first: registers only, and their landing place lies beyond every live byte of LDS, so this needs no barrier
    const double sd0 = wave_sum_d_dpp((double)st0), sd1 = wave_sum_d_dpp((double)st1), sd2 = wave_sum_d_dpp((double)st2), sd3 = wave_sum_d_dpp((double)st3);
    const int base = L.net_off[NET];
    const int nsz = L.net_size[NET];
    const int rstride = (nsz + 3) & ~3;
    int dred_off = MF_WAVES * rstride + 4;
    if (dred_off < m.total) dred_off = m.total;
    dred_off = (dred_off + 1) & ~1;
    double* dred = reinterpret_cast<double*>(smem + dred_off);
    if (lane == 0) { dred[wave * 4 + 0] = sd0; dred[wave * 4 + 1] = sd1; dred[wave * 4 + 2] = sd2; dred[wave * 4 + 3] = sd3; }
    // the two sample halves of the db2 partials meet; the layer-1 gradients lose the factor c they inherited from the scaled W2
#pragma unroll
    for (int tn = 0; tn < 2; tn++) gb2[tn] += __shfl_xor(gb2[tn], 32, 64);
    {   // the factor 2^S of the fp16 range leaves dW2, db2, dW1, db1 (exact)
        const float invS = pow2i(-S_w);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) gW2[i][j][r] *= invS;
        gb2[0] *= invS; gb2[1] *= invS;
        gb1 *= invS;
#pragma unroll
        for (int o = 0; o < OBS; o++) gW1[o] *= invS;
        if constexpr (DW1M) {   // ... and the fixed 2^-7 of the dz1 terms
            const float f = invS * 0x1p7f;
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) gW1m[b][r] = (gW1m[b][r] * f) * TANH_C_INV;
        }
    }
    gb1 *= TANH_C_INV;
#pragma unroll
    for (int o = 0; o < OBS; o++) gW1[o] *= TANH_C_INV;
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
    if constexpr (DW1M) {
        const int j16 = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int u = 16 * b + 4 * kg + r;
                if (j16 < OBS) red[L.w1[NET] - base + u * OBS + j16] = gW1m[b][r];
                else if (j16 == OBS) red[L.b1[NET] - base + u] = gW1m[b][r];
            }
    } else {
#pragma unroll
        for (int o = 0; o < OBS; o++) red[L.w1[NET] - base + lane * OBS + o] = gW1[o];
        red[L.b1[NET] - base + lane] = gb1;
    }
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

template <int DIST, int OBS, int AMAX, bool EXACT, bool STAMP>
__global__ __launch_bounds__(MF_THREADS, 1) void fwd_bwd_mfma_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 1-D grid: the first n_blocks[0] workgroups run the critic, the rest the actor
    const int b = blockIdx.x;
    constexpr bool DW1M = EXACT;   // the reference's two shapes: 234 / 237 vector registers leave room for 12 more; the generic variants (245 - 247) do not
    if (b < a.n_blocks[0]) mf_body<0, DIST, OBS, 1, true, STAMP, DW1M>(a, smem, b, a.n_blocks[0]);
    else mf_body<1, DIST, OBS, AMAX, EXACT, STAMP, DW1M>(a, smem, b - a.n_blocks[0], a.n_blocks[1]);
}


// =========================================================================================================
// Wave-specialised form of the same step for the reference's two shapes (CartPole 4 -> 2, MountainCar 2 -> 3 masked).
//
// mf_body above gives one wave everything from gather to weight gradient; its 80 gradient accumulators pin it at 234 - 250 registers (two
// waves per SIMD, nothing to spare), and every product that contracts over SAMPLES (dW3, dW2, dW1, the biases) makes the wave wait on its own
// LDS round trips.  Here a workgroup is TWELVE waves with two jobs (round 3 did the same to the rollout):
//   * waves 0-7  ("F": forward, loss, d(activation)): everything whose natural layout is lane = sample -- gather, layer 1, layer 2, head, PPO
//     loss, dz2, d(hidden 1), dz1.  They only WRITE the sample-major images: h1 (fp16 terms, formed for layer 2 anyway -- no third split), h2
//     (fp32), dz2 and dz1 (fp16 terms), the observation's terms, dOut.  ~130 registers, no gradient accumulator.
//   * waves 8-11 ("G": gradients): everything that contracts over samples, i.e. needs lane = unit -- dW3 / db3 on the vector ALU from the h2
//     image, dW2 / db2 and dW1 / db1 on the matrix cores from transposing reads of the term images.  Wave 8 + g serves F waves g and g + 4
//     (its own SIMD's), holds the 80 accumulators for both, and is otherwise asleep.
// Hand-over is through two LDS regions per F wave and two counters per F wave (events produced / consumed), polled with s_sleep: no barrier
// inside the tile loop, so no wave waits for a slower one.  Per tile an F wave produces three events
//   E0: RB = h2 image, sDo = dOut            -> G: dW3, db3
//   E1: RA = h1 terms + x terms, RB = dz2 terms, SEV = the tile's power-of-two scale S  -> G: dW2, db2
//   E2: RB = dz1 terms                        -> G: dW1, db1 (x terms from RA)
// and before it overwrites a region it waits for the count of consumed events that frees it (RA: everything of the previous tile; RB: the
// previous event).  G acknowledges an event as soon as its LDS reads have returned, before it issues the MFMAs.  Every wait is bounded: a
// protocol error raises the context's error flag instead of hanging the GPU.
// The scale S (see "Arithmetic" at the top) is each F wave's own -- a function of its own tiles only, quantised to multiples of four so that the two
// F waves of a G wave mostly agree; an event carries it, and G moves the accumulators it adds to by the power of two between their scale and the
// event's (exact).  With the fixed service order (a, b, a, b ...) the result does not depend on timing: same inputs, same bits.
// =========================================================================================================
constexpr int MG_FW = 8, MG_GW = 4, MG_WAVES = MG_FW + MG_GW, MG_THREADS = 64 * MG_WAVES;
enum { MG_PROD = 0, MG_CONS = 8, MG_SEV = 16, MG_ONE = 32, MG_ZERO = 34, MG_FLAG_WORDS = 64 };

struct MgSmem {
    int w2, w1, b1, b2, w3, b3, flags, wave0, wave_stride, ra, rb, s_do, dred, total;  // offsets in floats
};
__host__ __device__ inline MgSmem mg_smem(int obs, int aout) {
    MgSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.w2 = take(2 * 64 * NS / 2);
    m.w1 = take(64 * obs);
    m.b1 = take(64);
    m.b2 = take(64);
    m.w3 = take(aout * 64);
    m.b3 = take(aout);
    m.flags = take(MG_FLAG_WORDS);
    m.wave0 = o;
    int w = 0;
    auto takew = [&](int n) { int r = w; w += (n + 3) & ~3; return r; };
    m.ra = takew(MT * NS);       // two fp16 term images [term][sample][NS]; columns 64 .. 67 of a row carry the sample's observation terms
    m.rb = takew(MF_REGION);     // h2 as fp32 [sample][LS], then dz2 terms, then dz1 terms
    m.s_do = takew(aout * MT);   // [a][sample]
    m.wave_stride = w;
    o += MG_FW * w;
    m.dred = take(4 * 2 * MG_FW);  // loss sums of the F waves (doubles), beyond everything the epilogue reuses
    m.total = o;
    return m;
}

// Region slot of F wave fw: the two F waves a G wave serves (g and MG_GW + ((g + 1) & 3), on different SIMDs) own ADJACENT regions 2g, 2g + 1 --
// once it has served their last events nobody else touches that pair, and its gradient image (net_size floats <= 2 regions) can be parked there
// without waiting for the rest of the workgroup.
__host__ __device__ constexpr int mg_slot(int fw) { return fw < MG_GW ? 2 * fw : 2 * ((fw - MG_GW + MG_GW - 1) & (MG_GW - 1)) + 1; }

// wave-uniform wait for a workgroup-shared LDS counter to reach `want`; false when the bound runs out (protocol error)
typedef __attribute__((address_space(3))) volatile uint32_t lds_flag_t;
__device__ __forceinline__ bool mg_wait_ge(const lds_flag_t* flag, uint32_t want) {
    for (int spin = 0; spin < (1 << 21); spin++) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)*flag);
        if (v >= want) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
// The counter moves BEHIND this wave's LDS accesses so far.  No s_waitcnt: a wave's LDS instructions are executed by the LDS unit in the order they
// were issued (that is what makes lgkmcnt a counter), so whoever reads the new count afterwards finds the data written / the region already read.
// The asm is the compiler-side fence.  (-DMG_POST_WAIT restores the explicit wait for an A/B.)
__device__ __forceinline__ void mg_post(lds_flag_t* flag, uint32_t v, int lane) {
#ifdef MG_POST_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    if (lane == 0) *flag = v;
    asm volatile("" ::: "memory");
}

// -DMG_STAMP (diagnostic build, tools/ws_stamps.py): cycles F wave 0 / G wave 8 of workgroup 0 spend inside their waits and services
#ifdef MG_STAMP
#define MG_TIMED(slot, expr) do { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); expr; tstamp[slot] += __builtin_amdgcn_s_memtime() - t0_; } while (0)
#else
#define MG_TIMED(slot, expr) do { expr; } while (0)
#endif
// -DMG_TRACE (diagnostic build, tools/ws_trace.py; implies MG_STAMP's buffer): the hand-over timeline of ONE tile round (the third) of the actor's
// workgroup 0 -- F wave 0 ("a" of G wave 8), F wave 5 (its "b") and G wave 8 -- as cycles since the wave's first instruction, summed over launches.
#ifdef MG_TRACE
#define MG_ST_ADD(slot, v) do { } while (0)   // the trace owns the 24 slots
#else
#define MG_ST_ADD(slot, v) atomicAdd(a.stamps + NET * 12 + (slot), (v))
#endif
#ifdef MG_TRACE
#define MG_TR_F(slot_a, slot_b) do { if (NET == 1 && blk == 0 && it == 2 && lane == 0 && a.stamps) {                         \
        if (wave == 0 && (slot_a) >= 0) atomicAdd(a.stamps + (slot_a), (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));     \
        if (wave == 5 && (slot_b) >= 0) atomicAdd(a.stamps + (slot_b), (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); } } while (0)
#define MG_TR_G(slot) do { if (NET == 1 && blk == 0 && tr_round && wave == MG_FW && lane == 0 && a.stamps)                      \
        atomicAdd(a.stamps + (slot), (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); } while (0)
#else
#define MG_TR_F(slot_a, slot_b) do { } while (0)
#define MG_TR_G(slot) do { } while (0)
#endif
template <int NET, int DIST, int OBS, int AMAX>
__device__ __forceinline__ void mg_body(const UpdateArgs& a, float* smem, const int blk, const int nblk) {
    const NetLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 31, hi = lane >> 5;
    constexpr int AOUT = NET == 0 ? 1 : AMAX;
    const MgSmem m = mg_smem(OBS, AOUT);
    const uint16_t* sW2p = reinterpret_cast<const uint16_t*>(smem + m.w2);
    float* sW1 = smem + m.w1;
    float* sB1 = smem + m.b1;
    float* sB2 = smem + m.b2;
    float* sW3 = smem + m.w3;
    float* sB3 = smem + m.b3;
    lds_flag_t* flags = (lds_flag_t*)(smem + m.flags);   // explicit LDS address space: a generic volatile access is a FLAT instruction and waits for vmcnt too
    const float* __restrict__ P = a.params;
    const int n_tiles = (a.M + MT - 1) / MT;
    const int tile_step = nblk * MG_FW;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(NET == 0 ? a.rec_critic : a.rec_actor);
    const int tq = (lane & 15) >> 2, tp = lane & 3, tb = (lane >> 4) & 1;
    bool proto_ok = true;
#ifdef MG_STAMP
    unsigned long long tstamp[6] = { 0, 0, 0, 0, 0, 0 };
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

    // ---- an F wave requests its first tile's batch rows before anything else (see mf_body) ----
    int tile = blk * MG_FW + (wave < MG_FW ? wave : 0);
    auto fetch_row = [&](int tl) -> int {
        const int j = tl * MT + s;
        const bool ok = tl < n_tiles && j < a.M;
        const int v = a.idx[ok ? j : 0];
        return ok ? v : -1;
    };
    int row_n = -1, row_nn = -1;
    if (wave < MG_FW) { row_n = fetch_row(tile); row_nn = fetch_row(tile + tile_step); }

    // ---- weights of this net -> LDS (once per launch, all twelve waves; W1, b1, W2, b2 multiplied by c = 2 log2(e), W2 cut into its fp16 terms) ----
    constexpr int NW2 = (4096 + MG_THREADS - 1) / MG_THREADS;
    {
        float wv[NW2 + 5];
#pragma unroll
        for (int i = 0; i < NW2; i++) { const int e = tid + i * MG_THREADS; wv[i] = P[L.w2[NET] + (e < 4096 ? e : 0)]; }
        wv[NW2] = P[L.w1[NET] + (tid < 64 * OBS ? tid : 0)];
        wv[NW2 + 1] = P[L.w3[NET] + (tid < AOUT * 64 ? tid : 0)];
        wv[NW2 + 2] = P[L.b1[NET] + (tid & 63)];
        wv[NW2 + 3] = P[L.b2[NET] + (tid & 63)];
        wv[NW2 + 4] = P[L.b3[NET] + (tid < AOUT ? tid : 0)];
        uint16_t* wn = reinterpret_cast<uint16_t*>(smem + m.w2);
#pragma unroll
        for (int i = 0; i < NW2; i++) {
            const int e = tid + i * MG_THREADS;
            const int n = e >> 6, k = e & 63;
            uint32_t p1, p2;
            split2(wv[i] * TANH_C, 0.0f, p1, p2);
            if (e < 4096) { wn[n * NS + k] = (uint16_t)p1; wn[64 * NS + n * NS + k] = (uint16_t)p2; }
        }
        static_assert(64 * OBS <= MG_THREADS && AMAX * 64 <= MG_THREADS, "one staging slot per thread");
        if (tid < 64 * OBS) sW1[tid] = wv[NW2] * TANH_C;
        if (tid < AOUT * 64) sW3[tid] = wv[NW2 + 1];
        if (tid < 64) { sB1[tid] = wv[NW2 + 2] * TANH_C; sB2[tid] = wv[NW2 + 3] * TANH_C; }
        if (tid < AOUT) sB3[tid] = wv[NW2 + 4];
        if (tid < MG_FLAG_WORDS) flags[tid] = tid == MG_ONE ? 0x3c00u : 0u;   // MG_ONE: the fp16 cell {1, 0, 0, 0}
    }
    float4 x_n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (wave < MG_FW) x_n = rec[4 * (size_t)(row_n < 0 ? 0 : row_n)];
    __syncthreads();
#ifdef MG_STAMP
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
    if (blk == 0 && tid == 0 && a.stamps) MG_ST_ADD(7, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));   // prologue: weights -> LDS
#endif

    const int base = L.net_off[NET];
    double* dred = reinterpret_cast<double*>(smem + m.dred);

    if (wave < MG_FW) {
        // =================================================== F: forward, loss, d(activation) ===================================================
        float* wbase = smem + m.wave0 + mg_slot(wave) * m.wave_stride;
        uint16_t* ra16 = reinterpret_cast<uint16_t*>(wbase + m.ra);
        float* img = wbase + m.rb;
        uint16_t* zimg = reinterpret_cast<uint16_t*>(img);
        float* sDo = wbase + m.s_do;
        lds_flag_t* prod = flags + MG_PROD + wave;
        const lds_flag_t* cons = flags + MG_CONS + wave;
        float st0 = 0.0f, st1 = 0.0f, st2 = 0.0f, st3 = 0.0f;
        const float clip = a.hp.clip_coef;
        const float lo = uniform_f(1 - clip), hi_c = uniform_f(1 + clip);
        const float invM = uniform_f((float)a.inv_global_M);
        float mean_f = 0.0f, inv_std = 0.0f;
        if (NET == 1 && a.hp.norm_adv) {
            const float4 an = *a.adv_norm;
            mean_f = uniform_f(an.x);
            inv_std = uniform_f(an.y);
        }
        float w3max;
        {
            uint32_t mb = 0u;
#pragma unroll
            for (int k = 0; k < AOUT; k++) mb = max(mb, f2u(fabsf(sW3[k * 64 + lane])));
            w3max = u2f(wave_max_u(mb));
        }
        int S_w = 100;
        // Operands that stay in registers for the whole launch (an F wave has ~45 registers to spare; LDS is the busy pipe).  Each is a complete MFMA
        // operand (four dwords, zeros included): assembling one from single dwords costs three moves per use.  Lanes hi = 1 (k = 8 .. 15) hold zeros in
        // every A operand below, so whatever the B operand carries there drops out.
        //   * layer 1 on the f16 matrix cores like the rest, K = 16 with the bias as one more input: row u = s + 32 t of [c W1 | c b1] as fp16 terms at
        //     k = 0 .. OBS (B: the observation's terms and the constant 1).  Replaces 8 x 16-byte bias reads + 4 weight reads per tile, and the fp32
        //     MFMAs that exclude every other vector instruction of the SIMD while they run;
        //   * c b2: ONE product in front of layer 2's twelve, both terms side by side at k = 0, 1 against the constants 1, 1: 8 x 16-byte reads less;
        //   * W3 transposed for dz2 = W3^T dOut on the matrix cores (K = the logits).  Up to two logits the three term products sit side by side in ONE
        //     instruction: A = [w1 w1 w2] at k = 0 .. 5, B = [d1 d2 d1]; otherwise three instructions.
        // The scale 2^S bounds dz2 = W3^T dOut, not dOut: with a small W3 (the actor's head starts at gain 0.01) dOut 2^S alone overruns fp16.  W3 enters
        // its operand divided by P3 = the power of two just above max|W3| (|W3 / P3| <= 1) and dOut multiplied by 2^S P3 (<= 2^15): exact, the product is the same.
        const int e3 = w3max > 0.0f ? (int)((f2u(w3max) >> 23) & 0xffu) - 126 : 0;   // max|W3| < 2^e3
        const float w3norm = pow2i(e3 < -100 ? 100 : (e3 > 100 ? -100 : -e3));
        const float p3 = pow2i(e3 < -100 ? -100 : (e3 > 100 ? 100 : e3));
        constexpr int WR = AOUT <= 2 ? 1 : 2;  // dwords of one term of a lane's W3^T / dOut operand
        u32x4 w1a[2][2];                       // [t][term]
        u32x4 b2a[2];                          // [t]
        u32x4 w3a[2][WR == 1 ? 1 : 2];         // [t][term] (one combined operand up to two logits)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int u = s + 32 * t;
            float wrow[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int o = 0; o < OBS; o++) wrow[o] = sW1[u * OBS + o];
            uint32_t p1[3], p2[3];
            if constexpr (OBS == 4) { split2(wrow[0], wrow[1], p1[0], p2[0]); split2(wrow[2], wrow[3], p1[1], p2[1]); split2(sB1[u], 0.0f, p1[2], p2[2]); }
            else { split2(wrow[0], wrow[1], p1[0], p2[0]); split2(sB1[u], 0.0f, p1[1], p2[1]); p1[2] = 0u; p2[2] = 0u; }
            w1a[t][0] = u32x4{ hi ? 0u : p1[0], hi ? 0u : p1[1], hi ? 0u : p1[2], 0u };
            w1a[t][1] = u32x4{ hi ? 0u : p2[0], hi ? 0u : p2[1], hi ? 0u : p2[2], 0u };
            uint32_t q1, q2;
            split2(sB2[u], 0.0f, q1, q2);   // low halves: the two terms
            b2a[t] = u32x4{ hi ? 0u : ((q1 & 0xffffu) | (q2 << 16)), 0u, 0u, 0u };
            float w3c[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int k = 0; k < AOUT; k++) w3c[k] = sW3[k * 64 + u] * w3norm;
            uint32_t r1[2], r2[2];
            split2(w3c[0], w3c[1], r1[0], r2[0]);
            split2(w3c[2], w3c[3], r1[1], r2[1]);
            if constexpr (WR == 1) w3a[t][0] = u32x4{ hi ? 0u : r1[0], hi ? 0u : r1[0], hi ? 0u : r2[0], 0u };
            else { w3a[t][0] = u32x4{ hi ? 0u : r1[0], hi ? 0u : r1[1], 0u, 0u }; w3a[t][WR == 1 ? 0 : 1] = u32x4{ hi ? 0u : r2[0], hi ? 0u : r2[1], 0u, 0u }; }
        }
        const u32x4 ones2 = { 0x3c003c00u, 0u, 0u, 0u };   // B operand of the bias product: the constant 1 at k = 0 and k = 1
        const f32x16 zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        const int wave_half = (wave >> 2) & 1;
        uint32_t ev = 0;   // events produced so far (3 per tile)
        for (int it = 0; tile < n_tiles; tile += tile_step, it++) {
            if ((it ^ wave_half) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            MG_TR_F(0, 8);
            // ---------------- gather (K5) ----------------
            const bool valid = row_n >= 0;
            const int row = valid ? row_n : 0;
            // four scalars, not an array: a select between array elements by `hi` (layer 1's B operand) sends an array to scratch memory
            const float x0 = valid ? x_n.x : 0.0f, x1 = valid ? x_n.y : 0.0f, x2 = valid ? x_n.z : 0.0f, x3 = valid ? x_n.w : 0.0f;
            const float4 sc = rec[4 * (size_t)row + 1];
            float s_f0 = sc.x, s_f1 = sc.y;
            uint32_t s_actbits = f2u(sc.z), s_maskbits = f2u(sc.w);
            {
                row_n = row_nn;
                x_n = rec[4 * (size_t)(row_n < 0 ? 0 : row_n)];
                row_nn = fetch_row(tile + 2 * tile_step);
            }
            // ---------------- layer 1 (f16 MFMA on terms, K = 16: observation + the constant 1 that carries the bias) ----------------
            // observation terms: also what G gets for dW1 (columns 64 .. 67 of the sample's row in each h1 term image; column OBS is the constant 1 of db1
            // when it fits the 4-column cell)
            uint32_t xq[2][2];
            if constexpr (OBS == 4) { split2(x0, x1, xq[0][0], xq[1][0]); split2(x2, x3, xq[0][1], xq[1][1]); }
            else { split2(x0, x1, xq[0][0], xq[1][0]); split2(1.0f, 0.0f, xq[0][1], xq[1][1]); }
            float h1[32];
            {
                // k = 0 .. OBS: the observation's terms and the constant 1 (first term only); what lanes hi = 1 carry meets zeros in A
                u32x4 xb1, xb2;
                if constexpr (OBS == 4) { xb1 = u32x4{ xq[0][0], xq[0][1], 0x00003c00u, 0u }; xb2 = u32x4{ xq[1][0], xq[1][1], 0u, 0u }; }
                else { xb1 = u32x4{ xq[0][0], xq[0][1], 0u, 0u }; xb2 = u32x4{ xq[1][0], xq[1][1], 0u, 0u }; }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const f32x16 acc = mfma_x2(w1a[t][0], w1a[t][1], xb1, xb2, zero16);
#pragma unroll
                    for (int r = 0; r < 16; r++) h1[16 * t + r] = tanh_scaled(acc[r]);
                }
            }
            // ---------------- layer 2 forward; h1's terms (and the observation's) leave for G on the way ----------------
            float h2[32];
            {
                uint32_t hp[2][16];
#pragma unroll
                for (int j = 0; j < 16; j++) split2(h1[2 * j], h1[2 * j + 1], hp[0][j], hp[1][j]);
                if (it > 0) MG_TIMED(1, proto_ok &= mg_wait_ge(cons, ev));   // G has read everything of the previous tile: RA is free
#pragma unroll
                for (int term = 0; term < 2; term++) {
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            *reinterpret_cast<uint2*>(ra16 + term * MT * NS + s * NS + 8 * q + 4 * hi + 32 * t) = make_uint2(hp[term][8 * t + 2 * q], hp[term][8 * t + 2 * q + 1]);
                    if (hi == 0) *reinterpret_cast<uint2*>(ra16 + term * MT * NS + s * NS + 64) = make_uint2(xq[term][0], xq[term][1]);
                }
                auto afrag = [&](int g, int term) -> u32x4 {
                    const uint16_t* q = sW2p + term * 64 * NS + (s + 32 * (g >> 2)) * NS + (g & 3) * 16 + hi * 4;
                    const uint2 lo2 = *reinterpret_cast<const uint2*>(q), hi2 = *reinterpret_cast<const uint2*>(q + 8);
                    const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                    return r;
                };
                f32x16 acc;
                u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1);
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int t = g >> 2, c = g & 3;
                    if (c == 0) {   // c b2 enters as a product: row n of A carries the bias's two terms at k = 0, 1, every column of B the constant 1 there
                        acc = mfma_f16(b2a[t], ones2, zero16);
                    }
                    const u32x4 a1 = an1, a2 = an2;
                    if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); }
                    MF_PIN();
                    const u32x4 b1 = { hp[0][4 * c], hp[0][4 * c + 1], hp[0][4 * c + 2], hp[0][4 * c + 3] };
                    const u32x4 b2 = { hp[1][4 * c], hp[1][4 * c + 1], hp[1][4 * c + 2], hp[1][4 * c + 3] };
                    acc = mfma_x2(a1, a2, b1, b2, acc);
                    if (c == 3) {
#pragma unroll
                        for (int r = 0; r < 16; r++) h2[16 * t + r] = tanh_scaled(acc[r]);
                    }
                }
            }
            asm volatile("" : "+v"(s_f0), "+v"(s_f1), "+v"(s_actbits), "+v"(s_maskbits));
            // ---------------- head + loss (K6, K7): as mf_body ----------------
            float dOut[AMAX];
#pragma unroll
            for (int k = 0; k < AMAX; k++) dOut[k] = 0.0f;
            if (NET == 0) {
                float part = 0.0f;
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const float4 w = ld4(&sW3[8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                    part = __builtin_fmaf(h2[4 * g], w.x, part); part = __builtin_fmaf(h2[4 * g + 1], w.y, part);
                    part = __builtin_fmaf(h2[4 * g + 2], w.z, part); part = __builtin_fmaf(h2[4 * g + 3], w.w, part);
                }
                const float other = __shfl_xor(part, 32, 64);
                const float v = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[0];
                const float R = s_f0, vold = s_f1;
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
                if (valid && hi == 0) st0 += lossv;
                dOut[0] = valid ? g_v : 0.0f;
            } else {
                const float s_oldlp = s_f0, s_adv = s_f1;
                float z[AMAX], pr[AMAX];
                bool ok[AMAX];
#pragma unroll
                for (int k = 0; k < AMAX; k++) {
                    float part = 0.0f;
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        const float4 w = ld4(&sW3[k * 64 + 8 * (g & 3) + 4 * hi + 32 * (g >> 2)]);
                        part = __builtin_fmaf(h2[4 * g], w.x, part); part = __builtin_fmaf(h2[4 * g + 1], w.y, part);
                        part = __builtin_fmaf(h2[4 * g + 2], w.z, part); part = __builtin_fmaf(h2[4 * g + 3], w.w, part);
                    }
                    const float other = __shfl_xor(part, 32, 64);
                    z[k] = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[k];
                    ok[k] = true;
                    if (DIST == PPO_DIST_MASKED) ok[k] = ((s_maskbits >> k) & 1u) != 0u;
                    if (DIST == PPO_DIST_MASKED && !ok[k]) z[k] = -1e8f;
                }
                const int act_h = (int)(s_actbits & 0xffu);
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < AMAX; k++) mx = z[k] > mx ? z[k] : mx;
                float se = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) { pr[k] = fast_exp(z[k] - mx); se += pr[k]; }
                const float lse = fast_log(se) + mx;
                const float rse = __builtin_amdgcn_rcpf(se);
                float e1 = 0.0f, nlp = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) {
                    z[k] = z[k] - lse;
                    pr[k] = pr[k] * rse;
                    if (DIST == PPO_DIST_CATEGORICAL) {
                        const float l = z[k] > 1.17549435e-38f ? z[k] : 1.17549435e-38f;   // the reference's clamp (Categorical.cpp:112-119)
                        e1 += l * pr[k];
                    } else {
                        e1 += ok[k] ? z[k] * pr[k] : 0.0f;
                    }
                    if (k == act_h) nlp = z[k];
                }
                const float ent = -e1;
                const float logratio = nlp - s_oldlp;           // :585
                const float ratio = fast_exp(logratio);         // :586
                float adv = s_adv;
                if (a.hp.norm_adv) adv = (adv - mean_f) * inv_std;           // :593
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
                    st0 += l1 > l2 ? l1 : l2;
                    st1 += ent;
                    st2 += (ratio - 1.0f) - logratio;
                    st3 += (fabsf(ratio - 1.0f) > clip) ? 1.0f : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < AMAX; k++) {
                    float d = g_nlp * ((k == act_h ? 1.0f : 0.0f) - pr[k]);
                    if (DIST == PPO_DIST_MASKED) d += g_ent * (-pr[k] * (z[k] + ent));
                    dOut[k] = (valid && ok[k]) ? d : 0.0f;
                }
            }
            // ---------------- E0: h2 image + dOut -> G (dW3, db3).  RB is free: everything of the previous tile was consumed (waited for above) ----------------
            store_dlayout(img, h2, s, hi);
            if (hi == 0) {
#pragma unroll
                for (int k = 0; k < AOUT; k++) sDo[k * MT + s] = dOut[k];
            }
            mg_post(prod, ++ev, lane);
            MG_TR_F(1, 9);
            // ---------------- the tile's power-of-two scale S: this wave's own (a function of ITS tiles only: results do not depend on timing), moved
            //                  down when a tile's bound (sum_a |dOut[a]|) max|W3| 2^S would pass 2^13, and then to a multiple of 4 with the bound at
            //                  2^8 .. 2^11, so that the two F waves of a G wave mostly agree and G seldom has to move its accumulators ----------------
            {
                float dsum = 0.0f;
#pragma unroll
                for (int k = 0; k < AOUT; k++) dsum += fabsf(dOut[k]);
                const float tmax = u2f(wave_max_u(f2u(dsum))) * w3max;
                const int e = (int)((f2u(tmax) >> 23) & 0xffu) - 127;
                if (e + S_w > 13) {
                    int S_new = (11 - e) & ~3;
                    S_w = S_new < -100 ? -100 : S_new;
                }
            }
            const float scaleS = pow2i(S_w);
            // ---------------- dz2 = (sum_a dOut[a] W3[a][u]) (1 - h2^2) 2^S, D layout ----------------
            float dz2[32];
            {
                // B operand: dOut[s][.] 2^S as terms at the k slots of A's W3 terms
                uint32_t d1[2], d2[2];
                const float sc3 = scaleS * p3;
                split2(dOut[0] * sc3, (AOUT > 1 ? dOut[AOUT > 1 ? 1 : 0] : 0.0f) * sc3, d1[0], d2[0]);
                if constexpr (WR == 2) split2(dOut[AOUT > 2 ? 2 : 0] * sc3, (AOUT > 3 ? dOut[AOUT > 3 ? 3 : 0] : 0.0f) * sc3, d1[1], d2[1]);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    f32x16 acc;
                    if constexpr (WR == 1) acc = mfma_f16(w3a[t][0], u32x4{ d1[0], d2[0], d1[0], 0u }, zero16);   // [w1 w1 w2] . [d1 d2 d1]
                    else acc = mfma_x2(w3a[t][0], w3a[t][WR == 1 ? 0 : 1], u32x4{ d1[0], d1[1], 0u, 0u }, u32x4{ d2[0], d2[1], 0u, 0u }, zero16);
#pragma unroll
                    for (int r = 0; r < 16; r++) dz2[16 * t + r] = acc[r] * __builtin_fmaf(-h2[16 * t + r], h2[16 * t + r], 1.0f);
                }
            }
            uint32_t zp[2][16];
#pragma unroll
            for (int j = 0; j < 16; j++) split2(dz2[2 * j], dz2[2 * j + 1], zp[0][j], zp[1][j]);
            // ---------------- E1: dz2 terms -> RB once G has read the h2 image ----------------
            if (lane == 0) flags[MG_SEV + wave] = (uint32_t)S_w;
            MG_TR_F(2, -1);
            MG_TIMED(2, proto_ok &= mg_wait_ge(cons, ev));
            MG_TR_F(3, -1);
#pragma unroll
            for (int term = 0; term < 2; term++)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        *reinterpret_cast<uint2*>(zimg + term * MT * NS + s * NS + 8 * q + 4 * hi + 32 * t) = make_uint2(zp[term][8 * t + 2 * q], zp[term][8 * t + 2 * q + 1]);
            mg_post(prod, ++ev, lane);
            MG_TR_F(4, 10);
            // ---------------- c dh1^T[k][s] = sum_n c W2[n][k] dz2[s][n];  c 2^-7 dz1 = c dh1 (1 - h1^2) 2^-7, h1 rebuilt from its own terms in RA ----------------
            float dz1[32];
            {
                auto afrag = [&](int g, int term) -> u32x4 {
                    const uint16_t* q = sW2p + term * 64 * NS + (16 * (g & 3) + 4 * hi + tq) * NS + 32 * (g >> 2) + 16 * tb + 4 * tp;
                    const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 8 * NS);
                    const u32x4 r = { lo2.x, lo2.y, hi2.x, hi2.y };
                    return r;
                };
                f32x16 acc;
                u32x4 an1 = afrag(0, 0), an2 = afrag(0, 1);
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int t = g >> 2, c = g & 3;
                    if (c == 0) {
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
                    }
                    const u32x4 a1 = an1, a2 = an2;
                    if (g + 1 < 8) { an1 = afrag(g + 1, 0); an2 = afrag(g + 1, 1); }
                    MF_PIN();
                    const u32x4 b1 = { zp[0][4 * c], zp[0][4 * c + 1], zp[0][4 * c + 2], zp[0][4 * c + 3] };
                    const u32x4 b2 = { zp[1][4 * c], zp[1][4 * c + 1], zp[1][4 * c + 2], zp[1][4 * c + 3] };
                    acc = mfma_x2(a1, a2, b1, b2, acc);
                    if (c == 3) {
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            // h1 = t1 + t2 (one v_fma_mix_f32 per value: exact, the terms were cut from h1)
                            const uint2 u1 = *reinterpret_cast<const uint2*>(ra16 + s * NS + 8 * qq + 4 * hi + 32 * t);
                            const uint2 u2 = *reinterpret_cast<const uint2*>(ra16 + MT * NS + s * NS + 8 * qq + 4 * hi + 32 * t);
                            float hb[4];
                            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(hb[0]) : "v"(u1.x), "v"(u2.x));
                            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hb[1]) : "v"(u1.x), "v"(u2.x));
                            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(hb[2]) : "v"(u1.y), "v"(u2.y));
                            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hb[3]) : "v"(u1.y), "v"(u2.y));
#pragma unroll
                            for (int j = 0; j < 4; j++) dz1[16 * t + 4 * qq + j] = acc[4 * qq + j] * __builtin_fmaf(hb[j] * -0x1p-7f, hb[j], 0x1p-7f);
                        }
                    }
                }
            }
            // ---------------- E2: dz1 terms -> RB once G has read the dz2 terms ----------------
            {
                uint32_t zq[2][16];
#pragma unroll
                for (int j = 0; j < 16; j++) split2(dz1[2 * j], dz1[2 * j + 1], zq[0][j], zq[1][j]);
                MG_TR_F(5, -1);
                MG_TIMED(3, proto_ok &= mg_wait_ge(cons, ev));
                MG_TR_F(6, -1);
#pragma unroll
                for (int term = 0; term < 2; term++)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            *reinterpret_cast<uint2*>(zimg + term * MT * NS + s * NS + 8 * q + 4 * hi + 32 * t) = make_uint2(zq[term][8 * t + 2 * q], zq[term][8 * t + 2 * q + 1]);
                mg_post(prod, ++ev, lane);
                MG_TR_F(7, 11);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // loss sums: a lane added a handful of samples in fp32; lanes, waves and workgroups are added in binary64
        const double sd0 = wave_sum_d_dpp((double)st0), sd1 = wave_sum_d_dpp((double)st1), sd2 = wave_sum_d_dpp((double)st2), sd3 = wave_sum_d_dpp((double)st3);
        if (lane == 0) { dred[wave * 4 + 0] = sd0; dred[wave * 4 + 1] = sd1; dred[wave * 4 + 2] = sd2; dred[wave * 4 + 3] = sd3; }
        if (!proto_ok && a.error_flag) atomicOr(a.error_flag, PPO_ERRFLAG_UPDATE_PROTOCOL);
#ifdef MG_STAMP
        if (blk == 0 && tid == 0 && a.stamps) {
            MG_ST_ADD(0, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
            for (int i = 1; i < 4; i++) MG_ST_ADD(i, (unsigned long long)(tstamp[i]));
        }
#endif
        __syncthreads();   // gradient images parked (each G wave in the regions of its own two F waves: no barrier in front of that)
    } else {
        // =================================================== G: the products that contract over samples ===================================================
        const int g = wave - MG_FW;
        f32x16 gW2[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) gW2[i][j][r] = 0.0f;
        f32x4 gW1m[4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) gW1m[b][r] = 0.0f;
        float gW3[AMAX], gb3[AMAX], gb2[2] = { 0.0f, 0.0f };
#pragma unroll
        for (int k = 0; k < AMAX; k++) { gW3[k] = 0.0f; gb3[k] = 0.0f; }
        int S_g2 = 100, S_g1 = 100;   // scale of (dW2, db2) / of (dW1, db1)
        const int j16 = lane & 15, kg = lane >> 4;
        // The two F waves are served in a FIXED cyclic order, the second a good event behind the first:
        //     E0(a, i)  E0(b, i)  E1(a, i)  E1(b, i)  E2(a, i)  E2(b, i)
        // (straight-line code per round: with a data-dependent order, or with conditions inside the round, the register allocator duplicates the 80
        // accumulators across the branches and spills; the SIMD partners a = g, b = g + 4 advance at the same pace anyway).
        auto tiles_of = [&](int fw) -> int {
            const int t0 = blk * MG_FW + fw;
            return t0 < n_tiles ? (n_tiles - 1 - t0) / tile_step + 1 : 0;
        };
        // the two F waves of a G wave sit on DIFFERENT SIMDs (waves w and w + 4 share one): the fixed order keeps a G wave's pair in step, and SIMD
        // partners in step would want the same pipe at the same time
        const int fw_b = MG_GW + ((g + 1) & (MG_GW - 1));
        const int nt_a = tiles_of(g), nt_b = tiles_of(fw_b);
        bool g_ok = true;
#ifdef MG_TRACE
        bool tr_round = false;
#endif
#define MG_REGION_PTRS(fw)                                                                         \
        float* wbase = smem + m.wave0 + mg_slot(fw) * m.wave_stride;                               \
        const uint16_t* ra16 = reinterpret_cast<const uint16_t*>(wbase + m.ra);                    \
        const float* img = wbase + m.rb;                                                           \
        const uint16_t* zimg = reinterpret_cast<const uint16_t*>(img);                             \
        const float* sDo = wbase + m.s_do;                                                         \
        lds_flag_t* cons = flags + MG_CONS + (fw);                                                 \
        (void)ra16; (void)zimg; (void)sDo; (void)img
        // E1 / E2 carry the tile's scale S: the accumulators the event adds to move to it first (powers of two: exact, whichever way)
        auto factor_to = [&](const int S_ev, int& S_acc) __attribute__((always_inline)) -> float {
            int d = S_ev - S_acc;
            d = d < -126 ? -126 : (d > 126 ? 126 : d);
            S_acc = S_ev;
            return pow2i(d);
        };
        // ---- E0: dW3[a][u = lane] += sum_s dOut[s][a] h2[s][u]; db3[a] += sum_s dOut[s][a] (every lane forms the same sum) ----
        auto serve_e0 = [&](const int fw, const uint32_t evno) __attribute__((always_inline)) {
            MG_REGION_PTRS(fw);
            if (fw == g) MG_TR_G(12);
            MG_TIMED(0, g_ok &= mg_wait_ge(flags + MG_PROD + fw, evno));
            MG_TR_G(fw == g ? 13 : 15);
            // the image's 32 rows into registers, then the acknowledgement (RB is what the F wave is waiting for; dOut is not rewritten before the
            // next tile's E0, which comes behind everything of this tile), then the products
            float hv[MT];
#pragma unroll
            for (int i = 0; i < MT; i++) hv[i] = img[i * LS + lane];
            mg_post(cons, evno, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < AOUT; k++) {
                float accw3 = 0.0f, accb3 = 0.0f;
#pragma unroll
                for (int i = 0; i < MT; i += 4) {
                    const float4 d = ld4(&sDo[k * MT + i]);
                    accw3 = __builtin_fmaf(d.x, hv[i], accw3); accw3 = __builtin_fmaf(d.y, hv[i + 1], accw3);
                    accw3 = __builtin_fmaf(d.z, hv[i + 2], accw3); accw3 = __builtin_fmaf(d.w, hv[i + 3], accw3);
                    accb3 += (d.x + d.y) + (d.z + d.w);
                }
                gW3[k] += accw3; gb3[k] += accb3;
            }
            MG_TR_G(fw == g ? 14 : 16);
        };
        // ---- E1: dW2[n][k] += sum_s dz2[s][n] h1[s][k], db2[n] += sum_s dz2[s][n]: both operands by transposing reads of the term images.
        //      Lane (n | k = lane & 31, hi) holds the terms of samples 16 c + 8 hi + 0..7 of unit (lane & 31) + 32 t ----
        auto serve_e1 = [&](const int fw, const uint32_t evno) __attribute__((always_inline)) {
            MG_REGION_PTRS(fw);
            MG_TIMED(0, g_ok &= mg_wait_ge(flags + MG_PROD + fw, evno));
            MG_TR_G(fw == g ? 17 : 20);
            {
                const int S_ev = __builtin_amdgcn_readfirstlane((int)flags[MG_SEV + fw]);
                if (S_ev != S_g2) {
                    const float f = factor_to(S_ev, S_g2);
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int jj = 0; jj < 2; jj++)
#pragma unroll
                            for (int r = 0; r < 16; r++) gW2[i][jj][r] *= f;
                    gb2[0] *= f; gb2[1] *= f;
                }
            }
            auto trf = [&](const uint16_t* im, int c, int t, int term) __attribute__((always_inline)) -> u32x4 {
                const int off = term * MT * NS + (16 * c + 8 * hi + tq) * NS + 32 * t + 16 * tb + 4 * tp;
                const uint2 l2 = lds_read_tr16(im + off), h2_ = lds_read_tr16(im + off + 4 * NS);
                return u32x4{ l2.x, l2.y, h2_.x, h2_.y };
            };
            u32x4 A[2][2][2];   // [chunk][tn][term]: dz2, from RB -- the region the F wave wants back: read whole, acknowledged, and only then h1
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int term = 0; term < 2; term++) A[c][t][term] = trf(zimg, c, t, term);
            mg_post(cons, evno, lane);
            MG_TR_G(fw == g ? 18 : 21);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                u32x4 B[2][2];   // [tk][term]: h1, from RA (held until E2 anyway)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int term = 0; term < 2; term++) B[t][term] = trf(ra16, c, t, term);
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    float cacc = 0.0f;
#pragma unroll
                    for (int term = 1; term >= 0; term--)
#pragma unroll
                        for (int d = 0; d < 4; d++) cacc = add_pair(A[c][tn][term][d], cacc);
                    gb2[tn] += cacc;
                }
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
#pragma unroll
                    for (int tk = 0; tk < 2; tk++)
                        gW2[tn][tk] = mfma_x2(A[c][tn][0], A[c][tn][1], B[tk][0], B[tk][1], gW2[tn][tk]);
                __builtin_amdgcn_sched_barrier(0);   // the second chunk's h1 reads are issued behind the first chunk's MFMAs, not hoisted above them (16 more registers)
            }
            MG_TR_G(fw == g ? 19 : 22);
        };
        // ---- E2: c 2^-7 dW1[u][o] += sum_s dz1[s][u] x[s][o], db1 as the column of the constant 1: 12 MFMAs of 16 x 16 x 32.
        //      B operand: lane (j = lane & 15, kg) holds the x terms of samples 8 kg + 0..7, column j.  The 16-lane group reads a 4-sample x 16-column
        //      block: the lanes of column cell 0 point at the row's observation cell (columns 64 .. 67 of the h1 term image), cell 1 at the constant
        //      {1, 0, 0, 0} (first term, OBS == 4 only: with OBS == 2 the 1 sits in the observation cell), the others at zeros ----
        auto serve_e2 = [&](const int fw, const uint32_t evno) __attribute__((always_inline)) {
            MG_REGION_PTRS(fw);
            MG_TIMED(0, g_ok &= mg_wait_ge(flags + MG_PROD + fw, evno));
            {
                const int S_ev = __builtin_amdgcn_readfirstlane((int)flags[MG_SEV + fw]);
                if (S_ev != S_g1) {
                    const float f = factor_to(S_ev, S_g1);
#pragma unroll
                    for (int b = 0; b < 4; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) gW1m[b][r] *= f;
                }
            }
            u32x4 xb[2];
#pragma unroll
            for (int term = 0; term < 2; term++) {
                const uint16_t* cell = reinterpret_cast<const uint16_t*>(smem + m.flags) + 2 * ((OBS == 4 && tp == 1 && term == 0) ? MG_ONE : MG_ZERO);
                const uint16_t* q0 = tp == 0 ? ra16 + term * MT * NS + (8 * kg + tq) * NS + 64 : cell;
                const uint16_t* q1 = tp == 0 ? ra16 + term * MT * NS + (8 * kg + tq + 4) * NS + 64 : cell;
                const uint2 lo2 = lds_read_tr16(q0), hi2 = lds_read_tr16(q1);
                xb[term] = u32x4{ lo2.x, lo2.y, hi2.x, hi2.y };
            }
            u32x4 A1[4], A2[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint16_t* q = zimg + (8 * kg + tq) * NS + 16 * b + 4 * tp;
                const uint2 lo2 = lds_read_tr16(q), hi2 = lds_read_tr16(q + 4 * NS);
                const uint2 lo3 = lds_read_tr16(q + MT * NS), hi3 = lds_read_tr16(q + MT * NS + 4 * NS);
                A1[b] = u32x4{ lo2.x, lo2.y, hi2.x, hi2.y };
                A2[b] = u32x4{ lo3.x, lo3.y, hi3.x, hi3.y };
            }
            mg_post(cons, evno, lane);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                f32x4 acc4 = gW1m[b];
                acc4 = mfma16_f16(A2[b], xb[0], acc4);
                acc4 = mfma16_f16(A1[b], xb[1], acc4);
                acc4 = mfma16_f16(A1[b], xb[0], acc4);
                gW1m[b] = acc4;
            }
            if (fw != g) MG_TR_G(23);
        };
        __builtin_amdgcn_s_setprio(2);
        for (int i = 0; i < nt_a; i++) {   // nt_a >= nt_b (wave a's first tile comes first); nt_a - nt_b is 0 or 1
            const int nj = i < nt_b ? 2 : 1;
#ifdef MG_TRACE
            tr_round = i == 2;
#endif
#pragma nounroll
            for (int j = 0; j < nj; j++) serve_e0(j ? fw_b : g, 3u * i + 1u);
#pragma nounroll
            for (int j = 0; j < nj; j++) serve_e1(j ? fw_b : g, 3u * i + 2u);
#pragma nounroll
            for (int j = 0; j < nj; j++) serve_e2(j ? fw_b : g, 3u * i + 3u);
        }
#undef MG_REGION_PTRS
        __builtin_amdgcn_s_setprio(0);
        if (!g_ok && a.error_flag) atomicOr(a.error_flag, PPO_ERRFLAG_UPDATE_PROTOCOL);
#ifdef MG_STAMP
        if (blk == 0 && tid == MG_FW * 64 && a.stamps) {
            MG_ST_ADD(5, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
            MG_ST_ADD(6, (unsigned long long)(tstamp[0]));
        }
#endif
        // ---- the factors leave: 2^S (dW2, db2, dW1, db1), the fixed 2^-7 of the dz1 terms and the c of the scaled W2 (dW1, db1) ----
        gb2[0] += __shfl_xor(gb2[0], 32, 64); gb2[1] += __shfl_xor(gb2[1], 32, 64);
        {
            const float invS = pow2i(-S_g2);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) gW2[i][j][r] *= invS;
            gb2[0] *= invS; gb2[1] *= invS;
            const float f = pow2i(-S_g1) * 0x1p7f;
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) gW1m[b][r] = (gW1m[b][r] * f) * TANH_C_INV;
        }
        // this wave has served the last events of its two F waves: their regions (adjacent: mg_slot) are dead and nobody else's business
        float* red = smem + m.wave0 + 2 * g * m.wave_stride;
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int tk = 0; tk < 2; tk++)
#pragma unroll
                for (int r = 0; r < 16; r++) red[L.w2[NET] - base + umap(r, hi, tn) * 64 + s + 32 * tk] = gW2[tn][tk][r];
#pragma unroll
        for (int k = 0; k < AOUT; k++) red[L.w3[NET] - base + k * 64 + lane] = gW3[k];
        {   // column j16 of the [dW1 | db1] block: one address and one stride per lane, ONE predicate for the sixteen stores
            const int col0 = j16 < OBS ? L.w1[NET] - base + j16 : L.b1[NET] - base;
            const int ustride = j16 < OBS ? OBS : 1;
            if (j16 <= OBS) {
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) red[col0 + (16 * b + 4 * kg + r) * ustride] = gW1m[b][r];
            }
        }
        if (hi == 0) { red[L.b2[NET] - base + s] = gb2[0]; red[L.b2[NET] - base + s + 32] = gb2[1]; }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < AOUT; k++) red[L.b3[NET] - base + k] = gb3[k];
        }
#ifdef MG_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (blk == 0 && tid == MG_FW * 64 && a.stamps) MG_ST_ADD(10, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));   // own image stored
#endif
        __syncthreads();
#ifdef MG_STAMP
        if (blk == 0 && tid == MG_FW * 64 && a.stamps) MG_ST_ADD(11, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
#endif
    }
    // ---- every thread adds the four gradient images in a fixed order into the workgroup's slab ----
    const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
    float* slab = a.slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blk) * Pmax;
    {
        constexpr int NSZ = 64 * OBS + 64 + 4096 + 64 + 64 * AOUT + AOUT;   // = L.net_size[NET] (checked by the launcher)
        constexpr int NE = (NSZ + MG_THREADS - 1) / MG_THREADS;
        float t[NE][MG_GW];
#pragma unroll
        for (int i = 0; i < NE; i++) {   // every read of a thread in flight before the first addition
            const int e = tid + i * MG_THREADS;
#pragma unroll
            for (int w = 0; w < MG_GW; w++) t[i][w] = smem[m.wave0 + 2 * w * m.wave_stride + (e < NSZ ? e : 0)];
        }
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int e = tid + i * MG_THREADS;
            float acc = t[i][0];
#pragma unroll
            for (int w = 1; w < MG_GW; w++) acc += t[i][w];
            if (e < NSZ) slab[e] = acc;
        }
    }
    if (tid < 4) {
        double* o = a.stat_slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blk) * 8;
        double t = dred[tid];
#pragma unroll
        for (int w = 1; w < MG_FW; w++) t += dred[4 * w + tid];
        o[tid] = t;
    }
#ifdef MG_STAMP
    if (blk == 0 && tid == 0 && a.stamps) {
        MG_ST_ADD(8, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));          // the whole workgroup, first instruction to last
        MG_ST_ADD(9, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t_real0));      // the same from the prologue's barrier on, in 100 MHz ticks (clock calibration)
    }
#endif
}

template <int DIST, int OBS, int AMAX>
__global__ __launch_bounds__(MG_THREADS, 1) void fwd_bwd_mfma_ws_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;   // 1-D grid: the first n_blocks[0] workgroups run the critic, the rest the actor
    if (b < a.n_blocks[0]) mg_body<0, DIST, OBS, 1>(a, smem, b, a.n_blocks[0]);
    else mg_body<1, DIST, OBS, AMAX>(a, smem, b - a.n_blocks[0], a.n_blocks[1]);
}

// One 32-byte record per sample and net for the update kernel's gather (K5): critic {obs[0..3], return, old value, 0, 0}, actor
// {obs[0..3], old log-prob, advantage, actions (8 bits per head), mask bits} (obs zero-padded to 4).  The flattened rollout buffers
// stay what the C-ABI exposes (PPO_Discrete.cpp:557-562); this is the layout the 40 permuted passes of an update read.  The kernel
// reads returns and values anyway, so it also leaves the partial sums of the explained variance (K11, :647-648).
template <int OBS>
__global__ __launch_bounds__(256) void pack_records_kernel(const float* __restrict__ obs, const int32_t* __restrict__ actions, int n_heads,
                                                           const uint8_t* __restrict__ masks, int A, const float* __restrict__ logprobs,
                                                           const float* __restrict__ advantages, const float* __restrict__ returns,
                                                           const float* __restrict__ values, int64_t B, float4* __restrict__ rec_critic,
                                                           float4* __restrict__ rec_actor, double* __restrict__ ev_out, int32_t* error_flag) {
    __shared__ double red[4][4];
    double sy = 0, sy2 = 0, sd = 0, sd2 = 0;
    float x_absmax = 0.0f;   // the wave-specialised update kernel cuts the observation into fp16 terms: its range is checked here, once per update, not assumed
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (OBS == 4) x = *reinterpret_cast<const float4*>(obs + i * 4);
        else { const float2 v = *reinterpret_cast<const float2*>(obs + i * 2); x.x = v.x; x.y = v.y; }
        uint32_t ab = 0u, mb = 0xffffffffu;
        for (int h = 0; h < n_heads; h++) ab |= ((uint32_t)actions[i * n_heads + h] & 0xffu) << (8 * h);
        if (masks) { mb = 0u; for (int k = 0; k < A; k++) mb |= (masks[i * A + k] ? 1u : 0u) << k; }
        x_absmax = fmaxf(fmaxf(x_absmax, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
        const float R = returns[i], V = values[i];
        // the two nets' records of a sample are the two halves of ONE 64-byte sector (rec_actor = rec_critic + 2 float4; a sample is 4 float4 apart in both): the
        // critic's and the actor's workgroups that gather the same sample run in the same launch on the same XCD, so the second of them finds the sector in L2 --
        // as two separate arrays every 32-byte gather fetched a 64-byte sector of its own (21.5 MB per launch against 13.6 algorithmic)
        rec_critic[4 * i] = x;
        rec_critic[4 * i + 1] = make_float4(R, V, 0.0f, 0.0f);
        rec_actor[4 * i] = x;
        rec_actor[4 * i + 1] = make_float4(logprobs[i], advantages[i], u2f(ab), u2f(mb));
        const double y = R, d = (double)(R - V);
        sy += y; sy2 += y * y; sd += d; sd2 += d * d;
    }
    if (!(x_absmax < 65504.0f) && error_flag) atomicOr(error_flag, PPO_ERRFLAG_UPDATE_RANGE);   // beyond fp16 (or NaN): reported by the next call that reads the flag (api.hip: PPO_ERRFLAG_UPDATE_RANGE)
    sy = wave_sum_d_dpp(sy); sy2 = wave_sum_d_dpp(sy2); sd = wave_sum_d_dpp(sd); sd2 = wave_sum_d_dpp(sd2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = sy; red[w][1] = sy2; red[w][2] = sd; red[w][3] = sd2; }
    __syncthreads();
    if (threadIdx.x < 4) ev_out[blockIdx.x * 4 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------
// Critic over rows [0, n0) of obs0 and [0, n1) of obs1 on the matrix cores (m_values[step] = Critic(obs[step]),
// PPO_Discrete.cpp:534-536, and the bootstrap value, :280): the forward half of mf_body<0> -- one wave per 32-row tile, layer 1 as
// fp32 MFMA, layer 2 as two-term fp16 products, head as a 32-term dot product per half-lane.  Rows are contiguous, so the
// "gather" is one coalesced 16-byte load per row.
// ---------------------------------------------------------------------------------------------------------
template <int OBS>
__global__ __launch_bounds__(256, 2) void values_mfma_kernel(const float* __restrict__ P, NetLayout L, const float* __restrict__ obs0, int64_t n0,
                                                             float* __restrict__ out0, const float* __restrict__ obs1, int64_t n1,
                                                             float* __restrict__ out1) {
    __shared__ __attribute__((aligned(16))) uint16_t sW2p[2 * 64 * WS];
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
            const float w = wv[i] * TANH_C;   // tanh_scaled: c is folded into W1, b1, W2, b2
            uint32_t p1, p2;
            split2(w, 0.0f, p1, p2);
            const int pf = n * WS + slot_of_unit(k);
            sW2p[pf] = (uint16_t)p1; sW2p[64 * WS + pf] = (uint16_t)p2;
        }
    }
    if (tid < 64) { sB1[tid] = P[L.b1[0] + tid] * TANH_C; sB2[tid] = P[L.b2[0] + tid] * TANH_C; sW3[tid] = P[L.w3[0] + tid]; }
    const float b3 = P[L.b3[0]];
    constexpr int L1S = (OBS + 1) / 2;
    float w1op[2][L1S];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int stp = 0; stp < L1S; stp++) {
            const int o = 2 * stp + hi;
            w1op[t][stp] = o < OBS ? P[L.w1[0] + (s + 32 * t) * OBS + o] * TANH_C : 0.0f;
        }
    __syncthreads();
    const int64_t n = n0 + n1;
    const int64_t n_tiles = (n + MT - 1) / MT;
    // the observation row of a tile is requested one tile ahead (unconditional loads: a missing row reads row 0 and is zeroed at use).  A lane
    // loads only what it feeds to layer 1 -- elements hi, 2 + hi, ... of its sample's row, the B operand of k step stp -- so nothing is selected
    // per lane afterwards (a select between array elements by `hi` sends the array to scratch memory)
    auto load_x = [&](int64_t tl, float* xo) {
        const int64_t r = tl * MT + s;
        const bool ok = r < n;
        const float* src = !ok ? obs0 : (r < n0 ? obs0 + r * OBS : obs1 + (r - n0) * OBS);
#pragma unroll
        for (int stp = 0; stp < L1S; stp++) xo[stp] = src[(2 * stp + 1 < OBS) ? 2 * stp + hi : 2 * stp];   // odd OBS: the last step's upper half is zeroed at use
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    float x_n[L1S];
    load_x((int64_t)blockIdx.x * 4 + wave, x_n);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += tile_step) {
        const int64_t row = tile * MT + s;
        const bool valid = row < n;
        float xb[L1S];
#pragma unroll
        for (int stp = 0; stp < L1S; stp++) xb[stp] = (valid && (2 * stp + 1 < OBS || hi == 0)) ? x_n[stp] : 0.0f;
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
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1op[t][stp], xb[stp], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) h1[16 * t + r] = tanh_scaled(acc[r]);
        }
        uint32_t hp[2][16];
#pragma unroll
        for (int j = 0; j < 16; j++) split2(h1[2 * j], h1[2 * j + 1], hp[0][j], hp[1][j]);
        auto aptr = [&](int g, int term) {
            return reinterpret_cast<const u32x4*>(sW2p + term * 64 * WS + (s + 32 * (g >> 2)) * WS + (g & 3) * 16 + hi * 8);
        };
        float part = 0.0f;
        f32x16 acc;
        u32x4 an1 = *aptr(0, 0), an2 = *aptr(0, 1);
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
            const u32x4 a1 = an1, a2 = an2;
            if (g + 1 < 8) { an1 = *aptr(g + 1, 0); an2 = *aptr(g + 1, 1); }
            MF_PIN();
            const u32x4 b1 = { hp[0][4 * c], hp[0][4 * c + 1], hp[0][4 * c + 2], hp[0][4 * c + 3] };
            const u32x4 b2 = { hp[1][4 * c], hp[1][4 * c + 1], hp[1][4 * c + 2], hp[1][4 * c + 3] };
            acc = mfma_x2(a1, a2, b1, b2, acc);
            if (c == 3) {
#pragma unroll
                for (int q = 0; q < 4; q++) {   // units 8q + 4hi + 32t .. +3 are registers 4q .. 4q+3
                    const float4 w = ld4(&sW3[8 * q + 4 * hi + 32 * t]);
                    part = __builtin_fmaf(tanh_scaled(acc[4 * q]), w.x, part); part = __builtin_fmaf(tanh_scaled(acc[4 * q + 1]), w.y, part);
                    part = __builtin_fmaf(tanh_scaled(acc[4 * q + 2]), w.z, part); part = __builtin_fmaf(tanh_scaled(acc[4 * q + 3]), w.w, part);
                }
            }
        }
        const float other = __shfl_xor(part, 32, 64);
        const float v = (part + other) + b3;
        if (valid && hi == 0) { if (row < n0) out0[row] = v; else out1[row - n0] = v; }
    }
}

}  // namespace

// One 8-wave workgroup per CU (256 vector registers per wave: two waves per SIMD is all a CU holds; LDS ~95 KB), half of them per net.
void update_blocks_mfma(int M, int n_blocks[2]) {
    const int total = 256;                                       // resident workgroups of the chip
    const int tiles = (M + MT - 1) / MT;
    const int per_net = (tiles + MF_WAVES - 1) / MF_WAVES;       // workgroups that still get a tile per wave
    n_blocks[0] = n_blocks[1] = per_net <= total / 2 ? (per_net > 0 ? per_net : 1) : total / 2;
}

hipError_t launch_minibatch_fwd_bwd_mfma(const UpdateArgs& a, hipStream_t s) {
    if (a.M <= 0 || !a.rec_critic || !a.rec_actor) return hipErrorInvalidValue;
    if (a.L.act > 4) return hipErrorNotSupported;  // wider heads run on the VALU kernel
    const int aout = a.L.act > 1 ? a.L.act : 1;
    size_t shmem = (size_t)mf_smem(a.L.obs, aout).total * sizeof(float);
    {   // the epilogue parks one gradient image per wave (+ 4 doubles of loss sums each) in the same block
        const int nmax = a.L.net_size[0] > a.L.net_size[1] ? a.L.net_size[0] : a.L.net_size[1];
        size_t dred_off = (size_t)MF_WAVES * ((nmax + 3) & ~3) + 4;
        if (dred_off < (size_t)mf_smem(a.L.obs, aout).total) dred_off = (size_t)mf_smem(a.L.obs, aout).total;
        const size_t need = (dred_off + 2 + 8 * MF_WAVES + 8) * sizeof(float);
        if (need > shmem) shmem = need;
    }
    if (shmem > 160 * 1024) return hipErrorNotSupported;
    const dim3 grid((unsigned)(a.n_blocks[0] + a.n_blocks[1])), block(MF_THREADS);
#define PPO_LAUNCH_MF(DIST, OBS, AMAX, EXACT)                                                                          \
    do {                                                                                                               \
        static std::atomic<unsigned long long> lds_ok{0}, lds_ok_stamp{0};                                              \
        hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void*>(&fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, false>)); \
        if (e == hipSuccess && a.stamps) e = allow_dynamic_lds(lds_ok_stamp, reinterpret_cast<const void*>(&fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, true>)); \
        if (e != hipSuccess) return e;                                                                                 \
        if (a.stamps) hipLaunchKernelGGL((fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, true>), grid, block, shmem, s, a); \
        else hipLaunchKernelGGL((fwd_bwd_mfma_kernel<DIST, OBS, AMAX, EXACT, false>), grid, block, shmem, s, a);       \
    } while (0)
    // the reference's two shapes get fully folded head code: CartPole (obs 4, one head of 2) and MountainCar (obs 2, one masked
    // head of 3); anything else with <= 4 logits runs the generic variant
    const bool one = a.L.n_heads == 1;
    // the reference's two shapes: the wave-specialised kernel (twelve waves, 160 KB of LDS)
#ifdef MG_STAMP
    if (!a.single_wave && one) {
#else
    if (!a.single_wave && !a.stamps && one) {
#endif
        const bool cart = a.L.obs == 4 && a.L.act == 2 && a.hp.dist_kind == PPO_DIST_CATEGORICAL;
        const bool mcar = a.L.obs == 2 && a.L.act == 3 && a.hp.dist_kind == PPO_DIST_MASKED;
        const int body = 64 * a.L.obs + 64 + 4096 + 64;   // the kernel's epilogue has the two nets' sizes as compile-time constants
        // ... and a G wave parks its gradient image in the two regions of its F waves
        const bool sizes = a.L.net_size[0] == body + 64 + 1 && a.L.net_size[1] == body + 64 * a.L.act + a.L.act &&
                           2 * mg_smem(a.L.obs, a.L.act).wave_stride >= a.L.net_size[1] && 2 * mg_smem(a.L.obs, 1).wave_stride >= a.L.net_size[0];
        if ((cart || mcar) && sizes) {
            const size_t ws_shmem = (size_t)mg_smem(a.L.obs, a.L.act).total * sizeof(float);
            if (ws_shmem > 160 * 1024) return hipErrorNotSupported;
            const dim3 ws_block(MG_THREADS);
            static std::atomic<unsigned long long> ws_ok_c{0}, ws_ok_m{0};
            if (cart) {
                hipError_t e = allow_dynamic_lds(ws_ok_c, reinterpret_cast<const void*>(&fwd_bwd_mfma_ws_kernel<PPO_DIST_CATEGORICAL, 4, 2>));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((fwd_bwd_mfma_ws_kernel<PPO_DIST_CATEGORICAL, 4, 2>), grid, ws_block, ws_shmem, s, a);
            } else {
                hipError_t e = allow_dynamic_lds(ws_ok_m, reinterpret_cast<const void*>(&fwd_bwd_mfma_ws_kernel<PPO_DIST_MASKED, 2, 3>));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((fwd_bwd_mfma_ws_kernel<PPO_DIST_MASKED, 2, 3>), grid, ws_block, ws_shmem, s, a);
            }
            return hipGetLastError();
        }
    }
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
    return hipGetLastError();
}

// rec_critic / rec_actor: ONE array [B][16] floats, the critic's record in floats 0 .. 7 of a sample and the actor's in 8 .. 15 (rec_actor = rec_critic + 8); ev_sums: [PPO_EV_BLOCKS][4] partial sums of the explained variance
hipError_t launch_pack_records(const NetLayout& L, const float* obs, const int32_t* actions, const uint8_t* masks, const float* logprobs,
                               const float* advantages, const float* returns, const float* values, int64_t B, float* rec_critic, float* rec_actor,
                               double* ev_sums, int32_t* error_flag, hipStream_t s) {
    if (L.act > 4 || L.n_heads > 4) return hipErrorNotSupported;
    const dim3 grid(PPO_EV_BLOCKS), block(256);
    if (L.obs == 4)
        hipLaunchKernelGGL((pack_records_kernel<4>), grid, block, 0, s, obs, actions, L.n_heads, masks, L.act, logprobs, advantages, returns, values, B,
                           reinterpret_cast<float4*>(rec_critic), reinterpret_cast<float4*>(rec_actor), ev_sums, error_flag);
    else if (L.obs == 2)
        hipLaunchKernelGGL((pack_records_kernel<2>), grid, block, 0, s, obs, actions, L.n_heads, masks, L.act, logprobs, advantages, returns, values, B,
                           reinterpret_cast<float4*>(rec_critic), reinterpret_cast<float4*>(rec_actor), ev_sums, error_flag);
    else
        return hipErrorNotSupported;
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
