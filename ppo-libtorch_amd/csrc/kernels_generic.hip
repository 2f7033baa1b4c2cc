// ppo-libtorch_amd/csrc/kernels_generic.hip -- networks other than the reference's 2 x 64 (see generic.hpp).
//
// Layers are the hand-written matrix-core products of kernels_gemm.hip (bias / tanh / tanh' fused into their epilogues); this file holds
// the glue that is not a GEMM, written for wave64: categorical heads (one thread per row), the PPO loss and its gradient, the gather,
// clip + AdamW over an arbitrary tensor list, and the synthetic environment of BASELINE configs[4].
#include <string>

#include "generic.hpp"

namespace {

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// One thread per row: per-head (masked) categorical over the row's logits; samples (Philox, same keying as the specialised path:
// counter (row, step / 4, head, 0), word step % 4) unless an action is forced.  Agent.cpp:137-170.
template <int DIST>
__global__ __launch_bounds__(128) void heads_kernel(GenLayout L, const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                    const int64_t* __restrict__ forced, int64_t n, int64_t seed, int64_t row_offset, int64_t step_index,
                                                    int64_t* action, float* logprob, float* entropy) {
    const int64_t r = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (r >= n) return;
    float lp_sum = 0.0f, en_sum = 0.0f;
    int off = 0;
    for (int h = 0; h < L.n_heads; h++) {
        const int A = L.head_dims[h];
        float z[PPO_MAX_ACT], p[PPO_MAX_ACT];
        for (int k = 0; k < A; k++) z[k] = logits[r * L.act + off + k];
        const uint8_t* mrow = (DIST == PPO_DIST_MASKED && mask) ? mask + r * L.act + off : nullptr;
        const float en = categorical_head<DIST>(z, p, mrow, A);
        int a;
        if (forced) {
            a = (int)forced[r * L.n_heads + h];
        } else {
            const uint4 w = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)(row_offset + r), (uint32_t)(step_index >> 2), (uint32_t)h, 0u);
            const uint32_t ws = (step_index & 3) == 0 ? w.x : ((step_index & 3) == 1 ? w.y : ((step_index & 3) == 2 ? w.z : w.w));
            a = sample_head(p, A, (float)(ws >> 8) * 0x1p-24f);
        }
        if (action) action[r * L.n_heads + h] = a;
        float lp = 0.0f;
        for (int k = 0; k < A; k++) if (k == a) lp = z[k];
        if (h == 0) { lp_sum = lp; en_sum = en; } else { lp_sum += lp; en_sum += en; }   // stack(...).sum(0), Agent.cpp:165-168
        off += A;
    }
    if (logprob) logprob[r] = lp_sum;
    if (entropy) entropy[r] = en_sum;
}

// Minibatch gather (PPO_Discrete.cpp:576-582): observation rows and per-row scalars into dense workspaces.
__global__ __launch_bounds__(256) void gather_kernel(GenLayout L, const float* __restrict__ obs, const int32_t* __restrict__ actions,
                                                     const uint8_t* __restrict__ masks, const float* __restrict__ logprobs, const float* __restrict__ adv,
                                                     const float* __restrict__ ret, const float* __restrict__ values, const int32_t* __restrict__ idx,
                                                     int64_t M, float* xin, uint16_t* xin_bf, int ld_bf, int32_t* row_act, uint8_t* row_mask, float* f0, float* f1,
                                                     float* f2, float* f3) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per row
    if (r >= M) return;
    const int64_t src = idx[r];
    if (xin_bf) {   // bf16 storage: rounded to nearest even, the row's padding stays zero
        if ((L.obs & 3) == 0) {   // 16-byte loads, 8-byte stores (a row of 376 floats: two passes of the wave instead of six)
            typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
            for (int o = 4 * lane; o < L.obs; o += 256) {
                const f32x4a4 v = *reinterpret_cast<const f32x4a4*>(obs + src * L.obs + o);
                const bf16x2v lo = { (__bf16)v[0], (__bf16)v[1] }, hi = { (__bf16)v[2], (__bf16)v[3] };
                *reinterpret_cast<uint2*>(xin_bf + r * ld_bf + o) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
            }
        } else {
            for (int o = lane; o < L.obs; o += 64) { const __bf16 b = (__bf16)obs[src * L.obs + o]; xin_bf[r * ld_bf + o] = __builtin_bit_cast(uint16_t, b); }
        }
    } else {
        for (int o = lane; o < L.obs; o += 64) xin[r * L.obs + o] = obs[src * L.obs + o];
    }
    if (lane < L.n_heads) row_act[r * L.n_heads + lane] = actions[src * L.n_heads + lane];
    if (masks && lane < L.act) row_mask[r * L.act + lane] = masks[src * L.act + lane];
    if (lane == 0) { f0[r] = logprobs[src]; f1[r] = adv[src]; f2[r] = ret[src]; f3[r] = values[src]; }
}

// PPO loss and its gradient w.r.t. logits / value, one thread per row (PPO_Discrete.cpp:585-631 and the autograd of it; the formulas
// are those of the specialised kernels).  Block partial sums of {pg, entropy, kl, clip count, value loss} go to loss_part.
template <int DIST>
__global__ __launch_bounds__(256) void loss_kernel(GenLayout L, LossParams hp, const float* __restrict__ logits, const float* __restrict__ val,
                                                   const int32_t* __restrict__ row_act, const uint8_t* __restrict__ row_mask,
                                                   const float* __restrict__ oldlp, const float* __restrict__ advs, const float* __restrict__ rets,
                                                   const float* __restrict__ oldv, int64_t M, float invM, const float* __restrict__ stat2,
                                                   float* dlogits, float* dval, double* loss_part, uint16_t* dlogits_bf, uint16_t* dval_bf,
                                                   float* head_db_part) {
    // bf16 storage (dlogits_bf != nullptr): the two head gradients leave as bf16 rows of pitch 128 (the operand of the head layers' backward
    // products), and the block's f32 column sums of them -- the head layers' bias gradients -- as head_db_part[block][act + 1]
    __shared__ double red[5][4];
    __shared__ float sdb[4][PPO_MAX_ACT + 1];
    float dbs[PPO_MAX_ACT + 1];
    for (int k = 0; k <= PPO_MAX_ACT; k++) dbs[k] = 0.0f;
    double s[5] = { 0, 0, 0, 0, 0 };
    const float mean_f = stat2[0], inv_std = stat2[1];   // advantage mean and 1 / (std + 1e-8) of the (global) minibatch
    const float clip = hp.clip_coef, lo = 1 - clip, hi_c = 1 + clip;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += (int64_t)gridDim.x * 256) {
        float nlp = 0.0f, ent = 0.0f;
        float z[PPO_MAX_ACT], p[PPO_MAX_ACT], headH[PPO_MAX_HEADS];
        int off = 0;
        for (int h = 0; h < L.n_heads; h++) {
            const int A = L.head_dims[h];
            for (int k = 0; k < A; k++) z[off + k] = logits[r * L.act + off + k];
            const uint8_t* mrow = (DIST == PPO_DIST_MASKED && row_mask) ? row_mask + r * L.act + off : nullptr;
            // bf16 storage: log-softmax on the hardware exp2 / log2 units (~1 ULP each; the logits themselves carry bf16 rounding)
            headH[h] = dlogits_bf ? categorical_head_fast<DIST>(z + off, p + off, mrow, A) : categorical_head<DIST>(z + off, p + off, mrow, A);
            const int a = row_act[r * L.n_heads + h];
            float lp = 0.0f;
            for (int k = 0; k < A; k++) if (k == a) lp = z[off + k];
            if (h == 0) { nlp = lp; ent = headH[h]; } else { nlp += lp; ent += headH[h]; }
            off += A;
        }
        const float logratio = nlp - oldlp[r];
        const float ratio = dlogits_bf ? fast_exp(logratio) : expf(logratio);
        float adv = advs[r];
        if (hp.norm_adv) adv = (adv - mean_f) * inv_std;
        const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
        const float l1 = -adv * ratio, l2 = -adv * rc;
        const bool inside = (ratio >= lo && ratio <= hi_c);
        float d_ratio;
        if (l1 > l2) d_ratio = -adv;
        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half/half
        const float g_nlp = invM * d_ratio * ratio;
        const float g_ent = -hp.ent_coef * invM;
        uint16_t drow[PPO_MAX_ACT];   // bf16 storage: the row's head gradients, stored below as 16-byte pieces
        off = 0;
        for (int h = 0; h < L.n_heads; h++) {
            const int A = L.head_dims[h];
            const int a = row_act[r * L.n_heads + h];
            for (int k = 0; k < A; k++) {
                const bool ok = !(DIST == PPO_DIST_MASKED && row_mask) || row_mask[r * L.act + off + k] != 0;
                float d = g_nlp * ((k == a ? 1.0f : 0.0f) - p[off + k]);
                if (DIST == PPO_DIST_MASKED) d += g_ent * (-p[off + k] * (z[off + k] + headH[h]));
                d = ok ? d : 0.0f;
                if (dlogits_bf) { const __bf16 b = (__bf16)d; drow[off + k] = __builtin_bit_cast(uint16_t, b); dbs[off + k] += d; }
                else dlogits[r * L.act + off + k] = d;
            }
            off += A;
        }
        if (dlogits_bf) {
#pragma unroll
            for (int c8 = 0; c8 < PPO_MAX_ACT / 8; c8++) {
                if (8 * c8 >= L.act) break;
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k0 = 8 * c8 + 2 * j;
                    w[j] = (k0 < L.act ? (uint32_t)drow[k0] : 0u) | ((k0 + 1 < L.act ? (uint32_t)drow[k0 + 1] : 0u) << 16);
                }
                *reinterpret_cast<uint4*>(dlogits_bf + r * 128 + 8 * c8) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        s[0] += (double)(l1 > l2 ? l1 : l2);
        s[1] += (double)ent;
        s[2] += (double)((ratio - 1.0f) - logratio);
        s[3] += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
        // value loss (:603-625)
        const float v = val[r], R = rets[r], vold = oldv[r];
        const float un = (v - R) * (v - R);
        float g_v, lossv;
        if (hp.clip_vloss) {
            const float dv = v - vold;
            const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
            const float vc = vold + dvc;
            const float cl = (vc - R) * (vc - R);
            lossv = un > cl ? un : cl;
            const bool vin = (dv >= -clip && dv <= clip);
            const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
            const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
            g_v = hp.vf_coef * 0.5f * invM * d;
        } else {
            lossv = un;
            g_v = hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
        }
        if (dval_bf) {
            const __bf16 b = (__bf16)g_v;
            *reinterpret_cast<uint4*>(dval_bf + r * 128) = make_uint4((uint32_t)__builtin_bit_cast(uint16_t, b), 0u, 0u, 0u);
            dbs[L.act] += g_v;
        }
        else dval[r] = g_v;
        s[4] += (double)lossv;
    }
    if (dlogits_bf) {
        for (int k = 0; k <= L.act; k++) {
            const float t = wave_sum(dbs[k]);
            if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][k] = t;
        }
    }
    for (int k = 0; k < 5; k++) {
        const double t = wave_sum_d_dpp(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) loss_part[blockIdx.x * 8 + threadIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
    if (dlogits_bf && threadIdx.x <= L.act)
        head_db_part[blockIdx.x * (L.act + 1) + threadIdx.x] = ((sdb[0][threadIdx.x] + sdb[1][threadIdx.x]) + sdb[2][threadIdx.x]) + sdb[3][threadIdx.x];
}

// The same loss for bf16 storage with every per-row array in REGISTERS: loss_kernel indexes z / p / dbs with run-time head offsets, which sends
// them to scratch memory (416 B per thread; 41 us for 65 536 rows).  Here the loops over heads (NH) and logits (AM) have compile-time bounds and a
// logit takes part in a head's pass when `off <= k < off + A` -- wave-uniform predicates.  Same operations in the same order per row as
// categorical_head_fast + loss_kernel, so both give the same bits.
// One 32-byte record per sample for the loss kernel of a step that reads its rows in place (once per update, with the bf16 observations):
// { old log-prob, advantage, return, old value | actions (8 bits per head), mask bits (1 = valid; all ones without masks), 0, 0 }.  Needs <= 4 heads of <= 256 actions
// and <= 32 logits (gen_rows_packable).
__global__ __launch_bounds__(256) void pack_rows_kernel(GenLayout L, const int32_t* __restrict__ actions, const uint8_t* __restrict__ masks,
                                                       const float* __restrict__ logprobs, const float* __restrict__ adv, const float* __restrict__ ret,
                                                       const float* __restrict__ values, int64_t B, float4* __restrict__ rec, const int32_t* __restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= B) return;
    const int64_t i = idx ? (int64_t)idx[j] : j;   // idx: the records of those B rows only, each in its own place
    uint32_t ab = 0u, mb = 0xffffffffu;
    for (int h = 0; h < L.n_heads; h++) ab |= ((uint32_t)actions[i * L.n_heads + h] & 0xffu) << (8 * h);
    if (masks) { mb = 0u; for (int k = 0; k < L.act; k++) mb |= (masks[i * L.act + k] ? 1u : 0u) << k; }
    rec[2 * i] = make_float4(logprobs[i], adv[i], ret[i], values[i]);
    rec[2 * i + 1] = make_float4(__builtin_bit_cast(float, ab), __builtin_bit_cast(float, mb), 0.0f, 0.0f);
}

template <int DIST, int NH, int AM>
__global__ __launch_bounds__(256) void loss_reg_kernel(GenLayout L, LossParams hp, const float* __restrict__ logits, const float* __restrict__ val,
                                                       const int32_t* __restrict__ row_act, const uint8_t* __restrict__ row_mask,
                                                       const float* __restrict__ oldlp, const float* __restrict__ advs, const float* __restrict__ rets,
                                                       const float* __restrict__ oldv, int64_t M, float invM, const AdvStat* __restrict__ adv_stat,
                                                       double global_M, double* loss_part, uint16_t* dlogits_bf, uint16_t* dval_bf, float* head_db_part,
                                                       const int32_t* __restrict__ idx, const float4* __restrict__ rec) {
    // idx != nullptr: row_act / row_mask / oldlp / advs / rets / oldv are the rollout's own arrays and row r of the minibatch is their row idx[r] (no gathered copies)
    __shared__ double red[5][4];
    __shared__ float sdb[4][AM + 1];
    float dbs[AM + 1];
#pragma unroll
    for (int k = 0; k <= AM; k++) dbs[k] = 0.0f;
    double s[5] = { 0, 0, 0, 0, 0 };
    // mean and 1 / (Bessel std + 1e-8) of the minibatch's advantages from the PPO_ADV_PARTS partial sums (adv_finish_kernel's arithmetic, formed
    // here by every thread from wave-uniform loads: one launch less per step); adv_stat == nullptr: no normalisation
    float mean_f = 0.0f, std_f = 0.0f;
    if (adv_stat) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += adv_stat[i].s1; t2 += adv_stat[i].s2; }
        const double mean = t1 / global_M;
        const double var = (t2 - t1 * mean) / (global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
    }
    const float inv_std = 1.0f / (std_f + 1e-8f);
    const float clip = hp.clip_coef, lo = 1 - clip, hi_c = 1 + clip;
    const int act = L.act, n_heads = L.n_heads;
    const bool masked = DIST == PPO_DIST_MASKED && row_mask != nullptr;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += (int64_t)gridDim.x * 256) {
        const int64_t q = idx ? (int64_t)idx[r] : r;   // the row of the per-sample arrays
        // everything the row needs is requested here, behind the index: one more memory round trip for the thread, not one per head (a thread is a row and a CU
        // holds four waves of this kernel: nothing else hides a load that is issued where it is used)
        // rec != nullptr: ONE 32-byte record per sample holds all of it (pack_rows_kernel, once per update): { old log-prob, advantage, return, old value,
        // actions (8 bits per head), mask bits } -- a single gathered load per row instead of six 4-byte ones + the action and mask bytes (each a sector of its own)
        int acts[NH];
        float in_oldlp, in_adv, in_ret, in_oldv;
        uint32_t mask_bits = 0xffffffffu;
        if (rec) {
            const float4 r0 = rec[2 * q], r1 = rec[2 * q + 1];
            in_oldlp = r0.x; in_adv = r0.y; in_ret = r0.z; in_oldv = r0.w;
            const uint32_t ab = __builtin_bit_cast(uint32_t, r1.x);
            mask_bits = __builtin_bit_cast(uint32_t, r1.y);
#pragma unroll
            for (int h = 0; h < NH; h++) acts[h] = h < 4 ? (int)((ab >> (8 * h)) & 0xffu) : 0;
        } else {
#pragma unroll
            for (int h = 0; h < NH; h++) acts[h] = h < n_heads ? row_act[q * n_heads + h] : 0;
            in_oldlp = oldlp[q]; in_adv = advs[q]; in_ret = rets[q]; in_oldv = oldv[q];
        }
        const float in_val = val[r];
        float z[AM], p[AM];
        bool ok[AM];
#pragma unroll
        for (int k = 0; k < AM; k++) {
            z[k] = k < act ? logits[r * act + k] : 0.0f;
            ok[k] = !masked || (k < act && (rec ? ((mask_bits >> k) & 1u) != 0 : row_mask[q * act + k] != 0));
            if (masked && !ok[k]) z[k] = -1e8f;
            p[k] = 0.0f;
        }
        float nlp = 0.0f, ent = 0.0f, headH[NH];
        int off = 0;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            headH[h] = 0.0f;
            if (h < n_heads) {
                const int A = L.head_dims[h];
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < AM; k++) if (k >= off && k < off + A) mx = z[k] > mx ? z[k] : mx;
                float se = 0.0f;
#pragma unroll
                for (int k = 0; k < AM; k++) if (k >= off && k < off + A) { p[k] = fast_exp(z[k] - mx); se += p[k]; }
                const float lse = fast_log(se) + mx;
                const float rse = __builtin_amdgcn_rcpf(se);
                const int a = acts[h];
                float e = 0.0f, lp = 0.0f;
#pragma unroll
                for (int k = 0; k < AM; k++) if (k >= off && k < off + A) {
                    z[k] = z[k] - lse;
                    p[k] = p[k] * rse;
                    if (DIST == PPO_DIST_CATEGORICAL) {
                        const float l = z[k] > 1.17549435e-38f ? z[k] : 1.17549435e-38f;
                        e += l * p[k];
                    } else {
                        const float plp = z[k] * p[k];
                        e += ok[k] ? plp : 0.0f;
                    }
                    if (k - off == a) lp = z[k];
                }
                headH[h] = -e;
                if (h == 0) { nlp = lp; ent = headH[h]; } else { nlp += lp; ent += headH[h]; }
                off += A;
            }
        }
        const float logratio = nlp - in_oldlp;
        const float ratio = fast_exp(logratio);
        float adv = in_adv;
        if (hp.norm_adv) adv = (adv - mean_f) * inv_std;
        const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
        const float l1 = -adv * ratio, l2 = -adv * rc;
        const bool inside = (ratio >= lo && ratio <= hi_c);
        float d_ratio;
        if (l1 > l2) d_ratio = -adv;
        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half/half
        const float g_nlp = invM * d_ratio * ratio;
        const float g_ent = -hp.ent_coef * invM;
        uint32_t drow[AM];   // the row's head gradients as bf16 bit patterns (zero beyond the policy's width)
#pragma unroll
        for (int k = 0; k < AM; k++) drow[k] = 0u;
        off = 0;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            if (h < n_heads) {
                const int A = L.head_dims[h];
                const int a = acts[h];
#pragma unroll
                for (int k = 0; k < AM; k++) if (k >= off && k < off + A) {
                    float d = g_nlp * ((k - off == a ? 1.0f : 0.0f) - p[k]);
                    if (DIST == PPO_DIST_MASKED) d += g_ent * (-p[k] * (z[k] + headH[h]));
                    d = ok[k] ? d : 0.0f;
                    const __bf16 b = (__bf16)d;
                    drow[k] = (uint32_t)__builtin_bit_cast(uint16_t, b);
                    dbs[k] += d;
                }
                off += A;
            }
        }
#pragma unroll
        for (int c8 = 0; c8 < AM / 8; c8++) {
            if (8 * c8 < act)
                *reinterpret_cast<uint4*>(dlogits_bf + r * 128 + 8 * c8) = make_uint4(drow[8 * c8] | (drow[8 * c8 + 1] << 16), drow[8 * c8 + 2] | (drow[8 * c8 + 3] << 16),
                                                                                     drow[8 * c8 + 4] | (drow[8 * c8 + 5] << 16), drow[8 * c8 + 6] | (drow[8 * c8 + 7] << 16));
        }
        s[0] += (double)(l1 > l2 ? l1 : l2);
        s[1] += (double)ent;
        s[2] += (double)((ratio - 1.0f) - logratio);
        s[3] += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
        // value loss (:603-625)
        const float v = in_val, R = in_ret, vold = in_oldv;
        const float un = (v - R) * (v - R);
        float g_v, lossv;
        if (hp.clip_vloss) {
            const float dv = v - vold;
            const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
            const float vc = vold + dvc;
            const float cl = (vc - R) * (vc - R);
            lossv = un > cl ? un : cl;
            const bool vin = (dv >= -clip && dv <= clip);
            const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
            const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
            g_v = hp.vf_coef * 0.5f * invM * d;
        } else {
            lossv = un;
            g_v = hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
        }
        const __bf16 bv = (__bf16)g_v;
        *reinterpret_cast<uint4*>(dval_bf + r * 128) = make_uint4((uint32_t)__builtin_bit_cast(uint16_t, bv), 0u, 0u, 0u);
        dbs[AM] += g_v;
        s[4] += (double)lossv;
    }
#pragma unroll
    for (int k = 0; k <= AM; k++) {
        if (k < act || k == AM) {
            const float t = wave_sum(dbs[k]);
            if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][k] = t;
        }
    }
    for (int k = 0; k < 5; k++) {
        const double t = wave_sum_d_dpp(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) loss_part[blockIdx.x * 8 + threadIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
    if ((int)threadIdx.x <= act) {   // head_db_part[block][0 .. act - 1] = d(logits) column sums, [act] = the d(value) sum
        const int k = (int)threadIdx.x < act ? (int)threadIdx.x : AM;
        head_db_part[blockIdx.x * (act + 1) + threadIdx.x] = ((sdb[0][k] + sdb[1][k]) + sdb[2][k]) + sdb[3][k];
    }
}

// loss_reg_kernel for the shapes of a step that reads packed row records (rec, idx) with at most 4 heads of at most 4 logits (configs[4]: [3, 3, 3, 2]): the same
// operations in the same order per row, but every per-head loop runs over the head's own <= 4 slots (compile-time indices, wave-uniform "slot exists" predicates)
// instead of all AM logits under a per-element range test -- loss_reg_kernel<., 4, 16> is 5 500 instructions per row, three quarters of them predication,
// selects and scalar-register spills, and a CU holds four waves of it: the kernel was bound by its own instruction stream (20 us).  The head gradients leave as
// 2-byte stores at their columns; the columns between the policy's width and 32 are never written (zero since allocation), as the consumer expects.
template <int DIST>
__global__ __launch_bounds__(256) void loss_small_kernel(GenLayout L, LossParams hp, const float* __restrict__ logits, const float* __restrict__ val, int64_t M,
                                                         float invM, const AdvStat* __restrict__ adv_stat, double global_M, double* loss_part, uint16_t* dlogits_bf,
                                                         uint16_t* dval_bf, float* head_db_part, const int32_t* __restrict__ idx, const float4* __restrict__ rec) {
    constexpr int NH = 4, W = 4;
    __shared__ double red[5][4];
    __shared__ float sdb[4][NH * W + 1];
    float dbs[NH][W], dbv = 0.0f;
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int j = 0; j < W; j++) dbs[h][j] = 0.0f;
    double s[5] = { 0, 0, 0, 0, 0 };
    float mean_f = 0.0f, std_f = 0.0f;
    if (adv_stat) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += adv_stat[i].s1; t2 += adv_stat[i].s2; }
        const double mean = t1 / global_M;
        const double var = (t2 - t1 * mean) / (global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
    }
    const float inv_std = 1.0f / (std_f + 1e-8f);
    const float clip = hp.clip_coef, lo = 1 - clip, hi_c = 1 + clip;
    const int act = L.act;
    int A[NH], off[NH];   // wave-uniform: a head's width (0: no such head) and first column
    {
        int o = 0;
#pragma unroll
        for (int h = 0; h < NH; h++) { A[h] = h < L.n_heads ? L.head_dims[h] : 0; off[h] = o; o += A[h]; }
    }
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += (int64_t)gridDim.x * 256) {
        const int64_t q = (int64_t)idx[r];
        const float4 r0 = rec[2 * q], r1 = rec[2 * q + 1];
        const float in_oldlp = r0.x, in_adv = r0.y, in_ret = r0.z, in_oldv = r0.w, in_val = val[r];
        const uint32_t ab = __builtin_bit_cast(uint32_t, r1.x), mask_bits = DIST == PPO_DIST_MASKED ? __builtin_bit_cast(uint32_t, r1.y) : 0xffffffffu;
        float z[NH][W], p[NH][W];
        bool ok[NH][W];
        // all sixteen slots are requested unconditionally (a slot that does not exist re-reads the row's last logit): behind a per-slot branch the loads of a
        // head left in a basic block of their own, each with its own wait -- one memory round trip per head for a thread nothing else covers
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int j = 0; j < W; j++) { const int k = off[h] + j; z[h][j] = logits[r * act + (k < act ? k : act - 1)]; }
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int j = 0; j < W; j++) {
                const bool in = j < A[h];
                z[h][j] = in ? z[h][j] : 0.0f;
                ok[h][j] = in && ((mask_bits >> (off[h] + j)) & 1u) != 0;
                if (DIST == PPO_DIST_MASKED && in && !ok[h][j]) z[h][j] = -1e8f;
                p[h][j] = 0.0f;
            }
        float nlp = 0.0f, ent = 0.0f, headH[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            headH[h] = 0.0f;
            if (A[h] > 0) {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < W; j++) if (j < A[h]) mx = z[h][j] > mx ? z[h][j] : mx;
                float se = 0.0f;
#pragma unroll
                for (int j = 0; j < W; j++) if (j < A[h]) { p[h][j] = fast_exp(z[h][j] - mx); se += p[h][j]; }
                const float lse = fast_log(se) + mx;
                const float rse = __builtin_amdgcn_rcpf(se);
                const int a = (int)((ab >> (8 * h)) & 0xffu);
                float e = 0.0f, lp = 0.0f;
#pragma unroll
                for (int j = 0; j < W; j++) if (j < A[h]) {
                    z[h][j] = z[h][j] - lse;
                    p[h][j] = p[h][j] * rse;
                    if (DIST == PPO_DIST_CATEGORICAL) {
                        const float l = z[h][j] > 1.17549435e-38f ? z[h][j] : 1.17549435e-38f;
                        e += l * p[h][j];
                    } else {
                        const float plp = z[h][j] * p[h][j];
                        e += ok[h][j] ? plp : 0.0f;
                    }
                    if (j == a) lp = z[h][j];
                }
                headH[h] = -e;
                if (h == 0) { nlp = lp; ent = headH[h]; } else { nlp += lp; ent += headH[h]; }
            }
        }
        const float logratio = nlp - in_oldlp;
        const float ratio = fast_exp(logratio);
        float adv = in_adv;
        if (hp.norm_adv) adv = (adv - mean_f) * inv_std;
        const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
        const float l1 = -adv * ratio, l2 = -adv * rc;
        const bool inside = (ratio >= lo && ratio <= hi_c);
        float d_ratio;
        if (l1 > l2) d_ratio = -adv;
        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half/half
        const float g_nlp = invM * d_ratio * ratio;
        const float g_ent = -hp.ent_coef * invM;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int a = (int)((ab >> (8 * h)) & 0xffu);
#pragma unroll
            for (int j = 0; j < W; j++) if (j < A[h]) {
                float d = g_nlp * ((j == a ? 1.0f : 0.0f) - p[h][j]);
                if (DIST == PPO_DIST_MASKED) d += g_ent * (-p[h][j] * (z[h][j] + headH[h]));
                d = (DIST != PPO_DIST_MASKED || ok[h][j]) ? d : 0.0f;
                const __bf16 b = (__bf16)d;
                dlogits_bf[r * 128 + off[h] + j] = __builtin_bit_cast(uint16_t, b);
                dbs[h][j] += d;
            }
        }
        s[0] += (double)(l1 > l2 ? l1 : l2);
        s[1] += (double)ent;
        s[2] += (double)((ratio - 1.0f) - logratio);
        s[3] += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
        // value loss (:603-625)
        const float v = in_val, R = in_ret, vold = in_oldv;
        const float un = (v - R) * (v - R);
        float g_v, lossv;
        if (hp.clip_vloss) {
            const float dv = v - vold;
            const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
            const float vc = vold + dvc;
            const float cl = (vc - R) * (vc - R);
            lossv = un > cl ? un : cl;
            const bool vin = (dv >= -clip && dv <= clip);
            const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
            const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
            g_v = hp.vf_coef * 0.5f * invM * d;
        } else {
            lossv = un;
            g_v = hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
        }
        const __bf16 bv = (__bf16)g_v;
        dval_bf[r * 128] = __builtin_bit_cast(uint16_t, bv);
        dbv += g_v;
        s[4] += (double)lossv;
    }
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int j = 0; j < W; j++)
            if (j < A[h]) {
                const float t = wave_sum(dbs[h][j]);
                if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][off[h] + j] = t;
            }
    {
        const float t = wave_sum(dbv);
        if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][NH * W] = t;
    }
    for (int k = 0; k < 5; k++) {
        const double t = wave_sum_d_dpp(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) loss_part[blockIdx.x * 8 + threadIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
    if ((int)threadIdx.x <= act) {   // head_db_part[block][0 .. act - 1] = d(logits) column sums, [act] = the d(value) sum
        const int k = (int)threadIdx.x < act ? (int)threadIdx.x : NH * W;
        head_db_part[blockIdx.x * (act + 1) + threadIdx.x] = ((sdb[0][k] + sdb[1][k]) + sdb[2][k]) + sdb[3][k];
    }
}

// The slot-wise loss with FOUR LANES PER ROW -- lane (row, head) -- and 1024 threads per workgroup: a row's heads run side by side (each lane the <= 4 slots of its
// head), the row's log-prob and entropy are added over the four lanes in head order (the order loss_reg_kernel adds them in), lane 0 of the row forms the ratio,
// the policy gradient's scalars and the value loss, and every lane writes its own head's gradient.  A third of the instructions per lane and four times the waves:
// with a row per thread a CU held four waves of this kernel and nothing covered a load or a transcendental (17.8 us for 65 536 rows).
template <int DIST>
__global__ __launch_bounds__(1024) void loss_lanes_kernel(GenLayout L, LossParams hp, const float* __restrict__ logits, const float* __restrict__ val, int64_t M,
                                                          float invM, const AdvStat* __restrict__ adv_stat, double global_M, double* loss_part, uint16_t* dlogits_bf,
                                                          uint16_t* dval_bf, float* head_db_part, const int32_t* __restrict__ idx, const float4* __restrict__ rec) {
    constexpr int NH = 4, W = 4, WAVES = 16;
    __shared__ double red[5][WAVES];
    __shared__ float sdb[WAVES][NH * W + 1];
    float dbs[W], dbv = 0.0f;
#pragma unroll
    for (int j = 0; j < W; j++) dbs[j] = 0.0f;
    double s[5] = { 0, 0, 0, 0, 0 };
    float mean_f = 0.0f, std_f = 0.0f;
    if (adv_stat) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += adv_stat[i].s1; t2 += adv_stat[i].s2; }
        const double mean = t1 / global_M;
        const double var = (t2 - t1 * mean) / (global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
    }
    const float inv_std = 1.0f / (std_f + 1e-8f);
    const float clip = hp.clip_coef, lo = 1 - clip, hi_c = 1 + clip;
    const int act = L.act, n_heads = L.n_heads;
    const int h = threadIdx.x & 3;                     // this lane's head
    int A = 0, off = 0;
#pragma unroll
    for (int hh = 0; hh < NH; hh++) { const int a = hh < n_heads ? L.head_dims[hh] : 0; if (hh < h) off += a; if (hh == h) A = a; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 256 + (threadIdx.x >> 2); r < M; r += (int64_t)gridDim.x * 256) {   // every row's four lanes take the same trips
        const int64_t q = (int64_t)idx[r];
        const float4 r0 = rec[2 * q], r1 = rec[2 * q + 1];
        const uint32_t ab = __builtin_bit_cast(uint32_t, r1.x), mask_bits = DIST == PPO_DIST_MASKED ? __builtin_bit_cast(uint32_t, r1.y) : 0xffffffffu;
        const float in_val = val[r];
        float z[W], p[W];
        bool ok[W];
#pragma unroll
        for (int j = 0; j < W; j++) { const int k = off + j; z[j] = logits[r * act + (k < act ? k : act - 1)]; }
#pragma unroll
        for (int j = 0; j < W; j++) {
            const bool in = j < A;
            ok[j] = in && ((mask_bits >> (off + j)) & 1u) != 0;
            z[j] = in ? ((DIST == PPO_DIST_MASKED && !ok[j]) ? -1e8f : z[j]) : -INFINITY;   // a slot that does not exist: exp -> 0, never the maximum
            p[j] = 0.0f;
        }
        float lp = 0.0f, hH = 0.0f;
        const int a = (int)((ab >> (8 * h)) & 0xffu);
        if (A > 0) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < W; j++) if (j < A) mx = z[j] > mx ? z[j] : mx;
            float se = 0.0f;
#pragma unroll
            for (int j = 0; j < W; j++) if (j < A) { p[j] = fast_exp(z[j] - mx); se += p[j]; }
            const float lse = fast_log(se) + mx;
            const float rse = __builtin_amdgcn_rcpf(se);
            float e = 0.0f;
#pragma unroll
            for (int j = 0; j < W; j++) if (j < A) {
                z[j] = z[j] - lse;
                p[j] = p[j] * rse;
                if (DIST == PPO_DIST_CATEGORICAL) {
                    const float l = z[j] > 1.17549435e-38f ? z[j] : 1.17549435e-38f;
                    e += l * p[j];
                } else {
                    const float plp = z[j] * p[j];
                    e += ok[j] ? plp : 0.0f;
                }
                if (j == a) lp = z[j];
            }
            hH = -e;
        }
        // the row's sums in head order, on every lane of the row (the four lanes are consecutive): ((h0 + h1) + h2) + h3 over the heads that exist
        const int base = lane & ~3;
        float nlp = __shfl(lp, base, 64), ent = __shfl(hH, base, 64);
#pragma unroll
        for (int hh = 1; hh < NH; hh++) {
            const float l2 = __shfl(lp, base + hh, 64), e2 = __shfl(hH, base + hh, 64);
            if (hh < n_heads) { nlp += l2; ent += e2; }
        }
        const float logratio = nlp - r0.x;
        const float ratio = fast_exp(logratio);
        float adv = r0.y;
        if (hp.norm_adv) adv = (adv - mean_f) * inv_std;
        const float rc = ratio < lo ? lo : (ratio > hi_c ? hi_c : ratio);
        const float l1 = -adv * ratio, l2 = -adv * rc;
        const bool inside = (ratio >= lo && ratio <= hi_c);
        float d_ratio;
        if (l1 > l2) d_ratio = -adv;
        else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
        else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);   // torch::max splits ties half/half
        const float g_nlp = invM * d_ratio * ratio;
        const float g_ent = -hp.ent_coef * invM;
#pragma unroll
        for (int j = 0; j < W; j++) if (j < A) {
            float d = g_nlp * ((j == a ? 1.0f : 0.0f) - p[j]);
            if (DIST == PPO_DIST_MASKED) d += g_ent * (-p[j] * (z[j] + hH));
            d = (DIST != PPO_DIST_MASKED || ok[j]) ? d : 0.0f;
            const __bf16 b = (__bf16)d;
            dlogits_bf[r * 128 + off + j] = __builtin_bit_cast(uint16_t, b);
            dbs[j] += d;
        }
        if (h == 0) {   // the row's scalars and the value loss (:603-625), once per row
            s[0] += (double)(l1 > l2 ? l1 : l2);
            s[1] += (double)ent;
            s[2] += (double)((ratio - 1.0f) - logratio);
            s[3] += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
            const float v = in_val, R = r0.z, vold = r0.w;
            const float un = (v - R) * (v - R);
            float g_v, lossv;
            if (hp.clip_vloss) {
                const float dv = v - vold;
                const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
                const float vc = vold + dvc;
                const float cl = (vc - R) * (vc - R);
                lossv = un > cl ? un : cl;
                const bool vin = (dv >= -clip && dv <= clip);
                const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
                const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);
                g_v = hp.vf_coef * 0.5f * invM * d;
            } else {
                lossv = un;
                g_v = hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
            }
            const __bf16 bv = (__bf16)g_v;
            dval_bf[r * 128] = __builtin_bit_cast(uint16_t, bv);
            dbv += g_v;
            s[4] += (double)lossv;
        }
    }
    // column sums of the head gradients: column off(hh) + j lives on the lanes of head hh
#pragma unroll
    for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int j = 0; j < W; j++) {
            const float t = wave_sum(h == hh ? dbs[j] : 0.0f);
            int o = 0, a = 0;
#pragma unroll
            for (int g = 0; g < NH; g++) { const int w = g < n_heads ? L.head_dims[g] : 0; if (g < hh) o += w; if (g == hh) a = w; }
            if (lane == 0 && j < a) sdb[wave][o + j] = t;
        }
    {
        const float t = wave_sum(dbv);
        if (lane == 0) sdb[wave][NH * W] = t;
    }
    for (int k = 0; k < 5; k++) {
        const double t = wave_sum_d_dpp(s[k]);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double t = 0.0;
        for (int w = 0; w < WAVES; w++) t += red[threadIdx.x][w];
        loss_part[blockIdx.x * 8 + threadIdx.x] = t;
    }
    if ((int)threadIdx.x <= act) {   // head_db_part[block][0 .. act - 1] = d(logits) column sums, [act] = the d(value) sum
        const int k = (int)threadIdx.x < act ? (int)threadIdx.x : NH * W;
        float t = 0.0f;
        for (int w = 0; w < WAVES; w++) t += sdb[w][k];
        head_db_part[blockIdx.x * (act + 1) + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void loss_sums_kernel(const double* __restrict__ loss_part, int blocks, double* sums_out, float* grads_tail) {
    __shared__ double red[5][4];
    for (int k = 0; k < 5; k++) {
        double v = 0.0;
        for (int b = threadIdx.x; b < blocks; b += 256) v += loss_part[b * 8 + k];
        v = wave_sum_d_dpp(v);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const double t = threadIdx.x < 5 ? ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3] : 0.0;
        sums_out[threadIdx.x] = t;
        grads_tail[threadIdx.x] = (float)t;   // float copies ride behind the gradient so ONE all-reduce carries both
    }
}

// clip_grad_norm_ (clip_grad.h:58-85) + AdamW over an arbitrary tensor list: one workgroup per tensor for the squared norms, then
// element-parallel AdamW where every thread forms the same clip coefficient.  Arithmetic per element = clip_adamw_kernel.
__global__ __launch_bounds__(256) void gen_norm_kernel(const float* __restrict__ grads, GenLayout L, double* __restrict__ norm2) {
    // grid (tensor, part): GEN_NORM_PARTS contiguous slices per tensor, added in order by the consumer
    __shared__ double red[4];
    const int t = blockIdx.x, part = blockIdx.y;
    const int t0 = L.tensor_off[t], len = L.tensor_off[t + 1] - t0;
    const int a0 = t0 + (int)(((int64_t)len * part) / GEN_NORM_PARTS), a1 = t0 + (int)(((int64_t)len * (part + 1)) / GEN_NORM_PARTS);
    double acc = 0.0;
    int p = a0 + threadIdx.x;
    for (; p + 3 * 256 < a1; p += 4 * 256) {   // four loads in flight per thread (a slice of a 256 x 376 tensor is 24 elements per thread)
        const float g0 = grads[p], g1 = grads[p + 256], g2 = grads[p + 512], g3 = grads[p + 768];
        acc += (double)g0 * (double)g0; acc += (double)g1 * (double)g1; acc += (double)g2 * (double)g2; acc += (double)g3 * (double)g3;
    }
    for (; p < a1; p += 256) acc += (double)grads[p] * (double)grads[p];
    acc = wave_sum_d_dpp(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) norm2[t * GEN_NORM_PARTS + part] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void gen_adamw_kernel(float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ exp_avg,
                                                        float* __restrict__ exp_avg_sq, GenLayout L, float max_norm, const double* __restrict__ norm2,
                                                        const AdamCoef* __restrict__ coef_p, const double* __restrict__ loss_sums, double global_M,
                                                        LossParams hp, int world, int do_step, StepStats* stats_out, double* clipfrac_accum) {
    // total norm: thread t adds tensor t's partial sums (all its loads in flight), thread 0 adds the tensors -- the same sums in the same order
    // as one thread doing all of it (which every thread used to do: 320 dependent loads in front of the element-wise step)
    __shared__ double s_n2[4 * GEN_MAX_LAYERS];
    __shared__ float s_total;
    if ((int)threadIdx.x < L.n_tensors) {
        double part[GEN_NORM_PARTS];
#pragma unroll
        for (int k = 0; k < GEN_NORM_PARTS; k++) part[k] = norm2[threadIdx.x * GEN_NORM_PARTS + k];
        double n2 = 0.0;
#pragma unroll
        for (int k = 0; k < GEN_NORM_PARTS; k++) n2 += part[k];
        const float nrm = (float)sqrt(n2);
        s_n2[threadIdx.x] = (double)nrm * nrm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int t = 0; t < L.n_tensors; t++) tot += s_n2[t];
        s_total = (float)sqrt(tot);
    }
    __syncthreads();
    const float total = s_total;
    float c = max_norm / (total + 1e-6f);
    if (c > 1.0f) c = 1.0f;
    const AdamCoef k = *coef_p;
    const float b1 = 0.9f, b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-5f;
    if (do_step) {
        for (int p = blockIdx.x * 256 + threadIdx.x; p < L.P; p += gridDim.x * 256) {
            const float g = grads[p] * c;
            const float pi = params[p] * k.decay;
            const float mi = __builtin_fmaf(g, omb1, exp_avg[p] * b1);
            const float vi = __builtin_fmaf(omb2 * g, g, exp_avg_sq[p] * b2);
            const float denom = sqrtf(vi) / k.sqrt_bc2 + eps;
            params[p] = pi + (k.neg_step * mi) / denom;
            exp_avg[p] = mi;
            exp_avg_sq[p] = vi;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double ls[5];
        for (int i = 0; i < 5; i++) ls[i] = world > 1 ? (double)grads[L.P + i] : loss_sums[i];
        const float pg = (float)(ls[0] / global_M), vl = 0.5f * (float)(ls[4] / global_M), el = (float)(ls[1] / global_M);
        StepStats o;
        o.pg_loss = pg; o.v_loss = vl; o.entropy_loss = el;
        o.approx_kl = (float)(ls[2] / global_M);
        o.clipfrac = (float)ls[3] / (float)global_M;
        o.loss = (pg - hp.ent_coef * el) + vl * hp.vf_coef;
        o.total_norm = total;
        o.pad = 0.0;
        *stats_out = o;
        if (clipfrac_accum && do_step) { clipfrac_accum[0] += o.clipfrac; clipfrac_accum[1] += 1.0; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Synthetic environment of BASELINE configs[4] (obs ~ N(0,1), reward ~ U(-1,1), done ~ Bernoulli(0.01), random masks with
// at least one valid action per head).  Memoryless and counter-based, so that the oracle restates it bit for bit with integer
// arithmetic only (oracle/ppo_oracle.c:orc_synthetic_*):
//   obs_j(env, step)  = (sum of the four bytes of Philox(seed; env, step, j / 4, 0x10)[j % 4] - 510) / sqrt(4 (256^2 - 1) / 12)
//   reward(env, step) = Philox(seed; env, step, 0, 0x11).x >> 8  scaled to [-1, 1);   done = (.y >> 8) < 0.01 * 2^24
//   mask_k(env, step) = bit k of Philox(seed; env, step, 0, 0x12).x, bit 0 of a head forced on when the head has none.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float syn_obs(int64_t seed, int64_t env, int64_t step, int j) {
    const uint4 w = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)env, (uint32_t)step, (uint32_t)(j >> 2), 0x10u);
    const uint32_t x = (j & 3) == 0 ? w.x : ((j & 3) == 1 ? w.y : ((j & 3) == 2 ? w.z : w.w));
    const int sum = (int)(x & 255u) + (int)((x >> 8) & 255u) + (int)((x >> 16) & 255u) + (int)(x >> 24);
    return (float)(sum - 510) * 0.0067658990621566772f;
}
__global__ __launch_bounds__(64) void synthetic_obs_kernel(GenLayout L, int N, int64_t seed, int64_t env_offset, int64_t step, float* obs_out, uint8_t* mask_out) {
    const int env = blockIdx.x;
    if (env >= N) return;
    const int64_t eg = env_offset + env;
    for (int j = threadIdx.x; j < L.obs; j += 64) obs_out[(size_t)env * L.obs + j] = syn_obs(seed, eg, step, j);
    if (mask_out && threadIdx.x == 0) {
        const uint32_t bits = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)eg, (uint32_t)step, 0u, 0x12u).x;
        int off = 0;
        for (int h = 0; h < L.n_heads; h++) {
            bool any = false;
            for (int k = 0; k < L.head_dims[h]; k++) { const bool v = (bits >> (off + k)) & 1u; any |= v; mask_out[(size_t)env * L.act + off + k] = v ? 1 : 0; }
            if (!any) mask_out[(size_t)env * L.act + off] = 1;
            off += L.head_dims[h];
        }
    }
}
__global__ __launch_bounds__(256) void synthetic_transition_kernel(int N, int64_t seed, int64_t env_offset, int64_t step, int max_episode_steps,
                                                                   int32_t* ep_len, float* ep_rew, float* reward, int32_t* done, int32_t* fin_len, float* fin_rew) {
    const int env = blockIdx.x * 256 + threadIdx.x;
    if (env >= N) return;
    const uint4 w = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)(env_offset + env), (uint32_t)step, 0u, 0x11u);
    const float r = (float)(w.x >> 8) * 0x1p-23f - 1.0f;
    int term = (w.y >> 8) < 167772u ? 1 : 0;
    int len = ep_len[env] + 1;
    float rew = ep_rew[env] + r;
    if (len == max_episode_steps) term = 1;
    int fl = 0;
    float fr = 0.0f;
    if (term) { fl = len; fr = rew; len = 0; rew = 0.0f; }
    ep_len[env] = len;
    ep_rew[env] = rew;
    reward[env] = r;
    done[env] = term;
    if (fin_len) { fin_len[env] = fl; fin_rew[env] = fr; }
}
// rollout stores of one step (PPO_MultiDiscrete.cpp:547-562): obs, masks, actions, log-probs, the PREVIOUS step's done flags
__global__ __launch_bounds__(256) void store_step_kernel(GenLayout L, int N, const float* __restrict__ obs, const uint8_t* __restrict__ mask,
                                                         const int64_t* __restrict__ act64, const float* __restrict__ lp, const int32_t* __restrict__ done_prev,
                                                         float* obs_t, uint8_t* mask_t, int32_t* act_t, float* lp_t, float* dones_t) {
    const int64_t total = (int64_t)N * L.obs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) obs_t[i] = obs[i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)N * L.act; i += (int64_t)gridDim.x * 256) if (mask_t) mask_t[i] = mask ? mask[i] : 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)N * L.n_heads; i += (int64_t)gridDim.x * 256) act_t[i] = (int32_t)act64[i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) { lp_t[i] = lp[i]; dones_t[i] = (float)done_prev[i]; }
}

// out[i] = sum over the S partial slabs in a fixed order: 64 elements per block, four lanes per element each adding a quarter of the slabs
// (eight loads in flight at a time), then ((q0 + q1) + q2) + q3
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, int64_t slab_stride, int S, int64_t n_w, const float* __restrict__ db_part,
                                                       int db_chunks, int64_t db_stride, int64_t n_b, float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float acc = 0.0f;
    if (i < n_w + n_b) {
        const bool w = i < n_w;
        const float* src = w ? slab + i : db_part + (i - n_w);
        const int64_t stride = w ? slab_stride : db_stride;
        const int n = w ? S : db_chunks;
        const int per = (n + 3) / 4, k0 = q * per, k1 = k0 + per < n ? k0 + per : n;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = src[(size_t)(k + j) * stride];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
        for (; k < k1; k++) acc += src[(size_t)k * stride];
    }
    red[q][e] = acc;
    __syncthreads();
    if (q == 0 && i < n_w + n_b) {
        const float t = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        if (i < n_w) gw[i] = t; else gb[i - n_w] = t;
    }
}

// The same sums for every layer of a net in ONE launch (bf16 storage: each layer has its own slab block): blockIdx.y = layer.
struct SlabJob { const float* slab; const float* db_part; float* gw; float* gb; int64_t n_w, n_b, db_stride; int S, db_chunks; int64_t stride; };   // stride: floats between this job's slabs (0: jobs.slab_stride)
struct SlabJobs { SlabJob j[2 * GEN_MAX_LAYERS]; int64_t slab_stride; double* sq_part; int sq_stride, sq_job0; };   // blockIdx.y = job: a net's layers, or both nets'
// sq_part != nullptr: the workgroup also leaves the sums of squares of its SLAB_EPB gradient elements, weights and bias apart, as
// sq_part[((sq_job0 + job) * sq_stride + blockIdx.x) * 2 + {0, 1}] (job slots in the parameter order: net * n_layers + layer) -- the gradient norm then needs no
// pass of its own over the gradient (gen_opt_fused_kernel)
constexpr int SLAB_EPB = 256;   // gradient elements per workgroup of slab_sum_layers_kernel (four groups of SLAB_EPB threads split the slabs)
__global__ __launch_bounds__(4 * SLAB_EPB) void slab_sum_layers_kernel(SlabJobs jobs) {
    __shared__ float red[4][SLAB_EPB];
    __shared__ double sq[SLAB_EPB / 64][2];
    const SlabJob& J = jobs.j[blockIdx.y];
    const int e = threadIdx.x % SLAB_EPB, q = threadIdx.x / SLAB_EPB;
    const int64_t i = (int64_t)blockIdx.x * SLAB_EPB + e;
    if ((int64_t)blockIdx.x * SLAB_EPB >= J.n_w + J.n_b) return;   // whole block past this layer's elements (uniform: no barrier is skipped by part of a block)
    float acc = 0.0f;
    if (i < J.n_w + J.n_b) {
        const bool w = i < J.n_w;
        const float* src = w ? J.slab + i : J.db_part + (i - J.n_w);
        const int64_t stride = w ? (J.stride ? J.stride : jobs.slab_stride) : J.db_stride;
        const int n = w ? J.S : J.db_chunks;
        const int per = (n + 3) / 4, k0 = q * per, k1 = k0 + per < n ? k0 + per : n;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = src[(size_t)(k + j) * stride];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
        for (; k < k1; k++) acc += src[(size_t)k * stride];
    }
    red[q][e] = acc;
    __syncthreads();
    if (q == 0) {
        float t = 0.0f;
        if (i < J.n_w + J.n_b) {
            t = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
            if (i < J.n_w) J.gw[i] = t; else J.gb[i - J.n_w] = t;
        }
        if (jobs.sq_part) {
            const double t2 = (double)t * (double)t;
            const double sw = wave_sum_d_dpp(i < J.n_w ? t2 : 0.0), sb = wave_sum_d_dpp(i < J.n_w ? 0.0 : t2);
            if ((e & 63) == 0) { sq[e >> 6][0] = sw; sq[e >> 6][1] = sb; }
        }
    }
    if (jobs.sq_part) {
        __syncthreads();
        if (threadIdx.x < 2) {
            double v = 0.0;
            for (int w = 0; w < SLAB_EPB / 64; w++) v += sq[w][threadIdx.x];
            jobs.sq_part[((size_t)(jobs.sq_job0 + blockIdx.y) * jobs.sq_stride + blockIdx.x) * 2 + threadIdx.x] = v;
        }
    }
}

// Gradient norm + clip + AdamW + the bf16 weight planes + the step's loss scalars in ONE launch behind the slab sums (single rank, both nets' fused backward):
//   norm   every workgroup adds the slab-sum workgroups' sums of squares per tensor (wave w takes jobs w, w + 4, ...: lanes stride the job's workgroups, one DPP
//          reduction per tensor) -- 37 KB out of L2 per workgroup at configs[4], instead of a kernel of its own over the gradient;
//   step   the arithmetic of gen_adamw_kernel, one batch of loads per thread (grid-stride, OPT_EPT elements in flight);
//   planes a weight's new value goes straight to its place in the bf16 plane and the fragment-order copy (weight_planes_kernel's layout; the padding is
//          zero from the allocation and never written), so the next forward pass needs no re-split launch;
//   stats  workgroup 0 also adds the loss kernel's block sums (loss_sums_kernel) and writes the step's scalars.
constexpr int OPT_EPT = 2;
struct GenOptArgs {
    float* params; const float* grads; float* exp_avg; float* exp_avg_sq; GenLayout L; float max_norm;
    const double* sq_part; int xb;
    const AdamCoef* coef; const double* loss_part; int loss_blocks; double* sums_out; float* grads_tail;
    double global_M; LossParams hp; int do_step; StepStats* stats_out; double* clipfrac_accum;
    uint16_t* planes; uint16_t* frags; int64_t wp_off[2][GEN_MAX_LAYERS]; int wp_kpad[GEN_MAX_LAYERS];
    const int32_t* error_flag;   // the context's error word (OptGuard, ppo_internal.hpp): a step behind PPO_ERRFLAG_SKIP_STEP is not applied
};
__global__ __launch_bounds__(256) void gen_opt_fused_kernel(const GenOptArgs a) {
    __shared__ double s_t[4 * GEN_MAX_LAYERS];
    __shared__ double s_ls[5][4];
    __shared__ float s_total;
    const GenLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_jobs = 2 * L.n_layers;
    const int32_t err = opt_guard_word(a.error_flag);
    for (int j = wave; j < n_jobs; j += 4) {
        const int net = j / L.n_layers, l = j % L.n_layers;
        const int64_t n_el = (int64_t)L.out_dim[net][l] * L.in_dim[l] + L.out_dim[net][l];
        const int nblk = (int)((n_el + SLAB_EPB - 1) / SLAB_EPB);
        const double* src = a.sq_part + (size_t)j * a.xb * 2;
        double aw = 0.0, ab = 0.0;
        int b = lane;
        for (; b + 3 * 64 < nblk; b += 4 * 64) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const double2*>(src + (size_t)(b + 64 * u) * 2);
#pragma unroll
            for (int u = 0; u < 4; u++) { aw += v[u].x; ab += v[u].y; }
        }
        for (; b < nblk; b += 64) { const double2 v = *reinterpret_cast<const double2*>(src + (size_t)b * 2); aw += v.x; ab += v.y; }
        aw = wave_sum_d_dpp(aw); ab = wave_sum_d_dpp(ab);
        if (lane == 0) {
            const float nw = (float)sqrt(aw), nb = (float)sqrt(ab);   // per-tensor norms as floats (clip_grad.h:58-66), then the norm of the norms
            s_t[2 * j] = (double)nw * nw; s_t[2 * j + 1] = (double)nb * nb;
        }
    }
    if (blockIdx.x == 0) {   // the loss kernel's block sums (uniform branch): the five columns requested together
        double v5[5] = { 0, 0, 0, 0, 0 };
        for (int b = tid; b < a.loss_blocks; b += 256) {
            double t[5];
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = a.loss_part[b * 8 + k];
#pragma unroll
            for (int k = 0; k < 5; k++) v5[k] += t[k];
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const double v = wave_sum_d_dpp(v5[k]);
            if (lane == 0) s_ls[k][wave] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int t = 0; t < 2 * n_jobs; t++) tot += s_t[t];
        s_total = (float)sqrt(tot);
    }
    if (blockIdx.x == 0 && tid < 8) {
        const double t = tid < 5 ? ((s_ls[tid][0] + s_ls[tid][1]) + s_ls[tid][2]) + s_ls[tid][3] : 0.0;
        a.sums_out[tid] = t;
        a.grads_tail[tid] = (float)t;
        if (tid < 5) s_ls[tid][0] = t;
    }
    __syncthreads();
    const float total = s_total;
    float c = a.max_norm / (total + 1e-6f);
    if (c > 1.0f) c = 1.0f;
    const AdamCoef k = *a.coef;
    const float b1 = 0.9f, b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-5f;
    if (a.do_step && !(err & PPO_ERRFLAG_SKIP_STEP)) {
        const int stride = gridDim.x * 256;
        for (int p0 = blockIdx.x * 256 + tid; p0 < L.P; p0 += OPT_EPT * stride) {
            float g[OPT_EPT], pv[OPT_EPT], mv[OPT_EPT], vv[OPT_EPT];
#pragma unroll
            for (int u = 0; u < OPT_EPT; u++) {
                const int p = p0 + u * stride;
                const int ps = p < L.P ? p : p0;
                g[u] = a.grads[ps]; pv[u] = a.params[ps]; mv[u] = a.exp_avg[ps]; vv[u] = a.exp_avg_sq[ps];
            }
#pragma unroll
            for (int u = 0; u < OPT_EPT; u++) {
                const int p = p0 + u * stride;
                if (p >= L.P) continue;
                const float gc = g[u] * c;
                const float pi = pv[u] * k.decay;
                const float mi = __builtin_fmaf(gc, omb1, mv[u] * b1);
                const float vi = __builtin_fmaf(omb2 * gc, gc, vv[u] * b2);
                const float denom = sqrtf(vi) / k.sqrt_bc2 + eps;
                const float pn = pi + (k.neg_step * mi) / denom;
                a.params[p] = pn;
                a.exp_avg[p] = mi;
                a.exp_avg_sq[p] = vi;
                if (a.planes) {
                    // which weight matrix (if any): the layout's offsets are uniform constants; bias elements and nothing else fall through
                    int net = -1, l = 0;
                    for (int nn = 0; nn < 2; nn++)
                        for (int ll = 0; ll < L.n_layers; ll++)
                            if (p >= L.w_off[nn][ll] && p < L.b_off[nn][ll]) { net = nn; l = ll; }
                    if (net >= 0) {
                        const int K = L.in_dim[l], e = p - L.w_off[net][l];
                        const int n = e / K, kk = e - n * K;
                        const int kpad = a.wp_kpad[l];
                        const __bf16 bq = (__bf16)pn;   // round to nearest even, as weight_planes_kernel
                        const uint16_t bits = __builtin_bit_cast(uint16_t, bq);
                        a.planes[a.wp_off[net][l] + (int64_t)n * kpad + kk] = bits;
                        a.frags[a.wp_off[net][l] + (((int64_t)(n >> 5) * (kpad / 16) + (kk >> 4)) * 64 + (n & 31) + 32 * ((kk >> 3) & 1)) * 8 + (kk & 7)] = bits;
                    }
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        double ls[5];
        for (int i = 0; i < 5; i++) ls[i] = s_ls[i][0];
        const float pg = (float)(ls[0] / a.global_M), vl = 0.5f * (float)(ls[4] / a.global_M), el = (float)(ls[1] / a.global_M);
        StepStats o;
        o.pg_loss = pg; o.v_loss = vl; o.entropy_loss = el;
        o.approx_kl = (float)(ls[2] / a.global_M);
        o.clipfrac = (float)ls[3] / (float)a.global_M;
        o.loss = (pg - a.hp.ent_coef * el) + vl * a.hp.vf_coef;
        o.total_norm = total;
        o.pad = 0.0;
        *a.stats_out = o;
        if (a.clipfrac_accum && a.do_step) { a.clipfrac_accum[0] += o.clipfrac; a.clipfrac_accum[1] += 1.0; }
    }
}

inline unsigned grid_for(int64_t n, int per_block) { const int64_t g = (n + per_block - 1) / per_block; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace

hipError_t gen_forward(const GenericCtx& g, const float* params, int net, const float* x, int64_t rows, float* const* acts, float* scratch0,
                       float* scratch1, float* out, hipStream_t s) {
    const GenLayout& L = g.L;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    if (g.bf16) {
        // bf16 storage: the input is rounded once into g.xin_bf (or gen_gather left it there), every hidden layer writes bf16
        if (x) { const hipError_t ce = launch_to_bf16_pad(x, rows, L.obs, g.xin_bf, g.ld_in0, s); if (ce != hipSuccess) return ce; }
        const uint16_t* in = g.xin_bf;
        int64_t ldi = g.ld_in0;
        for (int l = 0; l < L.n_layers; l++) {
            const int K = L.in_dim[l], N = L.out_dim[net][l];
            const bool last = l == L.n_layers - 1;
            uint16_t* hb = acts ? g.acts_bf[net][l] : g.tmp_bf[l & 1];
            const hipError_t e = launch_matmul_bf16(false, false, rows, N, K, in, ldi, g.wplanes + g.wp_off[net][l], g.wp_kpad[l], last ? (void*)out : (void*)hb,
                                                    last ? N : g.ld_h, !last, last ? PPO_MM_EPI_BIAS : PPO_MM_EPI_BIAS_TANH, params + L.b_off[net][l], 0, 1, 0,
                                                    nullptr, 0, s);
            if (e != hipSuccess) return e;
            in = hb; ldi = g.ld_h;
        }
        return hipGetLastError();
    }
    const float* in = x;
    for (int l = 0; l < L.n_layers; l++) {
        const int K = L.in_dim[l], N = L.out_dim[net][l];
        const bool last = l == L.n_layers - 1;
        float* dst = last ? out : (acts ? acts[l] : ((l & 1) ? scratch1 : scratch0));
        // one launch per layer: product, bias and tanh (kernels_gemm.hip); weights from their bf16 planes
        const hipError_t e = launch_matmul(false, false, rows, N, K, in, K, params + L.w_off[net][l], K, dst, N, last ? PPO_MM_EPI_BIAS : PPO_MM_EPI_BIAS_TANH,
                                           params + L.b_off[net][l], 0, g.gemm_prec, 1, 0, nullptr, 0, g.wplanes + g.wp_off[net][l],
                                           (int64_t)g.wp_npad[net][l] * g.wp_kpad[l], g.wp_kpad[l], s);
        if (e != hipSuccess) return e;
        in = dst;
    }
    return hipGetLastError();
}

hipError_t gen_heads(const GenLayout& L, int dist_kind, const float* logits, const uint8_t* mask, const int64_t* forced, int64_t n, int64_t seed,
                     int64_t row_offset, int64_t step_index, int64_t* action, float* logprob, float* entropy, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    if (dist_kind == PPO_DIST_MASKED)
        hipLaunchKernelGGL(heads_kernel<PPO_DIST_MASKED>, grid, block, 0, s, L, logits, mask, forced, n, seed, row_offset, step_index, action, logprob, entropy);
    else
        hipLaunchKernelGGL(heads_kernel<PPO_DIST_CATEGORICAL>, grid, block, 0, s, L, logits, mask, forced, n, seed, row_offset, step_index, action, logprob, entropy);
    return hipGetLastError();
}

hipError_t gen_gather(const GenLayout& L, const float* obs, const int32_t* actions, const uint8_t* masks, const float* logprobs, const float* adv,
                      const float* ret, const float* values, const int32_t* idx, int64_t M, GenericCtx& g, hipStream_t s) {
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, L, obs, actions, masks, logprobs, adv, ret, values, idx, M, g.xin,
                       g.bf16 ? g.xin_bf : nullptr, g.ld_in0, g.row_act, g.row_mask, g.row_f[0], g.row_f[1], g.row_f[2], g.row_f[3]);
    return hipGetLastError();
}

// mean and 1 / (Bessel std + 1e-8) of the minibatch's advantages (PPO_Discrete.cpp:592-594) from the PPO_ADV_PARTS partial sums,
// formed on the device by one thread (no host round trip); the loss kernel reads the two floats
namespace {
__global__ void adv_finish_kernel(const AdvStat* adv_stat, double global_M, int norm_adv, float* out2) {
    float mean_f = 0.0f, std_f = 0.0f;
    if (norm_adv) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += adv_stat[i].s1; t2 += adv_stat[i].s2; }
        const double mean = t1 / global_M;
        const double var = (t2 - t1 * mean) / (global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
    }
    out2[0] = mean_f;
    out2[1] = 1.0f / (std_f + 1e-8f);
}
}  // namespace

bool gen_rows_packable(const GenLayout& L) {
    if (L.n_heads > 4 || L.act > 32) return false;
    for (int h = 0; h < L.n_heads; h++) if (L.head_dims[h] > 256) return false;
    return true;
}
hipError_t gen_pack_rows(const GenLayout& L, const GenRowSrc& src, int64_t B, float* rec, hipStream_t s, const int32_t* idx) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, L, src.actions, src.masks, src.logprobs, src.adv, src.ret, src.values, B,
                       reinterpret_cast<float4*>(rec), idx);
    return hipGetLastError();
}

hipError_t gen_loss(const GenLayout& L, const LossParams& hp, const GenericCtx& g, int64_t M, double inv_global_M, double global_M,
                    const AdvStat* adv_stat, hipStream_t s) {
    const dim3 grid(GEN_LOSS_BLOCKS), block(256);
    if (g.bf16) {   // bf16 storage: per-row arrays in registers, bounds 4 heads x 16 logits or the ABI's maximum
        // the per-row scalars: gathered copies, or (rows_idx set) the rollout's own arrays through the step's index list
        const bool direct = g.rows_idx != nullptr;
        const GenRowSrc& R = g.rows_src;
        const uint8_t* mask = hp.dist_kind == PPO_DIST_MASKED ? (direct ? R.masks : g.row_mask) : nullptr;
#define GEN_LOSS_REG(DIST, NH, AM)                                                                                                                     \
        hipLaunchKernelGGL((loss_reg_kernel<DIST, NH, AM>), grid, block, 0, s, L, hp, g.logits, g.val, direct ? R.actions : g.row_act, mask,              \
                           direct ? R.logprobs : g.row_f[0], direct ? R.adv : g.row_f[1], direct ? R.ret : g.row_f[2], direct ? R.values : g.row_f[3], M,   \
                           (float)inv_global_M, (adv_stat && hp.norm_adv) ? adv_stat : nullptr, global_M, g.loss_part, g.dout_bf[1], g.dout_bf[0],      \
                           g.head_db_part, g.rows_idx, direct ? g.rows_rec : nullptr)
        {   // <= 4 heads of <= 4 logits behind packed row records: the slot-wise kernel
            bool slots = direct && g.rows_rec != nullptr && L.n_heads <= 4;
            for (int h = 0; h < L.n_heads; h++) slots = slots && L.head_dims[h] <= 4;
#ifdef GEN_AB_LOSS_REG   // A/B build: the general kernel for every shape
            slots = false;
#endif
            if (slots) {
                const AdvStat* st = (adv_stat && hp.norm_adv) ? adv_stat : nullptr;
#ifndef GEN_AB_LOSS_SMALL   // A/B build: a row per thread
                if (hp.dist_kind == PPO_DIST_MASKED)
                    hipLaunchKernelGGL(loss_lanes_kernel<PPO_DIST_MASKED>, grid, dim3(1024), 0, s, L, hp, g.logits, g.val, M, (float)inv_global_M, st, global_M, g.loss_part, g.dout_bf[1],
                                       g.dout_bf[0], g.head_db_part, g.rows_idx, g.rows_rec);
                else
                    hipLaunchKernelGGL(loss_lanes_kernel<PPO_DIST_CATEGORICAL>, grid, dim3(1024), 0, s, L, hp, g.logits, g.val, M, (float)inv_global_M, st, global_M, g.loss_part,
                                       g.dout_bf[1], g.dout_bf[0], g.head_db_part, g.rows_idx, g.rows_rec);
                return hipGetLastError();
#endif
                if (hp.dist_kind == PPO_DIST_MASKED)
                    hipLaunchKernelGGL(loss_small_kernel<PPO_DIST_MASKED>, grid, block, 0, s, L, hp, g.logits, g.val, M, (float)inv_global_M, st, global_M, g.loss_part, g.dout_bf[1],
                                       g.dout_bf[0], g.head_db_part, g.rows_idx, g.rows_rec);
                else
                    hipLaunchKernelGGL(loss_small_kernel<PPO_DIST_CATEGORICAL>, grid, block, 0, s, L, hp, g.logits, g.val, M, (float)inv_global_M, st, global_M, g.loss_part,
                                       g.dout_bf[1], g.dout_bf[0], g.head_db_part, g.rows_idx, g.rows_rec);
                return hipGetLastError();
            }
        }
        const bool small = L.n_heads <= 4 && L.act <= 16;
        if (hp.dist_kind == PPO_DIST_MASKED) { if (small) GEN_LOSS_REG(PPO_DIST_MASKED, 4, 16); else GEN_LOSS_REG(PPO_DIST_MASKED, PPO_MAX_HEADS, PPO_MAX_ACT); }
        else { if (small) GEN_LOSS_REG(PPO_DIST_CATEGORICAL, 4, 16); else GEN_LOSS_REG(PPO_DIST_CATEGORICAL, PPO_MAX_HEADS, PPO_MAX_ACT); }
#undef GEN_LOSS_REG
        return hipGetLastError();
    }
    hipLaunchKernelGGL(adv_finish_kernel, dim3(1), dim3(1), 0, s, adv_stat, global_M, (adv_stat && hp.norm_adv) ? 1 : 0, g.row_f[4]);
    if (hp.dist_kind == PPO_DIST_MASKED)
        hipLaunchKernelGGL(loss_kernel<PPO_DIST_MASKED>, grid, block, 0, s, L, hp, g.logits, g.val, g.row_act, g.row_mask, g.row_f[0], g.row_f[1], g.row_f[2],
                           g.row_f[3], M, (float)inv_global_M, g.row_f[4], g.dlogits, g.dval, g.loss_part, g.bf16 ? g.dout_bf[1] : nullptr,
                           g.bf16 ? g.dout_bf[0] : nullptr, g.head_db_part);
    else
        hipLaunchKernelGGL(loss_kernel<PPO_DIST_CATEGORICAL>, grid, block, 0, s, L, hp, g.logits, g.val, g.row_act, nullptr, g.row_f[0], g.row_f[1], g.row_f[2],
                           g.row_f[3], M, (float)inv_global_M, g.row_f[4], g.dlogits, g.dval, g.loss_part, g.bf16 ? g.dout_bf[1] : nullptr,
                           g.bf16 ? g.dout_bf[0] : nullptr, g.head_db_part);
    return hipGetLastError();
}

// The fused backward pass (kernels_generic_bwd.hip): one launch per layer -- dZ_l and h_{l-1} are read ONCE for the weight gradient AND the gradient handed down;
// layer 0 has nothing below it and forms its weight gradient alone -- then the fixed-order sums of all layers' slabs and column sums in one launch.
// net_b >= 0: the same layers of a second net ride in the same launches (each net sized for half the chip), and one launch sums both nets' slabs.
namespace {
hipError_t fused_backward(const GenericCtx& g, int net_a, int net_b, int64_t rows, float* grads, hipStream_t s, bool half_chip) {
    const GenLayout& L = g.L;
    const int nets[2] = { net_a, net_b };
    const int n_nets = net_b >= 0 ? 2 : 1;
    const uint16_t* d[2];
    int64_t ldd = 128;
    int S_above[2] = { 0, 0 };
    for (int i = 0; i < n_nets; i++) d[i] = g.dout_bf[nets[i]];   // [rows + 128][128]: columns >= 32 are zero for good (L.act <= 32): row 0's columns 64 .. 71 are the kernel's 16 zero bytes
    SlabJobs jobs{};
    jobs.slab_stride = g.wslab_stride;
    int64_t most = 0;
    for (int l = L.n_layers - 1; l >= 0; l--) {
        const bool head = l == L.n_layers - 1;
        const bool in_place = l == 0 && g.rows_idx != nullptr;   // layer 0 reads the minibatch's rows of the update's bf16 observations through the index list
        const int64_t ldi = l == 0 ? g.ld_in0 : g.ld_h;
        const int cbk = gen_bwd_col_blocks((int)ldi, l > 0);
        int tpr = 1;
        const int S = gen_bwd_ranges(rows, cbk, half_chip || n_nets == 2, &tpr);
        // a layer's block holds GEN_SPLIT_MFMA + 1 slabs and cs_layer_stride / ld_h rows of column sums: more ranges than that would run into the next layer's
        // block (a whole-chip launch of a 128-column layer 0 asks for 256: refused here rather than summed wrong)
        if ((int64_t)S * g.wslab_stride > g.wslab_layer_stride || (int64_t)S * g.ld_h > g.cs_layer_stride) return hipErrorInvalidValue;
        GenBwdLayer q[2];
        for (int i = 0; i < n_nets; i++) {
            const int net = nets[i];
            const int K = L.in_dim[l], N = L.out_dim[net][l];
            float* lslab = (net == 1 ? g.wslab1 : g.wslab) + (size_t)l * g.wslab_layer_stride;
            uint16_t* nd = l > 0 ? g.dz_bf[net][l & 1] : nullptr;
            q[i] = GenBwdLayer{ d[i], ldd, l == 0 ? (in_place ? g.obs_bf : g.xin_bf) : g.acts_bf[net][l - 1], ldi, in_place ? g.rows_idx : nullptr,
                                l > 0 ? g.wplanes + g.wp_off[net][l] : nullptr, g.wp_kpad[l], nd, g.ld_h, lslab, g.wslab_stride,
                                l > 0 ? g.cs_part[net] + (size_t)l * g.cs_layer_stride : nullptr, g.ld_h, N, K, S, tpr };
            SlabJob& J = jobs.j[(n_nets == 2 ? net : 0) * L.n_layers + l];   // both nets: jobs in the parameter order (critic's layers, then the actor's)
            J.slab = lslab; J.S = S; J.n_w = (int64_t)N * K; J.n_b = N;
            // the bias gradient of layer l: block sums of the head gradient (loss kernel), or the column sums the launch of layer l + 1 left in block l + 1
            J.db_part = head ? g.head_db_part + (net == 0 ? L.act : 0) : g.cs_part[net] + (size_t)(l + 1) * g.cs_layer_stride;
            J.db_chunks = head ? GEN_LOSS_BLOCKS : S_above[i];
            J.db_stride = head ? L.act + 1 : g.ld_h;
            J.gw = grads + L.w_off[net][l]; J.gb = grads + L.b_off[net][l];
            most = std::max<int64_t>(most, J.n_w + N);
            S_above[i] = S;
            d[i] = nd;
        }
        if (head && g.head_fused > 0) {
            // the forward launch already ran this layer's backward (gen_fused_forward_loss): its per-workgroup partials stand where a row range's would
            for (int i = 0; i < n_nets; i++) {
                SlabJob& J = jobs.j[(n_nets == 2 ? nets[i] : 0) * L.n_layers + l];
                J.S = g.head_fused; J.stride = (int64_t)L.act * L.hidden;
                S_above[i] = g.head_fused;
            }
            ldd = g.ld_h;
            continue;
        }
        const hipError_t e = gen_fused_backward_layer(head ? 32 : g.ld_h, q[0], n_nets == 2 ? &q[1] : nullptr, rows, cbk, g.dout_bf[nets[0]] + 64, s);
        if (e != hipSuccess) return e;
        ldd = g.ld_h;
    }
    jobs.sq_part = (most + SLAB_EPB - 1) / SLAB_EPB <= g.sq_cap ? g.sq_part : nullptr;
    jobs.sq_stride = g.sq_cap; jobs.sq_job0 = n_nets == 2 ? 0 : net_a * L.n_layers;
    for (int i = 0; i < n_nets; i++) g.sq_valid[nets[i]] = jobs.sq_part != nullptr;
    hipLaunchKernelGGL(slab_sum_layers_kernel, dim3((unsigned)((most + SLAB_EPB - 1) / SLAB_EPB), (unsigned)(n_nets * L.n_layers)), dim3(4 * SLAB_EPB), 0, s, jobs);
    return hipGetLastError();
}
}  // namespace

// Backward of one net: dout = d(loss)/d(output layer) [rows, out]; fills the net's slice of the flat gradient.  Needs the activations
// kept by gen_forward(..., acts = g.acts[net]).  `ones` = g.dz[1] + rows * hidden is NOT used: bias gradients are a gemv with the
// ones vector kept in g.row_f[4] + 2 (see api.hip: filled once at creation).
hipError_t gen_backward(const GenericCtx& g, const float* params, int net, const float* x, int64_t rows, const float* dout, float* grads,
                        hipStream_t s, bool beside_other_net) {
    const GenLayout& L = g.L;
    g.sq_valid[net] = false;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    // dW[N, K] = d^T[N, rows] . in[rows, K] and db[N] = d^T . 1 contract over the minibatch rows: a single product would have N K / tile
    // workgroups walking all rows, so the rows are cut into ranges -- every range its own partial slab, enough of them for one or two workgroups
    // per CU, each a multiple of 64 rows -- and the slabs are added in a fixed order.
    auto ranges = [&](int N, int K) {
        const int64_t tiles = (int64_t)((N + 127) / 128) * ((K + 127) / 128);
        // 256 workgroups -- one eight-wave workgroup per CU: alone on the chip that is slower than 512 (31.5 against 22.6 us per product, 8.6 against
        // 10.0 us per slab sum), but the two nets' passes run side by side on two streams and there half the slab bytes win: 0.709 -> 0.705 ms per
        // minibatch step (A/B in one call; 128 workgroups: 0.762)
        int64_t want = ((g.bf16 ? 256 : 512) + tiles - 1) / tiles;   // f32 (one stream, two four-wave workgroups per CU): 512
        if (want > GEN_SPLIT_MFMA) want = GEN_SPLIT_MFMA;
        const int64_t range = ((rows + want - 1) / want + 63) / 64 * 64;
        return (int)((rows + range - 1) / range);
    };
#ifndef GEN_AB_UNFUSED_BWD   // A/B build (tools/build_variant.sh): the two tiled products per layer of rounds 1-4
    if (gen_fused_backward_ok(g)) return fused_backward(g, net, -1, rows, grads, s, beside_other_net);
#endif
    if (g.bf16) {
        // bf16 storage: x, dout are not used -- the layer inputs are g.xin_bf / g.acts_bf[net], the head gradient g.dout_bf[net] (written by the
        // loss kernel together with the head's bias-gradient block sums); the bias gradient of every other layer is summed, in f32, where its
        // d(pre-activation) is produced (the tanh' epilogue of the layer above), per 128-row tile
        const uint16_t* d = g.dout_bf[net];
        int64_t ldd = 128;
        float* wslab = net == 1 ? g.wslab1 : g.wslab;   // every scratch buffer of this pass belongs to the net: the other net's pass may be running
        if (rows % 64) {   // the contraction over rows runs in pairs of 32-row chunks: rows past the minibatch must contribute zeros
            const int64_t tail = 64 - rows % 64;
            for (int i = 0; i < 2; i++) {
                if (hipMemsetAsync(g.dz_bf[net][i] + rows * g.ld_h, 0, (size_t)tail * g.ld_h * 2, s) != hipSuccess) return hipErrorUnknown;
                if (i == net && hipMemsetAsync(g.dout_bf[i] + rows * 128, 0, (size_t)tail * 128 * 2, s) != hipSuccess) return hipErrorUnknown;
            }
        }
        // every layer's partial slabs (and the column sums that are the next bias gradient) go to the layer's OWN block; the fixed-order sums of all layers
        // are ONE launch at the end of the pass (they are needed by the norm / AdamW only): 2 launches per minibatch step instead of 10
        SlabJobs jobs{};
        jobs.slab_stride = g.wslab_stride;
        int64_t most = 0;
        for (int l = L.n_layers - 1; l >= 0; l--) {
            const int K = L.in_dim[l], N = L.out_dim[net][l];
            const bool head = l == L.n_layers - 1;
            const uint16_t* in = l == 0 ? g.xin_bf : g.acts_bf[net][l - 1];
            const int64_t ldi = l == 0 ? g.ld_in0 : g.ld_h;
            const int64_t n_w = (int64_t)N * K;
            const int S = ranges(N, K);
            float* lslab = wslab + (size_t)l * g.wslab_layer_stride;
            hipError_t e = launch_matmul_bf16(true, true, N, K, rows, d, ldd, in, ldi, lslab, K, false, PPO_MM_EPI_NONE, nullptr, 0, S, g.wslab_stride, nullptr, 0, s);
            if (e != hipSuccess) return e;
            // the bias gradient of layer l: block sums of the head gradient (loss kernel), or the column sums the d(input) product of layer l + 1 left
            // in block l + 1 of cs_part (written below, one iteration ago)
            SlabJob& J = jobs.j[l];
            J.slab = lslab; J.S = S; J.n_w = n_w; J.n_b = N;
            J.db_part = head ? g.head_db_part + (net == 0 ? L.act : 0) : g.cs_part[net] + (size_t)(l + 1) * g.cs_layer_stride;
            J.db_chunks = head ? GEN_LOSS_BLOCKS : (int)((rows + 127) / 128);
            J.db_stride = head ? L.act + 1 : g.ld_h;
            J.gw = grads + L.w_off[net][l]; J.gb = grads + L.b_off[net][l];
            most = std::max<int64_t>(most, n_w + N);
            if (l > 0) {
                uint16_t* nd = g.dz_bf[net][l & 1];
                // dH[rows, K] = d[rows, N] . W[N, K], then d(pre-activation) = dH (1 - h^2), stored bf16; its column sums per 128-row tile -> cs_part block l
                e = launch_matmul_bf16(false, true, rows, K, N, d, ldd, g.wplanes + g.wp_off[net][l], g.wp_kpad[l], nd, g.ld_h, true, PPO_MM_EPI_DTANH,
                                       g.acts_bf[net][l - 1], g.ld_h, 1, 0, g.cs_part[net] + (size_t)l * g.cs_layer_stride, g.ld_h, s);
                if (e != hipSuccess) return e;
                d = nd; ldd = g.ld_h;
            }
        }
        hipLaunchKernelGGL(slab_sum_layers_kernel, dim3((unsigned)((most + SLAB_EPB - 1) / SLAB_EPB), (unsigned)L.n_layers), dim3(4 * SLAB_EPB), 0, s, jobs);
        return hipGetLastError();
    }
    const float* d = dout;
    for (int l = L.n_layers - 1; l >= 0; l--) {
        const int K = L.in_dim[l], N = L.out_dim[net][l];
        const float* in = l == 0 ? x : g.acts[net][l - 1];
        const int64_t n_w = (int64_t)N * K;
        const int S = ranges(N, K);
        {   // ... and db beside it: the workgroups of the first column of tiles also sum their d tile over the rows (db_part[z][N])
            const hipError_t e = launch_matmul(true, true, N, K, rows, d, N, in, K, g.wslab, K, PPO_MM_EPI_NONE, nullptr, 0, g.gemm_prec, S, g.wslab_stride,
                                               g.db_part, N, nullptr, 0, 0, s);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n_w + N + 63) / 64)), dim3(256), 0, s, g.wslab, g.wslab_stride, S, n_w, g.db_part, S, (int64_t)N, (int64_t)N,
                           grads + L.w_off[net][l], grads + L.b_off[net][l]);
        if (l > 0) {
            float* nd = g.dz[(l & 1)];
            // dH[rows, K] = d[rows, N] . W[N, K], then d(pre-activation) = dH (1 - h^2): one launch
            const hipError_t e = launch_matmul(false, true, rows, K, N, d, N, params + L.w_off[net][l], K, nd, K, PPO_MM_EPI_DTANH, g.acts[net][l - 1], K,
                                               g.gemm_prec, 1, 0, nullptr, 0, g.wplanes + g.wp_off[net][l], (int64_t)g.wp_npad[net][l] * g.wp_kpad[l],
                                               g.wp_kpad[l], s);
            if (e != hipSuccess) return e;
            d = nd;
        }
    }
    return hipGetLastError();
}

// both nets' fused backward passes in the same launches, on one stream (bf16 storage, gen_fused_backward_ok): actor and critic side by side on the chip
hipError_t gen_backward_both(const GenericCtx& g, const float* params, int64_t rows, float* grads, hipStream_t s) {
    if (!gen_fused_backward_ok(g)) return hipErrorNotSupported;
    if (g.planes_dirty) { const hipError_t pe = gen_weight_planes(g, params, s); if (pe != hipSuccess) return pe; g.planes_dirty = false; }
    return fused_backward(g, 1, 0, rows, grads, s, true);
}

// The optimizer step behind the fused backward passes of both nets (gen_backward_both, or gen_backward per net) on a single rank: ONE launch (gen_opt_fused_kernel) for the loss sums, the gradient norm out of the slab sums'
// partials, clip + AdamW and the refreshed bf16 weight planes.  do_step false: loss scalars and the norm only.
hipError_t gen_opt_fused(const GenericCtx& g, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float max_grad_norm, const AdamCoef* coef,
                         double* sums_out, double global_M, LossParams hp, bool do_step, StepStats* stats_out, double* clipfrac_accum, const int32_t* error_flag,
                         hipStream_t s) {
    const GenLayout& L = g.L;
    if (!g.sq_part || g.sq_cap <= 0 || (PPO_OPT_GUARD && !error_flag)) return hipErrorInvalidValue;
    GenOptArgs a{};
    a.params = params; a.grads = grads; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.L = L; a.max_norm = max_grad_norm;
    a.sq_part = g.sq_part; a.xb = g.sq_cap;
    a.coef = coef; a.loss_part = g.loss_part; a.loss_blocks = GEN_LOSS_BLOCKS; a.sums_out = sums_out; a.grads_tail = grads + L.P;
    a.global_M = global_M; a.hp = hp; a.do_step = do_step ? 1 : 0; a.stats_out = stats_out; a.clipfrac_accum = clipfrac_accum; a.error_flag = error_flag;
    const bool planes = do_step && g.gemm_prec == PPO_MM_BF16 && g.wfrags != nullptr && !g.planes_dirty;   // dirty planes are rebuilt whole by their next user
    a.planes = planes ? g.wplanes : nullptr; a.frags = planes ? g.wfrags : nullptr;
    for (int net = 0; net < 2; net++) for (int l = 0; l < L.n_layers; l++) a.wp_off[net][l] = g.wp_off[net][l];
    for (int l = 0; l < L.n_layers; l++) a.wp_kpad[l] = g.wp_kpad[l];
    const int64_t per = (int64_t)256 * OPT_EPT;
    int64_t blocks = do_step ? (L.P + per - 1) / per : 1;
    if (blocks > 2048) blocks = 2048;   // every workgroup re-adds the partial sums of squares: a few per CU, not thousands
    hipLaunchKernelGGL(gen_opt_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    if (do_step && !planes) g.planes_dirty = true;
    return hipGetLastError();
}

hipError_t gen_fill(float* p, int64_t n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

hipError_t gen_loss_sums(const GenericCtx& g, double* sums_out, float* grads_tail, hipStream_t s) {
    hipLaunchKernelGGL(loss_sums_kernel, dim3(1), dim3(256), 0, s, g.loss_part, GEN_LOSS_BLOCKS, sums_out, grads_tail);
    return hipGetLastError();
}

hipError_t gen_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const GenLayout& L, float max_grad_norm, const AdamCoef* coef,
                          const double* loss_sums, double global_M, LossParams hp, int world, bool do_step, StepStats* stats_out,
                          double* clipfrac_accum, double* norm2_scratch, hipStream_t s) {
    hipLaunchKernelGGL(gen_norm_kernel, dim3(L.n_tensors, GEN_NORM_PARTS), dim3(256), 0, s, grads, L, norm2_scratch);
    const int blocks = do_step ? (int)grid_for(L.P, 256) : 1;
    hipLaunchKernelGGL(gen_adamw_kernel, dim3(blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, L, max_grad_norm, norm2_scratch, coef, loss_sums,
                       global_M, hp, world, do_step ? 1 : 0, stats_out, clipfrac_accum);
    return hipGetLastError();
}

// One env step of the synthetic env at global step `step_index`: rewards / dones of the step, then the observation (and mask) the
// agent sees next, i.e. those of step_index + 1.  obs_out == nullptr skips the observation (bookkeeping only).
hipError_t gen_synthetic_step(const GenLayout& L, int N, int64_t seed, int64_t env_offset, int64_t step_index, int max_episode_steps, int32_t* ep_len,
                              float* ep_rew, float* obs_out, uint8_t* mask_out, float* reward, int32_t* done, int32_t* fin_len, float* fin_rew,
                              hipStream_t s) {
    if (reward) hipLaunchKernelGGL(synthetic_transition_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, seed, env_offset, step_index, max_episode_steps, ep_len,
                                   ep_rew, reward, done, fin_len, fin_rew);
    if (obs_out) hipLaunchKernelGGL(synthetic_obs_kernel, dim3(N), dim3(64), 0, s, L, N, seed, env_offset, step_index + (reward ? 1 : 0), obs_out, mask_out);
    return hipGetLastError();
}

hipError_t gen_store_step(const GenLayout& L, int N, const float* obs, const uint8_t* mask, const int64_t* act64, const float* lp, const int32_t* done_prev,
                          float* obs_t, uint8_t* mask_t, int32_t* act_t, float* lp_t, float* dones_t, hipStream_t s) {
    hipLaunchKernelGGL(store_step_kernel, dim3(grid_for((int64_t)N * L.obs, 256)), dim3(256), 0, s, L, N, obs, mask, act64, lp, done_prev, obs_t, mask_t, act_t,
                       lp_t, dones_t);
    return hipGetLastError();
}
