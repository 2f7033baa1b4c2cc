// ppo-libtorch_amd/csrc/kernels_update_mfma.hip -- K5-K8 (gather + forward + PPO loss + backward) on the CDNA4 matrix cores.
//
// Same contract as fwd_bwd_kernel in kernels_update.hip (reference PPO_Discrete.cpp:576-638), different machine mapping:
// the three 64x64 contractions per sample and net -- layer-2 forward, d(hidden 1), and the weight gradient dW2 -- are
// 32x32x2 fp32 MFMAs (v_mfma_f32_32x32x2_f32: exact fp32, bit-for-bit an fmaf chain in k order), 192 MFMAs per 32-sample
// tile.  MFMA is used only here because only here is the minibatch (131 072 rows at BASELINE configs[1]) a real contraction.
//
// One WAVE owns a 32-sample tile of one net from gather to weight gradient; a workgroup is four such waves of the same net
// (blockIdx.y) sharing LDS copies of the weights.  Register layout of every hidden vector is the MFMA C/D layout
//     lane (s = lane & 31, hi = lane >> 5), element e = r + 16 t   <->   unit  U(r, hi, t) = (r & 3) + 8 (r >> 2) + 4 hi + 32 t
// i.e. lane = sample, registers = 32 of the 64 units.  Because the contraction index of an MFMA may be enumerated in any
// order as long as A and B agree, a D-layout vector is directly the B operand of the next product (k-index (e, hi) <-> unit
// U(e%16, hi, e/16)) when the weight operand is fetched in that same order: forward and d(hidden) need NO data movement.
// Only the products that contract over SAMPLES (dW2, dW3, dW1, bias gradients) need lane = unit: the tile is bounced through
// a private 32 x 68-float LDS image (4 times per tile, ~64 KB of LDS traffic against 12 288 MFMA cycles).
// Weight-gradient accumulators (64 registers for dW2) live in registers across all tiles of the wave; the four waves of a
// workgroup are then added in a fixed order through LDS and leave as ONE partial slab (deterministic, no float atomics).
#include "ppo_internal.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 256;
constexpr int MF_WAVES = 4;
constexpr int MT = 32;              // samples per wave tile
constexpr int LS = 68;              // padded LDS row stride (floats)

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
__host__ __device__ inline MfSmem mf_smem(int obs, int aout) {
    MfSmem m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.w2 = take(64 * LS);     // [n][k], padded rows
    m.w2t = take(64 * LS);    // [k][n]
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

template <int NET, int DIST, int OBS, int AMAX>
__device__ __forceinline__ void mf_body(const UpdateArgs& a, float* smem) {
    const NetLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, hi = lane >> 5;
    const int AOUT = NET == 0 ? 1 : L.act;
    const MfSmem m = mf_smem(OBS, AOUT);
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
    for (int e = tid; e < 64 * 64; e += MF_THREADS) {
        const float w = P[L.w2[NET] + e];
        const int n = e >> 6, k = e & 63;
        sW2[n * LS + k] = w;
        sW2T[k * LS + n] = w;
    }
    for (int e = tid; e < 64 * OBS; e += MF_THREADS) sW1[e] = P[L.w1[NET] + e];
    for (int e = tid; e < AOUT * 64; e += MF_THREADS) sW3[e] = P[L.w3[NET] + e];
    if (tid < 64) { sB1[tid] = P[L.b1[NET] + tid]; sB2[tid] = P[L.b2[NET] + tid]; }
    if (tid < AOUT) sB3[tid] = P[L.b3[NET] + tid];

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
        const double mean = a.adv_stat->s1 / a.global_M;
        const double var = (a.adv_stat->s2 - a.adv_stat->s1 * mean) / (a.global_M - 1.0);
        mean_f = (float)mean;
        std_f = (float)sqrt(var > 0.0 ? var : 0.0);
    }
    __syncthreads();

    const int n_tiles = (a.M + MT - 1) / MT;
    for (int tile = blockIdx.x * MF_WAVES + wave; tile < n_tiles; tile += gridDim.x * MF_WAVES) {
        // ---------------- gather (K5): lanes (s, 0) and (s, 1) read the same batch row ----------------
        const int j = tile * MT + s;
        const bool valid = j < a.M;
        const int row = valid ? a.idx[j] : 0;
        float x[OBS];
#pragma unroll
        for (int o = 0; o < OBS; o++) x[o] = valid ? a.obs[(size_t)row * OBS + o] : 0.0f;
        if (hi == 0) {
#pragma unroll
            for (int o = 0; o < OBS; o++) sX[o * MT + s] = x[o];
        }

        // ---------------- layer 1 (VALU): this lane's 32 units ----------------
        float h1[32];
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int u = umap(e & 15, hi, e >> 4);
            float z = sB1[u];
#pragma unroll
            for (int o = 0; o < OBS; o++) z = __builtin_fmaf(x[o], sW1[u * OBS + o], z);
            h1[e] = tanhf(z);
        }

        // ---------------- layer 2 forward (MFMA): z2^T[n][s] = b2[n] + sum_k W2[n][k] h1[s][k] ----------------
        float h2[32];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = sB2[umap(r, hi, t)];
#pragma unroll
            for (int tk = 0; tk < 2; tk++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 w = ld4(&sW2[(s + 32 * t) * LS + 8 * q + 4 * hi + 32 * tk]);  // W2[n][U(4q.., hi, tk)]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h1[16 * tk + 4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h1[16 * tk + 4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h1[16 * tk + 4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h1[16 * tk + 4 * q + 3], acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; r++) h2[16 * t + r] = tanhf(acc[r]);
        }

        // ---------------- head + loss (K6, K7): both half-lanes of a sample compute the same scalars ----------------
        float dOut[AMAX];
#pragma unroll
        for (int k = 0; k < AMAX; k++) dOut[k] = 0.0f;
        if (NET == 0) {
            float part = 0.0f;
#pragma unroll
            for (int e = 0; e < 32; e++) part = __builtin_fmaf(h2[e], sW3[umap(e & 15, hi, e >> 4)], part);
            const float other = __shfl_xor(part, 32, 64);
            const float v = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[0];  // same association in both halves
            const float R = a.returns[row], vold = a.values[row];
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
            const int A = L.act;
            float z[AMAX], pr[AMAX];
            bool ok[AMAX];
#pragma unroll
            for (int k = 0; k < AMAX; k++) {
                z[k] = 0.0f; pr[k] = 0.0f; ok[k] = true;
                if (k < A) {
                    float part = 0.0f;
#pragma unroll
                    for (int e = 0; e < 32; e++) part = __builtin_fmaf(h2[e], sW3[k * 64 + umap(e & 15, hi, e >> 4)], part);
                    const float other = __shfl_xor(part, 32, 64);
                    z[k] = ((hi == 0 ? part : other) + (hi == 0 ? other : part)) + sB3[k];
                    if (DIST == PPO_DIST_MASKED && a.masks) ok[k] = a.masks[(size_t)row * A + k] != 0;
                    if (DIST == PPO_DIST_MASKED && !ok[k]) z[k] = -1e8f;
                }
            }
            float nlp = 0.0f, ent = 0.0f;
            float headH[AMAX];
            int act_s[AMAX];
#pragma unroll
            for (int h = 0; h < AMAX; h++) { headH[h] = 0.0f; act_s[h] = (h < L.n_heads) ? a.actions[(size_t)row * L.n_heads + h] : 0; }
            int off = 0;
#pragma unroll
            for (int h = 0; h < AMAX; h++) {
                if (h >= L.n_heads) break;
                const int Ah = L.head_dims[h];
                const int act_h = act_s[h];
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) mx = z[k] > mx ? z[k] : mx;
                float se = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) { pr[k] = expf(z[k] - mx); se += pr[k]; }
                const float lse = logf(se) + mx;
                float e1 = 0.0f, lp = 0.0f;
#pragma unroll
                for (int k = 0; k < AMAX; k++) if (k >= off && k < off + Ah) {
                    z[k] = z[k] - lse;
                    pr[k] = pr[k] / se;
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
            const float logratio = nlp - a.logprobs[row];   // :585
            const float ratio = expf(logratio);             // :586
            float adv = a.advantages[row];
            if (a.hp.norm_adv) adv = (adv - mean_f) / (std_f + 1e-8f);   // :593
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
                if (h >= L.n_heads) break;
                const int Ah = L.head_dims[h];
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
#pragma unroll 8
                for (int ss = 0; ss < MT; ss++) acc = __builtin_fmaf(sDo[k * MT + ss], img[ss * LS + lane], acc);
                gW3[k] += acc;
                gb3[k] += hi == 0 ? dOut[k] : 0.0f;   // summed over lanes at the end
            }
        }

        // ---------------- dz2 = (sum_a dOut[a] W3[a][u]) (1 - h2^2), D layout ----------------
        float dz2[32];
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int u = umap(e & 15, hi, e >> 4);
            float d = 0.0f;
#pragma unroll
            for (int k = 0; k < AMAX; k++) if (k < AOUT) d = __builtin_fmaf(dOut[k], sW3[k * 64 + u], d);
            dz2[e] = d * (1.0f - h2[e] * h2[e]);
        }
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
        // ---------------- dW2[n][k] += sum_s dz2[s][n] h1[s][k] (MFMA, contraction over samples) ----------------
#pragma unroll
        for (int stp = 0; stp < 16; stp++) {
            const float b0 = img[(2 * stp + hi) * LS + s];        // h1[sample][k = s]
            const float b1 = img[(2 * stp + hi) * LS + s + 32];   // h1[sample][k = s + 32]
            gW2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][stp], b0, gW2[0][0], 0, 0, 0);
            gW2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[0][stp], b1, gW2[0][1], 0, 0, 0);
            gW2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[1][stp], b0, gW2[1][0], 0, 0, 0);
            gW2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA[1][stp], b1, gW2[1][1], 0, 0, 0);
        }

        // ---------------- dh1^T[k][s] = sum_n W2[n][k] dz2[s][n] (MFMA), dz1 = dh1 (1 - h1^2) ----------------
        float dz1[32];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 w = ld4(&sW2T[(s + 32 * t) * LS + 8 * q + 4 * hi + 32 * tn]);  // W2[U(4q.., hi, tn)][k = s + 32 t]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, dz2[16 * tn + 4 * q + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, dz2[16 * tn + 4 * q + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, dz2[16 * tn + 4 * q + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, dz2[16 * tn + 4 * q + 3], acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; r++) dz1[16 * t + r] = acc[r] * (1.0f - h1[16 * t + r] * h1[16 * t + r]);
        }
        wave_lds_fence();  // dW2's reads of the h1 image are done
        store_dlayout(img, dz1, s, hi);
        wave_lds_fence();
        // ---------------- dW1[u = lane][o] += sum_s dz1[s][u] x[s][o]; db1 ----------------
        {
            float accw[OBS], accb = 0.0f;
#pragma unroll
            for (int o = 0; o < OBS; o++) accw[o] = 0.0f;
#pragma unroll 8
            for (int ss = 0; ss < MT; ss++) {
                const float d = img[ss * LS + lane];
                accb += d;
#pragma unroll
                for (int o = 0; o < OBS; o++) accw[o] = __builtin_fmaf(d, sX[o * MT + ss], accw[o]);
            }
            gb1 += accb;
#pragma unroll
            for (int o = 0; o < OBS; o++) gW1[o] += accw[o];
        }
        wave_lds_fence();  // image and sX are rewritten by the next tile
    }

    // ---------------- add the four waves in a fixed order into one slab ----------------
    __syncthreads();
    float* red = smem;  // weights are dead: reuse the front of LDS as a [net_size] accumulator
    const int base = L.net_off[NET];
    const int nsz = L.net_size[NET];
    for (int e = tid; e < nsz; e += MF_THREADS) red[e] = 0.0f;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < MF_WAVES; w++) {
        if (wave == w) {
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int tk = 0; tk < 2; tk++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        red[L.w2[NET] - base + umap(r, hi, tn) * 64 + s + 32 * tk] += gW2[tn][tk][r];
#pragma unroll
            for (int k = 0; k < AMAX; k++) if (k < AOUT) red[L.w3[NET] - base + k * 64 + lane] += gW3[k];
#pragma unroll
            for (int o = 0; o < OBS; o++) red[L.w1[NET] - base + lane * OBS + o] += gW1[o];
            red[L.b1[NET] - base + lane] += gb1;
            if (hi == 0) { red[L.b2[NET] - base + s] += gb2[0]; red[L.b2[NET] - base + s + 32] += gb2[1]; }
#pragma unroll
            for (int k = 0; k < AMAX; k++) if (k < AOUT) {
                const float t = wave_sum(gb3[k]);
                if (lane == 0) red[L.b3[NET] - base + k] += t;
            }
        }
        __syncthreads();
    }
    const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
    float* slab = a.slab + ((size_t)NET * a.n_blocks_per_net + blockIdx.x) * Pmax;
    for (int e = tid; e < nsz; e += MF_THREADS) slab[e] = red[e];
    // loss sums
    st0 = wave_sum_d(st0); st1 = wave_sum_d(st1); st2 = wave_sum_d(st2); st3 = wave_sum_d(st3);
    __syncthreads();
    double* dred = reinterpret_cast<double*>(smem + ((nsz + 3) & ~3) + 4);
    if (lane == 0) { dred[wave * 4 + 0] = st0; dred[wave * 4 + 1] = st1; dred[wave * 4 + 2] = st2; dred[wave * 4 + 3] = st3; }
    __syncthreads();
    if (tid < 4) {
        double* o = a.stat_slab + ((size_t)NET * a.n_blocks_per_net + blockIdx.x) * 8;
        o[tid] = ((dred[0 + tid] + dred[4 + tid]) + dred[8 + tid]) + dred[12 + tid];
    }
}

template <int DIST, int OBS, int AMAX>
__global__ __launch_bounds__(MF_THREADS, 2) void fwd_bwd_mfma_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.y == 0) mf_body<0, DIST, OBS, 1>(a, smem);
    else mf_body<1, DIST, OBS, AMAX>(a, smem);
}

}  // namespace

int update_blocks_per_net_mfma(int M) {
    const int tiles = (M + MT - 1) / MT;
    const int blocks = (tiles + MF_WAVES - 1) / MF_WAVES;
    return blocks < 256 ? (blocks > 0 ? blocks : 1) : 256;  // one (actor, critic) workgroup pair per CU
}

hipError_t launch_minibatch_fwd_bwd_mfma(const UpdateArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipErrorInvalidValue;
    if (a.L.act > 4) return hipErrorNotSupported;  // wider heads run on the VALU kernel
    const int aout = a.L.act > 1 ? a.L.act : 1;
    const size_t shmem = (size_t)mf_smem(a.L.obs, aout).total * sizeof(float);
    const dim3 grid((unsigned)a.n_blocks_per_net, 2), block(MF_THREADS);
#define PPO_LAUNCH_MF(DIST, OBS)                                                                                       \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_bwd_mfma_kernel<DIST, OBS, 4>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
            if (e != hipSuccess) return e;                                                                             \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((fwd_bwd_mfma_kernel<DIST, OBS, 4>), grid, block, shmem, s, a);                             \
    } while (0)
    if (a.L.obs == 4) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) PPO_LAUNCH_MF(PPO_DIST_CATEGORICAL, 4); else PPO_LAUNCH_MF(PPO_DIST_MASKED, 4);
    } else if (a.L.obs == 2) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) PPO_LAUNCH_MF(PPO_DIST_CATEGORICAL, 2); else PPO_LAUNCH_MF(PPO_DIST_MASKED, 2);
    } else {
        return hipErrorNotSupported;
    }
#undef PPO_LAUNCH_MF
    return hipGetLastError();
}
