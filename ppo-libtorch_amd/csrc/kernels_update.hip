// ppo-libtorch_amd/csrc/kernels_update.hip -- update-side kernels for gfx950 (wave64).
//
//   K5  minibatch gather by permutation index        reference PPO_Discrete.cpp:569-582 (b_*.index({mb_inds}))
//   K6  teacher-forced forward (new log-prob, value) reference Agent.cpp:117-170
//   K7  PPO loss + statistics                        reference PPO_Discrete.cpp:585-631, :343-356
//   K8  backward through both MLPs                   reference PPO_Discrete.cpp:638 (autograd)
//   K9  global-norm clip  + K10 AdamW                reference PPO_Discrete.cpp:640-641 (LibTorch clip_grad.h:22-85, optim/adamw.cpp)
//   K11 explained variance                           reference PPO_Discrete.cpp:647-648
//   plus: minibatch advantage statistics (:591-594), permutation generator (torch::randperm, :569).
//
// K5-K8 are ONE kernel.  Actor and critic are independent networks whose only coupling is the scalar sum of their
// losses (Agent.cpp:25-59, PPO_Discrete.cpp:631), so blockIdx.y selects the net and a workgroup streams 64-sample tiles
// of its net end to end: gather -> layer 1 -> layer 2 -> head -> loss -> d head -> d layer 2 -> d layer 1, with the
// activations of the tile resident in LDS as [unit][sample] images (row stride 68 floats: 16-byte aligned rows that
// spread over the banks) and the 64x64 weight block resident in LDS in both orientations for the whole launch.
// The three 64x64 contractions per tile are register-tiled 4x4 per thread (16 FMAs per two 16-byte LDS reads).
// Weight gradients accumulate in registers across all tiles of the workgroup and are written ONCE as a partial slab;
// a second kernel adds the slabs in a fixed order (bit-reproducible, no float atomics).
#include "ppo_internal.hpp"

namespace {

constexpr int UPD_THREADS = 256;
constexpr int S = PPO_LDS_STRIDE;  // 68
constexpr int TILE = PPO_TILE;     // 64

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct SmemLayout {
    int w2t, w2, w1, b1, b2, w3, b3, x, h1, h2, dout, total;  // offsets in floats
};
__host__ __device__ inline SmemLayout smem_layout(int obs, int aout) {
    SmemLayout m;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.w2t = take(64 * 64);
    m.w2 = take(64 * 64);
    m.w1 = take(64 * obs);
    m.b1 = take(64);
    m.b2 = take(64);
    m.w3 = take(aout * 64);
    m.b3 = take(aout);
    m.x = take(obs * S);
    m.h1 = take(64 * S);
    m.h2 = take(64 * S);
    m.dout = take(aout * S);
    m.total = o;
    return m;
}

template <int NET, int DIST, int OBS>
__device__ __forceinline__ void fwd_bwd_body(const UpdateArgs& a, float* smem) {
    const NetLayout& L = a.L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ts = tid >> 4, tu = tid & 15;
    const int AOUT = NET == 0 ? 1 : L.act;
    const SmemLayout m = smem_layout(OBS, AOUT);
    float* sW2T = smem + m.w2t;
    float* sW2 = smem + m.w2;
    float* sW1 = smem + m.w1;
    float* sB1 = smem + m.b1;
    float* sB2 = smem + m.b2;
    float* sW3 = smem + m.w3;
    float* sB3 = smem + m.b3;
    float* sX = smem + m.x;
    float* sH1 = smem + m.h1;
    float* sH2 = smem + m.h2;
    float* sD = smem + m.dout;
    const float* __restrict__ P = a.params;

    // ---- weights of this net -> LDS (once per launch) ----
    for (int e = tid; e < 64 * 64; e += UPD_THREADS) {
        const float w = P[L.w2[NET] + e];  // [u][k]
        sW2[e] = w;
        sW2T[(e & 63) * 64 + (e >> 6)] = w;
    }
    for (int e = tid; e < 64 * OBS; e += UPD_THREADS) sW1[e] = P[L.w1[NET] + e];
    for (int e = tid; e < AOUT * 64; e += UPD_THREADS) sW3[e] = P[L.w3[NET] + e];
    if (tid < 64) { sB1[tid] = P[L.b1[NET] + tid]; sB2[tid] = P[L.b2[NET] + tid]; }
    if (tid < AOUT) sB3[tid] = P[L.b3[NET] + tid];

    // ---- per-thread gradient accumulators (whole launch) ----
    float gW2[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) gW2[i][j] = 0.0f;
    constexpr int W3R = (PPO_MAX_ACT * 64) / UPD_THREADS;  // 8
    float gW3[W3R];
#pragma unroll
    for (int r = 0; r < W3R; r++) gW3[r] = 0.0f;
    constexpr int W1R = (OBS * 64 + UPD_THREADS - 1) / UPD_THREADS;
    float gW1[W1R];
#pragma unroll
    for (int r = 0; r < W1R; r++) gW1[r] = 0.0f;
    float gb1 = 0.0f, gb2 = 0.0f, gb3 = 0.0f;

    // loss partial sums (wave 0 lanes)
    double st0 = 0.0, st1 = 0.0, st2 = 0.0, st3 = 0.0;  // NET1: pg, entropy, kl, clip count ; NET0: v

    const float clip = a.hp.clip_coef;
    const float lo = 1 - clip, hi = 1 + clip;  // int - float -> float (PPO_Discrete.cpp:598)
    const float invM = (float)a.inv_global_M;
    float mean_f = 0.0f, std_f = 0.0f;
    if (NET == 1 && a.hp.norm_adv) {
        // (adv - mean) / (std + 1e-8) over the minibatch, std Bessel-corrected (PPO_Discrete.cpp:591-594): formed once per update by adv_norm_kernel
        mean_f = a.adv_norm->x;
        std_f = a.adv_norm->z;
    }
    __syncthreads();

    const int n_tiles = (a.M + TILE - 1) / TILE;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---------------- gather (K5): wave 0, lane = sample ----------------
        bool valid = false;
        int row = 0;
        float s_oldlp = 0.0f, s_adv = 0.0f, s_ret = 0.0f, s_oldv = 0.0f;
        int s_act[PPO_MAX_HEADS];
        uint32_t s_mask = 0xffffffffu;
        if (tid < TILE) {
            const int j = tile * TILE + tid;
            valid = j < a.M;
            row = valid ? a.idx[j] : 0;
#pragma unroll
            for (int o = 0; o < OBS; o++) sX[o * S + tid] = valid ? a.obs[(size_t)row * OBS + o] : 0.0f;
            if (NET == 1) {
                s_oldlp = a.logprobs[row];
                s_adv = a.advantages[row];
                for (int h = 0; h < L.n_heads; h++) s_act[h] = a.actions[(size_t)row * L.n_heads + h];
                if (DIST == PPO_DIST_MASKED && a.masks) {
                    s_mask = 0u;
                    for (int k = 0; k < L.act; k++) s_mask |= (a.masks[(size_t)row * L.act + k] ? 1u : 0u) << k;
                }
            } else {
                s_ret = a.returns[row];
                s_oldv = a.values[row];
            }
        }
        __syncthreads();

        // ---------------- layer 1 forward: h1[u][s] = tanh(b1[u] + sum_o x[o][s] W1[u][o]) ----------------
        {
            float xs[OBS][4];
#pragma unroll
            for (int o = 0; o < OBS; o++) {
                const float4 v = ld4(&sX[o * S + ts * 4]);
                xs[o][0] = v.x; xs[o][1] = v.y; xs[o][2] = v.z; xs[o][3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = tu * 4 + j;
                float z[4] = { sB1[u], sB1[u], sB1[u], sB1[u] };
#pragma unroll
                for (int o = 0; o < OBS; o++) {
                    const float w = sW1[u * OBS + o];
#pragma unroll
                    for (int i = 0; i < 4; i++) z[i] = __builtin_fmaf(xs[o][i], w, z[i]);
                }
                st4(&sH1[u * S + ts * 4], make_float4(tanhf(z[0]), tanhf(z[1]), tanhf(z[2]), tanhf(z[3])));
            }
        }
        __syncthreads();

        // ---------------- layer 2 forward: h2[u][s] = tanh(b2[u] + sum_k h1[k][s] W2[u][k]) ----------------
        {
            float acc[4][4];  // [sample i][unit j]
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float b = sB2[tu * 4 + j];
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][j] = b;
            }
#pragma unroll 8
            for (int k = 0; k < 64; k++) {
                const float4 h = ld4(&sH1[k * S + ts * 4]);
                const float4 w = ld4(&sW2T[k * 64 + tu * 4]);
                const float hv[4] = { h.x, h.y, h.z, h.w }, wv[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_fmaf(hv[i], wv[j], acc[i][j]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                st4(&sH2[(tu * 4 + j) * S + ts * 4], make_float4(tanhf(acc[0][j]), tanhf(acc[1][j]), tanhf(acc[2][j]), tanhf(acc[3][j])));
        }
        __syncthreads();

        // ---------------- head + loss + d(head output) (K6, K7): wave 0, lane = sample ----------------
        if (tid < TILE) {
            if (NET == 0) {
                float v = sB3[0];
#pragma unroll 8
                for (int u = 0; u < 64; u++) v = __builtin_fmaf(sH2[u * S + tid], sW3[u], v);
                // value loss, PPO_Discrete.cpp:603-625
                const float R = s_ret, vold = s_oldv;
                const float un = (v - R) * (v - R);
                float g_v;
                float lossv;
                if (a.hp.clip_vloss) {
                    const float dv = v - vold;
                    const float dvc = dv < -clip ? -clip : (dv > clip ? clip : dv);
                    const float vc = vold + dvc;
                    const float cl = (vc - R) * (vc - R);
                    lossv = un > cl ? un : cl;
                    const bool vin = (dv >= -clip && dv <= clip);
                    const float d_un = 2.0f * (v - R), d_cl = vin ? 2.0f * (vc - R) : 0.0f;
                    const float d = un > cl ? d_un : (un < cl ? d_cl : 0.5f * d_un + 0.5f * d_cl);  // torch::max splits ties
                    g_v = a.hp.vf_coef * 0.5f * invM * d;
                } else {
                    lossv = un;
                    g_v = a.hp.vf_coef * 0.5f * invM * 2.0f * (v - R);
                }
                if (valid) st0 += (double)lossv;
                sD[tid] = valid ? g_v : 0.0f;
            } else {
                float z[PPO_MAX_ACT], p[PPO_MAX_ACT], headH[PPO_MAX_HEADS];
                for (int k = 0; k < L.act; k++) {
                    float acc = sB3[k];
#pragma unroll 8
                    for (int u = 0; u < 64; u++) acc = __builtin_fmaf(sH2[u * S + tid], sW3[k * 64 + u], acc);
                    z[k] = acc;
                }
                uint8_t mk[PPO_MAX_ACT];
                for (int k = 0; k < L.act; k++) mk[k] = (uint8_t)((s_mask >> k) & 1u);
                const bool use_mask = (DIST == PPO_DIST_MASKED) && a.masks != nullptr;
                float nlp = 0.0f, ent = 0.0f;
                int off = 0;
                for (int h = 0; h < L.n_heads; h++) {
                    const int Ah = L.head_dims[h];
                    headH[h] = categorical_head<DIST>(z + off, p + off, use_mask ? mk + off : nullptr, Ah);
                    const float l1 = z[off + s_act[h]];
                    if (h == 0) { nlp = l1; ent = headH[h]; } else { nlp += l1; ent += headH[h]; }
                    off += Ah;
                }
                // PPO_Discrete.cpp:585-599
                const float logratio = nlp - s_oldlp;
                const float ratio = expf(logratio);
                float adv = s_adv;
                if (a.hp.norm_adv) adv = (adv - mean_f) / (std_f + 1e-8f);
                const float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
                const float l1 = -adv * ratio, l2 = -adv * rc;
                const bool inside = (ratio >= lo && ratio <= hi);
                float d_ratio;
                if (l1 > l2) d_ratio = -adv;
                else if (l1 < l2) d_ratio = inside ? -adv : 0.0f;
                else d_ratio = 0.5f * -adv + (inside ? 0.5f * -adv : 0.0f);
                const float g_nlp = invM * d_ratio * ratio;
                const float g_ent = -a.hp.ent_coef * invM;
                if (valid) {
                    st0 += (double)(l1 > l2 ? l1 : l2);
                    st1 += (double)ent;
                    st2 += (double)((ratio - 1.0f) - logratio);         // :352
                    st3 += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;     // :348
                }
                off = 0;
                for (int h = 0; h < L.n_heads; h++) {
                    const int Ah = L.head_dims[h];
                    for (int k = 0; k < Ah; k++) {
                        const bool ok = !use_mask || mk[off + k];
                        const float pk = p[off + k];
                        float d = g_nlp * ((k == s_act[h] ? 1.0f : 0.0f) - pk);
                        // CategoricalMasked entropy is the true entropy: dH/dz_k = -p_k (log p_k + H).  The plain Categorical's
                        // clamp turns entropy into -FLT_MIN*sum(p): gradient is a denormal times ent_coef/M, i.e. zero.
                        if (DIST == PPO_DIST_MASKED) d += g_ent * (-pk * (z[off + k] + headH[h]));
                        sD[(off + k) * S + tid] = (valid && ok) ? d : 0.0f;
                    }
                    off += Ah;
                }
            }
        }
        __syncthreads();

        // ---------------- dW3, db3 (reads h2 before it is overwritten) ----------------
#pragma unroll
        for (int r = 0; r < W3R; r++) {
            const int q = tid + r * UPD_THREADS;
            if (q < AOUT * 64) {
                const int aa = q >> 6, u = q & 63;
                float acc = 0.0f;
#pragma unroll
                for (int s4 = 0; s4 < TILE; s4 += 4) {
                    const float4 d = ld4(&sD[aa * S + s4]);
                    const float4 h = ld4(&sH2[u * S + s4]);
                    acc = __builtin_fmaf(d.x, h.x, acc); acc = __builtin_fmaf(d.y, h.y, acc);
                    acc = __builtin_fmaf(d.z, h.z, acc); acc = __builtin_fmaf(d.w, h.w, acc);
                }
                gW3[r] += acc;
            }
        }
        if (tid < AOUT) {
            float acc = 0.0f;
#pragma unroll
            for (int s4 = 0; s4 < TILE; s4 += 4) { const float4 d = ld4(&sD[tid * S + s4]); acc += (d.x + d.y) + (d.z + d.w); }
            gb3 += acc;
        }
        __syncthreads();

        // ---------------- dz2[u][s] = (sum_a dOut[a][s] W3[a][u]) * (1 - h2^2), in place of h2 ----------------
        {
            float dh[4][4];  // [unit j][sample i]
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) dh[j][i] = 0.0f;
            for (int aa = 0; aa < AOUT; aa++) {
                const float4 d = ld4(&sD[aa * S + ts * 4]);
                const float dv[4] = { d.x, d.y, d.z, d.w };
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float w = sW3[aa * 64 + tu * 4 + j];
#pragma unroll
                    for (int i = 0; i < 4; i++) dh[j][i] = __builtin_fmaf(dv[i], w, dh[j][i]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float* ph = &sH2[(tu * 4 + j) * S + ts * 4];
                const float4 h = ld4(ph);
                st4(ph, make_float4(dh[j][0] * (1.0f - h.x * h.x), dh[j][1] * (1.0f - h.y * h.y), dh[j][2] * (1.0f - h.z * h.z),
                                    dh[j][3] * (1.0f - h.w * h.w)));
            }
        }
        __syncthreads();

        // ---------------- dW2[u][k] += sum_s dz2[u][s] h1[k][s]   (u = tu + 16 j, k = ts + 16 i) ; db2 ----------------
#pragma unroll 4
        for (int s4 = 0; s4 < TILE; s4 += 4) {
            float4 dz[4], hh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) dz[j] = ld4(&sH2[(tu + 16 * j) * S + s4]);
#pragma unroll
            for (int i = 0; i < 4; i++) hh[i] = ld4(&sH1[(ts + 16 * i) * S + s4]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float acc = gW2[j][i];
                    acc = __builtin_fmaf(dz[j].x, hh[i].x, acc); acc = __builtin_fmaf(dz[j].y, hh[i].y, acc);
                    acc = __builtin_fmaf(dz[j].z, hh[i].z, acc); acc = __builtin_fmaf(dz[j].w, hh[i].w, acc);
                    gW2[j][i] = acc;
                }
        }
        if (tid < 64) {
            float acc = 0.0f;
#pragma unroll
            for (int s4 = 0; s4 < TILE; s4 += 4) { const float4 d = ld4(&sH2[tid * S + s4]); acc += (d.x + d.y) + (d.z + d.w); }
            gb2 += acc;
        }

        // ---------------- dh1[k][s] = sum_u dz2[u][s] W2[u][k]  (registers; h1 is still being read by dW2) ----------------
        float dh1[4][4];  // [unit k_j][sample i]
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) dh1[j][i] = 0.0f;
#pragma unroll 8
        for (int u = 0; u < 64; u++) {
            const float4 d = ld4(&sH2[u * S + ts * 4]);
            const float4 w = ld4(&sW2[u * 64 + tu * 4]);
            const float dv[4] = { d.x, d.y, d.z, d.w }, wv[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) dh1[j][i] = __builtin_fmaf(dv[i], wv[j], dh1[j][i]);
        }
        __syncthreads();
        // dz1 = dh1 * (1 - h1^2), in place of h1
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float* ph = &sH1[(tu * 4 + j) * S + ts * 4];
            const float4 h = ld4(ph);
            st4(ph, make_float4(dh1[j][0] * (1.0f - h.x * h.x), dh1[j][1] * (1.0f - h.y * h.y), dh1[j][2] * (1.0f - h.z * h.z),
                                dh1[j][3] * (1.0f - h.w * h.w)));
        }
        __syncthreads();

        // ---------------- dW1[u][o] += sum_s dz1[u][s] x[o][s] ; db1 ----------------
#pragma unroll
        for (int r = 0; r < W1R; r++) {
            const int q = tid + r * UPD_THREADS;
            if (q < OBS * 64) {
                const int u = q & 63, o = q >> 6;
                float acc = 0.0f;
#pragma unroll
                for (int s4 = 0; s4 < TILE; s4 += 4) {
                    const float4 d = ld4(&sH1[u * S + s4]);
                    const float4 x = ld4(&sX[o * S + s4]);
                    acc = __builtin_fmaf(d.x, x.x, acc); acc = __builtin_fmaf(d.y, x.y, acc);
                    acc = __builtin_fmaf(d.z, x.z, acc); acc = __builtin_fmaf(d.w, x.w, acc);
                }
                gW1[r] += acc;
            }
        }
        if (tid < 64) {
            float acc = 0.0f;
#pragma unroll
            for (int s4 = 0; s4 < TILE; s4 += 4) { const float4 d = ld4(&sH1[tid * S + s4]); acc += (d.x + d.y) + (d.z + d.w); }
            gb1 += acc;
        }
        __syncthreads();
    }

    // ---------------- write this workgroup's partial gradient slab (net-local flat layout) ----------------
    const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
    float* slab = a.slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blockIdx.x) * Pmax;
    const int base = L.net_off[NET];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) slab[L.w2[NET] - base + (tu + 16 * j) * 64 + (ts + 16 * i)] = gW2[j][i];
#pragma unroll
    for (int r = 0; r < W3R; r++) {
        const int q = tid + r * UPD_THREADS;
        if (q < AOUT * 64) slab[L.w3[NET] - base + q] = gW3[r];
    }
#pragma unroll
    for (int r = 0; r < W1R; r++) {
        const int q = tid + r * UPD_THREADS;
        if (q < OBS * 64) slab[L.w1[NET] - base + (q & 63) * OBS + (q >> 6)] = gW1[r];
    }
    if (tid < 64) { slab[L.b1[NET] - base + tid] = gb1; slab[L.b2[NET] - base + tid] = gb2; }
    if (tid < AOUT) slab[L.b3[NET] - base + tid] = gb3;
    if (tid < 64) {
        st0 = wave_sum_d(st0); st1 = wave_sum_d(st1); st2 = wave_sum_d(st2); st3 = wave_sum_d(st3);
        if (lane == 0) {
            double* o = a.stat_slab + ((size_t)(NET == 0 ? 0 : a.n_blocks[0]) + blockIdx.x) * 8;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
        }
    }
}

template <int DIST, int OBS>
__global__ __launch_bounds__(UPD_THREADS, 2) void fwd_bwd_kernel(UpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.y == 0) fwd_bwd_body<0, DIST, OBS>(a, smem);
    else fwd_bwd_body<1, DIST, OBS>(a, smem);
}

// grads[p] = sum_b slab[net(p)][b][p - net_off] in a fixed order (16 contiguous groups of slabs, each summed in order by one
// wave with its loads in flight together, then the 16 partials added in order): bit-reproducible, no float atomics.
// The last block adds the per-workgroup loss sums the same way.
// The slab reductions (this kernel and reduce_grads_sumsq_kernel, which must add in the same order: the fused and the step-by-step paths are
// bit-identical) run RED_WAVES waves per workgroup, each adding 1 / RED_WAVES of the slabs with all its loads in flight (measured, A/B in one call, per optimizer
// step by events: 16 waves 13.4 us, 8 waves 12.7, 4 waves 13.2 and erratic)
constexpr int RED_WAVES = 8;
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_grads_kernel(const float* __restrict__ slab, const double* __restrict__ stat_slab, int nb0, int nb1,
                                                            NetLayout L, float* __restrict__ grads, double* __restrict__ sums_out) {
    __shared__ double part[RED_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (blockIdx.x + 1 < gridDim.x) {
        const int p = blockIdx.x * 64 + lane;
        const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
        double acc = 0.0;
        if (p < L.P) {
            const bool net1 = p >= L.net_off[1];   // selects, not L.net_off[net]: see reduce_grads_sumsq_kernel
            const int nb = net1 ? nb1 : nb0;
            const float* col = slab + (size_t)(net1 ? nb0 : 0) * Pmax + (p - (net1 ? L.net_off[1] : L.net_off[0]));
            const int b0 = (nb * w) / RED_WAVES, b1 = (nb * (w + 1)) / RED_WAVES;
            int b = b0;
            for (; b + 16 <= b1; b += 16) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = col[(size_t)(b + i) * Pmax];
#pragma unroll
                for (int i = 0; i < 16; i++) acc += (double)v[i];
            }
            for (; b + 4 <= b1; b += 4) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = col[(size_t)(b + i) * Pmax];
#pragma unroll
                for (int i = 0; i < 4; i++) acc += (double)v[i];
            }
            for (; b < b1; b++) acc += (double)col[(size_t)b * Pmax];
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0 && p < L.P) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < RED_WAVES; k++) s += part[k][lane];
            grads[p] = (float)s;
        }
    } else {
        // loss sums: [0]=pg [1]=entropy [2]=kl [3]=clip count (actor workgroups), [4]=value loss (critic workgroups): the five columns together, one
        // round trip and one barrier, in the order reduce_grads_sumsq_kernel adds them (the fused and the step-by-step paths give the same bits)
        double v5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
        for (int b = threadIdx.x; b < nb0 || b < nb1; b += 64 * RED_WAVES) {
            double t[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
            if (b < nb1) {
                const double* r = stat_slab + ((size_t)nb0 + b) * 8;
                t[0] = r[0]; t[1] = r[1]; t[2] = r[2]; t[3] = r[3];
            }
            if (b < nb0) t[4] = stat_slab[(size_t)b * 8];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) v5[kk] += t[kk];
        }
#pragma unroll
        for (int kk = 0; kk < 5; kk++) {
            const double r = wave_sum_d_dpp(v5[kk]);
            if (lane == 0) part[w][kk] = r;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            double sacc = 0.0;
            for (int i = 0; i < RED_WAVES; i++) sacc += part[i][threadIdx.x];
            sums_out[threadIdx.x] = sacc;
            grads[L.P + threadIdx.x] = (float)sacc;   // float copies ride behind the gradient so ONE all-reduce carries both
        }
        if (threadIdx.x >= 5 && threadIdx.x < 8) { sums_out[threadIdx.x] = 0.0; grads[L.P + threadIdx.x] = 0.0f; }
    }
}

// K9: per-tensor squared L2 norms of the gradient (clip_grad.h:58-66), one workgroup per tensor, every load in flight at once.
constexpr int NORM_THREADS = 256;
__global__ __launch_bounds__(NORM_THREADS) void grad_norm_kernel(const float* __restrict__ grads, NetLayout L, double* __restrict__ norm2) {
    __shared__ double red[NORM_THREADS / 64];
    const int t = blockIdx.x;
    const int a0 = L.tensor_off[t], a1 = L.tensor_off[t + 1];
    double acc = 0.0;
    for (int p0 = a0 + threadIdx.x; p0 < a1; p0 += 16 * NORM_THREADS) {
        float g[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { const int p = p0 + i * NORM_THREADS; g[i] = p < a1 ? grads[p] : 0.0f; }
#pragma unroll
        for (int i = 0; i < 16; i++) acc += (double)g[i] * (double)g[i];
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) norm2[t] = ((red[0] + red[1]) + red[2]) + red[3];
}

// K9 (clip coefficient) + K10 (AdamW), element-parallel.  GRADS is left holding the UNCLIPPED gradient (the reference scales
// .grad in place, clip_grad.h:79-81, but nothing reads it before zero_grad); the clip coefficient is applied on the fly.
constexpr int ADAM_THREADS = 256;   // four waves: opt_total_norm spreads the twelve tensors' partials over waves 0..3
// the fp16-range maxima of OptGuard (ppo_internal.hpp) from scratch, after the host wrote parameters
__global__ void weight_range_kernel(const float* __restrict__ params, NetLayout L, uint32_t* wr_dev, uint32_t* wr_host) {
    __shared__ uint32_t m[3];
    if (threadIdx.x < 3) m[threadIdx.x] = 0u;
    __syncthreads();
    for (int p = threadIdx.x; p < L.P; p += blockDim.x) atomicMax(&m[wr_class(L, p)], __builtin_bit_cast(uint32_t, params[p]) & 0x7fffffffu);
    __syncthreads();
    if (threadIdx.x < 3) {
        wr_dev[threadIdx.x] = m[threadIdx.x];
        wr_dev[4 + threadIdx.x] = m[threadIdx.x];   // what the host mirror holds (opt_mirror_range)
        __hip_atomic_store(wr_host + threadIdx.x, m[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(ADAM_THREADS) void clip_adamw_kernel(float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ exp_avg,
                                                                  float* __restrict__ exp_avg_sq, NetLayout L, float max_norm,
                                                                  const double* __restrict__ norm2, const AdamCoef* __restrict__ coef_p,
                                                                  const double* __restrict__ loss_sums, double global_M, LossParams hp,
                                                                  int world, int do_step, StepStats* stats_out, double* clipfrac_accum, OptGuard guard) {
    const int tid = threadIdx.x;
    // this thread's element, the norms and the statistics scalars are all requested before anything is consumed
    const int pu = blockIdx.x * ADAM_THREADS + tid;
    bool upd = do_step && pu < L.P;
    float u_g = 0.0f, u_p = 0.0f, u_m = 0.0f, u_v = 0.0f;
    if (upd) { u_g = grads[pu]; u_p = params[pu]; u_m = exp_avg[pu]; u_v = exp_avg_sq[pu]; }
    const int32_t err = opt_guard_word(guard.error_flag);   // requested with the step's other loads
    upd = upd && !(err & PPO_ERRFLAG_SKIP_STEP);   // a step the device knows is garbage is not applied (OptGuard, ppo_internal.hpp)
    double n2[12];
#pragma unroll
    for (int t = 0; t < 12; t++) n2[t] = t < L.n_tensors ? norm2[t] : 0.0;
    const AdamCoef k = *coef_p;
    double ls[5] = { 0, 0, 0, 0, 0 }, cf0 = 0.0, cf1 = 0.0;
    const bool stat_thread = tid == 0 && blockIdx.x == 0;
    if (stat_thread) {
        if (world > 1) { for (int i = 0; i < 5; i++) ls[i] = grads[L.P + i]; }   // sums travelled (as floats) behind the gradient
        else { for (int i = 0; i < 5; i++) ls[i] = loss_sums[i]; }
        if (clipfrac_accum && do_step) { cf0 = clipfrac_accum[0]; cf1 = clipfrac_accum[1]; }
    }
    // total_norm = || (||g_1||, ..., ||g_12||) ||_2 with float per-tensor norms (clip_grad.h:58-70); every thread forms the same bits.  Lane t of a
    // wave forms tensor t's norm (one binary64 sqrt per wave, not twelve in a row), the squares are added in tensor order through scalar registers.
    double sq = 0.0;
    {
        const int ln = tid & 63;
        double mine = 0.0;
#pragma unroll
        for (int t = 0; t < 12; t++) mine = ln == t ? n2[t] : mine;
        if (ln < 12) { const float nrm = (float)sqrt(mine); sq = (double)nrm * nrm; }
    }
    double tot = 0.0;
#pragma unroll
    for (int t = 0; t < 12; t++)
        tot += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sq), t), __builtin_amdgcn_readlane(__double2loint(sq), t));
    const float total = (float)sqrt(tot);
    float c = max_norm / (total + 1e-6f);   // clip_grad.h:76-78
    if (c > 1.0f) c = 1.0f;
    const float b1 = 0.9f, b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-5f;
    if (upd) {
        const float g = u_g * c;                   // param.grad().mul_(clip_coef_clamped), clip_grad.h:79-81
        const float pi = u_p * k.decay;            // p.mul_(1 - lr*wd)
        const float mi = __builtin_fmaf(g, omb1, u_m * b1);          // exp_avg.mul_(b1).add_(g, 1-b1)   (ATen fmadd)
        const float vi = __builtin_fmaf(omb2 * g, g, u_v * b2);      // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        const float denom = sqrtf(vi) / k.sqrt_bc2 + eps;
        const float pn = pi + (k.neg_step * mi) / denom;             // addcdiv_(exp_avg, denom, -step_size)
        params[pu] = pn;
        exp_avg[pu] = mi;
        exp_avg_sq[pu] = vi;
    }
    if (stat_thread) {
        // scalars as the reference forms them (PPO_Discrete.cpp:599,619,628,631,349,352), means over the global minibatch
        const float pg = (float)(ls[0] / global_M);
        const float vl = 0.5f * (float)(ls[4] / global_M);
        const float el = (float)(ls[1] / global_M);
        StepStats o;
        o.pg_loss = pg;
        o.v_loss = vl;
        o.entropy_loss = el;
        o.approx_kl = (float)(ls[2] / global_M);
        o.clipfrac = (float)ls[3] / (float)global_M;
        o.loss = (pg - hp.ent_coef * el) + vl * hp.vf_coef;
        o.total_norm = total;
        o.pad = 0.0;
        *stats_out = o;
        if (clipfrac_accum && do_step) { clipfrac_accum[0] = cf0 + o.clipfrac; clipfrac_accum[1] = cf1 + 1.0; }  // m_clipfracs (:349), mean at :755
    }
}

// Single-rank optimizer step in TWO launches instead of three: the slab reduction also leaves, per workgroup, the sums of squares
// of its 64 gradients split by tensor (K8-reduce + the first half of K9), and the AdamW kernel adds those partials in a fixed order
// instead of re-reading the gradient in a norm kernel of its own.  (With several ranks the RCCL all-reduce changes the gradient
// between the two, so that path keeps reduce -> all-reduce -> norm -> AdamW.)  A one-launch version with a grid-wide rendezvous
// was measured slower (18 us against 16: two device-scope fences and the polling cost more than the launch they save), and so was
// a last-arriver version without any waiting (every workgroup publishes, takes a ticket, the last one applies AdamW to all P
// parameters: 23 us with fences, 21 us with device-coherent stores / loads and no fence at all) -- on this part every
// device-scope operation (store acknowledgement, atomic, coherent load) is a 1-2 us trip to the memory side, and three of them in a
// row cost more than a kernel boundary.
struct FusedOptArgs {
    const float* slab; const double* stat_slab; int nb0, nb1;
    NetLayout L;
    float* grads; double* sums_out;
    const float* p_src; const float* m_src; const float* v_src;   // optimizer state before the step ...
    float* params; float* exp_avg; float* exp_avg_sq;             // ... and where the state after it goes (may be the same buffers)
    float max_norm; const AdamCoef* coef; double global_M; LossParams hp;
    StepStats* stats_out; double* clipfrac_accum;
    double* partial;            // [reduce workgroups][12] sums of squares
    // XCHG: the one-shot direct exchange of exchange_allreduce_kernel (below), folded into the reduction: a sharded step is then the same two
    // launches as a single-rank one -- reduce + exchange + sums of squares, then norm + clip + AdamW
    void* peers[8]; int rank, n_ranks; size_t slot_bytes; unsigned long long seq; int32_t* timeout_flag;
    OptGuard guard;
};
// ---- the exchange buffer of a rank (fine-grained device memory, mapped by every peer; ExchangeComm in api.hip) ----
//   payload  [2 parities][8 source ranks][slot_bytes]     rank s WRITES its share of call k into slot [k & 1][s] of EVERY rank's buffer
//   flags    u64 [2][8]                                    ... then stamps flag [k & 1][s] of every rank's buffer with k
//   counters u32 [2]                                       arrivals of the owner's own workgroups (local)
// A rank PUSHES: stores over the links are posted, and what it then waits for and reads is its OWN memory -- one one-way trip per
// all-reduce, where pulling (poll the peer's flag, then read the peer's payload) paid two round trips and kept the links busy with polls.
// Reuse: a rank writes call k + 2 into a peer's parity-k slot only after its call k + 1 completed, i.e. after every peer had published
// k + 1 -- which a peer does in a kernel that follows, in stream order, the one in which it read call k.
__device__ __forceinline__ char* xchg_slot(void* base, size_t slot_bytes, int parity, int src) { return static_cast<char*>(base) + ((size_t)parity * 8 + src) * slot_bytes; }
__device__ __forceinline__ unsigned long long* xchg_flag(void* base, size_t slot_bytes, int parity, int src) {
    return reinterpret_cast<unsigned long long*>(static_cast<char*>(base) + 16 * slot_bytes) + parity * 8 + src;
}
__device__ __forceinline__ unsigned int* xchg_count(void* base, size_t slot_bytes, int parity) {
    return reinterpret_cast<unsigned int*>(static_cast<char*>(base) + 16 * slot_bytes + 128) + parity;
}
// The caller's stores into the peers' slots are done: make them visible system-wide, count this workgroup in (one lane), and let the last
// workgroup of the grid stamp this rank's flag in every peer's buffer (lane p stamps peer p).  Called by one whole wave per workgroup.
__device__ __forceinline__ void xchg_publish_done(void* const* peers, int rank, int n, size_t slot_bytes, int parity, unsigned long long seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    const int lane = threadIdx.x & 63;
    int last = 0;
    if (lane == 0) {
        unsigned int* cnt = xchg_count(peers[rank], slot_bytes, parity);
        const unsigned int arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) { __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1; }
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last && lane < n && lane != rank) __hip_atomic_store(xchg_flag(peers[lane], slot_bytes, parity, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Waits (bounded) until every peer's share of call `seq` has landed in THIS rank's buffer: lane r of the calling wave polls local flag r.
// Returns the mask of peers whose slot may be read; a wait that runs out sets timeout_flag[0] (zero / non-zero: every polling lane of every
// workgroup that gives up adds to it) and marks the communicator dead: later calls do not wait again, and the host reports PPO_ERR_COMM from
// its next synchronising call (api.hip: comm_health).  The limit, in ticks of the 100 MHz s_memrealtime counter, sits behind the flag as a
// u64 at timeout_flag + 2 (ppo_comm_set_wait_limit; default 30 s -- larger than any host-side skew between ranks such as a checkpoint write).
__device__ __forceinline__ unsigned xchg_wait_all(void* own, int rank, int n, size_t slot_bytes, int parity, unsigned long long seq, int32_t* timeout_flag) {
    const int lane = threadIdx.x & 63;
    const bool dead = __hip_atomic_load(timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const unsigned long long limit = *reinterpret_cast<const unsigned long long*>(timeout_flag + 2);
    int ok = 1;
    if (lane < n && lane != rank) {
        const unsigned long long* flag = xchg_flag(own, slot_bytes, parity, lane);
        ok = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        for (;;) {
            if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { ok = 1; break; }
            if (dead || __builtin_amdgcn_s_memrealtime() - t0 > limit) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) atomicAdd(timeout_flag, 1);
    }
    const unsigned long long m = __ballot(ok != 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // every lane reads the slots after the polling lanes saw the flags
    return (unsigned)(m & 0xffu) | ~0xffu;
}
template <bool XCHG>
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_grads_sumsq_kernel(FusedOptArgs a) {
    __shared__ double part[RED_WAVES][64];
    const NetLayout& L = a.L;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (blockIdx.x + 1 < gridDim.x) {
        const int p = blockIdx.x * 64 + lane;
        const int Pmax = L.net_size[0] > L.net_size[1] ? L.net_size[0] : L.net_size[1];
        double acc = 0.0;
        if (p < L.P) {
            // (selects between the two nets' constants, not L.net_off[net]: indexing the kernel-argument struct with a per-lane value is a VECTOR load
            // from the argument segment -- a memory round trip in front of the slab loads)
            const bool net1 = p >= L.net_off[1];
            const int nb = net1 ? a.nb1 : a.nb0;
            const float* col = a.slab + (size_t)(net1 ? a.nb0 : 0) * Pmax + (p - (net1 ? L.net_off[1] : L.net_off[0]));
            const int b0 = (nb * w) / RED_WAVES, b1 = (nb * (w + 1)) / RED_WAVES;
            int b = b0;
            for (; b + 16 <= b1; b += 16) {   // a wave's 16 slabs of the 128-workgroup update kernel: ONE batch of loads, one round trip
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = col[(size_t)(b + i) * Pmax];
#pragma unroll
                for (int i = 0; i < 16; i++) acc += (double)v[i];
            }
            for (; b + 4 <= b1; b += 4) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = col[(size_t)(b + i) * Pmax];
#pragma unroll
                for (int i = 0; i < 4; i++) acc += (double)v[i];
            }
            for (; b < b1; b++) acc += (double)col[(size_t)b * Pmax];
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0) {
            float g = 0.0f;
            if (p < L.P) {
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < RED_WAVES; i++) sacc += part[i][lane];
                g = (float)sacc;
            }
            if constexpr (XCHG) {
                // this rank's share goes into its slot of every peer's buffer; the sum over ranks is then formed in rank order out of the OWN buffer
                const int parity = (int)(a.seq & 1ull);
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (r < a.n_ranks && r != a.rank && p < L.P)
                        __hip_atomic_store(reinterpret_cast<float*>(xchg_slot(a.peers[r], a.slot_bytes, parity, a.rank)) + p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                xchg_publish_done(a.peers, a.rank, a.n_ranks, a.slot_bytes, parity, a.seq);
                const unsigned okm = xchg_wait_all(a.peers[a.rank], a.rank, a.n_ranks, a.slot_bytes, parity, a.seq, a.timeout_flag);
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    v[r] = 0.0f;
                    if (r < a.n_ranks && r != a.rank && ((okm >> r) & 1u) && p < L.P)
                        v[r] = __hip_atomic_load(reinterpret_cast<const float*>(xchg_slot(a.peers[a.rank], a.slot_bytes, parity, r)) + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                float acc = 0.0f;
#pragma unroll
                for (int r = 0; r < 8; r++) if (r < a.n_ranks) acc += r == a.rank ? g : v[r];   // rank order: every rank forms the same sum
                g = acc;
            }
            if (p < L.P) a.grads[p] = g;
            // sums of squares per tensor of this workgroup's 64 gradients (a workgroup touches at most a few tensors)
            int tl = 0;
#pragma unroll
            for (int t = 0; t < 12; t++) if (t < L.n_tensors && p >= L.tensor_off[t]) tl = t;   // fixed bound: the offsets stay scalar constants
            const double g2 = p < L.P ? (double)g * (double)g : 0.0;
            for (int t = 0; t < 12; t++) {
                double v = 0.0;
                const bool touches = t < L.n_tensors && L.tensor_off[t] < (int)(blockIdx.x + 1) * 64 && L.tensor_off[t + 1] > (int)blockIdx.x * 64;
                if (touches) v = wave_sum_d_dpp(tl == t ? g2 : 0.0);
                if (lane == 0) a.partial[(size_t)blockIdx.x * 12 + t] = v;
            }
        }
    } else {
        // loss sums: [0]=pg [1]=entropy [2]=kl [3]=clip count (actor workgroups), [4]=value loss (critic workgroups).  All five columns are requested
        // together and reduced together: one memory round trip and one barrier (they used to be five of each, one column after the other --
        // this workgroup alone set the kernel's duration).  Fixed order: a thread's workgroups in index order, lanes on the DPP network, waves in order.
        double v5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
        for (int b = threadIdx.x; b < a.nb0 || b < a.nb1; b += 64 * RED_WAVES) {
            double t[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
            if (b < a.nb1) {
                const double* r = a.stat_slab + ((size_t)a.nb0 + b) * 8;
                t[0] = r[0]; t[1] = r[1]; t[2] = r[2]; t[3] = r[3];
            }
            if (b < a.nb0) t[4] = a.stat_slab[(size_t)b * 8];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) v5[kk] += t[kk];
        }
#pragma unroll
        for (int kk = 0; kk < 5; kk++) {
            const double r = wave_sum_d_dpp(v5[kk]);
            if (lane == 0) part[w][kk] = r;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            double sacc = 0.0;
            for (int i = 0; i < RED_WAVES; i++) sacc += part[i][threadIdx.x];
            a.sums_out[threadIdx.x] = sacc;
            part[RED_WAVES - 1][8 + threadIdx.x] = sacc;   // for the exchange below (a slot no wave's partial uses)
        }
        if (threadIdx.x >= 5 && threadIdx.x < 8) a.sums_out[threadIdx.x] = 0.0;
        if constexpr (XCHG) {
            // the loss sums ride the exchange as eight floats behind the gradient (what ppo_allreduce_grads carries on the unfused path)
            __syncthreads();
            if (threadIdx.x < 64) {
                const int parity = (int)(a.seq & 1ull);
                const float mine = threadIdx.x < 5 ? (float)part[RED_WAVES - 1][8 + threadIdx.x] : 0.0f;
                for (int r = 0; r < a.n_ranks; r++)
                    if (r != a.rank && threadIdx.x < 8)
                        __hip_atomic_store(reinterpret_cast<float*>(xchg_slot(a.peers[r], a.slot_bytes, parity, a.rank)) + L.P + threadIdx.x, mine, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                xchg_publish_done(a.peers, a.rank, a.n_ranks, a.slot_bytes, parity, a.seq);
                const unsigned okm = xchg_wait_all(a.peers[a.rank], a.rank, a.n_ranks, a.slot_bytes, parity, a.seq, a.timeout_flag);
                float acc = 0.0f;
                for (int r = 0; r < a.n_ranks; r++) {
                    if (r == a.rank) { acc += mine; continue; }
                    if (!((okm >> r) & 1u)) continue;
                    if (threadIdx.x < 8)
                        acc += __hip_atomic_load(reinterpret_cast<const float*>(xchg_slot(a.peers[a.rank], a.slot_bytes, parity, r)) + L.P + threadIdx.x, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (threadIdx.x < 8) a.sums_out[threadIdx.x] = (double)acc;
            }
        }
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void clip_adamw_sumsq_kernel(FusedOptArgs a) {
    __shared__ double n2s[12];
    const NetLayout& L = a.L;
    const int tid = threadIdx.x;
    const int p = blockIdx.x * ADAM_THREADS + tid;
    const bool own = p < L.P;
    float u_g = 0.0f, u_p = 0.0f, u_m = 0.0f, u_v = 0.0f;
    if (own) { u_g = a.grads[p]; u_p = a.p_src[p]; u_m = a.m_src[p]; u_v = a.v_src[p]; }
    const int32_t err = opt_guard_word(a.guard.error_flag);   // requested with the step's other loads
    const AdamCoef k = *a.coef;
    double ls[5] = { 0, 0, 0, 0, 0 }, cf0 = 0.0, cf1 = 0.0;
    const bool stat_thread = tid == 0 && blockIdx.x == 0;
    if (stat_thread) {
        for (int i = 0; i < 5; i++) ls[i] = a.sums_out[i];
        if (a.clipfrac_accum) { cf0 = a.clipfrac_accum[0]; cf1 = a.clipfrac_accum[1]; }
    }
    const float total = opt_total_norm(L, a.partial, n2s, tid);
    const float c = opt_clip_coef(total, a.max_norm);
    if (own && !(err & PPO_ERRFLAG_SKIP_STEP)) {   // a step the device knows is garbage is not applied (OptGuard, ppo_internal.hpp)
        adamw_apply(u_g, c, k, u_p, u_m, u_v);
        a.params[p] = u_p;
        a.exp_avg[p] = u_m;
        a.exp_avg_sq[p] = u_v;
    }
    if (stat_thread) opt_write_stats(ls, cf0, cf1, a.global_M, a.hp, total, a.stats_out, a.clipfrac_accum);
}

// Sum and sum of squares of the advantages of every minibatch of the coming update: PPO_ADV_PARTS workgroups per minibatch,
// each over a contiguous slice, eight gathered loads in flight per thread.
__global__ __launch_bounds__(256) void adv_stats_kernel(const float* __restrict__ adv, const int32_t* __restrict__ perm, int64_t B,
                                                        int64_t MB, int n_mb_per_epoch, AdvStat* out) {
    __shared__ double r1[4], r2[4];
    const int mb = blockIdx.x, part = blockIdx.y;
    const int e = mb / n_mb_per_epoch, m = mb % n_mb_per_epoch;
    const int64_t start = (int64_t)m * MB;
    const int64_t end = start + MB < B ? start + MB : B;
    const int64_t len = end - start;
    const int64_t p0 = start + (len * part) / PPO_ADV_PARTS, p1 = start + (len * (part + 1)) / PPO_ADV_PARTS;
    const int32_t* idx = perm + (size_t)e * B;
    double s1 = 0.0, s2 = 0.0;
    int64_t j = p0 + threadIdx.x;
    for (; j + 7 * 256 < p1; j += 8 * 256) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = adv[idx[j + i * 256]];
#pragma unroll
        for (int i = 0; i < 8; i++) { const double x = v[i]; s1 += x; s2 += x * x; }
    }
    for (; j < p1; j += 256) { const double x = adv[idx[j]]; s1 += x; s2 += x * x; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[mb * PPO_ADV_PARTS + part].s1 = ((r1[0] + r1[1]) + r1[2]) + r1[3];
        out[mb * PPO_ADV_PARTS + part].s2 = ((r2[0] + r2[1]) + r2[2]) + r2[3];
    }
}

// Keyed bijection of [0, 2^bits) (xor / odd multiply / xorshift are each invertible mod 2^bits), cycle-walked into [0, B).
__device__ __forceinline__ uint32_t mix_bits(uint32_t x, uint32_t mask, int bits, uint4 k) {
    const int sh = bits > 1 ? bits / 2 : 1;
    x ^= k.x & mask; x = (x * 0x9E3779B1u) & mask; x ^= x >> sh;
    x ^= k.y & mask; x = (x * 0x85EBCA77u) & mask; x ^= x >> sh;
    x ^= k.z & mask; x = (x * 0xC2B2AE3Du) & mask; x ^= x >> sh;
    x ^= k.w & mask; x = (x * 0x27D4EB2Fu) & mask; x ^= x >> sh;
    return x;
}
__global__ void permutation_kernel(int32_t* perm, int64_t B, int E, int64_t seed, int64_t update_index, int64_t rank_salt) {
    // the round keys depend only on (seed, update, epoch, rank): one Philox call per workgroup, not per element
    __shared__ uint4 s_key;
    __shared__ int s_bits;
    const int e = blockIdx.y;
    if (threadIdx.x == 0) {
        int bits = 1;
        while ((1ll << bits) < B) bits++;
        s_bits = bits;
        s_key = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)update_index, (uint32_t)e, (uint32_t)rank_salt, 2u);
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B || e >= E) return;
    const int bits = s_bits;
    const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    const uint4 k = s_key;
    uint32_t x = (uint32_t)i;
    do { x = mix_bits(x, mask, bits, k); } while (x >= (uint64_t)B);
    perm[(size_t)e * B + i] = (int32_t)x;
}

// permutation_kernel and adv_stats_kernel in ONE pass over the update's E x B index space: a workgroup forms its slice of one minibatch's
// permutation (same bijection, same keys), stores it, and sums the advantages it selects -- the indices never make a round trip
// through memory between the two.  Grid (minibatch of the update, part), as adv_stats_kernel.
__global__ __launch_bounds__(256) void perm_adv_stats_kernel(const float* __restrict__ adv, int32_t* __restrict__ perm, int64_t B, int64_t MB,
                                                             int n_mb_per_epoch, int64_t seed, int64_t update_index, int64_t rank_salt,
                                                             AdvStat* out) {
    __shared__ double r1[4], r2[4];
    __shared__ uint4 s_key;
    const int mb = blockIdx.x, part = blockIdx.y;
    const int e = mb / n_mb_per_epoch, m = mb % n_mb_per_epoch;
    if (threadIdx.x == 0) s_key = philox4x32_10((uint32_t)seed, (uint32_t)((uint64_t)seed >> 32), (uint32_t)update_index, (uint32_t)e, (uint32_t)rank_salt, 2u);
    int bits = 1;
    while ((1ll << bits) < B) bits++;
    const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    __syncthreads();
    const uint4 k = s_key;
    const int64_t start = (int64_t)m * MB;
    const int64_t end = start + MB < B ? start + MB : B;
    const int64_t len = end - start;
    const int64_t p0 = start + (len * part) / PPO_ADV_PARTS, p1 = start + (len * (part + 1)) / PPO_ADV_PARTS;
    int32_t* idx = perm + (size_t)e * B;
    double s1 = 0.0, s2 = 0.0;
    int64_t j = p0 + threadIdx.x;
    for (; j + 7 * 256 < p1; j += 8 * 256) {
        uint32_t x[8];
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x[i] = (uint32_t)(j + i * 256);
            do { x[i] = mix_bits(x[i], mask, bits, k); } while (x[i] >= (uint64_t)B);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = adv[x[i]];
#pragma unroll
        for (int i = 0; i < 8; i++) idx[j + i * 256] = (int32_t)x[i];
#pragma unroll
        for (int i = 0; i < 8; i++) { const double t = v[i]; s1 += t; s2 += t * t; }
    }
    for (; j < p1; j += 256) {
        uint32_t x = (uint32_t)j;
        do { x = mix_bits(x, mask, bits, k); } while (x >= (uint64_t)B);
        idx[j] = (int32_t)x;
        const double t = adv[x]; s1 += t; s2 += t * t;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[mb * PPO_ADV_PARTS + part].s1 = ((r1[0] + r1[1]) + r1[2]) + r1[3];
        out[mb * PPO_ADV_PARTS + part].s2 = ((r2[0] + r2[1]) + r2[2]) + r2[3];
    }
}

// In-process all-reduce: out[i] = bufs[0][i] + bufs[1][i] + ... (rank order), written back to every rank's buffer.
struct PtrPack8 { void* p[8]; };
template <class Tp>
__global__ void local_allreduce_kernel(PtrPack8 pk, int n, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Tp acc = static_cast<const Tp*>(pk.p[0])[i];
    for (int r = 1; r < n; r++) acc += static_cast<const Tp*>(pk.p[r])[i];
    for (int r = 0; r < n; r++) static_cast<Tp*>(pk.p[r])[i] = acc;
}

// One-shot direct exchange all-reduce (api.hip: ExchangeComm; buffer layout and protocol above, at xchg_slot).  Call number `seq` (1-based, the
// same on every rank) uses parity seq & 1.
//   publish:  every thread stores its elements of buf into this rank's slot of EVERY peer's buffer (system scope), release; the workgroup that
//             arrives last on the parity's counter stamps this rank's flag in every peer's buffer with seq
//   gather:   wave 0 waits until every peer's flag in the OWN buffer has reached seq (lane r polls flag r, bounded: xchg_wait_all), then every thread adds
//             the peers' elements out of the own buffer in rank order
//   result:   buf = the sum, formed in the same order on every rank.
// A timeout leaves the sum incomplete and sets timeout_flag: the kernel always ends, and the host turns the flag into PPO_ERR_COMM.
struct XchgPtrs8 { void* p[8]; };
// The grid is bounded (launch_exchange_allreduce: at most XCHG_MAX_BLOCKS workgroups, each walking the buffer in strides): every workgroup
// waits for flags that a peer stamps only when ITS last workgroup has arrived, so all workgroups of a rank must be resident at once --
// a grid sized by the element count (1 800 workgroups for configs[4]'s 460 k gradient floats) can exceed what the chip holds and then
// waits for workgroups that cannot start.
// ... and bounded well BELOW the number of CUs (round 6: 64, it was 256): a waiting exchange workgroup holds a slot of its CU, and the matrix-core kernels of the generic
// path need a CU's whole register file (two waves of 250 registers per SIMD) and all but ~1 KB of its LDS -- beside a resident exchange wave they cannot start.  One
// rank per GPU never meets that; ranks REHEARSED on one GPU do: with the exchange on all 256 CUs, the rank that arrives first waits for a peer whose forward launch can
// find no CU (seen once in this round's runs as a 30 s wait that ran out in tests/test_gpu_exchange.py, after the forward kernel's registers grew from 216 to 253).  With 64
// workgroups even three waiting ranks leave a quarter of the chip to the one that still computes.  The payload of a step (37 KB; 2.4 MB at configs[4]) does not need more.
constexpr int XCHG_MAX_BLOCKS = 64;
template <class Tp>
__global__ __launch_bounds__(256) void exchange_allreduce_kernel(Tp* __restrict__ buf, size_t count, XchgPtrs8 peers, int rank, int n, size_t slot_bytes,
                                                                 unsigned long long seq, int32_t* timeout_flag) {
    __shared__ unsigned s_okm;
    const int parity = (int)(seq & 1ull);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const Tp mine = buf[i];
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (r < n && r != rank)
                __hip_atomic_store(reinterpret_cast<Tp*>(xchg_slot(peers.p[r], slot_bytes, parity, rank)) + i, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: this thread's stores have landed before the workgroup is counted in
    __syncthreads();
    if (threadIdx.x < 64) {
        xchg_publish_done(peers.p, rank, n, slot_bytes, parity, seq);
        const unsigned okm = xchg_wait_all(peers.p[rank], rank, n, slot_bytes, parity, seq, timeout_flag);
        if (threadIdx.x == 0) s_okm = okm;
    }
    __syncthreads();
    const unsigned okm = s_okm;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const Tp mine = buf[i];
        Tp v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {   // all peers' elements requested together ...
            v[r] = Tp(0);
            if (r < n && r != rank && ((okm >> r) & 1u))
                v[r] = __hip_atomic_load(reinterpret_cast<const Tp*>(xchg_slot(peers.p[rank], slot_bytes, parity, r)) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        Tp acc = Tp(0);
#pragma unroll
        for (int r = 0; r < 8; r++) if (r < n) acc += r == rank ? mine : v[r];   // ... added in rank order
        buf[i] = acc;
    }
}

// K11 partial sums: per block {sum y, sum y^2, sum d, sum d^2} with y = returns, d = returns - values (float subtraction).
__global__ __launch_bounds__(256) void explained_variance_kernel(const float* __restrict__ returns, const float* __restrict__ values,
                                                                  int64_t B, double* out) {
    __shared__ double red[4][4];
    double sy = 0, sy2 = 0, sd = 0, sd2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const double y = returns[i], d = (double)(returns[i] - values[i]);
        sy += y; sy2 += y * y; sd += d; sd2 += d * d;
    }
    sy = wave_sum_d(sy); sy2 = wave_sum_d(sy2); sd = wave_sum_d(sd); sd2 = wave_sum_d(sd2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = sy; red[w][1] = sy2; red[w][2] = sd; red[w][3] = sd2; }
    __syncthreads();
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// This rank's contribution to the job-global statistics block (ppo_internal.hpp: PPO_GSTAT_*).  One workgroup: zero the block, add the
// PPO_EV_BLOCKS explained-variance rows in a fixed order (thread j: rows j and j + 256, then a fixed tree), copy the ring into the rank's slot.
__global__ __launch_bounds__(256) void gstats_pack_kernel(const double* __restrict__ ev_sums, const EpisodeRing* __restrict__ ring, int rank, double* __restrict__ g) {
    __shared__ double red[256][4];
    static_assert(PPO_EV_BLOCKS == 512, "two explained-variance rows per thread");
    const int j = threadIdx.x;
    for (int i = j; i < PPO_GSTAT_DOUBLES; i += 256) g[i] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) red[j][q] = ev_sums[j * 4 + q] + ev_sums[(j + 256) * 4 + q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (j < o) {
#pragma unroll
            for (int q = 0; q < 4; q++) red[j][q] += red[j + o][q];
        }
        __syncthreads();
    }
    if (j < 4) g[j] = red[0][j];
    double* slot = g + PPO_GSTAT_HEAD + (size_t)rank * PPO_GSTAT_RANK;
    const int size = ring->size;
    if (j == 0) { slot[0] = (double)ring->total; slot[1] = (double)size; }
    if (j < size && j < 100) {
        slot[4 + 3 * j + 0] = (double)ring->key[j];
        slot[4 + 3 * j + 1] = (double)ring->len[j];
        slot[4 + 3 * j + 2] = (double)ring->rew[j];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
int update_blocks_per_net(int M) {
    const int tiles = (M + PPO_TILE - 1) / PPO_TILE;
    return tiles < 256 ? (tiles > 0 ? tiles : 1) : 256;  // one (actor, critic) workgroup pair per CU
}

hipError_t launch_minibatch_fwd_bwd(const UpdateArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipErrorInvalidValue;
    const int aout = a.L.act > 1 ? a.L.act : 1;
    const size_t shmem = (size_t)smem_layout(a.L.obs, aout).total * sizeof(float);
    const dim3 grid((unsigned)a.n_blocks[0], 2), block(UPD_THREADS);  // the VALU kernel uses equal shares
#define PPO_LAUNCH_UPD(DIST, OBS)                                                                                     \
    do {                                                                                                              \
        static std::atomic<unsigned long long> lds_ok{0};                                                             \
        const hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void*>(&fwd_bwd_kernel<DIST, OBS>));    \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL((fwd_bwd_kernel<DIST, OBS>), grid, block, shmem, s, a);                                    \
    } while (0)
    if (a.L.obs == 4) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) PPO_LAUNCH_UPD(PPO_DIST_CATEGORICAL, 4); else PPO_LAUNCH_UPD(PPO_DIST_MASKED, 4);
    } else if (a.L.obs == 2) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) PPO_LAUNCH_UPD(PPO_DIST_CATEGORICAL, 2); else PPO_LAUNCH_UPD(PPO_DIST_MASKED, 2);
    } else if (a.L.obs == 8) {
        if (a.hp.dist_kind == PPO_DIST_CATEGORICAL) PPO_LAUNCH_UPD(PPO_DIST_CATEGORICAL, 8); else PPO_LAUNCH_UPD(PPO_DIST_MASKED, 8);
    } else {
        return hipErrorInvalidValue;
    }
#undef PPO_LAUNCH_UPD
    return hipGetLastError();
}

hipError_t launch_reduce_grads(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                               double* sums_out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_grads_kernel, dim3((L.P + 63) / 64 + 1), dim3(64 * RED_WAVES), 0, s, slab, stat_slab, n_blocks[0], n_blocks[1], L, grads, sums_out);
    return hipGetLastError();
}

hipError_t launch_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const NetLayout& L, float max_grad_norm,
                             const AdamCoef* coef, const double* loss_sums, double global_M, LossParams hp, int world, bool do_step,
                             StepStats* stats_out, double* clipfrac_accum, double* norm2_scratch, hipStream_t s, OptGuard guard) {
    if (PPO_OPT_GUARD && !guard.error_flag) return hipErrorInvalidValue;   // the kernel reads the word without a null check
    hipLaunchKernelGGL(grad_norm_kernel, dim3(L.n_tensors), dim3(NORM_THREADS), 0, s, grads, L, norm2_scratch);
    const int blocks = do_step ? (L.P + ADAM_THREADS - 1) / ADAM_THREADS : 1;
    hipLaunchKernelGGL(clip_adamw_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq, L, max_grad_norm, norm2_scratch,
                       coef, loss_sums, global_M, hp, world, do_step ? 1 : 0, stats_out, clipfrac_accum, guard);
    return hipGetLastError();
}
hipError_t launch_weight_range(const float* params, const NetLayout& L, uint32_t* wr_dev, uint32_t* wr_host, hipStream_t s) {
    hipLaunchKernelGGL(weight_range_kernel, dim3(1), dim3(1024), 0, s, params, L, wr_dev, wr_host);
    return hipGetLastError();
}

int fused_opt_blocks(const NetLayout& L) { return (L.P + 63) / 64 + 1; }
hipError_t launch_reduce_clip_adamw(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                                    double* sums_out, float* params, float* exp_avg, float* exp_avg_sq, float max_grad_norm,
                                    const AdamCoef* coef, double global_M, LossParams hp, StepStats* stats_out, double* clipfrac_accum,
                                    double* partial, hipStream_t s, OptGuard guard) {
    if (PPO_OPT_GUARD && !guard.error_flag) return hipErrorInvalidValue;   // the kernel reads the word without a null check
    FusedOptArgs a{};
    a.guard = guard;
    a.slab = slab; a.stat_slab = stat_slab; a.nb0 = n_blocks[0]; a.nb1 = n_blocks[1]; a.L = L; a.grads = grads; a.sums_out = sums_out;
    a.params = params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.max_norm = max_grad_norm; a.coef = coef; a.global_M = global_M;
    a.p_src = params; a.m_src = exp_avg; a.v_src = exp_avg_sq;
    a.hp = hp; a.stats_out = stats_out; a.clipfrac_accum = clipfrac_accum; a.partial = partial;
    hipLaunchKernelGGL(reduce_grads_sumsq_kernel<false>, dim3(fused_opt_blocks(L)), dim3(64 * RED_WAVES), 0, s, a);
    hipLaunchKernelGGL(clip_adamw_sumsq_kernel, dim3((L.P + ADAM_THREADS - 1) / ADAM_THREADS), dim3(ADAM_THREADS), 0, s, a);
    return hipGetLastError();
}
// The same two launches for a sharded context on the direct-exchange transport: the reduction also publishes this rank's gradient (and loss
// sums) to its exchange slot and adds the peers' in rank order.  global_M = the GLOBAL minibatch size.
hipError_t launch_reduce_exchange_clip_adamw(const float* slab, const double* stat_slab, const int n_blocks[2], const NetLayout& L, float* grads,
                                             double* sums_out, float* params, float* exp_avg, float* exp_avg_sq, float max_grad_norm, const AdamCoef* coef,
                                             double global_M, const LossParams& hp, StepStats* stats_out, double* clipfrac_accum, double* partial,
                                             void* const* peers, int rank, int n_ranks, size_t slot_bytes, uint64_t seq, int32_t* timeout_flag,
                                             hipStream_t s, OptGuard guard) {
    if (n_ranks < 1 || n_ranks > 8 || (size_t)(L.P + 8) * sizeof(float) > slot_bytes) return hipErrorInvalidValue;
    if (PPO_OPT_GUARD && !guard.error_flag) return hipErrorInvalidValue;   // the kernel reads the word without a null check
    FusedOptArgs a{};
    a.guard = guard;
    a.slab = slab; a.stat_slab = stat_slab; a.nb0 = n_blocks[0]; a.nb1 = n_blocks[1]; a.L = L; a.grads = grads; a.sums_out = sums_out;
    a.params = params; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.max_norm = max_grad_norm; a.coef = coef; a.global_M = global_M;
    a.p_src = params; a.m_src = exp_avg; a.v_src = exp_avg_sq;
    a.hp = hp; a.stats_out = stats_out; a.clipfrac_accum = clipfrac_accum; a.partial = partial;
    for (int r = 0; r < 8; r++) a.peers[r] = r < n_ranks ? peers[r] : nullptr;
    a.rank = rank; a.n_ranks = n_ranks; a.slot_bytes = slot_bytes; a.seq = seq; a.timeout_flag = timeout_flag;
    hipLaunchKernelGGL(reduce_grads_sumsq_kernel<true>, dim3(fused_opt_blocks(L)), dim3(64 * RED_WAVES), 0, s, a);
    hipLaunchKernelGGL(clip_adamw_sumsq_kernel, dim3((L.P + ADAM_THREADS - 1) / ADAM_THREADS), dim3(ADAM_THREADS), 0, s, a);
    return hipGetLastError();
}
// One thread per minibatch slot: the PPO_ADV_PARTS partial sums in order, then mean and Bessel std in binary64 as the update kernels used to do in
// every launch (32 dependent scalar loads behind the weight loads: ~2 us of the actor's prologue, 40 times per update).
__global__ void adv_norm_kernel(const AdvStat* __restrict__ stats, int n, int per_epoch, int64_t B, int64_t MB, int64_t explicit_M, int world,
                                float4* __restrict__ out, double* zero2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && zero2) { zero2[0] = 0.0; zero2[1] = 0.0; }   // the update's clip-fraction accumulator (m_clipfracs reset, PPO_Discrete.cpp:564): a memset launch less
    if (k >= n) return;
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < PPO_ADV_PARTS; i++) { t1 += stats[(size_t)k * PPO_ADV_PARTS + i].s1; t2 += stats[(size_t)k * PPO_ADV_PARTS + i].s2; }
    const int64_t start = (int64_t)(k % per_epoch) * MB;
    const int64_t M = explicit_M > 0 ? explicit_M : (MB < B - start ? MB : B - start);
    const double global_M = (double)M * world;
    const double mean = t1 / global_M;
    const double var = (t2 - t1 * mean) / (global_M - 1.0);
    const float mean_f = (float)mean;
    const float std_f = (float)sqrt(var < 0.0 ? 0.0 : var);
    out[k] = make_float4(mean_f, 1.0f / (std_f + 1e-8f), std_f, 0.0f);
}
hipError_t launch_adv_norm(const AdvStat* stats, int n, int per_epoch, int64_t B, int64_t MB, int64_t explicit_M, int world, float4* out, hipStream_t s, double* zero2) {
    if (n <= 0) return zero2 ? hipMemsetAsync(zero2, 0, 2 * sizeof(double), s) : hipSuccess;
    hipLaunchKernelGGL(adv_norm_kernel, dim3((n + 63) / 64), dim3(64), 0, s, stats, n, per_epoch > 0 ? per_epoch : 1, B, MB, explicit_M, world, out, zero2);
    return hipGetLastError();
}

hipError_t launch_adv_stats(const float* advantages, const int32_t* perm, int64_t B, int64_t MB, int n_mb_total, AdvStat* out,
                            hipStream_t s) {
    const int per_epoch = (int)((B + MB - 1) / MB);
    hipLaunchKernelGGL(adv_stats_kernel, dim3(n_mb_total, PPO_ADV_PARTS), dim3(256), 0, s, advantages, perm, B, MB, per_epoch, out);
    return hipGetLastError();
}

hipError_t launch_permutations(int32_t* perm, int64_t B, int E, int64_t seed, int64_t update_index, int64_t rank_salt, hipStream_t s) {
    hipLaunchKernelGGL(permutation_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)E), dim3(256), 0, s, perm, B, E, seed, update_index, rank_salt);
    return hipGetLastError();
}

hipError_t launch_permutations_adv_stats(const float* advantages, int32_t* perm, int64_t B, int E, int64_t MB, int64_t seed, int64_t update_index,
                                         int64_t rank_salt, AdvStat* out, hipStream_t s) {
    const int per_epoch = (int)((B + MB - 1) / MB);
    hipLaunchKernelGGL(perm_adv_stats_kernel, dim3(E * per_epoch, PPO_ADV_PARTS), dim3(256), 0, s, advantages, perm, B, MB, per_epoch, seed, update_index, rank_salt, out);
    return hipGetLastError();
}

hipError_t launch_explained_variance(const float* returns, const float* values, int64_t B, double* sums4, hipStream_t s) {
    hipLaunchKernelGGL(explained_variance_kernel, dim3(PPO_EV_BLOCKS), dim3(256), 0, s, returns, values, B, sums4);
    return hipGetLastError();
}

hipError_t launch_gstats_pack(const double* ev_sums, const EpisodeRing* ring, int rank, double* gstats, hipStream_t s) {
    if (rank < 0 || rank >= 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gstats_pack_kernel, dim3(1), dim3(256), 0, s, ev_sums, ring, rank, gstats);
    return hipGetLastError();
}

struct PtrPack { void* p[8]; };
hipError_t launch_local_allreduce(const PtrPack& pk, int n, size_t count, bool f64, hipStream_t s) {
    PtrPack8 q;
    for (int i = 0; i < 8; i++) q.p[i] = pk.p[i];
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (f64) hipLaunchKernelGGL(local_allreduce_kernel<double>, grid, block, 0, s, q, n, count);
    else hipLaunchKernelGGL(local_allreduce_kernel<float>, grid, block, 0, s, q, n, count);
    return hipGetLastError();
}

struct XchgPtrs { void* p[8]; };
hipError_t launch_exchange_allreduce(void* buf, size_t count, bool f64, const XchgPtrs& peers, int rank, int n, size_t slot_bytes, uint64_t seq,
                                     int32_t* timeout_flag, hipStream_t s) {
    XchgPtrs8 q;
    for (int i = 0; i < 8; i++) q.p[i] = peers.p[i];
    const size_t blocks = (count + 255) / 256;
    const dim3 grid((unsigned)(blocks < 1 ? 1 : (blocks > (size_t)XCHG_MAX_BLOCKS ? (size_t)XCHG_MAX_BLOCKS : blocks))), block(256);
    if (f64) hipLaunchKernelGGL(exchange_allreduce_kernel<double>, grid, block, 0, s, static_cast<double*>(buf), count, q, rank, n, slot_bytes, (unsigned long long)seq, timeout_flag);
    else hipLaunchKernelGGL(exchange_allreduce_kernel<float>, grid, block, 0, s, static_cast<float*>(buf), count, q, rank, n, slot_bytes, (unsigned long long)seq, timeout_flag);
    return hipGetLastError();
}
