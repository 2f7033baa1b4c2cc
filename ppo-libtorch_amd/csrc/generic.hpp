// ppo-libtorch_amd/csrc/generic.hpp -- networks other than the reference's 2 x 64 (BASELINE configs[4]: obs 376, 4 x 256, heads
// [3,3,3,2]; SURVEY 8(a) a6 "hidden width / depth configurable", cf. the commented 256-wide third layer of Agent.cpp:27,32,45-46).
//
// Same contract and the same C-ABI as the specialised path; different machine mapping: at these widths every layer is a real GEMM
// (M = rows of the step or of the minibatch, N = K = hidden), so layers are the hand-written matrix-core products of kernels_gemm.hip
// (bias, tanh, tanh' and the bias gradient fused into them) glued by small kernels for what is not a GEMM -- the categorical heads,
// the PPO loss and its gradient, the gather, clip + AdamW over an arbitrary tensor list.
// Parameter order is the reference's generalised: critic layers (W [out][in] row-major, b) then actor layers (Agent.cpp:65-66).
#pragma once
#include <cstdint>
#include <string>

#include "ppo_internal.hpp"

constexpr int GEN_MAX_LAYERS = 8;   // linear layers per net (n_hidden + 1)

struct GenLayout {
    int obs, act, n_heads, hidden, n_hidden;
    int head_dims[PPO_MAX_HEADS];
    int n_layers;                          // n_hidden + 1
    int in_dim[GEN_MAX_LAYERS];            // per layer (same for both nets)
    int out_dim[2][GEN_MAX_LAYERS];        // [net][layer]: hidden ... hidden, then 1 (critic) / act (actor)
    int w_off[2][GEN_MAX_LAYERS], b_off[2][GEN_MAX_LAYERS];
    int net_off[2], net_size[2];
    int P;
    int n_tensors;
    int tensor_off[4 * GEN_MAX_LAYERS + 1];
};

inline GenLayout make_gen_layout(int obs, int hidden, int n_hidden, int n_heads, const int* head_dims) {
    GenLayout L{};
    L.obs = obs; L.hidden = hidden; L.n_hidden = n_hidden; L.n_heads = n_heads; L.n_layers = n_hidden + 1;
    for (int h = 0; h < n_heads; h++) { L.head_dims[h] = head_dims[h]; L.act += head_dims[h]; }
    int o = 0, t = 0;
    for (int net = 0; net < 2; net++) {
        L.net_off[net] = o;
        for (int l = 0; l < L.n_layers; l++) {
            L.in_dim[l] = l == 0 ? obs : hidden;
            L.out_dim[net][l] = l == n_hidden ? (net == 0 ? 1 : L.act) : hidden;
            L.tensor_off[t++] = o; L.w_off[net][l] = o; o += L.out_dim[net][l] * L.in_dim[l];
            L.tensor_off[t++] = o; L.b_off[net][l] = o; o += L.out_dim[net][l];
        }
        L.net_size[net] = o - L.net_off[net];
    }
    L.tensor_off[t] = o;
    L.n_tensors = t;
    L.P = o;
    return L;
}

// the rollout buffers a step's index list points into (PPO_Discrete.cpp:557-562 flattened, :576-582 indexed)
struct GenRowSrc { const int32_t* actions; const uint8_t* masks; const float* logprobs; const float* adv; const float* ret; const float* values; };

// Device workspace and library handle of one generic context (owned by ppo_ctx; see api.hip).
struct GenericCtx {
    GenLayout L{};
    int gemm_prec = 0;             // PPO_MM_F32X3 (default) or PPO_MM_BF16 (PPO_GENERIC_PREC=bf16)
    // weights of every layer pre-split into bf16 planes [t][n_pad][k_pad] (n_pad, k_pad: multiples of 128, zero padded) for the B operand of
    // the forward and d(input) products; rewritten (one launch) when the parameters have changed since the last use
    uint16_t* wplanes = nullptr;
    uint16_t* wfrags = nullptr;    // bf16 storage: the same values in MFMA-fragment order (kernels_gemm.hip: weight_planes_kernel), same offsets
    int64_t wp_off[2][GEN_MAX_LAYERS] = {};
    int wp_npad[2][GEN_MAX_LAYERS] = {}, wp_kpad[GEN_MAX_LAYERS] = {};
    mutable bool planes_dirty = true;
    int64_t rows_max = 0;          // rows the workspaces are sized for: max(minibatch, T*N + N for the critic batch is chunked to it)
    float* acts[2][GEN_MAX_LAYERS] = {};   // [net][l]: post-tanh activations of hidden layer l, [rows_max, hidden]
    float* dz[2] = {};             // ping-pong d(pre-activation), [rows_max, hidden]
    float* xin = nullptr;          // gathered observations of the minibatch, [rows_max, obs]
    float* logits = nullptr;       // [rows_max, act]
    float* dlogits = nullptr;      // [rows_max, act]
    float* val = nullptr;          // [rows_max]
    float* dval = nullptr;         // [rows_max]
    float* row_f[5] = {};          // gathered per-row scalars: old log-prob, advantage, return, old value; [4] = {adv mean, 1/(std+eps)} then a ones vector
    int32_t* row_act = nullptr;    // gathered actions [rows_max, n_heads]
    uint8_t* row_mask = nullptr;   // gathered masks [rows_max, act]
    double* loss_part = nullptr;   // [GEN_LOSS_BLOCKS, 8] partial loss sums
    float* wslab = nullptr;        // [GEN_SPLIT + 1][max over layers of out * in + out]: row-chunk partials of one layer's dW | db
    float* wslab1 = nullptr;       // bf16 storage: the actor's own slab set (its backward pass runs beside the critic's); with bf16 storage a slab set holds
                                   // one block of (GEN_SPLIT_MFMA + 1) slabs PER LAYER (wslab_layer_stride apart)
    int64_t wslab_layer_stride = 0;
    int64_t wslab_stride = 0;
    float* db_part = nullptr;      // [GEN_DB_CHUNKS][max out]: row-chunk partials of one layer's bias gradient
    // ---- bf16 storage (ppo_config.compute_dtype = PPO_DTYPE_BF16): layer inputs, hidden activations and back-propagated gradients live in HBM
    //      as bf16, every buffer [rows_max + 128][pitch] with pitch a multiple of 128 and zeros in the padding (launch_matmul_bf16 has no guards)
    bool bf16 = false;
    int ld_in0 = 0, ld_h = 0;      // pitches of the network input (pad128(obs)) and of a hidden vector (pad128(hidden))
    uint16_t* xin_bf = nullptr;    // [.][ld_in0] layer-0 input: the gathered (or converted) observations
    // A minibatch step of the fused kernels reads its rows IN PLACE: the rollout's observations are rounded to bf16 once per update (obs_bf, [T N + 128][ld_in0],
    // zero padded) and the layer-0 loads of both passes, and the loss kernel's per-row scalars, go through the step's index list (rows_idx) -- no gathered copy
    // is written and read back (207 MB of the 1.77 GB a 65 536-row step moved, and the 50 us kernel that made it).  rows_idx == nullptr: the dense g.xin_bf / row_*.
    uint16_t* obs_bf = nullptr;
    mutable const int32_t* rows_idx = nullptr;
    mutable GenRowSrc rows_src{};
    float* row_rec = nullptr;                       // [T N][8] one 32-byte record of per-row scalars per sample (gen_pack_rows, once per update; gen_rows_packable shapes)
    mutable const float4* rows_rec = nullptr;       // = row_rec for a step whose records are valid, else null: the loss kernel then indexes rows_src's arrays
    uint16_t* acts_bf[2][GEN_MAX_LAYERS] = {};   // [net][l] [.][ld_h] kept activations of a minibatch step
    uint16_t* tmp_bf[2] = {};      // [.][ld_h] ping-pong activations of a forward pass that keeps nothing (rollout step, critic batch)
    uint16_t* dz_bf[2][2] = {};    // [net][.] [.][ld_h] ping-pong d(pre-activation); one pair per net: the two backward passes run on two streams
    uint16_t* dout_bf[2] = {};     // [net] [.][128] d(loss)/d(value), d(loss)/d(logits); zero beyond the head's width
    float* cs_part[2] = {};        // [net] [layer][rows_max / 128 + 1][ld_h] per-m-tile column sums of a d(pre-activation) (the next bias gradient); one block per layer,
                                   // so that a net's slab sums can run as ONE launch behind its backward pass
    int64_t cs_layer_stride = 0;   // floats between two layers' blocks
    double* sq_part = nullptr;     // [2 n_layers][sq_cap][2] sums of squares of the gradient per slab-sum workgroup (weights, bias): fused backward -> gen_opt_fused
    int sq_cap = 0;                // slab-sum workgroups of the largest layer
    mutable bool sq_valid[2] = {}; // [net]: the net's last backward pass left its part of sq_part
    float* head_db_part = nullptr; // [GEN_LOSS_BLOCKS][act + 1] block sums of d(loss)/d(logits) | d(value) (the head layers' bias gradients)
    mutable int head_fused = 0;    // > 0: this step's forward launch also ran the loss and the head layers' backward (gen_fused_forward_loss) and left that many
                                   // per-workgroup partials (dW_head slabs, column sums of dZ_top in cs_part's top block, dZ_top in dz_bf[net][(n_layers - 1) & 1])
    int64_t* act64 = nullptr;      // [N, n_heads] actions of the current rollout step (int64, the stand-alone API's type)
    float* step_lp = nullptr;      // [N] log-prob / entropy of the current rollout step
    float* step_en = nullptr;
};
constexpr int GEN_LOSS_BLOCKS = 256;
constexpr int GEN_DB_CHUNKS = 256, GEN_NORM_PARTS = 16;
constexpr int GEN_SPLIT_MFMA = 128;   // row ranges of a weight-gradient product on the matrix cores: (out / 128)(in / 128) tiles x ranges >= 2 workgroups per CU

// kernels_gemm.hip: c[z][M, N] (z < splits, slabs c_zstride floats apart) = epilogue(sum over the z-th range of k of A(m, k) B(n, k));
// see ppo_matmul in ppo_hip.h.  splits > 1 cuts the contraction into equal ranges (multiples of 64) and requires PPO_MM_EPI_NONE.
// colsum (trans_a only, may be null): colsum[z * colsum_zstride + m] = sum over the z-th range of k of A(m, k).
hipError_t launch_matmul(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* c,
                         int64_t ldc, int epilogue, const float* aux, int64_t ld_aux, int precision, int splits, int64_t c_zstride, float* colsum,
                         int64_t colsum_zstride, const uint16_t* bplanes, int64_t bp_plane, int64_t bp_ld, hipStream_t s);
struct GenericCtx;
hipError_t gen_weight_planes(const GenericCtx& g, const float* params, hipStream_t s);
// bf16-storage variant of launch_matmul and the f32 -> padded bf16 converter (kernels_gemm.hip)
hipError_t launch_matmul_bf16(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                              void* c, int64_t ldc, bool c_bf16, int epilogue, const void* aux, int64_t ld_aux, int splits, int64_t c_zstride,
                              float* colsum, int64_t colsum_stride, hipStream_t s);
hipError_t launch_to_bf16_pad(const float* src, int64_t rows, int K, uint16_t* dst, int ld, hipStream_t s, const int32_t* idx = nullptr);   // idx: those rows only, in place

// kernels_generic_fused.hip: bf16-storage networks whose layer inputs fit LDS -- a net's whole forward pass, or the whole T-step rollout of
// the synthetic env, in one launch
// One workgroup of the batch kernel streams ALL of a net's weights from L2 for each of its 64-row tiles.  In its first form (one tile per
// workgroup, input staged by a scalar loop, lane = output column) it lost to the five tiled products at 65 536 rows (142 us per net against
// 112); persistent, with the next tile's input prefetched into registers, the product transposed so that the epilogue stores 8-byte row
// pieces, and a layer's weight fragments requested across the epilogue of the layer above, it wins there too (configs[4] minibatch step
// 0.871 -> 0.822 ms, A/B in one call), so there is no row limit any more.
constexpr int64_t GEN_FUSED_MAX_ROWS = 1 << 20;
bool gen_fused_ok(const GenericCtx& g);
bool gen_fused_forward_ok(const GenericCtx& g);
// idx (bf16 input only, may be null): row r of the pass is row idx[r] of x_bf
hipError_t gen_fused_forward(const GenericCtx& g, const float* params, int net, const float* x_f32, const uint16_t* x_bf, int64_t ld_x, int64_t rows, bool keep,
                             float* out, hipStream_t s, const int32_t* idx = nullptr);
// both nets' passes over the same rows (activations kept) in one launch
hipError_t gen_fused_forward_both(const GenericCtx& g, const float* params, const uint16_t* x_bf, int64_t ld_x, int64_t rows, float* logits, float* val,
                                  hipStream_t s, const int32_t* idx = nullptr);
// both nets' passes + heads + PPO loss + the head layers' backward in one launch (rows read in place through idx, records g.rows_rec); sets g.head_fused
bool gen_fused_loss_ok(const GenericCtx& g, int64_t rows);
hipError_t gen_fused_forward_loss(const GenericCtx& g, const float* params, const uint16_t* x_bf, int64_t ld_x, int64_t rows, const LossParams& hp, double inv_global_M,
                                  double global_M, const AdvStat* adv_stat, const int32_t* idx, hipStream_t s);
hipError_t gen_fused_rollout(const GenericCtx& g, const float* params, int dist_kind, int N, int T, int max_episode_steps, int64_t seed, int64_t env_offset,
                             int64_t step_base, int32_t* ep_len, float* ep_rew, float* obs, uint8_t* masks, int32_t* actions, float* logprobs, float* rewards,
                             float* dones, int32_t* fin_len, float* fin_rew, float* next_obs, int32_t* next_done, uint8_t* cur_mask, const int64_t* forced,
                             hipStream_t s);

// kernels_generic_bwd.hip: a layer's weight gradient and the gradient handed to the layer below in ONE launch (bf16 storage, widths padded to 128 / 256)
bool gen_fused_backward_ok(const GenericCtx& g);
int gen_bwd_col_blocks(int ld_in, bool has_below);
int gen_bwd_ranges(int64_t rows, int col_blocks, bool half_chip, int* tiles_per_range);
// one net's operands of a layer's launch: d = dZ_l [., ldd], h = the layer's input [., ldh] (idx != nullptr, layer 0 only: row r is row idx[r] of h), w = the
// layer's bf16 weight plane (null: layer 0, nothing below), dz_out = dZ_{l-1}, slab = S partial weight gradients [n_real][k_real], colsum = per-range column sums
struct GenBwdLayer {
    const uint16_t* d; int64_t ldd; const uint16_t* h; int64_t ldh; const int32_t* idx; const uint16_t* w; int64_t ldw; uint16_t* dz_out; int64_t ld_out;
    float* slab; int64_t slab_stride; float* colsum; int64_t ld_cs; int n_real, k_real, S, tiles_per_range;
};
// `other` != nullptr: the same layer of the other net in the same launch
hipError_t gen_fused_backward_layer(int n_pad, const GenBwdLayer& one, const GenBwdLayer* other, int64_t rows, int col_blocks, const uint16_t* zeros, hipStream_t s);

// kernels_generic.hip
struct ppo_ctx;
// out[rows, out_dim(last)] = net(x[rows, obs]); acts != nullptr keeps every hidden layer's activations (for the backward pass).
// bf16 storage: x == nullptr means the input already sits in g.xin_bf (gen_gather put it there); acts != nullptr keeps g.acts_bf[net]
hipError_t gen_forward(const GenericCtx& g, const float* params, int net, const float* x, int64_t rows, float* const* acts, float* scratch0,
                       float* scratch1, float* out, hipStream_t s);
hipError_t gen_heads(const GenLayout& L, int dist_kind, const float* logits, const uint8_t* mask, const int64_t* forced, int64_t n, int64_t seed,
                     int64_t row_offset, int64_t step_index, int64_t* action, float* logprob, float* entropy, hipStream_t s);
hipError_t gen_gather(const GenLayout& L, const float* obs, const int32_t* actions, const uint8_t* masks, const float* logprobs, const float* adv,
                      const float* ret, const float* values, const int32_t* idx, int64_t M, GenericCtx& g, hipStream_t s);
hipError_t gen_loss(const GenLayout& L, const LossParams& hp, const GenericCtx& g, int64_t M, double inv_global_M, double global_M,
                    const AdvStat* adv_stat, hipStream_t s);
// the per-row scalars of all T N samples as one 32-byte record each (the loss kernel of an in-place step reads one record per row)
bool gen_rows_packable(const GenLayout& L);
hipError_t gen_pack_rows(const GenLayout& L, const GenRowSrc& src, int64_t B, float* rec, hipStream_t s, const int32_t* idx = nullptr);   // idx: those B rows only
// beside_other_net: the other net's backward pass runs at the same time on another stream (the fused launches then size themselves for half the chip)
hipError_t gen_backward(const GenericCtx& g, const float* params, int net, const float* x, int64_t rows, const float* dout, float* grads,
                        hipStream_t s, bool beside_other_net = false);
// both nets' fused backward passes in the same launches on one stream (needs gen_fused_backward_ok)
hipError_t gen_backward_both(const GenericCtx& g, const float* params, int64_t rows, float* grads, hipStream_t s);
// the optimizer step behind gen_backward_both on a single rank, one launch: loss sums, gradient norm (from the slab sums' partials), clip + AdamW, bf16 weight planes
hipError_t gen_opt_fused(const GenericCtx& g, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float max_grad_norm, const AdamCoef* coef,
                         double* sums_out, double global_M, LossParams hp, bool do_step, StepStats* stats_out, double* clipfrac_accum, const int32_t* error_flag,
                         hipStream_t s);
hipError_t gen_loss_sums(const GenericCtx& g, double* sums_out, float* grads_tail, hipStream_t s);
hipError_t gen_fill(float* p, int64_t n, float v, hipStream_t s);
hipError_t gen_clip_adamw(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const GenLayout& L, float max_grad_norm, const AdamCoef* coef,
                          const double* loss_sums, double global_M, LossParams hp, int world, bool do_step, StepStats* stats_out,
                          double* clipfrac_accum, double* norm2_scratch, hipStream_t s);
hipError_t gen_synthetic_step(const GenLayout& L, int N, int64_t seed, int64_t env_offset, int64_t step_index, int max_episode_steps, int32_t* ep_len,
                              float* ep_rew, float* obs_out, uint8_t* mask_out, float* reward, int32_t* done, int32_t* fin_len, float* fin_rew,
                              hipStream_t s);
hipError_t gen_store_step(const GenLayout& L, int N, const float* obs, const uint8_t* mask, const int64_t* act64, const float* lp, const int32_t* done_prev,
                          float* obs_t, uint8_t* mask_t, int32_t* act_t, float* lp_t, float* dones_t, hipStream_t s);
