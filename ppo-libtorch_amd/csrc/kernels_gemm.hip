// ppo-libtorch_amd/csrc/kernels_gemm.hip -- the Linear layers of networks wider than 2 x 64 (BASELINE configs[4]: obs 376, 4 x 256,
// heads [3,3,3,2]) on the CDNA4 matrix cores: Agent.cpp:25-59's `torch::nn::Linear` forward, and the two products of its backward,
// as ONE kernel template
//
//      c[m][n] = epilogue( sum_k A(m, k) B(n, k) ),     A(m, k) = TA ? a[k * lda + m] : a[m * lda + k],   B likewise
//
//   forward      h = tanh(x W^T + b)        A = x [rows, K]            B = W [N, K]              epilogue bias (+ tanh)
//   d(input)     dz' = (dz W) (1 - h'^2)    A = dz [rows, N]           B = W read as [k = n][K]  epilogue tanh'
//   d(weight)    dW = dz^T x                A = dz read as [k = row][N]  B = x read as [k = row][K]   rows cut into ranges -> slabs
//
// Activations stay fp32 in HBM (the weights also exist as pre-split bf16 planes, see PlaneStage).  A workgroup (4 waves, 128 x 128 output
// tile, k step 32) loads its operand tiles as fp32, cuts every
// value by truncation into three bf16 terms (x = t1 + t2 + t3 exactly; see kernels_update_mfma.hip) while staging them into LDS, and
// issues each fp32 product as the six bf16 products a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: fp32 accuracy at (1/6 of) the bf16 matrix rate, which on gfx950 is ~2.7x the rate of the fp32 MFMA a library sgemm
// uses.  PREC 1 keeps one (round-to-nearest) bf16 term per operand: the "bf16 with MFMA GEMMs" arithmetic BASELINE configs[4] names.
//
// LDS tiles: a k-contiguous operand is stored [row][32 k] (80-byte rows: a lane's 16-byte fragment read is conflict-free); an operand
// whose contraction index is the slow one in memory is stored as it comes, [32 k][128 cols], and its fragments are fetched with the
// transposing LDS read ds_read_b64_tr_b16 -- no transpose in registers, coalesced global loads either way.
// Two workgroups per CU: while one stages (vector ALU: splits, LDS writes) the other's waves can keep the matrix pipe busy; the operands of
// the next TWO chunks are in flight in registers during the current chunk's MFMAs.  Measured on the 65 536 x 256 x 1024 product
// (rocprofv3 PMC): matrix pipe busy 60 % of the CU's cycles, no LDS bank conflicts, ~200 vector instructions per 48 MFMAs in the unguarded
// loop; issue priority for one of the two workgroups, or starting it half a period late, changed nothing and is not in the code.
#include <cstdlib>
#include <type_traits>

#include "ppo_internal.hpp"
#include "generic.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte global load from a 4-byte aligned address

constexpr int BK = 32;   // contraction elements per chunk (two MFMA k steps)
constexpr int KS = 40;   // bf16 per row of a k-contiguous tile: 32 + 8 pad (80 B; 16 rows x 16 B land on 64 distinct banks)
__host__ __device__ constexpr int tr_stride(int bx) { return bx == 128 ? 160 : 32; }   // bf16 per k row of an as-it-comes tile (= 16 dwords mod 64)
__host__ __device__ constexpr int tile_elems(int bx, bool trans) { return trans ? BK * tr_stride(bx) : bx * KS; }

__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

struct GemmArgs {
    const float* a; int64_t lda;
    const float* b; int64_t ldb;
    float* c; int64_t ldc;
    int M, N;
    int64_t K;            // contraction length
    int64_t k_chunk;      // contraction range of one slab (a multiple of 2 BK); slab z is written at c + z * c_zstride
    int64_t c_zstride;
    int m_tiles, n_tiles, splits;   // tile grid; the launch is 1-D and XCD-aware (see the top of gemm_kernel)
    int epi;              // PPO_MM_EPI_*
    const float* aux;     // bias [N] or h [M, ld_aux]
    int64_t ld_aux;
    const uint16_t* bplanes;   // BP kernels: B pre-split into T bf16 planes [t][rows_pad][bp_ld] (gen_weight_planes), zero padded to tile multiples
    int64_t bp_plane, bp_ld;   // elements per plane, elements per plane row
    float* colsum;        // may be null.  TA (f32 A): colsum[z * colsum_zstride + m] = sum over the z-th k range of A(m, k) (the bias gradient beside dW).
                          // !TA with PPO_MM_EPI_DTANH: colsum[tm * colsum_zstride + n] = sum over the rows of m tile tm of the f32 result (the NEXT layer
                          // down's bias gradient, formed where d(pre-activation) is produced, before it is rounded for storage)
    int64_t colsum_zstride;
};
// ABF kernels: A is bf16 in memory (`a` really points at uint16_t, lda in elements), rows padded to the tile and the contraction padded with
// zeros to a multiple of 64 (+ slack for the prefetch): staged by PlaneStage, no guards, no vector work.  CBF kernels: the result is stored as
// bf16 (`c` really points at uint16_t, ldc in elements; round to nearest even) -- a layer then moves 2 bytes per activation element.

// Four consecutive floats from global memory, branch-free (a load inside a branch makes the compiler wait for every outstanding load at the
// join, which serialises the eight loads of a chunk into eight memory round trips): elements that do not exist are read from `safe` (any
// valid address of the operand) and zeroed by the consumer (Stage::store), which first pins the raw values with an empty asm -- that keeps
// the compiler from (a) proving the value unused and predicating the load after all, and (b) hoisting the zeroing up behind the load,
// where it would wait for the data a whole chunk early.
//   VEC:  the four elements exist or not together (host-checked: the contiguous extent is a multiple of 4) -> one 16-byte load
//   !VEC: four 4-byte loads, element e exists iff e < nvalid
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* p, const float* safe, int nvalid) {
    if constexpr (VEC) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(nvalid > 0 ? p : safe);
        const f32x4 r = { v[0], v[1], v[2], v[3] };
        return r;
    }
    const f32x4 r = { *(nvalid > 0 ? p : safe), *(nvalid > 1 ? p + 1 : safe), *(nvalid > 2 ? p + 2 : safe), *(nvalid > 3 ? p + 3 : safe) };
    return r;
}
// the pin takes the quad as ONE 128-bit register operand, so it stays in the registers the load wrote (four scalar operands made the
// allocator copy elements out right behind the load, which waits for it)
__device__ __forceinline__ float4 pin_and_zero(f32x4 v, int nvalid) {
    asm volatile("" : "+v"(v));
    return make_float4(nvalid > 0 ? v[0] : 0.0f, nvalid > 1 ? v[1] : 0.0f, nvalid > 2 ? v[2] : 0.0f, nvalid > 3 ? v[3] : 0.0f);
}

// (x0, x1) -> packed bf16 pairs of the three truncation terms (low half = x0's)
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t u0 = f2u(x0), u1 = f2u(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - u2f(u0 & 0xffff0000u), r1 = x1 - u2f(u1 & 0xffff0000u);
    const uint32_t v0 = f2u(r0), v1 = f2u(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - u2f(v0 & 0xffff0000u), q1 = r1 - u2f(v1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(f2u(q1), f2u(q0), 0x07060302u);
}
__device__ __forceinline__ uint32_t pack_rne(float x0, float x1) {
    const bf16x2 v = { (__bf16)x0, (__bf16)x1 };
    return __builtin_bit_cast(uint32_t, v);
}
// four consecutive elements -> 8 bytes in each of the T term planes at bf16 index `off` (a multiple of 4)
template <int T>
__device__ __forceinline__ void split_store(uint16_t* plane0, int plane_elems, int off, const float4 v) {
    if constexpr (T == 3) {
        uint32_t a1, a2, a3, b1, b2, b3;
        split3(v.x, v.y, a1, a2, a3);
        split3(v.z, v.w, b1, b2, b3);
        *reinterpret_cast<uint2*>(plane0 + off) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(plane0 + plane_elems + off) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(plane0 + 2 * plane_elems + off) = make_uint2(a3, b3);
    } else {
        *reinterpret_cast<uint2*>(plane0 + off) = make_uint2(pack_rne(v.x, v.y), pack_rne(v.z, v.w));
    }
}

__device__ __forceinline__ uint2 lds_read_tr16(const uint16_t* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}
// MFMA operand fragment of the 32 tile rows (or columns) xb .. xb + 31 for k step ks: lane (i = lane & 31, kg = lane >> 5) gets the
// eight contraction elements 16 ks + 8 kg .. + 7 of row i
template <bool TRANS, int BX>
__device__ __forceinline__ u32x4 frag(const uint16_t* plane, int xb, int ks, int lane) {
    if constexpr (!TRANS) {
        return *reinterpret_cast<const u32x4*>(plane + (xb + (lane & 31)) * KS + 16 * ks + 8 * (lane >> 5));
    } else {
        // ds_read_b64_tr_b16: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p + 3, and receives column (lane % 16) of the four rows
        constexpr int STR = tr_stride(BX);
        const uint16_t* q = plane + (16 * ks + 8 * (lane >> 5) + ((lane & 15) >> 2)) * STR + xb + 16 * ((lane & 31) >> 4) + 4 * (lane & 3);
        const uint2 lo = lds_read_tr16(q), hi = lds_read_tr16(q + 4 * STR);
        const u32x4 r = { lo.x, lo.y, hi.x, hi.y };
        return r;
    }
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// Per-thread share of one operand tile: BX rows (or columns) x 32 contraction elements as float4s.
//   k-contiguous (TRANS = false): thread (row = stage_row(tid) + 32 p, k quad = tid % 8), p < BX / 32.  stage_row spreads the four rows
//       a 32-lane half writes at once over LDS rows r, r + 4, r + 8, r + 12: with 80-byte rows their 64-byte pieces then fall on disjoint
//       banks (rows r and r + 1 would overlap: measured 33 % of the LDS cycles were bank conflicts with the plain tid / 8 mapping)
//   as-it-comes  (TRANS = true):  thread (column quad = tid % (BX / 4), k rows 4 (tid / (BX / 4)) .. + 3)
// The thread's global addresses are formed once (rows / columns that do not exist point at the operand's first element and never move) and
// advance by one chunk per load; every load is unconditional (see load4).
__device__ __forceinline__ int stage_row(int tid) {
    const int l = tid & 63, w = tid >> 6;
    return 4 * ((l >> 3) & 3) + 2 * (l >> 5) + (w & 1) + 16 * (w >> 1);
}
template <int NV>
struct Loaded {
    f32x4 v[NV];
    int kvalid;   // !TRANS: leading contraction elements of every quad that exist;  TRANS: leading k rows (of the thread's KR) that exist
};
// NT: threads of the workgroup (256, or 512 for the two-waves-per-SIMD double-buffered variant)
template <int BX, bool TRANS, bool VEC, int NT = 256>
struct Stage {
    static constexpr int RP = NT / 8;                                   // plain: tile rows covered per pass of the workgroup
    static constexpr int KR = BX == 128 ? 32 * (BX / 4) / NT : 4;       // as-it-comes: k rows per thread
    static constexpr int NV = TRANS ? KR : (BX / RP > 0 ? BX / RP : 1);
    const float* ptr[NV];     // next chunk's address of quad p (its first element)
    int xvalid[NV];           // !TRANS: 4 if the row exists else 0;  TRANS: valid columns of the quad (0 .. 4), same for every p
    int64_t k;                // contraction index of the thread's first element in the next chunk
    int64_t step;             // pointer advance per chunk
    const float* safe;
    bool active;
    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int x0, int X, int64_t k0, int tid) {
        safe = src;
        if constexpr (!TRANS) {
            active = true;
            k = k0 + 4 * (tid & 7);
            step = BK;
#pragma unroll
            for (int p = 0; p < NV; p++) {
                const int row = x0 + stage_row(tid) + RP * p;
                xvalid[p] = row < X ? 4 : 0;
                ptr[p] = src + (int64_t)(row < X ? row : 0) * ld + k;
            }
        } else {
            constexpr int MQ = BX / 4;
            const int mq = tid % MQ, kq = tid / MQ;
            active = BX >= 128 || kq < 8;   // BX = 32: whole waves are idle
            const int col = x0 + 4 * mq;
            const int cleft = X - col;
            k = k0 + KR * kq;
            step = BK * ld;
#pragma unroll
            for (int j = 0; j < KR; j++) {
                xvalid[j] = cleft >= 4 ? 4 : (cleft > 0 ? cleft : 0);
                ptr[j] = src + (k + j) * ld + (cleft > 0 ? col : 0);
            }
        }
    }
    // issues the loads of the next chunk and advances.  GUARD = false: the caller knows every element of the chunk exists (interior tile, chunk
    // inside the contraction range): no address select, no zeroing -- a third of the loop's vector instructions
    template <bool GUARD>
    __device__ __forceinline__ void load(Loaded<NV>& o, int64_t kend) {
        if constexpr (GUARD) {
            const int64_t left = kend - k;
            constexpr int W = TRANS ? KR : 4;   // elements (plain) or k rows (as-it-comes) the thread covers along k
            o.kvalid = left >= W ? W : (left > 0 ? (int)left : 0);
        }
        if constexpr (!TRANS) {
#pragma unroll
            for (int p = 0; p < NV; p++) {
                o.v[p] = load4<VEC>(ptr[p], safe, GUARD ? (xvalid[p] ? o.kvalid : 0) : 4);
                ptr[p] += step;
            }
        } else {
            if (active) {
#pragma unroll
                for (int j = 0; j < KR; j++) {
                    o.v[j] = load4<VEC>(ptr[j], safe, GUARD ? (j < o.kvalid ? xvalid[j] : 0) : 4);
                    ptr[j] += step;
                }
            }
        }
        k += BK;
    }
    template <int T, bool GUARD>
    __device__ __forceinline__ void store(const Loaded<NV>& o, uint16_t* planes, int tid, float (&cs)[4], bool do_cs) const {
        constexpr int PE = tile_elems(BX, TRANS);
        if constexpr (!TRANS) {
            const int kq = tid & 7;
#pragma unroll
            for (int p = 0; p < NV; p++)
                split_store<T>(planes, PE, (stage_row(tid) + RP * p) * KS + 4 * kq, pin_and_zero(o.v[p], GUARD ? (xvalid[p] ? o.kvalid : 0) : 4));
        } else {
            constexpr int MQ = BX / 4, STR = tr_stride(BX);
            const int mq = tid % MQ, kq = tid / MQ;
            if (active) {
#pragma unroll
                for (int j = 0; j < KR; j++) {
                    const float4 q = pin_and_zero(o.v[j], GUARD ? (j < o.kvalid ? xvalid[j] : 0) : 4);
                    if (do_cs) { cs[0] += q.x; cs[1] += q.y; cs[2] += q.z; cs[3] += q.w; }   // running sums of the thread's four columns over k
                    split_store<T>(planes, PE, (KR * kq + j) * STR + 4 * mq, q);
                }
            }
        }
    }
};

// B operand that arrives already split: the weights of a layer as T bf16 planes (written once per optimizer step by gen_weight_planes;
// every workgroup of the forward and d(input) products used to re-split the same 128 x K weights).  No vector work at all: 16-byte loads,
// 16-byte LDS stores into the same tile layouts.  The planes are zero padded to tile multiples in both directions: no guards.
//   plain  (rows = n, k contiguous): thread (row = 4 ((l >> 2) & 3) + (l >> 4) + 16 w + 64 p, 8 k at 8 (l & 3)): the four rows a
//          256-byte LDS cycle writes are r, r + 4, r + 8, r + 12 (disjoint banks at the 80-byte pitch)
//   as-it-comes (rows = k, n contiguous): thread (k row = tid / 16 + 16 p, 8 columns at 8 (tid & 15))
template <int BX, bool TRANS, int T, int NT = 256>
struct PlaneStage {
    static constexpr int ROWS = TRANS ? NT / (BX / 8) : NT / 4;          // tile rows (plain) or k rows (as-it-comes) covered per pass
    static constexpr int NVP = TRANS ? (32 / ROWS > 0 ? 32 / ROWS : 1) : (BX / ROWS > 0 ? BX / ROWS : 1);
    struct Set { u32x4 v[T][NVP]; };
    const uint16_t* ptr;    // plane 0, this thread's first element of the next chunk
    int64_t plane, step, pstep;
    int lds_off;            // bf16 index of the thread's first store in a plane tile
    bool active;
    __device__ __forceinline__ void init(const uint16_t* planes, int64_t plane_elems, int64_t ld, int x0, int64_t k0, int tid) {
        plane = plane_elems;
        const int l = tid & 63, w = tid >> 6;
        if constexpr (!TRANS) {
            const int row = 4 * ((l >> 2) & 3) + (l >> 4) + 16 * w;
            active = row < BX;
            ptr = planes + (int64_t)(x0 + (active ? row : 0)) * ld + k0 + 8 * (l & 3);
            step = BK; pstep = ROWS * ld;
            lds_off = row * KS + 8 * (l & 3);
        } else {
            constexpr int C8 = BX / 8;   // 16-byte pieces per k row
            const int c8 = tid % C8, krow = tid / C8;
            active = krow < 32;
            ptr = planes + (k0 + (active ? krow : 0)) * ld + x0 + 8 * c8;
            step = BK * ld; pstep = ROWS * ld;
            lds_off = krow * tr_stride(BX) + 8 * c8;
        }
    }
    __device__ __forceinline__ void load(Set& o) {
        if ((BX == 128 && NT <= 512) || active) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int p = 0; p < NVP; p++) o.v[t][p] = *reinterpret_cast<const u32x4*>(ptr + t * plane + p * pstep);
        }
        ptr += step;
    }
    __device__ __forceinline__ void store(const Set& o, uint16_t* tiles) const {
        constexpr int PE = tile_elems(BX, TRANS);
        constexpr int LP = ROWS * (TRANS ? tr_stride(BX) : KS);   // LDS distance of the thread's second piece
        if ((BX == 128 && NT <= 512) || active) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int p = 0; p < NVP; p++) *reinterpret_cast<u32x4*>(tiles + t * PE + lds_off + p * LP) = o.v[t][p];
        }
    }
};

// BM x BN output tile, WM x WN waves (WM * WN = 4), each wave FM x FN blocks of 32 x 32.  VEC: the contiguous extent of every operand that
// is split on the fly is a multiple of 4 (every layer product of a network whose widths are; heads and odd shapes take the 4-byte loads).
// BP: B comes as pre-split planes (PlaneStage).
// DB: ONE eight-wave workgroup per CU (two waves per SIMD) with two LDS tile sets: chunk c + 1 is staged into the other set while chunk c
// is multiplied, one barrier per chunk, the staging's vector instructions and LDS writes in the same instruction stream as the MFMAs (the
// compiler does interleave them: 1 MFMA / 7 VALU runs).  Used where it measured faster (small batches, plain bf16; see launch_prec); the
// fp32-accurate product at minibatch size stays on two four-wave workgroups per CU.  (A four-wave version of DB -- one wave per SIMD --
// was 30 % slower than either: one wave issues a vector instruction every ~5 cycles, two waves one every ~2.4.)
template <int BM, int BN, int WM, int WN, bool TA, bool TB, int T, bool VEC, bool BP, bool DB, bool ABF = false, bool CBF = false>
__global__ __launch_bounds__(64 * WM * WN, DB ? 1 : 2) void gemm_kernel(const GemmArgs g) {
    static_assert(!ABF || (T == 1 && BP), "bf16 activations come with the plain-bf16 arithmetic and plane-staged B");
    constexpr int NT = 64 * WM * WN;   // 256 threads; 512 (two waves per SIMD of ONE workgroup) for the double-buffered variant
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    constexpr int EA = tile_elems(BM, TA), EB = tile_elems(BN, TB);
    extern __shared__ __attribute__((aligned(16))) uint16_t dyn_lds[];   // (DB ? 2 : 1) x (T EA + T EB) bf16
    uint16_t* const sA = dyn_lds;
    uint16_t* const sB = dyn_lds + T * EA;
    uint16_t* const sA1 = dyn_lds + (DB ? T * (EA + EB) : 0);            // second tile set (DB)
    uint16_t* const sB1 = sA1 + T * EA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  Tiles that read the same
    // operand rows -- the n tiles of one m tile (they share the A rows: with N = 256 and the plain order every A row was fetched by two
    // XCDs), or all tiles of one row range of a split product -- are therefore given ids that differ by multiples of 8: same XCD,
    // consecutive dispatch slots, second reader hits L2.  (Measured neutral at 65 536 x 256: the 67 MB operand sits in the Infinity Cache.)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int members = g.splits > 1 ? g.m_tiles * g.n_tiles : g.n_tiles;
    const int member = slot % members, grp = (slot / members) * 8 + xcd;
    int tm, tn, tz;
    if (g.splits > 1) { tz = grp; tm = member / g.n_tiles; tn = member % g.n_tiles; if (tz >= g.splits) return; }
    else { tz = 0; tm = grp; tn = member; if (tm >= g.m_tiles) return; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int64_t kbeg = (int64_t)tz * g.k_chunk;
    const int64_t kend = kbeg + g.k_chunk < g.K ? kbeg + g.k_chunk : g.K;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // Two chunks of operands are in flight in registers: the loads of chunk c + 2 are issued when chunk c has been staged, so a load has two
    // chunks of MFMAs (not one) to cover its trip to HBM.  The contraction range is walked in PAIRS of chunks (the host rounds k_chunk
    // to a multiple of 64; chunks past the end load zeros) so that the loop body has no branch around a load.
    typename std::conditional<ABF, PlaneStage<BM, TA, T, NT>, Stage<BM, TA, VEC, NT>>::type sa;
    typename std::conditional<ABF, typename PlaneStage<BM, TA, T, NT>::Set, Loaded<Stage<BM, TA, VEC, NT>::NV>>::type va0, va1;
    if constexpr (ABF) sa.init(reinterpret_cast<const uint16_t*>(g.a), 0, g.lda, m0, kbeg, tid);
    else sa.init(g.a, g.lda, m0, g.M, kbeg, tid);
    float cs[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    const bool do_cs = !ABF && TA && g.colsum != nullptr && tn == 0;   // every n tile stages the same A tile: the first one sums it
    auto load_a = [&](auto& set, auto guard) {
        if constexpr (ABF) sa.load(set);
        else sa.template load<decltype(guard)::value>(set, kend);
    };
    auto store_a = [&](const auto& set, uint16_t* dst, auto guard) {
        if constexpr (ABF) sa.store(set, dst);
        else sa.template store<T, decltype(guard)::value>(set, dst, tid, cs, do_cs);
    };
    // B: split on the fly like A, or (BP) fetched as ready-made bf16 planes
    typename std::conditional<BP, PlaneStage<BN, TB, T, NT>, Stage<BN, TB, VEC, NT>>::type sb;
    typename std::conditional<BP, typename PlaneStage<BN, TB, T, NT>::Set, Loaded<Stage<BN, TB, VEC, NT>::NV>>::type vb0, vb1;
    if constexpr (BP) sb.init(g.bplanes, g.bp_plane, g.bp_ld, n0, kbeg, tid);
    else sb.init(g.b, g.ldb, n0, g.N, kbeg, tid);
    auto load_b = [&](auto& set, auto guard) {
        if constexpr (BP) sb.load(set);
        else sb.template load<decltype(guard)::value>(set, kend);
    };
    auto store_b = [&](const auto& set, uint16_t* dst, auto guard) {
        if constexpr (BP) sb.store(set, dst);
        else { float none[4] = { 0.0f, 0.0f, 0.0f, 0.0f }; sb.template store<T, decltype(guard)::value>(set, dst, tid, none, false); }
    };
    // Plane loads (BP) hit L2 and are fetched ONE chunk ahead into a single register set (two sets of six 16-byte registers on top of A's
    // spilled); within a slot they are issued BEFORE A's loads, so that waiting for them leaves A's newer set in flight.
    load_b(vb0, std::true_type{}); load_a(va0, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);   // set 0 strictly before set 1: the in-order load counter then lets the loop wait for set 0 alone
    if constexpr (!BP) load_b(vb1, std::true_type{});
    load_a(va1, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);
    auto compute = [&](const uint16_t* cA, const uint16_t* cB) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            u32x4 af[FM][T], bf[FN][T];
#pragma unroll
            for (int i = 0; i < FM; i++)
#pragma unroll
                for (int t = 0; t < T; t++) af[i][t] = frag<TA, BM>(cA + t * EA, (wm * FM + i) * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < FN; j++)
#pragma unroll
                for (int t = 0; t < T; t++) bf[j][t] = frag<TB, BN>(cB + t * EB, (wn * FN + j) * 32, ks, lane);
            // small terms first; the FM x FN accumulators of one product are independent MFMAs
            constexpr int NP = T == 3 ? 6 : 1;
            constexpr int pa[6] = { T == 3 ? 2 : 0, 0, 1, 1, 0, 0 }, pb[6] = { 0, 2, 1, 0, 1, 0 };
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int i = 0; i < FM; i++)
#pragma unroll
                    for (int j = 0; j < FN; j++) acc[i][j] = mfma_bf16(af[i][pa[p]], bf[j][pb[p]], acc[i][j]);
        }
    };
    auto pair = [&](auto guard) {   // stages and multiplies the two chunks in registers, fetches the two after them
        if constexpr (!DB) {
            store_a(va0, sA, guard);
            store_b(vb0, sB, guard);
            __syncthreads();
            load_b(vb0, guard);
            __builtin_amdgcn_sched_barrier(0);
            load_a(va0, guard);
            __builtin_amdgcn_sched_barrier(0);   // the loads go out BEFORE the MFMAs they are meant to hide behind (the scheduler sinks them otherwise)
            compute(sA, sB);
            __syncthreads();
            store_a(va1, sA, guard);
            if constexpr (BP) store_b(vb0, sB, guard); else store_b(vb1, sB, guard);
            __syncthreads();
            if constexpr (BP) load_b(vb0, guard); else load_b(vb1, guard);
            __builtin_amdgcn_sched_barrier(0);
            load_a(va1, guard);
            __builtin_amdgcn_sched_barrier(0);
            compute(sA, sB);
            __syncthreads();
        } else {
            // on entry: tile set 0 holds chunk c; va1 (and vb1 / vb0) chunk c + 1, va0 chunk c + 2
            store_a(va1, sA1, guard);
            if constexpr (BP) store_b(vb0, sB1, guard); else store_b(vb1, sB1, guard);
            if constexpr (BP) load_b(vb0, guard); else load_b(vb1, guard);
            load_a(va1, guard);
            compute(sA, sB);
            __syncthreads();
            store_a(va0, sA, guard);
            store_b(vb0, sB, guard);
            load_b(vb0, guard);
            load_a(va0, guard);
            compute(sA1, sB1);
            __syncthreads();
        }
    };
    if constexpr (DB) {   // chunk 0 into tile set 0; its register set goes on to chunk 2
        store_a(va0, sA, std::true_type{});
        store_b(vb0, sB, std::true_type{});
        load_b(vb0, std::true_type{});
        load_a(va0, std::true_type{});
        __syncthreads();
    }
    int64_t kc = kbeg;
    if ((ABF || m0 + BM <= g.M) && (BP || n0 + BN <= g.N)) {
        // interior tile: while this pair AND the pair fetched during it lie inside the range, nothing needs a guard
        for (; kc + (DB ? 6 : 4) * BK <= kend; kc += 2 * BK) pair(std::false_type{});   // DB fetches one chunk further ahead
    }
    for (; kc < kend; kc += 2 * BK) pair(std::true_type{});

    if constexpr (TA && !ABF) {
        if (do_cs) {   // the k-row groups of threads that share a column quad are added in a fixed order through LDS (the tiles are dead)
            constexpr int MQ = BM / 4, NG = BM == 128 ? NT / MQ : 8;
            float* red = reinterpret_cast<float*>(sA);   // [NG][BM]
            const int mq = tid % MQ, kq = tid / MQ;
            if (kq < NG) {
#pragma unroll
                for (int e = 0; e < 4; e++) red[kq * BM + 4 * mq + e] = cs[e];
            }
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) {
                float t = red[tid];
#pragma unroll
                for (int q = 1; q < NG; q++) t += red[q * BM + tid];
                g.colsum[(int64_t)tz * g.colsum_zstride + m0 + tid] = t;
            }
        }
    }

    // ---- epilogue: D layout = lane's column n = lane & 31, register r <-> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block ----
    float* __restrict__ c = g.c + (int64_t)tz * g.c_zstride;
    const int hi = lane >> 5;
    const bool interior = m0 + BM <= g.M && n0 + BN <= g.N;
    const bool want_cs = !TA && g.epi == PPO_MM_EPI_DTANH && g.colsum != nullptr;
    // CBF: the tile is gathered in LDS as bf16 [BM][BN + 8] (the operand tiles are dead) and leaves in 16-byte row pieces
    constexpr int CS = BN + 8;
    uint16_t* const sC = dyn_lds;
    float* const sRed = reinterpret_cast<float*>(dyn_lds + (CBF ? BM * CS : 0));   // [WM][BN] column sums of the waves along m
    static_assert(!CBF || (size_t)BM * CS * 2 + (size_t)WM * BN * 4 <= (size_t)(DB ? 2 : 1) * T * (EA + EB) * 2, "the C tile must fit the operand tiles' LDS");
    if (CBF || want_cs) __syncthreads();   // every wave is done reading the operand tiles
#pragma unroll
    for (int j = 0; j < FN; j++) {
        const int nl = (wn * FN + j) * 32 + (lane & 31);
        const int n = n0 + nl;
        const bool n_ok = n < g.N;
        const float bias = ((g.epi == PPO_MM_EPI_BIAS || g.epi == PPO_MM_EPI_BIAS_TANH) && n_ok) ? g.aux[n] : 0.0f;
        float csum = 0.0f;
#pragma unroll
        for (int i = 0; i < FM; i++) {
            const int ml = (wm * FM + i) * 32 + 4 * hi;
            const int mb = m0 + ml;
            float v[16];
            if (g.epi == PPO_MM_EPI_BIAS_TANH) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = tanh_mufu(acc[i][j][r] + bias);
            } else if (g.epi == PPO_MM_EPI_DTANH) {
                float h[16];   // all sixteen loads in flight together (clamped address where the element does not exist)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    const int64_t at = (n_ok && m < g.M) ? (int64_t)m * g.ld_aux + n : 0;
                    if constexpr (ABF) h[r] = u2f((uint32_t)reinterpret_cast<const uint16_t*>(g.aux)[at] << 16);   // the stored (bf16) activation
                    else h[r] = g.aux[at];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = acc[i][j][r] * (1.0f - h[r] * h[r]);
                if (want_cs) {
#pragma unroll
                    for (int r = 0; r < 16; r++) csum += (mb + (r & 3) + 8 * (r >> 2) < g.M) ? v[r] : 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = acc[i][j][r] + bias;
            }
            if constexpr (CBF) {
#pragma unroll
                for (int r = 0; r < 16; r++) sC[(ml + (r & 3) + 8 * (r >> 2)) * CS + nl] = (uint16_t)(pack_rne(v[r], 0.0f) & 0xffffu);
            } else if (interior) {
#pragma unroll
                for (int r = 0; r < 16; r++) c[(int64_t)(mb + (r & 3) + 8 * (r >> 2)) * g.ldc + n] = v[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (n_ok && m < g.M) c[(int64_t)m * g.ldc + n] = v[r];
                }
            }
        }
        if (want_cs) {
            csum += __shfl_xor(csum, 32, 64);
            if (hi == 0) sRed[wm * BN + nl] = csum;
        }
    }
    if constexpr (CBF) {
        __syncthreads();
        // rows of the tile that exist, all BN columns (the destination's row pitch covers the padded width): 16 bytes per thread and pass
        uint16_t* __restrict__ cb = reinterpret_cast<uint16_t*>(g.c);
        constexpr int P8 = BN / 8;
        for (int e = tid; e < BM * P8; e += NT) {
            const int row = e / P8, c8 = e % P8;
            if (m0 + row < g.M) *reinterpret_cast<u32x4*>(cb + (int64_t)(m0 + row) * g.ldc + n0 + 8 * c8) = *reinterpret_cast<const u32x4*>(sC + row * CS + 8 * c8);
        }
    } else if (want_cs) {
        __syncthreads();
    }
    if (want_cs && tid < BN && n0 + tid < g.N) {
        float t = sRed[tid];
#pragma unroll
        for (int q = 1; q < WM; q++) t += sRed[q * BN + tid];   // fixed order
        g.colsum[(int64_t)tm * g.colsum_zstride + n0 + tid] = t;
    }
}

// one instantiation: LDS size, (for the double-buffered variant) the attribute that allows more than 64 KB, launch
template <int BM, int BN, int WM, int WN, bool TA, bool TB, int T, bool VEC, bool BP, bool DB, bool ABF = false, bool CBF = false>
hipError_t launch_one(const GemmArgs& g, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)(DB ? 2 : 1) * T * (tile_elems(BM, TA) + tile_elems(BN, TB)) * sizeof(uint16_t);
    auto kern = gemm_kernel<BM, BN, WM, WN, TA, TB, T, VEC, BP, DB, ABF, CBF>;
    if constexpr (DB) {
        static std::atomic<unsigned long long> lds_ok{0};
        const hipError_t attr = allow_dynamic_lds(lds_ok, reinterpret_cast<const void*>(kern), (int)lds);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
    return hipGetLastError();
}
template <int BM, int BN, int WM, int WN, int T, bool VEC>
hipError_t launch_cfg(const GemmArgs& g_in, bool ta, bool tb, int splits, bool db, hipStream_t s) {
    GemmArgs g = g_in;
    g.m_tiles = (g.M + BM - 1) / BM; g.n_tiles = (g.N + BN - 1) / BN; g.splits = splits;
    const int64_t groups = splits > 1 ? splits : g.m_tiles, members = splits > 1 ? (int64_t)g.m_tiles * g.n_tiles : g.n_tiles;
    const int64_t blocks = (groups + 7) / 8 * 8 * members;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    if constexpr (BM == 128 && BN == 128) {
        if (db) {   // double-buffered variant: the big tile only, the orientations the layer products use
            // eight waves (4 x 2, each 32 x 64): two per SIMD, all of ONE workgroup
            if (g.bplanes) return tb ? launch_one<BM, BN, 4, 2, false, true, T, VEC, true, true>(g, grid, s) : launch_one<BM, BN, 4, 2, false, false, T, VEC, true, true>(g, grid, s);
            if (ta && tb) return launch_one<BM, BN, 4, 2, true, true, T, VEC, false, true>(g, grid, s);
            if (!ta && !tb) return launch_one<BM, BN, 4, 2, false, false, T, VEC, false, true>(g, grid, s);
            if (!ta && tb) return launch_one<BM, BN, 4, 2, false, true, T, VEC, false, true>(g, grid, s);
        }
    }
    if (g.bplanes) {   // plain A, B from planes (forward and d(input) of a layer)
        if constexpr (BM == 128) return tb ? launch_one<BM, BN, WM, WN, false, true, T, VEC, true, false>(g, grid, s) : launch_one<BM, BN, WM, WN, false, false, T, VEC, true, false>(g, grid, s);
        else return hipErrorInvalidValue;
    }
    if (!ta && !tb) return launch_one<BM, BN, WM, WN, false, false, T, VEC, false, false>(g, grid, s);
    if (!ta && tb) return launch_one<BM, BN, WM, WN, false, true, T, VEC, false, false>(g, grid, s);
    if (ta && !tb) return launch_one<BM, BN, WM, WN, true, false, T, VEC, false, false>(g, grid, s);
    return launch_one<BM, BN, WM, WN, true, true, T, VEC, false, false>(g, grid, s);
}
template <int T, bool VEC>
hipError_t launch_prec(const GemmArgs& g, bool ta, bool tb, int splits, hipStream_t s) {
    // Which main loop: two workgroups per CU with one tile set each, or ONE eight-wave workgroup with two tile sets (DB)?  Measured on
    // configs[4] (A/B in one call): the fp32-accurate product at minibatch size is 7 % faster with two workgroups (1.87 against 2.00 ms per
    // step); the same product on a rollout step's 2048 rows -- few workgroups, nothing to overlap with but itself -- is 11 % faster double-
    // buffered (13.1 against 14.8 ms per rollout), and so is plain bf16, which stages little (3 %).
    const bool db = T == 1 || (g.M <= 8192 && splits == 1);
    if (g.N <= 32) return launch_cfg<128, 32, 4, 1, T, VEC>(g, ta, tb, splits, false, s);
    if (g.M <= 32 && !g.bplanes) return launch_cfg<32, 128, 1, 4, T, VEC>(g, ta, tb, splits, false, s);
    return launch_cfg<128, 128, 2, 2, T, VEC>(g, ta, tb, splits, db, s);
}

// One thread per element of a padded plane row: weights [N, K] f32 -> T planes [t][n_pad][k_pad] bf16 (truncation terms, or one
// round-to-nearest term), zeros in the padding.
struct PlaneJob { int64_t src_off, dst_off, first; int N, K, n_pad, k_pad; };
struct PlaneJobs { PlaneJob j[2 * GEN_MAX_LAYERS]; int n; int64_t total; };
// frags (T = 1, may be null): a second copy of the same bf16 values in MFMA-fragment order for the fused forward kernels
// (kernels_generic_fused.hip): [column block of 32 rows n][k step of 16][lane][8], lane (n % 32, kg) holding k = 16 ks + 8 kg .. + 7 of row n --
// a wave's B-fragment load is then 1 KiB of consecutive bytes instead of 64 pieces of 32 different rows.
template <int T>
__global__ __launch_bounds__(256) void weight_planes_kernel(const float* __restrict__ params, uint16_t* __restrict__ planes, uint16_t* __restrict__ frags,
                                                            const PlaneJobs jobs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= jobs.total) return;
    int q = 0;
    for (int k = 1; k < jobs.n; k++) if (i >= jobs.j[k].first) q = k;
    const PlaneJob& J = jobs.j[q];
    const int64_t e = i - J.first;
    const int n = (int)(e / J.k_pad), k = (int)(e % J.k_pad);
    const float x = (n < J.N && k < J.K) ? params[J.src_off + (int64_t)n * J.K + k] : 0.0f;
    const int64_t pe = (int64_t)J.n_pad * J.k_pad;
    uint16_t* d = planes + J.dst_off + e;
    if constexpr (T == 3) {
        const uint32_t u0 = f2u(x);
        const float r1 = x - u2f(u0 & 0xffff0000u);
        const uint32_t u1 = f2u(r1);
        const float r2 = r1 - u2f(u1 & 0xffff0000u);
        d[0] = (uint16_t)(u0 >> 16); d[pe] = (uint16_t)(u1 >> 16); d[2 * pe] = (uint16_t)(f2u(r2) >> 16);
    } else {
        const uint16_t b = (uint16_t)(pack_rne(x, 0.0f) & 0xffffu);
        d[0] = b;
        if (frags) {
            const int64_t ksteps = J.k_pad / 16;
            frags[J.dst_off + (((int64_t)(n >> 5) * ksteps + (k >> 4)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)] = b;
        }
    }
}

}  // namespace

// c[z][M, N] (z < splits, slabs c_zstride apart) = epilogue(sum over the z-th k range of A(m, k) B(n, k)); splits > 1 cuts the contraction
// into ranges of k_chunk (rounded up to a multiple of 32) and requires the plain epilogue.
hipError_t launch_matmul(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* c,
                         int64_t ldc, int epilogue, const float* aux, int64_t ld_aux, int precision, int splits, int64_t c_zstride, float* colsum,
                         int64_t colsum_zstride, const uint16_t* bplanes, int64_t bp_plane, int64_t bp_ld, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (M > 0x7fffffff || N > 0x7fffffff || K < 0 || splits < 1) return hipErrorInvalidValue;
    if (splits > 1 && epilogue != PPO_MM_EPI_NONE) return hipErrorInvalidValue;
    GemmArgs g;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.c = c; g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = K;
    int64_t kc = (K + splits - 1) / splits;
    kc = (kc + 2 * BK - 1) / (2 * BK) * (2 * BK);   // the kernel walks pairs of chunks
    if (kc < 2 * BK) kc = 2 * BK;
    g.k_chunk = kc;
    g.c_zstride = c_zstride;
    g.epi = epilogue; g.aux = aux; g.ld_aux = ld_aux;
    g.colsum = colsum; g.colsum_zstride = colsum_zstride;   // TA: sums of A over each k range; plain A with PPO_MM_EPI_DTANH: column sums of the result per m tile
    g.bplanes = (!trans_a && splits == 1) ? bplanes : nullptr; g.bp_plane = bp_plane; g.bp_ld = bp_ld;
    // contiguous extent of an operand: k (plain) or its row / column index (transposed)
    const bool vec = ((trans_a ? M : K) % 4 == 0) && (g.bplanes != nullptr || (trans_b ? N : K) % 4 == 0);
    if (precision == PPO_MM_BF16) return vec ? launch_prec<1, true>(g, trans_a, trans_b, splits, s) : launch_prec<1, false>(g, trans_a, trans_b, splits, s);
    return vec ? launch_prec<3, true>(g, trans_a, trans_b, splits, s) : launch_prec<3, false>(g, trans_a, trans_b, splits, s);
}

// The same products with bf16 STORAGE (ppo_config.compute_dtype = PPO_DTYPE_BF16): both operands are bf16 in memory -- A(m, k) and B(n, k)
// read plain or transposed like launch_matmul's, rows padded to 128 and the contraction zero-padded to a multiple of 64 by the caller's
// buffers (PlaneStage has no guards) -- products on v_mfma_f32_32x32x16_bf16 with f32 accumulation, the result f32 or (c_bf16) bf16 rounded
// to nearest even.  One tile shape: 128 x 128, eight waves, double-buffered.  colsum: see GemmArgs.
hipError_t launch_matmul_bf16(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                              void* c, int64_t ldc, bool c_bf16, int epilogue, const void* aux, int64_t ld_aux, int splits, int64_t c_zstride,
                              float* colsum, int64_t colsum_stride, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (M > 0x7fffffff || N > 0x7fffffff || K < 0 || splits < 1) return hipErrorInvalidValue;
    if (splits > 1 && (epilogue != PPO_MM_EPI_NONE || c_bf16)) return hipErrorInvalidValue;
    GemmArgs g{};
    g.a = reinterpret_cast<const float*>(a); g.lda = lda; g.b = nullptr; g.ldb = ldb; g.c = reinterpret_cast<float*>(c); g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = K;
    int64_t kc = (K + splits - 1) / splits;
    kc = (kc + 2 * BK - 1) / (2 * BK) * (2 * BK);
    if (kc < 2 * BK) kc = 2 * BK;
    g.k_chunk = kc; g.c_zstride = c_zstride;
    g.epi = epilogue; g.aux = reinterpret_cast<const float*>(aux); g.ld_aux = ld_aux;
    g.colsum = colsum; g.colsum_zstride = colsum_stride;
    g.bplanes = b; g.bp_plane = 0; g.bp_ld = ldb;
    g.m_tiles = (g.M + 127) / 128; g.n_tiles = (g.N + 127) / 128; g.splits = splits;
    const int64_t groups = splits > 1 ? splits : g.m_tiles, members = splits > 1 ? (int64_t)g.m_tiles * g.n_tiles : g.n_tiles;
    const int64_t blocks = (groups + 7) / 8 * 8 * members;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    if (!trans_a && !trans_b) return c_bf16 ? launch_one<128, 128, 4, 2, false, false, 1, true, true, true, true, true>(g, grid, s)
                                            : launch_one<128, 128, 4, 2, false, false, 1, true, true, true, true, false>(g, grid, s);
    if (!trans_a && trans_b && c_bf16) return launch_one<128, 128, 4, 2, false, true, 1, true, true, true, true, true>(g, grid, s);
    if (trans_a && trans_b && !c_bf16) return launch_one<128, 128, 4, 2, true, true, 1, true, true, true, true, false>(g, grid, s);
    return hipErrorNotSupported;
}

// f32 [rows, K] -> bf16 [rows, ld] (round to nearest even), columns K .. ld - 1 zero: the layer input of a bf16 network
// idx != nullptr: only rows idx[0 .. rows - 1] are converted, each IN PLACE (row idx[j] of src -> row idx[j] of dst): what a stand-alone step of M rows needs of a
// batch of B (duplicate indices write the same values twice)
__global__ __launch_bounds__(256) void to_bf16_pad_kernel(const float* __restrict__ src, int64_t rows, int K, uint16_t* __restrict__ dst, int ld, const int32_t* __restrict__ idx) {
    const int64_t total = rows * (ld / 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t j = i / (ld / 2), r = idx ? (int64_t)idx[j] : j;
        const int k = 2 * (int)(i % (ld / 2));
        const float x0 = k < K ? src[r * K + k] : 0.0f, x1 = k + 1 < K ? src[r * K + k + 1] : 0.0f;
        reinterpret_cast<uint32_t*>(dst)[r * (ld / 2) + k / 2] = pack_rne(x0, x1);
    }
}
hipError_t launch_to_bf16_pad(const float* src, int64_t rows, int K, uint16_t* dst, int ld, hipStream_t s, const int32_t* idx) {
    if (rows <= 0) return hipSuccess;
    const int64_t total = rows * (ld / 2);
    const int64_t gb = (total + 255) / 256;
    hipLaunchKernelGGL(to_bf16_pad_kernel, dim3((unsigned)(gb < 4096 ? gb : 4096)), dim3(256), 0, s, src, rows, K, dst, ld, idx);
    return hipGetLastError();
}

// Pre-split weights of every layer of both nets (generic.hpp: GenericCtx::wplanes): one launch.
hipError_t gen_weight_planes(const GenericCtx& g, const float* params, hipStream_t s) {
    const GenLayout& L = g.L;
    PlaneJobs jobs{};
    int64_t first = 0;
    for (int net = 0; net < 2; net++)
        for (int l = 0; l < L.n_layers; l++) {
            PlaneJob& J = jobs.j[jobs.n++];
            J.src_off = L.w_off[net][l]; J.dst_off = g.wp_off[net][l]; J.first = first;
            J.N = L.out_dim[net][l]; J.K = L.in_dim[l]; J.n_pad = g.wp_npad[net][l]; J.k_pad = g.wp_kpad[l];
            first += (int64_t)J.n_pad * J.k_pad;
        }
    jobs.total = first;
    const dim3 grid((unsigned)((first + 255) / 256)), block(256);
    if (g.gemm_prec == PPO_MM_BF16) hipLaunchKernelGGL(weight_planes_kernel<1>, grid, block, 0, s, params, g.wplanes, g.wfrags, jobs);
    else hipLaunchKernelGGL(weight_planes_kernel<3>, grid, block, 0, s, params, g.wplanes, (uint16_t*)nullptr, jobs);
    return hipGetLastError();
}
