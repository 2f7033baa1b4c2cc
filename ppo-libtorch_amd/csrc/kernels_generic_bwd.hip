// ppo-libtorch_amd/csrc/kernels_generic_bwd.hip -- the backward pass of one Linear (+ tanh) layer of a bf16-storage network (generic.hpp; BASELINE
// configs[4]: obs 376, 4 x 256, heads [3,3,3,2]) as ONE kernel: loss.backward() through `torch::nn::Linear` / `Tanh` of Agent.cpp:25-59 generalised,
//
//      dW_l  = dZ_l^T h_{l-1}                       (the weight gradient: contracts over the minibatch rows)
//      dZ_{l-1} = (dZ_l W_l) (1 - h_{l-1}^2)        (what the layer below receives; its column sums are that layer's bias gradient)
//
// Until round 5 these were two launches of the tiled product (kernels_gemm.hip), each reading dZ_l and h_{l-1} from HBM: 168 MB per hidden layer and net
// at 65 536 rows, at 8 % of the matrix pipe (K = 256 is eight k steps of a 128 x 128 tile: prologue, epilogue and staging dominate).  Here a workgroup owns
// a RANGE of rows and a 64-wide COLUMN BLOCK c of the layer's input side, and per 64-row tile has dZ_l [64 x N] and h_{l-1}[:, block c] [64 x 64] brought
// into LDS ONCE, by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass), into a ring of THREE buffers: two tiles (80 KB per CU) are in flight
// while the third is multiplied -- with one tile ahead the kernel ran at the latency of its loads (a tile every ~2 us, 75 us per launch at 65 536 rows).
// Its eight waves have two jobs, one wave of each kind per SIMD:
//
//   waves 0-3 (P1)  D[kcol][row] = sum_n W[n][kcol] dZ[row][n]: W_l[:, block c] stays in registers for the whole launch as the A operand (64 registers at
//                   N = 256), the dZ rows are the B operand (16-byte LDS reads): the result has lane = row, registers = 4-groups of consecutive columns, so
//                   the tanh' epilogue reads h and writes the bf16 result as 8-byte pieces; per-lane column sums accumulate over the tiles and meet once at the end
//   waves 4-7 (P2)  D[n][kcol] += sum_row dZ[row][n] h[row][kcol]: both operands by transposing LDS reads (ds_read_b64_tr_b16); the accumulators of the
//                   workgroup's [N x 64] slice of dW (64 registers per wave at N = 256) stay in registers over all its tiles and leave as one slab per row range
//
// 16 + 16 MFMAs per SIMD and tile, one barrier per tile.  The four (ld / 64) column-block workgroups of a row range run on ONE XCD in consecutive dispatch
// slots: the dZ_l tile the first one pulls from HBM the other three find in that XCD's L2 (measured: 78 MB fetched per launch for 67 MB of operands).
// The LDS images are XOR-swizzled by row (16-byte chunk index ^ f(row)), not padded: the same dZ tile is read by rows (P1) and by columns (P2), and no single
// pitch serves both; the DMA writes lane-linearly, so the swizzle sits on its per-lane SOURCE address and on every read (the argument is at swz_d / swz_h).
// Layer 0 (no layer below) runs P2 on all eight waves; the head (N = logits padded to 32) is the same kernel with two k steps.
// Arithmetic = launch_matmul_bf16's: bf16 operands, f32 accumulation, tanh' and the column sums in f32 before the result is rounded to bf16.
// Everything is summed in a fixed order: same inputs, same bits.
#include "generic.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BW_WAVES = 8, BW_THREADS = 64 * BW_WAVES;
constexpr int BW_ROWS = 64;    // rows per tile
constexpr int BW_KC = 64;      // columns of the layer's input side per workgroup
constexpr int BW_RING = 3;     // LDS buffers of the tile ring

struct BwdArgs {
    const uint16_t* d; int64_t ldd;        // dZ_l [rows + pad][ldd] bf16 (columns >= the layer's width are zero)
    const uint16_t* h; int64_t ldh;        // h_{l-1}, or the layer-0 input [.][ldh] bf16
    const int32_t* idx;                    // layer 0 only (no P1), may be null: row r of the minibatch is row idx[r] of h (the update's observations, read in place)
    const uint16_t* w; int64_t ldw;        // bf16 plane of W_l [n_pad][ldw], zero padded (P1 only)
    uint16_t* dz_out; int64_t ld_out;      // dZ_{l-1} (P1 only)
    float* slab; int64_t slab_stride;      // [S][n_real][k_real] partial weight gradients, one per row range
    float* colsum; int64_t ld_cs;          // [S][ld_cs] column sums of dZ_{l-1} over the range's rows (P1 only)
    const uint16_t* zeros;                 // >= 16 bytes of zeros: what the DMA fetches for rows past the minibatch
    int64_t rows;                          // rows of the minibatch
    int n_real, k_real;                    // the layer's real widths (out, in)
    int S, CB, tiles_per_range;
};

__device__ __forceinline__ float bw_u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t bw_pack(float x0, float x1) { const bf16x2 v = { (__bf16)x0, (__bf16)x1 }; return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint2 bw_read_tr16(const uint16_t* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ f32x16 bw_mfma(const u32x4 a, const u32x4 b, const f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// LDS images: [row][W] bf16, W = 8 CPR, no padding; the 16-byte chunk j of row r sits at chunk j ^ f(r).
//   dZ tile (CPR = 4, 16 or 32; a row is 16, 64 or 128 dwords): f(r) = ((r & 3) << 2 | (r >> 2) & 3) & (CPR - 1), a bijection of r & 15.
//     read by rows (ds_read_b128, lane = row, 16-lane groups hold 16 different r & 15): 16 different chunks ^ f -> 16 different bank quads.
//     read by columns (ds_read_b64_tr_b16: a 32-lane half covers 4 consecutive rows x 32 columns = 4 chunks per row): (c0 + j) ^ f(r) sends the
//     rows' four chunks to four DIFFERENT aligned groups of four chunks ((c0 >> 2) ^ (r & 3)): 4 x 16 dwords, all 64 banks once.
//   h / W / result tiles (64 columns: CPR = 8, a row is 32 dwords, rows r and r + 2 share banks): bit 2 of f(r) = (r >> 1) & 1 moves rows r + 2, r + 3
//     to the other half of their bank window: the same transposing read is conflict free; the low bits (r >> 2) & 3 spread the epilogue's 8-byte accesses
//     (lane = row, all lanes at one logical column) over the eight chunks: 2-way instead of 8-way (measured: 35 % of the LDS cycles were conflicts without).
template <int CPR>
__device__ __forceinline__ int swz_d(int r) { return (((r & 3) << 2) | ((r >> 2) & 3)) & (CPR - 1); }
__device__ __forceinline__ int swz_h(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
// A bare s_barrier with the compiler held to it: the intrinsic is "no memory" for LLVM, and the DMA's LDS writes are invisible to it, so nothing else would keep
// the next tile's ds_reads (which see no store they could depend on) from being scheduled ABOVE the barrier, where the other waves' pieces may still be under way
// (first version: wrong gradients now and then).  The waits in front of it are the caller's (counted vmcnt, lgkmcnt(0)).
__device__ __forceinline__ void bw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// 64 lanes x 16 bytes -> 1 KiB at lds_wave_base (wave-uniform), lane-linear.  As inline asm, not __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) cannot tell
// which LDS bytes a DMA will write and puts s_waitcnt vmcnt(0) in front of EVERY later ds_read of the array (seen in this kernel's assembly: four per
// tile), which drains the ring it is there to keep full.  An asm DMA is invisible to that bookkeeping: the only waits on the vector-memory counter in the
// tile loop are the counted ones written there (the loop holds no compiler-visible load whose own wait the extra operations could make too short; its
// stores are never waited for).  M0 = the destination's LDS byte address, saved and restored around the instruction (the compiler reserves M0).
// The destination is the LDS BYTE ADDRESS as an integer (bw_lds_addr of the array's start + offsets): handed over as a generic pointer, every instruction paid a
// generic -> local conversion (null check, aperture base) and two v_readfirstlane on top of its address arithmetic.
// How a tile reaches LDS: the LDS-DMA ring (global_load_lds_dwordx4 by inline asm, counted waits).  The fetch, not HBM and not the products, is what the launch waits
// for: a timing-only build without any fetch runs the hidden layers' launch in 35 us instead of 53 and layer 0's in 30 instead of 54; with every source cache-resident it
// is still 53; a CU's address path moves a tile's 40 KB in ~900 cycles (tools/probes/l2_stream: 45 - 59 B/clk per CU by LDS-DMA, ~30 by plain loads, whatever the number
// of CUs) and the waves that issue the instructions are blocked meanwhile (in-kernel stamps: 800 - 2000 cycles per tile).  Plain loads into registers + ds_write (two
// register sets, the tile loop unrolled by two) were built: 62.8 / 55.5 us against 51.8 / 53.0; so was issuing the pieces one at a time between the products
// (BW_DMA_SPREAD: the stall moves into the products) and leaving the fetch to the P2 waves alone (BW_ALL_MOVERS = 0: below).
#ifndef BW_NT   /* exploration builds: bit 0 = non-temporal DMA of the h tile (read by ONE workgroup), bit 1 = of the dZ tile (read by the CB workgroups of a row range), bit 2 = non-temporal stores of dZ_{l-1} */
#define BW_NT 0
#endif
__device__ __forceinline__ uint32_t bw_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
template <bool NT = false>
__device__ __forceinline__ void bw_glds16(const uint16_t* src, uint32_t dst) {
#ifdef BW_GLDS_BUILTIN
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
    return;
#endif
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst);   // wave-uniform by construction (it depends on the wave's index): the compiler has to be told
    uint32_t keep;
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(d) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(d) : "memory");
}
// The same with the source as a wave-uniform base (SGPR pair) + a 32-bit byte offset per lane: no 64-bit address arithmetic on the vector ALU per instruction.
template <bool NT = false>
__device__ __forceinline__ void bw_glds16_s(const uint16_t* sbase, uint32_t voff_bytes, uint32_t dst) {
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst);
    uint32_t keep;
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(d) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(d) : "memory");
}

// 64 lanes x 4 bytes -> 256 bytes at lds_wave_base, lane-linear: the index list's entries of a tile's rows (layer 0, rows read in place).  A DMA, not a load into a
// register: a register that an asm load has in flight is, for the compiler, a value it may copy at any time -- hipcc (ROCm 7.2) placed such a copy IN FRONT of the
// counted wait in one instantiation (harmless there by luck: tests/test_asm_hazards.py found it) -- whereas nothing can touch the LDS words before the ds_read that
// follows the wait.
__device__ __forceinline__ void bw_glds4(const int32_t* src, uint32_t dst) {
    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(d) : "memory");
}

// MFMA operand fragment by TRANSPOSING reads out of an image [k rows][W columns]: lane (i = lane & 31, kg = lane >> 5) receives column x0 + i of the
// eight rows 16 ks + 8 kg .. + 7.  Lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 (see kernels_gemm.hip: frag).
template <int W, bool DZ>
__device__ __forceinline__ u32x4 frag_tr(const uint16_t* img, int x0, int ks, int lane) {
    const int r = 16 * ks + 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int col = x0 + 16 * ((lane & 31) >> 4) + 4 * (lane & 3);
    const int chunk = col >> 3, within = col & 7;
    const int f0 = DZ ? swz_d<W / 8>(r) : swz_h(r), f1 = DZ ? swz_d<W / 8>(r + 4) : swz_h(r + 4);
    const uint2 lo = bw_read_tr16(img + r * W + ((chunk ^ f0) << 3) + within);
    const uint2 hi = bw_read_tr16(img + (r + 4) * W + ((chunk ^ f1) << 3) + within);
    const u32x4 v = { lo.x, lo.y, hi.x, hi.y };
    return v;
}

// The two element offsets of frag_tr's reads at k step 0 (a k step further is 16 rows further: both swizzles repeat with period 16 in the row, so the offset of k step
// ks is this + 16 ks W -- an immediate).  Computed ONCE per launch: formed inside the tile loop, the swizzle arithmetic of the 32 transposing reads of a tile was 230
// vector instructions per wave and tile against 16 MFMAs (counters: 15.4 M vector-ALU instructions per launch, the vector ALU 40 % busy).
template <int W, bool DZ>
__device__ __forceinline__ void frag_off(int x0, int lane, int& lo, int& hi) {
    const int r = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int col = x0 + 16 * ((lane & 31) >> 4) + 4 * (lane & 3);
    const int chunk = col >> 3, within = col & 7;
    const int f0 = DZ ? swz_d<W / 8>(r) : swz_h(r), f1 = DZ ? swz_d<W / 8>(r + 4) : swz_h(r + 4);
    lo = r * W + ((chunk ^ f0) << 3) + within;
    hi = (r + 4) * W + ((chunk ^ f1) << 3) + within;
}
__device__ __forceinline__ u32x4 frag_at(const uint16_t* img, int lo, int hi) {
    const uint2 a = bw_read_tr16(img + lo), b = bw_read_tr16(img + hi);
    const u32x4 v = { a.x, a.y, b.x, b.y };
    return v;
}

// KCB = 64-column blocks of the input side a workgroup owns: 1 beside P1 (its weight block and result tile are sized for 64 columns), 2 for layer 0, whose only
// product is the weight gradient (the dZ_0 tile is then fetched by ld / 128 workgroups instead of ld / 64).
// Two launches in one (the same layer of the two nets, each sized for half the chip): workgroups [0, split) run a[0], the rest a[1]; split is a multiple of 8,
// so both halves keep the XCD order below.  split == gridDim.x: one net.
struct BwdPair { BwdArgs a[2]; int split; };
template <int NB, bool P1, int KCB = 1>
__global__ __launch_bounds__(BW_THREADS, 1) void bwd_layer_kernel(const BwdPair pp) {
    const int second = (int)blockIdx.x >= pp.split ? 1 : 0;
    const BwdArgs& a = pp.a[second];
    const int bid = (int)blockIdx.x - (second ? pp.split : 0);
    static_assert(KCB == 1 || !P1, "P1 is built for one 64-column block");
    constexpr int HSZ = KCB * BW_ROWS * BW_KC;   // elements of one h buffer: KCB images [64][64]
    constexpr int N = 32 * NB;                 // width of dZ_l as staged (the real width zero padded)
    constexpr int CPR = N / 8;                 // 16-byte chunks per dZ row
    constexpr int KS1 = N / 16;                // k steps of P1
    constexpr int DSZ = BW_ROWS * (N < 64 ? 64 : N);      // elements of one dZ buffer (>= the weight block [N][64])
    constexpr int P2W = P1 ? 4 : 8;            // waves that run P2
    constexpr int NBW = (NB + P2W - 1) / P2W;  // n blocks of a P2 wave (each against both 32-column halves of the block)
    // Movers: the waves that issue a tile's DMA.  Every wave fetches its eighth of the tile (BW_ALL_MOVERS = 1, shipped).  0: beside P1 only the four P2 waves do -- they
    // have the slack (the P1 waves carry the epilogue and are the longer path of an iteration), and a wave is blocked while the CU's address path moves what it issued
    // (~900 cycles for the tile's 40 KB: stamps, NOTES) -- measured in one call: 53.5 against 52.6 us for the hidden layers' launch: the blocked time only moves.
#ifndef BW_ALL_MOVERS
#define BW_ALL_MOVERS 1
#endif
    constexpr int MW = (P1 && !BW_ALL_MOVERS) ? 4 : BW_WAVES;      // movers
    constexpr int DPW = (CPR + MW - 1) / MW;   // 1-KiB DMA pieces of the dZ tile per mover (the tile has CPR of them)
    constexpr int HPW = BW_WAVES / MW;         // 1-KiB pieces of an h image per mover (an image has eight)
    static_assert(NB == 1 || NB == 4 || NB == 8, "dZ widths of 32 (a head), 128 and 256");
    extern __shared__ __attribute__((aligned(16))) uint16_t bw_lds[];
    uint16_t* const sD = bw_lds;                          // [3][64][N]  (buffer 2 first holds the weight block [N][64], once)
    uint16_t* const sH = sD + BW_RING * DSZ;              // [3][64][64]
    uint16_t* const sO = sH + BW_RING * HSZ;              // [2][64][64] result tile of P1
    float* const sRed = reinterpret_cast<float*>(sO + (P1 ? 2 : 0) * BW_ROWS * BW_KC);   // [2][64]
    int32_t* const sIdx = reinterpret_cast<int32_t*>(sRed + 2 * BW_KC) + BW_WAVES * 256;  // layer 0: [2][8 waves][64] index-list entries, behind the waves' spare KiBs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs; the CB column blocks of a row range take consecutive slots of ONE XCD
    const int xcd = bid & 7, slot = bid >> 3;
    const int z = (slot / a.CB) * 8 + xcd, c = slot % a.CB;   // c counts blocks of 64 KCB columns
    if (z >= a.S) return;
#ifdef BW_INTERLEAVE   /* exploration: range z takes tiles z, z + S, z + 2 S, ...: at any moment the workgroups of a launch read ONE contiguous band of rows */
    const int64_t all_tiles = (a.rows + BW_ROWS - 1) / BW_ROWS;
    int n_tiles = z < all_tiles ? (int)((all_tiles - z + a.S - 1) / a.S) : 0;
    if (n_tiles > a.tiles_per_range) n_tiles = a.tiles_per_range;
    auto tile_row = [&](int t) -> int64_t { return ((int64_t)t * a.S + z) * BW_ROWS; };
#else
    const int64_t row_begin = (int64_t)z * a.tiles_per_range * BW_ROWS;
    int n_tiles = a.tiles_per_range;
    {
        const int64_t left = a.rows - row_begin;
        const int64_t have = left <= 0 ? 0 : (left + BW_ROWS - 1) / BW_ROWS;
        if (have < n_tiles) n_tiles = (int)have;
    }
    auto tile_row = [&](int t) -> int64_t { return row_begin + (int64_t)t * BW_ROWS; };
#endif
    const int li = lane & 31, kg = lane >> 5;
    const bool p1_wave = P1 && wave < 4;
    const int cb1 = wave & 1, rb1 = (wave >> 1) & 1;   // P1: the wave's 32 columns (of the block's 64) and 32 rows (of the tile's 64)
    const int pj = P1 ? wave - 4 : wave;               // P2: the wave's index among the P2 waves (< 0: not one)
    const bool mover = MW == BW_WAVES || pj >= 0;      // (wave-uniform) this wave issues DMA
    const int mv = MW == BW_WAVES ? wave : (pj >= 0 ? pj : 0);   // its index among the movers

    // What a wave keeps for the whole launch lives in ONE register block `st`: a P1 wave's weight fragments (4 registers per k step) or a P2 wave's
    // accumulators (16 per n block and column half).  The two kinds of waves never need both, but two arrays would both be live through the tile loop for the
    // register allocator (one function, wave-uniform branches): 64 + 64 registers at N = 256, and the kernel spilled into the loop -- every scratch reload
    // is a vector-memory operation whose s_waitcnt drains the DMA ring.
    // The block is typed as INTEGERS (bf16 pairs are not floats: kept in float-typed registers, the weight fragments came back changed -- a pair whose upper
    // half is zero is a denormal float, and something on the way canonicalised it); the accumulators are bit-cast to floats around their MFMAs.
    constexpr int NST = NBW * 2 * KCB > (KS1 + 3) / 4 ? NBW * 2 * KCB : (KS1 + 3) / 4;
    typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
    u32x16 st[NST];
#pragma unroll
    for (int i = 0; i < NST; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) st[i][r] = 0u;
    auto wfrag = [&](int ks) -> u32x4 {
        const u32x4 v = { st[ks >> 2][4 * (ks & 3)], st[ks >> 2][4 * (ks & 3) + 1], st[ks >> 2][4 * (ks & 3) + 2], st[ks >> 2][4 * (ks & 3) + 3] };
        return v;
    };
    // ---- prologue (P1): W_l[:, block c] -> LDS [n][64] -> the P1 waves' A fragments of W^T (m = column, k = n), kept for the launch ----
    if constexpr (P1) {
        uint16_t* const sW = sD + 2 * DSZ;
        for (int e = tid; e < N * 8; e += BW_THREADS) {
            const int n = e >> 3, ch = e & 7;
            *reinterpret_cast<u32x4*>(sW + n * BW_KC + ((ch ^ swz_h(n)) << 3)) = *reinterpret_cast<const u32x4*>(a.w + (int64_t)n * a.ldw + BW_KC * c + 8 * ch);
        }
        __syncthreads();
        if (p1_wave) {
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                const u32x4 w = frag_tr<BW_KC, false>(sW, 32 * cb1, ks, lane);
#pragma unroll
                for (int e = 0; e < 4; e++) st[ks >> 2][4 * (ks & 3) + e] = w[e];
            }
        }
        __syncthreads();   // every wave has its fragments: buffer 2 may be overwritten by the DMA of tile 2
    }

    // ---- LDS-DMA of tile t into buffer t % 3: the image is lane-linear, so lane j of piece p fetches the chunk whose SWIZZLED place is 64 p + j.
    //      EVERY wave issues DPW + 1 instructions per tile (a wave without a piece of a narrow dZ tile fetches zeros into a spare KiB): the counted
    //      s_waitcnt vmcnt in the loop relies on it ----
    // Gathered h rows (layer 0): the source row of this lane's row of a tile comes from a.idx.  The entries of tile t + 3 travel by DMA into the wave's own 256 bytes
    // of sIdx[(t + 3) & 1] at the TOP of iteration t, IN FRONT of that iteration's tile DMAs, so the counted wait at the end of the iteration -- all but the last
    // VM_TILE operations -- covers them; iteration t + 1 reads them back with a ds_read the compiler can see (its own lgkmcnt wait) and issues tile t + 3.  EVERY
    // wave of a layer-0 instantiation issues the instruction, gathered or not (from the zeros when there is no index list): the counted waits rely on it.
    // Tiles 0 .. 2: plain loads in the prologue, where nothing is in flight yet.
    auto index_row = [&](int t) -> int64_t {
        int64_t r = tile_row(t) + ((64 * wave + lane) >> 3);
        return r >= a.rows ? a.rows - 1 : r;
    };
    const bool gathered = !P1 && a.idx != nullptr;
    // this lane's share of a tile's DMA, fixed for the launch: piece p = wave + 8 i of the dZ tile (lane j of piece p fetches the chunk whose swizzled place is 64 p + j)
    // and one piece of each h image -- row within the tile, and the byte offset from the tile's first row
    int d_row[DPW], h_row[HPW];
    uint32_t d_off[DPW], h_off[KCB][HPW];
#pragma unroll
    for (int i = 0; i < DPW; i++) {
        const int q = 64 * (mv + MW * i) + lane;
        d_row[i] = q / CPR;
        d_off[i] = (uint32_t)(d_row[i] * (int)a.ldd + 8 * ((q % CPR) ^ swz_d<CPR>(d_row[i]))) * 2u;
    }
#pragma unroll
    for (int j = 0; j < HPW; j++) {
        const int q = 64 * (mv + MW * j) + lane;
        h_row[j] = q >> 3;
#pragma unroll
        for (int k = 0; k < KCB; k++) h_off[k][j] = (uint32_t)(h_row[j] * (int)a.ldh + BW_KC * (KCB * c + k) + 8 * ((q & 7) ^ swz_h(h_row[j]))) * 2u;
    }
    const uint32_t lds0 = bw_lds_addr(bw_lds), spare0 = bw_lds_addr(sRed + 2 * BW_KC), idx0 = bw_lds_addr(sIdx) + 256u * wave;
    auto index_dma = [&](int t) {   // layer 0 only
        bw_glds4(gathered ? a.idx + index_row(t) : reinterpret_cast<const int32_t*>(a.zeros), idx0 + (uint32_t)(t & 1) * (BW_WAVES * 256u));
    };
    // One of the VM_TILE = DPW + KCB LDS-DMA instructions a wave issues for tile t (into ring buffer b = t % BW_RING): p < DPW = a KiB of the dZ tile, else an h image's.
    // They are issued ONE AT A TIME between the products of the tile loop (round 6): eight waves issuing their five at the top of the iteration queued on the CU's one
    // address path -- 40 KB at 64 B/clk -- and every wave sat 800 - 930 cycles of a 3 400-cycle iteration in front of its first product (in-kernel stamps, NOTES).
    auto issue_piece = [&](int t, int b, int hsrc, int p) {
#ifdef BW_ABL_NODMA   /* timing-only: no tile ever arrives */
        return;
#endif
        const int64_t r0 = tile_row(t);
        const bool whole = r0 + BW_ROWS <= a.rows;   // wave-uniform: every row of the tile exists (all tiles but the minibatch's last)
#ifndef BW_ABL
#define BW_ABL 0   /* timing-only ablations (wrong results): 1 = every tile's DMA reads the first rows (cache-resident), 2 = every result tile is stored over the first rows */
#endif
        const int64_t r0s = (BW_ABL & 1) ? (int64_t)(t & 1) * BW_ROWS : r0;
        if (p < DPW) {
            const int i = p < DPW ? p : 0;
            const uint32_t dD = lds0 + (uint32_t)(b * DSZ) * 2u;                          // byte addresses in LDS (wave-uniform)
            const uint16_t* const dbase = a.d + r0s * a.ldd;   // wave-uniform base: SGPR pair + the lane's 32-bit offset
            const int pc = mv + MW * i;
            const bool have = pc < CPR;
            const uint32_t dst = have ? dD + 1024u * pc : spare0 + 1024u * wave;
            if (whole && have) bw_glds16_s<(BW_NT & 2) != 0>(dbase, d_off[i], dst);
            else bw_glds16<(BW_NT & 2) != 0>(have && r0 + d_row[i] < a.rows ? reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(dbase) + d_off[i]) : a.zeros, dst);
        } else {
            const int hp = p - DPW;
            const int k = hp / HPW < KCB ? hp / HPW : 0, j = hp % HPW;   // image k (columns 64 (KCB c + k) ..), its piece mv + MW j
            const uint32_t dH = lds0 + (uint32_t)(BW_RING * DSZ + b * HSZ) * 2u;
            const uint16_t* const hbase = a.h + r0s * a.ldh;
            const uint32_t dst = dH + (uint32_t)(k * BW_ROWS * BW_KC) * 2u + 1024u * (mv + MW * j);
            if (whole && !gathered) bw_glds16_s<(BW_NT & 1) != 0>(hbase, h_off[k][j], dst);
            else {
                const uint16_t* src = gathered ? reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(a.h + (int64_t)hsrc * a.ldh) + (h_off[k][j] - (uint32_t)(h_row[j] * (int)a.ldh) * 2u))
                                               : reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(hbase) + h_off[k][j]);
                bw_glds16<(BW_NT & 1) != 0>(r0 + h_row[j] < a.rows ? src : a.zeros, dst);
            }
        }
    };
    auto issue = [&](int t, int b, int hsrc) {   // all of this mover's pieces of tile t
        if (!mover) return;
#pragma unroll
        for (int p = 0; p < DPW + KCB * HPW; p++) issue_piece(t, b, hsrc, p);
    };
    // the result tile of tile t leaves in 16-byte row pieces (rows past the minibatch are zeros: they keep the destination's padding zero)
    auto store_out = [&](int t) {
        const int64_t r0 = (BW_ABL & 2) ? (int64_t)(t & 1) * BW_ROWS : tile_row(t);
        const int row = tid >> 3, ch = tid & 7;
        const u32x4 piece = *reinterpret_cast<const u32x4*>(sO + (t & 1) * BW_ROWS * BW_KC + row * BW_KC + ((ch ^ swz_h(row)) << 3));
        if (BW_NT & 4) __builtin_nontemporal_store(piece, reinterpret_cast<u32x4*>(a.dz_out + (r0 + row) * a.ld_out + BW_KC * c + 8 * ch));
        else *reinterpret_cast<u32x4*>(a.dz_out + (r0 + row) * a.ld_out + BW_KC * c + 8 * ch) = piece;
    };
    // vector-memory operations a wave issues per iteration of the steady state: the DMA pieces of one tile, and (P1) one store of the result tile
    constexpr int VM_TILE = DPW + KCB * HPW;     // DMA instructions of one tile (per mover)
    constexpr int VM_PER_ITER = VM_TILE + (P1 ? 1 : 0);

    float csum[16];
#pragma unroll
    for (int r = 0; r < 16; r++) csum[r] = 0.0f;

    int hs_next = 0;   // gathered: this lane's source row in tile t + 2 when iteration t issues that tile
    {
        int h0 = 0, h1 = 0;
        if (gathered && n_tiles > 0) { h0 = a.idx[index_row(0)]; h1 = a.idx[index_row(1)]; hs_next = a.idx[index_row(2)]; }
        if (n_tiles > 0) issue(0, 0, h0);
        if (n_tiles > 1) issue(1, 1, h1);
    }
    // lane-constant LDS offsets of everything the tile loop reads (elements; see frag_off)
    int p1_rd[KS1], ep_at[4];          // P1: the row's 16-byte B fragments of dZ; the epilogue's 8-byte places in the h / result images
    int a_lo[NBW], a_hi[NBW], h_lo[2 * KCB], h_hi[2 * KCB];   // P2: transposing reads of dZ (A) and h (B) at k step 0
    {
        const int row = 32 * rb1 + li, fd = swz_d<CPR>(row), fh = swz_h(row);
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) p1_rd[ks] = row * N + (((2 * ks + kg) ^ fd) << 3);
#pragma unroll
        for (int q = 0; q < 4; q++) ep_at[q] = row * BW_KC + (((4 * cb1 + q) ^ fh) << 3) + 4 * kg;
#pragma unroll
        for (int i = 0; i < NBW; i++) frag_off<N, true>(32 * ((pj < 0 ? 0 : pj) * NBW + i), lane, a_lo[i], a_hi[i]);
#pragma unroll
        for (int h = 0; h < 2 * KCB; h++) { frag_off<BW_KC, false>(32 * (h & 1), lane, h_lo[h], h_hi[h]); h_lo[h] += (h >> 1) * BW_ROWS * BW_KC; h_hi[h] += (h >> 1) * BW_ROWS * BW_KC; }
    }
    if (!mover) { }   // (a wave that fetches nothing waits for nothing: the movers' waits and the barrier cover the tile)
    else if (n_tiles > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM_TILE) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 has landed
    bw_barrier();
#ifdef BW_STAMP   /* diagnostic build: cycle sums of the tile loop's phases for one P1 and one P2 wave of workgroup 0, printed by the device */
    unsigned long long st_sum[5] = { 0, 0, 0, 0, 0 }, st_t0 = 0, st_begin = __builtin_amdgcn_s_memtime();
#define BW_ST(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_sum[i] += now_ - st_t0; st_t0 = now_; } while (0)
#else
#define BW_ST(i) do { } while (0)
#endif
    int rb = 0;   // ring buffer of tile t (t % BW_RING, kept by rotation: no division in the loop)
    // The body of one iteration as a lambda, run TWICE per trip of the loop below: unrolled by two the layer-0 launch is 6 - 8 us faster (46.7 against 53 - 55 us: found
    // by accident when an exploration build needed two named register sets; the ring index and the parities become compile-time facts of each copy)
    auto body = [&](const int t) {
#ifdef BW_STAMP
        st_t0 = __builtin_amdgcn_s_memtime();
#endif
        const uint16_t* const tD = sD + rb * DSZ;
        const uint16_t* const tH = sH + rb * HSZ;
        const int rb2 = rb == 0 ? BW_RING - 1 : rb - 1;   // (t + 2) % 3
        if (t + 2 < n_tiles) {                      // into the buffer every wave finished reading before the barrier that ended iteration t - 1
            if constexpr (!P1) {
                // tile t + 2's entries landed under the wait that ended iteration t - 1 (tile 2's came with the prologue's loads)
                if (t > 0) hs_next = gathered ? sIdx[((t + 2) & 1) * (BW_WAVES * 64) + 64 * wave + lane] : 0;
                index_dma(t + 3);
            }
        }
        const bool pf = t + 2 < n_tiles;   // (uniform) this iteration issues tile t + 2's pieces, between its products: `pieces(lo, hi)` below
#ifndef BW_DMA_SPREAD
#define BW_DMA_SPREAD 0   /* 1: a tile's DMA instructions one at a time between the products (measured: the stall moves into the products, layer 0 gets slower); 0: all at the top */
#endif
#ifndef BW_STAGGER
#define BW_STAGGER 0   /* 1: beside P1 the P2 waves issue their pieces BEHIND their products, the P1 waves in front of theirs, so that a SIMD's two waves are blocked in the address path at different times -- measured 55.9 against 53.1 us: not taken */
#endif
        const bool late = BW_STAGGER && P1 && !p1_wave;   // (wave-uniform)
        if (!BW_DMA_SPREAD && pf && !late) issue(t + 2, rb2, hs_next);
        auto pieces = [&](int lo, int hi) {
            if (BW_DMA_SPREAD && pf) {
#pragma unroll
                for (int p = 0; p < VM_TILE; p++) if (mover && p >= lo && p < hi) issue_piece(t + 2, rb2, hs_next, p);
            }
        };
        BW_ST(0);   // (index entries of tile t + 3 requested)
        if constexpr (P1) { if (t > 0) store_out(t - 1); }
        BW_ST(1);   // result tile of t - 1 stored
        if (p1_wave) {
            // ---- P1: D[kcol][row] = sum_n W[n][kcol] dZ[row][n]; lane = row, registers = columns ----
            // The rows' fragments come from LDS a batch of k steps AHEAD of their products (read one by one in front of its MFMA, as the compiler laid the plain loop
            // out with a single register quad, every product waited out an LDS round trip: 2 100 cycles for 16 products; stamps, NOTES)
            constexpr int BT = KS1 >= 4 ? 4 : KS1, NBT = KS1 / BT;
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[r] = 0.0f;
            u32x4 bq[2][BT];
#pragma unroll
            for (int j = 0; j < BT; j++) bq[0][j] = *reinterpret_cast<const u32x4*>(tD + p1_rd[j]);
#pragma unroll
            for (int b = 0; b < NBT; b++) {
                if (b + 1 < NBT) {
#pragma unroll
                    for (int j = 0; j < BT; j++) bq[(b + 1) & 1][j] = *reinterpret_cast<const u32x4*>(tD + p1_rd[(b + 1) * BT + j]);
                }
                __builtin_amdgcn_sched_barrier(0);   // the next batch's reads are issued in front of this batch's products
#pragma unroll
                for (int j = 0; j < BT; j++) acc1 = bw_mfma(wfrag(b * BT + j), bq[b & 1][j], acc1);
                pieces(b * VM_TILE / NBT, (b + 1) * VM_TILE / NBT);
            }
            // epilogue: register r <-> column 32 cb1 + (r & 3) + 8 (r >> 2) + 4 kg of the block; tanh' from the staged h, result to its own image
            uint16_t* const tO = sO + (t & 1) * BW_ROWS * BW_KC;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint2 hv = *reinterpret_cast<const uint2*>(tH + ep_at[q]);
                const float h0 = bw_u2f(hv.x << 16), h1 = bw_u2f(hv.x & 0xffff0000u), h2 = bw_u2f(hv.y << 16), h3 = bw_u2f(hv.y & 0xffff0000u);
                const float v0 = acc1[4 * q] * (1.0f - h0 * h0), v1 = acc1[4 * q + 1] * (1.0f - h1 * h1);
                const float v2 = acc1[4 * q + 2] * (1.0f - h2 * h2), v3 = acc1[4 * q + 3] * (1.0f - h3 * h3);
                csum[4 * q] += v0; csum[4 * q + 1] += v1; csum[4 * q + 2] += v2; csum[4 * q + 3] += v3;
                *reinterpret_cast<uint2*>(tO + ep_at[q]) = make_uint2(bw_pack(v0, v1), bw_pack(v2, v3));
            }
        } else if (pj >= 0 && pj * NBW < NB) {
            // ---- P2: dW[n][kcol] += dZ^T h over the tile's 64 rows (4 k steps): the wave's n blocks against every 32-column half of the workgroup's columns ----
            // (the fragments of k step ks + 1 requested before the products of k step ks -- the same double buffering as P1's -- was measured and is not taken:
            // the hidden layers' launch did not move, layer 0's got 4 us slower)
            constexpr int KS2 = BW_ROWS / 16;
#pragma unroll
            for (int ks = 0; ks < KS2; ks++) {
                u32x4 bf[2 * KCB], af[NBW];
#pragma unroll
                for (int h = 0; h < 2 * KCB; h++) bf[h] = frag_at(tH + 16 * ks * BW_KC, h_lo[h], h_hi[h]);
#pragma unroll
                for (int i = 0; i < NBW; i++) af[i] = frag_at(tD + 16 * ks * N, a_lo[i], a_hi[i]);   // (a block past NB reads inside the tile and is not used)
#ifndef BW_P2_NOBARRIER
                __builtin_amdgcn_sched_barrier(0);   // every transposing read of the k step in front of its products (hipcc sinks each read to its product otherwise)
#endif
#pragma unroll
                for (int i = 0; i < NBW; i++) {
                    const int nb = pj * NBW + i;
                    if (nb < NB) {
#pragma unroll
                        for (int h = 0; h < 2 * KCB; h++)
                            st[2 * KCB * i + h] = __builtin_bit_cast(u32x16, bw_mfma(af[i], bf[h], __builtin_bit_cast(f32x16, st[2 * KCB * i + h])));
                    }
                }
                pieces(ks * VM_TILE / KS2, (ks + 1) * VM_TILE / KS2);
            }
        } else {
            pieces(0, VM_TILE);   // a wave without products (narrow layers) still issues its share of the DMA
        }
        if (!BW_DMA_SPREAD && pf && late) issue(t + 2, rb2, hs_next);
        BW_ST(2);   // products (+ P1's epilogue; the P2 waves' share of tile t + 2's DMA behind them)
        // this wave's pieces of tile t + 1 have landed: everything it issued up to them is done, i.e. all but what this iteration issued (the pieces of
        // tile t + 2 and the store of tile t - 1) -- counted, so that tile t + 2 stays in flight across the barrier (a plain __syncthreads() would drain
        // it: the compiler's fence waits for vmcnt(0)).  The last iterations issue less: they wait for everything.
        if (!mover) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave that fetches nothing: its result tile is in LDS; the movers' waits + the barrier cover the tiles
        } else
#ifdef BW_FULL_WAIT
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (false) {
#else
        if (t + 2 < n_tiles) {
#endif
            if (t > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_PER_ITER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(VM_TILE) : "memory");   // the first iteration has no result tile to store yet
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        rb = rb == BW_RING - 1 ? 0 : rb + 1;
        BW_ST(3);   // counted wait for tile t + 1
        bw_barrier();
        BW_ST(4);   // barrier
    };
    for (int t = 0; t < n_tiles; t += 2) {
        body(t);
        if (t + 1 < n_tiles) body(t + 1);
    }
#ifdef BW_STAMP
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && NB == 8)
        printf("BWST P1=%d wave %d tiles %d total %llu | issue %llu store %llu compute %llu wait %llu barrier %llu\n", (int)P1, wave, n_tiles,
               (unsigned long long)(__builtin_amdgcn_s_memtime() - st_begin), st_sum[0], st_sum[1], st_sum[2], st_sum[3], st_sum[4]);
#endif
    if constexpr (P1) { if (n_tiles > 0) store_out(n_tiles - 1); }

    // ---- the workgroup's slice of the weight gradient: register r of lane (i, kg) <-> n = 32 nb + (r & 3) + 8 (r >> 2) + 4 kg, column = block's 32 half + i ----
    float* const slab = a.slab + (int64_t)z * a.slab_stride;
    if (!p1_wave && pj >= 0) {
#pragma unroll
        for (int i = 0; i < NBW; i++) {
            const int nb = pj * NBW + i;
            if (nb < NB) {
#pragma unroll
                for (int h = 0; h < 2 * KCB; h++) {
                    const int kcol = BW_KC * KCB * c + 32 * h + li;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (n < a.n_real && kcol < a.k_real) slab[(int64_t)n * a.k_real + kcol] = bw_u2f(st[2 * KCB * i + h][r]);
                    }
                }
            }
        }
    }
    if constexpr (P1) {
        // column sums of dZ_{l-1} over the range's rows: lanes (= rows) of a half first, then the two row blocks in fixed order
        if (p1_wave) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = csum[r];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
                csum[r] = v;
            }
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) sRed[rb1 * BW_KC + 32 * cb1 + (r & 3) + 8 * (r >> 2) + 4 * kg] = csum[r];
            }
        }
        __syncthreads();
        if (tid < BW_KC) a.colsum[(int64_t)z * a.ld_cs + BW_KC * c + tid] = sRed[tid] + sRed[BW_KC + tid];
    }
}

template <int NB, bool P1, int KCB>
constexpr size_t bwd_lds_bytes() {   // ring of dZ and h tiles, (P1) two result tiles, the column-sum hand-over, a spare KiB per wave for DMA pieces a narrow tile does not have
    return (BW_RING * ((size_t)BW_ROWS * (32 * NB < 64 ? 64 : 32 * NB) + (size_t)KCB * BW_ROWS * BW_KC) + (P1 ? 2 : 0) * (size_t)BW_ROWS * BW_KC) * sizeof(uint16_t) +
           2 * BW_KC * sizeof(float) + (size_t)BW_WAVES * 1024 + (P1 ? 0 : 2 * (size_t)BW_WAVES * 256);   // layer 0: + the ring of index-list entries
}

template <int NB, bool P1, int KCB>
hipError_t bwd_launch(const BwdPair& pp, unsigned blocks, hipStream_t s) {
    constexpr size_t lds = bwd_lds_bytes<NB, P1, KCB>();
    static_assert(lds <= 160 * 1024, "one workgroup per CU must fit");
    auto kern = bwd_layer_kernel<NB, P1, KCB>;
    if constexpr (lds > 64 * 1024) {
        static std::atomic<unsigned long long> lds_ok{0};
        const hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BW_THREADS), lds, s, pp);
    return hipGetLastError();
}
inline unsigned bwd_blocks(const BwdArgs& a) { return (unsigned)((a.S + 7) / 8 * 8 * a.CB); }

}  // namespace

// Column blocks a launch cuts the layer's input side into: 64 columns beside P1; 128 for layer 0 (no layer below) when its width allows
int gen_bwd_col_blocks(int ld_in, bool has_below) { return (!has_below && ld_in % (2 * BW_KC) == 0) ? ld_in / (2 * BW_KC) : ld_in / BW_KC; }

// Row ranges of a layer's fused backward launch: S ranges of whole 64-row tiles, S x (column blocks) workgroups ~ one per CU -- or per CU of HALF the chip when
// the other net's launch runs beside it on a second stream (a workgroup holds a CU by its LDS): half the partial slabs, written and read again, for the same work.
int gen_bwd_ranges(int64_t rows, int col_blocks, bool half_chip, int* tiles_per_range) {
    const int64_t tiles = (rows + BW_ROWS - 1) / BW_ROWS;
    int64_t want = (half_chip ? 128 : 256) / col_blocks;
    // ranges go round-robin over the 8 XCDs (the kernel's block order) and a workgroup holds a CU: a count that is not a multiple of 8 gives XCD 0 one range
    // more than the others -- with three column blocks and both nets in the launch 36 workgroups for its 32 CUs, and the four that wait for a CU double the
    // launch's duration (configs[4] layer 0: 42 ranges -> 97 us for the pair, 40 ranges -> see NOTES)
    if (want >= 8) want = want / 8 * 8;
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    const int64_t per = (tiles + want - 1) / want;
    *tiles_per_range = (int)per;
    return (int)((tiles + per - 1) / per);
}

// dZ widths the kernel is instantiated for: a head (<= 32 logits in a 128-pitch buffer) or a hidden vector padded to 128 or 256; inputs in 64-column blocks
bool gen_fused_backward_ok(const GenericCtx& g) {
    return g.bf16 && (g.ld_h == 128 || g.ld_h == 256) && g.L.act <= 32 && g.ld_in0 % BW_KC == 0 && g.L.n_hidden >= 1;
}

// One layer: d = dZ_l [., ldd] (width n_pad in {32, 128, 256}), h = the layer's input [., ldh], w = the layer's bf16 weight plane (null: layer 0, no layer
// below).  col_blocks from gen_bwd_col_blocks, S / tiles_per_range from gen_bwd_ranges.  Writes S slabs [n_real][k_real], dZ_{l-1} and its column sums per range.
// `other` != nullptr: the same layer of the other net rides in the same launch (same n_pad, column blocks and kind of layer; its own operands and ranges).
namespace {
BwdArgs bwd_args(const GenBwdLayer& q, int64_t rows, int col_blocks, const uint16_t* zeros) {
    BwdArgs a{};
    a.zeros = zeros; a.idx = q.idx;
    a.d = q.d; a.ldd = q.ldd; a.h = q.h; a.ldh = q.ldh; a.w = q.w; a.ldw = q.ldw; a.dz_out = q.dz_out; a.ld_out = q.ld_out; a.slab = q.slab; a.slab_stride = q.slab_stride;
    a.colsum = q.colsum; a.ld_cs = q.ld_cs; a.rows = rows; a.n_real = q.n_real; a.k_real = q.k_real; a.S = q.S; a.CB = col_blocks; a.tiles_per_range = q.tiles_per_range;
    return a;
}
}  // namespace
hipError_t gen_fused_backward_layer(int n_pad, const GenBwdLayer& one, const GenBwdLayer* other, int64_t rows, int col_blocks, const uint16_t* zeros, hipStream_t s) {
    if (one.idx && one.w) return hipErrorInvalidValue;   // gathered inputs exist for layer 0 only
    if (other && ((other->w != nullptr) != (one.w != nullptr) || other->ldh != one.ldh)) return hipErrorInvalidValue;
    BwdPair pp{};
    pp.a[0] = bwd_args(one, rows, col_blocks, zeros);
    unsigned blocks = bwd_blocks(pp.a[0]);
    pp.split = (int)blocks;
    if (other) { pp.a[1] = bwd_args(*other, rows, col_blocks, zeros); blocks += bwd_blocks(pp.a[1]); }
    const bool p1 = one.w != nullptr;
    const bool wide = !p1 && col_blocks * 2 * BW_KC == one.ldh;   // 128-column blocks
    if (!p1 && !wide && col_blocks * BW_KC != one.ldh) return hipErrorInvalidValue;
    if (n_pad == 32) return p1 ? bwd_launch<1, true, 1>(pp, blocks, s) : (wide ? bwd_launch<1, false, 2>(pp, blocks, s) : bwd_launch<1, false, 1>(pp, blocks, s));
    if (n_pad == 128) return p1 ? bwd_launch<4, true, 1>(pp, blocks, s) : (wide ? bwd_launch<4, false, 2>(pp, blocks, s) : bwd_launch<4, false, 1>(pp, blocks, s));
    if (n_pad == 256) return p1 ? bwd_launch<8, true, 1>(pp, blocks, s) : (wide ? bwd_launch<8, false, 2>(pp, blocks, s) : bwd_launch<8, false, 1>(pp, blocks, s));
    return hipErrorNotSupported;
}
