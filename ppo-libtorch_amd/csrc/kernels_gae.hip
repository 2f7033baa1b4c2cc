// ppo-libtorch_amd/csrc/kernels_gae.hip -- K4: GAE advantage / return scan over time-major [T, N] buffers (gfx950).
//
// reference PPO_Discrete::calcAdvantage, PPO/PPO_Discrete.cpp:274-331:
//     nnt_t   = 1 - dones[t+1]                (t == T-1: 1 - next_done)
//     delta_t = (r_t + (gamma * v_{t+1}) * nnt_t) - v_t
//     A_t     = delta_t + ((gamma*lambda) * nnt_t) * A_{t+1}
//     R_t     = A_t + v_t
//
// EXACT mode (what training uses): bit-identical to the reference needs its evaluation order along t and separately
// rounded mul/add (the library is built with -ffp-contract=off).  What is parallel without changing a single rounding:
//   (i)  envs are independent                      -> a workgroup owns a strip of EPB env columns;
//   (ii) delta_t and c_t = (gamma*lambda)*nnt_t are element-wise -> ALL 256 threads compute them while streaming
//        r, v, dones with coalesced 16-byte loads, and park them in an LDS tile of TC time steps;
//   (iii) only the 2-op recurrence A_t = delta_t + c_t * A_{t+1} is serial: EPB lanes of wave 0 walk the tile top-down
//        out of LDS (one env per lane, carry in a register across tiles), writing A_t back in place;
//   (iv) all threads then stream A_t and R_t = A_t + v_t to HBM with coalesced 16-byte stores.
// A done flag cuts the chain (c_t = 0 => A_t = delta_t exactly): that is the "segmented" part; it needs no special
// handling in the serial walk and costs nothing.
// Algorithmic HBM traffic: 12 B read + 8 B written per (t, n) element, + 8 B per env (next_value, next_done).
// Round 2's kernel moved 2.0x the algorithmic READS past the L2 (PMC: 12.7 MB against 6.3 MB at 4 096 envs).  The larger half of that was NOT the
// second request for v_{t+1}: a 16-column strip reads 64-byte row pieces, half of a 128-byte memory line, and its neighbour strip, which owns the
// other half, was the NEXT workgroup -- on the next XCD, behind another L2 -- so every line was fetched twice from the memory side.  Workgroups are
// now mapped to strips so that the two halves of a line are read on the same XCD (strip_of_block below).
// v_{t+1} is still requested by the thread that forms delta_t although a neighbouring lane requests the same row as its v_t: that request is
// served by L1 / L2.  Three ways of avoiding it were built and measured, each A/B in one call (in-trace us at 4 096 / 8 192 / 32 768 envs;
// shipped: 5.0 / 6.25 / 18.55): handing row t + 1 over between lanes with ds_bpermute (memory-side reads 1.00x of algorithmic: profiles/
// r03_v1_gae_traffic.json) 5.3 / 7.1 / 19.1; a thread owning 2 or 4 CONSECUTIVE rows of a float4 column and fetching v of rows r .. r + ITERS
// once 5.05 / 6.8 / 19.0; round 2's hand-over through LDS behind one more barrier 5.6 / 6.9 / 21.2.  The duplicate request is the cheapest.
// Folding the update kernel's record pack (pack_records_kernel: it consumes exactly the advantages and returns the store phase holds) into
// this kernel was also built and measured in round 2: ONE launch of 18.2 us (21.6 in its first form) against 5.1 + 10.5 us for the two --
// the pack is a job for B = T N threads, and inside the scan it has N / 16 workgroups of four waves to run on.  Not shipped.
// Nor is a walk whose next 16 rows are pulled out of LDS (pinned by sched_barrier) before the current 16-step chain runs: 5.18 against 4.91 us
// at 4 096 envs, 6.42 / 6.25, 19.06 / 18.6 -- the walk is bound by its 256 dependent mul / add per env, not by the LDS hand-overs.
//
// FAST mode (ppo_gae_fast; NOT what training uses -- bit-exactness is the bar): the recurrence as a segmented scan of affine maps.  A_t = delta_t + c_t A_{t+1} is
// the map f_t(x) = delta_t + c_t x applied to A_{t+1}; maps compose associatively, (c, d) o (c', d') = (c c', d + c d'), and a done flag (c_t = 0) cuts
// the segment by itself.  Phase B then is: every one of the 256 threads composes the maps of ITS chunk of rows of one env column (256 / EPB chunks per
// column), the chunk maps meet in LDS, each thread runs the maps of the chunks above its own over the tile's carry (<= 15 steps) and replays its chunk
// from that value.  The longest dependent chain falls from 2 x 128 operations on EPB lanes to 2 x (8 + 15 + 8) on all 256 -- and the answer is no longer
// the reference's bit pattern: only the carry into a chunk is associated differently (inside a chunk the replay IS the reference's order), measured
// <= 6 ULP of the largest |A| the chain has carried, 8 - 40 % of the elements differ at all (tests/test_gpu_parity.py::test_gae_fast_mode_stays_within_ulps,
// profiles/r03_v4_gae_fast_report.jsonl).  What it buys: nothing at the sizes of BASELINE.json (4.9 / 6.0 / 18.8 us against 4.9 / 6.3 / 18.7 at 4096 / 8192 /
// 32768 envs x 128 steps) -- the scan is launch + one memory round trip, not its chain; 16 % at T = 2048 x 32 envs, where the chain is long and the grid one workgroup.
#include <cstdlib>

#include "ppo_internal.hpp"

namespace {

constexpr int GAE_TC = 128;       // time steps per LDS tile
constexpr int GAE_THREADS = 256;
constexpr int GAE_WALK = 16;      // rows the serial walk pulls into registers at a time

// Workgroup -> strip.  Workgroups are dealt to the 8 XCDs round-robin by index (each XCD has its own L2).  A strip narrower than a 128-byte
// memory line (16 columns = 64 bytes per row) shares every line with its neighbour strip: pair the two on ONE XCD -- workgroup b runs on
// XCD b % 8 as its (b / 8)-th workgroup; consecutive workgroups of an XCD take the two halves of a pair.  Needs the grid to be a multiple of 16.
template <int EPB>
__device__ __forceinline__ int strip_of_block(int b, int grid) {
    if (EPB * 4 >= 128 || (grid & 15) != 0) return b;
    const int xcd = b & 7, k = b >> 3;
    return (((k >> 1) << 3) + xcd) * 2 + (k & 1);
}

#ifndef GAE_KEEPV_ALL
#define GAE_KEEPV_ALL false   /* -DGAE_KEEPV_ALL=true: v_t in registers in the 16- / 32-column kernels of the smaller sizes too (A/B) */
#endif
#ifndef GAE_NO_PIPELINE
#define GAE_NO_PIPELINE 0   /* -DGAE_NO_PIPELINE=1: the walk without software-pipelined LDS reads (A/B) */
#endif
// MODE 0: GAE.  MODE 1: n-step returns (PPO_Discrete.cpp:309-329: ret_t = r_t + (gamma*nnt_t)*ret_{t+1}; adv = ret - v).
// TC: time steps per LDS tile (GAE_TC; 64 / 32 for the wide strips of HBM-resident sizes, launch_scan).  KEEPV (VEC only): v_t stays in the
// registers of the thread that loaded it until the same thread stores R_t = A_t + v_t -- no third LDS tile (EPB = 32, TC = 128: 32 KB instead of 48).
typedef float gae_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void gae_st4(float* p, const float4 v) {
    if (NT) { gae_f4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<gae_f4*>(p)); }
    else *reinterpret_cast<float4*>(p) = v;
}
// NT: A_t and R_t leave by non-temporal stores (HBM-resident sizes: the outputs do not evict the inputs from the memory-side cache).
template <int EPB, int MODE, bool VEC, bool FAST = false, int TC = GAE_TC, bool KEEPV = false, bool NT = false>
__global__ __launch_bounds__(GAE_THREADS) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                           const float* __restrict__ dones, const float* __restrict__ next_value,
                                                           const int32_t* __restrict__ next_done, int T, int N, float gamma,
                                                           float gae_lambda, float* __restrict__ adv, float* __restrict__ ret) {
    // sA: delta_t (MODE 0) / r_t (MODE 1) on input of the walk, A_t / ret_t on output.  sC: chain coefficient.  sV: v_t.
    static_assert(!KEEPV || (VEC && !FAST), "v_t in registers: the vector path of the exact mode only");
    static_assert(TC % GAE_WALK == 0 && (!VEC || (TC * (EPB / 4)) % GAE_THREADS == 0), "whole walk chunks, whole float4 slots per thread");
    __shared__ __attribute__((aligned(16))) float sA[TC * EPB];
    __shared__ __attribute__((aligned(16))) float sC[TC * EPB];
    __shared__ __attribute__((aligned(16))) float sV[KEEPV ? 4 : TC * EPB];
    const int tid = threadIdx.x;
    const int n0 = strip_of_block<EPB>((int)blockIdx.x, (int)gridDim.x) * EPB;
    const float gl = gamma * gae_lambda;  // the C++ float product of PPO_Discrete.cpp:301

    float carry = 0.0f;                                   // A_{t+1} (MODE 0) / ret_{t+1} (MODE 1) entering the tile
    const bool walker = tid < EPB && (n0 + tid) < N;
    if (MODE == 1 && walker) carry = next_value[n0 + tid];  // next_return = next_value at t = T-1 (:318)
    // FAST: chunk maps [chunk][column] and the tile's carry per column (two copies: the one a tile reads, the one it leaves)
    constexpr int NCH = GAE_THREADS / EPB, CH = TC / NCH > 0 ? TC / NCH : 1;
    __shared__ float sCk[FAST ? NCH * EPB : 1], sDk[FAST ? NCH * EPB : 1], sCarry[FAST ? 2 * EPB : 1];
    int tile_par = 0;
    if (FAST) {
        if (tid < EPB) sCarry[tid] = carry;
        __syncthreads();
    }

    constexpr int KV = (KEEPV && VEC) ? TC * (EPB / 4) / GAE_THREADS : 1;
    float4 keepv[KV];                                      // KEEPV: v_t of this thread's float4 slots, from phase A to phase C
    for (int t_hi = T; t_hi > 0; t_hi -= TC) {            // tile covers rows [t_lo, t_hi)
        const int t_lo = t_hi > TC ? t_hi - TC : 0;
        const int rows = t_hi - t_lo;

        // ---- phase A: stream r_t, v_t, v_{t+1}, dones_{t+1}; delta_t and c_t are element-wise (every rounding as the
        //      reference's tensor expression) and go straight to LDS.  v_{t+1} is the row a neighbouring thread loads as v_t:
        //      it is served by L1/L2, not HBM. ----
        if (VEC) {
            constexpr int C4 = EPB / 4;                    // float4 columns per row
            constexpr int ITERS = TC * C4 / GAE_THREADS;
            // every load of the tile is issued before the first use: one memory round trip per tile, not one per row group
            float4 rw[ITERS], vv[ITERS], nvv[ITERS], dd[ITERS];
#pragma unroll
            for (int i = 0; i < ITERS; i++) {
                const int e = tid + i * GAE_THREADS;
                const int r = e / C4, c = (e % C4) * 4;
                const int t = t_lo + r;
                if (r < rows) {
                    const size_t g = (size_t)t * N + n0 + c;
                    rw[i] = *reinterpret_cast<const float4*>(rewards + g);
                    vv[i] = *reinterpret_cast<const float4*>(values + g);
                    if (t + 1 < T) {
                        nvv[i] = *reinterpret_cast<const float4*>(values + g + N);   // served by L1 / L2: a neighbouring lane requests the same row as its v_t
                        dd[i] = *reinterpret_cast<const float4*>(dones + g + N);
                    } else {
                        nvv[i] = *reinterpret_cast<const float4*>(next_value + n0 + c);
                        const int4 d = *reinterpret_cast<const int4*>(next_done + n0 + c);
                        dd[i] = make_float4((float)d.x, (float)d.y, (float)d.z, (float)d.w);   // 0/1: exact, 1 - d below is the same value
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < ITERS; i++) {
                const int e = tid + i * GAE_THREADS;
                const int r = e / C4, c = (e % C4) * 4;
                if (r < rows) {
                    const float4 nnt = make_float4(1.0f - dd[i].x, 1.0f - dd[i].y, 1.0f - dd[i].z, 1.0f - dd[i].w);
                    const float4 nv = nvv[i], v = vv[i];
                    float4 a4, c4;
                    if (MODE == 0) {
                        a4 = make_float4((rw[i].x + (gamma * nv.x) * nnt.x) - v.x, (rw[i].y + (gamma * nv.y) * nnt.y) - v.y,   // :300
                                         (rw[i].z + (gamma * nv.z) * nnt.z) - v.z, (rw[i].w + (gamma * nv.w) * nnt.w) - v.w);
                        c4 = make_float4(gl * nnt.x, gl * nnt.y, gl * nnt.z, gl * nnt.w);                                     // :301
                    } else {
                        a4 = rw[i];
                        c4 = make_float4(gamma * nnt.x, gamma * nnt.y, gamma * nnt.z, gamma * nnt.w);                         // :324
                    }
                    *reinterpret_cast<float4*>(&sA[r * EPB + c]) = a4;
                    *reinterpret_cast<float4*>(&sC[r * EPB + c]) = c4;
                    if (KEEPV) keepv[KEEPV ? i : 0] = v;
                    else *reinterpret_cast<float4*>(&sV[r * EPB + c]) = v;
                }
            }
        } else {
            for (int e = tid; e < rows * EPB; e += GAE_THREADS) {
                const int r = e / EPB, c = e % EPB;
                const int t = t_lo + r;
                float a1 = 0.0f, c1 = 0.0f, v = 0.0f;
                if (n0 + c < N) {
                    const size_t g = (size_t)t * N + n0 + c;
                    const float rw = rewards[g];
                    v = values[g];
                    const float nv = (t + 1 < T) ? values[g + N] : next_value[n0 + c];
                    const float nnt = (t + 1 < T) ? 1.0f - dones[g + N] : (float)(1 - next_done[n0 + c]);
                    if (MODE == 0) { a1 = (rw + (gamma * nv) * nnt) - v; c1 = gl * nnt; }
                    else { a1 = rw; c1 = gamma * nnt; }
                }
                sA[e] = a1;
                sC[e] = c1;
                sV[e] = v;
            }
        }
        __syncthreads();

        // ---- phase B: the serial 2-op chain A_t = delta_t + c_t * A_{t+1}, one env per lane of wave 0, top row first.
        //      GAE_WALK rows are pulled into registers at a time so the LDS reads are in flight together and only the
        //      mul/add chain itself is serial. ----
        if constexpr (FAST) {
            const int k = tid / EPB, c = tid % EPB;         // chunk k of column c: tile rows [k CH, (k + 1) CH), walked from the top
            const int r_hi = (k + 1) * CH < rows ? (k + 1) * CH : rows, r_lo = k * CH < rows ? k * CH : rows;
            float d[CH], cc[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int r = (k + 1) * CH - 1 - i;
                const bool in = r < rows;
                d[i] = in ? sA[r * EPB + c] : 0.0f;         // rows past a ragged tile: the identity map
                cc[i] = in ? sC[r * EPB + c] : 1.0f;
            }
            float Ck = 1.0f, Dk = 0.0f;                     // this chunk as ONE map: A(r_lo) = Dk + Ck A(r_hi)
#pragma unroll
            for (int i = 0; i < CH; i++) { Dk = d[i] + cc[i] * Dk; Ck = cc[i] * Ck; }
            sCk[k * EPB + c] = Ck;
            sDk[k * EPB + c] = Dk;
            __syncthreads();
            float x = sCarry[tile_par * EPB + c];
            for (int j = NCH - 1; j > k; j--) x = sDk[j * EPB + c] + sCk[j * EPB + c] * x;   // the value entering this chunk
#pragma unroll
            for (int i = 0; i < CH; i++) {                  // replay: inside the chunk this is the reference's own order
                x = d[i] + cc[i] * x;
                const int r = (k + 1) * CH - 1 - i;
                if (r < rows) sA[r * EPB + c] = x;
            }
            if (k == 0) sCarry[(tile_par ^ 1) * EPB + c] = x;   // A of the tile's first row: the carry of the next (earlier) tile
            tile_par ^= 1;
            (void)r_hi; (void)r_lo;
        } else if (walker && rows == TC && !GAE_NO_PIPELINE) {
            // full tile: the chunks' LDS reads are software-pipelined -- chunk k + 1's 32 reads are in flight while chunk k's 16-step chain runs, so
            // only the chain itself (2 dependent operations per row) and the first chunk's read latency are serial
            float last = carry;
            float d[2][GAE_WALK], cc[2][GAE_WALK];
#pragma unroll
            for (int i = 0; i < GAE_WALK; i++) { d[0][i] = sA[(TC - 1 - i) * EPB + tid]; cc[0][i] = sC[(TC - 1 - i) * EPB + tid]; }
#pragma unroll
            for (int k = 0; k < TC / GAE_WALK; k++) {
                const int cur = k & 1, nxt = cur ^ 1, r = TC - k * GAE_WALK;
                if (k + 1 < TC / GAE_WALK) {
#pragma unroll
                    for (int i = 0; i < GAE_WALK; i++) { d[nxt][i] = sA[(r - GAE_WALK - 1 - i) * EPB + tid]; cc[nxt][i] = sC[(r - GAE_WALK - 1 - i) * EPB + tid]; }
                }
#pragma unroll
                for (int i = 0; i < GAE_WALK; i++) {
                    last = d[cur][i] + cc[cur][i] * last;
                    d[cur][i] = last;
                }
#pragma unroll
                for (int i = 0; i < GAE_WALK; i++) sA[(r - 1 - i) * EPB + tid] = d[cur][i];
            }
            carry = last;
        } else if (walker) {
            float last = carry;
            int r = rows;
            // full chunks: unconditional code (16 + 16 LDS reads in flight, then the 16-step mul/add chain, then 16 writes);
            // a per-element guard here costs more than the chain itself (12 us -> 4.9 us at 4096 x 128)
            while (r >= GAE_WALK) {
                float d[GAE_WALK], cc[GAE_WALK];
#pragma unroll
                for (int i = 0; i < GAE_WALK; i++) {
                    d[i] = sA[(r - 1 - i) * EPB + tid];
                    cc[i] = sC[(r - 1 - i) * EPB + tid];
                }
#pragma unroll
                for (int i = 0; i < GAE_WALK; i++) {
                    last = d[i] + cc[i] * last;
                    d[i] = last;
                }
#pragma unroll
                for (int i = 0; i < GAE_WALK; i++) sA[(r - 1 - i) * EPB + tid] = d[i];
                r -= GAE_WALK;
            }
            for (; r > 0; r--) {   // ragged tail (T not a multiple of 16)
                last = sA[(r - 1) * EPB + tid] + sC[(r - 1) * EPB + tid] * last;
                sA[(r - 1) * EPB + tid] = last;
            }
            carry = last;
        }
        __syncthreads();

        // ---- phase C: stream out advantages and returns ----
        if (VEC) {
            constexpr int C4 = EPB / 4;
            constexpr int ITERS = TC * C4 / GAE_THREADS;
#pragma unroll
            for (int i = 0; i < ITERS; i++) {              // the same slots as phase A: slot i's v_t may still be in this thread's registers
                const int e = tid + i * GAE_THREADS;
                const int r = e / C4, c = (e % C4) * 4;
                if (r >= rows) continue;
                const float4 a4 = *reinterpret_cast<const float4*>(&sA[r * EPB + c]);
                const float4 v4 = KEEPV ? keepv[KEEPV ? i : 0] : *reinterpret_cast<const float4*>(&sV[r * EPB + c]);
                float4 o_adv, o_ret;
                if (MODE == 0) {
                    o_adv = a4;
                    o_ret = make_float4(a4.x + v4.x, a4.y + v4.y, a4.z + v4.z, a4.w + v4.w);   // :305
                } else {
                    o_ret = a4;
                    o_adv = make_float4(a4.x - v4.x, a4.y - v4.y, a4.z - v4.z, a4.w - v4.w);   // :327
                }
                const size_t g = (size_t)(t_lo + r) * N + n0 + c;
                gae_st4<NT>(adv + g, o_adv);
                gae_st4<NT>(ret + g, o_ret);
            }
        } else {
            for (int e = tid; e < rows * EPB; e += GAE_THREADS) {
                const int r = e / EPB, c = e % EPB;
                if (n0 + c >= N) continue;
                const float a1 = sA[e], v1 = sV[e];
                const size_t g = (size_t)(t_lo + r) * N + n0 + c;
                if (MODE == 0) { adv[g] = a1; ret[g] = a1 + v1; }
                else { ret[g] = a1; adv[g] = a1 - v1; }
            }
        }
        __syncthreads();  // LDS tile is reused by the next (earlier) time tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The exact scan PIPELINED IN TIME (round 5).  gae_kernel above runs three phases behind two barriers -- every load, then the walk, then every store -- so the
// memory side sees 6.3 MB of reads, then 0.5 us of nothing, then 4.2 MB of writes, and at BASELINE's sizes the launch is one round trip of each after the
// other (4.95 us in trace at 4096 envs against 3.36 for a plain stream of the same bytes).  Here the 128-row tile is cut into four groups of 32 rows, latest
// time first -- the order the recurrence needs them in:
//   waves 0-3 (movers)  issue ALL their loads up front, group 3 first; as a group's loads return (they return in issue order) form delta_t, c_t for it, park
//                       them in LDS and count the group ready; later, when the walker has counted the group walked, stream its A_t and R_t = A_t + v_t out;
//   wave 4 (walker)     one env per lane as before; starts on group 3 as soon as that group is ready, while groups 2 .. 0 are still in flight, and hands every
//                       finished group back at once -- so the stores of group 3 leave while group 2 is walked and groups 1, 0 arrive.
// No barrier anywhere: LDS event counters (one wave's LDS operations execute in issue order: a count behind data writes needs no wait), polled with
// s_sleep.  Every wait is on a wave of the same workgroup that never waits on anything but memory or an earlier group, so the waits cannot cycle; they are
// bounded anyway.  Arithmetic, association order and the -ffp-contract=off build are gae_kernel's: the results are the same bits (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int GP_THREADS = GAE_THREADS + 64;                 // four mover waves + the walker
#ifndef GAE_PIPE_MIN_ENVS
#define GAE_PIPE_MIN_ENVS 4096
#endif
#ifndef GAE_PIPE_MAX_BLOCKS
#define GAE_PIPE_MAX_BLOCKS 256                               /* strips up to which the pipelined kernel is launched: one workgroup per CU (12 288 envs x 32 columns = 384: 9.0 - 9.3 us against 7.2) */
#endif
typedef __attribute__((address_space(3))) volatile uint32_t gp_flag_t;
// Bounded wait.  false = the count never came (cannot happen by construction, above): the caller poisons what it would have produced with NaN and raises the
// context's error word where there is one -- never a silently wrong advantage (mg_wait_ge of the update kernel does the same).
__device__ __forceinline__ bool gp_wait_ge(gp_flag_t* flag, uint32_t want) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 22); spin++) {
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*flag) >= want) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return ok;
}
template <int EPB, int MODE, int GP_GROUPS>
__global__ __launch_bounds__(GP_THREADS) void gae_pipe_kernel(const float* __restrict__ rewards, const float* __restrict__ values, const float* __restrict__ dones,
                                                              const float* __restrict__ next_value, const int32_t* __restrict__ next_done, int T, int N, float gamma,
                                                              float gae_lambda, float* __restrict__ adv, float* __restrict__ ret, int32_t* __restrict__ error_flag) {
    static_assert(EPB == 16 || EPB == 32, "strips of 16 or 32 columns");
    constexpr int GP_GROWS = GAE_TC / GP_GROUPS;       // rows of a group: 32 (four groups) or 64 (two)
    constexpr int C4 = EPB / 4;                        // float4 columns per row
    constexpr int NG = GAE_TC * C4 / GAE_THREADS;      // float4 slots of a mover thread: 2 (16 columns) or 4 (32); slot j = rows tid / C4 + (256 / C4) j: a wave's
                                                       // slot j lies inside ONE group (16 or 8 consecutive rows)
    constexpr int MOVERS_PER_GROUP = GAE_THREADS / 64 * NG / GP_GROUPS;   // (wave, slot) pairs that fill a group: what its ready count reaches per tile
    __shared__ __attribute__((aligned(16))) float sA[GAE_TC * EPB];
    __shared__ __attribute__((aligned(16))) float sC[GAE_TC * EPB];
    __shared__ uint32_t s_ready[GP_GROUPS], s_done[GP_GROUPS];   // (v_t stays in the mover's registers: no third tile)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = strip_of_block<EPB>((int)blockIdx.x, (int)gridDim.x) * EPB;
    const float gl = gamma * gae_lambda;   // the C++ float product of PPO_Discrete.cpp:301
    if (tid < GP_GROUPS) { s_ready[tid] = 0u; s_done[tid] = 0u; }
    __syncthreads();   // the only barrier: the counters exist
    gp_flag_t* const f_ready = (gp_flag_t*)s_ready;
    gp_flag_t* const f_done = (gp_flag_t*)s_done;
    const int n_tiles = T / GAE_TC;

    if (wave == GAE_THREADS / 64) {
        // ======================================================= the walker =======================================================
        const bool walker = lane < EPB && n0 + lane < N;
        float last = 0.0f;
        if (MODE == 1 && walker) last = next_value[n0 + lane];   // next_return = next_value at t = T-1 (:318)
        for (int tile = 0; tile < n_tiles; tile++) {
#pragma unroll
            for (int gi = 0; gi < GP_GROUPS; gi++) {
                const int g = GP_GROUPS - 1 - gi;
                if (!gp_wait_ge(f_ready + g, (uint32_t)(MOVERS_PER_GROUP * (tile + 1)))) {
                    last = __builtin_nanf("");   // the group never landed: everything below it in this strip is NaN, and the error word says why
                    if (lane == 0 && error_flag) atomicOr(error_flag, PPO_ERRFLAG_GAE_PROTOCOL);
                }
                if (walker) {
                    // the group's rows, top row first, 16 at a time: chunk k + 1's reads are in flight while chunk k's chain runs
                    constexpr int NCHUNK = GP_GROWS / GAE_WALK;
                    float d[2][GAE_WALK], cc[2][GAE_WALK];
                    const int top = (g + 1) * GP_GROWS;
#pragma unroll
                    for (int i = 0; i < GAE_WALK; i++) { d[0][i] = sA[(top - 1 - i) * EPB + lane]; cc[0][i] = sC[(top - 1 - i) * EPB + lane]; }
#pragma unroll
                    for (int k = 0; k < NCHUNK; k++) {
                        const int cur = k & 1, nxt = cur ^ 1, r = top - k * GAE_WALK;
                        if (k + 1 < NCHUNK) {
#pragma unroll
                            for (int i = 0; i < GAE_WALK; i++) { d[nxt][i] = sA[(r - GAE_WALK - 1 - i) * EPB + lane]; cc[nxt][i] = sC[(r - GAE_WALK - 1 - i) * EPB + lane]; }
                        }
#pragma unroll
                        for (int i = 0; i < GAE_WALK; i++) { last = d[cur][i] + cc[cur][i] * last; d[cur][i] = last; }
#pragma unroll
                        for (int i = 0; i < GAE_WALK; i++) sA[(r - 1 - i) * EPB + lane] = d[cur][i];
                    }
                }
                asm volatile("" ::: "memory");
                if (lane == 0) f_done[g] = (uint32_t)(tile + 1);   // behind the writes of A in this wave's LDS queue
            }
        }
        return;
    }

    // ========================================================= the movers =========================================================
    // the thread's slots, latest rows first: the jj-th one is rows r0 + (256 / C4) (NG - 1 - jj)
    const int r0 = tid / C4, c = (tid % C4) * 4;
    auto row_of = [&](int jj) { return r0 + (GAE_THREADS / C4) * (NG - 1 - jj); };
    for (int tile = 0; tile < n_tiles; tile++) {
        const int t_lo = T - (tile + 1) * GAE_TC;
        float4 rw[NG], vv[NG], nvv[NG];
        uint4 dd[NG];
        // every load of the tile before the first use, latest group first; the row above the last one (t + 1 == T) reads next_value / next_done: address
        // selects, not branches (a load inside a branch makes the compiler wait for every outstanding load at the join)
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const int r = row_of(j), t = t_lo + r;
            const size_t gidx = (size_t)t * N + n0 + c;
            const bool lastrow = t + 1 >= T;
            rw[j] = *reinterpret_cast<const float4*>(rewards + gidx);
            vv[j] = *reinterpret_cast<const float4*>(values + gidx);
            nvv[j] = *reinterpret_cast<const float4*>(lastrow ? next_value + n0 + c : values + gidx + N);
            dd[j] = *reinterpret_cast<const uint4*>(lastrow ? reinterpret_cast<const void*>(next_done + n0 + c) : reinterpret_cast<const void*>(dones + gidx + N));
        }
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const int r = row_of(j), g = r / GP_GROWS, t = t_lo + r;
            const bool lastrow = t + 1 >= T;
            // dones are 0 / 1 floats, next_done 0 / 1 integers: either way 1 - d is exact
            const float d0 = lastrow ? (float)(int)dd[j].x : __builtin_bit_cast(float, dd[j].x), d1 = lastrow ? (float)(int)dd[j].y : __builtin_bit_cast(float, dd[j].y);
            const float d2 = lastrow ? (float)(int)dd[j].z : __builtin_bit_cast(float, dd[j].z), d3 = lastrow ? (float)(int)dd[j].w : __builtin_bit_cast(float, dd[j].w);
            const float4 nnt = make_float4(1.0f - d0, 1.0f - d1, 1.0f - d2, 1.0f - d3);
            const float4 nv = nvv[j], v = vv[j];
            float4 a4, c4;
            if (MODE == 0) {
                a4 = make_float4((rw[j].x + (gamma * nv.x) * nnt.x) - v.x, (rw[j].y + (gamma * nv.y) * nnt.y) - v.y,   // :300
                                 (rw[j].z + (gamma * nv.z) * nnt.z) - v.z, (rw[j].w + (gamma * nv.w) * nnt.w) - v.w);
                c4 = make_float4(gl * nnt.x, gl * nnt.y, gl * nnt.z, gl * nnt.w);                                     // :301
            } else {
                a4 = rw[j];
                c4 = make_float4(gamma * nnt.x, gamma * nnt.y, gamma * nnt.z, gamma * nnt.w);                         // :324
            }
            if (tile > 0 && !gp_wait_ge(f_done + g, (uint32_t)tile)) {   // (own slot: this thread stored the previous tile's values of it already; the walker is done with the row)
                a4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                if (lane == 0 && error_flag) atomicOr(error_flag, PPO_ERRFLAG_GAE_PROTOCOL);
            }
            *reinterpret_cast<float4*>(&sA[r * EPB + c]) = a4;
            *reinterpret_cast<float4*>(&sC[r * EPB + c]) = c4;
            asm volatile("" ::: "memory");
            if (lane == 0) atomicAdd(&s_ready[g], 1u);   // behind this wave's writes in its LDS queue
        }
        // ---- the groups come back walked, latest first: stream A_t and R_t out ----
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const int r = row_of(j), g = r / GP_GROWS;
            const bool walked = gp_wait_ge(f_done + g, (uint32_t)(tile + 1));
            float4 a4 = *reinterpret_cast<const float4*>(&sA[r * EPB + c]);
            if (!walked) {
                a4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                if (lane == 0 && error_flag) atomicOr(error_flag, PPO_ERRFLAG_GAE_PROTOCOL);
            }
            const float4 v4 = vv[j];
            float4 o_adv, o_ret;
            if (MODE == 0) { o_adv = a4; o_ret = make_float4(a4.x + v4.x, a4.y + v4.y, a4.z + v4.z, a4.w + v4.w); }   // :305
            else { o_ret = a4; o_adv = make_float4(a4.x - v4.x, a4.y - v4.y, a4.z - v4.z, a4.w - v4.w); }              // :327
            const size_t gidx = (size_t)(t_lo + r) * N + n0 + c;
            *reinterpret_cast<float4*>(adv + gidx) = o_adv;
            *reinterpret_cast<float4*>(ret + gidx) = o_ret;
        }
    }
}

template <int MODE, bool FAST = false>
hipError_t launch_scan(const float* rewards, const float* values, const float* dones, const float* next_value,
                       const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* adv, float* ret,
                       int32_t* error_flag, hipStream_t s) {
    if (T <= 0 || N <= 0) return hipSuccess;
    if (T > INT32_MAX || N > INT32_MAX) return hipErrorInvalidValue;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec_ok = (N % 4 == 0) && al16(rewards) && al16(values) && al16(dones) && al16(next_value) && al16(next_done) &&
                        al16(adv) && al16(ret);
    // Strip width: wide strips coalesce better (EPB*4-byte rows), narrow strips give more workgroups.  Keep >= ~2 per CU.
    // Strip width, measured on MI355X (tools/gae_sweep.py): 32 columns (48 KB of LDS, three workgroups per CU overlapping their
    // load / walk / store phases) is best from 32 768 envs up (4.5 TB/s); below ~8 192 envs 16 columns give every CU a workgroup.
#ifdef GAE_EPB64_FROM   /* exploration builds: 64-column strips (96 KB of LDS: one workgroup per CU) from this many envs -- measured at 131 072 / 2^20 envs: 2.88 / 2.88 TB/s against 4.34 / 4.06 */
    const int epb = N >= GAE_EPB64_FROM ? 64 : (N >= 8192 ? 32 : 16);
#else
    const int epb = N >= 8192 ? 32 : 16;
#endif
#ifndef GAE_NO_TIME_PIPELINE   /* -DGAE_NO_TIME_PIPELINE: the three-phase kernel at every size (A/B) */
    // Whole tiles, whole strips, and a grid of about one workgroup per CU: the kernel pipelined in time (same bits).  Measured in trace, builds alternated in
    // one call, two rounds (three-phase | 16 columns x 2 groups | 16 x 4 | 32 x 2 | 32 x 4, us): 4096 envs 4.86 - 5.12 | 4.37 - 4.71 | 4.27 - 4.57 | 4.65 - 4.97 |
    // 5.2; 8192 envs 6.2 - 7.5 | 8.1 - 8.6 | 8.1 - 8.4 | 5.9 - 6.0 | 5.66 - 5.74; 16 384 envs 9.1 | 15.4 | 15.4 | 10.0 | 9.9; 32 768 envs 18.5 | 30 | 30 | 20.7 |
    // 20.4 -- with two and more workgroups per CU the three-phase kernel's workgroups overlap their phases among themselves and the hand-overs only cost.
#ifdef GP_FORCE   /* exploration builds: -DGP_FORCE=<epb * 10 + groups>, e.g. 164 */
    const int gp_epb = GP_FORCE / 10, gp_groups = GP_FORCE % 10;
#else
    const int gp_epb = 32, gp_groups = 4;
#endif
    // ... and only above 4096 envs: at 4096 (16-column strips) the pipelined kernel is ahead in a trace of the scan alone (4.3 - 4.8 us against 4.9 - 5.1) but
    // not inside the training iteration (5.34 against 5.27 us in trace) and slower back to back (6.6 against 5.1 us: five waves per workgroup leave the next
    // launch's workgroups less room to start under the tail of this one).  8192 envs = BASELINE configs[3]'s size: 5.7 us, 0.46 of 8 TB/s in trace.
    if (!FAST && vec_ok && T % GAE_TC == 0 && N % gp_epb == 0 && N / gp_epb <= (int64_t)GAE_PIPE_MAX_BLOCKS && N > GAE_PIPE_MIN_ENVS) {
        const dim3 grid((unsigned)(N / gp_epb)), block(GP_THREADS);
#define PPO_GP_LAUNCH(E, G) hipLaunchKernelGGL((gae_pipe_kernel<E, MODE, G>), grid, block, 0, s, rewards, values, dones, next_value, next_done, (int)T, (int)N, gamma, gae_lambda, adv, ret, error_flag)
        if (gp_epb == 32 && gp_groups == 4) PPO_GP_LAUNCH(32, 4);
        else if (gp_epb == 32) PPO_GP_LAUNCH(32, 2);
        else if (gp_groups == 4) PPO_GP_LAUNCH(16, 4);
        else PPO_GP_LAUNCH(16, 2);
#undef PPO_GP_LAUNCH
        return hipGetLastError();
    }
#endif
    // From GAE_WIDE_FROM envs (two and more workgroups per CU even with 64-column strips): strips of 64 columns x a time tile of 32 rows, v_t kept in registers.
    // The 64-column strip reads 256-byte row pieces (two memory lines per row and array instead of one) and fills all 64 lanes of the walking wave; the SHORT
    // time tile keeps its LDS at 16 KB so that eight workgroups share a CU and overlap each other's load / walk / store phases -- round 5's 64-column experiment
    // kept 128 rows (96 KB, ONE workgroup per CU) and lost for that reason.  Same chain, same operation order: same bits (the carry crosses tiles in a register
    // as before).  Measured in trace, builds alternated in one call (profiles/NOTES.md "GAE: wide strips", fraction of 8 TB/s at 16 384 / 32 768 / 65 536 /
    // 131 072 / 2^20 envs): 32 x 128 (round 5) 0.58 / 0.57 / 0.61 / 0.53 - 0.55 / 0.51; 64 x 128 - / - / 0.50 / 0.36 / 0.37; 64 x 64 - / - / 0.69 / 0.57 - 0.59 /
    // 0.60 - 0.63; 64 x 32 - / - / 0.75 / 0.60 / 0.62 - 0.64; 64 x 16 and 128 x 16, 128 x 32: within 0.02 of 64 x 32; 64 x 32 with non-temporal stores
    // 0.66 / 0.68 / 0.76 / 0.75 - 0.79 / 0.63 - 0.69 (non-temporal loads as well: worse at 65 536 and 131 072, the same at 2^20).
    // Non-temporal stores from GAE_NT_FROM envs only: below that the batch (20 B x T x N) fits the memory-side cache with room to spare and the kernels that read
    // the advantages next (pack_records, perm_adv_stats) find them there.
#ifndef GAE_WIDE_FROM
#define GAE_WIDE_FROM 16384
#endif
#ifndef GAE_NT_FROM
#define GAE_NT_FROM 65536
#endif
#ifndef GAE_WIDE_CFG   /* exploration builds: -DGAE_WIDE_CFG=<epb * 1000 + tc>, e.g. 64064, 64016, 128016, 32128 */
#define GAE_WIDE_CFG 64032
#endif
#define PPO_GAE_LAUNCH(EPB, TC, KEEPV, NT)                                                                                    \
    do {                                                                                                                      \
        const dim3 grid((unsigned)((N + EPB - 1) / EPB)), block(GAE_THREADS);                                                  \
        if (vec_ok && N % EPB == 0)                                                                                           \
            hipLaunchKernelGGL((gae_kernel<EPB, MODE, true, FAST, TC, KEEPV && !FAST, NT>), grid, block, 0, s, rewards, values, dones, next_value, next_done, \
                               (int)T, (int)N, gamma, gae_lambda, adv, ret);                                                  \
        else                                                                                                                  \
            hipLaunchKernelGGL((gae_kernel<EPB, MODE, false, FAST, TC, false, false>), grid, block, 0, s, rewards, values, dones, next_value, next_done, \
                               (int)T, (int)N, gamma, gae_lambda, adv, ret);                                                  \
    } while (0)
    constexpr int wide_epb = GAE_WIDE_CFG / 1000, wide_tc = GAE_WIDE_CFG % 1000;
    if (!FAST && vec_ok && N >= GAE_WIDE_FROM && N % wide_epb == 0) {
        if (N >= GAE_NT_FROM) PPO_GAE_LAUNCH(wide_epb, wide_tc, true, true);
        else PPO_GAE_LAUNCH(wide_epb, wide_tc, true, false);
    }
    else if (epb == 64) PPO_GAE_LAUNCH(64, GAE_TC, false, false);
    else if (epb == 32) PPO_GAE_LAUNCH(32, GAE_TC, GAE_KEEPV_ALL, false);
    else PPO_GAE_LAUNCH(16, GAE_TC, GAE_KEEPV_ALL, false);
#undef PPO_GAE_LAUNCH
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gae(const float* rewards, const float* values, const float* dones, const float* next_value,
                      const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* adv, float* ret,
                      int32_t* error_flag, hipStream_t s) {
    return launch_scan<0>(rewards, values, dones, next_value, next_done, T, N, gamma, gae_lambda, adv, ret, error_flag, s);
}

hipError_t launch_gae_fast(const float* rewards, const float* values, const float* dones, const float* next_value,
                           const int32_t* next_done, int64_t T, int64_t N, float gamma, float gae_lambda, float* adv, float* ret,
                           hipStream_t s) {
    return launch_scan<0, true>(rewards, values, dones, next_value, next_done, T, N, gamma, gae_lambda, adv, ret, nullptr, s);
}

hipError_t launch_nstep(const float* rewards, const float* values, const float* dones, const float* next_value,
                        const int32_t* next_done, int64_t T, int64_t N, float gamma, float* adv, float* ret, int32_t* error_flag, hipStream_t s) {
    return launch_scan<1>(rewards, values, dones, next_value, next_done, T, N, gamma, 1.0f, adv, ret, error_flag, s);
}
