// Probe: what ANY kernel with the GAE scan's memory shape costs on this part, without the scan.
//   hipcc --offload-arch=gfx950 -O3 -o gae_floor gae_floor.hip && ./gae_floor [N ...]        (T = 128)
// Three kernels over time-major [T, N] fp32 buffers, each launched `reps` times back to back (wall time / reps) -- run the binary under
// `rocprofv3 --kernel-trace --stats` for their in-trace durations:
//   stream_kernel   reads r, v, d (12 B / element) and writes two arrays (8 B / element) with 16-byte accesses, one pass, no LDS, no chain:
//                   the floor of "launch + one memory round trip + store drain" at the scan's byte count
//   column_kernel   the scan's own decomposition (one workgroup per strip of 16 env columns, 128 rows staged in LDS between two
//                   barriers) with the chain replaced by nothing: what the strip shape itself costs
//   empty_kernel    256 workgroups that do nothing: launch + grid ramp
// The scan (kernels_gae.hip) is then: column_kernel + the 128-step serial walk.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 0; }

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ r, const float4* __restrict__ v, const float4* __restrict__ d, float4* __restrict__ o1,
                                                     float4* __restrict__ o2, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = r[i], b = v[i], c = d[i];
        o1[i] = make_float4(a.x + b.x * c.x, a.y + b.y * c.y, a.z + b.z * c.z, a.w + b.w * c.w);
        o2[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
}

template <int EPB>
__global__ __launch_bounds__(256) void column_kernel(const float* __restrict__ r, const float* __restrict__ v, const float* __restrict__ d, float* __restrict__ o1,
                                                     float* __restrict__ o2, int T, int N) {
    __shared__ __attribute__((aligned(16))) float sA[128 * EPB], sB[128 * EPB];
    constexpr int C4 = EPB / 4, ITERS = 128 * C4 / 256;
    const int tid = threadIdx.x, n0 = blockIdx.x * EPB;
    float4 a[ITERS], b[ITERS], c[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int e = tid + i * 256, row = e / C4, col = (e % C4) * 4;
        const size_t g = (size_t)row * N + n0 + col;
        a[i] = *reinterpret_cast<const float4*>(r + g); b[i] = *reinterpret_cast<const float4*>(v + g); c[i] = *reinterpret_cast<const float4*>(d + g);
    }
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int e = tid + i * 256, row = e / C4, col = (e % C4) * 4;
        *reinterpret_cast<float4*>(&sA[row * EPB + col]) = make_float4(a[i].x + b[i].x * c[i].x, a[i].y + b[i].y * c[i].y, a[i].z + b[i].z * c[i].z, a[i].w + b[i].w * c[i].w);
        *reinterpret_cast<float4*>(&sB[row * EPB + col]) = c[i];
    }
    __syncthreads();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int e = tid + i * 256, row = e / C4, col = (e % C4) * 4;
        const int rr = (row + 1) & 127;
        const float4 x = *reinterpret_cast<const float4*>(&sA[rr * EPB + col]), y = *reinterpret_cast<const float4*>(&sB[rr * EPB + col]);
        const size_t g = (size_t)row * N + n0 + col;
        *reinterpret_cast<float4*>(o1 + g) = x;
        *reinterpret_cast<float4*>(o2 + g) = make_float4(x.x + b[i].x + y.x, x.y + b[i].y + y.y, x.z + b[i].z + y.z, x.w + b[i].w + y.w);
    }
}

// chain_kernel: the scan's strip decomposition with NO global traffic -- the tile's delta / coefficient rows are made in LDS by the workgroup itself,
// then one lane per env column walks the 128-step chain A_t = d_t + c_t * A_{t+1} exactly as kernels_gae.hip does (two separately rounded operations
// per row, LDS reads of the next 16 rows in flight while a chunk's chain runs), then one store per column keeps the result alive.
// chain_us - empty_us = what the serial walk costs when nothing else is in its way: the part of the scan no overlap can remove besides the memory
// round trip of its FIRST rows and the drain of its LAST ones (column_us is that round trip + drain).
template <int EPB>
__global__ __launch_bounds__(256) void chain_kernel(float* __restrict__ o1, int N, float seed) {
    __shared__ __attribute__((aligned(16))) float sA[128 * EPB], sC[128 * EPB];
    const int tid = threadIdx.x, n0 = blockIdx.x * EPB;
    for (int e = tid; e < 128 * EPB; e += 256) { sA[e] = seed + (float)(e & 7); sC[e] = (e & 31) ? 0.931f : 0.0f; }
    __syncthreads();
    if (tid < EPB) {
        float last = 0.0f;
        float d[2][16], cc[2][16];
#pragma unroll
        for (int i = 0; i < 16; i++) { d[0][i] = sA[(127 - i) * EPB + tid]; cc[0][i] = sC[(127 - i) * EPB + tid]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int cur = k & 1, nxt = cur ^ 1, r = 128 - k * 16;
            if (k + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 16; i++) { d[nxt][i] = sA[(r - 17 - i) * EPB + tid]; cc[nxt][i] = sC[(r - 17 - i) * EPB + tid]; }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float m = cc[cur][i] * last;          // separately rounded, like the reference's tensor expression (-ffp-contract=off)
                last = d[cur][i] + m;
                d[cur][i] = last;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) sA[(r - 1 - i) * EPB + tid] = d[cur][i];
        }
        o1[n0 + tid] = last;
    }
}

int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = { 4096, 8192, 32768 };
    const int T = 128;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int N : sizes) {
        const size_t n = (size_t)T * N;
        float *r, *v, *d, *o1, *o2;
        CK(hipMalloc(&r, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
        CK(hipMemset(r, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); CK(hipMemset(d, 0, n * 4));
        const int reps = 200;
        auto timeit = [&](auto launch) {
            for (int i = 0; i < 5; i++) launch();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; i++) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return 1e3 * ms / reps;
        };
        const unsigned g_stream = (unsigned)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048);
        const double t_empty = timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, r); });
        const double t_stream = timeit([&] { hipLaunchKernelGGL(stream_kernel, dim3(g_stream), dim3(256), 0, s, (const float4*)r, (const float4*)v, (const float4*)d, (float4*)o1, (float4*)o2, n / 4); });
        const double t_col = N >= 8192 ? timeit([&] { hipLaunchKernelGGL(column_kernel<32>, dim3(N / 32), dim3(256), 0, s, r, v, d, o1, o2, T, N); })
                                       : timeit([&] { hipLaunchKernelGGL(column_kernel<16>, dim3(N / 16), dim3(256), 0, s, r, v, d, o1, o2, T, N); });
        const double t_chain = N >= 8192 ? timeit([&] { hipLaunchKernelGGL(chain_kernel<32>, dim3(N / 32), dim3(256), 0, s, o1, N, 0.5f); })
                                         : timeit([&] { hipLaunchKernelGGL(chain_kernel<16>, dim3(N / 16), dim3(256), 0, s, o1, N, 0.5f); });
        const double bytes = 20.0 * n;
        printf("{\"N\": %d, \"T\": %d, \"bytes\": %.0f, \"empty_us\": %.2f, \"stream_us\": %.2f, \"stream_GBs\": %.0f, \"column_us\": %.2f, \"column_GBs\": %.0f, "
               "\"chain_us\": %.2f, \"chain_minus_empty_us\": %.2f, \"column_plus_chain_us\": %.2f, \"us_at_40pct_of_8TBs\": %.2f}\n", N, T, bytes, t_empty, t_stream,
               bytes / t_stream / 1e3, t_col, bytes / t_col / 1e3, t_chain, t_chain - t_empty, t_col + (t_chain - t_empty), bytes / 3.2e12 * 1e6);
        CK(hipFree(r)); CK(hipFree(v)); CK(hipFree(d)); CK(hipFree(o1)); CK(hipFree(o2));
    }
    return 0;
}
