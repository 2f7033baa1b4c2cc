// Probe: what does rocprofv3's FETCH_SIZE report for the two read patterns of this build?  (MI355X_MICROARCH.md "HBM": wide coalesced
// streaming reads are tallied at half their size on gfx950; "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern".)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -f csv -d out -o run -- ./fetch_calib
//   stream_read   reads 64 MiB with 16-byte loads, coalesced                                  -> known bytes: 67 108 864
//   gather_read   the update kernel's gather: 262 144 random 32-byte records out of a 16 MiB table of 524 288 records (a permutation, every
//                 record at most once), two 16-byte loads per record, lanes l and l + 32 of a wave reading the same record
//                                                                                             -> useful bytes: 8 388 608
// Both run 20 times; a 512 MiB memset between launches evicts the table from the 256 MiB Infinity Cache is NOT done: the update kernel
// itself finds its records in the Infinity Cache, and FETCH_SIZE counts L2-side requests either way.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ p, size_t n4, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(512) void gather_read(const float4* __restrict__ rec, const int* __restrict__ idx, int n, float* out) {
    float acc = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31;
    for (int tile = blockIdx.x * 8 + wave; tile * 32 < n; tile += gridDim.x * 8) {
        const int row = idx[tile * 32 + s];
        const float4 a = rec[2 * (size_t)row], b = rec[2 * (size_t)row + 1];
        acc += a.x + a.w + b.x + b.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t stream_bytes = 64u << 20;
    const int n_rec = 524288, n_gather = 262144;
    float4 *p, *rec; int* idx; float* out;
    CK(hipMalloc(&p, stream_bytes)); CK(hipMalloc(&rec, (size_t)n_rec * 32)); CK(hipMalloc(&idx, n_gather * 4)); CK(hipMalloc(&out, 16));
    CK(hipMemset(p, 0, stream_bytes)); CK(hipMemset(rec, 0, (size_t)n_rec * 32));
    std::vector<int> h(n_rec);
    for (int i = 0; i < n_rec; i++) h[i] = i;
    unsigned long long st = 88172645463325252ull;
    for (int i = n_rec - 1; i > 0; i--) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; const int j = (int)(st % (unsigned long long)(i + 1)); std::swap(h[i], h[j]); }
    CK(hipMemcpy(idx, h.data(), n_gather * 4, hipMemcpyHostToDevice));
    for (int r = 0; r < 20; r++) {
        hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, p, stream_bytes / 16, out);
        hipLaunchKernelGGL(gather_read, dim3(256), dim3(512), 0, 0, rec, idx, n_gather, out);
    }
    CK(hipDeviceSynchronize());
    printf("stream_read known bytes %zu; gather_read useful bytes %d (+ %d of indices)\n", stream_bytes, n_gather * 32, n_gather * 4);
    return 0;
}
