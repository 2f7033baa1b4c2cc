// Probe: the residual of a bf16 truncation split, r = x - bf16_trunc(x), as ONE v_dot2c_f32_bf16 (x + (-1) * t.lo + 0 * t.hi) instead of
// v_and_b32 + v_sub_f32: (1) is it bit-exact, including tiny / huge / denormal residuals?  (2) does it co-execute with bf16 MFMAs like a
// plain v_fma_f32 does?   hipcc --offload-arch=gfx950 -O3 -o dot2_split dot2_split.hip && ./dot2_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float resid_lo(uint32_t packed, float x) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x0000bf80u), x, false);
}
__device__ __forceinline__ float resid_hi(uint32_t packed, float x) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0xbf800000u), x, false);
}
__global__ void exact_kernel(const float* x, uint32_t* bad, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float x0 = x[i], x1 = x[i + 1];
    const uint32_t u0 = __builtin_bit_cast(uint32_t, x0), u1 = __builtin_bit_cast(uint32_t, x1);
    const uint32_t p = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = resid_lo(p, x0), r1 = resid_hi(p, x1);
    const float e0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), e1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    if (__builtin_bit_cast(uint32_t, r0) != __builtin_bit_cast(uint32_t, e0) || __builtin_bit_cast(uint32_t, r1) != __builtin_bit_cast(uint32_t, e1)) {
        const uint32_t k = atomicAdd(bad, 1u);
        if (k < 8) { bad[1 + 4 * k] = u0; bad[2 + 4 * k] = __builtin_bit_cast(uint32_t, r0); bad[3 + 4 * k] = __builtin_bit_cast(uint32_t, e0); bad[4 + 4 * k] = u1; }
    }
}

// MODE 0: partners (waves 0-3 MFMA, waves 4-7 vector op), 1: vector op only.  OP 0: v_fma_f32, 1: v_dot2c_f32_bf16, 2: v_and + v_sub pair
template <int MODE, int OP, int NM, int NV>
__global__ __launch_bounds__(512, 1) void rate_kernel(float* out, unsigned long long* cyc, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
    uint32_t pk = __builtin_bit_cast(uint32_t, seed) | 0x3f80u;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; i++) { pa[i] = (__bf16)(seed * 0.5f); pb[i] = (__bf16)(seed * 0.25f); }
    const bool do_m = MODE == 0 && wave < 4, do_v = MODE == 1 || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
                else if (OP == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[k & 7]) : "v"(pk), "v"(pk));
                else { asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %1, %1, %0" : "=&v"(v[(k + 1) & 7]), "+v"(v[k & 7])); }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MODE, int OP, int NM, int NV>
void rate(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 200;
    rate_kernel<MODE, OP, NM, NV><<<256, 512>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    rate_kernel<MODE, OP, NM, NV><<<256, 512>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-52s NM=%d NV=%d  %8.1f cycles/iter (mfma wave) %8.1f (vector wave)\n", name, NM, NV, (double)h[0] / iters, (double)h[1] / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int n = 1 << 24;
    float* hx = (float*)malloc(n * 4);
    uint32_t s = 12345u;
    for (int i = 0; i < n; i++) {   // every exponent, random mantissas; a sprinkle of special patterns
        s = s * 1664525u + 1013904223u;
        uint32_t u = s;
        if ((i & 1023) == 0) u &= 0x807fffffu;             // denormals
        if ((i & 1023) == 1) u = (u & 0x80000000u) | 0x00800000u | (u & 0xffffu);  // smallest normals, low bits only
        if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xbfffffffu;   // no inf / nan
        memcpy(&hx[i], &u, 4);
    }
    float* dx; uint32_t* bad;
    hipMalloc(&dx, n * 4); hipMalloc(&bad, 64 * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 64 * 4);
    exact_kernel<<<n / 2 / 256, 256>>>(dx, bad, n);
    uint32_t hb[64]; hipMemcpy(hb, bad, 64 * 4, hipMemcpyDeviceToHost);
    printf("exactness over %d values: %u mismatches\n", n, hb[0]);
    for (uint32_t k = 0; k < hb[0] && k < 8; k++) printf("   x=%08x dot2=%08x and/sub=%08x (pair %08x)\n", hb[1 + 4 * k], hb[2 + 4 * k], hb[3 + 4 * k], hb[4 + 4 * k]);
    rate<1, 0, 0, 256>("v_fma_f32 alone");
    rate<1, 1, 0, 256>("v_dot2c_f32_bf16 alone");
    rate<1, 2, 0, 128>("v_and + v_sub pairs alone (128 pairs)");
    rate<0, 0, 16, 256>("v_fma_f32 beside 16 bf16 MFMAs");
    rate<0, 1, 16, 256>("v_dot2c_f32_bf16 beside 16 bf16 MFMAs");
    rate<0, 2, 16, 128>("v_and + v_sub pairs beside 16 bf16 MFMAs");
    return 0;
}
