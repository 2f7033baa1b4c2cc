// Probe: do f32-input MFMAs (v_mfma_f32_32x32x2_f32) and plain f32 VALU work overlap on one SIMD of gfx950?
//   mode 0: every wave issues NM dependent MFMAs per iteration
//   mode 1: every wave issues NV dependent-chain-free v_fma per iteration
//   mode 2: every wave issues both, interleaved (same wave)
//   mode 3: waves 0-3 of a 512-thread workgroup issue the MFMAs, waves 4-7 (their SIMD partners) the VALU work
// One workgroup per CU, 256 workgroups.  Prints cycles per iteration (s_memtime) of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef USE_BF16
#define MFMA(acc) __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0)
#else
#define MFMA(acc) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#endif
#ifdef PACKED
#define VFMA(x) x = __builtin_fmaf(x, a, b)
#else
#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#endif
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(512, 1) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; i++) { pa[i] = (__bf16)(seed * 0.5f); pb[i] = (__bf16)(seed * 0.25f); }
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < NM; m++) {
                acc = MFMA(acc);
#pragma unroll
                for (int k = 0; k < NV / NM; k++) VFMA(v[k & 7]);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; m++) acc = MFMA(acc);
            }
            if (do_v) {
#pragma unroll
                for (int k = 0; k < NV; k++) VFMA(v[k & 7]);
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

template <int MODE, int NM, int NV>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NM, NV><<<256, 512>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, NM, NV><<<256, 512>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-44s NM=%d NV=%d  %8.1f cycles/iter (wave0) %8.1f (wave4)  %.1f us\n", name, NM, NV, (double)h[0] / iters, (double)h[1] / iters, ms * 1e3);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 16, 0>("mfma only (2 waves/SIMD both mfma)");
    run<1, 0, 256>("valu only (2 waves/SIMD both valu)");
    run<2, 16, 256>("same wave interleaved");
    run<3, 16, 256>("partners: waves0-3 mfma, waves4-7 valu");
    run<3, 16, 128>("partners: waves0-3 mfma, waves4-7 valu");
    run<2, 16, 128>("same wave interleaved");
    return 0;
}
