// Probe: fp32 carried as TWO fp16 terms (x = t1 + t2, both rounded toward zero: v_cvt_pkrtz_f16_f32 saturates instead of overflowing)
// with the residual as one v_fma_mix_f32 (fma(t1.half, -1.0, x)).
//   (1) exactness of the residual and the size of x - (t1 + t2) over every exponent fp16 can hold
//   (2) issue cycles of the candidate instructions, alone (two waves per SIMD) and beside a partner wave issuing MFMAs
//   (3) v_mfma_f32_32x32x16_f16: cycles, and whether fp16 denormal operands are honoured
//   hipcc --offload-arch=gfx950 -O3 -o f16_split f16_split.hip && ./f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pkrtz(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ float resid_lo(uint32_t p, float x) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float resid_hi(uint32_t p, float x) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}

// stats[0] = residual mismatches, stats[1] = max rel error (as float bits, values with |x| >= 2^-3), stats[2] = max abs error (bits)
__global__ void exact_kernel(const float* x, uint32_t* stats, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float x0 = x[i], x1 = x[i + 1];
    const uint32_t p = pkrtz(x0, x1);
    const f16x2 h = __builtin_bit_cast(f16x2, p);
    const float r0 = resid_lo(p, x0), r1 = resid_hi(p, x1);
    const float e0 = x0 - (float)h[0], e1 = x1 - (float)h[1];
    if (__builtin_bit_cast(uint32_t, r0) != __builtin_bit_cast(uint32_t, e0) || __builtin_bit_cast(uint32_t, r1) != __builtin_bit_cast(uint32_t, e1))
        atomicAdd(stats, 1u);
    const uint32_t q = pkrtz(r0, r1);
    const f16x2 g = __builtin_bit_cast(f16x2, q);
    const float d0 = fabsf(r0 - (float)g[0]), d1 = fabsf(r1 - (float)g[1]);   // what two terms leave behind
    if (fabsf(x0) >= 0.125f) atomicMax(stats + 1, __builtin_bit_cast(uint32_t, d0 / fabsf(x0)));
    if (fabsf(x1) >= 0.125f) atomicMax(stats + 1, __builtin_bit_cast(uint32_t, d1 / fabsf(x1)));
    atomicMax(stats + 2, __builtin_bit_cast(uint32_t, d0));
    atomicMax(stats + 2, __builtin_bit_cast(uint32_t, d1));
}

#define OPS(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
    X(1, "v_exp_f32", "v_exp_f32 %0, %1") \
    X(2, "v_rcp_f32", "v_rcp_f32 %0, %1") \
    X(3, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %1, %2") \
    X(4, "v_fma_mix_f32 (f16 src0)", "v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]") \
    X(5, "v_perm_b32", "v_perm_b32 %0, %1, %2, %1") \
    X(6, "v_and_b32", "v_and_b32 %0, %1, %2") \
    X(7, "v_dot2c_f32_bf16", "v_dot2c_f32_bf16 %0, %1, %2") \
    X(8, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %1, %2") \
    X(9, "v_mul_f32", "v_mul_f32 %0, %1, %2") \
    X(10, "v_pk_fma_f32", "v_pk_fma_f32 %0, %3, %3, %0") \
    X(11, "v_pk_mul_f32", "v_pk_mul_f32 %0, %3, %3") \
    X(12, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %1") \
    X(13, "v_ldexp_f32", "v_ldexp_f32 %0, %1, %2")

// MODE 0: waves 0-3 issue NM MFMAs per iteration, waves 4-7 (their SIMD partners) NV vector ops; 1: every wave NV vector ops (two per SIMD)
template <int MODE, int OP, int NM, int NV, bool F16>
__global__ __launch_bounds__(512, 1) void rate_kernel(float* out, unsigned long long* cyc, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
    double w[8];
    for (int i = 0; i < 8; i++) w[i] = seed + i;
    const float a = seed * 0.5f, b = seed * 0.25f;
    const double d = seed * 0.5;
    bf16x8 pa, pb;
    f16x8 ha, hb;
    for (int i = 0; i < 8; i++) { pa[i] = (__bf16)(seed * 0.5f); pb[i] = (__bf16)(seed * 0.25f); ha[i] = (_Float16)(seed * 0.5f); hb[i] = (_Float16)(seed * 0.25f); }
    const bool do_m = MODE == 0 && wave < 4, do_v = MODE == 1 || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) {
                if (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
#define X(id, name, text) if (OP == id) { if (id == 10 || id == 11) asm volatile(text : "+v"(w[k & 7]) : "v"(a), "v"(b), "v"(d)); else asm volatile(text : "+v"(v[k & 7]) : "v"(a), "v"(b), "v"(d)); }
                OPS(X)
#undef X
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    for (int i = 0; i < 8; i++) s += v[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MODE, int OP, int NM, int NV, bool F16>
void rate(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) { rate_kernel<MODE, OP, NM, NV, F16><<<256, 512>>>(out, cyc, iters, 1.0f); hipDeviceSynchronize(); }
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    if (MODE == 1) printf("%-28s alone, 2 waves/SIMD: %6.2f cycles per instruction and SIMD\n", name, (double)h[1] / iters / NV / 2);
    else printf("%-28s beside %2d %s MFMAs: mfma wave %7.1f cycles/iter (%.1f per MFMA), vector wave %7.1f (%.2f per instruction)\n", name, NM,
                F16 ? "f16 " : "bf16", (double)h[0] / iters, (double)h[0] / iters / NM, (double)h[1] / iters, (double)h[1] / iters / NV);
    hipFree(out); hipFree(cyc);
}

// one 32x32x16 f16 product with denormal A entries: A[i][k] = 2^-20 (fp16 denormal), B[k][j] = 2^10  ->  every C = 16 * 2^-10
__global__ void denorm_kernel(float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = __builtin_bit_cast(_Float16, (uint16_t)0x0010); b[i] = (_Float16)1024.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    const uint32_t p = pkrtz(9.5367431640625e-07f, 1e30f);   // 2^-20 -> denormal half; 1e30 -> saturates at 65504 under RTZ?
    const f16x2 h = __builtin_bit_cast(f16x2, p);
    if (threadIdx.x == 0) { out[1] = (float)h[0]; out[2] = (float)h[1]; out[3] = resid_lo(p, 9.5367431640625e-07f * 1.5f); }
}

int main() {
    const int n = 1 << 24;
    float* hx = (float*)malloc(n * 4);
    uint32_t s = 12345u;
    for (int i = 0; i < n; i++) {   // exponents 2^-30 .. 2^15, random mantissas and signs
        s = s * 1664525u + 1013904223u;
        const uint32_t mant = s & 0x807fffffu;
        s = s * 1664525u + 1013904223u;
        const uint32_t e = 127 - 30 + (s >> 8) % 46;
        const uint32_t u = mant | (e << 23);
        memcpy(&hx[i], &u, 4);
    }
    float* dx; uint32_t* st;
    hipMalloc(&dx, n * 4); hipMalloc(&st, 16);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemset(st, 0, 16);
    exact_kernel<<<n / 2 / 256, 256>>>(dx, st, n);
    uint32_t hs[4]; hipMemcpy(hs, st, 16, hipMemcpyDeviceToHost);
    float rel, ab; memcpy(&rel, &hs[1], 4); memcpy(&ab, &hs[2], 4);
    printf("two fp16 terms over %d values: residual mismatches %u, max |x - t1 - t2| / |x| = 2^%.2f for |x| >= 1/8, max abs = 2^%.2f\n", n, hs[0],
           log2(rel), log2(ab));
    float* o; hipMalloc(&o, 64);
    denorm_kernel<<<1, 64>>>(o);
    float ho[4]; hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
    printf("f16 MFMA with denormal A: C = %g (exact: %g); pkrtz(2^-20) = %g, pkrtz(1e30) = %g, residual of 1.5 * 2^-20 = %g\n", ho[0], 16.0 * 1024.0 / 1048576.0,
           ho[1], ho[2], ho[3]);
#define X(id, name, text) rate<1, id, 0, 256, false>(name);
    OPS(X)
#undef X
    rate<0, 0, 16, 128, false>("v_fma_f32");
    rate<0, 0, 16, 128, true>("v_fma_f32");
    rate<0, 3, 16, 128, true>("v_cvt_pkrtz_f16_f32");
    rate<0, 4, 16, 128, true>("v_fma_mix_f32");
    rate<0, 1, 16, 32, false>("v_exp_f32");
    rate<0, 2, 16, 32, false>("v_rcp_f32");
    rate<0, 5, 16, 128, false>("v_perm_b32");
    rate<0, 10, 16, 64, false>("v_pk_fma_f32");
    rate<0, 0, 16, 256, false>("v_fma_f32 (256)");
    return 0;
}
