// Probe: which vector instructions of the split / staging code co-execute with bf16 MFMAs issued by the SIMD's other wave?
// Waves 0-3 of a 512-thread workgroup issue 16 dependent v_mfma_f32_32x32x16_bf16 per iteration, waves 4-7 (their SIMD partners) 256 of the
// instruction under test.  Prints cycles per iteration of both, alone and together.
// Reading the output: the inline asm carries a vcc clobber, so the compiler puts an s_nop behind every instruction under test -- absolute
// cycles per instruction are ~2x the real ones; what the probe answers is the COMPARISON: and / sub / perm / cmp / lshl_add_u64 / f64 add all
// behave like v_fma_f32 beside MFMAs (none of them blocks or is blocked), transcendentals cost twice a plain instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_beside_mfma valu_beside_mfma.hip && ./valu_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define OPS(X) X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") X(1, "v_and_b32", "v_and_b32 %0, %1, %0") X(2, "v_sub_f32", "v_sub_f32 %0, %0, %1") \
    X(3, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") X(4, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(5, "v_cmp_lt_i32 (-> vcc)", "v_cmp_lt_i32 vcc, %0, %1") X(6, "v_lshl_add_u64", "v_lshl_add_u64 %3, %3, 0, %4") \
    X(7, "v_mov_b32", "v_mov_b32 %0, %1") X(8, "v_add_f32", "v_add_f32 %0, %0, %1") X(9, "v_exp_f32", "v_exp_f32 %0, %0") \
    X(10, "v_rcp_f32", "v_rcp_f32 %0, %0") X(11, "v_add_f64", "v_add_f64 %3, %3, %4") X(12, "v_mul_f32", "v_mul_f32 %0, %0, %1")

template <int MODE, int OP, int NM, int NV>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
    float a = seed * 0.5f, b = seed * 0.25f;
    double d0 = seed, d1 = seed * 3.0;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; i++) { pa[i] = (__bf16)(seed * 0.5f); pb[i] = (__bf16)(seed * 0.25f); }
    const bool do_m = MODE != 1 && wave < 4, do_v = MODE != 2 && (MODE == 1 || wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int q = 0; q < NV; q++) {
#define X(id, name, text) if (OP == id) asm volatile(text : "+v"(v[q & 7]) : "v"(a), "v"(b), "v"(d0), "v"(d1) : "vcc");
                OPS(X)
#undef X
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)d0;
    for (int r = 0; r < 16; r++) s += acc[r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MODE, int OP>
double run(int which) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 16);
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) { k<MODE, OP, 16, 256><<<256, 512>>>(out, cyc, iters, 1.0f); (void)hipDeviceSynchronize(); }
    unsigned long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)h[which] / iters;
}
template <int OP>
void one(const char* name) {
    const double alone = run<1, OP>(1), m_alone = run<2, OP>(0), v_tog = run<0, OP>(1), m_tog = run<0, OP>(0);
    printf("%-26s alone %7.1f (%.2f cyc/instr)   beside MFMA %7.1f   MFMA wave: alone %6.1f beside %6.1f   overlap %.0f%%\n", name, alone, alone / 256, v_tog,
           m_alone, m_tog, 100.0 * (alone + m_alone - (v_tog > m_tog ? v_tog : m_tog)) / (alone < m_alone ? alone : m_alone));
}
int main() {
#define X(id, name, text) one<id>(name);
    OPS(X)
#undef X
    return 0;
}
