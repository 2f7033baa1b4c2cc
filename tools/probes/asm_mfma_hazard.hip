// Probe: may an MFMA read, as its A / B operand, a VGPR that the instruction right in front of it wrote with v_cvt_pk_f16_f32 / v_fma_mix_f32?
// (kernels_update_mfma.hip forms its fp16 terms with inline asm, which LLVM's hazard recognizer does not look into.)  Two kernels compute the
// same 32x32x16 product; one issues conversion and MFMA back to back inside ONE asm block, the other puts s_nop 7 between them.  Equal bits on
// every lane = the hardware interlocks (or needs no wait states).   hipcc --offload-arch=gfx950 -O3 -o asm_mfma_hazard asm_mfma_hazard.hip
// ROUND 3: THIS PROBE'S CONCLUSION WAS WRONG.  Its MFMA sat in an accumulator chain (stalled at issue long enough for the operand to land); an
// MFMA with nothing to wait for reads the OLD register: rollout16_kernel's logits were off by 1e-3 with one s_nop 0 between the asm conversion and
// the MFMA.  The rule is 2 wait states (hipcc pads `s_nop 1` itself when the conversion is its own instruction: tools/probes/mfma16_split.hip,
// tools/check_asm_hazards.py); pk_f16 / r16_pk are compiler-formed since.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ZERO16(b) "v_mov_b32 v" #b ", 0\n\t"
template <bool NOPS>
__global__ void k(const float* __restrict__ x, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const float a0 = x[lane * 8 + 0], a1 = x[lane * 8 + 1], b0 = x[lane * 8 + 4], b1 = x[lane * 8 + 5];
    float o[4];
    // fixed registers: A = v[40:43], B = v[44:47], accumulator v[48:63], residual v64
#define BODY(PAD)                                                                                                                  \
    asm volatile(                                                                                                                  \
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"                                        \
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"                                        \
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t"  \
        "v_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"  \
        "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\ts_nop 7\n\t"                              \
        "v_cvt_pk_f16_f32 v44, %6, %7\n\t"                                                                                         \
        "v_cvt_pk_f16_f32 v40, %4, %5\n\t"                                                                                         \
        "v_fma_mix_f32 v64, v40, -1.0, %4 op_sel_hi:[1,0,0]\n\t"                                                                   \
        "v_cvt_pk_f16_f32 v41, v64, %5\n\t" PAD                                                                                    \
        "v_mfma_f32_32x32x16_f16 v[48:63], v[40:43], v[44:47], v[48:63]\n\t"                                                       \
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                                                                       \
        "v_mov_b32 %0, v48\n\tv_mov_b32 %1, v53\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v63\n\t"                                    \
        : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                      \
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", \
          "v59", "v60", "v61", "v62", "v63", "v64")
    if (NOPS) { BODY("s_nop 7\n\ts_nop 7\n\t"); } else { BODY(""); }
    for (int r = 0; r < 4; r++) out[lane * 16 + r] = o[r];
    for (int r = 4; r < 16; r++) out[lane * 16 + r] = 0.0f;
}
int main() {
    float hx[64 * 8];
    for (int i = 0; i < 64 * 8; i++) hx[i] = (float)((i * 37) % 101) / 13.0f - 3.0f;
    float *dx, *o0, *o1;
    hipMalloc(&dx, sizeof hx); hipMalloc(&o0, 64 * 16 * 4); hipMalloc(&o1, 64 * 16 * 4);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    int bad = 0;
    for (int rep = 0; rep < 50; rep++) {
        k<true><<<1, 64>>>(dx, o0);
        k<false><<<1, 64>>>(dx, o1);
        float h0[1024], h1[1024];
        hipMemcpy(h0, o0, sizeof h0, hipMemcpyDeviceToHost); hipMemcpy(h1, o1, sizeof h1, hipMemcpyDeviceToHost);
        bad += memcmp(h0, h1, sizeof h0) != 0;
        if (rep == 0) printf("sample: %g %g %g (with nops) | %g %g %g (back to back)\n", h0[0], h0[17], h0[100], h1[0], h1[17], h1[100]);
    }
    printf("back-to-back conversion -> MFMA differs from the padded sequence in %d of 50 runs\n", bad);
    return 0;
}
