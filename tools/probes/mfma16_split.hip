// Probe: the 3-product fp32-as-two-fp16-terms 16x16x16 MFMA on small magnitudes (W ~ 1e-3, h ~ 3e-2) against a double sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p1, uint32_t& p2) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(x0), "v"(x1));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p1), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p1), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p2) : "v"(r0), "v"(r1));
}
__device__ __forceinline__ f4 mf(uint2 a, uint2 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0); }
__global__ void k(const float* W, const float* Hh, float* out, float sw) {   // W[16][16] (i, k), Hh[16][16] (j, k)
    const int lane = threadIdx.x, e = lane & 15, kg = lane >> 4;
    float w[4], h[4];
    for (int v = 0; v < 4; v++) { w[v] = sw * W[e * 16 + 4 * kg + v]; h[v] = Hh[e * 16 + 4 * kg + v]; }
    uint32_t a1, a2, b1, b2, c1, c2, d1, d2;
    split2(w[0], w[1], a1, a2); split2(w[2], w[3], b1, b2);
    split2(h[0], h[1], c1, c2); split2(h[2], h[3], d1, d2);
    f4 z = { 0, 0, 0, 0 };
    z = mf(make_uint2(a2, b2), make_uint2(c1, d1), z);
    z = mf(make_uint2(a1, b1), make_uint2(c2, d2), z);
    z = mf(make_uint2(a1, b1), make_uint2(c1, d1), z);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = z[r] / sw;
}
int main() {
    float hW[256], hH[256], *W, *H, *o, ho[256];
    srand(1);
    for (int i = 0; i < 256; i++) { hW[i] = 2.5e-3f * (rand() / (float)RAND_MAX - 0.5f); hH[i] = 6e-2f * (rand() / (float)RAND_MAX - 0.5f); }
    hipMalloc(&W, 1024); hipMalloc(&H, 1024); hipMalloc(&o, 1024);
    hipMemcpy(W, hW, 1024, hipMemcpyHostToDevice); hipMemcpy(H, hH, 1024, hipMemcpyHostToDevice);
    for (float sw : { 1.0f, 256.0f, 4096.0f }) {
        k<<<1, 64>>>(W, H, o, sw); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
        double worst = 0, mag = 0;
        for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 4; r++) {
            const int j = lane & 15, i = 4 * (lane >> 4) + r; double s = 0;
            for (int kk = 0; kk < 16; kk++) s += (double)hW[i * 16 + kk] * hH[j * 16 + kk];
            worst = fmax(worst, fabs(s - ho[lane * 4 + r])); mag = fmax(mag, fabs(s));
        }
        printf("scale %g: max |err| = %.3e  (max |z| = %.3e)\n", sw, worst, mag);
    }
    return 0;
}
