// Probe: operand/result layout of v_mfma_f32_16x16x16_f16 on gfx950 (A[i][k] = 16 i + k, B = identity under the assumed lane -> k map).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int lane = threadIdx.x, e = lane & 15, kg = lane >> 4;
    h4 a, b;
    for (int v = 0; v < 4; v++) { a[v] = (_Float16)(float)(16 * e + 4 * kg + v); b[v] = (_Float16)((4 * kg + v) == e ? 1.0f : 0.0f); }
    f4 c = { 0, 0, 0, 0 };
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 4; r++) { const int j = lane & 15, i = 4 * (lane >> 4) + r; if (h[lane * 4 + r] != 16 * i + j) bad++; }
    printf("bad=%d  lane0: %g %g %g %g  lane1: %g %g  lane16: %g %g\n", bad, h[0], h[1], h[2], h[3], h[4], h[5], h[64], h[65]);
    return 0;
}
