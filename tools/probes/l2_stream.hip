// tools/probes/l2_stream.hip -- what one CU (and all of them together) can pull out of L2 with 16-byte-per-lane loads: the question behind bwd_layer_kernel's
// tile fetch and generic_forward_kernel's weight streaming.  Every workgroup (512 threads, 1 per CU) reads the same `tile_kb` KiB region
// (cache-resident after the first pass) `reps` times, as plain global_load_dwordx4 into registers or as global_load_lds_dwordx4 into LDS.
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip && ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: loads into registers (summed), 1: LDS-DMA
__global__ __launch_bounds__(512, 1) void stream_kernel(const uint16_t* __restrict__ src, int tile_bytes, int reps, int distinct, unsigned* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = reinterpret_cast<const char*>(src) + (size_t)(distinct ? blockIdx.x : 0) * tile_bytes;
    u32x4 acc = { 0u, 0u, 0u, 0u };
    const int pieces = tile_bytes / 1024;   // 1 KiB per wave instruction
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        for (int p = wave; p < pieces; p += 8) {
            const char* a = base + (size_t)p * 1024 + lane * 16;
            if (MODE == 0) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(a);
                acc += v;
            } else {
                const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + (uint32_t)(p & 31) * 1024u;
                const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(a), "s"(d) : "memory");
            }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int tile_kb_list[] = { 40, 128 };
    uint16_t* src; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&src, (size_t)256 * 128 * 1024); hipMemset(src, 1, (size_t)256 * 128 * 1024);
    hipMalloc(&sink, 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int tile_kb : tile_kb_list)
        for (int distinct = 0; distinct < 2; distinct++)
            for (int mode = 0; mode < 2; mode++)
                for (int blocks : { 1, 8, 64, 256 }) {
                    const int reps = 200;
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    for (int w = 0; w < 2; w++) {
                        hipEventRecord(e0);
                        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(512), 0, 0, src, tile_kb * 1024, reps, distinct, sink, cyc);
                        else hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(512), 64 * 1024, 0, src, tile_kb * 1024, reps, distinct, sink, cyc);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                    }
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
                    double mean = 0; for (auto c : h) mean += (double)c; mean /= blocks;
                    // s_memtime ticks at 100 MHz on this part: bytes per core clock are derived from the event time and a nominal 2.1 GHz
                    const double bytes = (double)tile_kb * 1024 * reps;
                    printf("{\"tile_kb\": %d, \"distinct_tiles\": %d, \"mode\": \"%s\", \"workgroups\": %d, \"us\": %.1f, \"GBps_per_cu\": %.1f, \"B_per_clk_per_cu_at_2.1GHz\": %.1f, \"aggregate_TBps\": %.2f, \"memtime_ticks\": %.0f}\n",
                           tile_kb, distinct, mode ? "lds_dma" : "registers", blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9, bytes * blocks / (ms * 1e-3) / 1e12, mean);
                }
    return 0;
}
