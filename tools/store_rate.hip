// How fast can the chip take the training kernels' store pattern, and does the rate follow the
// shader clock?  Persistent workgroups (one per CU, 8 waves); every wave writes its own "wave
// tile" of BLOCKS one-KiB blocks (16 B per lane, streamed), tile after tile, like the saving
// forward does.  MFMAS > 0: that many v_mfma_f32_16x16x32_f16 on random operands between two
// stores (lowers the clock the chip holds, as the real kernels' matrix work does).
// FILL: the workgroups sweep memory in dispatch order instead (8 KiB per workgroup and step).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/store_rate.hip -o tools/bin/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 167;

template <int MFMAS, bool FILL>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ dst, const half8* __restrict__ in, long wave_tiles,
                                                    unsigned long long* __restrict__ stamps, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a = in[threadIdx.x & 0xffff], b = in[(threadIdx.x * 7 + 3) & 0xffff];
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const long tiles_per_pass = (long)gridDim.x * 8;
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += tiles_per_pass) {
        for (int blk = 0; blk < BLOCKS; ++blk) {
            char* p;
            if (FILL) {  // pass-major sweep: all workgroups write one contiguous window per block step
                const long pass = wt / tiles_per_pass;
                p = dst + ((pass * BLOCKS + blk) * tiles_per_pass + (wt % tiles_per_pass)) * 1024 + lane * 16;
            } else {
                p = dst + (wt * BLOCKS + blk) * 1024 + lane * 16;
            }
            __builtin_nontemporal_store(a, (half8*)p);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = c1 - c0, stamps[2 * blockIdx.x + 1] = r1 - r0;
}

template <int MFMAS, bool FILL>
static void run(const char* name, char* dst, const half8* in, long wave_tiles, unsigned long long* stamps, float* sink, int cus) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 30; ++rep) store_kernel<MFMAS, FILL><<<cus, 512>>>(dst, in, wave_tiles, stamps, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) store_kernel<MFMAS, FILL><<<cus, 512>>>(dst, in, wave_tiles, stamps, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<unsigned long long> h(2 * cus);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk(cus);
    for (int i = 0; i < cus; ++i) clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
    std::sort(clk.begin(), clk.end());
    const double bytes = (double)wave_tiles * BLOCKS * 1024;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s %d MFMA per store: %6.2f TB/s  %.3f ms  clock %.3f GHz  %.2f B per clock per CU\n", name, MFMAS, tbs, ms, clk[cus / 2],
           bytes / (ms * 1e-3) / cus / (clk[cus / 2] * 1e9));
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const long wave_tiles = 24576;  // 786,432 points: the fine network's share of a 4096-ray step (4.2 GB)
    char* dst; half8* in; unsigned long long* stamps; float* sink;
    (void)hipMalloc(&dst, (size_t)wave_tiles * BLOCKS * 1024), (void)hipMalloc(&in, 65536 * sizeof(half8));
    (void)hipMalloc(&stamps, cus * 16), (void)hipMalloc(&sink, 4096);
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f);
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0, false>("wave-tile pattern", dst, in, wave_tiles, stamps, sink, cus);
    run<4, false>("wave-tile pattern", dst, in, wave_tiles, stamps, sink, cus);
    run<8, false>("wave-tile pattern", dst, in, wave_tiles, stamps, sink, cus);
    run<14, false>("wave-tile pattern", dst, in, wave_tiles, stamps, sink, cus);
    run<0, true>("pass-major sweep", dst, in, wave_tiles, stamps, sink, cus);
    run<8, true>("pass-major sweep", dst, in, wave_tiles, stamps, sink, cus);
    run<14, true>("pass-major sweep", dst, in, wave_tiles, stamps, sink, cus);
    return 0;
}
