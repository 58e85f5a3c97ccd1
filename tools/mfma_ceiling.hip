// What fp16 MFMA rate does this MI355X sustain?  A bare loop of v_mfma_f32_16x16x32_f16 (or
// 32x32x16) on random operands held in registers: no LDS, no memory traffic, every CU busy, one
// or two waves per SIMD.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_ceiling.hip -o gpurun_out/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, bool ZERO>
__global__ __launch_bounds__(512) void loop_kernel(const half8* __restrict__ in, float* __restrict__ out, int iters,
                                                   unsigned long long* __restrict__ stamps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(t * 8 + i) & 0xffff], b[i] = in[(t * 8 + 4 + i) & 0xffff];
        if (ZERO) a[i] = (half8)0, b[i] = (half8)0;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + j) & 3], b[j], acc[j], 0, 0, 0);
            }
        }
        for (int j = 0; j < 4; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + j) & 3], b[j], acc[j], 0, 0, 0);
            }
        }
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) sum += acc[j][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = sum;
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = c1 - c0, stamps[2 * blockIdx.x + 1] = r1 - r0;
}

template <int SHAPE, bool ZERO>
static void run(const char* name, int waves_per_simd, const half8* in, float* out, unsigned long long* stamps, int cus) {
    const int threads = 64 * 4 * waves_per_simd, blocks = cus;
    const int iters = 20000;
    const double flop_per_mfma = SHAPE == 16 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    const int mfma_per_iter = SHAPE == 16 ? 32 : 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 200; ++rep) loop_kernel<SHAPE, ZERO><<<blocks, threads>>>(in, out, iters, stamps);  // ~2 s of load first: the clock settles
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) loop_kernel<SHAPE, ZERO><<<blocks, threads>>>(in, out, iters, stamps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk(blocks);
    for (int i = 0; i < blocks; ++i) clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
    std::sort(clk.begin(), clk.end());
    const double waves = (double)blocks * threads / 64;
    const double tflops = waves * iters * mfma_per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma = clk[blocks / 2] * 1e9 * ms * 1e-3 / ((double)iters * mfma_per_iter * waves_per_simd);
    printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s  %.3f ms  clock %.3f GHz  %.2f cycles per MFMA per SIMD\n", name, waves_per_simd,
           tflops, ms, clk[blocks / 2], cyc_per_mfma);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    half8* in;
    float* out;
    unsigned long long* stamps;
    hipMalloc(&in, 65536 * sizeof(half8)), hipMalloc(&out, (size_t)cus * 512 * 4), hipMalloc(&stamps, cus * 16);
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("CUs: %d\n", cus);
    run<16, false>("16x16x32 f16, random", 1, in, out, stamps, cus);
    run<16, false>("16x16x32 f16, random", 2, in, out, stamps, cus);
    run<32, false>("32x32x16 f16, random", 1, in, out, stamps, cus);
    run<32, false>("32x32x16 f16, random", 2, in, out, stamps, cus);
    run<16, true>("16x16x32 f16, zeros", 2, in, out, stamps, cus);
    return 0;
}
