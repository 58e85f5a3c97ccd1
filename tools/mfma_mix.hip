// Micro-benchmark: the hidden-layer inner loop of the inference kernel, feature by feature.
// 8 waves per workgroup (2 per SIMD), one workgroup per CU.  Per "tile": 8 k-steps x NPT MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }

// FLAGS: 1 = A from LDS (else registers), 2 = conversions (cvt + max per slice), 4 = barrier per 2 tiles, 8 = bias read per tile
template <int NPT, int FLAGS>
__global__ __launch_bounds__(512) void mix_kernel(const half8* __restrict__ in, float* __restrict__ out, int tiles, unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) char lds[80 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 80 * 1024 / 16; i += 512) ((half8*)lds)[i] = in[i & 0xffff];
    __syncthreads();
    half8 b[NPT][8], o[NPT][8];
    for (int p = 0; p < NPT; ++p) for (int k = 0; k < 8; ++k) b[p][k] = in[(threadIdx.x * 31 + p * 8 + k) & 0xffff], o[p][k] = b[p][k];
    f32x4 acc[2][NPT];
    for (int p = 0; p < NPT; ++p) acc[0][p] = f32x4{0, 0, 0, 0}, acc[1][p] = f32x4{0, 0, 0, 0};
    unsigned rd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + lane * 16);
    asm volatile("" : "+v"(rd));
    const char* rdp = (const char*)(__attribute__((address_space(3))) const char*)(uintptr_t)rd;
    half8 a[4];
    for (int i = 0; i < 4; ++i) a[i] = *(const half8*)(rdp + i * 1024);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles; t += 2) {
        sfor<0, 2>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            sfor<0, 8>([&](auto kc) {
                constexpr int k = decltype(kc)::value, F = T * 8 + k;
                const half8 av = a[F % 4];
                if constexpr (FLAGS & 1) a[F % 4] = *(const half8*)(rdp + ((F + 4) % 64) * 1024 + T * 16384);
                sfor<0, NPT>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    acc[T][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b[p][k], acc[T][p], 0, 0, 0);
                });
                if constexpr ((FLAGS & 2) != 0) {
                    if constexpr (k >= 1 && k - 1 < 2 * NPT) {
                        constexpr int S = k - 1, p = S >> 1, q = S & 1;
                        const f32x2 x = {acc[T ^ 1][p][2 * q], acc[T ^ 1][p][2 * q + 1]};
                        const half2v c = __builtin_convertvector(x, half2v);
                        unsigned r;
                        asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, c)));
                        half2v pr = __builtin_bit_cast(half2v, r);
                        o[p][T][2 * q] = pr[0], o[p][T][2 * q + 1] = pr[1];
                    }
                }
                if constexpr ((FLAGS & 8) != 0 && k == 5) {
                    const f32x4 bias = *(const f32x4*)(rdp + 70 * 1024 + T * 64 - lane * 16 + (lane >> 4) * 16);
                    sfor<0, NPT>([&](auto pc) { acc[T ^ 1][decltype(pc)::value] = bias; });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr ((FLAGS & 4) != 0) asm volatile("s_barrier" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int p = 0; p < NPT; ++p) for (int j = 0; j < 4; ++j) sum += acc[0][p][j] + acc[1][p][j] + (float)o[p][0][j] + (float)o[p][1][j];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = c1 - c0, stamps[2 * blockIdx.x + 1] = r1 - r0;
}

template <int NPT, int FLAGS>
static void run(const char* name, const half8* in, float* out, unsigned long long* stamps, int cus) {
    const int tiles = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 12; ++rep) mix_kernel<NPT, FLAGS><<<cus, 512>>>(in, out, tiles, stamps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) mix_kernel<NPT, FLAGS><<<cus, 512>>>(in, out, tiles, stamps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    std::vector<unsigned long long> h(2 * cus);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk(cus), cyc(cus);
    for (int i = 0; i < cus; ++i) clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0), cyc[i] = (double)h[2 * i];
    std::sort(clk.begin(), clk.end()), std::sort(cyc.begin(), cyc.end());
    const double mfma_per_simd = 2.0 * tiles * 8 * NPT;
    const double tflops = (double)cus * 8 * tiles * 8 * NPT * 16384.0 / (ms * 1e-3) / 1e12;
    // cycles from the launch time and the in-kernel clock (a single wave's stamps would flatter the
    // variants without a barrier, where the older wave of a SIMD finishes first)
    const double cycles = clk[cus / 2] * 1e9 * ms * 1e-3;
    printf("%-44s NPT=%d: %7.1f TFLOP/s  clock %.3f GHz  %.2f cycles per MFMA per SIMD (pipe busy %.3f)\n", name, NPT, tflops, clk[cus / 2],
           cycles / mfma_per_simd, 16.0 * mfma_per_simd / cycles);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    half8* in; float* out; unsigned long long* stamps;
    (void)hipMalloc(&in, 65536 * sizeof(half8)), (void)hipMalloc(&out, (size_t)cus * 512 * 4), (void)hipMalloc(&stamps, cus * 16);
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.25f);
    (void)hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<3, 0>("registers only", in, out, stamps, cus);
    run<3, 1>("+ A from LDS", in, out, stamps, cus);
    run<3, 3>("+ A from LDS + conversions", in, out, stamps, cus);
    run<3, 11>("+ A from LDS + conversions + bias", in, out, stamps, cus);
    run<3, 15>("+ A from LDS + conversions + bias + barrier", in, out, stamps, cus);
    run<2, 15>("+ A from LDS + conversions + bias + barrier", in, out, stamps, cus);
    run<3, 2>("conversions only", in, out, stamps, cus);
    return 0;
}
