// Does the chip move more bytes per second when a reading kernel and a writing kernel run side by
// side than when either runs alone?  (The training step alternates write-bound kernels - saving
// forward, gradient chain - with a read-bound one, the weight-gradient kernel.)
// Persistent workgroups of 8 waves; a wave reads or writes its own run of one-KiB blocks (16 B per
// lane, streamed), tile after tile, in the training kernels' pattern.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/rw_mix.hip -o tools/bin/rw_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 160;

__global__ __launch_bounds__(512) void write_kernel(char* __restrict__ dst, long wave_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8)
        for (int blk = 0; blk < BLOCKS; ++blk)
            __builtin_nontemporal_store(v, (f32x4*)(dst + (wt * BLOCKS + blk) * 1024 + lane * 16));
}

__global__ __launch_bounds__(512) void read_kernel(const char* __restrict__ src, long wave_tiles, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8)
#pragma unroll 8
        for (int blk = 0; blk < BLOCKS; ++blk)
            acc += __builtin_nontemporal_load((const f32x4*)(src + (wt * BLOCKS + blk) * 1024 + lane * 16));
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[threadIdx.x] = acc[0];
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const long tiles_w = 24576, tiles_r = 24576;  // 4.0 GB each
    const double gb = (double)tiles_w * BLOCKS * 1024 / 1e9;
    char *w, *r;
    float* sink;
    (void)hipMalloc(&w, (size_t)tiles_w * BLOCKS * 1024), (void)hipMalloc(&r, (size_t)tiles_r * BLOCKS * 1024);
    (void)hipMalloc(&sink, 4096);
    (void)hipMemset(r, 1, (size_t)tiles_r * BLOCKS * 1024);
    hipStream_t s0, s1;
    (void)hipStreamCreate(&s0), (void)hipStreamCreate(&s1);
    const int reps = 5;
    auto timed = [&](const char* name, int wg_w, int wg_r, double bytes_gb) {
        for (int pass = 0; pass < 2; ++pass) {  // first pass warms up
            (void)hipDeviceSynchronize();
            const double t0 = now_ms();
            for (int i = 0; i < reps; ++i) {
                if (wg_w) write_kernel<<<wg_w, 512, 0, s0>>>(w, tiles_w);
                if (wg_r) read_kernel<<<wg_r, 512, 0, s1>>>(r, tiles_r, sink);
            }
            (void)hipDeviceSynchronize();
            const double ms = (now_ms() - t0) / reps;
            if (pass) printf("%-44s %7.3f ms  %5.2f TB/s\n", name, ms, bytes_gb / ms);
        }
    };
    timed("write alone, 256 workgroups", 256, 0, gb);
    timed("read alone, 256 workgroups", 0, 256, gb);
    timed("read alone, 512 workgroups", 0, 512, gb);
    timed("write 256 + read 256 side by side", 256, 256, 2 * gb);
    timed("write 128 + read 128 side by side", 128, 128, 2 * gb);
    timed("write 256 + read 512 side by side", 256, 512, 2 * gb);
    return 0;
}
