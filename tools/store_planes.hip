// Where should a persistent kernel that saves 10 layer outputs per 256-point tile put them?  Same
// stores in every variant (4 GB of one-KiB streamed stores, 16 v_mfma_f32_16x16x32_f16 between two
// stores of a wave, 256 workgroups x 8 waves, tiles dealt round-robin); only the address differs:
//   tile-major   [tile][wave][layer][16 KiB]                      (a wave tile is one 160 KiB run)
//   plane-major  [layer][place][wave][16 KiB], place = tile       (fixed)
//   plane-major, place = the order in which the workgroups START their tiles (one ticket per tile
//                from one counter, taken at the start of the pass)
// DRIFT > 0: workgroup b issues DRIFT * (b % 5) more MFMAs per store, so that the workgroups do
// not stay in step (in the real kernels they do not either).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/store_planes.hip -o tools/bin/store_planes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int LAYERS = 10, RUN = 16;  // KiB blocks per wave and layer

template <int MODE, int MFMAS, int DRIFT>
__global__ __launch_bounds__(512) void planes_kernel(char* __restrict__ dst, long tiles, unsigned* __restrict__ counter,
                                                     unsigned* __restrict__ table, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    __shared__ unsigned place_s;
    const int extra = DRIFT * (int)(blockIdx.x % 5);
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        long place = tile;
        if (MODE == 2) {
            if (threadIdx.x == 0) {
                place_s = atomicAdd(counter, 1u);
                table[tile] = place_s;
            }
            __syncthreads();
            place = place_s < tiles ? place_s : tiles - 1;
            __syncthreads();
        }
        for (int l = 0; l < LAYERS; ++l)
            for (int blk = 0; blk < RUN; ++blk) {
                char* p = MODE == 0 ? dst + (((tile * 8 + wave) * LAYERS + l) * RUN + blk) * 1024
                                    : dst + (((l * tiles + place) * 8 + wave) * RUN + blk) * 1024;
                __builtin_nontemporal_store(v, (f32x4*)(p + lane * 16));
#pragma unroll
                for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
                for (int m = 0; m < extra; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            }
    }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

template <class F>
static void timed(const char* name, double gb, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-96s %7.3f ms  %5.2f TB/s\n", name, ms / 5, gb / (ms / 5));
    fflush(stdout);
}

int main() {
    const long tiles = 3072;  // 786,432 points
    const size_t bytes = (size_t)tiles * 8 * LAYERS * RUN * 1024;
    const double gb = bytes / 1e9;
    char* w;
    unsigned *counter, *table;
    float* sink;
    if (hipMalloc(&w, bytes) != hipSuccess || hipMalloc(&counter, 4096) != hipSuccess ||
        hipMalloc(&table, tiles * 4) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    printf("# %.2f GB per launch, %ld tiles\n", gb, tiles);
#define RUN3(M, D)                                                                                          \
    timed("tile-major, " #M " MFMA per store, drift " #D, gb, [&] { planes_kernel<0, M, D><<<256, 512>>>(w, tiles, counter, table, sink); }); \
    timed("plane-major, place = tile, " #M " MFMA per store, drift " #D, gb, [&] { planes_kernel<1, M, D><<<256, 512>>>(w, tiles, counter, table, sink); }); \
    timed("plane-major, place = ticket at tile start, " #M " MFMA per store, drift " #D, gb, [&] {          \
        (void)hipMemsetAsync(counter, 0, 4096, 0);                                                          \
        planes_kernel<2, M, D><<<256, 512>>>(w, tiles, counter, table, sink); });
    RUN3(0, 0)
    RUN3(16, 0)
    RUN3(16, 1)
    RUN3(16, 4)
    RUN3(8, 2)
    // the tickets of the last launch are a permutation of the tiles
    unsigned* host = (unsigned*)malloc(tiles * 4);
    (void)hipMemcpy(host, table, tiles * 4, hipMemcpyDeviceToHost);
    char* seen = (char*)calloc(tiles, 1);
    long bad = 0;
    for (long t = 0; t < tiles; ++t) {
        if (host[t] >= tiles || seen[host[t]]) ++bad; else seen[host[t]] = 1;
    }
    printf("# tickets: %s\n", bad ? "NOT a permutation" : "a permutation of the tiles");
    return bad != 0;
}
