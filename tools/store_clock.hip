// Can a persistent kernel get the store rate of request-order placement WITHOUT asking anybody for
// a place?  Same stores as tools/store_patterns.hip's paced kernels (persistent 256 x 8 waves, one
// KiB per wave and store, MFMAS matrix instructions between two stores of a wave, a workgroup tile =
// 10 "layers" of 8 x 16 KiB), only the address differs:
//   fixed     [tile][wave][layer][16 KiB]                           (what the training kernels do)
//   ticket    region of the workgroup's XCD, granule = next ticket of the region's counter, asked for
//             one layer ahead (store_patterns' "queues with a region each": the rate to reach)
//   clock     region of the workgroup's XCD, granule (k, j): j = the workgroup's number inside its
//             region, k = max(k of its previous layer + 1, (now - t0) / delta) from the chip-wide
//             100 MHz counter (s_memrealtime), t0 stamped by a one-thread kernel before the launch.
//             No atomics, no hand-off: workgroups that run late skip buckets (holes in the address
//             space, never written), workgroups that run early fall back to counting.  The buffer
//             holds HOLES times the dense bucket count; a k beyond it falls back to the fixed place
//             in a second buffer (counted).
//   clock/w   the same per WAVE: every wave reads the counter itself and takes slot (k, j, wave) of
//             RUN KiB; nothing is shared between the waves of a workgroup (no LDS word, no barrier).
// DRIFT > 0: workgroup b issues DRIFT * (b % 5) more MFMAs per store (workgroups out of step).
// Every variant records where each (tile, layer) went; the host checks that no two share a place.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/store_clock.hip -o tools/bin/store_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <functional>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int LAYERS = 10, RUN = 16, GRAN = 8 * RUN;  // KiB
constexpr int REGIONS = 8;

__global__ void stamp_kernel(unsigned long long* t0) { *t0 = __builtin_amdgcn_s_memrealtime(); }

struct Args {
    char* dst;              // placed area: REGIONS regions of region_granules granules
    char* fallback;         // fixed area, clock mode only
    long tiles;
    unsigned* counter;      // ticket mode: REGIONS counters, 4 KiB apart
    unsigned* table;        // [tile][LAYERS]: granule number in dst, or 0x80000000 | fixed granule in fallback
    unsigned* wtable;       // per-wave mode: [tile][LAYERS][8]: slot of RUN KiB in dst, or 0x80000000 | fixed slot
    unsigned region_granules;
    const unsigned long long* t0;
    unsigned delta;         // clock ticks (10 ns) per bucket
    unsigned* overflow;     // [gridDim]: layers that fell back
    unsigned* kmax;         // [gridDim]: last bucket used
    float* sink;
};

template <int MODE, int MFMAS, int DRIFT>
__global__ __launch_bounds__(512) void place_kernel(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 ha, hb;
    for (int i = 0; i < 8; ++i) ha[i] = (_Float16)(0.01f * (lane + i)), hb[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    __shared__ unsigned place_s[2];
    const unsigned region = blockIdx.x % REGIONS, j = blockIdx.x / REGIONS, per_region = gridDim.x / REGIONS;
    const int extra = DRIFT * (int)(blockIdx.x % 5);
    unsigned pending = 0;
    long k_prev = -1;
    unsigned fell = 0;
    const unsigned long long t0 = MODE >= 2 ? *a.t0 : 0ull;
    if (MODE == 1) {
        if (threadIdx.x == 0) pending = atomicAdd(a.counter + region * 1024, 1u);
    }
    int parity = 0;
    for (long tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        for (int l = 0; l < LAYERS; ++l) {
            if (MODE == 1 || MODE == 2) {
                if (threadIdx.x == 0) {
                    unsigned place;
                    if (MODE == 1) {
                        place = pending;
                        pending = atomicAdd(a.counter + region * 1024, 1u);  // the next layer's, a layer ahead
                        if (place >= a.region_granules) place = a.region_granules - 1;  // not reached
                        place += region * a.region_granules;
                    } else {
                        const long k_clock = (long)((__builtin_amdgcn_s_memrealtime() - t0) / a.delta);
                        long k = k_prev + 1 > k_clock ? k_prev + 1 : k_clock;
                        k_prev = k;
                        const unsigned long long g = (unsigned long long)k * per_region + j;
                        if (g < a.region_granules) place = (unsigned)g + region * a.region_granules;
                        else place = 0x80000000u | (unsigned)(tile * LAYERS + l), ++fell;
                    }
                    place_s[parity] = place;
                    a.table[tile * LAYERS + l] = place;
                }
                __syncthreads();
            }
            char* base;
            if (MODE == 0) base = a.dst + ((tile * 8 + wave) * LAYERS + l) * (long)RUN * 1024;
            else if (MODE == 3) {
                // wave-uniform, nothing shared: this wave's own reading of the counter
                const long k_clock = (long)((__builtin_amdgcn_s_memrealtime() - t0) / a.delta);
                const long k = k_prev + 1 > k_clock ? k_prev + 1 : k_clock;
                k_prev = k;
                const unsigned long long g = ((unsigned long long)k * per_region + j) * 8 + wave;
                const unsigned long long room = (unsigned long long)a.region_granules * 8;
                const long fixed = (tile * LAYERS + l) * 8 + wave;
                unsigned slot;
                if (g < room) slot = (unsigned)(g + region * room);
                else slot = 0x80000000u | (unsigned)fixed, ++fell;
                if (lane == 0) a.wtable[fixed] = slot;
                base = (slot & 0x80000000u) ? a.fallback + (long)(slot & 0x7fffffffu) * RUN * 1024
                                            : a.dst + (long)slot * RUN * 1024;
            } else {
                const unsigned place = place_s[parity];
                base = (place & 0x80000000u) ? a.fallback + ((long)(place & 0x7fffffffu) * GRAN + wave * RUN) * 1024
                                             : a.dst + ((long)place * GRAN + wave * RUN) * 1024;
            }
            for (int blk = 0; blk < RUN; ++blk) {
                __builtin_nontemporal_store(v, (f32x4*)(base + blk * 1024 + lane * 16));
#pragma unroll
                for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
                for (int m = 0; m < extra; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
            }
            parity ^= 1;
        }
    }
    if (MODE >= 2 && threadIdx.x == 0) a.overflow[blockIdx.x] = fell, a.kmax[blockIdx.x] = (unsigned)k_prev;
    if (acc[0] == 12345.f) a.sink[threadIdx.x] = acc[0];
}

static float run5(const char* name, double gb, const std::function<void()>& launch) {
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
    printf("%-86s %7.3f ms  %5.2f TB/s", name, ms / 5, gb / (ms / 5));
    fflush(stdout);
    return ms / 5;
}

int main() {
    const long tiles = 3072;  // 786,432 points
    const int grid = 256;
    const long dense = tiles * LAYERS;                       // granules of data
    const int HOLES = 3;                                     // address space of the placed area / data
    const unsigned region_granules = (unsigned)(dense * HOLES / REGIONS);
    const size_t placed_bytes = (size_t)region_granules * REGIONS * GRAN * 1024, fixed_bytes = (size_t)dense * GRAN * 1024;
    const double gb = fixed_bytes / 1e9;
    Args a{};
    unsigned long long* t0;
    if (hipMalloc(&a.dst, placed_bytes) != hipSuccess || hipMalloc(&a.fallback, fixed_bytes) != hipSuccess ||
        hipMalloc(&a.counter, REGIONS * 4096) != hipSuccess || hipMalloc(&a.table, dense * 4) != hipSuccess || hipMalloc(&a.wtable, dense * 8 * 4) != hipSuccess ||
        hipMalloc(&a.overflow, grid * 4) != hipSuccess || hipMalloc(&a.kmax, grid * 4) != hipSuccess ||
        hipMalloc(&a.sink, 4096) != hipSuccess || hipMalloc(&t0, 8) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    a.tiles = tiles, a.region_granules = region_granules, a.t0 = t0;
    printf("# %.2f GB per launch, %ld tiles x %d layers of %d KiB; placed area %.1f GB (%d x the data)\n", gb, tiles, LAYERS, GRAN,
           placed_bytes / 1e9, HOLES);
    std::vector<unsigned> table(dense), ov(grid), km(grid);
    auto check = [&](bool clock) {
        (void)hipMemcpy(table.data(), a.table, dense * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> s(table);
        std::sort(s.begin(), s.end());
        long dup = 0, oob = 0;
        for (long i = 1; i < dense; ++i) dup += s[i] == s[i - 1];
        for (long i = 0; i < dense; ++i)
            if (!(s[i] & 0x80000000u) && s[i] >= region_granules * REGIONS) ++oob;
        printf("   places: %ld duplicates, %ld out of range", dup, oob);
        if (clock) {
            (void)hipMemcpy(ov.data(), a.overflow, grid * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(km.data(), a.kmax, grid * 4, hipMemcpyDeviceToHost);
            long fell = 0;
            unsigned kmx = 0;
            for (int i = 0; i < grid; ++i) fell += ov[i], kmx = std::max(kmx, km[i]);
            printf(", %ld of %ld layers fell back, last bucket %u of %ld dense", fell, dense, kmx, dense / grid);
        }
        printf("\n");
        return dup == 0 && oob == 0;
    };
    std::vector<unsigned> wtable(dense * 8);
    auto check_waves = [&]() {
        (void)hipMemcpy(wtable.data(), a.wtable, dense * 8 * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> s(wtable);
        std::sort(s.begin(), s.end());
        long dup = 0, oob = 0, fb = 0;
        for (long i = 1; i < dense * 8; ++i) dup += s[i] == s[i - 1];
        for (long i = 0; i < dense * 8; ++i) {
            if (s[i] & 0x80000000u) ++fb;
            else if (s[i] >= (unsigned long long)region_granules * 8 * REGIONS) ++oob;
        }
        (void)hipMemcpy(km.data(), a.kmax, grid * 4, hipMemcpyDeviceToHost);
        unsigned kmx = 0;
        for (int i = 0; i < grid; ++i) kmx = std::max(kmx, km[i]);
        printf("   slots: %ld duplicates, %ld out of range, %ld of %ld fell back, last bucket (wave 0) %u of %ld dense\n", dup, oob, fb,
               dense * 8, kmx, dense / grid);
        return dup == 0 && oob == 0;
    };
    bool ok = true;
#define SET(M, D)                                                                                                   \
    {                                                                                                               \
        const float t_fixed = run5("fixed wave tiles, " #M " MFMA per store, drift " #D, gb, [&] { place_kernel<0, M, D><<<grid, 512>>>(a); }); \
        printf("\n");                                                                                               \
        run5("tickets per (tile, layer), one layer ahead, " #M " MFMA per store, drift " #D, gb, [&] {              \
            (void)hipMemsetAsync(a.counter, 0, REGIONS * 4096, 0);                                                  \
            place_kernel<1, M, D><<<grid, 512>>>(a); });                                                            \
        ok &= check(false);                                                                                         \
        for (int pct : {70, 85, 100, 115}) {                                                                        \
            /* bucket length = pct % of the time a layer takes at the fixed layout's rate */                       \
            a.delta = (unsigned)(t_fixed * 1e5 / (tiles / grid * LAYERS) * pct / 100);                              \
            if (a.delta < 1) a.delta = 1;                                                                           \
            char name[128];                                                                                         \
            snprintf(name, sizeof name, "clock buckets of %u ticks (%d %% of a layer), " #M " MFMA per store, drift " #D, a.delta, pct); \
            run5(name, gb, [&] { stamp_kernel<<<1, 1>>>(t0); place_kernel<2, M, D><<<grid, 512>>>(a); });           \
            ok &= check(true);                                                                                      \
            snprintf(name, sizeof name, "  the same per wave (nothing shared in a workgroup), %u ticks", a.delta);  \
            run5(name, gb, [&] { stamp_kernel<<<1, 1>>>(t0); place_kernel<3, M, D><<<grid, 512>>>(a); });           \
            ok &= check_waves();                                                                                    \
        }                                                                                                           \
    }
    SET(0, 0)
    SET(14, 0)
    SET(14, 1)
    SET(8, 2)
    printf(ok ? "# every placement gave each (tile, layer) a place of its own\n" : "# PLACEMENT ERROR\n");
    return ok ? 0 : 1;
}
