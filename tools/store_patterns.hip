// Which property of a store pattern sets the rate the chip takes it at?  All variants write the
// same 4 GB with 16-byte-per-lane streamed stores (one KiB per wave and instruction).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/store_patterns.hip -o tools/bin/store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 160;

// A wave owns a run of BLOCKS KiB ("wave tile"); workgroups stride over the tiles.
template <bool NT>
__global__ __launch_bounds__(512) void tile_kernel(char* __restrict__ dst, long wave_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long wt = (long)blockIdx.x * waves + wave; wt < wave_tiles; wt += (long)gridDim.x * waves)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            f32x4* p = (f32x4*)(dst + (wt * BLOCKS + blk) * 1024 + lane * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
}

// The chip sweeps memory front to back: step s of a workgroup writes the KiB blocks
// [(s * gridDim + blockIdx) * waves + wave].
template <bool NT>
__global__ __launch_bounds__(512) void sweep_kernel(char* __restrict__ dst, long kib) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long b = (long)blockIdx.x * waves + wave; b < kib; b += (long)gridDim.x * waves) {
        f32x4* p = (f32x4*)(dst + b * 1024 + lane * 16);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}

// Same sweep, but a wave writes RUN consecutive KiB per step.
template <int RUN>
__global__ __launch_bounds__(512) void sweep_run_kernel(char* __restrict__ dst, long kib) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long b = ((long)blockIdx.x * waves + wave) * RUN; b < kib; b += (long)gridDim.x * waves * RUN)
#pragma unroll
        for (int r = 0; r < RUN; ++r) __builtin_nontemporal_store(v, (f32x4*)(dst + (b + r) * 1024 + lane * 16));
}

// Workgroups take the next CHUNK KiB of the sweep from a counter (dispatch order without new waves).
template <int CHUNK>
__global__ __launch_bounds__(512) void dynamic_kernel(char* __restrict__ dst, long kib, unsigned* __restrict__ counter) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    __shared__ unsigned next;
    for (;;) {
        if (threadIdx.x == 0) next = atomicAdd(counter, 1u);
        __syncthreads();
        const long base = (long)next * CHUNK;
        __syncthreads();
        if (base >= kib) break;
#pragma unroll
        for (int r = 0; r < CHUNK / 8; ++r)
            __builtin_nontemporal_store(v, (f32x4*)(dst + (base + r * 8 + wave) * 1024 + lane * 16));
    }
}

// Short-lived waves in dispatch order, RUN KiB each (consecutive).
template <int RUN>
__global__ __launch_bounds__(256) void oneshot_kernel(char* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    const long b = ((long)blockIdx.x * 4 + wave) * RUN;
#pragma unroll 8
    for (int r = 0; r < RUN; ++r) __builtin_nontemporal_store(v, (f32x4*)(dst + (b + r) * 1024 + lane * 16));
}

// Wave tiles from persistent workgroups, at most DEPTH + 1 stores of a wave in flight.
template <int DEPTH>
__global__ __launch_bounds__(1024) void throttled_kernel(char* __restrict__ dst, long wave_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long wt = (long)blockIdx.x * waves + wave; wt < wave_tiles; wt += (long)gridDim.x * waves)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            __builtin_nontemporal_store(v, (f32x4*)(dst + (wt * BLOCKS + blk) * 1024 + lane * 16));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        }
}

// Persistent workgroups that write wave tiles at the pace of a compute kernel (MFMAS matrix
// instructions between two stores), into memory handed out in the order it is asked for: a
// workgroup takes GRAN KiB at a time from a counter (one request ahead of its use) and its eight
// waves fill it block step by block step, so that at any moment the chip writes one moving window
// of about gridDim * GRAN KiB whatever the phase of each workgroup.  GRAN = 0: the fixed layout.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int GRAN, int MFMAS, int COUNTERS = 1>
__global__ __launch_bounds__(512) void paced_kernel(char* __restrict__ dst, long wave_tiles, unsigned* __restrict__ counter,
                                                    unsigned granules, float* __restrict__ sink) {
    static_assert(GRAN == 0 || (GRAN % 8 == 0 && (BLOCKS * 8) % GRAN == 0), "a workgroup round is a whole number of granules");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    __shared__ unsigned base[2];
    constexpr int kSteps = GRAN > 0 ? GRAN / 8 : 1;  // block steps per granule
    int parity = 0;
    // COUNTERS request queues (4 KiB apart), workgroup i uses queue i % COUNTERS, whose k-th granule is
    // number k * COUNTERS + i % COUNTERS: the queues advance together as long as the workgroups do.
    const unsigned queue = blockIdx.x % COUNTERS;
    counter += queue * 1024;
    if (GRAN > 0) {
        if (threadIdx.x == 0) base[0] = atomicAdd(counter, 1u);
        __syncthreads();
    }
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8) {
        for (int blk = 0; blk < BLOCKS; ++blk) {
            char* p;
            if (GRAN > 0) {
                const bool more = wt + (long)gridDim.x * 8 < wave_tiles || blk + kSteps < BLOCKS;
                if (blk % kSteps == 0 && more && threadIdx.x == 0) base[parity ^ 1] = atomicAdd(counter, 1u);  // the next granule
                unsigned g = base[parity] * COUNTERS + queue;
                if (g >= granules) g = granules - 1;  // not reached: exactly `granules` requests are made
                p = dst + ((long)g * GRAN + (blk % kSteps) * 8 + wave) * 1024 + lane * 16;
            } else {
                p = dst + (wt * BLOCKS + blk) * 1024 + lane * 16;
            }
            __builtin_nontemporal_store(v, (f32x4*)p);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            if (GRAN > 0 && blk % kSteps == kSteps - 1) {
                __syncthreads();
                parity ^= 1;
            }
        }
    }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

// The paced kernel with the request for a granule made two granules before its use, so that the
// counter's latency is off the workgroup's path (the buffer needs 2 * gridDim granules of slack:
// every workgroup ends with two requests it does not use).
// REGIONS: queue q hands out the granules of its own contiguous 1 / COUNTERS of the buffer
// (otherwise granule k of queue q is number k * COUNTERS + q).
// WAVE_MAJOR: inside a granule a wave's blocks are consecutive (GRAN / 8 KiB per wave) instead of
// the eight waves' blocks of one step.
template <int GRAN, int MFMAS, int COUNTERS, bool REGIONS = false, bool WAVE_MAJOR = false>
__global__ __launch_bounds__(512) void paced2_kernel(char* __restrict__ dst, long wave_tiles, unsigned* __restrict__ counter,
                                                     unsigned granules_with_slack, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    __shared__ unsigned base[2];
    constexpr int kSteps = GRAN / 8;
    static_assert(GRAN % 8 == 0 && BLOCKS % kSteps == 0, "a workgroup round is a whole number of granules");
    const unsigned queue = blockIdx.x % COUNTERS;
    counter += queue * 1024;
    unsigned pending = 0;
    if (threadIdx.x == 0) base[0] = atomicAdd(counter, 1u), pending = atomicAdd(counter, 1u);
    __syncthreads();
    int parity = 0;
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            if (blk % kSteps == 0 && threadIdx.x == 0) base[parity ^ 1] = pending, pending = atomicAdd(counter, 1u);
            unsigned g = REGIONS ? queue * (granules_with_slack / COUNTERS) + base[parity] : base[parity] * COUNTERS + queue;
            if (g >= granules_with_slack) g = granules_with_slack - 1;  // not reached
            const int in_granule = WAVE_MAJOR ? wave * kSteps + blk % kSteps : (blk % kSteps) * 8 + wave;
            __builtin_nontemporal_store(v, (f32x4*)(dst + ((long)g * GRAN + in_granule) * 1024 + lane * 16));
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            if (blk % kSteps == kSteps - 1) {
                __syncthreads();
                parity ^= 1;
            }
        }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

// The same placement without a counter: queue q (workgroups q, q + COUNTERS, ...) owns a region,
// and inside it granule step t of the j-th workgroup of the queue is granule t * n + j: in time
// order as long as the n workgroups of a queue stay in step, which nothing enforces.
template <int GRAN, int MFMAS, int COUNTERS>
__global__ __launch_bounds__(512) void fixed_region_kernel(char* __restrict__ dst, long wave_tiles, unsigned granules,
                                                           float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    constexpr int kSteps = GRAN / 8, kPerRound = BLOCKS / kSteps;
    static_assert(GRAN % 8 == 0 && BLOCKS % kSteps == 0, "a workgroup round is a whole number of granules");
    const unsigned queue = blockIdx.x % COUNTERS, j = blockIdx.x / COUNTERS, n = gridDim.x / COUNTERS;
    unsigned round = 0;
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8, ++round)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            unsigned g = queue * (granules / COUNTERS) + (round * kPerRound + blk / kSteps) * n + j;
            if (g >= granules) g = granules - 1;  // not reached
            __builtin_nontemporal_store(v, (f32x4*)(dst + ((long)g * GRAN + (blk % kSteps) * 8 + wave) * 1024 + lane * 16));
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

// Every wave asks for its own runs of RUN KiB (no workgroup-level hand-off): lane 0 adds RUN to the
// counter of the workgroup's region one run before the run is used.
template <int RUN, int MFMAS, int COUNTERS>
__global__ __launch_bounds__(512) void wave_runs_kernel(char* __restrict__ dst, long wave_tiles, unsigned* __restrict__ counter,
                                                        unsigned region_kib, float* __restrict__ sink) {
    static_assert(BLOCKS % RUN == 0, "a wave tile is a whole number of runs");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    const unsigned queue = blockIdx.x % COUNTERS;
    counter += queue * 1024;
    unsigned cur = 0, nxt = 0;
    if (lane == 0) cur = atomicAdd(counter, (unsigned)RUN);
    cur = __builtin_amdgcn_readfirstlane(cur);
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            const bool more = wt + (long)gridDim.x * 8 < wave_tiles || blk + RUN < BLOCKS;
            if (blk % RUN == 0 && more && lane == 0) nxt = atomicAdd(counter, (unsigned)RUN);
            unsigned kib = cur + blk % RUN;
            if (kib >= region_kib) kib = region_kib - 1;  // not reached
            __builtin_nontemporal_store(v, (f32x4*)(dst + ((long)queue * region_kib + kib) * 1024 + lane * 16));
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            if (blk % RUN == RUN - 1) cur = __builtin_amdgcn_readfirstlane(nxt);
        }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

// Short-lived waves, one KiB each; the workgroup's place in memory is its index with the low
// SHUFFLE bits reversed (SHUFFLE = 0: dispatch order), so neighbours in time are 2^SHUFFLE chunks apart.
template <int SHUFFLE>
__global__ void shuffled_kernel(char* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    unsigned id = blockIdx.x;
    if (SHUFFLE > 0) {
        const unsigned low = id & ((1u << SHUFFLE) - 1);
        id = (id & ~((1u << SHUFFLE) - 1)) | (__brev(low) >> (32 - SHUFFLE));
    }
    __builtin_nontemporal_store(v, (f32x4*)(dst + ((long)id * waves + wave) * 1024 + lane * 16));
}

// Short-lived 4-wave workgroups, 4 KiB each; the low 12 bits of the workgroup's index are rotated
// left by ROT, so workgroups that follow each other in time are 2^ROT chunks apart inside a 16 MiB
// window.
template <int ROT>
__global__ void rotated_kernel(char* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    const unsigned id = blockIdx.x, low = id & 4095u;
    const unsigned rot = ((low << ROT) | (low >> (12 - ROT))) & 4095u;
    __builtin_nontemporal_store(v, (f32x4*)(dst + ((long)((id & ~4095u) | rot) * 4 + wave) * 1024 + lane * 16));
}

// Persistent wave tiles whose one-KiB blocks are cut into pieces of LANES * 16 bytes, each piece in
// a different 4 KiB chunk: block b of a tile puts piece k at chunk k of its group of 64 / LANES
// chunks, so that every store instruction is spread over 64 / LANES chunks and the group fills
// completely after 4096 / (LANES * 16) blocks.  LANES = 64: whole blocks (the plain layout).
template <int LANES, int MFMAS>
__global__ __launch_bounds__(512) void spread_kernel(char* __restrict__ dst, long wave_tiles, float* __restrict__ sink) {
    constexpr int kPieces = 64 / LANES, kPiece = LANES * 16, kPerGroup = 4096 / kPiece;  // blocks per group
    static_assert(BLOCKS % kPerGroup == 0, "a tile is a whole number of groups");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * (lane - i));
    f32x4 acc = {0, 0, 0, 0};
    const long lane_off = (long)(lane / LANES) * 4096 + (lane % LANES) * 16;
    for (long wt = (long)blockIdx.x * 8 + wave; wt < wave_tiles; wt += (long)gridDim.x * 8)
        for (int blk = 0; blk < BLOCKS; ++blk) {
            char* p = dst + wt * BLOCKS * 1024 + (long)(blk / kPerGroup) * (kPieces * 4096) + (blk % kPerGroup) * kPiece + lane_off;
            __builtin_nontemporal_store(v, (f32x4*)p);
#pragma unroll
            for (int m = 0; m < MFMAS; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    if (acc[0] == 12345.f) sink[threadIdx.x] = acc[0];
}

template <class F>
static void timed(const char* name, double gb, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.3f ms  %5.2f TB/s\n", name, ms / 5, gb / (ms / 5));
}

int main() {
    const long tiles = 24576, kib = tiles * BLOCKS;
    const double gb = (double)kib * 1024 / 1e9;
    char* w;
    (void)hipMalloc(&w, (size_t)kib * 1024 + ((size_t)64 << 20));  // 64 MiB of slack for paced2_kernel
    timed("wave tiles, persistent 256 x 8 waves, nt", gb, [&] { tile_kernel<true><<<256, 512>>>(w, tiles); });
    timed("wave tiles, persistent 256 x 8 waves, plain", gb, [&] { tile_kernel<false><<<256, 512>>>(w, tiles); });
    timed("wave tiles, persistent 512 x 8 waves, nt", gb, [&] { tile_kernel<true><<<512, 512>>>(w, tiles); });
    timed("wave tiles, persistent 1024 x 8 waves, nt", gb, [&] { tile_kernel<true><<<1024, 512>>>(w, tiles); });
    timed("wave tiles, one workgroup per 8 tiles (3072 workgroups), nt", gb, [&] { tile_kernel<true><<<tiles / 8, 512>>>(w, tiles); });
    timed("wave tiles, persistent 256 x 4 waves, nt", gb, [&] { tile_kernel<true><<<256, 256>>>(w, tiles); });
    timed("sweep, persistent 256 x 8 waves, nt", gb, [&] { sweep_kernel<true><<<256, 512>>>(w, kib); });
    timed("sweep, persistent 256 x 8 waves, plain", gb, [&] { sweep_kernel<false><<<256, 512>>>(w, kib); });
    timed("sweep, persistent 1024 x 8 waves, nt", gb, [&] { sweep_kernel<true><<<1024, 512>>>(w, kib); });
    timed("sweep, one KiB per wave, one workgroup per 4 KiB (grid = size), nt", gb, [&] { sweep_kernel<true><<<kib / 4, 256>>>(w, kib); });
    timed("sweep, one KiB per wave, grid = size, plain", gb, [&] { sweep_kernel<false><<<kib / 4, 256>>>(w, kib); });
    timed("sweep, 4 KiB runs per wave, persistent 256 x 8", gb, [&] { sweep_run_kernel<4><<<256, 512>>>(w, kib); });
    timed("sweep, 16 KiB runs per wave, persistent 256 x 8", gb, [&] { sweep_run_kernel<16><<<256, 512>>>(w, kib); });
    timed("sweep, 16 KiB runs per wave, persistent 1024 x 8", gb, [&] { sweep_run_kernel<16><<<1024, 512>>>(w, kib); });
    timed("wave tiles, 256 x 8 waves, 1 store in flight per wave", gb, [&] { throttled_kernel<0><<<256, 512>>>(w, tiles); });
    timed("wave tiles, 256 x 8 waves, 2 stores in flight per wave", gb, [&] { throttled_kernel<1><<<256, 512>>>(w, tiles); });
    timed("wave tiles, 256 x 8 waves, 4 stores in flight per wave", gb, [&] { throttled_kernel<3><<<256, 512>>>(w, tiles); });
    timed("wave tiles, 256 x 8 waves, 8 stores in flight per wave", gb, [&] { throttled_kernel<7><<<256, 512>>>(w, tiles); });
    timed("wave tiles, 256 x 8 waves, 16 stores in flight per wave", gb, [&] { throttled_kernel<15><<<256, 512>>>(w, tiles); });
    timed("wave tiles, 256 x 16 waves, 1 store in flight per wave", gb, [&] { throttled_kernel<0><<<256, 1024>>>(w, tiles); });
    timed("wave tiles, 256 x 16 waves, 2 stores in flight per wave", gb, [&] { throttled_kernel<1><<<256, 1024>>>(w, tiles); });
    timed("wave tiles, 512 x 16 waves, 1 store in flight per wave", gb, [&] { throttled_kernel<0><<<512, 1024>>>(w, tiles); });
    timed("wave tiles, 512 x 16 waves, 2 stores in flight per wave", gb, [&] { throttled_kernel<1><<<512, 1024>>>(w, tiles); });
    timed("one-shot waves, 1 KiB each, workgroup = 1 wave", gb, [&] { shuffled_kernel<0><<<kib, 64>>>(w); });
    timed("one-shot waves, 1 KiB each, workgroup = 8 waves", gb, [&] { shuffled_kernel<0><<<kib / 8, 512>>>(w); });
    timed("one-shot waves, 1 KiB each, workgroup = 16 waves", gb, [&] { shuffled_kernel<0><<<kib / 16, 1024>>>(w); });
    timed("one-shot, 4-wave workgroups shuffled within 64 KiB", gb, [&] { shuffled_kernel<4><<<kib / 4, 256>>>(w); });
    timed("one-shot, 4-wave workgroups shuffled within 1 MiB", gb, [&] { shuffled_kernel<8><<<kib / 4, 256>>>(w); });
    timed("one-shot, 4-wave workgroups shuffled within 16 MiB", gb, [&] { shuffled_kernel<12><<<kib / 4, 256>>>(w); });
    timed("one-shot, 4-wave workgroups shuffled within 256 MiB", gb, [&] { shuffled_kernel<16><<<kib / 4, 256>>>(w); });
#define ROTATED(R) timed("one-shot 4 KiB chunks, neighbours in time 2^" #R " chunks apart", gb, [&] { rotated_kernel<R><<<kib / 4, 256>>>(w); })
    ROTATED(0); ROTATED(1); ROTATED(2); ROTATED(3); ROTATED(4); ROTATED(5); ROTATED(6); ROTATED(7); ROTATED(8); ROTATED(9); ROTATED(10); ROTATED(11);
    unsigned* counter;
    float* sink;
    (void)hipMalloc(&counter, 64 * 4096), (void)hipMalloc(&sink, 4096);
#define QUEUED(G, M, C)                                                                                      \
    timed("paced, " #M " MFMA per store, granule " #G " KiB from " #C " queues", gb, [&] {                    \
        (void)hipMemsetAsync(counter, 0, 64 * 4096, 0);                                                      \
        paced_kernel<G, M, C><<<256, 512>>>(w, tiles, counter, (unsigned)(kib / G), sink);                   \
    })
#define AHEAD(G, M, C)                                                                                       \
    timed("paced, " #M " MFMA per store, granule " #G " KiB from " #C " queues, asked for 2 ahead", gb, [&] { \
        (void)hipMemsetAsync(counter, 0, 64 * 4096, 0);                                                      \
        paced2_kernel<G, M, C><<<256, 512>>>(w, tiles, counter, (unsigned)(kib / G) + 512u, sink);            \
    })
#define REGION(G, M, C)                                                                                      \
    timed("paced, " #M " MFMA per store, granule " #G " KiB, " #C " queues with a region each, 2 ahead", gb, [&] { \
        (void)hipMemsetAsync(counter, 0, 64 * 4096, 0);                                                      \
        paced2_kernel<G, M, C, true><<<256, 512>>>(w, tiles, counter, (unsigned)(kib / G) + 512u, sink);      \
    })
#define FIXED(G, M, C)                                                                                       \
    timed("paced, " #M " MFMA per store, granule " #G " KiB, " #C " regions, fixed placement", gb, [&] {      \
        fixed_region_kernel<G, M, C><<<256, 512>>>(w, tiles, (unsigned)(kib / G), sink);                      \
    })
    FIXED(8, 0, 8); FIXED(16, 0, 8); FIXED(32, 0, 8); FIXED(64, 0, 8); FIXED(128, 0, 8); FIXED(32, 0, 1); FIXED(32, 0, 32); FIXED(64, 0, 32);
    FIXED(8, 14, 8); FIXED(16, 14, 8); FIXED(32, 14, 8); FIXED(64, 14, 8); FIXED(128, 14, 8); FIXED(32, 14, 1); FIXED(32, 14, 32); FIXED(64, 14, 32);
#define REGIONW(G, M, C)                                                                                     \
    timed("paced, " #M " MFMA per store, granule " #G " KiB wave-major, " #C " queues with a region each, 2 ahead", gb, [&] { \
        (void)hipMemsetAsync(counter, 0, 64 * 4096, 0);                                                      \
        paced2_kernel<G, M, C, true, true><<<256, 512>>>(w, tiles, counter, (unsigned)(kib / G) + 512u, sink); \
    })
#define WAVERUNS(R, M, C)                                                                                    \
    timed("paced, " #M " MFMA per store, every wave its own runs of " #R " KiB, " #C " regions", gb, [&] {   \
        (void)hipMemsetAsync(counter, 0, 64 * 4096, 0);                                                      \
        wave_runs_kernel<R, M, C><<<256, 512>>>(w, tiles, counter, (unsigned)(kib / C), sink);               \
    })
    WAVERUNS(4, 14, 8); WAVERUNS(8, 14, 8); WAVERUNS(16, 14, 8); WAVERUNS(32, 14, 8); WAVERUNS(16, 14, 32); WAVERUNS(16, 14, 1);
    WAVERUNS(16, 8, 8); WAVERUNS(16, 0, 8); WAVERUNS(8, 0, 8);
    REGIONW(16, 14, 8); REGIONW(32, 14, 8); REGIONW(64, 14, 8); REGIONW(128, 14, 8); REGIONW(32, 14, 32); REGIONW(64, 14, 32);
    REGIONW(32, 8, 8); REGIONW(64, 8, 8); REGION(32, 8, 8); REGION(64, 8, 8); REGION(32, 8, 32);
    REGION(8, 0, 8); REGION(16, 0, 8); REGION(32, 0, 8); REGION(64, 0, 8); REGION(128, 0, 8); REGION(16, 0, 32); REGION(32, 0, 32); REGION(64, 0, 32);
    REGION(8, 14, 8); REGION(16, 14, 8); REGION(32, 14, 8); REGION(64, 14, 8); REGION(128, 14, 8); REGION(16, 14, 32); REGION(32, 14, 32); REGION(64, 14, 32);
    AHEAD(16, 0, 8); AHEAD(32, 0, 8); AHEAD(64, 0, 8); AHEAD(128, 0, 8); AHEAD(32, 0, 1); AHEAD(64, 0, 1);
    AHEAD(16, 14, 8); AHEAD(32, 14, 8); AHEAD(64, 14, 8); AHEAD(128, 14, 8); AHEAD(64, 14, 1);
    QUEUED(16, 0, 8); QUEUED(32, 0, 8); QUEUED(64, 0, 8); QUEUED(128, 0, 8);
    QUEUED(16, 0, 32); QUEUED(32, 0, 32); QUEUED(64, 0, 32);
    QUEUED(16, 14, 8); QUEUED(32, 14, 8); QUEUED(64, 14, 8); QUEUED(128, 14, 8);
    QUEUED(16, 14, 32); QUEUED(32, 14, 32); QUEUED(64, 14, 32);
#define PACED(G, M)                                                                                          \
    timed("paced, " #M " MFMA per store, granule " #G " KiB (0 = fixed wave tiles)", gb, [&] {               \
        (void)hipMemsetAsync(counter, 0, 4, 0);                                                              \
        paced_kernel<G, M><<<256, 512>>>(w, tiles, counter, G ? (unsigned)(kib / (G ? G : 1)) : 0u, sink);   \
    })
#define SPREAD(L, M) timed("wave tiles, " #M " MFMA per store, pieces of " #L " lanes in separate 4 KiB chunks", gb, [&] { spread_kernel<L, M><<<256, 512>>>(w, tiles, sink); })
    SPREAD(64, 0); SPREAD(32, 0); SPREAD(16, 0); SPREAD(8, 0);
    SPREAD(64, 14); SPREAD(32, 14); SPREAD(16, 14); SPREAD(8, 14);
    PACED(0, 0); PACED(64, 0); PACED(128, 0); PACED(256, 0); PACED(640, 0);
    PACED(0, 8); PACED(64, 8); PACED(128, 8); PACED(256, 8); PACED(640, 8);
    PACED(0, 14); PACED(64, 14); PACED(128, 14); PACED(256, 14);
    timed("dynamic 8 KiB chunks, persistent 256 x 8", gb, [&] { (void)hipMemsetAsync(counter, 0, 4, 0); dynamic_kernel<8><<<256, 512>>>(w, kib, counter); });
    timed("dynamic 64 KiB chunks, persistent 256 x 8", gb, [&] { (void)hipMemsetAsync(counter, 0, 4, 0); dynamic_kernel<64><<<256, 512>>>(w, kib, counter); });
    timed("dynamic 64 KiB chunks, persistent 512 x 8", gb, [&] { (void)hipMemsetAsync(counter, 0, 4, 0); dynamic_kernel<64><<<512, 512>>>(w, kib, counter); });
    timed("dynamic 512 KiB chunks, persistent 256 x 8", gb, [&] { (void)hipMemsetAsync(counter, 0, 4, 0); dynamic_kernel<512><<<256, 512>>>(w, kib, counter); });
    timed("one-shot waves, 2 KiB each", gb, [&] { oneshot_kernel<2><<<kib / 8, 256>>>(w); });
    timed("one-shot waves, 4 KiB each", gb, [&] { oneshot_kernel<4><<<kib / 16, 256>>>(w); });
    timed("one-shot waves, 16 KiB each", gb, [&] { oneshot_kernel<16><<<kib / 64, 256>>>(w); });
    timed("one-shot waves, 160 KiB each", gb, [&] { oneshot_kernel<160><<<kib / 640, 256>>>(w); });
    timed("hipMemsetAsync", gb, [&] { (void)hipMemsetAsync(w, 1, (size_t)kib * 1024, 0); });
    return 0;
}
