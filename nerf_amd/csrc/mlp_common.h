// Pieces shared by the fused MLP kernels (forward, backward chain, weight gradients): the
// L2 -> LDS weight stream with its register prefetch pipeline, counted vector-memory waits,
// direct-to-LDS loads and the range reduction of the positional encoding.  Data layouts:
// mlp_layout_wide.h (forward), mlp_layout.h (training).
#pragma once
#include <type_traits>
#include "common.h"
#include "mlp_layout.h"

namespace nerf {
using namespace layout;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__

constexpr int kPrefetch = 4;  // A fragments held in registers ahead of the MFMA (<= 8)
constexpr int kPointsPerWave = 32;  // training kernels (saved tensors are laid out in 32-point wave tiles)
constexpr int kWavesPerBlock = 8;
constexpr int kPointsPerBlock = kPointsPerWave * kWavesPerBlock;

template <int I, int N, class F>
DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// s_waitcnt vmcnt(n) for a value that is a compile-time constant after unrolling (the switch
// folds to one instruction); counts above the 6-bit field wait for 63, which is stricter.
DEV void wait_vmcnt_n(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
        case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
        case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
        case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        default:
            if (n < 0) __builtin_trap();
            asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            break;
    }
}

// This wave's vector-memory operations counted at issue: loads (ordinary and direct-to-LDS) and
// stores share one hardware counter and retire in issue order, so "everything issued up to mark m
// has completed" is s_waitcnt vmcnt(issued - m).  An issue site may undercount (the wait then is
// merely stricter: an uncounted older operation retires first anyway, an uncounted younger one
// makes the wait ask for more than needed); it must never overcount.
struct VmQueue {
    int issued = 0;
    DEV void wait_until(int mark) const { wait_vmcnt_n(issued - mark); }
};

// From here on this wave's fp32 -> fp16 conversions saturate: a result beyond the fp16 range
// becomes +-65504 instead of +-inf (MODE.FP16_OVFL, bit 23 of hardware register 1; infinities and
// NaNs pass through).  Same result as clamping the fp32 value to +-65504 before rounding, without
// the two clamp instructions per pair of values.
DEV void saturate_fp16_conversions() {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1u);
}

DEV f32x16 mfma(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 16 bytes per lane, global memory -> LDS, issued from inline asm so that the compiler does not
// know about it (the builtin form makes it drain lgkmcnt / vmcnt to zero): where LDS
// addresses are not compile-time constants it cannot tell later LDS reads from loads still in
// flight and would drain the whole queue (s_waitcnt vmcnt(0)) before every read.  The caller
// owns the ordering: a counted s_waitcnt vmcnt, then a barrier if other waves read the data.
// lds_dst must be wave-uniform.
template <bool STREAMED = false>
DEV void glds16_untracked(const void* gsrc, void* lds_dst) {
    const unsigned addr = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    if constexpr (STREAMED)  // read-once data: do not displace what other kernels' waves keep in L2
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(addr)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(addr)
                     : "memory");
}

// The same with the source given as a wave-uniform base (SGPR pair) plus a 32-bit per-lane byte
// offset: address arithmetic stays on the scalar unit and a lane keeps one register per distinct
// offset pattern instead of a 64-bit pointer per load.
template <bool STREAMED = false>
DEV void glds16_untracked_sbase(const void* base, unsigned lane_off, unsigned lds_addr) {
    const unsigned addr = __builtin_amdgcn_readfirstlane(lds_addr);
    // `base` must be the same in every lane; say so to the compiler (an "s" operand it believes
    // divergent is silently passed in vector registers)
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const void* uniform_base = (const void*)(uintptr_t)(
        ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b));  // the builtin returns a signed int
    unsigned keep;
    if constexpr (STREAMED)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(uniform_base), "s"(addr)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(uniform_base), "s"(addr)
                     : "memory");
}

// The weight stream (NCHUNKS chunks of 16 KiB, RING LDS slots, DEPTH chunks in flight) and the
// register pipeline of A fragments.  Fragments must be taken in stream order, F = 0, 1, 2, ...
//
// WRAP: the workgroup consumes the same stream once per tile of its persistent loop, and the
// first DEPTH chunks of the next pass are requested during the tail of the current one, so that
// a pass never starts with an empty ring (after the last pass they are loaded for nothing: the
// kernel drains the queue before it ends).  Ring slots are compile-time (chunk % RING), so
// NCHUNKS must be a multiple of RING; chunks NLOAD .. NCHUNKS-1 exist only as slot numbers
// (nothing is loaded for them, nothing reads them).
template <int NCHUNKS, int RING, int DEPTH, int NLOAD = NCHUNKS, bool WRAP = false, int WAVES = kWavesPerBlock>
struct Stream {
    static_assert(WAVES == 4 || WAVES == 8 || WAVES == 16, "a chunk is 16 KiB: 4, 2 or 1 KiB per wave");
    static constexpr int kLoadsPerWave = kFragsPerChunk / WAVES;  // 1 KiB loads of one wave per chunk
    static_assert(DEPTH <= RING - 2 && kPrefetch <= 8, "ring slack");
    static_assert(!WRAP || NCHUNKS % RING == 0, "a wrapping stream keeps its slot phase");
    static_assert(NLOAD <= NCHUNKS && NCHUNKS - NLOAD <= 1, "at most one slot-only chunk");
    static constexpr int kRingBytes = RING * kChunkBytes;
    VmQueue& vm;
    const char* gsrc;   // this wave's share of chunk 0 (wave-uniform: loads take it as a scalar base)
    unsigned wr;        // LDS address of this wave's share of ring slot 0 (wave-uniform)
    int lane_off;       // lane * 16
    // This lane's read addresses of ring bytes 0 and 65536, kept as two opaque registers so that
    // every fragment read is base + 16-bit immediate (a ds_read offset ends at 65535; left to
    // itself the compiler forms the upper addresses with one vector instruction per read).
    const char* rd_lo;
    const char* rd_hi;
    int mark[RING];     // vm.issued right after each resident chunk's loads
    half8 a[kPrefetch];

    DEV Stream(VmQueue& q, const char* packed, char* lds_base, int wave, int lane)
        : vm(q), gsrc(packed + wave * (kLoadsPerWave * kFragBytes)),
          wr(__builtin_amdgcn_readfirstlane(
              (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds_base + wave * (kLoadsPerWave * kFragBytes)))),
          lane_off(lane * 16) {
        unsigned lo = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds_base + lane * 16);
        unsigned hi = lo + 65536u;
        asm volatile("" : "+v"(lo));
        asm volatile("" : "+v"(hi));
        rd_lo = (const char*)(__attribute__((address_space(3))) const char*)(uintptr_t)lo;
        rd_hi = (const char*)(__attribute__((address_space(3))) const char*)(uintptr_t)hi;
    }

#ifdef EXP_STALE_RING  // timing experiment: loads only while `live` (later passes read stale ring data)
    bool live = true;
#endif
    template <int C>
    DEV void issue() {
        if constexpr (C < NLOAD) {
            const char* g = gsrc + (size_t)C * kChunkBytes;
            const unsigned l = wr + (C % RING) * kChunkBytes;
#ifdef EXP_STALE_RING
            if (live)
#endif
            {
                static_for<0, kLoadsPerWave>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    glds16_untracked_sbase(g + i * kFragBytes, (unsigned)lane_off, l + i * kFragBytes);
                });
            }
            vm.issued += kLoadsPerWave;
        }
        mark[C % RING] = vm.issued;
    }

    // Makes chunk C readable by every wave, then refills the ring slot of chunk C + DEPTH - RING.
    template <int C>
    DEV void sync() {
        __builtin_amdgcn_sched_barrier(0);
        vm.wait_until(mark[C % RING]);
        asm volatile("s_barrier" ::: "memory");
        if constexpr (C + DEPTH < NCHUNKS) issue<C + DEPTH>();
        else if constexpr (WRAP) issue<C + DEPTH - NCHUNKS>();
        __builtin_amdgcn_sched_barrier(0);
    }

    // WRAP, after the last fragment of a pass: the barrier of the slot-only last chunk (every wave
    // is past the pass, so the slot two chunks back is free), then the next pass's chunk that
    // belongs there.  The next pass begins with start(), not prologue().
    DEV void finish_pass() {
        static_assert(!WRAP || NLOAD == NCHUNKS - 1, "finish_pass stands in for the sync of the slot-only chunk");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        issue<DEPTH - 1>();
        __builtin_amdgcn_sched_barrier(0);
    }

    // WRAP, at the top of every pass (the first one included, after prologue()): restarts the
    // queue bookkeeping from constants, so that every wait of a pass is an immediate.  Chunks
    // 0 .. DEPTH-1 are in flight, issued in that order; whatever else was issued between them
    // only makes the true distance of a chunk from the queue's tail larger than the loads per
    // younger chunk assumed here, so the waits are at worst stricter than needed.
    DEV void rebase() {
        constexpr int kBase = 1 << 20;
        static_for<0, DEPTH>([&](auto c) { mark[decltype(c)::value % RING] = kBase - kLoadsPerWave * (DEPTH - 1 - decltype(c)::value); });
        vm.issued = kBase;
        // a new value per pass as far as the compiler can tell: otherwise it computes the 64-bit
        // source address of every chunk once, outside the loop, and keeps them all in registers
        asm volatile("" : "+s"(gsrc), "+s"(wr));
    }

    template <int F>
    DEV half8 read() const {
        constexpr int C = F / kFragsPerChunk;
        constexpr int off = (C % RING) * kChunkBytes + (F % kFragsPerChunk) * kFragBytes;
        if constexpr (off < 65536) return *(const half8*)(rd_lo + off);
        else return *(const half8*)(rd_hi + (off - 65536));
    }

    // Issues the first DEPTH chunks (call early, before any other work).
    DEV void prologue() {
        static_for<0, DEPTH>([&](auto c) { issue<decltype(c)::value>(); });
    }

    DEV void start() {
        sync<0>();
        static_for<0, kPrefetch>([&](auto i) { a[decltype(i)::value] = read<decltype(i)::value>(); });
    }

    // Returns fragment F and starts the read of fragment F + kPrefetch into its register slot.
    template <int F>
    DEV half8 take() {
        const half8 cur = a[F % kPrefetch];
        if constexpr ((F + 8) % kFragsPerChunk == 0 && (F + 8) / kFragsPerChunk < NCHUNKS)
            sync<(F + 8) / kFragsPerChunk>();
#ifndef EXP_NO_LDS_READ  // timing experiment: the first kPrefetch fragments are used over and over
        if constexpr (F + kPrefetch < NCHUNKS * kFragsPerChunk) a[F % kPrefetch] = read<F + kPrefetch>();
#endif
        return cur;
    }
};

// x / (2 pi) reduced to about [-0.5, 0.5] revolutions with a two-term constant, so that the
// hardware sin/cos (which take revolutions) see a small argument whatever the magnitude of x.
DEV float to_revolutions(float x) {
    const float kHi = 0x1.45f306p-3f, kLo = 0x1.b9391p-28f;  // hi + lo = 1/(2 pi)
    const float k = rintf(x * kHi);
    float f = __builtin_fmaf(x, kHi, -k);
    return __builtin_fmaf(x, kLo, f);
}

DEV float double_revolutions(float f) {
    const float g = f + f;  // exact
    return g - rintf(g);    // exact
}

template <int NSLOT>
DEV void pack_slots(const float (&v)[NSLOT], half8* out) {
#pragma unroll
    for (int s = 0; s < NSLOT / 8; ++s) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            f32x2 x = {v[8 * s + j], v[8 * s + j + 1]};
            half2v p = __builtin_convertvector(x, half2v);
            out[s][j] = p[0], out[s][j + 1] = p[1];
        }
    }
}

}  // namespace nerf
