// Fused NeRF MLP forward for inference on gfx950: query point -> positional encoding -> 8x256 MLP
// with the skip after layer 4 -> alpha / bottleneck / view layer / rgb, one kernel, activations
// resident in registers from the first layer to the last (see mlp_layout_wide.h).
//
// Work decomposition.  A workgroup is 8 waves (2 per SIMD); each wave owns NPT 16-point tiles for
// the whole network: NPT = 3 for inference (48 points per wave, 384 per workgroup tile), NPT = 2
// for the training forward (its saved activations are laid out in 32-point wave tiles).
// Workgroups are persistent (one per CU, tiles dealt round-robin): while a tile's last layers
// run, the weight stream already refills the ring for the next tile and the next tile's rays
// and depths are fetched, so a CU never waits for a workgroup launch, a cold ring or its inputs.
// A layer is computed one 16-row output tile at a time: one A fragment per k-step feeds NPT
// v_mfma_f32_16x16x32_f16 (one per point tile), 2 x NPT x 4 fp32 accumulator registers; the
// first k-step of a tile adds to the tile's bias registers (read from LDS a few k-steps earlier).  A finished tile is converted (fp16 rounding, then a packed ReLU) in 2 NPT
// slices placed between the MFMAs of the FOLLOWING tile, so the matrix pipe never waits behind a
// block of conversion instructions; two consecutive tiles make one B fragment of the next layer.
// Two fragment sets ping-pong between layers.
//
// Why three point tiles.  Under this kernel the chip is power-limited (in-kernel clock 1.8-1.95
// of 2.4 GHz), and what costs power besides the MFMAs is operand delivery for the weights: the
// L2 -> LDS stream (once per workgroup tile) and the LDS -> register fragment reads (once per
// wave and k-step).  Measured by leaving each out (results wrong, timing only): no stream +17 %
// (clock 1.77 -> 2.08 GHz), no fragment reads +12 %.  A third point tile divides both by 1.5 per
// point and pays +6..7 % end to end with bit-identical results.  Its 96 more registers come from
// parking what is not needed between layers in LDS: the point encoding (used by layers 0 and 5)
// and the view directions (encoded only before the view layer); live state per wave is then
// 2 sets x 3 point tiles x 8 fragments x 4 registers + accumulators (24) + the A-fragment pipeline
// (16), 246 of the 256 registers two waves per SIMD allow, no spills.  (A spill would be a
// disaster here, not a nuisance: the compiler waits for vmcnt(0) before using a reloaded
// register, which drains the whole weight stream.)
//
// Weight stream: as in mlp_common.h (L2 -> LDS ring of 16 KiB chunks shared by the 8 waves,
// counted waits, one barrier per chunk executed half a chunk early).
//
// Bound: MFMA.  2344 MFMAs of 16 cycles per 32 points = 1.200 MFLOP/point issued for 1.187
// MFLOP/point of algorithmic work.
#include "mlp_common.h"
#include "mlp_layout_wide.h"

#ifdef FWD_STAMPS  // diagnostic build: shader-clock stamps at the layer boundaries of one pass
#define LSTAMP(i) (lstamp[i] = __builtin_amdgcn_s_memtime())
#else
#define LSTAMP(i) ((void)0)
#endif

namespace nerf {

struct FwdWideArgs {
    const char* packed;
    const float* rays;  // MODE 0: [n_rays, width], z [n_rays, S]
    const float* z;
    int width;
    const float* pts;       // MODE 1: [M,3]
    const float* viewdirs;  // MODE 1: [M/S,3]
    float4* raw;            // [M] (r,g,b,sigma)
    char* act;              // training: saved activations [ceil(M/256)*8][kActBlocks][1 KiB], else null
    int64_t M;              // points = n_rays * S
    int64_t n_rays;
    int S;                  // points per ray
#ifdef FWD_STAMPS  // diagnostic build only: per-workgroup shader-clock and wall-clock stamps
    unsigned long long* stamps;  // [workgroups][4]: s_memtime begin/end, s_memrealtime begin/end
#endif
};

// The ring: 5 slots of 16 KiB, 3 chunks in flight (measured equal to 8 / 6).  The 74 chunks of a
// network count as 75 ring positions, a multiple of the slot count, so that a workgroup can run
// the stream again for its next tile with the same compile-time slot of every chunk.
constexpr int kWRing = 5;
constexpr int kWDepth = 3;
constexpr int kWRingChunks = lw::kChunks + 1;
template <int WAVES>
using WideStreamT = Stream<kWRingChunks, kWRing, kWDepth, lw::kChunks, true, WAVES>;
constexpr int kWRingBytes = WideStreamT<8>::kRingBytes;
// Waves per workgroup of the inference kernel.  8 (two per SIMD): a wave's live state fills the 256
// registers that leaves (see the head of the file); 16 waves (four per SIMD, 128 registers each)
// spill even with two point tiles.  The training forward is always 8 (its saved-activation buffer
// is sized in 256-point tiles).
#ifndef NERF_FWD_WAVES
#define NERF_FWD_WAVES 8
#endif
// 16-point tiles per wave of the inference kernel (every A fragment read from LDS feeds one MFMA
// per point tile; the weight stream is read once per WAVES x 16 x NPT points).
#ifndef NERF_FWD_NPT
#define NERF_FWD_NPT 3
#endif

DEV f32x4 mfma_wide(half8 a, half8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <class STREAM, int NPT>
struct WidePipe {
    static constexpr int kNpt = NPT;  // 16-point tiles per wave
    STREAM& st;
    const char* bias;   // bias table in LDS, already offset by this lane group's 16 bytes
    f32x4 acc[2][NPT];  // [tile parity][point tile]
    f32x4 seed;         // the biases of the NEXT output tile: srcC of its first k-step (no register copies)
};

// Where a training forward saves its fragments: [wave tile][block][lane] (inference: unused).
// Streamed (nt) stores: the saved tensors are read once, by other kernels, and must not displace
// the weight stream from L2 (measured write rate of the kernel: 4.1 TB/s with nt, 3.5 with sc1,
// 3.2 with the default policy).  Stores are counted in the wave's vector-memory queue.
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
struct WSink {
    char* base;  // this lane's address inside block 0 of its wave tile
    VmQueue* vm;
    template <int BLK, class V>
    DEV void put(const V& v) const {
        __builtin_nontemporal_store(v, (V*)(base + BLK * kFragBytes));
        vm->issued += 1;
    }
};

// 16-byte register load issued from inline asm and counted in the wave's vector-memory queue (a
// load the compiler knows about would make it wait for vmcnt(0), i.e. for the whole weight stream
// and every pending store, before the first use).  The training forward uses it to fetch its own
// saved encodings back shortly before the layers that need them, instead of holding 24
// registers through the whole network.
struct FragLoad {
    half8 v;
    int mark;
    // wave-uniform base (scalar registers) + 32-bit per-lane byte offset: no 64-bit address to keep
    DEV void issue(VmQueue& vm, const void* base, unsigned lane_off) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        const void* uniform_base = (const void*)(uintptr_t)(
            ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
            (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b));
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lane_off), "s"(uniform_base) : "memory");
        vm.issued += 1;
        mark = vm.issued;
    }
    DEV half8 get(const VmQueue& vm) {
        vm.wait_until(mark);
        asm volatile("" : "+v"(v));
        return v;
    }
};

// What happens to the finished accumulators of one output tile (every point tile of the wave):
// two slices per point tile, each two registers -> one packed dword.
struct WNoPrev {
    static constexpr bool kActive = false;
    template <int S, class ACC> DEV void slice(const ACC&) const {}
};
// Tile number TL inside its layer (of NT): tiles 2s, 2s+1 fill the low / high half of fragment s.
// STORE: a finished fragment s of point tile pt is saved as block ABLK + 2s + pt, and for
// MBLK >= 0 the layer's ReLU mask bits are collected in mk[] and saved as block MBLK after the
// last tile.  With i = 2 TL + (r >> 1) numbering the register pairs of the layer: dword
// 2 pt + (i >> 4) of the lane, bit (i & 15) + 16 (r & 1) = register r of tile TL > 0, i.e. unit
// 16 TL + 4 (lane >> 4) + r of point 16 pt + (lane & 15).
template <bool RELU, int TL, bool STORE = false, int ABLK = 0, int MBLK = -1, int NT = 0>
struct WToFrag {
    static constexpr bool kActive = true;
    half8* dst0;  // fragment array of point tile 0, point tile pt follows at + pt * kStride
    WSink sink;
    unsigned* mk;
    static constexpr int kStride = 8;
    template <int S, class ACC> DEV void slice(const ACC& v) const {
        constexpr int pt = S >> 1, q = S & 1;
        half2v p;
        if constexpr (RELU) {
            // Round to fp16 first, then one packed max does the ReLU of both values (rounding is
            // monotone and keeps the sign, so this equals relu-then-round): two instructions per
            // pair instead of three, worth 1.7 % of the kernel, whose vector issue port is full.
            // The conversion saturates at 65504 (saturate_fp16_conversions).
            const f32x2 x = {v[pt][2 * q], v[pt][2 * q + 1]};
            const half2v c = __builtin_convertvector(x, half2v);
            unsigned r;
            asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, c)));
            p = __builtin_bit_cast(half2v, r);
        } else {
            const f32x2 x = {v[pt][2 * q], v[pt][2 * q + 1]};
            p = __builtin_convertvector(x, half2v);  // saturating
            // the conversion stays in its slot (left free, the compiler sinks it to the fragment's
            // first use and keeps the fp32 accumulators alive until then)
            unsigned r = __builtin_bit_cast(unsigned, p);
            asm volatile("" : "+v"(r));
            p = __builtin_bit_cast(half2v, r);
        }
        half8& f = dst0[pt * kStride + (TL >> 1)];
        constexpr int e = 4 * (TL & 1) + 2 * q;
        f[e] = p[0], f[e + 1] = p[1];
        if constexpr (STORE) {
            if constexpr (MBLK >= 0) {
                // both halves clamped to {0, 1} by one packed min, shifted into place by one
                // shift-or: two instructions per pair of units
                // (inline asm: the compiler expands a packed u16 min into five instructions)
                unsigned flags;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(flags) : "v"(__builtin_bit_cast(unsigned, p)), "s"(0x00010001u));
                constexpr int si = 2 * TL + q, sh = si & 15, w = 2 * pt + (si >> 4);
                if constexpr (sh == 0) mk[w] = flags; else mk[w] |= flags << sh;
            }
            if constexpr ((TL & 1) == 1 && q == 1) sink.template put<ABLK + 2 * (TL >> 1) + pt>(f);
            if constexpr (MBLK >= 0 && TL == NT - 1 && S == 3)
                sink.template put<MBLK>(uint4v{mk[0], NT > 8 ? mk[1] : 0u, mk[2], NT > 8 ? mk[3] : 0u});
        }
    }
};
struct WKeepRow0 {
    static constexpr bool kActive = true;
    float* out;  // [point tiles]
    template <int S, class ACC> DEV void slice(const ACC& v) const {
        if constexpr ((S & 1) == 0) out[S >> 1] = v[S >> 1][0];
    }
};

// The 2 npt slices of a finished tile are spread evenly over the k-steps of the next one, from the
// second k-step to the last but one: the first tile of a layer reads the fragment that the last
// tile of the layer before completes in its LAST k-step (two-step tiles only ever follow a tile
// of their own layer and take every slice in their second step).
constexpr int w_first_slice_step(int nk) { return nk >= 2 ? 1 : 0; }
constexpr int w_slices_per_step(int nk, int npt) {
    const int steps = nk >= 3 ? nk - 2 : 1;
    return (2 * npt + steps - 1) / steps;
}
constexpr int w_last_slice_step(int nk, int npt) { return w_first_slice_step(nk) + (2 * npt - 1) / w_slices_per_step(nk, npt); }
constexpr int w_bias_step(int nk, int npt) { return w_last_slice_step(nk, npt) > nk - 3 ? w_last_slice_step(nk, npt) : nk - 3; }
static_assert(w_slices_per_step(2, 2) == 4 && w_slices_per_step(4, 2) == 2 && w_slices_per_step(8, 2) == 1 &&
              w_slices_per_step(9, 2) == 1 && w_slices_per_step(10, 2) == 1, "two point tiles: the schedule the training forward was tuned on");
static_assert(w_last_slice_step(2, 3) < 2 && w_last_slice_step(4, 3) < 3 && w_last_slice_step(8, 3) < 7 &&
              w_last_slice_step(9, 3) < 8 && w_last_slice_step(10, 3) < 9 && w_last_slice_step(4, 2) < 3, "slice schedule");

// Output tile T (= its bias tile): NK k-steps of fragments F0..; getB(pt, k) is the B fragment of
// point tile pt.  Every k-step is one scheduling region: two MFMAs, the LDS read of the A
// fragment kPrefetch steps ahead, a slice of the previous tile and, late in the tile, the LDS read
// of the next tile's biases.
template <int F0, int NK, int T, class PIPE, class GetB, class Prev>
DEV void wide_tile(PIPE& p, GetB&& getB, const Prev& prev) {
    static_for<0, NK>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int NPT = PIPE::kNpt, NS = 2 * NPT;
        const half8 a = p.st.template take<F0 + k>();
        static_for<0, NPT>([&](auto ptc) {
            constexpr int pt = decltype(ptc)::value;
            // the first k-step takes the bias tile as its addend: the accumulators are never seeded by copies
            if constexpr (k == 0) p.acc[T & 1][pt] = mfma_wide(a, getB(ptc, kc), p.seed);
            else p.acc[T & 1][pt] = mfma_wide(a, getB(ptc, kc), p.acc[T & 1][pt]);
        });
        if constexpr (Prev::kActive && k >= w_first_slice_step(NK)) {
            constexpr int sp = w_slices_per_step(NK, NPT), s0 = (k - w_first_slice_step(NK)) * sp;
            static_for<(s0 < NS ? s0 : NS), (s0 + sp < NS ? s0 + sp : NS)>(
                [&](auto sc) { prev.template slice<decltype(sc)::value>(p.acc[(T & 1) ^ 1]); });
        }
        if constexpr (k == w_bias_step(NK, NPT) && T + 1 < lw::T_END) p.seed = *(const f32x4*)(p.bias + (T + 1) * 64);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One layer of NT output tiles of NK k-steps; its output becomes fragments 0..NT/2-1 of `out`
// (every point tile).  `before` finishes the tile that preceded the layer; the returned object
// finishes the layer's last tile.
// TB, TE: only tiles TB .. TE-1 of the layer (a layer may be run in two parts whose B operands
// come from different places).
template <int F0, int NT, int NK, int T0, bool RELU, bool STORE, int ABLK, int MBLK, int TB = 0, int TE = NT, class PIPE, class GetB, class Prev>
DEV auto wide_layer(PIPE& p, const WSink& sink, unsigned* mk, half8* out, GetB&& getB, const Prev& before) {
    wide_tile<F0 + TB * NK, NK, T0 + TB>(p, getB, before);
    static_for<TB + 1, TE>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        wide_tile<F0 + t * NK, NK, T0 + t>(p, getB, WToFrag<RELU, t - 1, STORE, ABLK, MBLK, NT>{out, sink, mk});
    });
    return WToFrag<RELU, TE - 1, STORE, ABLK, MBLK, NT>{out, sink, mk};
}

// Local encoding slots of lane group g for a point: see lw::pe_feature.
DEV void encode_point(const float (&x)[3], int g, half8 (&out)[2]) {
    const float e = g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : 0.f;
    const float base = g == 0 ? 1.f : g == 1 ? 4.f : g == 2 ? 32.f : 128.f;
    const bool odd = g & 1;
    float f[3], s[3][3], c[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) f[d] = to_revolutions(x[d] * base);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            s[o][d] = __builtin_amdgcn_sinf(f[d]);
            c[o][d] = __builtin_amdgcn_cosf(f[d]);
            if (o < 2) f[d] = double_revolutions(f[d]);
        }
    }
    float v[16];
    v[0] = e;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        v[1 + d] = odd ? c[0][d] : s[0][d];
        v[4 + d] = odd ? s[1][d] : c[0][d];
        v[7 + d] = odd ? c[1][d] : s[1][d];
        v[10 + d] = odd ? s[2][d] : c[1][d];
        v[13 + d] = odd ? c[2][d] : s[2][d];
    }
    pack_slots<16>(v, out);
}

// Local view-direction slots of lane group g: see lw::view_feature.
DEV void encode_view(const float (&x)[3], int g, half8 (&out)[1]) {
    const float e = g == 0 ? x[0] : g == 1 ? x[1] : g == 2 ? x[2] : 0.f;
    const float base = g == 0 ? 1.f : g == 1 ? 2.f : g == 2 ? 4.f : 8.f;
    float v[8];
    v[0] = e, v[7] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float f = to_revolutions(x[d] * base);
        v[1 + d] = __builtin_amdgcn_sinf(f);
        v[4 + d] = __builtin_amdgcn_cosf(f);
    }
    pack_slots<8>(v, out);
}

// The words a lane reads for one query point.  MODE 0 (rays + depths): q4 = o.xyz, d.x; q2a = d.yz;
// q2b = viewdir.xy; s1 = viewdir.z; s2 = depth.  MODE 1 (explicit points): q2a = p.xy; s1 = p.z;
// q2b = viewdir.xy; s2 = viewdir.z.
struct PointWords {
    f32x4 q4;
    f32x2 q2a, q2b;
    float s1, s2;
};

// Which ray a point belongs to, without a division per tile.  The first point of this wave in a
// tile is base = tile * PB + wave * PW = q * S + r (PW points per wave, PB per workgroup tile); the
// workgroup's next tile lies gridDim.x * PB points further, which advances (q, r) by a fixed
// (dq, dr).  All of it is wave-uniform and runs on the scalar unit; a lane adds its point number
// (< PW) to r and carries at most once when S >= PW (shorter rays take the division).
struct RayCursor {
    int64_t base, q, dq, step;
    int r, dr, S, pw;
    DEV RayCursor(int64_t tile, int wave, int S_, int points_per_block, int points_per_wave) : S(S_), pw(points_per_wave) {
        base = tile * points_per_block + wave * points_per_wave;
        q = base / S, r = (int)(base - q * S);
        step = (int64_t)gridDim.x * points_per_block;
        dq = step / S, dr = (int)(step - dq * S);
    }
    DEV void advance() {
        base += step, q += dq, r += dr;
        if (r >= S) r -= S, ++q;
    }
    DEV int64_t ray_of(int l) const { return S >= pw ? q + (r + l >= S ? 1 : 0) : (base + l) / S; }
};

template <int MODE>
DEV void point_addresses(const FwdWideArgs& a, const RayCursor& cur, int lane, int pt,
                         const float*& pa, const float*& pb, const float*& pz) {
    const int l = 16 * pt + (lane & 15);
    const int64_t p = cur.base + l, last = a.M - 1;
    const int64_t pc = p < last ? p : last;       // past the end: the last point, never stored
    int64_t ray = cur.ray_of(l);
    ray = ray < a.n_rays - 1 ? ray : a.n_rays - 1;
    if (MODE == 0) pa = a.rays + ray * a.width, pb = nullptr, pz = a.z + pc;
    else pa = a.pts + pc * 3, pb = a.viewdirs + ray * 3, pz = nullptr;
}

// Plain loads (first tile of a workgroup: nothing else is in flight yet).
template <int MODE>
DEV PointWords load_point(const FwdWideArgs& a, const RayCursor& cur, int lane, int pt) {
    const float *pa, *pb, *pz;
    point_addresses<MODE>(a, cur, lane, pt, pa, pb, pz);
    PointWords w{};
    if (MODE == 0) {
        w.q4 = f32x4{pa[0], pa[1], pa[2], pa[3]}, w.q2a = f32x2{pa[4], pa[5]};
        w.q2b = f32x2{pa[8], pa[9]}, w.s1 = pa[10], w.s2 = *pz;
    } else {
        w.q2a = f32x2{pa[0], pa[1]}, w.s1 = pa[2], w.q2b = f32x2{pb[0], pb[1]}, w.s2 = pb[2];
    }
    return w;
}

// The same loads issued from inline asm and counted in the wave's vector-memory queue, for the
// NEXT tile while the current one computes (a load the compiler knows about would be waited for
// with vmcnt(0), draining the weight stream).
struct PointFetch {
    PointWords w;
    int mark;
    template <int MODE>
    DEV void issue(VmQueue& vm, const FwdWideArgs& a, const RayCursor& cur, int lane, int pt) {
        const float *pa, *pb, *pz;
        point_addresses<MODE>(a, cur, lane, pt, pa, pb, pz);
        if (MODE == 0) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.q4) : "v"(pa) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(w.q2a) : "v"(pa) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off offset:32" : "=v"(w.q2b) : "v"(pa) : "memory");
            asm volatile("global_load_dword %0, %1, off offset:40" : "=v"(w.s1) : "v"(pa) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(w.s2) : "v"(pz) : "memory");
            vm.issued += 5;
        } else {
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(w.q2a) : "v"(pa) : "memory");
            asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(w.s1) : "v"(pa) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(w.q2b) : "v"(pb) : "memory");
            asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(w.s2) : "v"(pb) : "memory");
            vm.issued += 4;
        }
        mark = vm.issued;
    }
    DEV PointWords get(const VmQueue& vm) {
        vm.wait_until(mark);
        asm volatile("" : "+v"(w.q4), "+v"(w.q2a), "+v"(w.q2b), "+v"(w.s1), "+v"(w.s2));
        return w;
    }
};

template <int MODE>
DEV void point_xyz(const PointWords& w, float (&x)[3], float (&vd)[3]) {
    if (MODE == 0) {
        const float zz = w.s2;
        x[0] = w.q4[0] + w.q4[3] * zz;  // pts = rays_o + rays_d * z
        x[1] = w.q4[1] + w.q2a[0] * zz;
        x[2] = w.q4[2] + w.q2a[1] * zz;
        vd[0] = w.q2b[0], vd[1] = w.q2b[1], vd[2] = w.s1;
    } else {
        x[0] = w.q2a[0], x[1] = w.q2a[1], x[2] = w.s1;
        vd[0] = w.q2b[0], vd[1] = w.q2b[1], vd[2] = w.s2;
    }
}

template <int MODE, bool STORE, int WAVES, int NPT>
__global__ __launch_bounds__(WAVES * 64) void mlp_fwd_wide_kernel(FwdWideArgs a) {
    // Three and more point tiles per wave: the point encoding waits in LDS (each wave its own
    // 2 NPT KiB) between layer 0 and layer 5 instead of in 8 NPT registers, and the view
    // directions (16 B per point) wait there until the view layer, where they are encoded.
    constexpr bool kEncLds = NPT >= 3;
    constexpr int kEncWaveBytes = 2 * NPT * kFragBytes + NPT * 256;
    constexpr int kEncBytes = kEncLds ? WAVES * kEncWaveBytes : 0;
    __shared__ __attribute__((aligned(16))) char lds[kWRingBytes + lw::kBiasBytes + kEncBytes];
    static_assert(!STORE || (NPT == 2 && WAVES == kWavesPerBlock), "saved activations are laid out in 32-point wave tiles, 8 per workgroup tile");
    constexpr int kPW = 16 * NPT;      // points per wave
    constexpr int kPB = WAVES * kPW;   // points per workgroup tile
    using WideStream = WideStreamT<WAVES>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int64_t ntiles = (a.M + kPB - 1) / kPB;
    saturate_fp16_conversions();
#ifdef FWD_STAMPS
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    VmQueue vm;
    WideStream st(vm, a.packed, lds, wave, lane);
    // biases first (oldest in the queue), 1 KiB per wave-load, then the first chunks
    for (int blk = wave; blk < lw::kBiasBytes / kFragBytes; blk += WAVES)
        glds16_untracked(a.packed + lw::kBiasOffset + blk * kFragBytes + lane * 16,
                         lds + kWRingBytes + blk * kFragBytes);
    st.prologue();

    constexpr int kS = 8;  // fragments (k-steps of 32 features) per point tile in a set
    half8 fa[NPT * kS], fb[NPT * kS], pe[NPT][2], vpe[NPT][1];
    auto from_pe = [&](auto pt, auto kc) { return pe[decltype(pt)::value][decltype(kc)::value]; };
    auto from_a = [&](auto pt, auto kc) { return fa[decltype(pt)::value * kS + decltype(kc)::value]; };
    auto from_b = [&](auto pt, auto kc) { return fb[decltype(pt)::value * kS + decltype(kc)::value]; };
    static_assert(WToFrag<true, 0>::kStride == kS, "fragment set stride");

    // the bias table's address for this lane group, opaque to the compiler so that every bias
    // read is this register + an immediate (the table lies beyond a ds_read's 16-bit offset)
    unsigned bias_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + kWRingBytes + g * 16);
    asm volatile("" : "+v"(bias_addr));
    WidePipe<WideStream, NPT> pp{st, (const char*)(__attribute__((address_space(3))) const char*)(uintptr_t)bias_addr, {}, {}};

    // ---- this lane's two query points (one per point tile) of the workgroup's first tile ----
    RayCursor where(blockIdx.x, wave, a.S, kPB, kPW);
    PointWords cur[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) cur[pt] = load_point<MODE>(a, where, lane, pt);

#ifdef FWD_STAMPS
    int pass_no = 0;
#endif
    // The lane number as a value the compiler cannot connect with its earlier uses: what a pass
    // derives from it (lane group, point offsets, store addresses) is formed again where it is
    // needed (one or two instructions) instead of being carried, or spilled, through the network.
    auto fresh_lane = [&]() {
        int off = st.lane_off;
        asm volatile("" : "+v"(off));
        return off >> 4;
    };
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef FWD_STAMPS
        unsigned long long lstamp[16];
#endif
        LSTAMP(0);
        st.rebase();
        const int lane_top = fresh_lane();
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            float x[3], vd[3];
            point_xyz<MODE>(cur[pt], x, vd);
            encode_point(x, lane_top >> 4, pe[pt]);
            if constexpr (kEncLds) {  // every lane group of a point writes the same 16 bytes
                *(f32x4*)(lds + kWRingBytes + lw::kBiasBytes + wave * kEncWaveBytes + 2 * NPT * kFragBytes + pt * 256 + (lane_top & 15) * 16) =
                    f32x4{vd[0], vd[1], vd[2], 0.f};
            } else {
                encode_view(vd, lane_top >> 4, vpe[pt]);
            }
        }
        char* const enc_lds = lds + kWRingBytes + lw::kBiasBytes + wave * kEncWaveBytes + lane_top * 16;
        if constexpr (kEncLds) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                *(half8*)(enc_lds + (2 * pt) * kFragBytes) = pe[pt][0];
                *(half8*)(enc_lds + (2 * pt + 1) * kFragBytes) = pe[pt][1];
            }
        }

        WSink sink{nullptr, &vm};
        const char* act_tile = nullptr;  // this wave's tile of saved activations (wave-uniform)
        if constexpr (STORE) {
            const int64_t wave_tile = tile * WAVES + wave;
            act_tile = a.act + (size_t)wave_tile * kActBlocks * kFragBytes;
            sink.base = a.act + (size_t)wave_tile * kActBlocks * kFragBytes + lane * 16;
            static_for<0, 2>([&](auto sc) {
                constexpr int k = decltype(sc)::value;
                sink.template put<A_PE + 2 * k>(pe[0][k]);
                sink.template put<A_PE + 2 * k + 1>(pe[1][k]);
            });
            sink.template put<A_VPE>(vpe[0][0]);
            sink.template put<A_VPE + 1>(vpe[1][0]);
        }
        const int enc_saved = vm.issued;  // the saved encodings are in memory once the queue is past this mark
        FragLoad lpe[4], lvpe[2];
        PointFetch nxt[NPT];
        unsigned mk[4] = {0u, 0u, 0u, 0u};

        LSTAMP(1);
        st.start();
        LSTAMP(2);
        pp.seed = *(const f32x4*)(pp.bias);
        auto c0 = wide_layer<lw::F_L0, 16, 2, lw::T_L0, true, STORE, A_H0, A_MASK + 0>(pp, sink, mk, fa, from_pe, WNoPrev{});                     // layer 0
        LSTAMP(3);
        auto c1 = wide_layer<lw::F_L1 + 0 * 128, 16, 8, lw::T_L1 + 0, true, STORE, A_H0 + 16, A_MASK + 1>(pp, sink, mk, fb, from_a, c0);        // layers 1..4
        LSTAMP(4);
        auto c2 = wide_layer<lw::F_L1 + 1 * 128, 16, 8, lw::T_L1 + 16, true, STORE, A_H0 + 32, A_MASK + 2>(pp, sink, mk, fa, from_b, c1);
        LSTAMP(5);
        auto c3 = wide_layer<lw::F_L1 + 2 * 128, 16, 8, lw::T_L1 + 32, true, STORE, A_H0 + 48, A_MASK + 3>(pp, sink, mk, fb, from_a, c2);
        LSTAMP(6);
        if constexpr (STORE) {  // the point encoding comes back during layer 4 (blocks A_PE + 2k + pt)
            vm.wait_until(enc_saved);
            static_for<0, 4>([&](auto ic) { lpe[decltype(ic)::value].issue(vm, act_tile + (A_PE + decltype(ic)::value) * kFragBytes, lane * 16); });
        }
        auto c4 = wide_layer<lw::F_L1 + 3 * 128, 16, 8, lw::T_L1 + 48, true, STORE, A_H0 + 64, A_MASK + 4>(pp, sink, mk, fa, from_b, c3);
        LSTAMP(7);
        half8 pe5[NPT][2];
        static_for<0, 2 * NPT>([&](auto ic) {
            constexpr int i = decltype(ic)::value, k = i / NPT, pt = i % NPT;  // STORE (NPT = 2): block A_PE + 2 k + pt
            if constexpr (STORE) pe5[pt][k] = lpe[i].get(vm);
            else if constexpr (kEncLds) pe5[pt][k] = *(const half8*)(enc_lds + (2 * pt + k) * kFragBytes);
            else pe5[pt][k] = pe[pt][k];
        });
        // layer 5: [PE ; h].  With the encoding parked in LDS it is held in registers only while the
        // layer's output set has room to spare (its last two fragments per point tile are born in
        // tiles 12..15); those last tiles read it from LDS again, k-step by k-step.
        constexpr int kL5Split = kEncLds ? 12 : 16;
        auto c5a = wide_layer<lw::F_L5, 16, 10, lw::T_L5, true, STORE, A_H0 + 80, A_MASK + 5, 0, kL5Split>(pp, sink, mk, fb, [&](auto pt, auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < 2) return pe5[decltype(pt)::value][k]; else return fa[decltype(pt)::value * kS + k - 2];
        }, c4);
        auto c5 = [&]() {
            if constexpr (kEncLds)
                return wide_layer<lw::F_L5, 16, 10, lw::T_L5, true, STORE, A_H0 + 80, A_MASK + 5, kL5Split, 16>(pp, sink, mk, fb, [&](auto pt, auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < 2) return *(const half8*)(enc_lds + (2 * decltype(pt)::value + k) * kFragBytes);
                    else return fa[decltype(pt)::value * kS + k - 2];
                }, c5a);
            else return c5a;
        }();
        LSTAMP(8);
        auto c6 = wide_layer<lw::F_L6 + 0 * 128, 16, 8, lw::T_L6 + 0, true, STORE, A_H0 + 96, A_MASK + 6>(pp, sink, mk, fa, from_b, c5);        // layers 6, 7
        LSTAMP(9);
        auto c7 = wide_layer<lw::F_L6 + 1 * 128, 16, 8, lw::T_L6 + 16, true, STORE, A_H0 + 112, A_MASK + 7>(pp, sink, mk, fb, from_a, c6);
        LSTAMP(10);
        // alpha head (row 0 of its tile), then the bottleneck (linear)
        float sigma[NPT];
        wide_tile<lw::F_ALPHA, 8, lw::T_ALPHA>(pp, from_b, c7);
        if constexpr (STORE) {  // the view encoding comes back during the bottleneck layer
            lvpe[0].issue(vm, act_tile + A_VPE * kFragBytes, lane * 16);
            lvpe[1].issue(vm, act_tile + (A_VPE + 1) * kFragBytes, lane * 16);
        }
        // (linear, and not saved for the backward pass: see the note at kActBlocks)
        auto c8 = wide_layer<lw::F_BOTT, 16, 8, lw::T_BOTT, false, false, 0, -1>(pp, sink, mk, fa, from_b, WKeepRow0{sigma});
        LSTAMP(11);
        half8 vpe9[NPT];
        if constexpr (STORE) vpe9[0] = lvpe[0].get(vm), vpe9[1] = lvpe[1].get(vm);
        else {
            const int lane_view = fresh_lane();
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                if constexpr (kEncLds) {
                    const f32x4 d = *(const f32x4*)(lds + kWRingBytes + lw::kBiasBytes + wave * kEncWaveBytes + 2 * NPT * kFragBytes + pt * 256 + (lane_view & 15) * 16);
                    const float vd[3] = {d[0], d[1], d[2]};
                    half8 e[1];
                    encode_view(vd, lane_view >> 4, e);
                    vpe9[pt] = e[0];
                } else {
                    vpe9[pt] = vpe[pt][0];
                }
            }
        }
        // the next tile's rays and depths are requested before the last two layers (past the
        // last tile the addresses are clamped to the last point and the words are not used)
        where.advance();
        const int lane_nxt = fresh_lane();
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) nxt[pt].template issue<MODE>(vm, a, where, lane_nxt, pt);
        // view layer: [bottleneck ; view PE] -> 128, relu
        auto c9 = wide_layer<lw::F_VIEW, 8, 9, lw::T_VIEW, true, STORE, A_V, A_MASK + 8>(pp, sink, mk, fb, [&](auto pt, auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < 8) return fa[decltype(pt)::value * kS + k]; else return vpe9[decltype(pt)::value];
        }, c8);
        LSTAMP(12);
        wide_tile<lw::F_RGB, 4, lw::T_RGB>(pp, from_b, c9);                                                  // rgb rows 0..2
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const f32x4& c = pp.acc[lw::T_RGB & 1][pt];
            const int ln = fresh_lane();
            const int64_t p = tile * kPB + wave * kPW + 16 * pt + (ln & 15);
            if (p < a.M && (ln >> 4) == 0) a.raw[p] = make_float4(c[0], c[1], c[2], sigma[pt]);
        }
        LSTAMP(13);
#ifdef EXP_STALE_RING
        st.live = false;
#endif
        st.finish_pass();
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) cur[pt] = nxt[pt].get(vm);
        LSTAMP(14);
#ifdef FWD_STAMPS
        if (++pass_no == 6 && a.stamps && lane == 0 && blockIdx.x < 32) {
            unsigned long long* o = a.stamps + 4 * 256 + ((size_t)blockIdx.x * WAVES + wave) * 16;
#pragma unroll
            for (int i = 0; i < 15; ++i) o[i] = lstamp[i];
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring loads requested for a pass that does not come
#ifdef FWD_STAMPS
    if (threadIdx.x == 0 && a.stamps) {
        unsigned long long* s = a.stamps + 4 * (size_t)blockIdx.x;
        s[0] = stamp_c0, s[1] = __builtin_amdgcn_s_memtime(), s[2] = stamp_r0, s[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef FWD_STAMPS
static unsigned long long* g_fwd_stamps = nullptr;
#endif

static int launch_fwd_wide(FwdWideArgs a, int mode, hipStream_t st, const char* what) {
    constexpr int kInferWaves = NERF_FWD_WAVES, kInferNpt = NERF_FWD_NPT, kInferPB = kInferWaves * 16 * kInferNpt;
    const int64_t tiles = a.act ? (a.M + kPointsPerBlock - 1) / kPointsPerBlock : (a.M + kInferPB - 1) / kInferPB;
    if (a.S > (1 << 24)) return NERF_FAIL(NERF_E_UNSUPPORTED, "%s: more than 2^24 points per ray (%d)", what, a.S);
    // persistent workgroups: one per CU (the ring and two waves per SIMD fill a CU), fewer for small calls
    const int cus = device_cu_count();
    if (cus <= 0) return NERF_FAIL(NERF_E_UNSUPPORTED, "%s: no current device", what);
    const int64_t blocks = tiles < cus ? tiles : cus;
#ifdef FWD_STAMPS
    a.stamps = g_fwd_stamps;
#endif
    if (a.act)
        mlp_fwd_wide_kernel<0, true, kWavesPerBlock, 2><<<(unsigned)blocks, kWavesPerBlock * 64, 0, st>>>(a);
    else if (mode == 0)
        mlp_fwd_wide_kernel<0, false, kInferWaves, kInferNpt><<<(unsigned)blocks, kInferWaves * 64, 0, st>>>(a);
    else
        mlp_fwd_wide_kernel<1, false, kInferWaves, kInferNpt><<<(unsigned)blocks, kInferWaves * 64, 0, st>>>(a);
    return check_launch(what);
}

int preload_mlp_forward_code(const char* what) {
    return preload_code_of(mlp_fwd_wide_kernel<0, false, NERF_FWD_WAVES, NERF_FWD_NPT>, what);
}

}  // namespace nerf

using namespace nerf;

#ifdef FWD_STAMPS
NERF_API void nerf_debug_set_fwd_stamps(void* p) { g_fwd_stamps = (unsigned long long*)p; }
#endif

NERF_API int nerf_mlp_forward_rays(const void* packed, const float* rays, int width,
                                   const float* z, int64_t n_rays, int S, float* raw,
                                   nerf_stream_t stream) {
    NERF_REQUIRE(packed && rays && z && raw, "nerf_mlp_forward_rays: null pointer");
    NERF_REQUIRE(width >= 11, "nerf_mlp_forward_rays: ray batch needs view directions (width >= 11, got %d)", width);
    NERF_REQUIRE(n_rays >= 0 && S > 0, "nerf_mlp_forward_rays: bad size");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0,
                 "nerf_mlp_forward_rays: packed and raw must be 16-byte aligned");
    if (n_rays == 0) return 0;
    FwdWideArgs a{};
    a.packed = (const char*)packed, a.rays = rays, a.z = z, a.width = width;
    a.raw = (float4*)raw, a.M = n_rays * S, a.n_rays = n_rays, a.S = S;
    return launch_fwd_wide(a, 0, as_stream(stream), "nerf_mlp_forward_rays");
}

NERF_API int nerf_mlp_forward_pts(const void* packed, const float* pts, const float* viewdirs,
                                  int64_t n, int S, float* raw, nerf_stream_t stream) {
    NERF_REQUIRE(packed && pts && viewdirs && raw, "nerf_mlp_forward_pts: null pointer");
    NERF_REQUIRE(n >= 0 && S > 0, "nerf_mlp_forward_pts: bad size");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0,
                 "nerf_mlp_forward_pts: packed and raw must be 16-byte aligned");
    if (n == 0) return 0;
    FwdWideArgs a{};
    a.packed = (const char*)packed, a.pts = pts, a.viewdirs = viewdirs;
    a.raw = (float4*)raw, a.M = n * S, a.n_rays = n, a.S = S;
    return launch_fwd_wide(a, 1, as_stream(stream), "nerf_mlp_forward_pts");
}

NERF_API size_t nerf_mlp_act_bytes(int64_t n_points) {
    const int64_t wave_tiles = (n_points + kPointsPerBlock - 1) / kPointsPerBlock * kWavesPerBlock;
    return (size_t)wave_tiles * layout::kActBlocks * layout::kFragBytes;
}

NERF_API int nerf_mlp_forward_train(const void* packed, const float* rays, int width, const float* z,
                                    int64_t n_rays, int S, float* raw, void* act,
                                    nerf_stream_t stream) {
    NERF_REQUIRE(packed && rays && z && raw && act, "nerf_mlp_forward_train: null pointer");
    NERF_REQUIRE(width >= 11, "nerf_mlp_forward_train: ray batch needs view directions (width >= 11, got %d)", width);
    NERF_REQUIRE(n_rays >= 0 && S > 0, "nerf_mlp_forward_train: bad size");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0 && ((uintptr_t)act & 15) == 0,
                 "nerf_mlp_forward_train: packed, raw and act must be 16-byte aligned");
    if (n_rays == 0) return 0;
    FwdWideArgs a{};
    a.packed = (const char*)packed, a.rays = rays, a.z = z, a.width = width;
    a.raw = (float4*)raw, a.act = (char*)act, a.M = n_rays * S, a.n_rays = n_rays, a.S = S;
    return launch_fwd_wide(a, 0, as_stream(stream), "nerf_mlp_forward_train");
}
