// Fused NeRF MLP forward for gfx950: query point -> positional encoding -> 8x256 MLP with the
// skip after layer 4 -> alpha / bottleneck / view layer / rgb, one kernel, activations resident in
// registers from the first layer to the last (see mlp_layout.h for the orientation trick).
//
// Work decomposition.  A workgroup is 8 waves (2 per SIMD); each wave owns 32 query points for
// the whole network, so a workgroup covers 256 points.  A layer is computed one 32-row output
// tile at a time: 16 fp32 accumulator registers, seeded with the biases, run through the tile's
// k-steps and are converted at once (ReLU folded into the fp16 clamp) into two B fragments of
// the NEXT layer's input; two fragment sets ping-pong between layers.  Live state per wave:
// 2 x 16 fragments x 4 registers + the point encoding kept for the skip (4 x 4) + one or two
// accumulator tiles.
//
// Weight stream.  All eight waves consume the same A fragments in the same order, so the packed
// network is streamed L2 -> LDS once per workgroup through an 8-slot ring of 16 KiB chunks with
// direct-to-LDS loads (global_load_lds_dwordx4, two per wave per chunk), kDepth chunks ahead of
// the consumer.  A chunk becomes readable through one counted s_waitcnt vmcnt + one s_barrier,
// executed half a chunk EARLY (while fragment 8 of the previous chunk is being consumed) so that
// the register prefetch of A fragments (kPrefetch deep) never drains at a chunk boundary.  The
// same point refills the ring slot whose last reader is at least eight MFMAs behind every wave.
// Biases (10 KiB) are copied to LDS once and seed the accumulators by four broadcast reads.
//
// Bound: MFMA.  1184 v_mfma_f32_32x32x16_f16 per 32 points = 1.212 MFLOP/point issued for
// 1.187 MFLOP/point of algorithmic work (2.1 % padding: 63->64, 283->288 inputs, 1->32 and 3->32
// output rows of the two heads).
#include "mlp_common.h"
#include "mlp_layout_wide.h"

namespace nerf {

struct FwdArgs {
    const char* packed;
    const float* rays;  // MODE 0: [n_rays, width], z [n_rays, S]
    const float* z;
    int width;
    const float* pts;       // MODE 1: [M,3]
    const float* viewdirs;  // MODE 1: [M/S,3]
    float4* raw;            // [M] (r,g,b,sigma)
    char* act;              // training: saved activations [ceil(M/256)*8][kActBlocks][1 KiB], else null
    int64_t M;              // points
    int S;                  // points per ray
};

constexpr int kRing = 8;    // LDS ring slots (16 KiB each)
constexpr int kDepth = 6;   // chunks in flight ahead of the chunk being synchronised
using FwdStream = Stream<kChunks, kRing, kDepth>;
constexpr int kRingBytes = FwdStream::kRingBytes;

// Where a training forward saves its fragments: [wave tile][block][lane] (null: inference).
struct ActSink {
    char* base;  // this lane's address inside block 0 of its wave tile
    VmQueue* vm;
    template <int BLK>
    DEV void put(const half8& v) const {
        __builtin_nontemporal_store(v, (half8*)(base + BLK * kFragBytes));  // streamed: keep L2 for the weights
        vm->issued += 1;
    }
    template <int BLK, class V>
    DEV void put_mask(const V& v) const {
        __builtin_nontemporal_store(v, (V*)(base + BLK * kFragBytes));
        vm->issued += 1;
    }
};

// What happens to a finished accumulator tile.  The work is cut into eight slices that run
// between the MFMAs of the FOLLOWING tile, so that no wave ever sits in a block of conversion
// instructions while the matrix pipe idles (and the last MFMA's result latency is never exposed).
struct NoPrev {
    static constexpr bool kActive = false;
    template <int S> DEV void slice(const f32x16&) const {}
};
// Registers 2S, 2S+1 -> packed dword S&3 of fragment S>>2: round to nearest even, clamped to the
// fp16 range; RELU folds the activation into the clamp (one v_med3_f32).  When STORE, finished
// fragments go to blocks BLK, BLK+1 of the saved activations and, for MBLK >= 0, the layer's ReLU
// mask bits (tile TL of NT) are collected in mk[] and stored as block MBLK after the last tile.
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
template <bool RELU, bool STORE, int BLK, int TL = 0, int NT = 0, int MBLK = -1>
struct ToFrags {
    static constexpr bool kActive = true;
    half8* dst;
    ActSink sink;
    unsigned* mk;
    template <int S> DEV void slice(const f32x16& v) const {
        const float lo = RELU ? 0.f : -65504.f;
        const f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * S], lo, 65504.f),
                         __builtin_amdgcn_fmed3f(v[2 * S + 1], lo, 65504.f)};
        const half2v p = __builtin_convertvector(x, half2v);
        half8& f = dst[S >> 2];
        f[2 * (S & 3)] = p[0], f[2 * (S & 3) + 1] = p[1];
        if constexpr (STORE && (S & 3) == 3) sink.template put<BLK + (S >> 2)>(f);
        if constexpr (STORE && MBLK >= 0) {
            const unsigned pw = __builtin_bit_cast(unsigned, p);
            const unsigned bits = ((pw & 0xFFFFu) ? 1u : 0u) | ((pw >> 16) ? 2u : 0u);
            constexpr int sh = 16 * (TL & 1) + 2 * S;
            if constexpr (sh == 0) mk[TL >> 1] = bits; else mk[TL >> 1] |= bits << sh;
            if constexpr (TL == NT - 1 && S == 7)
                sink.template put_mask<MBLK>(uint4v{mk[0], NT > 2 ? mk[1] : 0u, NT > 4 ? mk[2] : 0u, NT > 6 ? mk[3] : 0u});
        }
    }
};
struct KeepRow0 {
    static constexpr bool kActive = true;
    float* out;
    template <int S> DEV void slice(const f32x16& v) const {
        if constexpr (S == 0) *out = v[0];
    }
};

struct Pipe {
    FwdStream& st;
    const char* bias;
    f32x16 acc[2];  // tile T accumulates in acc[T & 1] while tile T-1 is converted out of the other
};

// Slices of the previous tile done after k-step K of a tile of NK steps: [first, last).
constexpr int slices_per_step(int nk) { return nk >= 10 ? 1 : nk == 8 ? 2 : 3; }
constexpr int first_slice_step(int nk) { return nk >= 10 ? 2 : 1; }
constexpr int last_slice_step(int nk) { return first_slice_step(nk) + 7 / slices_per_step(nk); }
constexpr int bias_step(int nk) { return last_slice_step(nk) > nk - 4 ? last_slice_step(nk) : nk - 4; }
static_assert(last_slice_step(4) < 4 && last_slice_step(8) < 6 && last_slice_step(16) < 14, "slice schedule");

// Output tile T (= its bias tile): NK k-steps of fragments F0..; getB(k) is the B fragment of k.
// Every k-step is one scheduling region: MFMA, the LDS read of the A fragment kPrefetch steps
// ahead, a slice of the previous tile and, late in the tile, the bias seed of the next one.
template <int F0, int NK, int T, class GetB, class Prev>
DEV void run_tile(Pipe& p, GetB&& getB, const Prev& prev) {
    static_for<0, NK>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        p.acc[T & 1] = mfma(p.st.template take<F0 + k>(), getB(kc), p.acc[T & 1]);
        if constexpr (Prev::kActive) {
            constexpr int sp = slices_per_step(NK), s0 = (k - first_slice_step(NK)) * sp;
            if constexpr (k >= first_slice_step(NK))
                static_for<(s0 < 8 ? s0 : 8), (s0 + sp < 8 ? s0 + sp : 8)>(
                    [&](auto sc) { prev.template slice<decltype(sc)::value>(p.acc[(T & 1) ^ 1]); });
        }
        if constexpr (k == bias_step(NK) && T + 1 < BT_END)
            p.acc[(T & 1) ^ 1] = load_bias_tile<T + 1>(p.bias);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One layer of NT output tiles; tile t becomes fragments 2t, 2t+1 of `out` (and, when STORE,
// blocks ABLK + 2t, ABLK + 2t + 1 of the saved activations).  `before` finishes the tile that
// preceded the layer; the returned object finishes the layer's last tile.
template <int F0, int NT, int NK, int T0, bool RELU, bool STORE, int ABLK, int MBLK, class GetB, class Prev>
DEV auto run_layer(Pipe& p, const ActSink& sink, unsigned* mk, half8* out, GetB&& getB, const Prev& before) {
    run_tile<F0, NK, T0>(p, getB, before);
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        run_tile<F0 + t * NK, NK, T0 + t>(
            p, getB, ToFrags<RELU, STORE, ABLK + 2 * (t - 1), t - 1, NT, MBLK>{out + 2 * (t - 1), sink, mk});
    });
    return ToFrags<RELU, STORE, ABLK + 2 * (NT - 1), NT - 1, NT, MBLK>{out + 2 * (NT - 1), sink, mk};
}

template <int MODE, bool STORE>
__global__ __launch_bounds__(512) void mlp_fwd_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    VmQueue vm;
    FwdStream st(vm, a.packed, lds, wave, lane);
    // biases first (oldest in the queue), 1 KiB per wave-load, then the first chunks
    for (int blk = wave; blk < kBiasBytes / kFragBytes; blk += kWavesPerBlock)
        glds16(a.packed + kBiasOffset + blk * kFragBytes + lane * 16, lds + kRingBytes + blk * kFragBytes);
    st.prologue();
    const char* bias = lds + kRingBytes + h * 64;

    // ---- this lane's query point and view direction ----
    const int64_t p = (int64_t)blockIdx.x * kPointsPerBlock + wave * kPointsPerWave + (lane & 31);
    const bool valid = p < a.M;
    const int64_t pc = valid ? p : a.M - 1;
    const int64_t ray = pc / a.S;
    float x[3], vd[3];
    if (MODE == 0) {
        const float* r = a.rays + ray * a.width;
        const float zz = a.z[pc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = r[c] + r[3 + c] * zz;  // pts = rays_o + rays_d * z
            vd[c] = r[8 + c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = a.pts[pc * 3 + c];
            vd[c] = a.viewdirs[ray * 3 + c];
        }
    }
    half8 pe[4], vpe[2];
    {
        float v[32];
        encode<5, 32>(x, h, 32.f, v);
        pack_slots<32>(v, pe);
    }
    {
        float v[16];
        encode<2, 16>(vd, h, 4.f, v);
        pack_slots<16>(v, vpe);
    }
    ActSink sink{nullptr, &vm};
    if constexpr (STORE) {
        const int64_t wave_tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
        sink.base = a.act + (size_t)wave_tile * kActBlocks * kFragBytes + lane * 16;
        static_for<0, 4>([&](auto i) { sink.template put<A_PE + decltype(i)::value>(pe[decltype(i)::value]); });
        static_for<0, 2>([&](auto i) { sink.template put<A_VPE + decltype(i)::value>(vpe[decltype(i)::value]); });
    }

    half8 fa[16], fb[16];
    auto from_pe = [&](auto kc) { return pe[decltype(kc)::value]; };
    auto from_a = [&](auto kc) { return fa[decltype(kc)::value]; };
    auto from_b = [&](auto kc) { return fb[decltype(kc)::value]; };

    Pipe pp{st, bias, {}};
    unsigned mk[4] = {0u, 0u, 0u, 0u};
    st.start();
    pp.acc[0] = load_bias_tile<0>(bias);
    auto c0 = run_layer<F_L0, 8, 4, BT_L0, true, STORE, A_H0, A_MASK + 0>(pp, sink, mk, fa, from_pe, NoPrev{});                 // layer 0
    auto c1 = run_layer<F_L1 + 0 * 128, 8, 16, BT_L1 + 0, true, STORE, A_H0 + 16, A_MASK + 1>(pp, sink, mk, fb, from_a, c0);   // layers 1..4
    auto c2 = run_layer<F_L1 + 1 * 128, 8, 16, BT_L1 + 8, true, STORE, A_H0 + 32, A_MASK + 2>(pp, sink, mk, fa, from_b, c1);
    auto c3 = run_layer<F_L1 + 2 * 128, 8, 16, BT_L1 + 16, true, STORE, A_H0 + 48, A_MASK + 3>(pp, sink, mk, fb, from_a, c2);
    auto c4 = run_layer<F_L1 + 3 * 128, 8, 16, BT_L1 + 24, true, STORE, A_H0 + 64, A_MASK + 4>(pp, sink, mk, fa, from_b, c3);
    auto c5 = run_layer<F_L5, 8, 20, BT_L5, true, STORE, A_H0 + 80, A_MASK + 5>(pp, sink, mk, fb, [&](auto kc) {               // layer 5: [PE ; h]
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 4) return pe[k]; else return fa[k - 4];
    }, c4);
    auto c6 = run_layer<F_L6 + 0 * 128, 8, 16, BT_L6 + 0, true, STORE, A_H0 + 96, A_MASK + 6>(pp, sink, mk, fa, from_b, c5);   // layers 6, 7
    auto c7 = run_layer<F_L6 + 1 * 128, 8, 16, BT_L6 + 8, true, STORE, A_H0 + 112, A_MASK + 7>(pp, sink, mk, fb, from_a, c6);
    // alpha head (row 0 of its tile), then the bottleneck (linear)
    float sigma;
    run_tile<F_HEADS, 16, BT_ALPHA>(pp, from_b, c7);
    auto c8 = run_layer<F_HEADS + 16, 8, 16, BT_BOTT, false, STORE, A_BOTT, -1>(pp, sink, mk, fa, from_b, KeepRow0{&sigma});
    // view layer: [bottleneck ; view PE] -> 128, relu
    auto c9 = run_layer<F_VIEW, 4, 18, BT_VIEW, true, STORE, A_V, A_MASK + 8>(pp, sink, mk, fb, [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 16) return fa[k]; else return vpe[k - 16];
    }, c8);
    run_tile<F_RGB, 8, BT_RGB>(pp, from_b, c9);                                                             // rgb rows 0..2
    const f32x16& acc_c = pp.acc[BT_RGB & 1];
    if (valid && h == 0) a.raw[p] = make_float4(acc_c[0], acc_c[1], acc_c[2], sigma);
}

static int launch_fwd(const FwdArgs& a, int mode, hipStream_t st, const char* what) {
    const int64_t blocks = (a.M + kPointsPerBlock - 1) / kPointsPerBlock;
    if (blocks > 0x7fffffffLL) return NERF_FAIL(NERF_E_UNSUPPORTED, "%s: too many points (%lld)", what, (long long)a.M);
    (void)mode;
    mlp_fwd_kernel<0, true><<<(unsigned)blocks, 512, 0, st>>>(a);
    return check_launch(what);
}

}  // namespace nerf

using namespace nerf;

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
    FwdArgs a{};
    a.packed = (const char*)packed + lw::kPackedBytes, a.rays = rays, a.z = z, a.width = width;
    a.raw = (float4*)raw, a.act = (char*)act, a.M = n_rays * S, a.S = S;
    return launch_fwd(a, 0, as_stream(stream), "nerf_mlp_forward_train");
}
