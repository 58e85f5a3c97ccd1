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

constexpr int kRing = 8;      // LDS ring slots (16 KiB each)
constexpr int kDepth = 6;     // chunks in flight ahead of the chunk being synchronised (<= kRing - 2)
constexpr int kPrefetch = 4;  // A fragments held in registers ahead of the MFMA (<= 8)
constexpr int kPointsPerWave = 32;
constexpr int kWavesPerBlock = 8;
constexpr int kPointsPerBlock = kPointsPerWave * kWavesPerBlock;
constexpr int kRingBytes = kRing * kChunkBytes;
static_assert(kDepth <= kRing - 2 && kPrefetch <= 8, "ring slack");

template <int I, int N, class F>
DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int N>
DEV void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

DEV f32x16 mfma(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// The weight stream and the register pipeline of A fragments.  Fragments must be taken in
// stream order, F = 0, 1, 2, ...
struct Stream {
    const char* gsrc;   // this lane's source address inside chunk 0
    char* lds;          // ring base; biases follow the ring
    int wave_off;       // this wave's 2 KiB share of a chunk
    int lane_off;       // lane * 16
    int bias_off;       // kRingBytes + (lane half) * 64
    half8 a[kPrefetch];

    template <int C>
    DEV void issue() const {
        const char* g = gsrc + (size_t)C * kChunkBytes;
        char* l = lds + (C % kRing) * kChunkBytes + wave_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + kFragBytes),
                                         (__attribute__((address_space(3))) void*)(l + kFragBytes), 16, 0, 0);
    }

    // Makes chunk C readable by every wave, then refills the ring slot of chunk C + kDepth - kRing.
    template <int C>
    DEV void sync() const {
        constexpr int younger = (kChunks - 1 - C < kDepth - 1) ? (kChunks - 1 - C) : (kDepth - 1);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<2 * younger>();
        asm volatile("s_barrier" ::: "memory");
        if constexpr (C + kDepth < kChunks) issue<C + kDepth>();
        __builtin_amdgcn_sched_barrier(0);
    }

    template <int F>
    DEV half8 read() const {
        constexpr int C = F / kFragsPerChunk;
        const char* p = lds + (C % kRing) * kChunkBytes + (F % kFragsPerChunk) * kFragBytes + lane_off;
        return *(const half8*)p;
    }

    DEV void start() {
        sync<0>();
        static_for<0, kPrefetch>([&](auto i) { a[decltype(i)::value] = read<decltype(i)::value>(); });
    }

    // Returns fragment F and starts the read of fragment F + kPrefetch into its register slot.
    template <int F>
    DEV half8 take() {
        const half8 cur = a[F % kPrefetch];
        if constexpr ((F + 8) % kFragsPerChunk == 0 && (F + 8) / kFragsPerChunk < kChunks)
            sync<(F + 8) / kFragsPerChunk>();
        if constexpr (F + kPrefetch < F_END) a[F % kPrefetch] = read<F + kPrefetch>();
        return cur;
    }

    // The 32 biases of output tile T in accumulator order (packed that way): four broadcast reads.
    template <int T>
    DEV f32x16 bias_tile() const {
        const f32x4* p = (const f32x4*)(lds + bias_off + T * 128);
        const f32x4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        return f32x16{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3],
                      q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
    }
};

// Registers 8*S..8*S+7 of an accumulator tile -> one fp16 B fragment (round to nearest even),
// clamped to the fp16 range; RELU folds the activation into the clamp (one v_med3_f32).
template <int S, bool RELU>
DEV half8 to_frag(const f32x16& v) {
    half8 out;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float lo = RELU ? 0.f : -65504.f;
        f32x2 x = {__builtin_amdgcn_fmed3f(v[8 * S + j], lo, 65504.f),
                   __builtin_amdgcn_fmed3f(v[8 * S + j + 1], lo, 65504.f)};
        half2v p = __builtin_convertvector(x, half2v);
        out[j] = p[0], out[j + 1] = p[1];
    }
    return out;
}

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

// Local encoding slots of this lane half: NF frequencies starting at 1 (h = 0) or
// base_freq_h1 (h = 1); see pe_feature / view_feature in mlp_layout.h for the slot order.
template <int NF, int NSLOT>
DEV void encode(const float (&x)[3], int h, float base_freq_h1, float (&v)[NSLOT]) {
    v[0] = h ? x[2] : x[0];
    v[1] = h ? 0.f : x[1];
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = to_revolutions(h ? x[c] * base_freq_h1 : x[c]);
#pragma unroll
    for (int a = 0; a < NF; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[2 + 6 * a + c] = __builtin_amdgcn_sinf(f[c]);
            v[2 + 6 * a + 3 + c] = __builtin_amdgcn_cosf(f[c]);
            f[c] = double_revolutions(f[c]);
        }
    }
#pragma unroll
    for (int i = 2 + 6 * NF; i < NSLOT; ++i) v[i] = 0.f;
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

struct FwdArgs {
    const char* packed;
    const float* rays;  // MODE 0: [n_rays, width], z [n_rays, S]
    const float* z;
    int width;
    const float* pts;       // MODE 1: [M,3]
    const float* viewdirs;  // MODE 1: [M/S,3]
    float4* raw;            // [M] (r,g,b,sigma)
    int64_t M;              // points
    int S;                  // points per ray
};

// One output tile: NK k-steps of fragments F0.., bias tile BT.  getB(k) is the B fragment of k.
template <int F0, int NK, int BT, class GetB>
DEV f32x16 run_tile(Stream& st, GetB&& getB) {
    f32x16 acc = st.template bias_tile<BT>();
    static_for<0, NK>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        acc = mfma(st.template take<F0 + k>(), getB(kc), acc);
    });
    return acc;
}

// One layer of NT output tiles; tile t becomes fragments 2t, 2t+1 of `out`.
template <int F0, int NT, int NK, int BT0, bool RELU, class GetB>
DEV void run_layer(Stream& st, half8* out, GetB&& getB) {
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const f32x16 acc = run_tile<F0 + t * NK, NK, BT0 + t>(st, getB);
        out[2 * t] = to_frag<0, RELU>(acc);
        out[2 * t + 1] = to_frag<1, RELU>(acc);
    });
}

template <int MODE>
__global__ __launch_bounds__(512) void mlp_fwd_kernel(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    Stream st;
    st.gsrc = a.packed + wave * (2 * kFragBytes) + lane * 16;
    st.lds = lds;
    st.wave_off = wave * (2 * kFragBytes);
    st.lane_off = lane * 16;
    st.bias_off = kRingBytes + h * 64;
    // biases first (oldest in the vmcnt queue), 1 KiB per wave-load, then the first chunks
    for (int blk = wave; blk < kBiasBytes / kFragBytes; blk += kWavesPerBlock)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a.packed + kBiasOffset + blk * kFragBytes + lane * 16),
            (__attribute__((address_space(3))) void*)(lds + kRingBytes + blk * kFragBytes), 16, 0, 0);
    static_for<0, kDepth>([&](auto c) { st.template issue<decltype(c)::value>(); });

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

    half8 fa[16], fb[16];
    auto from_pe = [&](auto kc) { return pe[decltype(kc)::value]; };
    auto from_a = [&](auto kc) { return fa[decltype(kc)::value]; };
    auto from_b = [&](auto kc) { return fb[decltype(kc)::value]; };

    st.start();
    run_layer<F_L0, 8, 4, BT_L0, true>(st, fa, from_pe);                  // layer 0: PE -> 256
    run_layer<F_L1 + 0 * 128, 8, 16, BT_L1 + 0, true>(st, fb, from_a);    // layers 1..4
    run_layer<F_L1 + 1 * 128, 8, 16, BT_L1 + 8, true>(st, fa, from_b);
    run_layer<F_L1 + 2 * 128, 8, 16, BT_L1 + 16, true>(st, fb, from_a);
    run_layer<F_L1 + 3 * 128, 8, 16, BT_L1 + 24, true>(st, fa, from_b);
    run_layer<F_L5, 8, 20, BT_L5, true>(st, fb, [&](auto kc) {            // layer 5: [PE ; h]
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 4) return pe[k]; else return fa[k - 4];
    });
    run_layer<F_L6 + 0 * 128, 8, 16, BT_L6 + 0, true>(st, fa, from_b);    // layers 6, 7
    run_layer<F_L6 + 1 * 128, 8, 16, BT_L6 + 8, true>(st, fb, from_a);
    // alpha head (row 0 of its tile), then the bottleneck (linear)
    const f32x16 acc_a = run_tile<F_HEADS, 16, BT_ALPHA>(st, from_b);
    const float sigma = acc_a[0];
    run_layer<F_HEADS + 16, 8, 16, BT_BOTT, false>(st, fa, from_b);
    // view layer: [bottleneck ; view PE] -> 128, relu
    run_layer<F_VIEW, 4, 18, BT_VIEW, true>(st, fb, [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 16) return fa[k]; else return vpe[k - 16];
    });
    const f32x16 acc_c = run_tile<F_RGB, 8, BT_RGB>(st, from_b);          // rgb rows 0..2
    if (valid && h == 0) a.raw[p] = make_float4(acc_c[0], acc_c[1], acc_c[2], sigma);
}

static int launch_fwd(const FwdArgs& a, int mode, hipStream_t st, const char* what) {
    const int64_t blocks = (a.M + kPointsPerBlock - 1) / kPointsPerBlock;
    if (blocks > 0x7fffffffLL) return NERF_FAIL(NERF_E_UNSUPPORTED, "%s: too many points (%lld)", what, (long long)a.M);
    if (mode == 0)
        mlp_fwd_kernel<0><<<(unsigned)blocks, 512, 0, st>>>(a);
    else
        mlp_fwd_kernel<1><<<(unsigned)blocks, 512, 0, st>>>(a);
    return check_launch(what);
}

}  // namespace nerf

using namespace nerf;

NERF_API int nerf_mlp_forward_rays(const void* packed, const float* rays, int width,
                                   const float* z, int64_t n_rays, int S, float* raw,
                                   nerf_stream_t stream) {
    NERF_REQUIRE(packed && rays && z && raw, "nerf_mlp_forward_rays: null pointer");
    NERF_REQUIRE(width >= 11, "nerf_mlp_forward_rays: ray batch needs view directions (width >= 11, got %d)", width);
    NERF_REQUIRE(n_rays >= 0 && S > 0, "nerf_mlp_forward_rays: bad size");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0,
                 "nerf_mlp_forward_rays: packed and raw must be 16-byte aligned");
    if (n_rays == 0) return 0;
    FwdArgs a{};
    a.packed = (const char*)packed, a.rays = rays, a.z = z, a.width = width;
    a.raw = (float4*)raw, a.M = n_rays * S, a.S = S;
    return launch_fwd(a, 0, as_stream(stream), "nerf_mlp_forward_rays");
}

NERF_API int nerf_mlp_forward_pts(const void* packed, const float* pts, const float* viewdirs,
                                  int64_t n, int S, float* raw, nerf_stream_t stream) {
    NERF_REQUIRE(packed && pts && viewdirs && raw, "nerf_mlp_forward_pts: null pointer");
    NERF_REQUIRE(n >= 0 && S > 0, "nerf_mlp_forward_pts: bad size");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0,
                 "nerf_mlp_forward_pts: packed and raw must be 16-byte aligned");
    if (n == 0) return 0;
    FwdArgs a{};
    a.packed = (const char*)packed, a.pts = pts, a.viewdirs = viewdirs;
    a.raw = (float4*)raw, a.M = n * S, a.S = S;
    return launch_fwd(a, 1, as_stream(stream), "nerf_mlp_forward_pts");
}
