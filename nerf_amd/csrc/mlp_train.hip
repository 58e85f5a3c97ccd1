// Backward of the NeRF MLP (the tf.GradientTape of run_nerf.py:770-826 restricted to the network):
// gradients of the 24 parameter arrays from d raw [M,4] and the activations the training forward
// saved.  Three kernels:
//
//  1. mlp_bwd_chain_kernel -- d h_l = W_{l+1} . d pre_{l+1}, masked by h_l > 0, from the rgb/alpha
//     heads down to layer 0.  Same structure as the forward kernel: transposed products with
//     v_mfma_f32_32x32x16_f16, the gradient of a 32-point wave tile stays in registers, the
//     (transposed-packed) weights stream L2 -> LDS once per workgroup.  ReLU masks are the bit
//     blocks the forward saved (one 16-byte register load per layer, issued a layer ahead).
//     Every d pre_l is written out as fp16 fragments.  Gradients are multiplied by `gscale` on
//     entry so that they sit inside the fp16 range.
//  2. mlp_dw_kernel -- per layer, dW^T[out][in] = sum over points of d pre[out][pt] * x[pt][in].
//     The contraction runs over points, which the stored fragments keep on the lane, so every
//     operand is rebuilt with the transposing LDS read (ds_read_b64_tr_b16).  One workgroup
//     owns a (layer, point range) pair, keeps the whole dW^T of the layer in accumulators
//     (wave w = output tile w) and stores it as its own partial tiles once at the end; a small
//     kernel then adds the partial tiles of a job in workgroup order (no atomics: the same inputs
//     give the same bits).  Bias
//     gradients are the sums of the gradient operand over the points.  Layers whose input is
//     a concatenation (layer 5) and products that share an operand (the view layer's gradient
//     against h7 and against the view encoding, the alpha head's gradient against h7) are one job
//     each, so every saved block is read once.  HBM-bound: ~8.9 KB/point read.
//  3. mlp_grad_unpack_kernel -- accumulator-ordered sums -> the parameter arrays, times 1/gscale.
//  4. mlp_head_compose_kernel -- the bottleneck is linear, so its gradient and that of the view
//     layer's bottleneck rows are small matrix products of P = sum_points h7 (x) d v_pre with the
//     two layers' own fp32 weights (mlp_layout.h, at kActBlocks); neither the bottleneck nor its
//     gradient is ever stored.
//
// The gradient scale.  The chain carries gradients in fp16, so d raw is multiplied by a power of
// two on entry and the result divided by it on exit.  By default the factor is formed on the
// device from max |d raw| (draw_absmax_kernel) so that the largest scaled element lies in
// [T, 2T) for a power of two T the caller chooses (256 unless told otherwise: 128x of headroom to
// the fp16 maximum for growth inside the chain, 2^22 of range below it before the smallest normal
// number, which is what the many small entries of d raw -- samples of little weight -- need; a
// caller whose networks amplify gradients more lowers T).  The weight-gradient kernel, which reads
// every stored layer gradient anyway, records the largest fp16 bit pattern it saw; the unpack
// kernel turns that into a status word (saturated / non-finite, or everything below the smallest
// normal number) and, when it is set, writes NaN instead of numbers nobody should trust.
#include "mlp_common.h"
#include "mlp_layout_wide.h"

namespace nerf {

// ---------------------------------------------------------------------------------------------
// status block (the first kStatusBytes of the caller's `partial` scratch) and the gradient scale
// ---------------------------------------------------------------------------------------------
constexpr int kStatusBytes = 256;
enum StatusWord {
    ST_ABSMAX = 0,   // in: bit pattern of max |d raw| (a non-negative float orders like its bits; NaN sorts above inf)
    ST_YMAX = 1,     // in: largest |fp16| bit pattern among the stored layer gradients
    ST_FLAGS = 4,    // out: NERF_GRAD_* bits
    ST_GSCALE = 5,   // out: the factor that was applied (float)
    ST_ABSMAX_OUT = 6,  // out: max |d raw| (float)
    ST_YMAX_OUT = 7,    // out: copy of ST_YMAX
};

// The power of two that brings mx into [2^target_exp, 2^(target_exp+1)); 1 when mx is zero,
// denormal or not finite.
__host__ __device__ inline float auto_gscale(unsigned absmax_bits, int target_exp) {
    const int e = (int)((absmax_bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    int f = 127 + target_exp - (e - 127);
    f = f < 1 ? 1 : f > 254 ? 254 : f;
    const unsigned bits = (unsigned)f << 23;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, bits);
#else
    float r;
    memcpy(&r, &bits, sizeof r);
    return r;
#endif
}

__global__ __launch_bounds__(256) void draw_absmax_kernel(const float4* __restrict__ d_raw, int64_t n,
                                                          unsigned* __restrict__ status) {
    __shared__ unsigned part[4];
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = d_raw[i];
        const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
        const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned q = ab > cd ? ab : cd;
        m = m > q ? m : q;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = m > o ? m : o;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup (a few hundred on one word; one per wave of a large batch would
    // queue thousands behind each other)
    if (threadIdx.x == 0) {
        const unsigned a = part[0] > part[1] ? part[0] : part[1], b = part[2] > part[3] ? part[2] : part[3];
        const unsigned all = a > b ? a : b;
        if (all) atomicMax(status + ST_ABSMAX, all);
    }
}

// ---------------------------------------------------------------------------------------------
// backward weight stream
// ---------------------------------------------------------------------------------------------
struct ParamPtrsB {
    const float* p[24];
};

// fp32 arrays kept behind the stream for mlp_head_compose_kernel, in blocks of 256 floats
constexpr int kBwdCopyBlocks = (256 * 256 + 256 + 256 * 128) / 256;

__global__ __launch_bounds__(64) void mlp_pack_bwd_kernel(ParamPtrsB pp, char* __restrict__ packed) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int F = blockIdx.x;
    if (F >= B_END) {  // bottleneck kernel, bottleneck bias, the view layer's first 256 rows: contiguous at kBwdBottW
        const int q = (F - B_END) * 64 + lane;  // float4 index
        const float4* src = q < 16384 ? (const float4*)pp.p[16] + q
                          : q < 16384 + 64 ? (const float4*)pp.p[17] + (q - 16384)
                                           : (const float4*)pp.p[18] + (q - 16384 - 64);
        ((float4*)(packed + kBwdBottW))[q] = *src;
        return;
    }
    const float* Wm = nullptr;  // kernel [in, out]
    int out_dim = 0, in = -1, in_rows = 0;
    int outs[8];
    for (int j = 0; j < 8; ++j) outs[j] = -1;
    if (F < B_BOTT) {                       // d v = W_rgb . d rgb
        Wm = pp.p[20], out_dim = 3, in_rows = 128, in = 32 * F + r;
        for (int j = 0; j < 8; ++j) { const int f = acc_feature(0, h, j); outs[j] = f < 3 ? f : -1; }
    } else if (F < B_H7) {                  // d bott = W_view[0:256] . d v_pre
        const int q = F - B_BOTT, t = q / 8, k = q % 8;
        Wm = pp.p[18], out_dim = 128, in_rows = 256, in = 32 * t + r;
        for (int j = 0; j < 8; ++j) outs[j] = acc_feature(k, h, j);
    } else if (F < B_H6) {                  // d h7 = W_bott . d bott + W_alpha . d alpha
        const int q = F - B_H7, t = q / 17, k = q % 17;
        in = 32 * t + r, in_rows = 256;
        if (k < 16) {
            Wm = pp.p[16], out_dim = 256;
            for (int j = 0; j < 8; ++j) outs[j] = acc_feature(k, h, j);
        } else {
            Wm = pp.p[22], out_dim = 1;
            for (int j = 0; j < 8; ++j) outs[j] = acc_feature(0, h, j) == 0 ? 0 : -1;
        }
    } else if (F < B_END_USED) {            // d h_l = W_{l+1} . d pre_{l+1}, l = 6..0
        const int q = F - B_H6, step = q / 128, t = (q % 128) / 16, k = q % 16;
        const int layer = 7 - step;         // the kernel used: layers 7, 6, 5, 4, 3, 2, 1
        Wm = pp.p[2 * layer], out_dim = 256, in_rows = (layer == 5) ? 319 : 256;
        in = 32 * t + r + (layer == 5 ? 63 : 0);
        for (int j = 0; j < 8; ++j) outs[j] = acc_feature(k, h, j);
    }
    _Float16* dst = (_Float16*)(packed + (size_t)F * kFragBytes) + lane * 8;
    for (int j = 0; j < 8; ++j) {
        const float v = (Wm && in < in_rows && outs[j] >= 0) ? Wm[(size_t)in * out_dim + outs[j]] : 0.f;
        dst[j] = (_Float16)v;
    }
}

// ---------------------------------------------------------------------------------------------
// 1. gradient chain
// ---------------------------------------------------------------------------------------------
constexpr int kBwdRing = 8, kBwdDepth = 6;
using BwdStream = Stream<kBwdChunks, kBwdRing, kBwdDepth>;
constexpr int kBwdLds = BwdStream::kRingBytes;
static_assert(kBwdLds <= 160 * 1024, "LDS budget");

typedef short short8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// The ReLU mask bits of one layer for this lane's point, loaded by inline asm and counted in the
// VmQueue like the weight stream's loads (a load the compiler knows about would make it wait for
// vmcnt(0), i.e. for the whole stream and every pending store, before the first use).
// The forward (16x16x32 tiles: lane = point & 15 + 16 g, g = 0..3) saved, per lane, dwords
// [point tile][2]: with i = 2 t + (r >> 1), bit (i & 15) + 16 (r & 1) of dword i >> 4 = unit
// 16 t + 4 g + r.  A lane of this kernel
// (32x32x16 tiles: point = lane & 31, half h) holds units with (unit >> 2) & 1 == h, i.e. the
// pieces g = h and g = h + 2 of its point: two 8-byte loads, v[0:1] and v[2:3].
struct MaskLoad {
    uint4v v;
    int mark;
    DEV void issue(VmQueue& vm, const char* block, int lane) {
        const int p = lane & 15, pt = (lane >> 4) & 1, h = lane >> 5;
        const char* g0 = block + 16 * (p + 16 * h) + 8 * pt;
        unsigned long long lo, hi;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lo) : "v"(g0) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(hi) : "v"(g0 + 16 * 32) : "memory");
        v = uint4v{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        vm.issued += 2;
        mark = vm.issued;
    }
    // The loaded value; every use is ordered behind the counted wait.
    DEV uint4v get(const VmQueue& vm) {
        vm.wait_until(mark);
        asm volatile("" : "+v"(v));
        return v;
    }
};

struct GradSink {
    char* base;
    VmQueue* vm;
    template <int BLK>
    DEV void put(const half8& v) const {
        __builtin_nontemporal_store(v, (half8*)(base + BLK * kFragBytes));  // streamed: keep L2 for the weights
        vm->issued += 1;
    }
};

struct ChainArgs {
    const char* packed_bwd;
    const char* act;
    const float4* d_raw;  // [M]
    char* gbuf;           // [wave tiles][kGradBlocks][1 KiB]
    int64_t M;
    float gscale;         // > 0: the caller's factor; otherwise formed from status[ST_ABSMAX]
    int target_exp;       // ... so that the largest scaled element lies in [2^target_exp, 2^(target_exp+1))
    const unsigned* status;
};

struct ChainPipe {
    BwdStream& st;
    f32x16 acc[2];  // tile t accumulates in acc[t & 1] while tile t-1 is converted out of the other
};

// What happens to a finished gradient tile: eight slices (two registers -> one packed dword,
// saturated to the fp16 range by the conversion, multiplied by the unit's ReLU mask bits) run between the MFMAs of the
// following tile; finished fragments are stored as blocks GBLK + 2*TL, +1 for the dW kernel
// (GBLK < 0: not stored -- the bottleneck's gradient only feeds the next layer of the chain).
struct CNoPrev {
    static constexpr bool kActive = false;
    template <int S> DEV void slice(const f32x16&) const {}
};
template <bool MASKED, int TL, int GBLK>
struct CToFrags {
    static constexpr bool kActive = true;
    half8* dst;   // the layer's fragment set
    GradSink sink;
    uint4v mk;    // the layer's mask bits for this lane's point (see MaskLoad)
    template <int S> DEV void slice(const f32x16& v) const {
        const f32x2 x = {v[2 * S], v[2 * S + 1]};
        half2v p = __builtin_convertvector(x, half2v);  // saturating (saturate_fp16_conversions)
        if constexpr (MASKED) {
            // registers 2S, 2S+1 of 32-unit tile TL = units 16 t + 4 g + r with t = 2 TL + (S >> 2),
            // g = h + 2 ((S >> 1) & 1), r = 2 (S & 1) (+1): pair i = 2 t + (S & 1) of the forward's
            // numbering, flags at bits (i & 15) and (i & 15) + 16 of dword i >> 4 (see MaskLoad)
            constexpr int i = 2 * (2 * TL + (S >> 2)) + (S & 1), piece = (S >> 1) & 1;
            const unsigned m = (mk[2 * piece + (i >> 4)] >> (i & 15)) & 0x00010001u;
            unsigned kept;  // each half times its flag (inline asm: the compiler scalarises packed u16 arithmetic)
            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(kept) : "v"(__builtin_bit_cast(unsigned, p)), "v"(m));
            p = __builtin_bit_cast(half2v, kept);
        }
        half8& f = dst[2 * TL + (S >> 2)];
        f[2 * (S & 3)] = p[0], f[2 * (S & 3) + 1] = p[1];
        if constexpr ((S & 3) == 3 && GBLK >= 0) sink.template put<GBLK + 2 * TL + (S >> 2)>(f);
    }
};

constexpr int c_first_slice_step(int nk) { return nk >= 10 ? 2 : nk >= 2 ? 1 : 0; }
constexpr int c_slices_per_step(int nk) { return nk >= 10 ? 1 : nk >= 5 ? 2 : 8; }

// Gradient tile PAR (parity of its number): NK k-steps of fragments F0..; one scheduling region
// per k-step (MFMA, LDS read of the A fragment kPrefetch steps ahead, slices of the previous tile).
template <int F0, int NK, int PAR, class GetB, class Prev>
DEV void chain_tile(ChainPipe& p, GetB&& getB, const Prev& prev) {
    static_for<0, NK>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const half8 a = p.st.template take<F0 + k>();
        if constexpr (k == 0)
            p.acc[PAR] = mfma(a, getB(kc), f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
        else
            p.acc[PAR] = mfma(a, getB(kc), p.acc[PAR]);
        if constexpr (Prev::kActive && k >= c_first_slice_step(NK)) {
            constexpr int sp = c_slices_per_step(NK), s0 = (k - c_first_slice_step(NK)) * sp;
            static_for<(s0 < 8 ? s0 : 8), (s0 + sp < 8 ? s0 + sp : 8)>(
                [&](auto sc) { prev.template slice<decltype(sc)::value>(p.acc[PAR ^ 1]); });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One layer of the chain: NT (even) gradient tiles of NK k-steps into the fragment set `out`.
template <int F0, int NT, int NK, bool MASKED, int GBLK, class GetB, class Prev>
DEV auto chain_layer(ChainPipe& p, const GradSink& sink, const uint4v& mk, half8* out, GetB&& getB, const Prev& before) {
    static_assert(NT % 2 == 0, "tile parity restarts at every layer");
    chain_tile<F0, NK, 0>(p, getB, before);
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        chain_tile<F0 + t * NK, NK, t & 1>(p, getB, CToFrags<MASKED, t - 1, GBLK>{out, sink, mk});
    });
    return CToFrags<MASKED, NT - 1, GBLK>{out, sink, mk};
}

__global__ __launch_bounds__(512) void mlp_bwd_chain_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[kBwdLds];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int64_t wave_tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    saturate_fp16_conversions();

    VmQueue vm;
    BwdStream st(vm, a.packed_bwd, lds, wave, lane);
    st.prologue();
    const char* masks = a.act + ((size_t)wave_tile * kActBlocks + A_MASK) * kFragBytes;
    MaskLoad mv, m7, m6, m5, m4, m3, m2, m1, m0;
    mv.issue(vm, masks + 8 * kFragBytes, lane);
    m7.issue(vm, masks + 7 * kFragBytes, lane);
    GradSink sink{a.gbuf + (size_t)wave_tile * kGradBlocks * kFragBytes + lane * 16, &vm};

    const int64_t p = (int64_t)blockIdx.x * kPointsPerBlock + wave * kPointsPerWave + (lane & 31);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < a.M && h == 0) g = a.d_raw[p];
    half8 dC = {0, 0, 0, 0, 0, 0, 0, 0}, dA = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const float gs = a.gscale > 0.f ? a.gscale : auto_gscale(a.status[ST_ABSMAX], a.target_exp);
        // saturating conversions (a NaN stays a NaN and is reported by the status word)
        const half2v xy = __builtin_convertvector(f32x2{g.x * gs, g.y * gs}, half2v);
        const half2v zw = __builtin_convertvector(f32x2{g.z * gs, g.w * gs}, half2v);
        dC[0] = xy[0], dC[1] = xy[1], dC[2] = zw[0];
        dA[0] = zw[1];
    }
    sink.template put<G_DA>(dA);
    sink.template put<G_DC>(dC);

    half8 fa[16], fb[16];
    auto from_a = [&](auto kc) { return fa[decltype(kc)::value]; };
    auto from_b = [&](auto kc) { return fb[decltype(kc)::value]; };
    const uint4v none = {0u, 0u, 0u, 0u};

    ChainPipe pp{st, {}};
    st.start();
    // each layer's mask bits are requested one layer before they are needed
    auto c0 = chain_layer<B_V, 4, 1, true, G_DV>(pp, sink, mv.get(vm), fa, [&](auto) { return dC; }, CNoPrev{});   // d v_pre
    m6.issue(vm, masks + 6 * kFragBytes, lane);
    auto c1 = chain_layer<B_BOTT, 8, 8, false, -1>(pp, sink, none, fb, from_a, c0);                                // d bott (not stored)
    auto c2 = chain_layer<B_H7, 8, 17, true, G_DPRE0 + 16 * 7>(pp, sink, m7.get(vm), fa, [&](auto kc) {            // d pre7
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 16) return fb[k]; else return dA;
    }, c1);
    m5.issue(vm, masks + 5 * kFragBytes, lane);
    auto c3 = chain_layer<B_H6 + 0 * 128, 8, 16, true, G_DPRE0 + 16 * 6>(pp, sink, m6.get(vm), fb, from_a, c2);    // d pre6
    m4.issue(vm, masks + 4 * kFragBytes, lane);
    auto c4 = chain_layer<B_H6 + 1 * 128, 8, 16, true, G_DPRE0 + 16 * 5>(pp, sink, m5.get(vm), fa, from_b, c3);    // d pre5
    m3.issue(vm, masks + 3 * kFragBytes, lane);
    auto c5 = chain_layer<B_H6 + 2 * 128, 8, 16, true, G_DPRE0 + 16 * 4>(pp, sink, m4.get(vm), fb, from_a, c4);    // d pre4
    m2.issue(vm, masks + 2 * kFragBytes, lane);
    auto c6 = chain_layer<B_H6 + 3 * 128, 8, 16, true, G_DPRE0 + 16 * 3>(pp, sink, m3.get(vm), fa, from_b, c5);    // d pre3
    m1.issue(vm, masks + 1 * kFragBytes, lane);
    auto c7 = chain_layer<B_H6 + 4 * 128, 8, 16, true, G_DPRE0 + 16 * 2>(pp, sink, m2.get(vm), fb, from_a, c6);    // d pre2
    m0.issue(vm, masks + 0 * kFragBytes, lane);
    auto c8 = chain_layer<B_H6 + 5 * 128, 8, 16, true, G_DPRE0 + 16 * 1>(pp, sink, m1.get(vm), fa, from_b, c7);    // d pre1
    auto c9 = chain_layer<B_H6 + 6 * 128, 8, 16, true, G_DPRE0 + 16 * 0>(pp, sink, m0.get(vm), fb, from_a, c8);    // d pre0
    static_for<0, 8>([&](auto sc) { c9.template slice<decltype(sc)::value>(pp.acc[1]); });
}

// ---------------------------------------------------------------------------------------------
// 2. weight gradients
// ---------------------------------------------------------------------------------------------
constexpr int kDwStageBytes = 40 * kFragBytes;  // one ring stage: a group of wave tiles of one job
constexpr int kDwStages = 4;
constexpr int kDwAhead = 3;      // input operands in flight ahead of their MFMA
constexpr int kNumJobs = 10;     // what a workgroup can be given (see make_jobs)
constexpr int kNumProducts = 12; // (layer input, layer gradient) products the unpack kernel reads

// One job = one gradient segment y (NY tiles of 32 outputs) against one or two input segments x
// (NX0 + NX1 tiles of 32 inputs), optionally against a second, single-row gradient y1 (the alpha
// head shares the bottleneck's input).
struct DwJob {
    int kind;            // which dw_body instantiation (see mlp_dw_kernel)
    int x0_blk, x1_blk;  // input segments: first block in the activation buffer
    int y_blk, y1_blk;   // gradient segments: first block in the gradient buffer
    int y_k;             // blocks of y (1 or even)
    int n_out;           // output tiles (<= 8): wave w owns output tile w
    int tile0;           // first 32x32 tile of this job in the summed buffer
    int n_acc;           // accumulator tiles of this job (what one workgroup stores)
    int slab0;           // first tile of this job's per-workgroup partial tiles: workgroup slice s stores
                         // tiles [slab0 + s * n_acc, slab0 + (s + 1) * n_acc)
    int used;            // workgroups of this job that own at least one wave tile (the others store nothing)
    int wg0, wgs;        // workgroups [wg0, wg0 + wgs) split the wave tiles of this job
};

struct DwArgs {
    const char* act;
    const char* gbuf;
    float* slabs;        // [partial tiles][16 regs][64 lanes]: every workgroup's own sums
    unsigned* status;
    int64_t wave_tiles;
    DwJob job[kNumJobs];
#ifdef DW_STAMPS  // diagnostic build only: per-workgroup (begin, end, job) timestamps
    unsigned long long* stamps;
#endif
};

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

// LDS image of one 32-feature tile (a pair of stored blocks) of a staged wave tile: 2 KiB laid
// out [point][4 granules][8 features], 64 B per point, so that the 32 lanes of a half-wave,
// which read 4 points x 64 B with one transposing read, touch every bank once.  Feature slot
// 8 * granule + element of the image becomes row (or column) 8 * granule + element of the
// 32 x 32 product.  The direct-to-LDS load writes lane-linear, so the permutation is applied to
// the SOURCE address: LDS lane L of load n (n = 0, 1) fetches point 16n + L/4, granule L & 3.
//  - gradient pairs (32x32x16 fragments of the chain kernel: block = 16 features, lane = point
//    + 32 half): granule = 2 (block of the pair) + half;
//  - input pairs (16x16x32 fragments of the forward kernel: block = one 16-point tile x 32
//    features, lane = point + 16 g): granule = g, and load n reads block n.
constexpr int kTileImgBytes = 2 * kFragBytes;

// Per-lane byte offset of a gradient-image load inside its pair of stored blocks (load n = 0, 1
// of the image adds 256 n bytes, which the caller folds into the wave-uniform base address).
DEV unsigned stage_lane_offset(int lane, bool one_block) {
    const int c = lane >> 2, sb = one_block ? 0 : (lane >> 1) & 1, hh = lane & 1;
    return (unsigned)(sb * kFragBytes + (32 * hh + c) * 16);
}

// Operand fragment (A or B alike) of a staged tile image x the 16 points 16*kp .. 16*kp+15:
// lane (r = feature slot, hh) gets points 8*hh .. 8*hh+7 of the 16.  `one_block`: the segment
// has a single k-step; the upper 16 feature slots read as zero.
DEV half8 tr_operand(const char* img, int kp, int lane, bool one_block) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int c = 16 * kp + 8 * (g >> 1) + q;
    const char* base = img + c * 64 + (g & 1) * 32 + (p >> 1) * 16 + (p & 1) * 8;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)base);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + 4 * 64));
    short8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (one_block && (g & 1)) v = (short8)0;
    return __builtin_bit_cast(half8, v);
}

// Stages a group of G wave tiles of a job into one ring stage (a tile index past the end
// re-loads the last tile: the loads stay in bounds and a wave issues the same number of loads
// for every stage).  The 1 KiB loads of a stage are dealt round-robin to the 8 waves.  Which
// block a load fetches is wave-uniform (scalar arithmetic);
// the lanes only add one of two offsets.  Everything is passed by value: an object
// whose address is taken would live in scratch and be re-read after every inline-asm load.
struct DwStageSrc {
    const char* act;
    const char* gbuf;
    int x0_blk, x1_blk, y_blk, y1_blk;
    bool y_one;
    int64_t t_lo, t_hi;
    unsigned off_two, off_one;  // per-lane offsets of gradient images: two-block / one-block
    unsigned off_x;             // per-lane offset of input images
};

template <int NX0, int NX1, int NY, int NY1, int G>
DEV void dw_stage_in(const DwStageSrc s, int64_t group, char* dst, int wave) {
    constexpr int NX = NX0 + NX1;
    constexpr int kTileUnits = 2 * (NX + NY + NY1), kUnits = kTileUnits * G;
    constexpr int kMine = (kUnits + kWavesPerBlock - 1) / kWavesPerBlock;
    // base addresses of the G wave tiles once per stage (64-bit scalar multiplies), then a few
    // 32-bit scalar operations per load
    const char* abase[G];
    const char* gbase[G];
    const unsigned dst_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
#pragma unroll
    for (int gt = 0; gt < G; ++gt) {
        int64_t wt = s.t_lo + group * G + gt;
        if (wt >= s.t_hi) wt = s.t_hi - 1;
        abase[gt] = s.act + (size_t)wt * kActBlocks * kFragBytes;
        gbase[gt] = s.gbuf + (size_t)wt * kGradBlocks * kFragBytes;
    }
#pragma unroll
    for (int m = 0; m < kMine; ++m) {
        const int u = wave + m * kWavesPerBlock;
        if (u >= kUnits) break;  // only in the last round, only for waves >= kUnits % 8
        int gt = 0, r = u;
        if constexpr (G > 1) gt = u / kTileUnits, r = u % kTileUnits;
        const int img = r >> 1, n = r & 1;
        const bool is_x = img < NX;
        const int blk = img < NX0 ? s.x0_blk + 2 * img
                      : img < NX ? s.x1_blk + 2 * (img - NX0)
                      : img < NX + NY ? s.y_blk + 2 * (img - NX)
                                      : s.y1_blk;
        const bool one = !is_x && (img >= NX + NY || s.y_one);
        const char* ab = abase[0];
        const char* gb = gbase[0];
#pragma unroll
        for (int k = 1; k < G; ++k)
            if (gt == k) ab = abase[k], gb = gbase[k];
        // load n of an image = points 16 n .. 16 n + 15: block n of an input pair (one point tile
        // per block), the second 256 bytes of each lane half of a gradient pair
        const char* src = (is_x ? ab : gb) + (unsigned)(blk * kFragBytes + n * (is_x ? kFragBytes : 256));
        glds16_untracked_sbase<true>(src, is_x ? s.off_x : one ? s.off_one : s.off_two, dst_addr + (unsigned)(u * kFragBytes));
    }
}

// One workgroup's share of a job: NX = NX0 + NX1 input tiles (+ the bias column when BIAS), NY
// gradient tiles and, when NY1, one more single-row gradient tile per wave tile.  G wave tiles form
// one ring stage (one barrier, one counted wait), chosen so that a stage is 26-40 KiB whatever the
// job: a workgroup then always has ~100 KiB of loads in flight, which is what keeps the narrow
// jobs (rgb, encoding rows) off the latency bound.  The loads of a stage are dealt round-robin to
// the 8 waves (kMine each, one fewer for the waves past the remainder), so the vmcnt bookkeeping
// is a compile-time constant per wave class.  The kernel's time is set by the number of 1 KiB
// loads a workgroup issues (measured: 48 ns each per CU, whatever the job), so no load is
// issued twice and workgroups are handed out in proportion to the loads of a job.
//
// Accumulator tiles of the partial-sum buffer, from jb.tile0: output tile w x input i at
// w * NI + i (i = NX: the bias, in column 0); the single-row gradient x tile i of the FIRST input
// segment at n_out * NI + i (i = NX0: its bias).
//
// ROW: the job has one output tile (rgb).  Giving it to one wave would leave that wave alone with
// every MFMA of the job, so the waves split the INPUT tiles instead: wave w accumulates the
// product with input tile w (wave 0 also takes the bias).  The single-row gradient of an NY1
// job is handled the same way (wave w: tile w of the first segment) on top of the wave's own
// output tile.
template <int NX0, int NX1, int NY, int NY1, int BIAS, int G, bool ROW = false>
DEV void dw_body(const DwArgs& a, const DwJob& jb, char* lds, int lane, int wave) {
    constexpr int NX = NX0 + NX1, NI = NX + BIAS;
    static_assert(!ROW || (NY == 1 && NY1 == 0 && NX <= kWavesPerBlock), "row jobs: one gradient tile, one input tile per wave");
    static_assert(NY1 == 0 || (NY1 == 1 && !ROW && NX0 <= kWavesPerBlock), "a second gradient is a single row against the first input segment, split over the waves");
    constexpr int kTileUnits = 2 * (NX + NY + NY1);            // 1 KiB loads per wave tile
    constexpr int kUnits = kTileUnits * G;                     // per stage
    constexpr int kMine = (kUnits + kWavesPerBlock - 1) / kWavesPerBlock;  // loads per stage of waves < kFullWaves,
    constexpr int kFullWaves = kUnits - (kMine - 1) * kWavesPerBlock;       // the other waves issue one fewer
    constexpr int kTileBytes = kTileUnits * kFragBytes;
    static_assert(kTileBytes * G <= kDwStageBytes, "stage size");
    const int slice = blockIdx.x - jb.wg0;
    const int64_t per = (a.wave_tiles + jb.wgs - 1) / jb.wgs;
    const int64_t t_lo = slice * per, t_hi = (t_lo + per < a.wave_tiles) ? t_lo + per : a.wave_tiles;
    const bool y_one = jb.y_k == 1;
    const int64_t ntile = t_hi - t_lo;
    const int64_t nstage = (ntile + G - 1) / G;

    const DwStageSrc src{a.act, a.gbuf, jb.x0_blk, jb.x1_blk, jb.y_blk, jb.y1_blk, y_one, t_lo, t_hi,
                         stage_lane_offset(lane, false), stage_lane_offset(lane, true),
                         (unsigned)((lane & 3) * 256 + (lane >> 2) * 16)};
    auto stage_in = [=](int64_t group, int sbuf) {
        dw_stage_in<NX0, NX1, NY, NY1, G>(src, group, lds + sbuf * kDwStageBytes, wave);
    };

    constexpr int NACC = ROW ? 1 : NX;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = zero16;
    f32x16 accr = zero16;  // NY1: single-row gradient x tile `wave` of the first input segment
    // Bias gradients = sums of the gradient operand over the points.  A lane of that operand holds
    // 8 points of output row (lane & 31), so one fp32 register per lane takes the sum (v_dot2 with
    // ones) and the two lane halves meet in the atomic add at the end.
    float bsum = 0.f, bsum_r = 0.f;
    // largest |fp16| bit pattern among the gradient operands this wave reads (two per dword): the
    // status word's evidence that nothing in the chain saturated, became NaN or sank into denormals
    unsigned ymax = 0u;
    auto track = [&](const half8& v) {
        typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
        typedef unsigned uint4t __attribute__((ext_vector_type(4)));
        const uint4t w = __builtin_bit_cast(uint4t, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ushort2v cur = __builtin_bit_cast(ushort2v, w[j] & 0x7fff7fffu);
            ymax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ushort2v, ymax), cur));
        }
    };
    auto point_sum = [](const half8& v, float s) {
        const half2v one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int j = 0; j < 8; j += 2) s = __builtin_amdgcn_fdot2(half2v{v[j], v[j + 1]}, one, s, false);
        return s;
    };

    constexpr int kAhead = kDwStages - 1;  // stages in flight beyond the one being consumed
#pragma unroll
    for (int k = 0; k < kAhead; ++k)
        if (k < nstage) stage_in(k, k);
    // The stage loop is unrolled over the ring so that every LDS address is a compile-time
    // offset: with a run-time buffer index the compiler cannot tell the transposing reads from
    // the direct-to-LDS loads still in flight and drains the whole queue before every read.
    for (int64_t it0 = 0; it0 < nstage; it0 += kDwStages) {
        static_for<0, kDwStages>([&](auto sc) {
            constexpr int buf = decltype(sc)::value;
            const int64_t it = it0 + buf;
            if (it < nstage) {
                // everything but this wave's loads of the stages after this one has landed
                const int64_t left = nstage - 1 - it;
                if (wave < kFullWaves) {
                    if (left >= kAhead - 1) wait_vmcnt_n((kAhead - 1) * kMine);
                    else if (left == 1) wait_vmcnt_n(kMine);
                    else wait_vmcnt_n(0);
                } else {
                    if (left >= kAhead - 1) wait_vmcnt_n((kAhead - 1) * (kMine - 1));
                    else if (left == 1) wait_vmcnt_n(kMine - 1);
                    else wait_vmcnt_n(0);
                }
                asm volatile("s_barrier" ::: "memory");
                // the buffer read in the previous iteration is free again (every wave passed the barrier)
                constexpr int nbuf = (buf + kAhead) % kDwStages;
                if (it + kAhead < nstage) stage_in(it + kAhead, nbuf);
                if constexpr (ROW) {
                    if (wave < NX) {
#pragma unroll
                        for (int gt = 0; gt < G; ++gt) {
                            if (it * G + gt < ntile) {
                                const char* stage = lds + buf * kDwStageBytes + gt * kTileBytes;
#pragma unroll
                                for (int kp = 0; kp < 2; ++kp) {
                                    const half8 ya = tr_operand(stage + NX * kTileImgBytes, kp, lane, y_one);
                                    acc[0] = mfma(ya, tr_operand(stage + wave * kTileImgBytes, kp, lane, false), acc[0]);
                                    if constexpr (BIAS) bsum = point_sum(ya, bsum);
                                    if (wave == 0) track(ya);
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int gt = 0; gt < G; ++gt) {
                        if (it * G + gt < ntile) {
                            const char* stage = lds + buf * kDwStageBytes + gt * kTileBytes;
                            if (wave < jb.n_out) {
                                // 2 NX products, input operands read kDwAhead products ahead of
                                // their MFMA (the compiler, left alone, reads each operand right
                                // before its use and every MFMA then waits a full LDS round trip)
                                half8 ya[2], q[kDwAhead];
                                ya[0] = tr_operand(stage + (NX + wave) * kTileImgBytes, 0, lane, y_one);
                                ya[1] = tr_operand(stage + (NX + wave) * kTileImgBytes, 1, lane, y_one);
                                static_for<0, kDwAhead>([&](auto mc) {
                                    constexpr int m = decltype(mc)::value;
                                    q[m] = tr_operand(stage + (m % NX) * kTileImgBytes, m / NX, lane, false);
                                });
                                __builtin_amdgcn_sched_barrier(0);
                                static_for<0, 2 * NX>([&](auto mc) {
                                    constexpr int m = decltype(mc)::value, i = m % NX, kp = m / NX;
                                    const half8 b = q[m % kDwAhead];
                                    if constexpr (m + kDwAhead < 2 * NX) {
                                        constexpr int m2 = m + kDwAhead;
                                        q[m % kDwAhead] = tr_operand(stage + (m2 % NX) * kTileImgBytes, m2 / NX, lane, false);
                                    }
                                    acc[i] = mfma(ya[kp], b, acc[i]);
                                    __builtin_amdgcn_sched_barrier(0);
                                });
                                if constexpr (BIAS) bsum = point_sum(ya[1], point_sum(ya[0], bsum));
                                track(ya[0]), track(ya[1]);
                            }
                            if constexpr (NY1) {
                                if (wave < NX0) {  // the single-row gradient against this wave's tile of the first segment
                                    half8 yr[2], xr[2];
#pragma unroll
                                    for (int kp = 0; kp < 2; ++kp) {
                                        yr[kp] = tr_operand(stage + (NX + NY) * kTileImgBytes, kp, lane, true);
                                        xr[kp] = tr_operand(stage + wave * kTileImgBytes, kp, lane, false);
                                    }
                                    accr = mfma(yr[1], xr[1], mfma(yr[0], xr[0], accr));
                                    if constexpr (BIAS) bsum_r = point_sum(yr[1], point_sum(yr[0], bsum_r));
                                    if (wave == 0) track(yr[0]), track(yr[1]);
                                }
                            }
                        }
                    }
                }
            }
        });
    }
    if (ntile <= 0) return;  // the whole workgroup: its slice is empty and the sum skips it
    // store this workgroup's partial dW^T as its own tiles (256 contiguous bytes per register);
    // every tile of the job is written in full by exactly one wave
    float* const mine = a.slabs + ((size_t)(jb.slab0 + slice * jb.n_acc) * 16) * 64 + lane;
    auto flush = [&](int tile, const f32x16& v) {
        float* dst = mine + ((size_t)tile * 16) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = v[r];
    };
    // a bias gradient goes where the product with a ones column would have put it: column 0 of
    // the tile, output row r -> register (r & 3) + 4 (r >> 3) of lane 32 ((r >> 2) & 1); the sum of
    // row r is split over lanes r and r + 32; the rest of the tile is zero
    auto flush_bias = [&](int tile, float v) {
        f32x16 t = zero16;
        static_for<0, 16>([&](auto rc) {
            constexpr int reg = decltype(rc)::value;
            static_for<0, 2>([&](auto hc) {
                constexpr int hsel = decltype(hc)::value, r = (reg & 3) + 4 * hsel + 8 * (reg >> 2);
                const float sum = read_lane(v, r) + read_lane(v, r + 32);
                if (lane == 32 * hsel) t[reg] = sum;
            });
        });
        flush(tile, t);
    };
    if constexpr (ROW) {
        if (wave < NX) flush(wave, acc[0]);
        if (BIAS && wave == 0) flush_bias(NX, bsum);
    } else {
        if (wave < jb.n_out) {
#pragma unroll
            for (int i = 0; i < NX; ++i) flush(wave * NI + i, acc[i]);
            if constexpr (BIAS) flush_bias(wave * NI + NX, bsum);
        }
        if constexpr (NY1) {  // its row of tiles: the NX0 input tiles, then the bias
            if (wave < NX0) flush(jb.n_out * NI + wave, accr);
            if (BIAS && wave == 0) flush_bias(jb.n_out * NI + NX0, bsum_r);
        }
    }
    // the workgroup's largest gradient pattern -> one atomic per workgroup (LDS is free: every
    // wave is past its last stage once it has passed the first barrier)
    {
        unsigned m = ymax & 0xffffu, hi = ymax >> 16;
        m = m > hi ? m : hi;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, off);
            m = m > o ? m : o;
        }
        unsigned* slot = (unsigned*)lds;
        __syncthreads();
        if (lane == 0) slot[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned all = 0u;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) all = all > slot[w] ? all : slot[w];
            if (all) atomicMax(a.status + ST_YMAX, all);
        }
    }
}

enum DwKind { DW_HIDDEN, DW_LAYER0, DW_LAYER5, DW_HEADS, DW_RGB };

__global__ __launch_bounds__(512) void mlp_dw_kernel(DwArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[kDwStages * kDwStageBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // which job this workgroup belongs to: a select chain over constant indices keeps the job
    // in scalar registers (indexing the argument array at run time would move it to scratch)
    int ji = 0;
    DwJob jb = a.job[0];
#pragma unroll
    for (int j = 1; j < kNumJobs; ++j)
        if ((int)blockIdx.x >= a.job[j].wg0) jb = a.job[j], ji = j;
    (void)ji;
#ifdef DW_STAMPS
    struct Stamp {
        unsigned long long* p; unsigned long long t0; int j;
        __device__ ~Stamp() {
            if (threadIdx.x == 0) p[3 * blockIdx.x] = t0, p[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(), p[3 * blockIdx.x + 2] = j;
        }
    } stamp{a.stamps, __builtin_amdgcn_s_memrealtime(), ji};
#endif
    switch (jb.kind) {
        case DW_HIDDEN: dw_body<8, 0, 8, 0, 1, 1>(a, jb, lds, lane, wave); break;        // layers 1-4, 6, 7
        case DW_LAYER0: dw_body<2, 0, 8, 0, 1, 2>(a, jb, lds, lane, wave); break;        // point encoding -> 256
        case DW_LAYER5: dw_body<8, 2, 8, 0, 1, 1>(a, jb, lds, lane, wave); break;        // [h4 ; point encoding] -> 256
        case DW_HEADS: dw_body<8, 1, 4, 1, 1, 1>(a, jb, lds, lane, wave); break;         // [h7 ; view encoding] x d v_pre, h7 x d alpha
        default: dw_body<4, 0, 1, 0, 1, 3, true>(a, jb, lds, lane, wave); break;         // rgb
    }
}

// ---------------------------------------------------------------------------------------------
// 2b. sum of the workgroups' partial tiles, in workgroup order
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
    const float* slabs;
    float* gperm;            // [tiles][16 regs][64 lanes]
    int tile0[kNumJobs + 1]; // first summed tile of each job; [kNumJobs] = total
    int n_acc[kNumJobs], slab0[kNumJobs], used[kNumJobs];
};

// One workgroup per summed tile, one float4 per thread.
__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(ReduceArgs a) {
    const int tile = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < kNumJobs; ++k)
        if (tile >= a.tile0[k]) j = k;
    const int local = tile - a.tile0[j], n_acc = a.n_acc[j], used = a.used[j];
    const float4* src = (const float4*)(a.slabs + ((size_t)(a.slab0[j] + local) * 16) * 64) + threadIdx.x;
    const size_t stride = (size_t)n_acc * 16 * 64 / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < used; ++w) {
        const float4 v = src[(size_t)w * stride];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    ((float4*)(a.gperm + (size_t)tile * 16 * 64))[threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// 3. unpack
// ---------------------------------------------------------------------------------------------
struct GradPtrs {
    float* g[24];
};

struct UnpackArgs {
    const float* gperm;
    GradPtrs out;
    float gscale;      // > 0: the caller's factor; otherwise the one formed from status[ST_ABSMAX]
    int target_exp;
    unsigned* status;
    int accumulate;
    int tile0[kNumProducts];  // per product (0..7 point layers, 8 bottleneck: none, 9 view, 10 rgb, 11 alpha):
    int n_in[kNumProducts];   // first accumulator tile, and input tiles per output tile (bias tile last)
    int first[25];  // first flat index of each parameter array (get_weights order); [24] = total
    float* pext;    // [257][128]: P = sum_points h7 (x) d v_pre in natural order, then s = sum_points d v_pre
};
constexpr int kPextRows = 257, kPextCols = 128;

// (inputs, outputs) of the 12 layers in get_weights() order
__device__ const int kLayerDims[12][2] = {{63, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {319, 256},
                                          {256, 256}, {256, 256}, {256, 256}, {283, 128}, {128, 3}, {256, 1}};

// Slot (tile, row) of OUTPUT feature f of a 256- or 128-wide layer in the chain kernel's gradient
// fragments (layout::acc_feature).
__device__ __forceinline__ void acc_slot(int f, int& tile, int& row) {
    const int s = 2 * (f >> 5) + ((f >> 4) & 1), f16 = f & 15;
    const int h = (f16 >> 2) & 1, j = ((f16 >> 3) << 2) | (f16 & 3);
    tile = s >> 1, row = 16 * (s & 1) + 8 * h + j;
}

// Slot (tile, row) of INPUT feature f of a saved 256- or 128-wide activation (the forward
// kernel's fragments, lw::acc_feature: element j of lane group g of k-step s is feature
// 32 s + 16 (j >> 2) + 4 g + (j & 3)); row = 8 g + j.
__device__ __forceinline__ void in_slot(int f, int& tile, int& row) {
    const int f32 = f & 31, g = (f32 >> 2) & 3, j = 4 * (f32 >> 4) + (f32 & 3);
    tile = f >> 5, row = 8 * g + j;
}

// Slot of feature f of the saved point encoding (lw::pe_feature: 16 local slots per lane group,
// two k-steps) / of the saved view encoding (lw::view_feature: 8 local slots, one k-step).
__device__ __forceinline__ void pe_slot(int f, int& tile, int& row) {
    tile = 0, row = 0;
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 16; ++i)
            if (lw::pe_feature(g, i) == f) tile = i >> 3, row = 8 * g + (i & 7);
}
__device__ __forceinline__ void view_slot(int f, int& tile, int& row) {
    tile = 0, row = 0;
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 8; ++j)
            if (lw::view_feature(g, j) == f) row = 8 * g + j;
}

// One thread per parameter scalar of the whole network.  pi = parameter array (get_weights
// order), e = flat index inside it.
__global__ void mlp_grad_unpack_kernel(UnpackArgs a) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned absmax = a.status[ST_ABSMAX], ymax = a.status[ST_YMAX];
    const float gs = a.gscale > 0.f ? a.gscale : auto_gscale(absmax, a.target_exp);
    // saturated (the chain clamps at the largest finite fp16, 0x7bff), infinite or NaN; or, with a
    // non-zero d raw, not one stored gradient as large as the smallest normal fp16 (0x0400)
    unsigned flags = 0u;
    if (ymax >= 0x7bffu || absmax >= 0x7f800000u) flags |= NERF_GRAD_SATURATED;
    if (absmax != 0u && ymax < 0x0400u) flags |= NERF_GRAD_UNDERFLOW;
    if (gi == 0) {
        a.status[ST_FLAGS] = flags;
        a.status[ST_GSCALE] = __builtin_bit_cast(unsigned, gs);
        a.status[ST_ABSMAX_OUT] = absmax;
        a.status[ST_YMAX_OUT] = ymax;
    }
    const float inv = 1.0f / gs;
    if (gi >= a.first[24]) {
        // the operand of mlp_head_compose_kernel: the view layer's product with h7 (and its bias sums)
        const int q = gi - a.first[24];
        if (q >= kPextRows * kPextCols) return;
        const int i = q / kPextCols, j = q % kPextCols;
        int ot, orow, it, irow;
        acc_slot(j, ot, orow);
        if (i < 256) in_slot(i, it, irow); else it = a.n_in[9] - 1, irow = 0;
        const int reg = (orow & 3) + 4 * (orow >> 3), lane = 32 * ((orow >> 2) & 1) + irow;
        float v = a.gperm[((size_t)(a.tile0[9] + ot * a.n_in[9] + it) * 16 + reg) * 64 + lane] * inv;
        if (flags) v = __builtin_nanf("");
        a.pext[q] = v;
        return;
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 24; ++i)
        if (gi >= a.first[i]) pi = i;
    const int e = gi - a.first[pi];
    const int out_dim = kLayerDims[pi >> 1][1];
    const bool is_bias = pi & 1;
    const int layer = pi >> 1;  // 0..7 point layers, 8 bottleneck, 9 view, 10 rgb, 11 alpha
    const int in = is_bias ? -1 : e / out_dim, out = is_bias ? e : e % out_dim;
    // the bottleneck's arrays and the view layer's bottleneck rows are written by mlp_head_compose_kernel
    if (layer == 8 || (layer == 9 && !is_bias && in < 256)) return;
    int ot, orow, it, irow;
    if (layer == 10 || layer == 11) ot = 0, orow = out;           // d rgb slots 0..2 / d alpha slot 0
    else acc_slot(out, ot, orow);
    if (is_bias) it = a.n_in[layer] - 1, irow = 0;
    else if (layer == 0) pe_slot(in, it, irow);
    else if (layer == 5) {       // input tiles: h4 (8), then the point encoding (2)
        if (in < 63) { pe_slot(in, it, irow); it += 8; } else in_slot(in - 63, it, irow);
    } else if (layer == 9) {     // input tiles: h7 (8, composed elsewhere), then the view encoding (1)
        view_slot(in - 256, it, irow);
        it += 8;
    } else in_slot(in, it, irow);
    const int job = layer;
    const int n_in = a.n_in[job];
    const int reg = (orow & 3) + 4 * (orow >> 3), lane = 32 * ((orow >> 2) & 1) + irow;
    float v = a.gperm[((size_t)(a.tile0[job] + ot * n_in + it) * 16 + reg) * 64 + lane] * inv;
    if (flags) v = __builtin_nanf("");
    float* dst = a.out.g[pi] + e;
    *dst = a.accumulate ? *dst + v : v;
}

// ---------------------------------------------------------------------------------------------
// 4. the two layers around the linear bottleneck
// ---------------------------------------------------------------------------------------------
// bott = h7 . W_b + b_b (run_nerf_helpers.py:103) feeds v_pre = [bott, views] . W_v + b_v (:104-110).
// With P[i][j] = sum over points of h7[i] d v_pre[j] and s[j] = sum of d v_pre[j] (pext rows 0..255 and
// row 256), and d bott[k] = sum_j W_v[k][j] d v_pre[j]:
//     dW_b[i][k] = sum_points h7[i] d bott[k]  = sum_j P[i][j] W_v[k][j]          (C1 rows 0..255)
//     db_b[k]    = sum_points d bott[k]        = sum_j s[j] W_v[k][j]             (C1 row 256)
//     dW_v[k][j] = sum_points bott[k] d v_pre[j] = sum_i W_b[i][k] P[i][j] + b_b[k] s[j]   (C2), k < 256
// in fp32 from the fp32 parameters: two products of 257 x 128 x 256 multiply-adds per network and
// step, one 16 x 16 output tile per workgroup, sums in a fixed order.
struct ComposeArgs {
    const float* pext;   // [257][128]
    const float* w_b;    // bottleneck kernel [256 in][256 out]
    const float* b_b;    // [256]
    const float* w_v;    // view layer kernel, rows 0..255: [256][128]
    float* dw_b;         // [256][256]
    float* db_b;         // [256]
    float* dw_v;         // [283][128]: rows 0..255 are written
    int accumulate;
};
constexpr int kComposeC1 = ((kPextRows + 15) / 16) * (256 / 16), kComposeC2 = (256 / 16) * (128 / 16);

__global__ __launch_bounds__(256) void mlp_head_compose_kernel(ComposeArgs a) {
    __shared__ float As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float sum = 0.f;
    if ((int)blockIdx.x < kComposeC1) {
        const int r0 = ((int)blockIdx.x / 16) * 16, k0 = ((int)blockIdx.x % 16) * 16;
#pragma unroll 1
        for (int j0 = 0; j0 < kPextCols; j0 += 16) {
            As[ty][tx] = r0 + ty < kPextRows ? a.pext[(r0 + ty) * kPextCols + j0 + tx] : 0.f;
            Bs[ty][tx] = a.w_v[(k0 + ty) * 128 + j0 + tx];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += As[ty][j] * Bs[tx][j];
            __syncthreads();
        }
        const int r = r0 + ty, k = k0 + tx;
        if (r > 256) return;
        float* dst = r < 256 ? a.dw_b + r * 256 + k : a.db_b + k;
        *dst = a.accumulate ? *dst + sum : sum;
    } else {
        const int b = (int)blockIdx.x - kComposeC1;
        const int k0 = (b / 8) * 16, j0 = (b % 8) * 16;
#pragma unroll 1
        for (int i0 = 0; i0 < 256; i0 += 16) {
            As[ty][tx] = a.w_b[(i0 + ty) * 256 + k0 + tx];
            Bs[ty][tx] = a.pext[(i0 + ty) * kPextCols + j0 + tx];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += As[i][ty] * Bs[i][tx];
            __syncthreads();
        }
        const int k = k0 + ty, j = j0 + tx;
        sum += a.b_b[k] * a.pext[256 * kPextCols + j];
        float* dst = a.dw_v + k * 128 + j;
        *dst = a.accumulate ? *dst + sum : sum;
    }
}

// The jobs of one network, their accumulator tiles in the partial-sum buffer and their share of
// `total_wgs` workgroups (in proportion to the 1 KiB loads a job issues per wave tile).  prod_tile0 / prod_n_in tell
// the unpack kernel where each layer's product lives.
static void make_jobs(DwJob* jb, int* total_tiles, int total_wgs, int* prod_tile0 = nullptr, int* prod_n_in = nullptr,
                      int* slab_tiles = nullptr) {
    int blocks[kNumJobs], n_acc[kNumJobs], layer_of[kNumJobs];
    int nj = 0;
    auto add = [&](int layer, int kind, int x0, int x1, int yb, int y1b, int yk, int n_out, int n_in, int row_tiles, int nloads) {
        jb[nj] = DwJob{kind, x0, x1, yb, y1b, yk, n_out, 0, 0, 0, 0, 0, 0};
        blocks[nj] = nloads, n_acc[nj] = n_in * n_out + row_tiles, layer_of[nj] = layer;
        if (prod_n_in) prod_n_in[layer] = n_in;
        ++nj;
    };
    add(0, DW_LAYER0, A_PE, 0, G_DPRE0, 0, 16, 8, 3, 0, 4 + 16);
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) add(5, DW_LAYER5, A_H0 + 16 * 4, A_PE, G_DPRE0 + 16 * 5, 0, 16, 8, 11, 0, 16 + 4 + 16);
        else add(l, DW_HIDDEN, A_H0 + 16 * (l - 1), 0, G_DPRE0 + 16 * l, 0, 16, 8, 9, 0, 32);
    }
    // view layer against [h7 ; view encoding] (the h7 part is P of mlp_head_compose_kernel) + alpha against h7
    add(9, DW_HEADS, A_H0 + 16 * 7, A_VPE, G_DV, G_DA, 8, 4, 10, 9, 16 + 2 + 8 + 2 + 2);
    add(10, DW_RGB, A_V, 0, G_DC, 0, 1, 1, 5, 0, 8 + 2 + 2);                             // (+2: short stages, unpipelined)
    int tiles = 0, bytes = 0;
    for (int j = 0; j < nj; ++j) {
        jb[j].tile0 = tiles;
        jb[j].n_acc = n_acc[j];
        if (prod_tile0) prod_tile0[layer_of[j]] = tiles;
        tiles += n_acc[j];
        bytes += blocks[j];
    }
    if (prod_tile0) {  // alpha rides on the heads job: its row of tiles follows the 4 x 10 main tiles
        prod_tile0[11] = prod_tile0[9] + 4 * 10;
        prod_n_in[11] = 9;
        prod_tile0[8] = -1, prod_n_in[8] = 0;  // the bottleneck's gradient is composed, not accumulated
    }
    *total_tiles = tiles;
    // largest-remainder split of exactly total_wgs workgroups (one more than the CU count would
    // run alone after all the others)
    int given = 0;
    double frac[kNumJobs];
    for (int j = 0; j < nj; ++j) {
        const double share = (double)blocks[j] / bytes * total_wgs;
        int n = (int)share;
        if (n < 1) n = 1;
        jb[j].wgs = n, frac[j] = share - n, given += n;
    }
    while (given < total_wgs) {
        int best = 0;
        for (int j = 1; j < nj; ++j)
            if (frac[j] > frac[best]) best = j;
        jb[best].wgs += 1, frac[best] -= 1.0, given += 1;
    }
    int wg = 0, slab = 0;
    for (int j = 0; j < nj; ++j) {
        jb[j].wg0 = wg;
        jb[j].slab0 = slab;
        wg += jb[j].wgs;
        slab += jb[j].wgs * jb[j].n_acc;
    }
    if (slab_tiles) *slab_tiles = slab;
}

// Workgroups of the weight-gradient kernel: one per CU, fewer when there are few wave tiles.
static int dw_workgroups(int64_t wave_tiles, int cus) {
    if (wave_tiles >= (int64_t)cus * 4) return cus;
    const int64_t n = wave_tiles / 4;
    return (int)(n > kNumJobs ? n : kNumJobs);
}

constexpr size_t kTileFloats = 16 * 64;

// Layout of `partial`: status block, the operand of mlp_head_compose_kernel, summed tiles, then the
// per-workgroup partial tiles.
constexpr size_t kPextBytes = ((size_t)kPextRows * kPextCols * sizeof(float) + 255) / 256 * 256;
static size_t partial_bytes_for(int wgs) {
    DwJob jb[kNumJobs];
    int tiles, slab;
    make_jobs(jb, &tiles, wgs, nullptr, nullptr, &slab);
    return kStatusBytes + kPextBytes + ((size_t)tiles + slab) * kTileFloats * sizeof(float);
}

}  // namespace nerf

using namespace nerf;

#ifdef DW_STAMPS
static unsigned long long* g_dw_stamps = nullptr;
NERF_API void nerf_debug_set_stamps(void* p) { g_dw_stamps = (unsigned long long*)p; }
#endif

NERF_API size_t nerf_mlp_packed_bwd_bytes(void) { return layout::kPackedBwdBytes; }

NERF_API int nerf_mlp_train_layout(int32_t* out, int n) {
    NERF_REQUIRE(out && n >= 10, "nerf_mlp_train_layout: need room for 10 numbers");
    const int32_t v[10] = {layout::kActBlocks, layout::A_PE, layout::A_H0, layout::A_V, layout::A_VPE, layout::A_MASK,
                           layout::kGradBlocks, layout::G_DV, layout::G_DPRE0, layout::G_DA};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    return 0;
}

NERF_API int nerf_mlp_pack_bwd(const float* const* params, void* packed_bwd, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed_bwd, "nerf_mlp_pack_bwd: null pointer");
    NERF_REQUIRE(((uintptr_t)packed_bwd & 15) == 0, "nerf_mlp_pack_bwd: packed_bwd must be 16-byte aligned");
    ParamPtrsB pp;
    for (int i = 0; i < 24; ++i) {
        NERF_REQUIRE(params[i], "nerf_mlp_pack_bwd: params[%d] is null", i);
        pp.p[i] = params[i];
    }
    hipStream_t st = as_stream(stream);
    for (int i = 16; i <= 18; ++i)
        NERF_REQUIRE(((uintptr_t)params[i] & 15) == 0, "nerf_mlp_pack_bwd: params[%d] must be 16-byte aligned", i);
    // the stream, then fp32 copies of what mlp_head_compose_kernel multiplies by (bottleneck kernel
    // and bias, the first 256 of the view layer's [283][128] rows)
    mlp_pack_bwd_kernel<<<layout::B_END + kBwdCopyBlocks, 64, 0, st>>>(pp, (char*)packed_bwd);
    return check_launch("nerf_mlp_pack_bwd");
}

NERF_API size_t nerf_mlp_grad_bytes(int64_t n_points) {
    const int64_t wave_tiles = (n_points + kPointsPerBlock - 1) / kPointsPerBlock * kWavesPerBlock;
    return (size_t)wave_tiles * layout::kGradBlocks * layout::kFragBytes;
}

NERF_API size_t nerf_mlp_partial_bytes(void) {
    // sized for one workgroup per CU of the current device (256 when there is none to ask)
    int cus = device_cu_count();
    if (cus < 256) cus = 256;
    return partial_bytes_for(cus);
}

NERF_API int nerf_mlp_backward(const void* packed_bwd, const void* act, const float* d_raw,
                               int64_t n_points, float gscale, void* gbuf, void* partial,
                               float* const* grads, int accumulate, nerf_stream_t stream) {
    NERF_REQUIRE(packed_bwd && act && d_raw && gbuf && partial && grads, "nerf_mlp_backward: null pointer");
    NERF_REQUIRE(n_points >= 0, "nerf_mlp_backward: bad size");
    NERF_REQUIRE(gscale == gscale && gscale < 3.0e38f && gscale > -3.0e38f, "nerf_mlp_backward: gscale must be finite (<= 0 selects the automatic scale)");
    // gscale < 0: automatic, with the largest scaled element in [-gscale, -2 gscale)
    int target_exp = 8;
    if (gscale < 0.f) {
        int e2 = 0;
        const float mant = frexpf(-gscale, &e2);  // -gscale = mant * 2^e2, mant in [0.5, 1)
        NERF_REQUIRE(mant == 0.5f && e2 - 1 >= -14 && e2 - 1 <= 14,
                     "nerf_mlp_backward: a negative gscale names the target of the automatic scale, a power of two in [2^-14, 2^14] (got %g)", (double)-gscale);
        target_exp = e2 - 1;
    }
    NERF_REQUIRE((((uintptr_t)packed_bwd | (uintptr_t)act | (uintptr_t)d_raw | (uintptr_t)gbuf | (uintptr_t)partial) & 15) == 0,
                 "nerf_mlp_backward: buffers must be 16-byte aligned");
    UnpackArgs ua{};
    for (int i = 0; i < 24; ++i) {
        NERF_REQUIRE(grads[i], "nerf_mlp_backward: grads[%d] is null", i);
        ua.out.g[i] = grads[i];
    }
    if (n_points == 0) return 0;
    hipStream_t st = as_stream(stream);
    const int64_t blocks = (n_points + kPointsPerBlock - 1) / kPointsPerBlock;
    if (blocks > 0x7fffffffLL) return NERF_FAIL(NERF_E_UNSUPPORTED, "nerf_mlp_backward: too many points");
    int cus = device_cu_count();
    if (cus <= 0) return NERF_FAIL(NERF_E_UNSUPPORTED, "nerf_mlp_backward: no current device");

    unsigned* status = (unsigned*)partial;
    hipError_t e = hipMemsetAsync(status, 0, kStatusBytes, st);
    if (e != hipSuccess) return NERF_FAIL((int)e, "nerf_mlp_backward(memset): %s", hipGetErrorString(e));
    {
        const int64_t want = (n_points + 255) / 256;
        draw_absmax_kernel<<<(unsigned)(want < 512 ? want : 512), 256, 0, st>>>((const float4*)d_raw, n_points, status);
        int rc = check_launch("nerf_mlp_backward(absmax)");
        if (rc) return rc;
    }
    ChainArgs ca{(const char*)packed_bwd, (const char*)act, (const float4*)d_raw, (char*)gbuf, n_points,
                 gscale > 0.f ? gscale : 0.f, target_exp, status};
    mlp_bwd_chain_kernel<<<(unsigned)blocks, 512, 0, st>>>(ca);
    int rc = check_launch("nerf_mlp_backward(chain)");
    if (rc) return rc;

    DwArgs da{};
    da.act = (const char*)act, da.gbuf = (const char*)gbuf, da.status = status;
    da.wave_tiles = blocks * kWavesPerBlock;
    int tiles, slab_tiles;
    const int wgs = dw_workgroups(da.wave_tiles, cus);
    make_jobs(da.job, &tiles, wgs, ua.tile0, ua.n_in, &slab_tiles);
    float* pext = (float*)((char*)partial + kStatusBytes);
    float* gperm = (float*)((char*)partial + kStatusBytes + kPextBytes);
    da.slabs = gperm + (size_t)tiles * kTileFloats;
    ReduceArgs ra{};
    ra.slabs = da.slabs, ra.gperm = gperm;
    for (int j = 0; j < kNumJobs; ++j) {
        DwJob& jb = da.job[j];
        const int64_t per = (da.wave_tiles + jb.wgs - 1) / jb.wgs;
        jb.used = (int)((da.wave_tiles + per - 1) / per);
        ra.tile0[j] = jb.tile0, ra.n_acc[j] = jb.n_acc, ra.slab0[j] = jb.slab0, ra.used[j] = jb.used;
    }
    ra.tile0[kNumJobs] = tiles;
    const int total_wgs = da.job[kNumJobs - 1].wg0 + da.job[kNumJobs - 1].wgs;
#ifdef DW_STAMPS
    da.stamps = g_dw_stamps;
#endif
    mlp_dw_kernel<<<total_wgs, 512, 0, st>>>(da);
    rc = check_launch("nerf_mlp_backward(dw)");
    if (rc) return rc;
    mlp_dw_reduce_kernel<<<tiles, 256, 0, st>>>(ra);
    rc = check_launch("nerf_mlp_backward(reduce)");
    if (rc) return rc;

    ua.gperm = gperm, ua.gscale = gscale > 0.f ? gscale : 0.f, ua.target_exp = target_exp, ua.status = status, ua.accumulate = accumulate;
    ua.pext = pext;
    static const int dims[12][2] = {{63, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {319, 256},
                                    {256, 256}, {256, 256}, {256, 256}, {283, 128}, {128, 3}, {256, 1}};
    ua.first[0] = 0;
    for (int i = 0; i < 24; ++i) {
        const int in_dim = dims[i >> 1][0], out_dim = dims[i >> 1][1];
        ua.first[i + 1] = ua.first[i] + ((i & 1) ? out_dim : in_dim * out_dim);
    }
    mlp_grad_unpack_kernel<<<(ua.first[24] + kPextRows * kPextCols + 255) / 256, 256, 0, st>>>(ua);
    rc = check_launch("nerf_mlp_backward(unpack)");
    if (rc) return rc;
    const char* pb = (const char*)packed_bwd;
    ComposeArgs co{pext, (const float*)(pb + layout::kBwdBottW), (const float*)(pb + layout::kBwdBottB),
                   (const float*)(pb + layout::kBwdViewW), grads[16], grads[17], grads[18], accumulate};
    mlp_head_compose_kernel<<<kComposeC1 + kComposeC2, 256, 0, st>>>(co);
    return check_launch("nerf_mlp_backward(compose)");
}

NERF_API int nerf_mlp_backward_status(const void* partial, int* flags, float* gscale_used, float* absmax_d_raw,
                                      nerf_stream_t stream) {
    NERF_REQUIRE(partial && flags, "nerf_mlp_backward_status: null pointer");
    unsigned host[8] = {};
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemcpyAsync(host, partial, sizeof host, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return NERF_FAIL((int)e, "nerf_mlp_backward_status: %s", hipGetErrorString(e));
    *flags = (int)host[ST_FLAGS];
    if (gscale_used) memcpy(gscale_used, &host[ST_GSCALE], sizeof(float));
    if (absmax_d_raw) memcpy(absmax_d_raw, &host[ST_ABSMAX_OUT], sizeof(float));
    return 0;
}
