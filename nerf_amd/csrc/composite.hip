// Alpha compositing along each ray (raw2outputs) as wavefront-scan kernels.  Rays of more than 64
// samples: one 64-lane wave per ray, one sample per lane per 64-sample chunk, exclusive
// transmittance product by a DPP scan carried across chunks.  Shorter rays: half a wave per ray,
// two samples per lane (composite_fwd_pair_kernel).  HBM-bound: raw (16 B/sample) and z (4 B/sample) are read once
// as coalesced 1 KiB / 256 B wave transactions, weights written once; raw and weights are
// streamed (read-once / written-once data does not displace the depths, which are read twice).
#include "common.h"

namespace nerf {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

typedef float comp_f32x4 __attribute__((ext_vector_type(4)));

// What a lane reads for one sample: raw, depth, the next depth and the density noise.
struct SampleIn {
    comp_f32x4 r;
    float zz, znext, nz;
};

__device__ __forceinline__ SampleIn load_sample(const float4* __restrict__ raw, const float* __restrict__ z,
                                                const float* __restrict__ noise, int64_t row, int s, int S) {
    SampleIn in;
    const bool valid = s < S;
    in.r = comp_f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid) in.r = __builtin_nontemporal_load((const comp_f32x4*)(raw + row + s));
    in.zz = valid ? z[row + s] : 0.f;
    in.znext = (s + 1 < S) ? z[row + s + 1] : 0.f;
    in.nz = (noise && valid) ? noise[row + s] : 0.f;
    return in;
}

// NCH = 64-sample chunks per ray, R = rays a wave has in flight: every load of the R rays is
// issued before the first scan (R x NCH x 1.3 KiB per wave, about 4 KiB: with less in flight the
// kernel is bound by load latency, 4.0 instead of 6+ TB/s).  NCH = 0: any S, chunk by chunk.
template <int NCH, int R>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
    int d_stride, int64_t n, int S, int white_bkgd, const float* __restrict__ noise,
    float noise_std, float* __restrict__ rgb_map, float* __restrict__ disp_map,
    float* __restrict__ acc_map, float* __restrict__ depth_map, float* __restrict__ weights) {
    static_assert(NCH > 0 || R == 1, "the chunk-by-chunk path takes one ray at a time");
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ray0 = wave0 * R; ray0 < n; ray0 += nwaves * R) {
        SampleIn pre[R][NCH > 0 ? NCH : 1];
        float dir[R][3];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t ray = ray0 + q < n ? ray0 + q : n - 1;  // past the end: loaded, not used
            if constexpr (NCH > 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) pre[q][c] = load_sample(raw, z, noise, ray * S, 64 * c + lane, S);
            }
            dir[q][0] = rays_d[ray * d_stride], dir[q][1] = rays_d[ray * d_stride + 1], dir[q][2] = rays_d[ray * d_stride + 2];
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t ray = ray0 + q;
            if (ray >= n) break;  // wave-uniform
            const int64_t row = ray * S;
            const float dx = dir[q][0], dy = dir[q][1], dz = dir[q][2];
            const float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
            float carry = 1.0f;
            float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
            auto chunk = [&](int base, const SampleIn& in) {
                const int s = base + lane;
                const bool valid = s < S;
                const float zz = in.zz;
                float dist = (s + 1 < S) ? in.znext - zz : 1e10f;
                dist = dist * norm;
                float sig = in.r[3];
                if (noise) sig = sig + in.nz * noise_std;
                const float alpha = valid ? 1.0f - __expf(-fmaxf(sig, 0.f) * dist) : 0.f;
                const float t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
                const float incl = wave_scan_mul(t);
                const float trans = carry * wave_shr1(1.0f, incl);
                const float w = alpha * trans;
                carry = carry * lane63(incl);
                if (valid) {
                    if (weights) __builtin_nontemporal_store(w, weights + row + s);
                    sr += w * sigmoidf(in.r[0]);
                    sg += w * sigmoidf(in.r[1]);
                    sb += w * sigmoidf(in.r[2]);
                    sd += w * zz;
                    sa += w;
                }
            };
            if constexpr (NCH > 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) chunk(64 * c, pre[q][c]);
            } else {
                for (int base = 0; base < S; base += 64) chunk(base, load_sample(raw, z, noise, row, base + lane, S));
            }
            sr = wave_sum(sr), sg = wave_sum(sg), sb = wave_sum(sb), sd = wave_sum(sd),
            sa = wave_sum(sa);
            if (lane == 0) {
                // disp = 1/max(1e-10, depth/acc); 0/0 stays NaN as in the reference's maximum().
                const float qd = sd / sa;
                const float disp = 1.0f / ((qd != qd) ? qd : fmaxf(1e-10f, qd));
                if (white_bkgd) {
                    const float bg = 1.0f - sa;
                    sr += bg, sg += bg, sb += bg;
                }
                rgb_map[ray * 3] = sr, rgb_map[ray * 3 + 1] = sg, rgb_map[ray * 3 + 2] = sb;
                if (disp_map) disp_map[ray] = disp;
                if (acc_map) acc_map[ray] = sa;
                if (depth_map) depth_map[ray] = sd;
            }
        }
    }
}

// Rays of at most 64 samples: half a wave (32 lanes) per ray, two consecutive samples per lane, so
// that one instruction stream composites two rays (the scans and the five map sums, which are
// what bounds the one-ray-per-wave form at this length, cost 5 DPP steps per PAIR of rays instead
// of 6 per ray, and no lane read).  P pairs of rays per wave are loaded before the first scan.
// Segmented scans: the four row shifts plus the row broadcast into rows 1 and 3 stay inside each
// 32-lane half.
__device__ __forceinline__ float half_scan_mul(float v) {
    v *= dpp_or<0x111, 0xf>(1.0f, v);
    v *= dpp_or<0x112, 0xf>(1.0f, v);
    v *= dpp_or<0x114, 0xf>(1.0f, v);
    v *= dpp_or<0x118, 0xf>(1.0f, v);
    v *= dpp_or<0x142, 0xa>(1.0f, v);  // row_bcast:15 -> rows 1 and 3
    return v;
}
__device__ __forceinline__ float half_scan_add(float v) {
    v += dpp_or<0x111, 0xf>(0.0f, v);
    v += dpp_or<0x112, 0xf>(0.0f, v);
    v += dpp_or<0x114, 0xf>(0.0f, v);
    v += dpp_or<0x118, 0xf>(0.0f, v);
    v += dpp_or<0x142, 0xa>(0.0f, v);
    return v;  // lanes 31 and 63 hold the sums of their halves
}

template <int P>
__global__ __launch_bounds__(256) void composite_fwd_pair_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
    int d_stride, int64_t n, int S, int white_bkgd, const float* __restrict__ noise,
    float noise_std, float* __restrict__ rgb_map, float* __restrict__ disp_map,
    float* __restrict__ acc_map, float* __restrict__ depth_map, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63, h = lane & 31, half = lane >> 5;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int s0 = 2 * h, s1 = 2 * h + 1;
    // every row starts on an 8-byte boundary: a lane's two weights can leave as one store
    const bool pair_stores = (S & 1) == 0 && ((uintptr_t)weights & 7) == 0;
    const bool pair_loads = (S & 1) == 0 && ((uintptr_t)z & 7) == 0;
    for (int64_t ray0 = wave0 * (2 * P); ray0 < n; ray0 += nwaves * (2 * P)) {
        SampleIn in0[P], in1[P];
        float dir[P][3];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            int64_t ray = ray0 + 2 * q + half;
            ray = ray < n ? ray : n - 1;  // past the end: loaded, not used
            const int64_t row = ray * S;
            if (pair_loads) {  // the lane's two depths as one 8-byte load; the depth after them comes from the next lane
                typedef float comp_f32x2 __attribute__((ext_vector_type(2)));
                const bool valid = s0 < S;  // S is even: s1 is valid whenever s0 is
                in0[q].r = in1[q].r = comp_f32x4{0.f, 0.f, 0.f, 0.f};
                comp_f32x2 zz = {0.f, 0.f};
                if (valid) {
                    in0[q].r = __builtin_nontemporal_load((const comp_f32x4*)(raw + row + s0));
                    in1[q].r = __builtin_nontemporal_load((const comp_f32x4*)(raw + row + s1));
                    zz = *(const comp_f32x2*)(z + row + s0);
                }
                in0[q].zz = zz[0], in1[q].zz = zz[1], in0[q].znext = zz[1];
                in0[q].nz = (noise && valid) ? noise[row + s0] : 0.f;
                in1[q].nz = (noise && valid) ? noise[row + s1] : 0.f;
            } else {
                in0[q] = load_sample(raw, z, noise, row, s0, S);
                in1[q] = load_sample(raw, z, noise, row, s1, S);
            }
            dir[q][0] = rays_d[ray * d_stride], dir[q][1] = rays_d[ray * d_stride + 1], dir[q][2] = rays_d[ray * d_stride + 2];
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int64_t ray = ray0 + 2 * q + half;
            const bool live = ray < n;
            const int64_t row = (live ? ray : n - 1) * S;
            const float dx = dir[q][0], dy = dir[q][1], dz = dir[q][2];
            const float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
            auto alpha_of = [&](const SampleIn& in, int s, float& t) {
                const bool valid = s < S;
                float dist = (s + 1 < S) ? in.znext - in.zz : 1e10f;
                dist = dist * norm;
                float sig = in.r[3];
                if (noise) sig = sig + in.nz * noise_std;
                const float alpha = valid ? 1.0f - __expf(-fmaxf(sig, 0.f) * dist) : 0.f;
                t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
                return alpha;
            };
            if (pair_loads) in1[q].znext = wave_shl1(0.f, in0[q].zz);  // every lane active (used only where s1 + 1 < S)
            float t0, t1;
            const float a0 = alpha_of(in0[q], s0, t0), a1 = alpha_of(in1[q], s1, t1);
            const float incl = half_scan_mul(t0 * t1);
            // product over the lanes before this one (the shift runs with every lane active: a lane
            // that is switched off cannot be read by its neighbour)
            float before = wave_shr1(1.0f, incl);
            before = (h == 0) ? 1.0f : before;
            const float w0 = a0 * before, w1 = a1 * (before * t0);
            float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
            if (s0 < S) {
                sr = w0 * sigmoidf(in0[q].r[0]), sg = w0 * sigmoidf(in0[q].r[1]), sb = w0 * sigmoidf(in0[q].r[2]);
                sd = w0 * in0[q].zz, sa = w0;
            }
            if (s1 < S) {
                sr += w1 * sigmoidf(in1[q].r[0]), sg += w1 * sigmoidf(in1[q].r[1]), sb += w1 * sigmoidf(in1[q].r[2]);
                sd += w1 * in1[q].zz, sa += w1;
            }
            if (weights && live) {
                if (pair_stores && s1 < S) {  // both weights of the lane as one 8-byte store
                    typedef float comp_f32x2 __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(comp_f32x2{w0, w1}, (comp_f32x2*)(weights + row + s0));
                } else {
                    if (s0 < S) __builtin_nontemporal_store(w0, weights + row + s0);
                    if (s1 < S) __builtin_nontemporal_store(w1, weights + row + s1);
                }
            }
            sr = half_scan_add(sr), sg = half_scan_add(sg), sb = half_scan_add(sb), sd = half_scan_add(sd),
            sa = half_scan_add(sa);
            if (h == 31 && live) {
                const float qd = sd / sa;
                const float disp = 1.0f / ((qd != qd) ? qd : fmaxf(1e-10f, qd));
                if (white_bkgd) {
                    const float bg = 1.0f - sa;
                    sr += bg, sg += bg, sb += bg;
                }
                rgb_map[ray * 3] = sr, rgb_map[ray * 3 + 1] = sg, rgb_map[ray * 3 + 2] = sb;
                if (disp_map) disp_map[ray] = disp;
                if (acc_map) acc_map[ray] = sa;
                if (depth_map) depth_map[ray] = sd;
            }
        }
    }
}

// Backward of the compositing: d raw [n,S,4] from the gradients of rgb_map / acc_map / depth_map.
// With c = sigmoid(raw_rgb), e = exp(-sigma*dist) (= 1 - alpha), t = e + 1e-10, T = exclusive
// product of t, w = alpha*T and G_i = dL/dw_i = g_rgb.c_i + g_depth*z_i + g_acc (white background:
// g_acc - sum(g_rgb)):
//   dL/dalpha_i = G_i*T_i - (sum_{j>i} G_j*w_j)/t_i     (the division is TF's cumprod gradient)
//   d raw_sigma = [sigma_pre > 0] * dist*e * dL/dalpha,    d raw_rgb = g_rgb * w * c*(1-c).
// Two sweeps per ray: ascending 64-sample chunks to get each chunk's incoming transmittance,
// then descending chunks with a suffix-sum scan carried from the far end.
constexpr int kMaxChunks = 16;

__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
    int d_stride, int64_t n, int S, int white_bkgd, const float* __restrict__ noise,
    float noise_std, const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
    const float* __restrict__ g_depth, float4* __restrict__ d_raw) {
    __shared__ float carry_in_all[4][kMaxChunks];
    const int lane = threadIdx.x & 63;
    float* carry_in = carry_in_all[threadIdx.x >> 6];
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nch = (S + 63) >> 6;
    for (int64_t ray = wave0; ray < n; ray += nwaves) {
        const float dx = rays_d[ray * d_stride], dy = rays_d[ray * d_stride + 1],
                    dz = rays_d[ray * d_stride + 2];
        const float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
        const int64_t row = ray * S;
        const float gr = g_rgb ? g_rgb[ray * 3] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f,
                    gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
        const float gd = g_depth ? g_depth[ray] : 0.f;
        float ga = g_acc ? g_acc[ray] : 0.f;
        if (white_bkgd) ga -= (gr + gg) + gb;

        auto sample = [&](int s, float4& r, float& zz, float& dist, float& sig, float& e) {
            const bool valid = s < S;
            r = valid ? raw[row + s] : make_float4(0.f, 0.f, 0.f, 0.f);
            zz = valid ? z[row + s] : 0.f;
            dist = ((s + 1 < S) ? z[row + s + 1] - zz : 1e10f) * norm;
            sig = r.w;
            if (noise) sig = sig + (valid ? noise[row + s] : 0.f) * noise_std;
            e = valid ? __expf(-fmaxf(sig, 0.f) * dist) : 1.0f;  // 1 - alpha
            return valid;
        };

        // sweep 1: transmittance entering each chunk
        float carry = 1.0f;
        for (int c = 0; c < nch; ++c) {
            float4 r; float zz, dist, sig, e;
            const bool valid = sample(c * 64 + lane, r, zz, dist, sig, e);
            const float t = valid ? (1.0f - (1.0f - e)) + 1e-10f : 1.0f;
            if (lane == 0) carry_in[c] = carry;
            carry = carry * lane63(wave_scan_mul(t));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // sweep 2: far to near
        float carry_r = 0.f;
        for (int c = nch - 1; c >= 0; --c) {
            const int s = c * 64 + lane;
            float4 r; float zz, dist, sig, e;
            const bool valid = sample(s, r, zz, dist, sig, e);
            const float alpha = valid ? 1.0f - e : 0.f;
            const float t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
            const float T = carry_in[c] * wave_shr1(1.0f, wave_scan_mul(t));
            const float w = alpha * T;
            const float cr = sigmoidf(r.x), cg = sigmoidf(r.y), cb = sigmoidf(r.z);
            const float G = ((gr * cr + gg * cg) + gb * cb) + (gd * zz + ga);
            const float Gw = valid ? G * w : 0.f;
            const float suf = wave_rscan_add(Gw);
            const float R = wave_shl1(0.f, suf) + carry_r;   // sum over samples after this one
            carry_r = carry_r + read_lane(suf, 0);
            if (valid) {
                const float dalpha = G * T - R / t;
                const float dsig = (sig > 0.f) ? dist * (1.0f - alpha) * dalpha : 0.f;
                d_raw[row + s] = make_float4(gr * w * cr * (1.0f - cr), gg * w * cg * (1.0f - cg),
                                             gb * w * cb * (1.0f - cb), dsig);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int preload_composite_code(const char* what) { return preload_code_of(composite_fwd_kernel<3, 1>, what); }

}  // namespace nerf

using namespace nerf;

NERF_API int nerf_composite_bwd(const float* raw, const float* z, const float* rays_d,
                                int d_stride, int64_t n, int S, int white_bkgd,
                                const float* noise, float noise_std, const float* g_rgb,
                                const float* g_acc, const float* g_depth, float* d_raw,
                                nerf_stream_t stream) {
    NERF_REQUIRE(raw && z && rays_d && d_raw, "nerf_composite_bwd: null pointer");
    NERF_REQUIRE(n >= 0 && S > 0 && d_stride >= 3, "nerf_composite_bwd: bad size");
    NERF_REQUIRE(S <= 64 * kMaxChunks, "nerf_composite_bwd: at most %d samples per ray", 64 * kMaxChunks);
    NERF_REQUIRE(((uintptr_t)raw & 15) == 0 && ((uintptr_t)d_raw & 15) == 0,
                 "nerf_composite_bwd: raw and d_raw must be 16-byte aligned");
    if (n == 0) return 0;
    if (noise_std <= 0.f) noise = nullptr;
    const int64_t want = (n + 3) / 4;
    const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
    composite_bwd_kernel<<<blocks, 256, 0, as_stream(stream)>>>(
        (const float4*)raw, z, rays_d, d_stride, n, S, white_bkgd, noise, noise_std, g_rgb, g_acc,
        g_depth, (float4*)d_raw);
    return check_launch("nerf_composite_bwd");
}

NERF_API int nerf_composite_fwd(const float* raw, const float* z, const float* rays_d,
                                int d_stride, int64_t n, int S, int white_bkgd,
                                const float* noise, float noise_std, float* rgb_map,
                                float* disp_map, float* acc_map, float* depth_map, float* weights,
                                nerf_stream_t stream) {
    NERF_REQUIRE(raw && z && rays_d && rgb_map, "nerf_composite_fwd: null pointer");
    NERF_REQUIRE(n >= 0 && S > 0 && d_stride >= 3, "nerf_composite_fwd: bad size (n=%lld S=%d d_stride=%d)",
                 (long long)n, S, d_stride);
    NERF_REQUIRE(((uintptr_t)raw & 15) == 0, "nerf_composite_fwd: raw must be 16-byte aligned");
    if (n == 0) return 0;
    if (noise_std <= 0.f) noise = nullptr;
    // one wave per group of R rays, 4 waves per block, and as many blocks as that takes: waves
    // that start in ray order keep the chip's reads and writes in one moving window (measured on
    // 1 M rays of 192 samples: 6.0 TB/s against 5.2 TB/s with 32768 blocks walking 8 groups each)
    auto launch = [&](auto kernel, int R) {
        const int64_t want = (n + 4 * R - 1) / (4 * R);
        const unsigned blocks = (unsigned)(want < 0x40000000 ? want : 0x40000000);
        kernel<<<blocks, 256, 0, as_stream(stream)>>>((const float4*)raw, z, rays_d, d_stride, n, S, white_bkgd, noise,
                                                        noise_std, rgb_map, disp_map, acc_map, depth_map, weights);
    };
    const int nch = (S + 63) / 64;
    // up to 64 samples: half a wave per ray, two pairs of rays in flight per wave (measured on 131072
    // rays of 64 samples: 40.8 us = 5.0 TB/s against 46.7-50.5 us with one wave per ray; one pair 45.7 us,
    // three pairs 43.8 us)
    if (nch == 1) launch(composite_fwd_pair_kernel<2>, 4);
    else if (nch == 2) launch(composite_fwd_kernel<2, 2>, 2);
    else if (nch == 3) launch(composite_fwd_kernel<3, 1>, 1);
    else if (nch == 4) launch(composite_fwd_kernel<4, 1>, 1);
    else if (nch <= 6) launch(composite_fwd_kernel<6, 1>, 1);
    else launch(composite_fwd_kernel<0, 1>, 1);
    return check_launch("nerf_composite_fwd");
}
