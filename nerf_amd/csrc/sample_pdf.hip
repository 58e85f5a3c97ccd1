// Hierarchical inverse-CDF resampling (sample_pdf) and the sorted merge with the coarse depths,
// one 64-lane wave per ray.  The ray's CDF and bin edges live in the wave's slice of LDS; every
// lane inverts the CDF for its own samples by binary search (searchsorted side='right'), and the
// two sorted depth lists are merged by rank (each element's output slot = its own index + the
// count of elements of the other list that precede it), so no general sort is needed when the
// uniforms are the deterministic linspace; a sample's rank then follows from the bin it was
// drawn from, without a search.  With caller-supplied random uniforms the samples are first
// sorted in LDS by a bitonic network.  Searches are fixed-length halving loops.  HBM traffic: z and weights read once
// (coalesced), merged depths written once.
#include "common.h"

namespace nerf {

__device__ __forceinline__ void wave_lds_sync() {
    // One wave's LDS operations execute in issue order; this only drains them and stops the
    // compiler from moving LDS accesses across the point.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct PdfArgs {
    // merge mode: z_vals/weights are [n,Sc]; direct mode: bins [n,nb], weights [n,nb-1]
    const float* bins_or_z;
    const float* weights;
    const float* u;
    float* samples;  // [n,ns] or null
    float* z_out;    // [n,Sc+ns] (merge mode) or null
    float* z_std;    // [n] or null
    int64_t n;
    int nb;       // number of bin edges (= Sc-1 in merge mode)
    int ns;       // samples to draw
    int ns_pow2;  // ns rounded up to a power of two (bitonic network size)
    int top_b, top_s, top_c;  // largest powers of two <= nb, ns, Sc: first steps of the halving searches
    int Sc;       // coarse depth count (merge mode), 0 in direct mode
    int det;
    int lds_floats_per_wave;
};

// What a lane reads of one ray, PRE 64-element chunks of each row: depths (or bin edges), the
// depth after each (merge mode) and the weights.
template <int PRE>
struct PdfRow {
    float z[PRE > 0 ? PRE : 1], zn[PRE > 0 ? PRE : 1], w[PRE > 0 ? PRE : 1];
};

// PRE > 0 (rows of at most 64 PRE elements): a wave reads its NEXT ray's row into registers
// before it works on the current one.  One ray is a serial chain of about 5 us (global loads,
// sum, scan, a dozen dependent LDS probes) and a SIMD holds 8 waves, so the kernel is bound by
// that chain, not by bytes or instructions; the prefetch takes the 2 us of load latency out of it.
// PRE = 0: rows of any length, read where they are needed.
template <bool MERGE, int PRE>
__global__ __launch_bounds__(256) void sample_pdf_kernel(PdfArgs a) {
    extern __shared__ float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    // Arrays that are searched are padded with +inf up to twice the first step of their halving
    // search, so that a probe needs neither an index clamp nor a range check.
    float* cdf = lds_all + (size_t)wave_in_block * a.lds_floats_per_wave;  // [2 top_b]
    float* bins = cdf + 2 * a.top_b;          // [nb]
    float* sA = bins + a.nb;                  // [2 top_c] (merge mode) coarse depths
    float* sB = sA + (MERGE ? 2 * a.top_c : 0);  // [2 top_s] samples
    float* sOut = sB + 2 * a.top_s;           // [Sc+ns]   (merge mode)
    const int nb = a.nb, nw = a.nb - 1, ns = a.ns, Sc = a.Sc;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

    auto fetch = [&](int64_t ray, PdfRow<PRE>& r) {
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int i = 64 * c + lane;
            if (MERGE) {
                const float* zr = a.bins_or_z + ray * Sc;
                r.z[c] = i < Sc ? zr[i] : __builtin_inff();
                r.zn[c] = i + 1 < Sc ? zr[i + 1] : 0.f;
                r.w[c] = i < nw ? a.weights[ray * Sc + 1 + i] : 0.f;  // weights[:, 1:-1]
            } else {
                r.z[c] = i < nb ? a.bins_or_z[ray * nb + i] : 0.f;
                r.zn[c] = 0.f;
                r.w[c] = i < nw ? a.weights[ray * nw + i] : 0.f;
            }
        }
    };
    PdfRow<PRE> next_row;
    if (PRE > 0 && wave0 < a.n) fetch(wave0, next_row);

    for (int64_t ray = wave0; ray < a.n; ray += nwaves) {
        // ---- stage bins (and the coarse depths) in LDS; weights + 1e-5 and their sum ----
        const PdfRow<PRE> row = next_row;
        if (PRE > 0 && ray + nwaves < a.n) fetch(ray + nwaves, next_row);
        const float* wsrc = MERGE ? a.weights + ray * Sc + 1 : a.weights + ray * nw;
        float part = 0.f;
        if constexpr (PRE > 0) {
#pragma unroll
            for (int c = 0; c < PRE; ++c) {
                const int i = 64 * c + lane;
                if (MERGE) {
                    if (i < 2 * a.top_c) sA[i] = row.z[c];
                    if (i + 1 < Sc) bins[i] = 0.5f * (row.zn[c] + row.z[c]);  // z_mid
                } else if (i < nb) bins[i] = row.z[c];
                if (i < nw) part += row.w[c] + 1e-5f;
            }
            if (MERGE)
                for (int i = 64 * PRE + lane; i < 2 * a.top_c; i += 64) sA[i] = __builtin_inff();
        } else {
            if (MERGE) {
                const float* zr = a.bins_or_z + ray * Sc;
                for (int i = lane; i < 2 * a.top_c; i += 64) {
                    const float zi = i < Sc ? zr[i] : __builtin_inff();
                    sA[i] = zi;
                    if (i + 1 < Sc) bins[i] = 0.5f * (zr[i + 1] + zi);  // z_mid
                }
            } else {
                const float* br = a.bins_or_z + ray * nb;
                for (int i = lane; i < nb; i += 64) bins[i] = br[i];
            }
            for (int i = lane; i < nw; i += 64) part += wsrc[i] + 1e-5f;
        }
        const float total = wave_sum(part);
        // ---- cdf = [0, cumsum(w/total)] ----
        float carry = 0.f;
        if (lane == 0) cdf[0] = 0.f;
        auto cdf_chunk = [&](int base, float wi) {
            const int i = base + lane;
            const float p = (i < nw) ? (wi + 1e-5f) / total : 0.f;
            const float incl = wave_scan_add(p);
            if (i < 2 * a.top_b - 1) cdf[i + 1] = i < nw ? carry + incl : __builtin_inff();
            carry = carry + lane63(incl);
        };
        if constexpr (PRE > 0) {
#pragma unroll
            for (int c = 0; c < PRE; ++c)
                if (64 * c < nw) cdf_chunk(64 * c, row.w[c]);
            for (int i = ((nw + 63) & ~63) + lane; i < 2 * a.top_b - 1; i += 64) cdf[i + 1] = __builtin_inff();
        } else {
            for (int base = 0; base < 2 * a.top_b - 1; base += 64) cdf_chunk(base, base + lane < nw ? wsrc[base + lane] : 0.f);
        }
        wave_lds_sync();
        // ---- invert the CDF ----
        // searchsorted(side='right') as a fixed number of halving steps (no divergence; the
        // lane's two samples of a round search side by side, which hides half the LDS latency)
        const int top = a.top_b;
        auto draw = [&](int k) { return a.det ? ((ns > 1) ? (float)k / (float)(ns - 1) : 0.f) : a.u[ray * ns + k]; };
        auto finish = [&](int k, float u, int lo) {  // lo = first index with cdf[idx] > u
            const int below = max(0, lo - 1), above = min(nb - 1, lo);
            const float c0 = cdf[below], c1 = cdf[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (u - c0) / denom;
            const float b0 = bins[below], b1 = bins[above];
            const float smp = b0 + t * (b1 - b0);
            if (a.samples) a.samples[ray * ns + k] = smp;
            sB[k] = smp;
            if (MERGE && a.det) {
                // Deterministic uniforms ascend, so the samples do and sample k's slot is k + the
                // number of coarse depths <= it.  It lies in [z_mid[below], z_mid[above]], so that
                // number is below + 1 plus the few depths just above that it still reaches.
                int c = below + 1;
                while (c < Sc && sA[c] <= smp) ++c;
                sOut[k + c] = smp;
            }
            return smp;
        };
        float ssum = 0.f;
        for (int k0 = lane; k0 < a.ns_pow2; k0 += 128) {
            const int k1 = k0 + 64;
            const bool v0 = k0 < ns, v1 = k1 < ns;
            const float u0 = v0 ? draw(k0) : 0.f, u1 = v1 ? draw(k1) : 0.f;
            int lo0 = 0, lo1 = 0;
            for (int step = top; step > 0; step >>= 1) {  // probes stay below 2 top; beyond nb they read +inf
                const int m0 = lo0 + step, m1 = lo1 + step;
                const float q0 = cdf[m0 - 1], q1 = cdf[m1 - 1];
                lo0 = q0 <= u0 ? m0 : lo0;
                lo1 = q1 <= u1 ? m1 : lo1;
            }
            if (v0) ssum += finish(k0, u0, lo0); else sB[k0] = __builtin_inff();  // padding of the sort network
            if (k1 < a.ns_pow2) { if (v1) ssum += finish(k1, u1, lo1); else sB[k1] = __builtin_inff(); }
        }
        for (int k = a.ns_pow2 + lane; k < 2 * a.top_s; k += 64) sB[k] = __builtin_inff();  // padding of the rank search
        if (a.z_std) {
            const float mean = wave_sum(ssum) / (float)ns;
            wave_lds_sync();
            float v = 0.f;
            for (int k = lane; k < ns; k += 64) {
                const float d = sB[k] - mean;
                v += d * d;
            }
            v = wave_sum(v);
            if (lane == 0) a.z_std[ray] = sqrtf(v / (float)ns);
        }
        if (MERGE) {
            wave_lds_sync();
            if (!a.det) {
                // bitonic sort of sB[0..ns_pow2) ascending (padding is +inf)
                const int P = a.ns_pow2;
                for (int k = 2; k <= P; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = lane; t < (P >> 1); t += 64) {
                            const int i = 2 * j * (t / j) + (t % j);
                            const float x = sB[i], y = sB[i + j];
                            const bool up = (i & k) == 0;
                            if ((x > y) == up) { sB[i] = y; sB[i + j] = x; }
                        }
                        wave_lds_sync();
                    }
                }
            }
            // rank merge: ties put the coarse depth first
            const int tops = a.top_s, topc = a.top_c;
            for (int i = lane; i < Sc; i += 64) {
                const float v = sA[i];
                int lo = 0;  // count of samples < v
                for (int step = tops; step > 0; step >>= 1) {
                    const int m = lo + step;
                    lo = sB[m - 1] < v ? m : lo;
                }
                sOut[i + lo] = v;
            }
            if (!a.det) {
                for (int j = lane; j < ns; j += 64) {
                    const float v = sB[j];
                    int lo = 0;  // count of coarse depths <= v
                    for (int step = topc; step > 0; step >>= 1) {
                        const int m = lo + step;
                        lo = sA[m - 1] <= v ? m : lo;
                    }
                    sOut[j + lo] = v;
                }
            }
            wave_lds_sync();
            float* zo = a.z_out + ray * (int64_t)(Sc + ns);
            for (int i = lane; i < Sc + ns; i += 64) zo[i] = sOut[i];
        }
        wave_lds_sync();  // LDS slice is reused by the wave's next ray
    }
}

static int launch_pdf(PdfArgs a, bool merge, hipStream_t st, const char* what) {
    int p2 = 1;
    while (p2 < a.ns) p2 <<= 1;
    a.ns_pow2 = p2;
    auto pow2_below = [](int v) { int t = 1; while (2 * t <= v) t *= 2; return t; };
    a.top_b = pow2_below(a.nb), a.top_s = pow2_below(a.ns), a.top_c = merge ? pow2_below(a.Sc) : 1;
    a.lds_floats_per_wave = 2 * a.top_b + a.nb + 2 * a.top_s + (merge ? (2 * a.top_c + a.Sc + a.ns) : 0);
    const size_t per_wave = (size_t)a.lds_floats_per_wave * sizeof(float);
    int waves = 4;
    while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
    if (per_wave * waves > 64 * 1024)
        return NERF_FAIL(NERF_E_UNSUPPORTED, "%s: %d bins / %d samples need %zu B of LDS per ray (max 65536)",
                         what, a.nb, a.ns, per_wave);
    const int64_t want = (a.n + waves - 1) / waves;
    const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
    const int row = merge ? a.Sc : a.nb;  // longest row a lane reads (the weights row is shorter)
    auto launch = [&](auto kernel) { kernel<<<blocks, waves * 64, per_wave * waves, st>>>(a); };
    if (merge) {
        if (row <= 64) launch(sample_pdf_kernel<true, 1>);
        else if (row <= 128) launch(sample_pdf_kernel<true, 2>);
        else if (row <= 256) launch(sample_pdf_kernel<true, 4>);
        else launch(sample_pdf_kernel<true, 0>);
    } else {
        if (row <= 64) launch(sample_pdf_kernel<false, 1>);
        else if (row <= 128) launch(sample_pdf_kernel<false, 2>);
        else if (row <= 256) launch(sample_pdf_kernel<false, 4>);
        else launch(sample_pdf_kernel<false, 0>);
    }
    return check_launch(what);
}

int preload_sample_pdf_code(const char* what) { return preload_code_of(sample_pdf_kernel<true, 1>, what); }

}  // namespace nerf

using namespace nerf;

NERF_API int nerf_sample_pdf(const float* bins, const float* weights, int64_t n, int nb, int ns,
                             int det, const float* u, float* samples, nerf_stream_t stream) {
    NERF_REQUIRE(bins && weights && samples, "nerf_sample_pdf: null pointer");
    NERF_REQUIRE(n >= 0 && nb >= 2 && ns >= 1, "nerf_sample_pdf: bad size (n=%lld nb=%d ns=%d)",
                 (long long)n, nb, ns);
    NERF_REQUIRE(det || u, "nerf_sample_pdf: det == 0 needs the uniforms u");
    if (n == 0) return 0;
    PdfArgs a{};
    a.bins_or_z = bins, a.weights = weights, a.u = u, a.samples = samples;
    a.n = n, a.nb = nb, a.ns = ns, a.Sc = 0, a.det = det;
    return launch_pdf(a, false, as_stream(stream), "nerf_sample_pdf");
}

NERF_API int nerf_resample_merge(const float* z_vals, const float* weights, int64_t n, int Sc,
                                 int Si, int det, const float* u, float* z_out, float* z_samples,
                                 float* z_std, nerf_stream_t stream) {
    NERF_REQUIRE(z_vals && weights && z_out, "nerf_resample_merge: null pointer");
    NERF_REQUIRE(n >= 0 && Sc >= 3 && Si >= 1, "nerf_resample_merge: bad size (n=%lld Sc=%d Si=%d)",
                 (long long)n, Sc, Si);
    NERF_REQUIRE(det || u, "nerf_resample_merge: det == 0 needs the uniforms u");
    if (n == 0) return 0;
    PdfArgs a{};
    a.bins_or_z = z_vals, a.weights = weights, a.u = u, a.samples = z_samples, a.z_out = z_out,
    a.z_std = z_std;
    a.n = n, a.nb = Sc - 1, a.ns = Si, a.Sc = Sc, a.det = det;
    return launch_pdf(a, true, as_stream(stream), "nerf_resample_merge");
}
