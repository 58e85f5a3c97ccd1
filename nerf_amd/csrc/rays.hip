// Ray generation, NDC projection, ray-batch assembly, standalone positional encoding and
// stratified depth sampling.  All fp32 element-wise work, one thread per ray / point / sample,
// written with the reference's operation order (no fused multiply-add: the library is built
// with -ffp-contract=off) so results match an fp32 evaluation of the reference bit for bit
// where the operations are the same.
#include <stdarg.h>
#include "common.h"

namespace nerf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct Mat34 {
    float m[12];
};

// get_rays (run_nerf_helpers.py:122-129): pixel (row j, column i) ->
// dirs = [(i - W/2)/f, -(j - H/2)/f, -1]; rays_d[k] = sum_c dirs[c]*c2w[k][c]; rays_o = c2w[:,3].
__global__ void get_rays_kernel(int H, int W, float half_w, float half_h, float focal, Mat34 c2w,
                                float* __restrict__ rays_o, float* __restrict__ rays_d) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W) return;
    int j = (int)(idx / W), i = (int)(idx - (int64_t)j * W);
    float d0 = ((float)i - half_w) / focal;
    float d1 = -((float)j - half_h) / focal;
    float d2 = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = d0 * c2w.m[4 * k + 0] + d1 * c2w.m[4 * k + 1];
        s = s + d2 * c2w.m[4 * k + 2];
        rays_d[idx * 3 + k] = s;
        rays_o[idx * 3 + k] = c2w.m[4 * k + 3];
    }
}

// Rays and colours of n pixels picked from a set of posed images: get_rays' arithmetic per pixel
// with that image's pose (bit-identical to a lookup in the table run_nerf.py:634-656 builds).
__global__ void gather_pixel_rays_kernel(int H, int W, float half_w, float half_h, float focal,
                                         int n_img, const float* __restrict__ poses,
                                         const float* __restrict__ images, int channels,
                                         const int32_t* __restrict__ img_index,
                                         const int32_t* __restrict__ pix_index, int64_t n,
                                         float* __restrict__ out_rays, float* __restrict__ out_rgb) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int im = img_index[idx], pix = pix_index[idx];
    im = im < 0 ? 0 : (im >= n_img ? n_img - 1 : im);  // the host validates; never read out of bounds
    pix = pix < 0 ? 0 : (pix >= H * W ? H * W - 1 : pix);
    const int j = pix / W, i = pix - j * W;
    const float* c2w = poses + (size_t)im * 12;
    float d0 = ((float)i - half_w) / focal;
    float d1 = -((float)j - half_h) / focal;
    float d2 = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = d0 * c2w[4 * k + 0] + d1 * c2w[4 * k + 1];
        s = s + d2 * c2w[4 * k + 2];
        out_rays[(n + idx) * 3 + k] = s;
        out_rays[idx * 3 + k] = c2w[4 * k + 3];
    }
    if (images) {
        const float* px = images + ((size_t)im * H * W + pix) * channels;
#pragma unroll
        for (int k = 0; k < 3; ++k) out_rgb[idx * 3 + k] = px[k];
    }
}

// ndc_rays (run_nerf_helpers.py:163-175).  cw = -1/(W/(2f)), ch = -1/(H/(2f)) are formed on
// the host in double, as the reference's Python scalars are, then rounded to fp32.
__device__ __forceinline__ void ndc_project(float near, float cw, float ch, float (&o)[3],
                                            float (&d)[3]) {
    float t = -(near + o[2]) / d[2];
    o[0] = o[0] + t * d[0];
    o[1] = o[1] + t * d[1];
    o[2] = o[2] + t * d[2];
    float o0 = cw * o[0] / o[2];
    float o1 = ch * o[1] / o[2];
    float o2 = 1.0f + 2.0f * near / o[2];
    float e0 = cw * (d[0] / d[2] - o[0] / o[2]);
    float e1 = ch * (d[1] / d[2] - o[1] / o[2]);
    float e2 = -2.0f * near / o[2];
    o[0] = o0, o[1] = o1, o[2] = o2;
    d[0] = e0, d[1] = e1, d[2] = e2;
}

__global__ void ndc_rays_kernel(int64_t n, float near, float cw, float ch,
                                const float* rays_o, const float* rays_d, float* out_o,
                                float* out_d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]};
    float d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
    ndc_project(near, cw, ch, o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) out_o[3 * i + k] = o[k], out_d[3 * i + k] = d[k];
}

// render()'s ray-batch assembly (run_nerf.py:79-108).
__global__ void build_ray_batch_kernel(int64_t n, const float* __restrict__ rays_o,
                                       const float* __restrict__ rays_d,
                                       const float* __restrict__ view_src, int ndc, float cw,
                                       float ch, const float* __restrict__ near_arr,
                                       const float* __restrict__ far_arr, float near, float far,
                                       float* __restrict__ out, int width) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]};
    float d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
    float* r = out + i * width;
    if (width > 8) {
        float v0 = view_src[3 * i], v1 = view_src[3 * i + 1], v2 = view_src[3 * i + 2];
        float nrm = sqrtf((v0 * v0 + v1 * v1) + v2 * v2);
        r[8] = v0 / nrm, r[9] = v1 / nrm, r[10] = v2 / nrm;
    }
    if (ndc) ndc_project(1.0f, cw, ch, o, d);
    r[0] = o[0], r[1] = o[1], r[2] = o[2];
    r[3] = d[0], r[4] = d[1], r[5] = d[2];
    r[6] = near_arr ? near_arr[i] : near;
    r[7] = far_arr ? far_arr[i] : far;
}

// Embedder.embed (run_nerf_helpers.py:46-56): [x, sin(x f), cos(x f)] per frequency f = 2^k.
__global__ void embed_kernel(int64_t m, const float* __restrict__ x, int L,
                             float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int od = 3 + 6 * L;
    float v[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    float* o = out + i * od;
    o[0] = v[0], o[1] = v[1], o[2] = v[2];
    float f = 1.0f;
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = v[c] * f;
            o[3 + 6 * k + c] = sinf(a);
            o[3 + 6 * k + 3 + c] = cosf(a);
        }
        f *= 2.0f;
    }
}

// Depth of sample i of S on a ray: t = i/(S-1); linear in depth or in disparity.
__device__ __forceinline__ float z_at(int i, int S, float near, float far, int lindisp) {
    float t = (S > 1) ? (float)i / (float)(S - 1) : 0.0f;
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

__global__ void sample_z_kernel(int64_t n, int S, const float* __restrict__ rays, int width,
                                int lindisp, const float* __restrict__ perturb_u,
                                float* __restrict__ z_out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * S) return;
    int64_t r = idx / S;
    int i = (int)(idx - r * S);
    float near = rays[r * width + 6], far = rays[r * width + 7];
    float z = z_at(i, S, near, far, lindisp);
    if (perturb_u) {
        // mids = .5*(z[1:]+z[:-1]); upper = [mids, z_last]; lower = [z_first, mids]
        float upper = (i < S - 1) ? 0.5f * (z_at(i + 1, S, near, far, lindisp) + z) : z;
        float lower = (i > 0) ? 0.5f * (z + z_at(i - 1, S, near, far, lindisp)) : z;
        z = lower + (upper - lower) * perturb_u[idx];
    }
    z_out[idx] = z;
}

static inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

int preload_rays_code(const char* what) { return preload_code_of(sample_z_kernel, what); }

}  // namespace nerf

using namespace nerf;

NERF_API int nerf_abi_version(void) { return NERF_ABI_VERSION; }
NERF_API const char* nerf_last_error(void) { return g_err; }

NERF_API int nerf_get_rays(int H, int W, double focal, const float* c2w, float* rays_o,
                           float* rays_d, nerf_stream_t stream) {
    NERF_REQUIRE(H > 0 && W > 0 && c2w && rays_o && rays_d, "nerf_get_rays: bad argument");
    Mat34 m;
    for (int i = 0; i < 12; ++i) m.m[i] = c2w[i];
    int64_t n = (int64_t)H * W;
    get_rays_kernel<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(
        H, W, (float)(W * .5), (float)(H * .5), (float)focal, m, rays_o, rays_d);
    return check_launch("nerf_get_rays");
}

NERF_API int nerf_gather_pixel_rays(int H, int W, double focal, int n_img, const float* poses,
                                    const float* images, int channels, const int32_t* img_index,
                                    const int32_t* pix_index, int64_t n, float* out_rays,
                                    float* out_rgb, nerf_stream_t stream) {
    NERF_REQUIRE(H > 0 && W > 0 && n_img > 0 && n >= 0, "nerf_gather_pixel_rays: bad size");
    NERF_REQUIRE((int64_t)H * W <= 0x7fffffffLL, "nerf_gather_pixel_rays: image too large for int32 pixel ids");
    NERF_REQUIRE(poses && img_index && pix_index && out_rays, "nerf_gather_pixel_rays: null pointer");
    NERF_REQUIRE((images == nullptr) == (out_rgb == nullptr), "nerf_gather_pixel_rays: images and out_rgb go together");
    NERF_REQUIRE(images == nullptr || channels >= 3, "nerf_gather_pixel_rays: images need >= 3 channels, got %d", channels);
    if (n == 0) return 0;
    gather_pixel_rays_kernel<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(
        H, W, (float)(W * .5), (float)(H * .5), (float)focal, n_img, poses, images, channels, img_index,
        pix_index, n, out_rays, out_rgb);
    return check_launch("nerf_gather_pixel_rays");
}

NERF_API int nerf_ndc_rays(int64_t n, int H, int W, double focal, double near,
                           const float* rays_o, const float* rays_d, float* out_o, float* out_d,
                           nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && rays_o && rays_d && out_o && out_d, "nerf_ndc_rays: bad argument");
    if (n == 0) return 0;
    float cw = (float)(-1. / (W / (2. * focal))), ch = (float)(-1. / (H / (2. * focal)));
    ndc_rays_kernel<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(n, (float)near, cw, ch,
                                                                       rays_o, rays_d, out_o, out_d);
    return check_launch("nerf_ndc_rays");
}

NERF_API int nerf_build_ray_batch(int64_t n, const float* rays_o, const float* rays_d,
                                  const float* view_src, int ndc, int H, int W, double focal,
                                  const float* near_arr, const float* far_arr, double near,
                                  double far, float* out, int width, nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && rays_o && rays_d && out, "nerf_build_ray_batch: null pointer");
    NERF_REQUIRE(width == 8 || width == 11, "nerf_build_ray_batch: width must be 8 or 11, got %d", width);
    NERF_REQUIRE((width == 11) == (view_src != nullptr),
                 "nerf_build_ray_batch: view_src must be given exactly when width is 11");
    NERF_REQUIRE((near_arr == nullptr) == (far_arr == nullptr),
                 "nerf_build_ray_batch: near_arr and far_arr go together");
    if (n == 0) return 0;
    float cw = 0.f, ch = 0.f;
    if (ndc) {
        NERF_REQUIRE(H > 0 && W > 0 && focal != 0., "nerf_build_ray_batch: ndc needs H, W, focal");
        cw = (float)(-1. / (W / (2. * focal))), ch = (float)(-1. / (H / (2. * focal)));
    }
    build_ray_batch_kernel<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(
        n, rays_o, rays_d, view_src, ndc, cw, ch, near_arr, far_arr, (float)near, (float)far, out,
        width);
    return check_launch("nerf_build_ray_batch");
}

NERF_API int nerf_embed(int64_t m, const float* x, int L, float* out, nerf_stream_t stream) {
    NERF_REQUIRE(m >= 0 && x && out && L >= 0 && L <= 32, "nerf_embed: bad argument");
    if (m == 0) return 0;
    embed_kernel<<<blocks_for(m, 256), 256, 0, as_stream(stream)>>>(m, x, L, out);
    return check_launch("nerf_embed");
}

NERF_API int nerf_sample_z(int64_t n, int S, const float* rays, int width, int lindisp,
                           const float* perturb_u, float* z_out, nerf_stream_t stream) {
    NERF_REQUIRE(n >= 0 && S > 0 && rays && z_out && width >= 8, "nerf_sample_z: bad argument");
    if (n == 0) return 0;
    sample_z_kernel<<<blocks_for(n * S, 256), 256, 0, as_stream(stream)>>>(n, S, rays, width,
                                                                          lindisp, perturb_u, z_out);
    return check_launch("nerf_sample_z");
}
