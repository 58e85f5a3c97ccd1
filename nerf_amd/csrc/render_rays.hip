// render_rays as ONE call of the C ABI: stratified depths -> fused encoding + coarse network ->
// compositing -> resampling + merge -> fused encoding + fine network -> compositing, six kernel
// launches on the caller's stream with every intermediate ([n,S] depths, [n,S,4] network outputs,
// [n,S] weights) in a workspace instead of in tensors the host language allocates per stage.
// The stages are the library's own entry points, so the results are bit-identical to calling them
// one by one.
#include <mutex>
#include "common.h"

namespace nerf {

constexpr size_t kAlign = 256;
inline size_t aligned(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

// Carves the workspace; returns the bytes used.
struct RenderWorkspace {
    float *z_c, *raw_c, *w_c, *z_f, *raw_f, *rgb_tmp;
    size_t bytes;
    RenderWorkspace(char* base, int64_t n, int Sc, int Si) {
        size_t at = 0;
        auto take = [&](size_t count) {
            float* p = (float*)(base + at);
            at += aligned(count * sizeof(float));
            return p;
        };
        const size_t N = (size_t)n, Sf = (size_t)Sc + Si;
        z_c = take(N * Sc), raw_c = take(N * Sc * 4), w_c = take(N * Sc);
        z_f = Si > 0 ? take(N * Sf) : nullptr;
        raw_f = Si > 0 ? take(N * Sf * 4) : nullptr;
        rgb_tmp = take(N * 3);  // the coarse colours when the caller does not want them
        bytes = at;
    }
};

static int render_rays_impl(const void* packed_coarse, const void* packed_fine, const float* rays, int width,
                            int64_t n, int Sc, int Si, int lindisp, int white_bkgd, const float* perturb_u,
                            const float* pdf_u, float* rgb_map, float* disp_map, float* acc_map, float* depth_map,
                            float* rgb0, float* disp0, float* acc0, float* z_std, char* workspace,
                            nerf_stream_t stream) {
    const RenderWorkspace ws(workspace, n, Sc, Si);
    int rc = nerf_sample_z(n, Sc, rays, width, lindisp, perturb_u, ws.z_c, stream);
    if (rc) return rc;
    rc = nerf_mlp_forward_rays(packed_coarse, rays, width, ws.z_c, n, Sc, ws.raw_c, stream);
    if (rc) return rc;
    if (Si <= 0)
        return nerf_composite_fwd(ws.raw_c, ws.z_c, rays + 3, width, n, Sc, white_bkgd, nullptr, 0.f, rgb_map, disp_map,
                                  acc_map, depth_map, nullptr, stream);
    rc = nerf_composite_fwd(ws.raw_c, ws.z_c, rays + 3, width, n, Sc, white_bkgd, nullptr, 0.f,
                            rgb0 ? rgb0 : ws.rgb_tmp, disp0, acc0, nullptr, ws.w_c, stream);
    if (rc) return rc;
    rc = nerf_resample_merge(ws.z_c, ws.w_c, n, Sc, Si, pdf_u ? 0 : 1, pdf_u, ws.z_f, nullptr, z_std, stream);
    if (rc) return rc;
    rc = nerf_mlp_forward_rays(packed_fine ? packed_fine : packed_coarse, rays, width, ws.z_f, n, Sc + Si, ws.raw_f, stream);
    if (rc) return rc;
    return nerf_composite_fwd(ws.raw_f, ws.z_f, rays + 3, width, n, Sc + Si, white_bkgd, nullptr, 0.f, rgb_map, disp_map,
                              acc_map, depth_map, nullptr, stream);
}

static int check_render_args(const char* what, const void* packed_coarse, const float* rays, int width, int64_t n, int Sc,
                             int Si, const float* rgb_map) {
    if (!packed_coarse || !rays || !rgb_map) return NERF_FAIL(NERF_E_ARG, "%s: null pointer", what);
    if (width < 11) return NERF_FAIL(NERF_E_ARG, "%s: ray batch needs view directions (width >= 11, got %d)", what, width);
    if (n < 0 || Sc < 1 || Si < 0) return NERF_FAIL(NERF_E_ARG, "%s: bad size", what);
    if (Si > 0 && Sc < 3) return NERF_FAIL(NERF_E_ARG, "%s: resampling needs at least 3 coarse samples", what);
    return 0;
}

}  // namespace nerf

using namespace nerf;

NERF_API size_t nerf_render_rays_workspace_bytes(int64_t n_rays, int N_samples, int N_importance) {
    if (n_rays <= 0 || N_samples < 1 || N_importance < 0) return 0;
    return RenderWorkspace(nullptr, n_rays, N_samples, N_importance).bytes;
}

NERF_API int nerf_render_rays(const void* packed_coarse, const void* packed_fine, const float* rays, int width,
                              int64_t n_rays, int N_samples, int N_importance, int lindisp, int white_bkgd,
                              const float* perturb_u, const float* pdf_u, float* rgb_map, float* disp_map,
                              float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0, float* z_std,
                              void* workspace, nerf_stream_t stream) {
    int rc = check_render_args("nerf_render_rays", packed_coarse, rays, width, n_rays, N_samples, N_importance, rgb_map);
    if (rc) return rc;
    NERF_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0, "nerf_render_rays: workspace must be a 16-byte aligned device buffer");
    if (n_rays == 0) return 0;
    return render_rays_impl(packed_coarse, packed_fine, rays, width, n_rays, N_samples, N_importance, lindisp, white_bkgd,
                            perturb_u, pdf_u, rgb_map, disp_map, acc_map, depth_map, rgb0, disp0, acc0, z_std,
                            (char*)workspace, stream);
}

// ---------------------------------------------------------------------------------------------
// The same call for callers that cannot allocate device memory and have no workspace to give
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
// intermediates live in a fixed array of the code object.  Each GPU the library runs on holds its
// own copy of the array at its own address until the process ends; the address is looked up once
// per device and only that pointer is ever passed on (the array's name is a host-side stand-in,
// not a device address).  Every stage overwrites what it uses, so nothing has to be cleared
// between calls; calls that overlap in time on one GPU would share the array and race.
// ---------------------------------------------------------------------------------------------
namespace nerf {

constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

static std::mutex g_scratch_lock;
static void* g_scratch_ptr[64] = {};  // per device

// The current device's copy of `scratch` (looked up on first use, with that device current).
static int static_scratch(char** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return NERF_FAIL(NERF_E_UNSUPPORTED, "nerf_render_rays_static: no current device");
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return NERF_FAIL(e != hipSuccess ? (int)e : NERF_E_UNSUPPORTED, "nerf_render_rays_static: hipGetSymbolAddress: %s",
                             hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = (char*)g_scratch_ptr[dev];
    return 0;
}

}  // namespace nerf

NERF_API size_t nerf_render_rays_static_bytes(void) { return kStaticScratchBytes; }

NERF_API int nerf_render_rays_static(const void* packed_coarse, const void* packed_fine, const float* rays, int width,
                                     int64_t n_rays, int N_samples, int N_importance, int lindisp, int white_bkgd,
                                     const float* perturb_u, const float* pdf_u, float* rgb_map, float* disp_map,
                                     float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0,
                                     float* z_std, nerf_stream_t stream) {
    int rc = check_render_args("nerf_render_rays_static", packed_coarse, rays, width, n_rays, N_samples, N_importance, rgb_map);
    if (rc) return rc;
    if (n_rays == 0) {
        // set-up only: the code object that holds the array, and those of the kernels a call
        // launches, are loaded on the current device now (the device's free memory drops by the
        // array's size and a few MiB of code here, once, and by nothing in later calls)
        char* ws = nullptr;
        rc = static_scratch(&ws);
        if (!rc) rc = preload_rays_code("nerf_render_rays_static");
        if (!rc) rc = preload_mlp_forward_code("nerf_render_rays_static");
        if (!rc) rc = preload_composite_code("nerf_render_rays_static");
        if (!rc) rc = preload_sample_pdf_code("nerf_render_rays_static");
        return rc;
    }
    const size_t need = nerf_render_rays_workspace_bytes(n_rays, N_samples, N_importance);
    if (need > kStaticScratchBytes)
        return NERF_FAIL(NERF_E_UNSUPPORTED,
                         "nerf_render_rays_static: %lld rays at %d + %d samples need %zu bytes of scratch, the built-in array "
                         "holds %zu (use fewer rays per call, or nerf_render_rays with a workspace)",
                         (long long)n_rays, N_samples, N_importance, need, kStaticScratchBytes);
    char* ws = nullptr;
    rc = static_scratch(&ws);
    if (rc) return rc;
    return render_rays_impl(packed_coarse, packed_fine, rays, width, n_rays, N_samples, N_importance, lindisp, white_bkgd,
                            perturb_u, pdf_u, rgb_map, disp_map, acc_map, depth_map, rgb0, disp0, acc0, z_std, ws, stream);
}
