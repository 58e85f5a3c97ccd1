/*
 * nerf_hip.h -- C ABI of libnerfhip.so, the MI355X (gfx950) implementation of NeRF's per-ray
 * render core.
 *
 * The reference (sherwoac/nerf) is pure Python/TensorFlow and has no FFI; its boundary for this
 * path is the Python call signatures of render()/render_rays()/create_nerf() and the helpers in
 * run_nerf_helpers.py.  The package nerf_amd/ mirrors those signatures on torch tensors and
 * binds the entry points below through ctypes.  Each entry point names the reference
 * function(s) whose arithmetic it replaces (file:line under the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the parameter is documented
 *     as "host"; the library allocates and frees nothing;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream) and keeps no state between calls;
 *   - all tensors are dense row-major fp32 unless stated; scalar camera parameters are double
 *     because the reference forms constants such as -1/(W/(2*focal)) in Python doubles before
 *     rounding them to fp32;
 *   - the return value is 0 on success, a negative NERF_E_* code for a rejected argument, or a
 *     positive hipError_t; nerf_last_error() returns a thread-local message for the last
 *     non-zero return.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 1

#define NERF_E_ARG (-1)        /* null pointer / non-positive size / unsupported width      */
#define NERF_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for              */

typedef void* nerf_stream_t;

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* ---- ray generation --------------------------------------------------------------------- */

/* get_rays / get_rays_np (run_nerf_helpers.py:122-129, :135-142).
 * c2w: HOST pointer to 12 floats, the row-major [3,4] camera-to-world matrix.
 * rays_o, rays_d: [H,W,3].  No half-pixel offset; directions are not normalised. */
int nerf_get_rays(int H, int W, double focal, const float* c2w, float* rays_o, float* rays_d,
                  nerf_stream_t stream);

/* ndc_rays (run_nerf_helpers.py:145-180), same operation order.  In/out [n,3]; out may alias in. */
int nerf_ndc_rays(int64_t n, int H, int W, double focal, double near, const float* rays_o,
                  const float* rays_d, float* out_o, float* out_d, nerf_stream_t stream);

/* render()'s ray-batch assembly (run_nerf.py:79-108): viewdirs = view_src/||view_src|| taken
 * before NDC, optional NDC projection with near plane 1, then
 * out[i] = [o(3), d(3), near, far, (viewdirs(3))], width = 8 or 11.
 * view_src: [n,3] or NULL (width 8).  near_arr/far_arr: per-ray [n] or NULL -> scalars. */
int nerf_build_ray_batch(int64_t n, const float* rays_o, const float* rays_d,
                         const float* view_src, int ndc, int H, int W, double focal,
                         const float* near_arr, const float* far_arr, double near, double far,
                         float* out, int width, nerf_stream_t stream);

/* Host helper of the image reader that stands in for imageio.imread (load_blender.py:112,
 * load_llff.py:134-140): reverses the PNG scanline filters.  raw: h rows of 1 filter byte +
 * stride bytes (host memory); out: h x stride bytes (host memory); bpp: bytes per pixel. */
int nerf_png_unfilter(const unsigned char* raw, int64_t h, int64_t stride, int bpp, unsigned char* out);

/* Training ray batches (run_nerf.py:615-656 builds a [N*H*W, ro+rd+rgb, 3] table with get_rays_np
 * for every training pose and slices N_rand shuffled rows per step, :711-731; the no_batching
 * branch, :736-766, gathers N_rand pixels of one image).  Both are "pixel ids -> (ray origin, ray
 * direction, observed colour)": the rays are formed here from the pose of the pixel's image with
 * get_rays' arithmetic, so no ray table is ever materialised.
 * poses: [n_img,3,4] camera-to-world; images: [n_img,H,W,channels] fp32 (first 3 channels are
 * used; NULL: no colours wanted); img_index, pix_index: [n] int32, pixel = row*W + column.
 * out_rays: [2,n,3] (origins then directions, the layout render(rays=...) takes); out_rgb: [n,3]. */
int nerf_gather_pixel_rays(int H, int W, double focal, int n_img, const float* poses,
                           const float* images, int channels, const int32_t* img_index,
                           const int32_t* pix_index, int64_t n, float* out_rays, float* out_rgb,
                           nerf_stream_t stream);

/* ---- positional encoding (standalone form) ------------------------------------------------ */

/* Embedder.embed (run_nerf_helpers.py:22-56): out[m, 3+6L] = [x, sin(x 2^0), cos(x 2^0), ...]. */
int nerf_embed(int64_t m, const float* x, int L, float* out, nerf_stream_t stream);

/* ---- depth sampling ----------------------------------------------------------------------- */

/* render_rays' stratified depths (NERFCO.nerf_renderer.render_rays, source absent from the
 * reference tree; in-tree twin tiny_nerf.ipynb "z_vals = tf.linspace(near, far, N_samples)").
 * rays: [n,width] ray batch (near at column 6, far at 7).  perturb_u: [n,S] uniforms in [0,1)
 * or NULL (no jitter).  z_out: [n,S]. */
int nerf_sample_z(int64_t n, int S, const float* rays, int width, int lindisp,
                  const float* perturb_u, float* z_out, nerf_stream_t stream);

/* ---- the MLP ------------------------------------------------------------------------------ */

/* Size in bytes of the packed fp16 weight stream + fp32 biases one network occupies. */
size_t nerf_mlp_packed_bytes(void);

/* Re-lays one init_nerf_model(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4],
 * use_viewdirs=True) network (run_nerf_helpers.py:80-117) into the MFMA fragment stream the
 * forward kernel consumes.  params: HOST array of 24 device pointers in get_weights() order --
 * kernel [in,out] and bias [out] of the 8 point layers, bottleneck, view layer, rgb, alpha. */
int nerf_mlp_pack(const float* const* params, void* packed, nerf_stream_t stream);

/* run_network + the Keras model call (run_nerf.py:209-214; run_nerf_helpers.py:22-56, :89-112,
 * :216-223) with the query points formed on the fly as o + d*z (render_rays).  Positional
 * encoding (L=10 points, L=4 view directions) is computed in-kernel; matmuls are fp16 MFMA with
 * fp32 accumulation.  Hidden activations are carried in fp16 and saturate at 65504 (a unit
 * that exceeds the fp16 range is held at the largest finite value, it does not become inf).
 * rays [n_rays,width>=11], z [n_rays,S] -> raw [n_rays,S,4] = (r,g,b,sigma) pre-activation.
 * At most 2^24 samples per ray. */
int nerf_mlp_forward_rays(const void* packed, const float* rays, int width, const float* z,
                          int64_t n_rays, int S, float* raw, nerf_stream_t stream);

/* Same network on explicit points: network_query_fn(inputs[n,S,3], viewdirs[n,3], network_fn)
 * (run_nerf.py:209-214; extract_mesh.ipynb "raw = net_fn(...)") -> raw [n,S,4]. */
int nerf_mlp_forward_pts(const void* packed, const float* pts, const float* viewdirs,
                         int64_t n, int S, float* raw, nerf_stream_t stream);

/* ---- compositing -------------------------------------------------------------------------- */

/* raw2outputs (NERFCO.nerf_renderer, source absent; in-tree core lines tiny_nerf.ipynb
 * render_rays cell: sigma = relu(raw[...,3]), rgb = sigmoid(raw[...,:3]), dists, alpha,
 * weights = alpha * exclusive cumprod(1-alpha+1e-10), rgb/depth/acc maps), plus upstream's
 * dists *= ||rays_d||, disp = 1/max(1e-10, depth/acc) and the white-background add.
 * rays_d is read at rays_d[i*d_stride + 0..2].  noise: [n,S] standard normal or NULL.
 * Any output pointer except rgb_map may be NULL. */
int nerf_composite_fwd(const float* raw, const float* z, const float* rays_d, int d_stride,
                       int64_t n, int S, int white_bkgd, const float* noise, float noise_std,
                       float* rgb_map, float* disp_map, float* acc_map, float* depth_map,
                       float* weights, nerf_stream_t stream);

/* ---- hierarchical resampling -------------------------------------------------------------- */

/* sample_pdf (run_nerf_helpers.py:185-213).  bins [n,nb], weights [n,nb-1] -> samples [n,ns].
 * det != 0: u = linspace(0,1,ns); else u: [n,ns] uniforms supplied by the caller. */
int nerf_sample_pdf(const float* bins, const float* weights, int64_t n, int nb, int ns, int det,
                    const float* u, float* samples, nerf_stream_t stream);

/* render_rays' resampling step: z_mid = .5(z[1:]+z[:-1]); z_samples = sample_pdf(z_mid,
 * weights[:,1:-1], Si, det); z_out = sort(concat(z_vals, z_samples)); z_std = std(z_samples).
 * z_vals, weights: [n,Sc] (z_vals ascending); z_out: [n,Sc+Si]; z_samples [n,Si] and z_std [n]
 * may be NULL.  With det == 0, u [n,Si] must be given; the kernel sorts the samples itself. */
int nerf_resample_merge(const float* z_vals, const float* weights, int64_t n, int Sc, int Si,
                        int det, const float* u, float* z_out, float* z_samples, float* z_std,
                        nerf_stream_t stream);

/* ---- the whole per-ray core in one call ----------------------------------------------------- */

/* Bytes of workspace nerf_render_rays needs (depths, network outputs and weights of both passes). */
size_t nerf_render_rays_workspace_bytes(int64_t n_rays, int N_samples, int N_importance);

/* render_rays (NERFCO.nerf_renderer.render_rays, source absent from the reference tree; call site
 * run_nerf.py:20 inside batchify_rays, keyword arguments run_nerf.py:216-238) at
 * raw_noise_std = 0: stratified depths (perturb_u [n,N_samples] uniforms or NULL = no jitter),
 * coarse network, raw2outputs, sample_pdf on the inner weights + sorted merge (pdf_u
 * [n,N_importance] uniforms or NULL = deterministic, i.e. perturb == 0), fine network
 * (packed_fine, or packed_coarse when NULL), raw2outputs.  rays: [n,width>=11] ray batch as
 * render() assembles it.  Outputs: rgb_map [n,3] (required), disp_map / acc_map / depth_map [n],
 * and with N_importance > 0 the coarse pass's rgb0 [n,3], disp0, acc0 [n] and z_std [n]; any of
 * them may be NULL.  With N_importance == 0 the maps are the coarse pass's and the last four are
 * not written.  Six launches on `stream`, bit-identical to calling the stages one by one. */
int nerf_render_rays(const void* packed_coarse, const void* packed_fine, const float* rays, int width,
                     int64_t n_rays, int N_samples, int N_importance, int lindisp, int white_bkgd,
                     const float* perturb_u, const float* pdf_u, float* rgb_map, float* disp_map,
                     float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0, float* z_std,
                     void* workspace, nerf_stream_t stream);

/* nerf_render_rays without the workspace argument, for callers that cannot allocate device memory
 * and have none to pass: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  The intermediates live in a fixed
 * 64 MiB array of the library's code object (nerf_render_rays_static_bytes()), which every GPU the
 * library runs on holds until the process ends; a call whose
 * nerf_render_rays_workspace_bytes(n_rays, N_samples, N_importance) exceeds it is rejected with
 * NERF_E_UNSUPPORTED before anything is launched (about 12,000 rays at 64 + 128 samples).  The call
 * allocates and frees nothing.  What the HIP runtime does by itself the first time something of a
 * code object is used on a GPU is another matter: it loads that code object there, and the GPU's
 * free memory drops (by the array's size when the array's address is first looked up, by a few MiB
 * when a kernel of another source file is first launched).  A call with n_rays == 0 does exactly
 * that and nothing else, for the array and for every kernel a call launches: callers make it once
 * per GPU before they start capturing or serving, and no later call changes the GPU's free memory.
 * All calls on one GPU share the array: calls that overlap in time on one GPU (different streams
 * or threads) race; calls on one stream, or on different GPUs, do not. */
size_t nerf_render_rays_static_bytes(void);
int nerf_render_rays_static(const void* packed_coarse, const void* packed_fine, const float* rays, int width,
                            int64_t n_rays, int N_samples, int N_importance, int lindisp, int white_bkgd,
                            const float* perturb_u, const float* pdf_u, float* rgb_map, float* disp_map,
                            float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0, float* z_std,
                            nerf_stream_t stream);

/* ---- training: backward of the path (run_nerf.py:770-826, tf.GradientTape over render()) ---- */

/* Gradient of raw2outputs: d_raw [n,S,4] from the gradients of rgb_map [n,3], acc_map [n] and
 * depth_map [n] (any of the three may be NULL = zero).  Same raw/z/rays_d/noise as the forward
 * call.  No gradient flows to z or rays_d (upstream stops it at the resampled depths). */
int nerf_composite_bwd(const float* raw, const float* z, const float* rays_d, int d_stride,
                       int64_t n, int S, int white_bkgd, const float* noise, float noise_std,
                       const float* g_rgb, const float* g_acc, const float* g_depth, float* d_raw,
                       nerf_stream_t stream);

/* Bytes of activation scratch nerf_mlp_forward_train needs for n_points query points. */
size_t nerf_mlp_act_bytes(int64_t n_points);

/* nerf_mlp_forward_rays that also saves the fp16 activations the backward pass needs into `act`
 * (nerf_mlp_act_bytes(n_rays*S) bytes) for nerf_mlp_backward: both encodings, the outputs of the
 * eight point layers and of the view layer, and their ReLU masks.  The bottleneck
 * (run_nerf_helpers.py:103) is linear and is not saved (nerf_mlp_backward composes its gradient). */
int nerf_mlp_forward_train(const void* packed, const float* rays, int width, const float* z,
                           int64_t n_rays, int S, float* raw, void* act, nerf_stream_t stream);

/* Transposed fp16 weight stream for the backward chain (nerf_mlp_packed_bwd_bytes() bytes),
 * from the same 24 parameter arrays as nerf_mlp_pack. */
size_t nerf_mlp_packed_bwd_bytes(void);
int nerf_mlp_pack_bwd(const float* const* params, void* packed_bwd, nerf_stream_t stream);

/* Where the saved tensors lie, for tools and tests that decode them: both buffers are
 * [32-point wave tile][block][1 KiB].  out[0..9] = blocks per tile of `act`, first block of the
 * point encoding, of h0 (16 blocks per layer output), of the view layer's output, of the view
 * encoding, of the ReLU mask blocks; blocks per tile of `gbuf`, first block of d(view layer
 * pre-activation), of d(pre-activation of layer 0) (16 per layer), of d(alpha).  n: room in out. */
int nerf_mlp_train_layout(int32_t* out, int n);

/* Scratch sizes of nerf_mlp_backward: fp16 layer gradients for n_points, and the fp32
 * partial-sum buffer (fixed size for the current device: a status block, the summed
 * accumulator tiles and one set of partial tiles per workgroup of the weight-gradient kernel). */
size_t nerf_mlp_grad_bytes(int64_t n_points);
size_t nerf_mlp_partial_bytes(void);

/* Status bits of the last nerf_mlp_backward that used a given `partial` buffer. */
#define NERF_GRAD_SATURATED 1 /* a layer gradient reached the fp16 maximum, or was inf / NaN  */
#define NERF_GRAD_UNDERFLOW 2 /* d_raw was non-zero, yet every fp16 layer gradient was denormal */

/* Gradient of the network parameters (tape.gradient(loss, grad_vars), run_nerf.py:825, restricted
 * to one network): d_raw [n_points,4] -> grads: HOST array of 24 device pointers in get_weights()
 * order, each shaped like its parameter.  `act` is what nerf_mlp_forward_train saved for the same
 * points.  The chain runs in fp16 with fp32 accumulation; d_raw is multiplied by a power of two
 * first and the result divided by it.  gscale == 0 (the normal use): the factor is formed on the
 * device from max|d_raw| so that the largest scaled element lies in [256, 512) -- no host
 * synchronisation.  gscale < 0: the same with the target [-gscale, -2 gscale), -gscale a power of
 * two in [2^-14, 2^14] (lower it for networks that amplify gradients by more than the 128x of
 * headroom the default leaves).  gscale > 0: the caller's factor is used as given.  Either way the call records
 * whether a layer gradient saturated the fp16 range (or was not finite) or everything sank below
 * the smallest normal fp16; when either happened the gradients are written as NaN, never as
 * plausible numbers (read the bits with nerf_mlp_backward_status).
 * Weight gradients are summed in a fixed order: the same inputs give the same bits.  The gradients
 * of the bottleneck layer and of the view layer's bottleneck rows are formed as products of
 * sum_points h7 (x) d v_pre with the two layers' fp32 weights (kept behind the stream in packed_bwd).
 * accumulate != 0 adds into grads instead of overwriting.  No gradient flows to rays or depths. */
int nerf_mlp_backward(const void* packed_bwd, const void* act, const float* d_raw, int64_t n_points,
                      float gscale, void* gbuf, void* partial, float* const* grads, int accumulate,
                      nerf_stream_t stream);

/* Reads the status block the last nerf_mlp_backward left in `partial` (enqueued on `stream`
 * behind it; waits for the stream): *flags = NERF_GRAD_* bits, *gscale_used, *absmax_d_raw
 * (either may be NULL).  The reference's counterpart is upstream's tf.debugging.check_numerics on
 * the outputs of render_rays (NERFCO.nerf_renderer, source absent from the reference tree). */
int nerf_mlp_backward_status(const void* partial, int* flags, float* gscale_used, float* absmax_d_raw,
                             nerf_stream_t stream);

/* img2mse and its gradient (run_nerf_helpers.py:11; run_nerf.py:781,:809): adds
 * sum((x-y)^2)*inv_count to *loss (which the caller zeroes; may be NULL) and writes
 * d_x = 2(x-y)*inv_count*upstream (may be NULL).  count = number of scalars. */
int nerf_mse_loss_fwd_bwd(const float* x, const float* y, int64_t count, float inv_count,
                          float upstream, float* loss, float* d_x, nerf_stream_t stream);

/* The loss of the plain photometric configuration and everything the step derives from it, in
 * one launch (run_nerf.py:781 img_loss, :797 psnr, :809-811 img_loss0, psnr0, loss += img_loss0;
 * run_nerf_helpers.py:11-12 img2mse, mse2psnr).  rgb, rgb0 (NULL when there is no coarse map) and
 * target hold `count` scalars each.  out[0..4] = loss (img_loss + img_loss0), img_loss, img_loss0,
 * psnr, psnr0 (overwritten, nothing to zero; out[2] = out[4] = 0 without rgb0);
 * d_rgb = grad_weight * d loss / d rgb, d_rgb0 likewise (either may be NULL).  grad_weight is 1 for
 * a single process; a data-parallel rank passes its ray count (see the gradient exchange of
 * nerf_mlp_backward's callers: gradients are summed over the ranks and divided by the summed count).
 * One workgroup does it all (deterministic sums, no atomics): meant for a step's ray batch, a few
 * thousand to a few hundred thousand scalars; nerf_mse_loss_fwd_bwd spreads larger inputs over the chip. */
int nerf_photometric_loss(const float* rgb, const float* rgb0, const float* target, int64_t count,
                          float grad_weight, float* out, float* d_rgb, float* d_rgb0, nerf_stream_t stream);

/* tf.keras.optimizers.Adam dense update (run_nerf.py:605,:826) over a flat buffer of n scalars:
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; theta -= lr_t m/(sqrt(v)+eps), where the caller
 * passes lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (eps is not bias-corrected, as in Keras).  Elements whose
 * gradient is NaN or infinite are skipped (parameter and moments unchanged): nerf_mlp_backward
 * writes NaN gradients when its fp16 chain saturated, and such a step must not reach the weights. */
int nerf_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, float lr_t,
                   float beta1, float beta2, float eps, nerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
