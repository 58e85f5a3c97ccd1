// Training-step element-wise kernels: the photometric loss with its gradient, and the Keras Adam
// update over a flat parameter buffer.
#include "common.h"

namespace nerf {

// img2mse (run_nerf_helpers.py:11) and its gradient in one pass:
//   loss += sum((x - y)^2) * inv_count,   d_x = 2 (x - y) * inv_count * upstream
// inv_count = 1/(number of elements of the GLOBAL batch), so that ray-sharded ranks sum to the
// global mean.  One atomic per workgroup into *loss.
__global__ __launch_bounds__(256) void mse_fwd_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ y, int64_t count,
                                                          float inv_count, float upstream,
                                                          float* __restrict__ loss,
                                                          float* __restrict__ d_x) {
    __shared__ float part[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        s += d * d;
        if (d_x) d_x[i] = 2.0f * d * inv_count * upstream;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, ((part[0] + part[1]) + (part[2] + part[3])) * inv_count);
}

// The plain photometric configuration of an optimisation step in one launch (run_nerf.py:781, :797,
// :809-811; run_nerf_helpers.py:11-12): img_loss = mean((rgb - target)^2), img_loss0 the same for
// the coarse map, loss = img_loss + img_loss0, psnr = -10 ln(img_loss) / ln 10, and the gradients
// of the loss with respect to the two maps, times grad_weight.  One workgroup: a ray batch is a
// few thousand scalars, a launch costs more than the sums, and this way the sums need neither
// atomics (the same inputs give the same bits) nor a result zeroed beforehand.
constexpr int kLossThreads = 1024;
__global__ __launch_bounds__(kLossThreads) void photometric_loss_kernel(
    const float* __restrict__ rgb, const float* __restrict__ rgb0, const float* __restrict__ target, int64_t count,
    float inv_count, float grad_weight, float* __restrict__ out, float* __restrict__ d_rgb, float* __restrict__ d_rgb0) {
    __shared__ float part[2][kLossThreads / 64];
    float s = 0.f, s0 = 0.f;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < count; i += kLossThreads) {
        const float t = target[i];
        const float d = rgb[i] - t;
        s += d * d;
        if (d_rgb) d_rgb[i] = 2.0f * d * inv_count * grad_weight;
        if (rgb0) {
            const float d0 = rgb0[i] - t;
            s0 += d0 * d0;
            if (d_rgb0) d_rgb0[i] = 2.0f * d0 * inv_count * grad_weight;
        }
    }
    s = wave_sum(s), s0 = wave_sum(s0);
    if ((threadIdx.x & 63) == 0) part[0][threadIdx.x >> 6] = s, part[1][threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, a0 = 0.f;
        for (int w = 0; w < kLossThreads / 64; ++w) a += part[0][w], a0 += part[1][w];
        const float img_loss = a * inv_count, img_loss0 = rgb0 ? a0 * inv_count : 0.f;
        out[0] = img_loss + img_loss0;
        out[1] = img_loss;
        out[2] = img_loss0;
        out[3] = -10.f * logf(img_loss) / logf(10.f);
        out[4] = rgb0 ? -10.f * logf(img_loss0) / logf(10.f) : 0.f;
    }
}

// tf.keras.optimizers.Adam, dense update (non-amsgrad), with lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
// formed on the host:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  theta -= lr_t m/(sqrt(v)+eps).
// An element whose gradient is not finite is left as it is, moments included: the backward pass
// writes NaN for a whole network when its fp16 gradient chain saturated, and that step is then
// skipped for the network instead of putting NaN into its weights and moments for good.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr_t, float b1, float b2, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    if (!(fabsf(gi) <= 3.0e38f)) return;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    theta[i] = theta[i] - lr_t * mi / (sqrtf(vi) + eps);
}

}  // namespace nerf

using namespace nerf;

NERF_API int nerf_mse_loss_fwd_bwd(const float* x, const float* y, int64_t count, float inv_count,
                                   float upstream, float* loss, float* d_x, nerf_stream_t stream) {
    NERF_REQUIRE(x && y && count >= 0, "nerf_mse_loss_fwd_bwd: bad argument");
    NERF_REQUIRE(loss || d_x, "nerf_mse_loss_fwd_bwd: nothing to compute");
    if (count == 0) return 0;
    const int64_t want = (count + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1024 ? want : 1024);
    mse_fwd_bwd_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, y, count, inv_count, upstream, loss, d_x);
    return check_launch("nerf_mse_loss_fwd_bwd");
}

NERF_API int nerf_photometric_loss(const float* rgb, const float* rgb0, const float* target, int64_t count,
                                   float grad_weight, float* out, float* d_rgb, float* d_rgb0, nerf_stream_t stream) {
    NERF_REQUIRE(rgb && target && out && count > 0, "nerf_photometric_loss: bad argument");
    NERF_REQUIRE(rgb0 || !d_rgb0, "nerf_photometric_loss: d_rgb0 without rgb0");
    NERF_REQUIRE(grad_weight == grad_weight, "nerf_photometric_loss: grad_weight is NaN");
    photometric_loss_kernel<<<1, kLossThreads, 0, as_stream(stream)>>>(rgb, rgb0, target, count, 1.0f / (float)count,
                                                                       grad_weight, out, d_rgb, d_rgb0);
    return check_launch("nerf_photometric_loss");
}

NERF_API int nerf_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, float lr_t,
                            float beta1, float beta2, float eps, nerf_stream_t stream) {
    NERF_REQUIRE(theta && g && m && v && n >= 0, "nerf_adam_step: bad argument");
    if (n == 0) return 0;
    adam_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(theta, g, m, v, n, lr_t,
                                                                             beta1, beta2, eps);
    return check_launch("nerf_adam_step");
}
