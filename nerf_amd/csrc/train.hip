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

NERF_API int nerf_adam_step(float* theta, const float* g, float* m, float* v, int64_t n, float lr_t,
                            float beta1, float beta2, float eps, nerf_stream_t stream) {
    NERF_REQUIRE(theta && g && m && v && n >= 0, "nerf_adam_step: bad argument");
    if (n == 0) return 0;
    adam_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(theta, g, m, v, n, lr_t,
                                                                             beta1, beta2, eps);
    return check_launch("nerf_adam_step");
}
