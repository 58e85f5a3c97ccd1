// Re-lays the 24 fp32 parameter arrays of one NeRF network (get_weights() order) into the fp16
// MFMA fragment stream + fp32 bias tiles of mlp_layout_wide.h, which both forward kernels
// (inference and the activation-saving training forward) read.  One 64-thread block per
// fragment / bias tile; runs once per weight update (1.2 MB out), so it is written for clarity.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_layout_wide.h"

namespace nerf {
using namespace layout;

struct ParamPtrs {
    const float* p[24];
};

// get_weights() slots (kernel at 2*i, bias at 2*i+1)
constexpr int P_BOTT = 8, P_VIEW = 9, P_RGB = 10, P_ALPHA = 11;

// Fragment F = (output tile of 16 rows, k-step of 32 inputs).
__device__ void pack_wide_fragment(const ParamPtrs& pp, char* __restrict__ packed, int F) {
    using namespace lw;
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    if (F >= lw::F_END) {
        const int T = F - lw::F_END;  // bias tile: 16 floats in row order
        if (lane >= 16) return;
        const float* src;
        int off, count = 16;
        if (T < T_L1) src = pp.p[1], off = 16 * T;
        else if (T < T_L5) src = pp.p[2 * (1 + (T - T_L1) / 16) + 1], off = 16 * ((T - T_L1) % 16);
        else if (T < T_L6) src = pp.p[11], off = 16 * (T - T_L5);
        else if (T < T_ALPHA) src = pp.p[2 * (6 + (T - T_L6) / 16) + 1], off = 16 * ((T - T_L6) % 16);
        else if (T == T_ALPHA) src = pp.p[2 * P_ALPHA + 1], off = 0, count = 1;
        else if (T < T_VIEW) src = pp.p[2 * P_BOTT + 1], off = 16 * (T - T_BOTT);
        else if (T < T_RGB) src = pp.p[2 * P_VIEW + 1], off = 16 * (T - T_VIEW);
        else src = pp.p[2 * P_RGB + 1], off = 0, count = 3;
        float* dst = (float*)(packed + lw::kBiasOffset) + 16 * T;
        dst[lane] = (lane < count) ? src[off + lane] : 0.f;
        return;
    }
    const float* Wm = nullptr;  // kernel [in, out], row-major
    int out_dim = 0;            // columns of Wm
    int oc = -1;                // output column of this lane's row, or -1
    int rows[8];                // input row per element, or -1
    for (int j = 0; j < 8; ++j) rows[j] = -1;
    if (F < lw::F_L1) {
        const int to = F / 2, s = F % 2;
        Wm = pp.p[0], out_dim = 256, oc = 16 * to + r;
        for (int j = 0; j < 8; ++j) rows[j] = lw::pe_feature(g, 8 * s + j);
    } else if (F < lw::F_L5 || (F >= lw::F_L6 && F < lw::F_ALPHA)) {
        const int base = (F < lw::F_L5) ? lw::F_L1 : lw::F_L6, l0 = (F < lw::F_L5) ? 1 : 6;
        const int l = l0 + (F - base) / 128, q = (F - base) % 128, to = q / 8, s = q % 8;
        Wm = pp.p[2 * l], out_dim = 256, oc = 16 * to + r;
        for (int j = 0; j < 8; ++j) rows[j] = lw::acc_feature(s, g, j);
    } else if (F < lw::F_L6) {
        const int q = F - lw::F_L5, to = q / 10, s = q % 10;
        Wm = pp.p[10], out_dim = 256, oc = 16 * to + r;
        for (int j = 0; j < 8; ++j)
            rows[j] = (s < 2) ? lw::pe_feature(g, 8 * s + j) : 63 + lw::acc_feature(s - 2, g, j);
    } else if (F < lw::F_BOTT) {
        const int s = F - lw::F_ALPHA;
        Wm = pp.p[2 * P_ALPHA], out_dim = 1, oc = (r == 0) ? 0 : -1;
        for (int j = 0; j < 8; ++j) rows[j] = lw::acc_feature(s, g, j);
    } else if (F < lw::F_VIEW) {
        const int q = F - lw::F_BOTT, to = q / 8, s = q % 8;
        Wm = pp.p[2 * P_BOTT], out_dim = 256, oc = 16 * to + r;
        for (int j = 0; j < 8; ++j) rows[j] = lw::acc_feature(s, g, j);
    } else if (F < lw::F_RGB) {
        const int q = F - lw::F_VIEW, to = q / 9, s = q % 9;
        Wm = pp.p[2 * P_VIEW], out_dim = 128, oc = 16 * to + r;
        for (int j = 0; j < 8; ++j) {
            if (s < 8) rows[j] = lw::acc_feature(s, g, j);
            else {
                const int f = lw::view_feature(g, j);
                rows[j] = (f < 0) ? -1 : 256 + f;
            }
        }
    } else if (F < lw::F_USED) {
        const int s = F - lw::F_RGB;
        Wm = pp.p[2 * P_RGB], out_dim = 3, oc = (r < 3) ? r : -1;
        for (int j = 0; j < 8; ++j) rows[j] = lw::acc_feature(s, g, j);
    }
    _Float16* dst = (_Float16*)(packed + (size_t)F * kFragBytes) + lane * 8;
    for (int j = 0; j < 8; ++j) {
        const float v = (Wm && oc >= 0 && rows[j] >= 0) ? Wm[(size_t)rows[j] * out_dim + oc] : 0.f;
        dst[j] = (_Float16)v;
    }
}

constexpr int kWideBlocks = lw::F_END + lw::T_END;

__global__ __launch_bounds__(64) void mlp_pack_kernel(ParamPtrs pp, char* __restrict__ packed) {
    pack_wide_fragment(pp, packed, blockIdx.x);
}

}  // namespace nerf

using namespace nerf;

NERF_API size_t nerf_mlp_packed_bytes(void) { return lw::kPackedBytes; }

NERF_API int nerf_mlp_pack(const float* const* params, void* packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack: null pointer");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0, "nerf_mlp_pack: packed must be 16-byte aligned");
    ParamPtrs pp;
    for (int i = 0; i < 24; ++i) {
        NERF_REQUIRE(params[i], "nerf_mlp_pack: params[%d] is null", i);
        pp.p[i] = params[i];
    }
    mlp_pack_kernel<<<kWideBlocks, 64, 0, as_stream(stream)>>>(pp, (char*)packed);
    return check_launch("nerf_mlp_pack");
}
