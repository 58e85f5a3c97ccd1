// Re-lays the 24 fp32 parameter arrays of one NeRF network (get_weights() order) into the fp16
// MFMA fragment stream + fp32 bias tiles described in mlp_layout.h.  One 64-thread block per
// fragment / bias tile; runs once per weight update (1.2 MB out), so it is written for clarity.
#include "common.h"
#include "mlp_layout.h"

namespace nerf {
using namespace layout;

struct ParamPtrs {
    const float* p[24];
};

// get_weights() slots (kernel at 2*i, bias at 2*i+1)
constexpr int P_BOTT = 8, P_VIEW = 9, P_RGB = 10, P_ALPHA = 11;

__global__ __launch_bounds__(64) void mlp_pack_kernel(ParamPtrs pp, char* __restrict__ packed) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int F = blockIdx.x;
    if (F >= F_END) {
        // bias tiles
        const int T = F - F_END;
        if (lane >= 32) return;
        const float* src;
        int off, count = 32;
        if (T < BT_L1) src = pp.p[1], off = 32 * T;
        else if (T < BT_L5) src = pp.p[2 * (1 + (T - BT_L1) / 8) + 1], off = 32 * ((T - BT_L1) % 8);
        else if (T < BT_L6) src = pp.p[11], off = 32 * (T - BT_L5);
        else if (T < BT_ALPHA) src = pp.p[2 * (6 + (T - BT_L6) / 8) + 1], off = 32 * ((T - BT_L6) % 8);
        else if (T == BT_ALPHA) src = pp.p[2 * P_ALPHA + 1], off = 0, count = 1;
        else if (T < BT_VIEW) src = pp.p[2 * P_BOTT + 1], off = 32 * (T - BT_BOTT);
        else if (T < BT_RGB) src = pp.p[2 * P_VIEW + 1], off = 32 * (T - BT_VIEW);
        else src = pp.p[2 * P_RGB + 1], off = 0, count = 3;
        // accumulator order: slot (h, reg) holds output row (reg&3) + 8(reg>>2) + 4h
        const int hh = lane >> 4, reg = lane & 15;
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        float* dst = (float*)(packed + kBiasOffset) + 32 * T;
        dst[lane] = (row < count) ? src[off + row] : 0.f;
        return;
    }

    // weight fragment F: lane (r = output row in the tile, h = k half), 8 consecutive k elements
    const float* Wm;   // kernel [in, out], row-major
    int out_dim;       // columns of Wm
    int oc;            // output column this lane's row maps to, or -1
    int rows[8];       // input row per element, or -1
    if (F < F_L1) {
        const int to = F / 4, k = F % 4;
        Wm = pp.p[0], out_dim = 256, oc = 32 * to + r;
        for (int j = 0; j < 8; ++j) rows[j] = pe_feature(h, 8 * k + j);
    } else if (F < F_L5 || (F >= F_L6 && F < F_HEADS)) {
        const int base = (F < F_L5) ? F_L1 : F_L6, l0 = (F < F_L5) ? 1 : 6;
        const int l = l0 + (F - base) / 128, q = (F - base) % 128, to = q / 16, k = q % 16;
        Wm = pp.p[2 * l], out_dim = 256, oc = 32 * to + r;
        for (int j = 0; j < 8; ++j) rows[j] = acc_feature(k, h, j);
    } else if (F < F_L6) {
        const int q = F - F_L5, to = q / 20, k = q % 20;
        Wm = pp.p[10], out_dim = 256, oc = 32 * to + r;
        for (int j = 0; j < 8; ++j)
            rows[j] = (k < 4) ? pe_feature(h, 8 * k + j) : 63 + acc_feature(k - 4, h, j);
    } else if (F < F_VIEW) {
        const int q = F - F_HEADS, tile = q / 16, k = q % 16;
        if (tile == 0) Wm = pp.p[2 * P_ALPHA], out_dim = 1, oc = (r == 0) ? 0 : -1;
        else Wm = pp.p[2 * P_BOTT], out_dim = 256, oc = 32 * (tile - 1) + r;
        for (int j = 0; j < 8; ++j) rows[j] = acc_feature(k, h, j);
    } else if (F < F_RGB) {
        const int q = F - F_VIEW, to = q / 18, k = q % 18;
        Wm = pp.p[2 * P_VIEW], out_dim = 128, oc = 32 * to + r;
        for (int j = 0; j < 8; ++j) {
            if (k < 16) rows[j] = acc_feature(k, h, j);
            else {
                const int f = view_feature(h, 8 * (k - 16) + j);
                rows[j] = (f < 0) ? -1 : 256 + f;
            }
        }
    } else {
        const int k = F - F_RGB;
        Wm = pp.p[2 * P_RGB], out_dim = 3, oc = (r < 3) ? r : -1;
        for (int j = 0; j < 8; ++j) rows[j] = acc_feature(k, h, j);
    }
    _Float16* dst = (_Float16*)(packed + (size_t)F * kFragBytes) + lane * 8;
    for (int j = 0; j < 8; ++j) {
        const float v = (oc >= 0 && rows[j] >= 0) ? Wm[(size_t)rows[j] * out_dim + oc] : 0.f;
        dst[j] = (_Float16)v;
    }
}

}  // namespace nerf

using namespace nerf;

NERF_API size_t nerf_mlp_packed_bytes(void) { return layout::kPackedBytes; }

NERF_API int nerf_mlp_pack(const float* const* params, void* packed, nerf_stream_t stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_pack: null pointer");
    NERF_REQUIRE(((uintptr_t)packed & 15) == 0, "nerf_mlp_pack: packed must be 16-byte aligned");
    ParamPtrs pp;
    for (int i = 0; i < 24; ++i) {
        NERF_REQUIRE(params[i], "nerf_mlp_pack: params[%d] is null", i);
        pp.p[i] = params[i];
    }
    mlp_pack_kernel<<<layout::F_END + layout::BT_END, 64, 0, as_stream(stream)>>>(pp, (char*)packed);
    return check_launch("nerf_mlp_pack");
}
