// Layout of one packed NeRF network as the INFERENCE kernel consumes it: the same transposed
// orientation as mlp_layout.h, tiled for v_mfma_f32_16x16x32_f16 (on MI355X the chip holds a
// higher clock on this shape than on 32x32x16, which is what bounds an MFMA-dense kernel).
//
// Orientation.  Y^T[out][point] = W^T[out][in] . X^T[in][point] with A = a 16(out) x 32(in) block
// of W^T (lane l: row l&15, k = 8(l>>4) .. +7), B = 32(in) x 16(points) of X^T (lane l: point
// l&15, the same k), D = 16(out) x 16(points) in fp32 (lane l: point l&15, rows 4(l>>4) + r in
// its 4 registers).  Two finished output tiles 2s, 2s+1, converted to fp16, ARE the B fragment of
// k-step s of the next layer: element j of lane group g is feature 32s + 16(j>>2) + 4g + (j&3).
// A wave owns two (training forward) or three (inference) point tiles and uses every A fragment
// for all of them.
//
// Stream.  1 KiB fragments (64 lanes x 8 fp16, lane-linear) in consumption order, 16 per 16 KiB
// chunk, then the fp32 biases, 16 per output tile in natural row order.
#pragma once
#include <stddef.h>
#include "mlp_layout.h"

namespace nerf {
namespace lw {

using layout::kFragBytes;
using layout::kFragsPerChunk;
using layout::kChunkBytes;

// first fragment of each layer: (k-steps of 32 per output tile) x (output tiles of 16)
constexpr int F_L0 = 0;       //  2 x 16   PE(64 slots)              -> 256, relu
constexpr int F_L1 = 32;      //  8 x 16   x4 layers (1..4)          -> 256, relu
constexpr int F_L5 = 544;     // 10 x 16   [PE(64) ; h(256)]         -> 256, relu
constexpr int F_L6 = 704;     //  8 x 16   x2 layers (6,7)           -> 256, relu
constexpr int F_ALPHA = 960;  //  8 x 1    alpha (row 0), linear
constexpr int F_BOTT = 968;   //  8 x 16   bottleneck, linear
constexpr int F_VIEW = 1096;  //  9 x 8    [bottleneck(256) ; viewPE(32 slots)] -> 128, relu
constexpr int F_RGB = 1168;   //  4 x 1    128 -> rgb (rows 0..2), linear
constexpr int F_USED = 1172;
constexpr int F_END = 1184;   // padded to a chunk boundary
constexpr int kChunks = F_END / kFragsPerChunk;  // 74
static_assert(F_END % kFragsPerChunk == 0 && F_END >= F_USED, "stream must end on a chunk boundary");

// first bias tile (16 floats) of each layer; tiles are consumed in this order
constexpr int T_L0 = 0, T_L1 = 16, T_L5 = 80, T_L6 = 96, T_ALPHA = 128, T_BOTT = 129, T_VIEW = 145,
              T_RGB = 153, T_END = 154;

constexpr size_t kBiasOffset = (size_t)kChunks * kChunkBytes;
constexpr int kBiasBytes = 10 * kFragBytes;  // 154 tiles x 64 B = 9856 B, padded to whole KiB
static_assert(T_END * 64 <= kBiasBytes, "bias region");
constexpr size_t kPackedBytes = kBiasOffset + kBiasBytes;

// Feature of the previous layer's output held by element j of lane group g in k-step s.
__host__ __device__ constexpr int acc_feature(int s, int g, int j) {
    return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
}

// Point positional encoding, 63 features in 64 slots.  Lane group g computes 16 of them, local
// order i = 8*kstep + j, from three consecutive octaves starting at f = 1, 4, 32, 128:
//   even g: [e, sin f, cos f, sin 2f, cos 2f, sin 4f]      odd g: [e, cos f, sin 2f, cos 2f, sin 4f, cos 4f]
// (each sin/cos entry is the (x, y, z) triple; e = x, y, z, pad for g = 0..3).  Returns the index
// into the reference's embedding (run_nerf_helpers.py:46-56) or -1 for the pad.
__host__ __device__ constexpr int pe_feature(int g, int i) {
    if (i == 0) return g < 3 ? g : -1;
    const int first_octave = g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 5 : 7;
    const int t = (i - 1) / 3 + (g & 1);  // triple number counted from "sin f"
    return 3 + 6 * (first_octave + t / 2) + 3 * (t & 1) + (i - 1) % 3;
}

// View-direction encoding, 27 features in 32 slots: lane group g computes 8 local slots
// [e, sin f (x,y,z), cos f (x,y,z), pad] with f = 2^g.
__host__ __device__ constexpr int view_feature(int g, int j) {
    if (j == 0) return g < 3 ? g : -1;
    if (j == 7) return -1;
    return 3 + 6 * g + (j - 1);
}

}  // namespace lw
}  // namespace nerf
