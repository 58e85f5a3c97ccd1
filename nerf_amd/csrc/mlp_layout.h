// Layouts shared by the training kernels of one NeRF network (D=8, W=256, 63+27 inputs, skip
// after layer 4, view head): the blocks of saved activations and layer gradients, and the
// backward weight stream of the gradient chain.  (The forward weight stream is mlp_layout_wide.h.)
//
// Orientation of the gradient chain.  d X^T[in][point] = W[in][out] . d Y^T[out][point] with
// v_mfma_f32_32x32x16_f16: A = a 32(in) x 16(out) block of W, B = 16(out) x 32(points) of d Y^T,
// D = 32(in) x 32(points) in fp32.  D keeps the point on the lane (column) and the 32 features in
// the lane's 16 registers (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), which is exactly the shape
// the next step's B operand needs: registers 8s..8s+7 of an accumulator tile, converted to fp16,
// ARE the B fragment of k-step s.  Gradients therefore never leave registers.  The k order inside
// such a fragment is permuted (element j of lane half h is feature 16s + 8(j>>2) + 4h + (j&3) of
// the tile); the weights are packed with the same permutation.
//
// Streams are stored as 1 KiB "fragments" (64 lanes x 8 fp16, lane-linear so a wave reads one
// with a single conflict-free ds_read_b128) in the order the kernel consumes them, 16 fragments
// per 16 KiB chunk.
#pragma once
#include <stddef.h>

namespace nerf {
namespace layout {

constexpr int kFragBytes = 1024;
constexpr int kFragsPerChunk = 16;
constexpr int kChunkBytes = kFragBytes * kFragsPerChunk;

// ---- training: activations saved by the forward pass, gradients saved by the backward chain ----
// Both are stored as the 1 KiB B fragments the kernels already hold, [wave tile of 32 points][block],
// so a wave writes or reads a block as one coalesced 1 KiB transaction.
// Activations come from the 16x16x32 forward (mlp_layout_wide.h): a block is k-step s (32
// features) of point tile pt (16 points), block = first + 2 s + pt; lane = point + 16 g holds the
// 8 features lw::acc_feature(s, g, 0..7) (encodings: lw::pe_feature / lw::view_feature).
constexpr int A_PE = 0;      //  4 blocks: point encoding (64 slots)
constexpr int A_H0 = 4;      // 16 blocks per layer output h0..h7 (after ReLU): A_H0 + 16*l
constexpr int A_V = 132;     //  8 blocks: view layer output (after ReLU)
constexpr int A_VPE = 140;   //  2 blocks: view-direction encoding (32 slots)
constexpr int A_MASK = 142;  //  1 block per layer output h0..h7 and one for v (A_MASK + 8): ReLU masks, one
                             //  bit per unit: lane = point + 16 g holds dwords [point tile][2]; with
                             //  i = 2 t + (r >> 1), bit (i & 15) + 16 (r & 1) of dword i >> 4 = unit
                             //  16 t + 4 g + r is > 0
constexpr int kActBlocks = 151;
// The bottleneck (run_nerf_helpers.py:103) is linear, so neither its output nor its gradient is
// saved: with P = sum over points of h7 (x) d v_pre and s = sum of d v_pre, the gradients of the two
// layers around it are products of P with the layers' own weights (see mlp_head_compose_kernel),
//     dW_bott = P . W_view[0:256]^T          db_bott = W_view[0:256] . s
//     dW_view[0:256] = W_bott^T . P + b_bott (x) s
// which takes 32 of 318 blocks out of what the weight-gradient kernel reads per 32 points, 16 out
// of what the forward pass writes and 16 out of what the gradient chain writes.

// Gradients come from the 32x32x16 chain: a block is k-step k (16 features) of the 32 points,
// lane = point + 32 h holds the 8 features acc_feature(k, h, 0..7).
constexpr int G_DV = 0;      //  8 blocks: d(view layer pre-activation)
constexpr int G_DPRE0 = 8;   // 16 blocks per layer: d(pre-activation of layer l): G_DPRE0 + 16*l
constexpr int G_DA = 136;    //  1 block: d(alpha) in feature slot 0
constexpr int G_DC = 137;    //  1 block: d(rgb) in feature slots 0..2
constexpr int kGradBlocks = 138;

// ---- backward weight stream: d h_in = W . d y, A = W[in][out] with the out index permuted as
// the incoming gradient fragments are ----
constexpr int B_V = 0;       //  1 x 4   d rgb(16 slots)            -> d v (128)
constexpr int B_BOTT = 4;    //  8 x 8   d v_pre(128)               -> d bottleneck (256)
constexpr int B_H7 = 68;     // 17 x 8   [d bott(256) ; d alpha(16)] -> d h7
constexpr int B_H6 = 204;    // 16 x 8   x7: d pre_{l+1} -> d h_l for l = 6..0 (layer 5: rows 63.. of its kernel)
constexpr int B_END_USED = 1100;
constexpr int B_END = 1104;  // padded to a chunk boundary
constexpr int kBwdChunks = B_END / kFragsPerChunk;  // 69
// behind the stream, in fp32 and natural order, the three arrays mlp_head_compose_kernel multiplies by:
// the bottleneck's kernel [256][256] and bias [256], and rows 0..255 of the view layer's kernel [256][128]
constexpr size_t kBwdBottW = (size_t)kBwdChunks * kChunkBytes;
constexpr size_t kBwdBottB = kBwdBottW + 256 * 256 * sizeof(float);
constexpr size_t kBwdViewW = kBwdBottB + 256 * sizeof(float);
constexpr size_t kPackedBwdBytes = kBwdViewW + 256 * 128 * sizeof(float);
static_assert(B_H6 + 7 * 128 == B_END_USED && B_END % kFragsPerChunk == 0, "backward stream");

// Feature of the previous layer's output held by element j of lane half h in k-step k
// (k-step k = accumulator tile k>>1, register group k&1).
__host__ __device__ constexpr int acc_feature(int k, int h, int j) {
    return 32 * (k >> 1) + 16 * (k & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}

}  // namespace layout
}  // namespace nerf
