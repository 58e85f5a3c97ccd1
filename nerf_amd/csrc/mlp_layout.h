// Layout of one packed NeRF network (D=8, W=256, 63+27 inputs, skip after layer 4, view head)
// as the forward kernel consumes it.
//
// Orientation.  Every layer is computed transposed, Y^T[out][point] = W^T[out][in] . X^T[in][point],
// with v_mfma_f32_32x32x16_f16: A = a 32(out) x 16(in) block of W^T, B = 16(in) x 32(points) of
// X^T, D = 32(out) x 32(points) in fp32.  D keeps the point on the lane (column) and the 32 output
// features in the lane's 16 registers (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), which is exactly
// the shape the next layer's B operand needs: registers 8s..8s+7 of an accumulator tile, converted
// to fp16, ARE the B fragment of k-step s.  Activations therefore never leave registers.  The
// k order inside such a fragment is permuted (element j of lane half h is feature
// 16s + 8(j>>2) + 4h + (j&3) of the tile); the weights are packed with the same permutation.
//
// Stream.  The A operands of the whole network are stored as 1 KiB "fragments" (64 lanes x 8 fp16,
// lane-linear so a wave reads one with a single conflict-free ds_read_b128) in the order the
// kernel consumes them, 16 fragments per 16 KiB chunk, followed by the fp32 biases, 32 per
// output tile stored in accumulator order: [tile][lane half h][register r] = bias of output row
// (r&3) + 8(r>>2) + 4h of the tile.  1184 fragments = 74 chunks.
#pragma once
#include <stddef.h>

namespace nerf {
namespace layout {

constexpr int kFragBytes = 1024;
constexpr int kFragsPerChunk = 16;
constexpr int kChunkBytes = kFragBytes * kFragsPerChunk;

// first fragment of each layer: (k-steps per output tile) x (output tiles)
constexpr int F_L0 = 0;        //  4 x 8   PE(64 slots)            -> 256, relu
constexpr int F_L1 = 32;       // 16 x 8   x4 layers (1..4)        -> 256, relu
constexpr int F_L5 = 544;      // 20 x 8   [PE(64) ; h(256)]       -> 256, relu
constexpr int F_L6 = 704;      // 16 x 8   x2 layers (6,7)         -> 256, relu
constexpr int F_HEADS = 960;   // 16 x 9   tile 0 = alpha (row 0), tiles 1..8 = bottleneck, linear
constexpr int F_VIEW = 1104;   // 18 x 4   [bottleneck(256) ; viewPE(32 slots)] -> 128, relu
constexpr int F_RGB = 1176;    //  8 x 1   128 -> rgb (rows 0..2), linear
constexpr int F_END = 1184;
constexpr int kChunks = F_END / kFragsPerChunk;  // 74
static_assert(F_END % kFragsPerChunk == 0, "stream must end on a chunk boundary");

// first bias tile (32 floats) of each layer
constexpr int BT_L0 = 0, BT_L1 = 8, BT_L5 = 40, BT_L6 = 48, BT_ALPHA = 64, BT_BOTT = 65,
              BT_VIEW = 73, BT_RGB = 77, BT_END = 78;

constexpr size_t kBiasOffset = (size_t)kChunks * kChunkBytes;
constexpr int kBiasBytes = 10 * kFragBytes;  // 78 tiles x 128 B = 9984 B, padded to whole KiB
static_assert(BT_END * 128 <= kBiasBytes, "bias region");
constexpr size_t kPackedBytes = kBiasOffset + kBiasBytes;

// ---- training: activations saved by the forward pass, gradients saved by the backward chain ----
// Both are stored as the 1 KiB B fragments the kernels already hold (lane = point within a
// 32-point wave tile, 8 fp16 features per lane), [wave tile][block], so a wave writes or reads a
// block as one coalesced 1 KiB transaction.
constexpr int A_PE = 0;      //  4 blocks: point encoding (64 slots)
constexpr int A_H0 = 4;      // 16 blocks per layer output h0..h7 (after ReLU): A_H0 + 16*l
constexpr int A_BOTT = 132;  // 16 blocks: bottleneck (linear)
constexpr int A_V = 148;     //  8 blocks: view layer output (after ReLU)
constexpr int A_VPE = 156;   //  2 blocks: view-direction encoding (32 slots)
constexpr int A_MASK = 158;  //  1 block per layer output h0..h7 and one for v (A_MASK + 8): ReLU masks, one
                             //  bit per unit: lane (point, half) holds 4 dwords; accumulator tile t of the
                             //  layer owns halfword t&1 of dword t>>1, bit r = register r of the tile > 0
constexpr int kActBlocks = 167;

constexpr int G_DV = 0;      //  8 blocks: d(view layer pre-activation)
constexpr int G_DBOTT = 8;   // 16 blocks: d(bottleneck)
constexpr int G_DPRE0 = 24;  // 16 blocks per layer: d(pre-activation of layer l): G_DPRE0 + 16*l
constexpr int G_DA = 152;    //  1 block: d(alpha) in feature slot 0
constexpr int G_DC = 153;    //  1 block: d(rgb) in feature slots 0..2
constexpr int kGradBlocks = 154;

// ---- backward weight stream: d h_in = W . d y, A = W[in][out] with the out index permuted as
// the incoming gradient fragments are ----
constexpr int B_V = 0;       //  1 x 4   d rgb(16 slots)            -> d v (128)
constexpr int B_BOTT = 4;    //  8 x 8   d v_pre(128)               -> d bottleneck (256)
constexpr int B_H7 = 68;     // 17 x 8   [d bott(256) ; d alpha(16)] -> d h7
constexpr int B_H6 = 204;    // 16 x 8   x7: d pre_{l+1} -> d h_l for l = 6..0 (layer 5: rows 63.. of its kernel)
constexpr int B_END_USED = 1100;
constexpr int B_END = 1104;  // padded to a chunk boundary
constexpr int kBwdChunks = B_END / kFragsPerChunk;  // 69
constexpr size_t kPackedBwdBytes = (size_t)kBwdChunks * kChunkBytes;
static_assert(B_H6 + 7 * 128 == B_END_USED && B_END % kFragsPerChunk == 0, "backward stream");

// Feature of the previous layer's output held by element j of lane half h in k-step k
// (k-step k = accumulator tile k>>1, register group k&1).
__host__ __device__ constexpr int acc_feature(int k, int h, int j) {
    return 32 * (k >> 1) + 16 * (k & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}

// Point positional encoding, 63 features in 64 slots: lane half h computes 32 of them in local
// order i = 8*kstep + j: [e0, e1, then for a = 0..4: sin(f x), sin(f y), sin(f z), cos(f x),
// cos(f y), cos(f z) with f = 2^(5h+a)], where (e0,e1) = (x,y) for h=0 and (z,pad) for h=1.
// Returns the index into the reference's embedding (run_nerf_helpers.py:46-56) or -1 for the pad.
__host__ __device__ constexpr int pe_feature(int h, int i) {
    if (i == 0) return h ? 2 : 0;
    if (i == 1) return h ? -1 : 1;
    return 3 + 6 * (5 * h + (i - 2) / 6) + (i - 2) % 6;
}

// View-direction encoding, 27 features in 32 slots: lane half h computes 16 local slots:
// [e0, e1, then for a = 0..1 the six sin/cos with f = 2^(2h+a), pad, pad].
__host__ __device__ constexpr int view_feature(int h, int i) {
    if (i == 0) return h ? 2 : 0;
    if (i == 1) return h ? -1 : 1;
    if (i >= 14) return -1;
    return 3 + 6 * (2 * h + (i - 2) / 6) + (i - 2) % 6;
}

}  // namespace layout
}  // namespace nerf
