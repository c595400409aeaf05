// One stream wave of the relevancy chain kernels for fp32 slabs (relevancy_chain_groups.hip, relevancy_chain_cols.hip): the head
// reduction  A_bar_l[chunk] = mean_h clamp(G_l[h][chunk] * A_l[h][chunk], 0)  (rule 5: CLIP_explainability.ipynb cell 6:26-29,
// DETR/modules/ExplanationGenerator.py:11-16) over this wave's share of a workgroup's (layer, 64-chunk block) items, handed to a sink.
//
// A wave is on its own -- no barrier, no LDS: it takes the items ws, ws + NWs, ... of the list in layer order (item = layer * NBLK +
// block, a block = 64 four-element chunks, one per lane, `rot` rotates the block order inside a layer) and walks the heads of an item
// IN ASCENDING ORDER (the sequential fp32 sum of the reference) with a register software pipeline of raw buffer loads: two sets of
// 4 heads x 2 arrays = 16 x 16 B per lane in flight, the next batch requested before the current one is reduced.  Every request is
// unconditional (clamped indices, weight 0) so that the compiler keeps the wait counts of the two register sets apart; the buffer
// resources end exactly at the tensor end, so the last partial chunk of an odd N^2 reads zeros, not memory behind the slab.
// Alone on the chip these waves stream at 0.58 of the 8 TB/s peak (profiles/r05_chain_relay_probe.txt); the VGPR file holds ~3x the
// bytes in flight an LDS-DMA ring of the same workgroup could.
#pragma once
#include "mmx_common.h"

#include <type_traits>

namespace mmx {

struct ChainStreamGeom {
    const void* const* attn;   // [layers] probability slabs  [B or 1][H][N][N] fp32
    const void* const* grad;   // [layers] gradient slabs     [B][H][N][N] fp32
    int l0;                    // first layer of this workgroup's list
    int NBLK;                  // 64-chunk blocks per layer
    int rot;                   // block rotation inside a layer (0 .. NBLK - 1)
    int nchunks;               // ceil(N * N / 4)
    int H, NN;                 // heads, N * N
    int b, B;                  // sample, batch
    int64_t attn_bstride;      // H * N * N, or 0: one probability slab shared by the batch
    int nt;                    // nt cache policy on the read-once slabs
    // MMX_CHAIN_CAUSAL (round 6): the probabilities come out of causally masked attention -- exact zeros above the diagonal, so
    // clamp(G * A, 0) is 0 there whatever G holds (finite) -- and a 4-element chunk that lies ENTIRELY above the diagonal is not
    // requested at all: its lanes present an out-of-range buffer offset, which the hardware answers with zeros without touching
    // memory.  Same bits as the full read (0 * g = +-0, and +0 + -0 = +0); about half the bytes of a causal tower.
    int causal = 0;
    int N = 0;                 // tokens (causal only)
    unsigned row_magic = 0;    // ceil(2^32 / N): row of flat element p = umulhi(p, row_magic) for p < N * N (causal only)
};

// sink(layer_in_list, chunk_index, mean) is called once per item by all 64 lanes (chunk_index may be >= nchunks in the last block)
template <typename Sink>
__device__ __forceinline__ void chain_stream_wave(const ChainStreamGeom& gm, int ws, int NWs, int nitems, int lane, Sink&& sink) {
    const int H = gm.H, NN = gm.NN, NBLK = gm.NBLK;
    const int mine = ws < nitems ? (nitems - ws + NWs - 1) / NWs : 0;
    const int HB = (H + 3) >> 2;                       // batches of 4 heads
    const int hstride = NN * 4;
    const float fH = static_cast<float>(H);
    const int64_t sampleG = static_cast<int64_t>(gm.b) * H * NN * 4, sampleA = static_cast<int64_t>(gm.b) * gm.attn_bstride * 4;
    const int64_t restG = static_cast<int64_t>(gm.B - gm.b) * H * NN * 4;
    const int bytesG = static_cast<int>(restG < 0x7fffffff ? restG : 0x7fffffff), bytesA = gm.attn_bstride ? bytesG : H * NN * 4;
    const int total = mine * HB;
    if (total <= 0) return;
    auto item_layer = [&](int i) { return (ws + i * NWs) / NBLK; };
    auto item_chunk = [&](int i) {
        int blk = (ws + i * NWs) % NBLK + gm.rot;
        if (blk >= NBLK) blk -= NBLK;
        return blk * 64 + lane;
    };
    // flat batch sequence k = item * HB + hb
    auto issue = [&](int k, u32x4 (&av)[4], u32x4 (&gv)[4], auto aux_tag) {
        constexpr int AUXG = decltype(aux_tag)::value & 2, AUXA = (decltype(aux_tag)::value & 1) ? 0 : AUXG;
        const int i = k / HB, hb = k - i * HB;
        const int lu = __builtin_amdgcn_readfirstlane(gm.l0 + item_layer(i));
        const auto rA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(gm.attn[lu]) + sampleA)), 0, bytesA, kRawBufferFlags);
        const auto rG = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(gm.grad[lu]) + sampleG)), 0, bytesG, kRawBufferFlags);
        const int cidx = min(item_chunk(i), gm.nchunks - 1);
        unsigned voff = static_cast<unsigned>(cidx) * 16u;
        unsigned skip = 0u;
        if (gm.causal) {                                // elements p .. p + 3 all above the diagonal: first one is, and the chunk does not wrap
            const unsigned p = static_cast<unsigned>(cidx) * 4u;
            const unsigned row = __umulhi(p, gm.row_magic), col = p - row * static_cast<unsigned>(gm.N);
            skip = (col > row && col + 3u < static_cast<unsigned>(gm.N)) ? 0xC0000000u : 0u;   // + 3 GB: past any resource bound (<= 2^31 - 1)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned off = (voff + static_cast<unsigned>(min(hb * 4 + u, H - 1) * hstride)) | skip;
            av[u] = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, AUXA);
            gv[u] = __builtin_amdgcn_raw_buffer_load_b128(rG, off, 0, AUXG);
        }
    };
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](int k, bool live, const u32x4 (&av)[4], const u32x4 (&gv)[4]) {
        const int i = k / HB, hb = k - i * HB;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // heads in ascending order: the sequential sum.  A padding head (H % 4 != 0) or the dead tail batch re-loads the clamped
            // head H - 1: it is SELECTED away, not multiplied by 0 (inf * 0 would plant a NaN where the reference has +inf)
            const bool on = live && hb * 4 + u < H;
            const f32x4 x = __builtin_bit_cast(f32x4, gv[u]) * __builtin_bit_cast(f32x4, av[u]);
            s[0] += on ? relu_nan(x[0]) : 0.f; s[1] += on ? relu_nan(x[1]) : 0.f;
            s[2] += on ? relu_nan(x[2]) : 0.f; s[3] += on ? relu_nan(x[3]) : 0.f;
        }
        if (live && hb == HB - 1) {
            sink(item_layer(i), item_chunk(i), f32x4{s[0] / fH, s[1] / fH, s[2] / fH, s[3] / fH});
            s = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto run = [&](auto aux_tag) {
        u32x4 a0[4], g0[4], a1[4], g1[4];
        issue(0, a0, g0, aux_tag);
        for (int k = 0; k < total; k += 2) {
            issue(min(k + 1, total - 1), a1, g1, aux_tag);
            consume(k, true, a0, g0);
            issue(min(k + 2, total - 1), a0, g0, aux_tag);
            consume(min(k + 1, total - 1), k + 1 < total, a1, g1);
        }
    };
    // aux tag: 0 default policy | 2 nt on both slabs | 3 nt on the gradient slab only (the batch shares the probabilities)
    if (!gm.nt) run(std::integral_constant<int, 0>{});
    else if (gm.attn_bstride == 0) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 2>{});
}

// Scatter of a reduced chunk (4 consecutive elements p .. p + 3 of the row-major N x N matrix) into an LDS image with row stride S, and
// the arrival count of its block per 16-row tile, in ELEMENTS: a block [p0, p1) is contiguous in row-major order and shorter than a
// tile (256 <= 16 N once there is more than one tile), so it ends in tile t0 or t0 + 1.  LDS operations of a wave execute in issue
// order: the counts land after the elements they count.  `cnt` = the NT counters of the image.
__device__ __forceinline__ void chain_stream_deliver(float* img, int S, unsigned* cnt, int cidx, int lane, int nchunks, int N, int NN,
                                                     unsigned row_magic, f32x4 mean) {
    if (cidx < nchunks) {
        const int p = cidx * 4;
        int row = static_cast<int>(__umulhi(static_cast<unsigned>(p), row_magic)), cc = p - row * N;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p + e < NN) img[row * S + cc] = mean[e];
            if (++cc == N) { cc = 0; ++row; }
        }
    }
    // the image stores above must be ISSUED before the count that announces them: the LDS executes a wave's operations in issue
    // order, this fence (no instruction, no waitcnt) keeps the compiler from moving a store below the atomic
    __atomic_signal_fence(__ATOMIC_RELEASE);
    if (lane == 0) {
        const int p0 = (cidx - lane) * 4, p1 = min(NN, p0 + 256);
        const int t0 = static_cast<int>(__umulhi(static_cast<unsigned>(p0), row_magic)) >> 4;
        const int n0 = min(p1, (t0 + 1) * 16 * N) - p0;
        atomicAdd(cnt + t0, static_cast<unsigned>(n0));
        if (p1 - p0 > n0) atomicAdd(cnt + t0 + 1, static_cast<unsigned>(p1 - p0 - n0));
    }
}

}  // namespace mmx
