// One layer of the self-attention chain for LONG sequences (128 < N <= 636), rules 5 + 6 in ONE launch:
//     R_out[b] = R_in[b] + A_bar[b] . R_in[b],      A_bar[b] = mean_h clamp(G[b, h] * A[b, h], 0)
// (CLIP_explainability.ipynb cell 6:26-32 at ViT-L/14@336's 577 tokens; ViT notebook cell 7:2-12 at 197).
//
// Rounds 1-5 ran a layer as two launches -- avg_heads_kernel (an HBM stream that WRITES A_bar, N^2 floats per sample) and the tiled
// exact-fp32 product (which reads it back) -- back to back: 416 us + 571 us at 577 tokens, B = 128, one unit idle while the other
// works.  Here ONE persistent workgroup per CU walks blocks of 32 rows of R_out with two kinds of waves:
//   stream waves (4)  reduce the 32 x N slice of the H head slabs of the NEXT block into one of two LDS images of A_bar (the loads of
//                     avg_heads_kernel, 16 per lane in flight, heads in ascending order: the sequential fp32 sum of the reference).
//                     A_bar never travels to memory;
//   matrix waves (4)  one per SIMD: the block row A_bar[32 x N] . R_in[N x N] of the CURRENT block on v_mfma_f32_32x32x2_f32 (exact
//                     fp32).  A operand: one ds_read_b128 per four MFMA k-steps; B operand: R_in straight from L2 by raw buffer loads
//                     (32 consecutive floats of one row per half wave, the row offset in an SGPR, rows past N read as 0 through the
//                     resource's bound) -- no LDS staging, no barrier inside a block; a wave owns 5 (2) column blocks of 32 and keeps
//                     their accumulators over all of k; the operands of the next 8 k are requested before the current 8 are used.
// One workgroup barrier per BLOCK hands the filled image over.  The HBM stream of block i + 1 runs under the MFMA loop of block i:
// the overlap the two-launch form never had (and the first, non-persistent version of this file did not get either: two
// co-resident workgroups ran their phases in step -- profiles/r06_chain_rows_probe.txt).  Blocks of one sample run on ONE XCD at a
// time (xcd_contiguous_id), so that sample's R_in (1.3 MB at 577 tokens) is served by that L2.
// R_in == nullptr: the chain's first layer, R_in = I: R_out = I + A_bar, written by the matrix waves from the image.
#include "mmx_common.h"

#include <type_traits>

namespace mmx {
namespace {

constexpr int kRows = 32;      // rows of R_out per block
constexpr int kStream = 256;   // threads of the stream waves (waves 4-7)

typedef float f32x16 __attribute__((ext_vector_type(16)));
// exact-fp32 MFMA: D(32x32) += A(32x2) . B(2x32).  lane l: a = A[l & 31][l >> 5], b = B[l >> 5][l & 31];
// acc[r] = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
__device__ __forceinline__ f32x16 mfma32x32x2(float a, float b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

struct RowsArgs {
    const void* attn;          // [B or 1][H][N][N]
    const void* grad;          // [B][H][N][N]
    const float* R_in;         // [B][N][N] or nullptr (identity)
    float* R_out;              // [B][N][N]
    int B, H, N;
    int64_t attn_bstride;      // H * N * N, or 0: one probability slab shared by the batch
    int SA;                    // row stride of an A_bar image (floats): N rounded up to 8, + 4
    int nblocks;               // B * ceil(N / 32)
    int debug;                 // profiling only (option debug_flags): 32 = no head reduction (A_bar = 0), 64 = no product
};

// the 32 x N block `blk` of A_bar -> image `img` (the k-padding stays zero from the initial clear; rows past N of a sample's last
// block are re-zeroed here, an earlier block may have left values there)
template <int DT>
__device__ __forceinline__ void stream_block(const RowsArgs& a, int blk, float* img, int st) {
    const int N = a.N, H = a.H, SA = a.SA;
    const int nrb = (N + kRows - 1) / kRows;
    const int b = blk / nrb, r0 = (blk % nrb) * kRows;
    const int rows = min(kRows, N - r0);
    const int64_t NN = static_cast<int64_t>(N) * N;
    const int64_t gbase = static_cast<int64_t>(b) * H * NN, abase = static_cast<int64_t>(b) * a.attn_bstride;
    const int64_t p0 = static_cast<int64_t>(r0) * N;                  // the block is the flat range [p0, p0 + rows * N) of every head
    const int cnt = rows * N;
    const float fH = static_cast<float>(H);
    if (rows < kRows)
        for (int e = st; e < (kRows - rows) * SA; e += kStream) img[rows * SA + e] = 0.f;
    // Flat sequence of batches t = (position chunk, batch of 8 heads): the loads of batch t + 1 are requested before batch t is
    // reduced (two register sets), so a lane keeps 16-32 loads in flight WITHOUT draining between position chunks -- four stream
    // waves per CU have to cover the memory latency with bytes in flight per lane, not with wave count.  Every request is
    // unconditional (clamped chunk / head indices, results selected away): the wait counts of the two sets stay apart.
    constexpr int HB = 8;
    const int nhb = (H + HB - 1) / HB;
    const int nchunks = (a.debug & 32) ? 0 : (cnt + 3) / 4;
    const int mine = st < nchunks ? (nchunks - st + kStream - 1) / kStream : 0;
    const int total = mine * nhb;
    // a chunk the aligned-dword form may not touch (it over-reads two 16-bit elements; the block's ragged end): element by element
    auto slow_chunk = [&](int q) { return !(q + 4 <= cnt && p0 + q + 5 < NN); };
    auto chunk_q = [&](int t) { return (st + (t / nhb) * kStream) * 4; };
    // (the gradient slab through a raw buffer with nt loads -- tried so that the read-once stream would not evict the R_in rows the
    // matrix waves live on -- made the stream 20 % slower, nt or not: plain loads, profiles/r06_chain_rows_probe.txt)
    auto issue = [&](int t, stream_raw<DT> (&ra)[HB], stream_raw<DT> (&rg)[HB]) {
        const int q = chunk_q(t), hb = t % nhb;
        const int64_t p = p0 + (slow_chunk(q) ? 0 : q);                // (a slow chunk's fast loads go to the block start: valid, unused)
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int h = min(hb * HB + u, H - 1);
            ra[u] = load4_stream_raw<DT>(a.attn, abase + h * NN + p);
            rg[u] = load4_stream_raw<DT>(a.grad, gbase + h * NN + p);
        }
    };
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](int t, bool live, const stream_raw<DT> (&ra)[HB], const stream_raw<DT> (&rg)[HB]) {
        const int q = chunk_q(t), hb = t % nhb;
        const bool slow = slow_chunk(q);
        const int64_t p = p0 + (slow ? 0 : q);
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int h = min(hb * HB + u, H - 1);
            const f32x4 x = stream_cvt<DT>(rg[u], gbase + h * NN + p) * stream_cvt<DT>(ra[u], abase + h * NN + p);
            const bool on = live && !slow && hb * HB + u < H;
            s[0] += on ? relu_nan(x[0]) : 0.f; s[1] += on ? relu_nan(x[1]) : 0.f;
            s[2] += on ? relu_nan(x[2]) : 0.f; s[3] += on ? relu_nan(x[3]) : 0.f;
        }
        if (live && hb == nhb - 1) {
            if (slow) {
                s = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int e = 0; e < 4 && q + e < cnt; ++e)
                    for (int h = 0; h < H; ++h)
                        s[e] += relu_nan(load1_as_f32<DT>(a.grad, gbase + h * NN + p0 + q + e) * load1_as_f32<DT>(a.attn, abase + h * NN + p0 + q + e));
            }
            int row = q / N, col = q - row * N;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q + e < cnt) img[row * SA + col] = s[e] / fH;
                if (++col == N) { col = 0; ++row; }
            }
            s = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (total > 0) {
        stream_raw<DT> a0[HB], g0[HB], a1[HB], g1[HB];
        issue(0, a0, g0);
        for (int t = 0; t < total; t += 2) {
            issue(min(t + 1, total - 1), a1, g1);
            consume(t, true, a0, g0);
            issue(min(t + 2, total - 1), a0, g0);
            consume(min(t + 1, total - 1), t + 1 < total, a1, g1);
        }
    }
}

// MB: column blocks of 32 per matrix wave: 4 waves x MB x 32 >= N
template <int DT, int MB>
__global__ __launch_bounds__(512, 2) void chain_rows_layer_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* img0 = reinterpret_cast<float*>(smem_raw);                 // [2][kRows][SA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
    const int N = a.N, SA = a.SA;
    const int nrb = (N + kRows - 1) / kRows;
    const int64_t NN = static_cast<int64_t>(N) * N;
    const int img_floats = kRows * SA;
    const bool streamer = __builtin_amdgcn_readfirstlane(wave) >= 4;
    // blocks in (sample, row block) order, row block fastest; in round `it` the resident workgroups take blocks it * G + w', w' the
    // XCD-contiguous id: an XCD works on 32 consecutive blocks = under two samples at a time
    const int G = gridDim.x, wid = xcd_contiguous_id(blockIdx.x, G);
    for (int e = tid; e < 2 * img_floats; e += 512) img0[e] = 0.f;    // k-padding and rows past N must read as 0
    __syncthreads();
    if (streamer && wid < a.nblocks) stream_block<DT>(a, wid, img0, tid - 256);
    __syncthreads();

    const int kgroups = (N + 7) / 8;                                   // k in groups of 8: four MFMA k-steps, slot (hi, j) <-> k = 8 kg + 4 hi + j
    const int cw = (wave & 3) * MB * 32;                               // first column of this matrix wave
    int cur = 0;
    for (int blk = wid; blk < a.nblocks; blk += G, cur ^= 1) {
        float* img = img0 + cur * img_floats;
        if (streamer) {
            if (blk + G < a.nblocks) stream_block<DT>(a, blk + G, img0 + (cur ^ 1) * img_floats, tid - 256);
        } else if (!(a.debug & 64)) {
            const int b = blk / nrb, r0 = (blk % nrb) * kRows;
            float* out = a.R_out + static_cast<int64_t>(b) * NN;
            if (a.R_in == nullptr) {                                   // first layer of a chain: R_out = I + A_bar
                const int rows = min(kRows, N - r0);
                for (int e = tid; e < rows * N; e += 256) {
                    const int row = e / N, col = e - row * N;
                    out[static_cast<int64_t>(r0 + row) * N + col] = img[row * SA + col] + (r0 + row == col ? 1.f : 0.f);
                }
            } else {
                const float* Rb = a.R_in + static_cast<int64_t>(b) * NN;
                // raw buffer over this sample's R_in: rows k >= N (the k-padding of the last group) lie past the bound and read as 0
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(Rb)), 0, static_cast<int>(NN * 4), kRawBufferFlags);
                // B operands: a matrix wave owns MB x 32 columns from cw on.  Its first 128 are read as ONE 16-byte load per lane and
                // k row (lane i: columns cw + 4 i .. + 3; component m' feeds MFMA m', whose output column n = i is then cw + 4 i + m':
                // the column order inside a wave is free), the remaining 32 (MB = 5) as a dword.  A wave can have 63 loads in flight
                // (vmcnt is 6 bits) and 23 % of these miss the L2: with one DWORD per MFMA the first version kept 60 x 256 B in flight
                // per wave and waited for memory half of the time (SQ_VALU_MFMA_BUSY 52 %).
                constexpr bool NARROW = MB == 5;                       // MB = 2 (N <= 256): 64 columns per wave, ONE 8-byte load per lane
                constexpr int WIDE = MB == 5 ? 4 : MB;                 // components of the wide load = column blocks served by it
                constexpr int WC = 32 * WIDE;                          // columns of the wide part (128 / 64)
                // No load may straddle a row end (the quad of a lane would mix two rows, and past the sample's end be cut by the
                // bound): the wave whose nominal range [cw, cw + 32 MB) crosses N reads its wide part ALIGNED TO THE ROW END
                // ([N - WC, N), overlapping its left neighbour: computed twice, stored once) and, if more than WC columns are its
                // own, the narrow part as [N - 32, N).  Stores are limited to the wave's own columns.
                const int own = min(N - cw, 32 * MB);                  // columns of this wave (<= 0: none, it only keeps step)
                const int wb = own >= WC ? cw : max(N - WC, 0);        // first column of the wide part
                const int nb = own >= 32 * MB ? cw + WC : N - 32;      // first column of the narrow part (MB = 5)
                const int wcol = wb + (WC / 32) * i;                   // lane i: columns wcol .. wcol + WC / 32 - 1
                const unsigned voffw = static_cast<unsigned>((4 * hi) * N + wcol) * 4u;
                const unsigned voffn = static_cast<unsigned>((4 * hi) * N + nb + i) * 4u;
                f32x16 acc[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
                const float* arow = img + i * SA + 4 * hi;             // A operand: row i of the block, k = 8 kg + 4 hi + (0 .. 3)
                constexpr int RING = 4;                                // k groups whose B operands are in flight
                float bq[RING][4][MB];                                  // [set][j][m]: B operands of one k group
                auto request = [&](int kg, auto set) {
                    constexpr int S = decltype(set)::value;
                    // ONE straight-line form for every group, the row offset in the VECTOR part of the address: the hardware checks the
                    // vector offset against the resource's bound (rows k >= N of the last group read as 0, not as the next sample's
                    // values, whose NaN times A_bar's 0 would be a NaN), and a second, SGPR-offset form for the inner groups gave
                    // hipcc's wait-count pass two paths with different load counts -- it then waited for the group just requested
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned rowoff = static_cast<unsigned>((8 * kg + j) * N * 4);
                        if constexpr (MB == 5) {
                            const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffw + rowoff, 0, 0));
#pragma unroll
                            for (int m = 0; m < 4; ++m) bq[S][j][m] = w[m];
                            bq[S][j][4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffn + rowoff, 0, 0));
                        } else {
                            const f32x2 w = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffw + rowoff, 0, 0));
                            bq[S][j][0] = w[0];
                            bq[S][j][1] = w[1];
                        }
                    }
                };
                auto multiply = [&](int kg, auto set) {
                    constexpr int S = decltype(set)::value;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 8 * kg);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m] = mfma32x32x2(av[j], bq[S][j][m], acc[m]);
                };
                // The row blocks of a sample run side by side on one XCD and would all ask for the same rows of R_in at the same time:
                // block rb starts its walk over k at group rb * kgroups / nrb and wraps around (the sum over k in another, still fixed,
                // order), so each block is the first at a different 1 / nrb of the rows
                const int rot = ((blk % nrb) * kgroups) / nrb;
                auto grp = [&](int t) { const int g = min(t, kgroups - 1) + rot; return g >= kgroups ? g - kgroups : g; };
                request(grp(0), std::integral_constant<int, 0>{});
                request(grp(1), std::integral_constant<int, 1>{});
                request(grp(2), std::integral_constant<int, 2>{});
                for (int t = 0; t < kgroups; t += RING) {              // set s holds step t + s; step t + 3 + s is requested into the set used last
                    request(grp(t + 3), std::integral_constant<int, 3>{});
                    multiply(grp(t), std::integral_constant<int, 0>{});
                    request(grp(t + 4), std::integral_constant<int, 0>{});
                    if (t + 1 < kgroups) multiply(grp(t + 1), std::integral_constant<int, 1>{});
                    request(grp(t + 5), std::integral_constant<int, 1>{});
                    if (t + 2 < kgroups) multiply(grp(t + 2), std::integral_constant<int, 2>{});
                    request(grp(t + 6), std::integral_constant<int, 2>{});
                    if (t + 3 < kgroups) multiply(grp(t + 3), std::integral_constant<int, 3>{});
                }
                // epilogue: accumulator m of lane i is column wcol + m (m < WIDE) or nb + i (m = 4);
                // rows r0 + (r & 3) + 8 (r >> 2) + 4 hi.  One column block at a time (its R_in values, not all of them, in registers).
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < N) {
                        const int64_t o = static_cast<int64_t>(row) * N;
#pragma unroll
                        for (int m = 0; m < WIDE; ++m) {
                            const int col = wcol + m;
                            if (col >= cw && col < min(N, cw + WC)) out[o + col] = Rb[o + col] + acc[m][r];
                        }
                        if constexpr (NARROW) {
                            const int col = nb + i;
                            if (col >= cw + WC && col < N) out[o + col] = Rb[o + col] + acc[4][r];
                        }
                    }
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();        // image `cur ^ 1` is complete and image `cur` is free: ONE barrier per block
    }
}

size_t rows_lds_bytes(int N, int* SA_out) {
    const int SA = (N + 7) / 8 * 8 + 4;
    if (SA_out) *SA_out = SA;
    return sizeof(float) * 2 * static_cast<size_t>(kRows) * SA;
}

template <int DT, int MB>
int launch_rows(const RowsArgs& a, size_t lds, hipStream_t s) {
    auto kern = chain_rows_layer_kernel<DT, MB>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    const int grid = min(a.nblocks, device_cu_count());               // persistent: one workgroup per CU (the images fill its LDS)
    kern<<<dim3(static_cast<unsigned>(grid)), 512, lds, s>>>(a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "chain_rows_layer_kernel");
    return MMX_OK;
}

template <int DT>
int dispatch_rows(const RowsArgs& a, size_t lds, hipStream_t s) {
    return a.N <= 256 ? launch_rows<DT, 2>(a, lds, s) : launch_rows<DT, 5>(a, lds, s);
}

}  // namespace

// One layer R_out = R_in + A_bar . R_in (R_in == nullptr: identity).  Serves 128 < N <= 636 (two 32-row images of A_bar in a CU's
// LDS, 5 column blocks of 32 per matrix wave); anything longer (DETR's 850-1050 tokens) runs on the two-launch path.
bool chain_rows_layer_applies(int N) {
    return N > 128 && N <= 636 && rows_lds_bytes(N, nullptr) <= 160 * 1024;
}

int chain_rows_layer_launch(const void* attn, const void* grad, const float* R_in, float* R_out, int B, int H, int N, int dtype,
                            int64_t attn_bstride, hipStream_t s, int debug) {
    RowsArgs a;
    a.debug = debug;
    a.attn = attn; a.grad = grad; a.R_in = R_in; a.R_out = R_out;
    a.B = B; a.H = H; a.N = N; a.attn_bstride = attn_bstride;
    a.nblocks = B * ((N + kRows - 1) / kRows);
    const size_t lds = rows_lds_bytes(N, &a.SA);
    switch (dtype) {
        case MMX_F32: return dispatch_rows<MMX_F32>(a, lds, s);
        case MMX_F16: return dispatch_rows<MMX_F16>(a, lds, s);
        case MMX_BF16: return dispatch_rows<MMX_BF16>(a, lds, s);
        default: set_error("chain_rows_layer: dtype %d", dtype); return MMX_EINVAL;
    }
}

}  // namespace mmx
