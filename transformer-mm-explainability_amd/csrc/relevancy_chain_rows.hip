// One layer of the self-attention chain for LONG sequences (N > 128), rules 5 + 6 in ONE launch:
//     R_out[b] = R_in[b] + A_bar[b] . R_in[b],      A_bar[b] = mean_h clamp(G[b, h] * A[b, h], 0)
// (CLIP_explainability.ipynb cell 6:26-32 at ViT-L/14@336's 577 tokens, DETR/modules/ExplanationGenerator.py:110-118 at 850-1050).
//
// Rounds 1-5 ran a layer as two launches -- avg_heads_kernel (an HBM stream that WRITES A_bar, N^2 floats per sample) and the tiled
// exact-fp32 product (which reads it back, once per column tile) -- back to back: 416 us + 571 us at 577 tokens, B = 128, one
// unit idle while the other works.  Here a workgroup owns a block of 16 ROWS of one sample's R_out:
//   phase 1  every wave streams the 16 x N slice of the H head slabs (the same 16-byte / aligned-dword loads as avg_heads_kernel,
//            heads in ascending order: the sequential fp32 sum of the reference) and leaves A_bar[16 rows][N] in LDS -- A_bar never
//            travels to memory;
//   phase 2  the block row A_bar[16 x N] . R_in[N x N] on v_mfma_f32_16x16x4_f32 (exact fp32): R_in goes through LDS in slabs of 8
//            rows (a flat copy of 8 N floats: consecutive lanes read consecutive floats whatever N is), two slabs in flight; a wave owns
//            every fourth 16-column block of the result, its accumulators stay in registers over all of k;
//   epilogue R_out rows = R_in rows + accumulators.
// Two or three workgroups share a CU (75 KB of LDS each at 577 tokens), so one workgroup's HBM stream runs under another's MFMA loop:
// the overlap the two-launch form never had.  R_in is re-read from L2 by the N / 16 workgroups of a sample (1.3 MB each at 577).
// R_in == nullptr: the chain's first layer, R_in = I: R_out = I + A_bar, no product.
#include "mmx_common.h"

#include <type_traits>

namespace mmx {
namespace {

constexpr int kRows = 16;      // rows of R_out per workgroup
constexpr int kSlab = 8;       // rows of R_in per LDS slab (two MFMAs of k = 4 per column block)

struct RowsArgs {
    const void* attn;          // [B or 1][H][N][N]
    const void* grad;          // [B][H][N][N]
    const float* R_in;         // [B][N][N] or nullptr (identity)
    float* R_out;              // [B][N][N]
    int B, H, N;
    int64_t attn_bstride;      // H * N * N, or 0: one probability slab shared by the batch
    int SA;                    // row stride of the A_bar image (floats): N rounded up to kSlab, + 4
    int nblk;                  // ceil(N / 16) column blocks
    int debug;                 // profiling only (option debug_flags): 32 = no head reduction (A_bar = 0), 64 = return after the head reduction
};

// MB: column blocks per wave (compile-time bound of the accumulator array): 4 waves x MB x 16 >= N
// CH: 16-byte chunks a thread stages per slab: 256 x CH x 4 >= 8 N + 16
template <int DT, int MB, int CH>
__global__ __launch_bounds__(256) void chain_rows_layer_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* abar = reinterpret_cast<float*>(smem_raw);                 // [kRows][SA]
    float* slab = abar + kRows * a.SA;                                // [2][kSlab * N] (+ pad), flat copies of R_in rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int N = a.N, H = a.H, SA = a.SA;
    const int nrb = (N + kRows - 1) / kRows;
    // the N / 16 workgroups of a sample all read that sample's R_in (1.3 MB at 577 tokens): consecutive ids = one XCD, so that an
    // XCD's 4 MB L2 holds the one or two samples its 64 resident workgroups are on (dispatch order b % 8 -> XCD spread every sample
    // over all eight L2s: 6 GB per layer from beyond the L2, the first version's bound)
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int b = wg / nrb, r0 = (wg % nrb) * kRows;
    const int rows = min(kRows, N - r0);
    const int64_t NN = static_cast<int64_t>(N) * N;

    // ---------------------------------------------------------------------------------------------------- phase 1: A_bar block
    for (int e = tid; e < kRows * SA; e += 256) abar[e] = 0.f;        // padding columns (k >= N) and rows past N must read as 0
    __syncthreads();
    {
        const int64_t gbase = static_cast<int64_t>(b) * H * NN, abase = static_cast<int64_t>(b) * a.attn_bstride;
        const int64_t p0 = static_cast<int64_t>(r0) * N;              // the block is the flat range [p0, p0 + rows * N) of every head
        const int cnt = rows * N;
        const float fH = static_cast<float>(H);
        for (int q = tid * 4; q < ((a.debug & 32) ? 0 : cnt); q += 1024) {
            const int64_t p = p0 + q;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            // the aligned-dword form over-reads two 16-bit elements: the last chunks of a slab go element by element
            if (q + 4 <= cnt && p + 5 < NN) {
                // heads in batches of 8: all 16 loads of a batch are requested before the first is used (two workgroups per CU are
                // 8 waves: the bytes in flight per lane, not the wave count, have to cover the memory latency here)
                constexpr int HB = 8;
                for (int h0 = 0; h0 < H; h0 += HB) {
                    stream_raw<DT> ra[HB], rg[HB];
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
                        const int h = min(h0 + u, H - 1);              // a short last batch re-loads the last head (selected away below)
                        ra[u] = load4_stream_raw<DT>(a.attn, abase + h * NN + p);
                        rg[u] = load4_stream_raw<DT>(a.grad, gbase + h * NN + p);
                    }
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
                        const f32x4 x = stream_cvt<DT>(rg[u], gbase + (h0 + u) * NN + p) * stream_cvt<DT>(ra[u], abase + (h0 + u) * NN + p);
                        const bool on = h0 + u < H;
                        s[0] += on ? relu_nan(x[0]) : 0.f; s[1] += on ? relu_nan(x[1]) : 0.f;
                        s[2] += on ? relu_nan(x[2]) : 0.f; s[3] += on ? relu_nan(x[3]) : 0.f;
                    }
                }
            } else {
                for (int e = 0; e < 4 && q + e < cnt; ++e)
                    for (int h = 0; h < H; ++h)
                        s[e] += relu_nan(load1_as_f32<DT>(a.grad, gbase + h * NN + p + e) * load1_as_f32<DT>(a.attn, abase + h * NN + p + e));
            }
            int row = q / N, col = q - row * N;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (q + e < cnt) abar[row * SA + col] = s[e] / fH;
                if (++col == N) { col = 0; ++row; }
            }
        }
    }
    __syncthreads();

    float* out = a.R_out + static_cast<int64_t>(b) * NN;
    if (a.debug & 64) {
        if (tid == 0) out[static_cast<int64_t>(r0) * N] = abar[0] + abar[SA + 1];
        return;
    }
    if (a.R_in == nullptr) {                                          // first layer of a chain: R_out = I + A_bar
        for (int e = tid; e < rows * N; e += 256) {
            const int row = e / N, col = e - row * N;
            out[static_cast<int64_t>(r0 + row) * N + col] = abar[row * SA + col] + (r0 + row == col ? 1.f : 0.f);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- phase 2: A_bar . R_in
    const float* Rb = a.R_in + static_cast<int64_t>(b) * NN;
    const int slabN = kSlab * N;                                       // floats of one slab
    const int slab_stride = (slabN + 16 * 4 + 3) & ~3;                 // + slack: the last column block reads up to 15 floats past a row end
    const int nslabs = (N + kSlab - 1) / kSlab;
    f32x4 stage[CH];
    auto fetch = [&](int sidx) {
        const int64_t f0 = static_cast<int64_t>(sidx) * slabN;        // flat offset of the slab inside R_in[b]
        if (sidx + 1 < nslabs) {
            // every slab but the last lies inside R_in[b]: straight-line code, every load unconditional (a chunk past the slab end
            // re-reads the slab's last chunk and is never stored) -- with per-chunk branches each load sat in its own basic block
            // behind a wait (the first version's product phase: 33 % of the fp32 MFMA peak)
#pragma unroll
            for (int j = 0; j < CH; ++j) stage[j] = ldg4_u(Rb + f0 + min((tid + 256 * j) * 4, slabN - 4));
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) {                             // the last slab: rows k >= N are zeros, element by element
                const int q = (tid + 256 * j) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < slabN)
                    for (int e = 0; e < 4; ++e) v[e] = f0 + q + e < NN ? Rb[f0 + q + e] : 0.f;
                stage[j] = v;
            }
        }
    };
    auto store = [&](int buf) {
        float* dst = slab + buf * slab_stride;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int q = (tid + 256 * j) * 4;
            if (q < slabN + 16) *reinterpret_cast<f32x4*>(dst + q) = q < slabN ? stage[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mine = __builtin_amdgcn_readfirstlane(a.nblk > wave ? (a.nblk - wave + 3) / 4 : 0);   // column blocks wave, wave + 4, ... of this wave

    fetch(0);
    store(0);
    if (nslabs > 1) fetch(1);
    __syncthreads();
    for (int sidx = 0; sidx < nslabs; ++sidx) {
        const int cur = sidx & 1;
        if (sidx + 1 < nslabs) {
            store(cur ^ 1);                                            // the other buffer: last read two iterations ago, one barrier since
            if (sidx + 2 < nslabs) fetch(sidx + 2);
        }
        // A operand: rows c of the block, k = 8 sidx + 2 g + j (j = 0, 1): k-slot (g, j) of the two MFMAs of this slab
        const f32x2 av = *reinterpret_cast<const f32x2*>(abar + c * SA + kSlab * sidx + 2 * g);
        const float* sl = slab + cur * slab_stride + (2 * g) * N + 16 * wave + c;
        // ONE basic block per slab: all B operands requested first, then the MFMAs (a per-block `if (blk < nblk)` made every block its
        // own basic block: read -> wait -> MFMA, an LDS round trip in front of every matrix instruction).  A wave owns `mine` = MB or
        // MB - 1 blocks for every N an instantiation serves but the shortest; those take the guarded loop.
        auto slab_products = [&](auto cnt) {
            constexpr int C = decltype(cnt)::value;
            float b0[C], b1[C];
#pragma unroll
            for (int m = 0; m < C; ++m) {
                b0[m] = sl[64 * m];
                b1[m] = sl[64 * m + N];
            }
#pragma unroll
            for (int m = 0; m < C; ++m) {
                acc[m] = mfma16x16x4(av[0], b0[m], acc[m]);
                acc[m] = mfma16x16x4(av[1], b1[m], acc[m]);
            }
        };
        if (mine == MB) {
            slab_products(std::integral_constant<int, MB>{});
        } else if (mine == MB - 1) {
            slab_products(std::integral_constant<int, MB - 1>{});
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m < mine) {
                    acc[m] = mfma16x16x4(av[0], sl[64 * m], acc[m]);
                    acc[m] = mfma16x16x4(av[1], sl[64 * m + N], acc[m]);
                }
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------------------------------------------- epilogue
    // accumulators: lane (column 16 blk + c), rows r0 + 4 g + r
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int col = 16 * (wave + 4 * m) + c;
        if (wave + 4 * m < a.nblk && col < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 4 * g + r;
                if (row < N) {
                    const int64_t o = static_cast<int64_t>(row) * N + col;
                    out[o] = Rb[o] + acc[m][r];
                }
            }
        }
    }
}

template <int DT, int MB, int CH>
int launch_rows(const RowsArgs& a, size_t lds, hipStream_t s) {
    auto kern = chain_rows_layer_kernel<DT, MB, CH>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    const int nrb = (a.N + kRows - 1) / kRows;
    kern<<<dim3(static_cast<unsigned>(a.B) * nrb), 256, lds, s>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "chain_rows_layer_kernel");
    return MMX_OK;
}

template <int DT>
int dispatch_rows(const RowsArgs& a, size_t lds, hipStream_t s) {
    const int per_wave = (a.nblk + 3) / 4;
    if (per_wave <= 4) return launch_rows<DT, 4, 3>(a, lds, s);        // N <= 256:  256 x 3 x 4 = 3072 >= 8 N + 16
    if (per_wave <= 10) return launch_rows<DT, 10, 5>(a, lds, s);      // N <= 636:  5120
    return launch_rows<DT, 17, 9>(a, lds, s);                          // N <= 1088: 9216
}

size_t rows_lds_bytes(int N, int* SA_out) {
    const int SA = (N + kSlab - 1) / kSlab * kSlab + 4;
    const int slab_stride = (kSlab * N + 16 * 4 + 3) & ~3;
    if (SA_out) *SA_out = SA;
    return sizeof(float) * (static_cast<size_t>(kRows) * SA + 2 * static_cast<size_t>(slab_stride));
}

}  // namespace

// One layer R_out = R_in + A_bar . R_in (R_in == nullptr: identity).  Applies to 128 < N <= 1088 except the two narrow bands just
// below 640 and 256 tokens where the staging passes of the smaller instantiation do not cover a slab's slack (8 N + 16 floats);
// those and anything longer run on the two-launch path.  Returns false if the shape is not served.
bool chain_rows_layer_applies(int N) {
    if (N <= 128 || N > 1088 || rows_lds_bytes(N, nullptr) > 160 * 1024) return false;
    const int per_wave = ((N + 15) / 16 + 3) / 4;
    const int cover = per_wave <= 4 ? 3072 : per_wave <= 10 ? 5120 : 9216;
    return 8 * N + 16 <= cover;
}

int chain_rows_layer_launch(const void* attn, const void* grad, const float* R_in, float* R_out, int B, int H, int N, int dtype,
                            int64_t attn_bstride, hipStream_t s, int debug) {
    RowsArgs a;
    a.debug = debug;
    a.attn = attn; a.grad = grad; a.R_in = R_in; a.R_out = R_out;
    a.B = B; a.H = H; a.N = N; a.attn_bstride = attn_bstride;
    a.nblk = (N + 15) / 16;
    const size_t lds = rows_lds_bytes(N, &a.SA);
    switch (dtype) {
        case MMX_F32: return dispatch_rows<MMX_F32>(a, lds, s);
        case MMX_F16: return dispatch_rows<MMX_F16>(a, lds, s);
        case MMX_BF16: return dispatch_rows<MMX_BF16>(a, lds, s);
        default: set_error("chain_rows_layer: dtype %d", dtype); return MMX_EINVAL;
    }
}

}  // namespace mmx
