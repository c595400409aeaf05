// Small-M exact-fp32 GEMM, C [M, N] = A [M, K] . B [K, N] (+ bias row | + Cin), for the one-sample / small-batch projections of
// the explainability passes: ViT-B/16 at one image (197 rows), LXMERT at B = 32 (448 / 1152 rows), DETR at one image (100 / 950
// rows).  At these sizes a GEMM is 0.1-1 GFLOP: the library's kernels (tuned for thousands of rows) take 13-30 us each, bound
// by the latency of one workgroup walking K behind barriers (TunableOp finds nothing faster: profiles/r03_small_gemm_probe.txt),
// and bmm_f32_kernel (relevancy_kernels.hip) is K-serial with scalar loads.  This kernel has NO LDS staging and NO barrier in its
// K loop:
//
//   * a WAVE owns 16 (or 32) rows x 64 columns of C for a contiguous 1 / KS share of K (KS = 4 or 8 waves per workgroup split
//     K; their partial tiles meet once, in LDS, and are summed in wave order -- deterministic).
//   * operands go global (L2) -> registers in MFMA layout with 16-byte buffer loads: lane (c = l & 15, g = l >> 4) loads
//     A[m0 + c][kb + 4g .. + 3] (four k-steps of its row) and B[kb + 4g + i][n0 + 4c .. + 3], i = 0..3 (for k-step i the B
//     values of FOUR column tiles: tile e = columns {n0 + 4c + e}).  16 k of a 16 x 64 tile = 1 + 4 loads, 16 MFMAs.
//   * the loads run D blocks (D x 16 k) ahead of the MFMAs in a ring of register sets; every issue and every consume is
//     unconditional (past the end: the last block again, its A operand zeroed), so the compiler waits with partial vmcnt.
//   * accumulator e of a lane holds C[m0 + 4g + r][n0 + 4c + e]: for one r the four accumulators are four CONSECUTIVE columns.
//
// Workgroup order: row tiles fastest, XCD-contiguous -- the workgroups that read one 64-column strip of B sit on one XCD's L2.
// Eligibility (host side, else the older paths): K % 16 == 0, N % 4 == 0, 16-byte aligned operands, operands < 2 GB.
#include "mmx_common.h"

namespace mmx {

namespace {

template <int TMW>
struct OperandSet {
    u32x4 a[TMW];
    u32x4 b[4];
};

// KS waves (K split), TMW row tiles of 16 per wave, D blocks of 16 k in flight.
template <int KS, int TMW, int D>
__global__ __launch_bounds__(64 * KS) void linear_stream_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               const float* Cin, float* C, int M, int N, int K,
                                                               int cin_is_row) {
    constexpr int TM = 16 * TMW, RS = 68;                  // rows per workgroup; LDS row stride of a partial tile (floats)
    extern __shared__ __attribute__((aligned(16))) float red[];          // [KS][TM][RS]
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform by construction: keeps the scalar offsets in SGPRs
    const int mtiles = (M + TM - 1) / TM;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (wg % mtiles) * TM, n0 = (wg / mtiles) * 64;
    const int kblocks = K >> 4;
    const int per = (kblocks + KS - 1) / KS;
    const int b0 = min(wave * per, kblocks - 1);           // first block of this wave (clamped: an idle wave re-reads a valid one)
    const int nb = max(min(kblocks, (wave + 1) * per) - wave * per, 0);

    const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(A)), 0, 0x7fffffff, kRawBufferFlags);
    const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(B)), 0, 0x7fffffff, kRawBufferFlags);
    // rows / columns past the edge read a valid element instead: what they produce is never stored
    unsigned a_off[TMW];
#pragma unroll
    for (int t = 0; t < TMW; ++t)
        a_off[t] = static_cast<unsigned>(min(m0 + 16 * t + c16, M - 1)) * static_cast<unsigned>(K) * 4u + g * 16u;
    const unsigned rowB = static_cast<unsigned>(N) * 4u;
    const unsigned b_off = static_cast<unsigned>(4 * g) * rowB + static_cast<unsigned>(min(n0 + 4 * c16, N - 4)) * 4u;

    auto issue = [&](OperandSet<TMW>& s, int it) {
        const int blk = b0 + min(it, max(nb - 1, 0));       // wave-uniform
        const int soff_a = blk * 64;
#pragma unroll
        for (int t = 0; t < TMW; ++t) s.a[t] = __builtin_amdgcn_raw_buffer_load_b128(rA, a_off[t], soff_a, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s.b[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, b_off, static_cast<int>((blk * 16 + i) * rowB), 0);
    };
    f32x4 acc[TMW][4];
#pragma unroll
    for (int t = 0; t < TMW; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const OperandSet<TMW>& s, bool valid) {
        f32x4 av[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t) {
            av[t] = __builtin_bit_cast(f32x4, s.a[t]);
            if (!valid) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = __builtin_bit_cast(f32x4, s.b[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < TMW; ++t) acc[t][e] = mfma16x16x4(av[t][i], bv[e], acc[t][e]);
        }
    };

    OperandSet<TMW> ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(ring[d], d);
    for (int it = 0; it < nb; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            consume(ring[d], it + d < nb);
            __builtin_amdgcn_sched_barrier(0);      // keep the order: without it hipcc gathers all D sets' MFMAs behind ONE vmcnt(0)
            issue(ring[d], it + d + D);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // partial tiles -> LDS: lane holds, per r, four consecutive columns (the four accumulators)
    float* mine = red + wave * (TM * RS);
#pragma unroll
    for (int t = 0; t < TMW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(mine + (16 * t + 4 * g + r) * RS + 4 * c16) =
                f32x4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
    __syncthreads();            // (also drains the ring's surplus loads)
    for (int idx = threadIdx.x; idx < TM * 16; idx += 64 * KS) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        const int gm = m0 + row, gn = n0 + c4;
        if (gm >= M || gn >= N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(red + row * RS + c4);
#pragma unroll
        for (int w = 1; w < KS; ++w) v = v + *reinterpret_cast<const f32x4*>(red + (w * TM + row) * RS + c4);
        const int64_t off = static_cast<int64_t>(gm) * N + gn;
        if (Cin) v = *reinterpret_cast<const f32x4*>(Cin + (cin_is_row ? gn : off)) + v;
        *reinterpret_cast<f32x4*>(C + off) = v;
    }
}

template <int KS, int TMW>
int launch_stream(const float* A, const float* B, const float* Cin, float* C, int M, int N, int K, int cin_is_row, hipStream_t s) {
    constexpr int D = 8;
    constexpr size_t lds = sizeof(float) * KS * 16 * TMW * 68;
    auto kern = linear_stream_kernel<KS, TMW, D>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return 0;
    }
    const int grid = ((M + 16 * TMW - 1) / (16 * TMW)) * ((N + 63) / 64);
    kern<<<dim3(grid), 64 * KS, lds, s>>>(A, B, Cin, C, M, N, K, cin_is_row);
    return 1;
}

}  // namespace

static int g_linear_stream = 0;      // option "linear_stream": 0 never (default), 1 for eligible small-M products
void linear_stream_enable(int on) { g_linear_stream = on; }

// returns 1 if the product was launched here, 0 if the caller should take its usual path
int linear_stream_try(const float* A, const float* B, const float* Cin, float* C, int M, int N, int K, int cin_is_row,
                      hipStream_t s) {
    if (!g_linear_stream || (K & 15) || (N & 3) || M < 1 || N < 4 || K < 16) return 0;
    if (M > 2048 || static_cast<int64_t>(M) * K > (1ll << 28) || static_cast<int64_t>(K) * N > (1ll << 28)) return 0;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) |
         reinterpret_cast<uintptr_t>(Cin)) & 15u)
        return 0;
    const int kblocks = K >> 4, strips = (N + 63) / 64;
    const bool two = M > 208;                                  // 32-row workgroups once the 16-row padding waste is small
    const int wgs = ((M + (two ? 31 : 15)) / (two ? 32 : 16)) * strips;
    const bool ks8 = wgs * 4 < 1024 && kblocks >= 32;          // few workgroups: split K eight ways to fill the SIMDs
    if (two) return ks8 ? launch_stream<8, 2>(A, B, Cin, C, M, N, K, cin_is_row, s) : launch_stream<4, 2>(A, B, Cin, C, M, N, K, cin_is_row, s);
    return ks8 ? launch_stream<8, 1>(A, B, Cin, C, M, N, K, cin_is_row, s) : launch_stream<4, 1>(A, B, Cin, C, M, N, K, cin_is_row, s);
}

}  // namespace mmx
