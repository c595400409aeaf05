// K_bmm_f32_tiles: C[b] = (Cin ? Cin[b] : 0) + A[b] . B[b], exact fp32 on v_mfma_f32_32x32x2_f32, for the large products of
// the N > 128 relevancy chain (R <- R + A_bar . R at 197 / 577 / 950-1050 tokens: DETR/modules/ExplanationGenerator.py:110-118,
// ViT notebook cell 7:33, CLIP_explainability.ipynb cell 6:32 on ViT-L/14@336) and of the batched rules 6 / 7 / 10.
//
// the round-1..4 kernel (bmm_f32_kernel, relevancy_kernels.hip: 64 x 64 tiles, scalar 4-byte global loads,
// scalar LDS stores, ds_read_b32 operands, two barriers per 32-wide K slab) ran at 32-40 % of the fp32 MFMA peak and lost to
// rocBLAS.  This one (950 tokens: 0.597 of the peak, rocBLAS 0.587; 577: 0.507 / 0.487; 1024: 0.706 / 0.717):
//   * TM x TN output tile per 256-thread workgroup (instantiated at 64 x 64), 4 waves as 2 x 2; a wave's (TM/2) x (TN/2) block is
//     1 (or 4) MFMA tiles of 32 x 32 (16 accumulator registers each); K slabs of 16.
//   * the k-order of a dot product is free: MFMA step t of the 8-wide k group j takes k = 8 j + 4 (lane >> 5) + t, so a lane's A
//     operands of FOUR consecutive steps are one ds_read_b128 from a k-fastest A tile (rows of 20 floats: conflict-free), and the
//     B operand is one ds_read_b32 from the n-fastest B tile (32 consecutive floats per half wave).
//   * 16-byte global loads (dword-aligned addresses suffice on gfx950; rows of 577 floats are not 16-byte aligned), 16-byte LDS
//     stores; edge chunks fall back to guarded scalar loads, out-of-range elements are zero.
//   * two LDS stages, ONE barrier per slab, PF slabs requested ahead into a ring of register sets (a slab's 0.2 us of MFMAs do
//     not cover an L2 round trip): slab s + 1 moves from its registers to the other stage before the MFMAs of slab s.
//   * 1-D grid, XCD-aware (xcd_contiguous_id): the tiles of one batch entry share an L2.
// Products it does not take (transposed A, tiny M or N, grids below one workgroup per CU) stay on bmm_f32_kernel.
#include "mmx_common.h"

namespace mmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTilesBK = 16;

template <int TM, int TN, int PF>
__global__ __launch_bounds__(256, 2) void bmm_f32_tiles_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               const float* Cin, float* C, int M, int N, int K, int64_t sa,
                                                               int64_t sb, int64_t sc, int nan_to_zero, int cin_is_row) {
    constexpr int WM = TM / 64, WN = TN / 64;          // 32 x 32 MFMA tiles per wave and dimension
    constexpr int LA = kTilesBK + 4;                   // A stage: [TM][20] floats, k fastest
    constexpr int LB = TN + 4;                         // B stage: [16][TN + 4] floats, n fastest
    constexpr int CA = TM * (kTilesBK / 4) / 256;      // 16-byte chunks of the A slab per thread
    constexpr int CB = kTilesBK * (TN / 4) / 256;      // ... of the B slab
    __shared__ __attribute__((aligned(16))) float As[2][TM * LA];
    __shared__ __attribute__((aligned(16))) float Bs[2][kTilesBK * LB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (N + TN - 1) / TN, tiles_m = (M + TM - 1) / TM;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int bx = wg % tiles_n, by = (wg / tiles_n) % tiles_m, bz = wg / (tiles_n * tiles_m);
    const int m0 = by * TM, n0 = bx * TN;
    const float* Ab = A + static_cast<int64_t>(bz) * sa;
    const float* Bb = B + static_cast<int64_t>(bz) * sb;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // per-thread chunk addresses, fixed for the whole K loop (the slab loop only adds k0 / k0 * N and compares against K)
    f32x4 ra[PF][CA], rb[PF][CB];      // slab t waits in register set t % PF
    const float* pa[CA];
    const float* pb[CB];
    int ka[CA], kb[CB];
    bool oka[CA], okb4[CB];
    int nb[CB];
#pragma unroll
    for (int c = 0; c < CA; ++c) {
        const int idx = tid + c * 256;
        const int gm = m0 + (idx >> 2);
        ka[c] = (idx & 3) * 4;
        oka[c] = gm < M;
        pa[c] = Ab + static_cast<int64_t>(oka[c] ? gm : 0) * K + ka[c];
    }
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int idx = tid + c * 256;
        kb[c] = idx / (TN / 4);
        nb[c] = n0 + (idx % (TN / 4)) * 4;
        okb4[c] = nb[c] + 3 < N;
        pb[c] = Bb + static_cast<int64_t>(kb[c]) * N + (nb[c] < N ? nb[c] : 0);
    }
    auto fetch = [&](int k0, f32x4 (&xa)[CA], f32x4 (&xb)[CB]) {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            if (oka[c] && k0 + ka[c] + 3 < K) {
                xa[c] = ldg4_u(pa[c] + k0);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oka[c])
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + ka[c] + e < K) v[e] = pa[c][k0 + e];
                xa[c] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float* src = pb[c] + static_cast<int64_t>(k0) * N;
            if (okb4[c] && k0 + kb[c] < K) {
                xb[c] = ldg4_u(src);
            } else {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k0 + kb[c] < K)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nb[c] + e < N) v[e] = src[e];
                xb[c] = v;
            }
        }
    };
    auto stash = [&](int stage, const f32x4 (&xa)[CA], const f32x4 (&xb)[CB]) {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int idx = tid + c * 256;
            *reinterpret_cast<f32x4*>(&As[stage][(idx >> 2) * LA + (idx & 3) * 4]) = xa[c];
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int idx = tid + c * 256;
            *reinterpret_cast<f32x4*>(&Bs[stage][(idx / (TN / 4)) * LB + (idx % (TN / 4)) * 4]) = xb[c];
        }
    };

    const int li = lane & 31, lg = lane >> 5;
    // 32 x 32 sub-tiles that lie completely outside C are skipped (wave-uniform): at 577 / 950 tokens the ragged last tile row and
    // column would otherwise cost 23 % / 16 % more MFMAs than the 32-granular cover of the matrix
    bool row_on[WM], col_on[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) row_on[i] = __builtin_amdgcn_readfirstlane(m0 + wr * (TM / 2) + i * 32) < M;
#pragma unroll
    for (int q = 0; q < WN; ++q) col_on[q] = __builtin_amdgcn_readfirstlane(n0 + wc * (TN / 2) + q * 32) < N;
    const int nslab = (K + kTilesBK - 1) / kTilesBK;
    fetch(0, ra[0], rb[0]);
    stash(0, ra[0], rb[0]);
#pragma unroll
    for (int j = 1; j <= PF; ++j)                            // slabs 1 .. PF in flight (set 0 is free again)
        if (j < nslab) fetch(j * kTilesBK, ra[j % PF], rb[j % PF]);
    lds_barrier();
    for (int s0 = 0; s0 < nslab; s0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int s = s0 + j;
            if (s >= nslab) break;
            const int stage = s & 1;
            if (s + 1 < nslab) {
                // every wave left stage ^ 1 at the barrier that ended slab s - 1: slab s + 1 (requested PF slabs ago) moves in before
                // the MFMAs of slab s, and its register set takes the request for slab s + 1 + PF
                stash(stage ^ 1, ra[(j + 1) % PF], rb[(j + 1) % PF]);
                if (s + 1 + PF < nslab) fetch((s + 1 + PF) * kTilesBK, ra[(j + 1) % PF], rb[(j + 1) % PF]);
            }
            const float* Asl = &As[stage][(wr * (TM / 2) + li) * LA + 4 * lg];
            const float* Bsl = &Bs[stage][(4 * lg) * LB + wc * (TN / 2) + li];
#pragma unroll
            for (int jj = 0; jj < kTilesBK / 8; ++jj) {
                f32x4 av[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) av[i] = *reinterpret_cast<const f32x4*>(Asl + i * 32 * LA + 8 * jj);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float bv[WN];
#pragma unroll
                    for (int q = 0; q < WN; ++q) bv[q] = Bsl[(8 * jj + t) * LB + q * 32];
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int q = 0; q < WN; ++q)
                            if (row_on[i] && col_on[q])
                                acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[q], acc[i][q], 0, 0, 0);
                }
            }
            if (s + 1 < nslab) lds_barrier();
        }
    }

    const int64_t cbase = static_cast<int64_t>(bz) * sc;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < WN; ++q)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int gm = m0 + wr * (TM / 2) + i * 32 + (v >> 2) * 8 + lg * 4 + (v & 3);
                const int gn = n0 + wc * (TN / 2) + q * 32 + li;
                if (gm < M && gn < N) {
                    const int64_t off = cbase + static_cast<int64_t>(gm) * N + gn;
                    float x = acc[i][q][v];
                    if (Cin) x = (cin_is_row ? Cin[gn] : Cin[off]) + x;      // cin_is_row: a bias row shared by every row
                    if (nan_to_zero && x != x) x = 0.f;
                    C[off] = x;
                }
            }
}

// Takes the product if it is one of the shapes this kernel is built for (plain A, a grid that fills the chip); returns false otherwise.
// One configuration: 64 x 64 tiles, three K slabs requested ahead.  Measured against 128 x 128 tiles and 1 / 2 slabs ahead
// (profiles/r05_chain_split_roofline.txt): 577 tokens 0.507 vs 0.416 of the fp32 MFMA peak, 950 tokens 0.597 vs 0.447, 1024 (no
// padding) 0.706 vs 0.702 -- 3200 small tiles leave each CU 12.5 -> 13 of them where 800 large ones leave 3.1 -> 4, and the edge
// tiles' idle waves give their matrix-pipe time to the other workgroups of the CU.
bool bmm_f32_tiles_try(const float* A, const float* B, const float* Cin, float* C, int batch, int M, int N, int K, int trans_a,
                       int64_t sa, int64_t sb, int64_t sc, int nan_to_zero, int cin_is_row, hipStream_t s) {
    if (trans_a || M < 96 || N < 96 || K < 32) return false;
    const int64_t wgs = static_cast<int64_t>((N + 63) / 64) * ((M + 63) / 64) * batch;
    if (wgs < device_cu_count()) return false;
    bmm_f32_tiles_kernel<64, 64, 3><<<dim3(static_cast<unsigned>(wgs)), 256, 0, s>>>(A, B, Cin, C, M, N, K, sa, sb, sc, nan_to_zero,
                                                                                    cin_is_row);
    // a failed launch is reported, not swallowed: the error stays pending (peek) for the caller's MMX_LAUNCH_CHECK, which turns it
    // into the entry point's return code; the message names this kernel
    const hipError_t e = hipPeekAtLastError();
    if (e != hipSuccess) hip_fail(e, "bmm_f32_tiles_kernel");
    return true;
}

}  // namespace mmx
