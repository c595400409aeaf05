// Attention-capture kernels for LONG sequences on gfx950 (DETR encoder 850-1050 image tokens, ViT-L/14@336 577):
// the whole-head kernels of attention_small.hip need the head's score matrix in LDS, the first-generation tiled
// kernels of attention_kernels.hip keep a [16 x Nk] score row block there and re-stage K / V for every 16 queries.
// Here nothing of size Nk lives on chip:
//
//   * a workgroup owns 64 query rows (forward, dQ half of backward) or 64 keys (dK/dV half); each of its 4 waves
//     owns 16 of them and keeps its MFMA A operand (Q, dO, or the P / dS columns) in registers;
//   * K / V (or Q / dO) stream through LDS in 64-row tiles; the NEXT tile's global loads are issued into registers
//     before the MFMAs of the current one and the barriers are LDS-only (s_waitcnt lgkmcnt), so loads and the
//     capture-slab stores stay in flight across them;
//   * forward is two sweeps over the keys: (1) running row max / row sum (lane-local online softmax, merged across
//     the 16 lanes of a row with DPP at the end), (2) recompute S, write P = exp(S - max) / sum straight to the
//     capture slab and accumulate O = P.V.  Recomputing S costs MFMA time the kernel has to spare: it is bound by
//     the N^2 slab traffic (P written once; in backward P read and dP written), which is the algorithmic minimum
//     because P and dP are the PRODUCT here (they feed the relevancy rules), unlike flash attention;
//   * every product runs on the exact-fp32 v_mfma_f32_16x16x4_f32.  The contraction index is visited in the order
//     (block, lane>>4, step) so that each lane's four consecutive steps read ONE ds_read_b128 / global float4.
//
// Operand geometry (lane l, i = l & 15, g = l >> 4):  A[m = i][k-slot g],  B[k-slot g][n = i],
// C/D[m = 4g + r][n = i], r = 0..3.  k-slot g of step s in block blk is contraction index 16 blk + 4 g + s.
#include "mmx_common.h"
#include "attention_args.h"

#include <type_traits>

namespace mmx {
namespace {

constexpr int kRows = 64;      // query rows (or keys, in the dK/dV kernel) per workgroup = 4 waves x 16
constexpr int kTile = 64;      // rows of the streamed operand per step
constexpr int kThreads = 256;
constexpr int kPS = kTile + 4; // row stride of a wave's private [16][64] P / dS tile (16-byte rows, conflict-free b128)

int g_attn_stream = 1;
int g_attn_fwd_split = 1;   // option "attn_fwd_split": the small-grid forward (attn_fwd_split_kernel)

// exp(x) = 2^(x log2 e) on the hardware v_exp_f32 (1 ulp).  The rounding of x * log2(e) adds |x| * 6e-8 of relative
// error; the ABSOLUTE error of a probability p = exp(x) is then at most max_x |x| e^x * 6e-8 = 2.2e-8, and its relative
// error stays below the 1e-5 parity bar for every x that does not underflow.  Two instructions instead of libm expf's
// ~15: PMC showed these kernels issue-bound on VALU work around the MFMAs, not HBM-bound.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ---- streamed [64 x D] operand tile: global -> registers (prefetch) -> LDS, row stride DP + 4 floats.
// The prefetch registers hold the RAW loaded words: masking (rows past the end, d >= D), the bf16 -> fp32 widening of a
// bf16 operand and the scale all happen in tile_store, one iteration later.  Any ALU op on the loaded value at fetch time
// makes the compiler wait for the load right there (s_waitcnt vmcnt(0) before the tile's MFMAs): PMC showed the bf16-input
// variant, which widened at fetch time, parked 48 % of its wave cycles against 33 %.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 r) {
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                 __uint_as_float(r[1] & 0xffff0000u)};
}
template <int DP, bool HALF = false>
struct TileRegs {
    typename std::conditional<HALF, u32x2, f32x4>::type raw[DP / 16];
    int row0;
};
template <int DP, bool HALF, typename T>
__device__ __forceinline__ void tile_fetch(TileRegs<DP, HALF>& reg, const T* base, int64_t sn, int row0, int rows_total,
                                           int D, int tid) {
    static_assert(sizeof(T) == (HALF ? 2 : 4), "operand type");
    constexpr int C4 = DP / 4;
    reg.row0 = row0;
#pragma unroll
    for (int e = 0; e < DP / 16; ++e) {
        const int f = tid + kThreads * e;
        const int row = row0 + f / C4, c = (f % C4) * 4;
        const bool ok = row < rows_total && c < D;
        // unconditional (clamped) load: a conditional load would serialise on vmcnt(0) per element
        const T* src = base + (ok ? static_cast<int64_t>(row) * sn + c : 0);
        if constexpr (HALF) reg.raw[e] = *reinterpret_cast<const u32x2*>(src);
        else reg.raw[e] = *reinterpret_cast<const f32x4*>(src);
    }
}

template <int DP, bool HALF>
__device__ __forceinline__ void tile_store(float* lds, const TileRegs<DP, HALF>& reg, float mul, int rows_total, int D, int tid) {
    constexpr int C4 = DP / 4, LS = DP + 4;
#pragma unroll
    for (int e = 0; e < DP / 16; ++e) {
        const int f = tid + kThreads * e;
        const bool ok = reg.row0 + f / C4 < rows_total && (f % C4) * 4 < D;
        f32x4 v;
        if constexpr (HALF) v = widen_bf16x4(reg.raw[e]);
        else v = reg.raw[e];
        *reinterpret_cast<f32x4*>(lds + (f / C4) * LS + (f % C4) * 4) = ok ? v * mul : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// A operand held in registers: rows row0 + i of a [rows x D] matrix, contraction index permuted as in the header
template <int DP>
__device__ __forceinline__ void load_a_rows(f32x4 (&a)[DP / 16], const float* base, int64_t sn, int row, int D,
                                            int g, float mul) {
#pragma unroll
    for (int blk = 0; blk < DP / 16; ++blk) {
        const int d0 = 16 * blk + 4 * g;
        const bool ok = d0 < D;
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(row) * sn + (ok ? d0 : 0));
        a[blk] = ok ? v * mul : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- load_a_rows for a bf16 operand in global memory (MMX_ATTN_IO_BF16: dO): 8-byte loads, widened to fp32
template <int DP>
__device__ __forceinline__ void load_a_rows_h(f32x4 (&a)[DP / 16], const unsigned short* base, int64_t sn, int row, int D,
                                              int g) {
#pragma unroll
    for (int blk = 0; blk < DP / 16; ++blk) {
        const int d0 = 16 * blk + 4 * g;
        const bool ok = d0 < D;
        const u32x2 v = *reinterpret_cast<const u32x2*>(base + static_cast<int64_t>(row) * sn + (ok ? d0 : 0));
        a[blk] = ok ? widen_bf16x4(v) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// bf16 epilogue of a C-layout accumulator (lane (i, g): rows 4g + r, column 16 dt + i).  Element-wise that is one 2-byte
// store per value; here lanes i and i ^ 1 swap one value per row pair (DPP quad_perm, no LDS), so the even lane owns rows
// r = 0, 2 and the odd lane rows r = 1, 3 as (column, column + 1) PAIRS: 4-byte stores, half as many.
__device__ __forceinline__ float dpp_swap1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
// v[r]: this lane's 4 row values of column `col`; row_ok(r) / row_idx(r) describe row r.  All lanes must call it.
template <typename RowOk, typename RowIdx>
__device__ __forceinline__ void store_rows_bf16_pairs(float* base, int64_t off0, int64_t sn, int col, int D, const f32x4& v,
                                                      int i, RowOk row_ok, RowIdx row_idx) {
    const bool odd = i & 1;
    unsigned short* out = reinterpret_cast<unsigned short*>(base);
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        const float mine = odd ? v[2 * rp + 1] : v[2 * rp];              // the value of the row this lane will write
        const float recv = dpp_swap1(odd ? v[2 * rp] : v[2 * rp + 1]);   // the partner's value of that same row
        const int r = 2 * rp + (odd ? 1 : 0);
        const int c0 = col - (odd ? 1 : 0);                               // even column of the pair
        if (row_ok(r) && c0 < D)
            *reinterpret_cast<unsigned*>(out + off0 + row_idx(r) * sn + c0) = odd ? pack2_bf16(recv, mine) : pack2_bf16(mine, recv);
    }
}

// C_t[16 x 16] = A_regs[16 x D] . T_t^T for the four 16-row sub-tiles t of the LDS tile ([row][d], contiguous in d).
// The four accumulator chains are issued round-robin: a dependent v_mfma_f32_16x16x4_f32 has 40 cycles of latency
// against a 32-cycle issue rate, and anything the compiler slips between two MFMAs on the SAME accumulator costs a
// further ~43 cycles (MI355X_MICROARCH.md, per-instruction constants) -- PMC on the first version of these kernels:
// 54 % of the wave cycles were issue stalls at 33 % MFMA utilisation.
template <int DP>
__device__ __forceinline__ void tile_abt4(f32x4 (&acc)[4], const f32x4 (&a)[DP / 16], const float* tile, int i, int g) {
    constexpr int LS = DP + 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p = tile + i * LS + 4 * g;
#pragma unroll
    for (int blk = 0; blk < DP / 16; ++blk) {
        f32x4 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x4*>(p + 16 * t * LS + 16 * blk);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16x16x4(a[blk][s], b[t][s], acc[t]);
    }
}

// acc[half][dt] (16 x 16 each, columns d = 16 dt + i) += W[16 x 64] . T[64 x D]: W = the wave's private tile (A
// operand, one b128 per 4 steps), T = the LDS tile read along its rows.  Two 16-row blocks of the contraction run
// side by side on separate accumulators (2 x DP/16 independent chains); the caller adds the halves at the very end.
template <int DP>
__device__ __forceinline__ void tile_wt(f32x4 (&acc)[2][DP / 16], const float* w, const float* tile, int i, int g) {
    constexpr int LS = DP + 4;
#pragma unroll
    for (int kb = 0; kb < kTile / 16; kb += 2) {
        const f32x4 av0 = *reinterpret_cast<const f32x4*>(w + i * kPS + 16 * kb + 4 * g);
        const f32x4 av1 = *reinterpret_cast<const f32x4*>(w + i * kPS + 16 * (kb + 1) + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* row0 = tile + (16 * kb + 4 * g + s) * LS + i;
            const float* row1 = row0 + 16 * LS;
#pragma unroll
            for (int dt = 0; dt < DP / 16; ++dt) {
                acc[0][dt] = mfma16x16x4(av0[s], row0[16 * dt], acc[0][dt]);
                acc[1][dt] = mfma16x16x4(av1[s], row1[16 * dt], acc[1][dt]);
            }
        }
    }
}

// ---- the same two products on v_mfma_f32_16x16x32_bf16 (MMX_ATTN_MMA_BF16).  The LDS tiles stay fp32: an operand is
// the 8 values of TWO of the b128 (or 2 x 4 b32) reads the fp32 form already makes, rounded to bf16 in registers
// (4 v_cvt_pk_bf16_f32), so one bf16 MFMA replaces eight exact-fp32 ones on the same LDS traffic.
// Contraction slot (g, j) of pair pr  <->  index 32 pr + 16 (j >> 2) + 4 g + (j & 3), for both operands.
template <int DP>
__device__ __forceinline__ void pack_a_rows(bf16x8 (&apk)[DP / 32], const f32x4 (&a)[DP / 16]) {
#pragma unroll
    for (int pr = 0; pr < DP / 32; ++pr) apk[pr] = pack_bf16(a[2 * pr], a[2 * pr + 1]);
}

// acc[t][4g + r][i] = sum_d  R[i or 4g+r ...]: SWAP = false: C[m][n] with m = the register operand's row (A = regs, B = LDS
// rows 16 t + i);  SWAP = true: A = LDS rows 16 t + i (m), B = regs (n).
template <int DP, bool SWAP>
__device__ __forceinline__ void tile_abt4_bf16(f32x4 (&acc)[4], const bf16x8 (&a)[DP / 32], const float* tile, int i, int g) {
    constexpr int LS = DP + 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p = tile + i * LS + 4 * g;
#pragma unroll
    for (int pr = 0; pr < DP / 32; ++pr) {
        bf16x8 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            b[t] = pack_bf16(*reinterpret_cast<const f32x4*>(p + 16 * t * LS + 32 * pr),
                             *reinterpret_cast<const f32x4*>(p + 16 * t * LS + 32 * pr + 16));
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = SWAP ? mfma16x16x32_bf16(b[t], a[pr], acc[t]) : mfma16x16x32_bf16(a[pr], b[t], acc[t]);
    }
}

template <int DP>
__device__ __forceinline__ void tile_wt_bf16(f32x4 (&acc)[2][DP / 16], const float* w, const float* tile, int i, int g) {
    constexpr int LS = DP + 4;
#pragma unroll
    for (int pr = 0; pr < kTile / 32; ++pr) {
        const bf16x8 av = pack_bf16(*reinterpret_cast<const f32x4*>(w + i * kPS + 32 * pr + 4 * g),
                                    *reinterpret_cast<const f32x4*>(w + i * kPS + 32 * pr + 16 + 4 * g));
        const float* row0 = tile + (32 * pr + 4 * g) * LS + i;
#pragma unroll
        for (int dt = 0; dt < DP / 16; ++dt) {
            f32x4 lo, hi;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                lo[s2] = row0[s2 * LS + 16 * dt];
                hi[s2] = row0[(16 + s2) * LS + 16 * dt];
            }
            acc[pr][dt] = mfma16x16x32_bf16(av, pack_bf16(lo, hi), acc[pr][dt]);
        }
    }
}

template <int DP>
constexpr size_t stream_lds_bytes(int tiles) {
    return sizeof(float) * (static_cast<size_t>(tiles) * kTile * (DP + 4) + 4 * 16 * kPS);
}

// =============================================================================================== forward
template <int DP, int DT, bool MM>
__global__ __launch_bounds__(kThreads, DP == 32 ? 3 : 2) void attn_fwd_stream_kernel(const AttnFwdArgs a) {
    typedef typename slab_elem<DT>::type slab_t;
    constexpr int LS = DP + 4, NB = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    float* Ks = smem;
    float* Vs = Ks + kTile * LS;
    float* Pw = Vs + kTile * LS + wave * 16 * kPS;
    // 1-D grid, XCD-contiguous logical order (row tile fastest): the row tiles of a head share that head's K / V in
    // one XCD's L2
    const int nrt = (a.Nq + kRows - 1) / kRows;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int h = (wg / nrt) % a.H, b = wg / (nrt * a.H);
    const int rw = (wg % nrt) * kRows + wave * 16;          // first query row of this wave
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const float ninf = -__builtin_inff();

    f32x4 qa[NB];
    load_a_rows<DP>(qa, qb, a.qs.sn, min(rw + i, a.Nq - 1), a.D, g, q_first ? a.scale : 1.f);
    bf16x8 qa_pk[DP / 32];
    if constexpr (MM) pack_a_rows<DP>(qa_pk, qa);
    auto s_tile = [&](f32x4 (&sacc)[4]) {
        if constexpr (MM) tile_abt4_bf16<DP, false>(sacc, qa_pk, Ks, i, g);
        else tile_abt4<DP>(sacc, qa, Ks, i, g);
    };

    // per-lane row pointers (+ this lane's key column i): inside the sweeps only a wave-uniform key offset is added,
    // and rows beyond Nq simply have no output pointer
    int rows[4];
    const float* mrow[4];
    slab_t* pout[4];
    const int64_t pbase = (static_cast<int64_t>(b) * a.H + h) * a.Nq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = rw + 4 * g + r;
        mrow[r] = a.mask ? a.mask + b * a.mask_sb + static_cast<int64_t>(min(rows[r], a.Nq - 1)) * a.mask_sq : nullptr;
        pout[r] = rows[r] < a.Nq ? reinterpret_cast<slab_t*>(a.probs) + (pbase + rows[r]) * a.Nk + i : nullptr;
    }
    // one score of the C tile: scale, additive mask, -inf beyond the last key.  EDGE = the tile may run past Nk
    // (only the last tile of a sweep); interior tiles skip every key-range test.
    auto score = [&](float s, int r, int k0, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        if (!q_first) s = s / a.scale;                      // scale = sqrt(d) divisor in MMX_SCALE_SCORES mode
        if (a.mask) s += mrow[r][EDGE ? min(k0 + i, a.Nk - 1) : k0 + i];
        return (!EDGE || k0 + i < a.Nk) ? s : ninf;
    };

    const int ntiles = (a.Nk + kTile - 1) / kTile;
    TileRegs<DP> kreg, vreg;

    // ---- sweep 1: lane-local running max m and sum l of exp(s - m) over this lane's keys
    float m[4] = {ninf, ninf, ninf, ninf}, l[4] = {0.f, 0.f, 0.f, 0.f};
    auto sweep1 = [&](int kt, auto edge) {
        f32x4 sacc[4];
        s_tile(sacc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sv[t] = score(sacc[t][r], r, kt * kTile + 16 * t, edge);
            const float mn = fmaxf(fmaxf(m[r], fmaxf(sv[0], sv[1])), fmaxf(sv[2], sv[3]));
            const float base = (mn == ninf) ? 0.f : mn;     // all keys masked so far: keep l = 0 instead of inf - inf
            l[r] = l[r] * exp_fast(m[r] - base) + exp_fast(sv[0] - base) + exp_fast(sv[1] - base) +
                   exp_fast(sv[2] - base) + exp_fast(sv[3] - base);
            m[r] = mn;
        }
    };
    tile_fetch<DP>(kreg, kb, a.ks.sn, 0, a.Nk, a.D, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        lds_barrier();
        tile_store<DP>(Ks, kreg, 1.f, a.Nk, a.D, tid);
        lds_barrier();
        if (kt + 1 < ntiles) tile_fetch<DP>(kreg, kb, a.ks.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
        if (kt + 1 < ntiles) sweep1(kt, std::false_type{}); else sweep1(kt, std::true_type{});
    }
    // merge the 16 lanes of each row (a fully masked row ends as 0 * inf = NaN like torch.softmax)
    float linv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mall = group16_max(m[r]);
        const float base = (mall == ninf) ? 0.f : mall;
        linv[r] = 1.f / group16_sum(l[r] * exp_fast(m[r] - base));
        m[r] = base;
    }

    // ---- sweep 2: P -> capture slab, O += P.V
    f32x4 oacc[2][NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) oacc[0][dt] = oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto sweep2 = [&](int kt, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        f32x4 sacc[4];
        s_tile(sacc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = kt * kTile + 16 * t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = exp_fast(score(sacc[t][r], r, k0, edge) - m[r]) * linv[r];
                if (EDGE && k0 + i >= a.Nk) p = 0.f;                        // (also keeps NaN rows out of the padding)
                if (pout[r] && (!EDGE || k0 + i < a.Nk)) slab_store<DT>(pout[r] + k0, p);
                Pw[(4 * g + r) * kPS + 16 * t + i] = p;
            }
        }
        if constexpr (MM) tile_wt_bf16<DP>(oacc, Pw, Vs, i, g);
        else tile_wt<DP>(oacc, Pw, Vs, i, g);               // same-wave LDS traffic is in order: no barrier needed
    };
    tile_fetch<DP>(kreg, kb, a.ks.sn, 0, a.Nk, a.D, tid);
    tile_fetch<DP>(vreg, vb, a.vs.sn, 0, a.Nk, a.D, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        lds_barrier();
        tile_store<DP>(Ks, kreg, 1.f, a.Nk, a.D, tid);
        tile_store<DP>(Vs, vreg, 1.f, a.Nk, a.D, tid);
        lds_barrier();
        if (kt + 1 < ntiles) {
            tile_fetch<DP>(kreg, kb, a.ks.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
            tile_fetch<DP>(vreg, vb, a.vs.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
        }
        if (kt + 1 < ntiles) sweep2(kt, std::false_type{}); else sweep2(kt, std::true_type{});
    }
    float* ob = a.o + b * a.os.sb + h * a.os.sh;
#pragma unroll
    for (int dt = 0; dt < NB; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rows[r] < a.Nq && 16 * dt + i < a.D)
                ob[static_cast<int64_t>(rows[r]) * a.os.sn + 16 * dt + i] = oacc[0][dt][r] + oacc[1][dt][r];
}

// =============================================================================================== forward, small grids
// One shared forward (DETR's K kept queries of ONE image: B = 1, 8 heads, 950 tokens) gives the kernel above 15 x 8 = 120
// workgroups, each walking all 15 key tiles twice behind workgroup barriers: 67 us for 0.9 GFLOP.  Here a workgroup owns 16
// query rows and its four waves SPLIT THE KEYS (wave w takes tiles w, w + 4, ...), each staging its own tiles in a private
// LDS region (no workgroup barrier inside the sweeps; a wave's LDS traffic is in order).  The per-wave softmax statistics
// and partial O = P.V are merged through LDS: 4x the workgroups, a quarter of the serial tile chain each.  Exact fp32, fp32
// slabs; head_dim <= 64 (ViT-B/16 at one image: 12 heads x 197 tokens = 48 workgroups of the kernel above).
template <int DP>
struct WaveTileRegs {
    f32x4 raw[DP / 4];
    int row0;
};
template <int DP>
__device__ __forceinline__ void wave_tile_fetch(WaveTileRegs<DP>& reg, const float* base, int64_t sn, int row0, int rows_total,
                                                int D, int lane) {
    constexpr int C4 = DP / 4;
    reg.row0 = row0;
#pragma unroll
    for (int e = 0; e < DP / 4; ++e) {
        const int f = lane + 64 * e;
        const int row = row0 + f / C4, c = (f % C4) * 4;
        const bool ok = row < rows_total && c < D;
        reg.raw[e] = *reinterpret_cast<const f32x4*>(base + (ok ? static_cast<int64_t>(row) * sn + c : 0));
    }
}
template <int DP>
__device__ __forceinline__ void wave_tile_store(float* lds, const WaveTileRegs<DP>& reg, int rows_total, int D, int lane) {
    constexpr int C4 = DP / 4, LS = DP + 4;
#pragma unroll
    for (int e = 0; e < DP / 4; ++e) {
        const int f = lane + 64 * e;
        const bool ok = reg.row0 + f / C4 < rows_total && (f % C4) * 4 < D;
        *reinterpret_cast<f32x4*>(lds + (f / C4) * LS + (f % C4) * 4) = ok ? reg.raw[e] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_wave_barrier();          // keep the compiler from moving this wave's tile reads above the stores
}

// (the partial-O exchange at the end reuses the tile regions: a 64-wide head's four K + V regions alone are 153 KB of the 160)
template <int DP>
constexpr size_t split_lds_bytes() {
    return sizeof(float) * (4 * (2 * kTile * (DP + 4) + 16 * kPS) + 4 * 16 * 2);
}

template <int DP>
__global__ __launch_bounds__(kThreads, 1) void attn_fwd_split_kernel(const AttnFwdArgs a) {
    constexpr int LS = DP + 4, NB = DP / 16, kWave = 2 * kTile * LS + 16 * kPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    float* Ks = smem + wave * kWave;
    float* Vs = Ks + kTile * LS;
    float* Pw = Vs + kTile * LS;
    float* stat = smem + 4 * kWave;                       // [4 waves][16 rows][max, sum]
    float* opart = smem;                                  // [4 waves][16 rows][DP], over the tile regions once the sweeps are done
    const int nrt = (a.Nq + 15) / 16;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int h = (wg / nrt) % a.H, b = wg / (nrt * a.H);
    const int rw = (wg % nrt) * 16;                       // the 16 query rows of this workgroup (all four waves)
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const float ninf = -__builtin_inff();

    f32x4 qa[NB];
    load_a_rows<DP>(qa, qb, a.qs.sn, min(rw + i, a.Nq - 1), a.D, g, q_first ? a.scale : 1.f);
    int rows[4];
    const float* mrow[4];
    float* pout[4];
    const int64_t pbase = (static_cast<int64_t>(b) * a.H + h) * a.Nq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = rw + 4 * g + r;
        mrow[r] = a.mask ? a.mask + b * a.mask_sb + static_cast<int64_t>(min(rows[r], a.Nq - 1)) * a.mask_sq : nullptr;
        pout[r] = rows[r] < a.Nq ? a.probs + (pbase + rows[r]) * a.Nk + i : nullptr;
    }
    auto score = [&](float sc, int r, int k0) {           // every tile takes the key-range test here (few tiles per wave)
        if (!q_first) sc = sc / a.scale;
        if (a.mask) sc += mrow[r][min(k0 + i, a.Nk - 1)];
        return k0 + i < a.Nk ? sc : ninf;
    };
    const int ntiles = (a.Nk + kTile - 1) / kTile;
    WaveTileRegs<DP> kreg, vreg;

    // ---- sweep 1 over this wave's tiles: lane-local running max / sum
    float m[4] = {ninf, ninf, ninf, ninf}, l[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < ntiles) wave_tile_fetch<DP>(kreg, kb, a.ks.sn, wave * kTile, a.Nk, a.D, lane);
    for (int kt = wave; kt < ntiles; kt += 4) {
        wave_tile_store<DP>(Ks, kreg, a.Nk, a.D, lane);
        if (kt + 4 < ntiles) wave_tile_fetch<DP>(kreg, kb, a.ks.sn, (kt + 4) * kTile, a.Nk, a.D, lane);
        f32x4 sacc[4];
        tile_abt4<DP>(sacc, qa, Ks, i, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sv[t] = score(sacc[t][r], r, kt * kTile + 16 * t);
            const float mn = fmaxf(fmaxf(m[r], fmaxf(sv[0], sv[1])), fmaxf(sv[2], sv[3]));
            const float base = (mn == ninf) ? 0.f : mn;
            l[r] = l[r] * exp_fast(m[r] - base) + exp_fast(sv[0] - base) + exp_fast(sv[1] - base) +
                   exp_fast(sv[2] - base) + exp_fast(sv[3] - base);
            m[r] = mn;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // this wave's statistics per row, then the four waves' (a fully masked row ends as 0 * inf = NaN like torch.softmax)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mall = group16_max(m[r]);
        const float base = (mall == ninf) ? 0.f : mall;
        const float lw = group16_sum(l[r] * exp_fast(m[r] - base));
        if (i == 0) {
            stat[(wave * 16 + 4 * g + r) * 2] = mall;
            stat[(wave * 16 + 4 * g + r) * 2 + 1] = lw;
        }
    }
    __syncthreads();
    float linv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mw[4], mall = ninf;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = stat[(w * 16 + 4 * g + r) * 2];
            mall = fmaxf(mall, mw[w]);
        }
        const float base = (mall == ninf) ? 0.f : mall;
        float lsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float bw = (mw[w] == ninf) ? 0.f : mw[w];
            lsum += stat[(w * 16 + 4 * g + r) * 2 + 1] * exp_fast(bw - base);
        }
        linv[r] = 1.f / lsum;
        m[r] = base;
    }

    // ---- sweep 2 over the same tiles: P -> capture slab, partial O += P.V
    f32x4 oacc[2][NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) oacc[0][dt] = oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave < ntiles) {
        wave_tile_fetch<DP>(kreg, kb, a.ks.sn, wave * kTile, a.Nk, a.D, lane);
        wave_tile_fetch<DP>(vreg, vb, a.vs.sn, wave * kTile, a.Nk, a.D, lane);
    }
    for (int kt = wave; kt < ntiles; kt += 4) {
        wave_tile_store<DP>(Ks, kreg, a.Nk, a.D, lane);
        wave_tile_store<DP>(Vs, vreg, a.Nk, a.D, lane);
        if (kt + 4 < ntiles) {
            wave_tile_fetch<DP>(kreg, kb, a.ks.sn, (kt + 4) * kTile, a.Nk, a.D, lane);
            wave_tile_fetch<DP>(vreg, vb, a.vs.sn, (kt + 4) * kTile, a.Nk, a.D, lane);
        }
        f32x4 sacc[4];
        tile_abt4<DP>(sacc, qa, Ks, i, g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = kt * kTile + 16 * t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = exp_fast(score(sacc[t][r], r, k0) - m[r]) * linv[r];
                if (k0 + i >= a.Nk) p = 0.f;                                // (also keeps NaN rows out of the padding)
                if (pout[r] && k0 + i < a.Nk) pout[r][k0] = p;
                Pw[(4 * g + r) * kPS + 16 * t + i] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();
        tile_wt<DP>(oacc, Pw, Vs, i, g);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                      // every wave is done with its tiles: their LDS becomes the exchange
#pragma unroll
    for (int dt = 0; dt < NB; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) opart[(wave * 16 + 4 * g + r) * DP + 16 * dt + i] = oacc[0][dt][r] + oacc[1][dt][r];
    __syncthreads();
    float* ob = a.o + b * a.os.sb + h * a.os.sh;
    for (int idx = tid; idx < 16 * DP; idx += kThreads) {
        const int row = idx / DP, d = idx % DP;
        if (rw + row < a.Nq && d < a.D)
            ob[static_cast<int64_t>(rw + row) * a.os.sn + d] = (opart[row * DP + d] + opart[(16 + row) * DP + d]) +
                                                              (opart[(32 + row) * DP + d] + opart[(48 + row) * DP + d]);
    }
}

// =============================================================================================== backward, query side
// dP = dO.V^T -> capture slab;  delta = rowsum(P * dP) -> workspace;  dS = P * (dP - delta);  dQ = dS.K
// REL (row-relevancy mode, see AttnBwdArgs::rel_v): the product the relevancy rules need from this layer -- one ROW of
// R + A_bar.R, i.e. v.A_bar with A_bar = mean_h clamp(dP * P, 0) -- is reduced here from the dP / P values the sweep
// already holds, so dP is neither stored nor re-read and no A_bar matrix exists.
template <int DP, int DT, bool MM, bool REL = false, bool IOH = false>
__global__ __launch_bounds__(kThreads, DP == 32 ? 3 : 2) void attn_bwd_q_stream_kernel(const AttnBwdArgs a) {
    typedef typename slab_elem<DT>::type slab_t;
    constexpr int LS = DP + 4, NB = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    float* Vs = smem;
    float* Ks = Vs + kTile * LS;
    float* Sw = Ks + kTile * LS + wave * 16 * kPS;
    const int nrt = (a.Nq + kRows - 1) / kRows;           // 1-D grid, XCD-contiguous logical order (see forward)
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int h = (wg / nrt) % a.H, b = wg / (nrt * a.H);
    const int rw = (wg % nrt) * kRows + wave * 16;
    const float* kb = a.need_dqkv ? a.k + b * a.ks.sb + h * a.ks.sh : nullptr;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const float* dob = a.dout + b * a.os.sb + h * a.os.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;

    f32x4 doa[NB];
    if constexpr (IOH)
        load_a_rows_h<DP>(doa, reinterpret_cast<const unsigned short*>(a.dout) + b * a.os.sb + h * a.os.sh, a.os.sn,
                          min(rw + i, a.Nq - 1), a.D, g);
    else
        load_a_rows<DP>(doa, dob, a.os.sn, min(rw + i, a.Nq - 1), a.D, g, 1.f);
    bf16x8 doa_pk[DP / 32];
    if constexpr (MM) pack_a_rows<DP>(doa_pk, doa);
    auto dp_tile = [&](f32x4 (&dp)[4]) {
        if constexpr (MM) tile_abt4_bf16<DP, false>(dp, doa_pk, Vs, i, g);
        else tile_abt4<DP>(dp, doa, Vs, i, g);
    };

    // per-lane row pointers incl. this lane's key column i; the sweeps add wave-uniform key offsets only
    int rows[4];
    const slab_t* prow[4];
    slab_t* dpout[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = rw + 4 * g + r;
        prow[r] = reinterpret_cast<const slab_t*>(a.probs) + b * a.probs_sb +
                  (static_cast<int64_t>(h) * a.Nq + min(rows[r], a.Nq - 1)) * a.Nk + i;
        dpout[r] = (rows[r] < a.Nq && a.dprobs) ? reinterpret_cast<slab_t*>(a.dprobs) + (head * a.Nq + rows[r]) * a.Nk + i
                                                : nullptr;
    }
    float vrow[4] = {0.f, 0.f, 0.f, 0.f};
    float* relw = smem + 2 * kTile * LS + 4 * 16 * kPS;      // REL: [4 waves x 4 row groups][64 keys] partial sums
    if constexpr (REL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vrow[r] = rows[r] < a.Nq ? a.rel_v[static_cast<int64_t>(b) * a.Nq + rows[r]] : 0.f;
    }
    // REL: keys of tile kt summed over the workgroup's 64 query rows (fixed order: deterministic) -> this workgroup's row
    auto rel_flush = [&](int kt) {
        const int kk = kt * kTile + tid;
        if (tid < kTile && kk < a.Nk) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += relw[j * kTile + tid];
            a.rel_part[(head * nrt + wg % nrt) * a.Nk + kk] = sum;
        }
    };
    const int ntiles = (a.Nk + kTile - 1) / kTile;
    TileRegs<DP> kreg, vreg;
    // this lane's 4 x 4 probabilities of a key tile.  EDGE = the tile may run past Nk (last tile only)
    auto load_p = [&](float (&p)[4][4], int kt, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = kt * kTile + 16 * t;
            const bool ok = !EDGE || k0 + i < a.Nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = slab_load<DT>(prow[r] + (EDGE ? min(k0, a.Nk - 1 - i) : k0));   // clamped, unconditional
                p[t][r] = ok ? v : 0.f;
            }
        }
    };

    float delta[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.need_dqkv && a.o) {
        // delta = rowsum(P * dP) = rowsum(dO * O)  (sum_k P_k (dO . V_k) = dO . sum_k P_k V_k): with the forward's O at
        // hand no sweep over the keys is needed.  Lane (row i, slot g) holds dO[row i][its d's]; the four slots of a
        // row are added with two cross-row shuffles, then the C-layout rows 4g + r pick their value up.
        const float* ob = a.o + b * a.oos.sb + h * a.oos.sh;
        f32x4 oa[NB];
        load_a_rows<DP>(oa, ob, a.oos.sn, min(rw + i, a.Nq - 1), a.D, g, 1.f);
        float part = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
            part += doa[blk][0] * oa[blk][0] + doa[blk][1] * oa[blk][1] + doa[blk][2] * oa[blk][2] + doa[blk][3] * oa[blk][3];
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);                         // every lane with lane & 15 == i now holds delta of row rw + i
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            delta[r] = __shfl(part, 4 * g + r);
            if (i == 0 && rows[r] < a.Nq) a.delta[head * a.Nq + rows[r]] = delta[r];
        }
    } else if (a.need_dqkv) {
        // ---- sweep 1: delta (dP is recomputed in sweep 2 instead of being read back)
        auto sweep1 = [&](int kt, auto edge) {
            float p[4][4];
            load_p(p, kt, edge);
            f32x4 dp[4];
            dp_tile(dp);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) delta[r] += p[t][r] * dp[t][r];
        };
        tile_fetch<DP>(vreg, vb, a.vs.sn, 0, a.Nk, a.D, tid);
        for (int kt = 0; kt < ntiles; ++kt) {
            lds_barrier();
            tile_store<DP>(Vs, vreg, 1.f, a.Nk, a.D, tid);
            lds_barrier();
            if (kt + 1 < ntiles) tile_fetch<DP>(vreg, vb, a.vs.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
            if (kt + 1 < ntiles) sweep1(kt, std::false_type{}); else sweep1(kt, std::true_type{});
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            delta[r] = group16_sum(delta[r]);
            if (i == 0 && rows[r] < a.Nq) a.delta[head * a.Nq + rows[r]] = delta[r];
        }
    }

    // ---- sweep 2: dP -> capture slab; dS -> wave tile; dQ += dS.K
    f32x4 qacc[2][NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) qacc[0][dt] = qacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ds_mul = q_first ? 1.f : 1.f / a.scale;      // exact for the power-of-two sqrt(d) of d = 16, 64
    auto sweep2 = [&](int kt, const float (&p)[4][4], auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        f32x4 dp[4];
        dp_tile(dp);
        if constexpr (REL) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float c = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) c += vrow[r] * relu_nan(p[t][r] * dp[t][r]);
                relw[(wave * 4 + g) * kTile + 16 * t + i] = c;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = kt * kTile + 16 * t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (dpout[r] && (!EDGE || k0 + i < a.Nk)) slab_store<DT>(dpout[r] + k0, dp[t][r]);
                if (a.need_dqkv) Sw[(4 * g + r) * kPS + 16 * t + i] = p[t][r] * (dp[t][r] - delta[r]) * ds_mul;
            }
        }
        if (a.need_dqkv) {
            if constexpr (MM) tile_wt_bf16<DP>(qacc, Sw, Ks, i, g);
            else tile_wt<DP>(qacc, Sw, Ks, i, g);
        }
    };
    tile_fetch<DP>(vreg, vb, a.vs.sn, 0, a.Nk, a.D, tid);
    if (a.need_dqkv) tile_fetch<DP>(kreg, kb, a.ks.sn, 0, a.Nk, a.D, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        lds_barrier();
        if constexpr (REL) { if (kt > 0) rel_flush(kt - 1); }
        tile_store<DP>(Vs, vreg, 1.f, a.Nk, a.D, tid);
        if (a.need_dqkv) tile_store<DP>(Ks, kreg, 1.f, a.Nk, a.D, tid);
        lds_barrier();
        // this tile's probabilities first, the next tile's K / V after them: vmcnt retires in order, so the wait in front
        // of the first use of p then leaves the prefetch in flight (issued the other way round it drained everything)
        float p[4][4];
        if (a.need_dqkv || REL) { if (kt + 1 < ntiles) load_p(p, kt, std::false_type{}); else load_p(p, kt, std::true_type{}); }
        if (kt + 1 < ntiles) {
            tile_fetch<DP>(vreg, vb, a.vs.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
            if (a.need_dqkv) tile_fetch<DP>(kreg, kb, a.ks.sn, (kt + 1) * kTile, a.Nk, a.D, tid);
        }
        if (kt + 1 < ntiles) sweep2(kt, p, std::false_type{}); else sweep2(kt, p, std::true_type{});
    }
    if constexpr (REL) {
        lds_barrier();
        rel_flush(ntiles - 1);
    }
    if (!a.need_dqkv) return;
    const int64_t dq0 = b * a.dqs.sb + h * a.dqs.sh;
    const float mul = q_first ? a.scale : 1.f;
    if constexpr (IOH) {
#pragma unroll
        for (int dt = 0; dt < NB; ++dt)
            store_rows_bf16_pairs(a.dq, dq0, a.dqs.sn, 16 * dt + i, a.D, (qacc[0][dt] + qacc[1][dt]) * mul, i,
                                  [&](int r) { return rw + 4 * g + r < a.Nq; },
                                  [&](int r) { return static_cast<int64_t>(rw + 4 * g + r); });
    } else {
#pragma unroll
        for (int dt = 0; dt < NB; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rows[r] < a.Nq && 16 * dt + i < a.D)
                    a.dq[dq0 + static_cast<int64_t>(rows[r]) * a.dqs.sn + 16 * dt + i] = (qacc[0][dt][r] + qacc[1][dt][r]) * mul;
    }
}

// =============================================================================================== backward, key side
// per 64 keys (16 per wave): dV = P^T.dO, dK = dS^T.Q with dS rebuilt from the two capture slabs and delta.
// The A operands (columns of P / dS) come straight from the slabs in MFMA layout: lane (key i, slot g) reads rows
// 16 rb + 4 g + s of key column j0 + i -- 16 consecutive keys per row segment, no LDS staging.
// MM (bf16 MFMA): dP of the tile is RECOMPUTED (dO tile from LDS . this wave's V rows held in registers -- two bf16
// MFMAs per 16 queries) instead of being read back from the slab the query-side kernel just wrote: one N^2 read less.
template <int DP, int DT, bool MM, bool IOH = false>
__global__ __launch_bounds__(kThreads) void attn_bwd_kv_stream_kernel(const AttnBwdArgs a) {
    typedef typename slab_elem<DT>::type slab_t;
    constexpr int LS = DP + 4, NB = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    float* Qs = smem;
    float* dOs = Qs + kTile * LS;
    float* dl = dOs + kTile * LS;                             // [64] delta of the staged query rows
    const int nkt = (a.Nk + kRows - 1) / kRows;           // 1-D grid, XCD-contiguous: a head's key tiles share Q / dO
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int h = (wg / nkt) % a.H, b = wg / (nkt * a.H);
    const int kw = (wg % nkt) * kRows + wave * 16;          // first key of this wave
    const int key = kw + i;
    const bool key_ok = key < a.Nk;
    const int keyc = min(key, a.Nk - 1);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* dob = a.dout + b * a.os.sb + h * a.os.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    const slab_t* pcol = reinterpret_cast<const slab_t*>(a.probs) + b * a.probs_sb + static_cast<int64_t>(h) * a.Nq * a.Nk + keyc;
    const slab_t* dpcol = reinterpret_cast<const slab_t*>(a.dprobs) + head * a.Nq * a.Nk + keyc;

    f32x4 kacc[NB], vacc[NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) kacc[dt] = vacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float ds_mul = q_first ? 1.f : 1.f / a.scale;
    bf16x8 v_pk[DP / 32];
    if constexpr (MM) {
        f32x4 va[NB];
        load_a_rows<DP>(va, a.v + b * a.vs.sb + h * a.vs.sh, a.vs.sn, keyc, a.D, g, 1.f);
        pack_a_rows<DP>(v_pk, va);
    }

    const int ntiles = (a.Nq + kTile - 1) / kTile;
    TileRegs<DP> qreg;
    TileRegs<DP, IOH> doreg;
    float dlreg = 0.f;
    auto fetch = [&](int qt) {
        tile_fetch<DP>(qreg, qb, a.qs.sn, qt * kTile, a.Nq, a.D, tid);
        if constexpr (IOH)
            tile_fetch<DP>(doreg, reinterpret_cast<const unsigned short*>(a.dout) + b * a.os.sb + h * a.os.sh, a.os.sn,
                           qt * kTile, a.Nq, a.D, tid);
        else
            tile_fetch<DP>(doreg, dob, a.os.sn, qt * kTile, a.Nq, a.D, tid);
        if (tid < kTile) {
            const int row = qt * kTile + tid;
            const float v = a.delta[head * a.Nq + min(row, a.Nq - 1)];
            dlreg = row < a.Nq ? v : 0.f;
        }
    };
    fetch(0);
    for (int qt = 0; qt < ntiles; ++qt) {
        // this tile's slab columns: issued before the barriers so they overlap the LDS staging
        float p[4][4], dp[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = qt * kTile + 16 * rb + 4 * g + s;
                const int64_t off = static_cast<int64_t>(min(row, a.Nq - 1)) * a.Nk;
                const bool ok = key_ok && row < a.Nq;
                const float pv = slab_load<DT>(pcol + off);
                p[rb][s] = ok ? pv : 0.f;
                if constexpr (!MM) {
                    const float dv = slab_load<DT>(dpcol + off);
                    dp[rb][s] = ok ? dv : 0.f;
                }
            }
        lds_barrier();
        tile_store<DP>(Qs, qreg, q_first ? a.scale : 1.f, a.Nq, a.D, tid);
        tile_store<DP>(dOs, doreg, 1.f, a.Nq, a.D, tid);
        if (tid < kTile) dl[tid] = dlreg;
        lds_barrier();
        if (qt + 1 < ntiles) fetch(qt + 1);
        if constexpr (MM) {
            f32x4 dpt[4];                                      // dpt[rb][r] = dP[16 rb + 4 g + r][key i]  (rows past Nq: dO = 0)
            tile_abt4_bf16<DP, true>(dpt, v_pk, dOs, i, g);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 dsv[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int rb = 2 * pr + hh;
                    const f32x4 dlv = *reinterpret_cast<const f32x4*>(dl + 16 * rb + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) dsv[hh][s] = p[rb][s] * (dpt[rb][s] - dlv[s]) * ds_mul;
                }
                const bf16x8 p_pk = pack_bf16(f32x4{p[2 * pr][0], p[2 * pr][1], p[2 * pr][2], p[2 * pr][3]},
                                              f32x4{p[2 * pr + 1][0], p[2 * pr + 1][1], p[2 * pr + 1][2], p[2 * pr + 1][3]});
                const bf16x8 ds_pk = pack_bf16(dsv[0], dsv[1]);
                const float* qrow = Qs + (32 * pr + 4 * g) * LS + i;
                const float* drow = dOs + (32 * pr + 4 * g) * LS + i;
#pragma unroll
                for (int dt = 0; dt < NB; ++dt) {
                    f32x4 qlo, qhi, dlo, dhi;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        qlo[s] = qrow[s * LS + 16 * dt]; qhi[s] = qrow[(16 + s) * LS + 16 * dt];
                        dlo[s] = drow[s * LS + 16 * dt]; dhi[s] = drow[(16 + s) * LS + 16 * dt];
                    }
                    vacc[dt] = mfma16x16x32_bf16(p_pk, pack_bf16(dlo, dhi), vacc[dt]);
                    kacc[dt] = mfma16x16x32_bf16(ds_pk, pack_bf16(qlo, qhi), kacc[dt]);
                }
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const f32x4 dlv = *reinterpret_cast<const f32x4*>(dl + 16 * rb + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float ds = p[rb][s] * (dp[rb][s] - dlv[s]) * ds_mul;
                    const float* qrow = Qs + (16 * rb + 4 * g + s) * LS + i;
                    const float* drow = dOs + (16 * rb + 4 * g + s) * LS + i;
#pragma unroll
                    for (int dt = 0; dt < NB; ++dt) {
                        vacc[dt] = mfma16x16x4(p[rb][s], drow[16 * dt], vacc[dt]);
                        kacc[dt] = mfma16x16x4(ds, qrow[16 * dt], kacc[dt]);
                    }
                }
            }
        }
    }
    const int64_t dk0 = b * a.dks.sb + h * a.dks.sh, dv0 = b * a.dvs.sb + h * a.dvs.sh;
    if constexpr (IOH) {
        auto ok = [&](int r) { return kw + 4 * g + r < a.Nk; };
        auto idx = [&](int r) { return static_cast<int64_t>(kw + 4 * g + r); };
#pragma unroll
        for (int dt = 0; dt < NB; ++dt) {
            store_rows_bf16_pairs(a.dk, dk0, a.dks.sn, 16 * dt + i, a.D, kacc[dt], i, ok, idx);
            store_rows_bf16_pairs(a.dv, dv0, a.dvs.sn, 16 * dt + i, a.D, vacc[dt], i, ok, idx);
        }
    } else {
#pragma unroll
        for (int dt = 0; dt < NB; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kw + 4 * g + r, d = 16 * dt + i;
                if (j < a.Nk && d < a.D) {
                    a.dk[dk0 + static_cast<int64_t>(j) * a.dks.sn + d] = kacc[dt][r];
                    a.dv[dv0 + static_cast<int64_t>(j) * a.dvs.sn + d] = vacc[dt][r];
                }
            }
    }
}

bool aligned16(const float* p, const Strides& s) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && s.sb % 4 == 0 && s.sh % 4 == 0 && s.sn % 4 == 0;
}

template <typename K, typename A>
int launch_stream(K kern, const A& args, dim3 grid, size_t lds, hipStream_t s, const char* name) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    kern<<<grid, kThreads, lds, s>>>(args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MMX_OK;
}

template <int DT, bool MM>
int launch_fwd_mm(const AttnFwdArgs& a, dim3 grid, hipStream_t s) {
    return a.D <= 32
        ? launch_stream(attn_fwd_stream_kernel<32, DT, MM>, a, grid, stream_lds_bytes<32>(2), s, "attn_fwd_stream_kernel<32>")
        : launch_stream(attn_fwd_stream_kernel<64, DT, MM>, a, grid, stream_lds_bytes<64>(2), s, "attn_fwd_stream_kernel<64>");
}

template <int DT>
int launch_fwd_dt(const AttnFwdArgs& a, dim3 grid, hipStream_t s) {
    return a.mma_bf16 ? launch_fwd_mm<DT, true>(a, grid, s) : launch_fwd_mm<DT, false>(a, grid, s);
}

template <int DT, bool MM>
int launch_bwd_mm(const AttnBwdArgs& a, dim3 gq, dim3 gk, hipStream_t s) {
    const bool small_d = a.D <= 32;
    int rc = small_d ? launch_stream(attn_bwd_q_stream_kernel<32, DT, MM>, a, gq, stream_lds_bytes<32>(2), s,
                                     "attn_bwd_q_stream_kernel<32>")
                     : launch_stream(attn_bwd_q_stream_kernel<64, DT, MM>, a, gq, stream_lds_bytes<64>(2), s,
                                     "attn_bwd_q_stream_kernel<64>");
    if (rc == MMX_OK && a.need_dqkv) {
        // the key-side kernel's LDS: two operand tiles + 64 deltas (the 4 x 16 x 68 floats of the wave tiles cover it)
        rc = small_d ? launch_stream(attn_bwd_kv_stream_kernel<32, DT, MM>, a, gk, stream_lds_bytes<32>(2), s,
                                     "attn_bwd_kv_stream_kernel<32>")
                     : launch_stream(attn_bwd_kv_stream_kernel<64, DT, MM>, a, gk, stream_lds_bytes<64>(2), s,
                                     "attn_bwd_kv_stream_kernel<64>");
    }
    return rc;
}

// bf16-MFMA kernels with the optional modes: REL (row relevancy) and IOH (bf16 gradient stream)
template <int DT, bool REL, bool IOH>
int launch_bwd_bf16(const AttnBwdArgs& a, dim3 gq, dim3 gk, hipStream_t s) {
    const bool small_d = a.D <= 32;
    constexpr size_t kRel = REL ? sizeof(float) * 16 * kTile : 0;
    int rc = small_d ? launch_stream(attn_bwd_q_stream_kernel<32, DT, true, REL, IOH>, a, gq, stream_lds_bytes<32>(2) + kRel, s,
                                     "attn_bwd_q_stream_kernel<32, bf16>")
                     : launch_stream(attn_bwd_q_stream_kernel<64, DT, true, REL, IOH>, a, gq, stream_lds_bytes<64>(2) + kRel, s,
                                     "attn_bwd_q_stream_kernel<64, bf16>");
    if (rc == MMX_OK && a.need_dqkv) {
        rc = small_d ? launch_stream(attn_bwd_kv_stream_kernel<32, DT, true, IOH>, a, gk, stream_lds_bytes<32>(2), s,
                                     "attn_bwd_kv_stream_kernel<32, bf16>")
                     : launch_stream(attn_bwd_kv_stream_kernel<64, DT, true, IOH>, a, gk, stream_lds_bytes<64>(2), s,
                                     "attn_bwd_kv_stream_kernel<64, bf16>");
    }
    return rc;
}

template <int DT>
int launch_bwd_dt(const AttnBwdArgs& a, dim3 gq, dim3 gk, hipStream_t s) {
    if (!a.mma_bf16) return launch_bwd_mm<DT, false>(a, gq, gk, s);
    if (a.rel_v) {
        int rc = a.io_bf16 ? launch_bwd_bf16<DT, true, true>(a, gq, gk, s) : launch_bwd_bf16<DT, true, false>(a, gq, gk, s);
        if (rc) return rc;
        return rel_row_update(a.rel_v, a.rel_part, a.rel_out, a.B, a.H * ((a.Nq + kRows - 1) / kRows), a.Nk, 1.0f / a.H, s);
    }
    return a.io_bf16 ? launch_bwd_bf16<DT, false, true>(a, gq, gk, s) : launch_bwd_bf16<DT, false, false>(a, gq, gk, s);
}

}  // namespace

void attn_stream_enable(int on) { g_attn_stream = on & 1; }
void attn_fwd_split_enable(int on) { g_attn_fwd_split = on & 1; }

// returns 1 if the streaming kernel was launched (rc in *rc_out), 0 if the shape / layout is not eligible.
// Slabs in fp16 / bf16 (slab_dt) exist on this path only: they ignore the "attn_stream" switch.
int attn_fwd_stream_try(const AttnFwdArgs& a, hipStream_t s, int* rc_out) {
    if ((!g_attn_stream && a.slab_dt == MMX_F32 && !a.mma_bf16) || a.D % 4 || a.D > 64) return 0;
    if (!aligned16(a.q, a.qs) || !aligned16(a.k, a.ks) || !aligned16(a.v, a.vs)) return 0;
    dim3 grid(((a.Nq + kRows - 1) / kRows) * a.H * a.B);
    // a grid that leaves most of the chip idle (one shared forward): 16-row workgroups whose waves split the keys
    if (g_attn_fwd_split && grid.x < 160 && a.slab_dt == MMX_F32 && !a.mma_bf16 && a.Nk > kTile) {
        const dim3 g16(((a.Nq + 15) / 16) * a.H * a.B);
        *rc_out = a.D <= 32 ? launch_stream(attn_fwd_split_kernel<32>, a, g16, split_lds_bytes<32>(), s, "attn_fwd_split_kernel<32>")
                            : launch_stream(attn_fwd_split_kernel<64>, a, g16, split_lds_bytes<64>(), s, "attn_fwd_split_kernel<64>");
        return 1;
    }
    switch (a.slab_dt) {
        case MMX_F32: *rc_out = launch_fwd_dt<MMX_F32>(a, grid, s); break;
        case MMX_F16: *rc_out = launch_fwd_dt<MMX_F16>(a, grid, s); break;
        case MMX_BF16: *rc_out = launch_fwd_dt<MMX_BF16>(a, grid, s); break;
        default: return 0;
    }
    return 1;
}

int attn_bwd_stream_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out) {
    if ((!g_attn_stream && a.slab_dt == MMX_F32 && !a.mma_bf16) || a.D % 4 || a.D > 64) return 0;
    if ((a.rel_v || a.io_bf16) && !a.mma_bf16) return 0;
    if (a.mma_bf16 && attn_bwd_bf16_try(a, s, rc_out)) return 1;      // second-generation bf16 kernels (attention_bf16.hip)
    if (!aligned16(a.v, a.vs) || !aligned16(a.dout, a.os)) return 0;
    if (a.need_dqkv && (!aligned16(a.q, a.qs) || !aligned16(a.k, a.ks))) return 0;
    dim3 gq(((a.Nq + kRows - 1) / kRows) * a.H * a.B), gk(((a.Nk + kRows - 1) / kRows) * a.H * a.B);
    switch (a.slab_dt) {
        case MMX_F32: *rc_out = launch_bwd_dt<MMX_F32>(a, gq, gk, s); break;
        case MMX_F16: *rc_out = launch_bwd_dt<MMX_F16>(a, gq, gk, s); break;
        case MMX_BF16: *rc_out = launch_bwd_dt<MMX_BF16>(a, gq, gk, s); break;
        default: return 0;
    }
    return 1;
}

}  // namespace mmx
