// Register-resident whole-head attention-capture kernels (second generation of the short-sequence path: CLIP 50 / 77
// tokens, LXMERT 14-36, BERT-sized N <= 128 keys).
//
// attention_small.hip keeps Q, K, V, S (and dO, P, dS) of a head in LDS and walks five workgroup-wide phases separated
// by barriers with ONE 512/1024-thread workgroup per CU: every phase waits for the slowest wave and nothing overlaps the
// global loads / stores of a head with the MFMAs of another (rocprofv3 r01: 13-37 % of HBM peak, latency-bound).
// Here a WAVE owns 16 query rows end to end and the score tile never leaves its registers:
//
//   * S^T = K.Q^T is accumulated per 16-key tile: lane l holds S[q = l&15][key = 16t + 4(l>>4) + r] -- four
//     CONSECUTIVE keys of ONE query row.  The softmax reductions are in-lane plus one v_permlane16_swap + one
//     v_permlane32_swap across the four 16-lane rows; P leaves the chip as one 16-byte store per tile, straight into the
//     capture slab.
//   * The contraction order inside an MFMA dot product is free, so the accumulator register P[q][16t + 4g + r] IS the
//     A operand of O = P.V for the k-step (t, r) (B operand V[16t + 4g + r][d]): P is never written to or re-read from
//     LDS.  The same holds for dQ = dS.K in the backward.
//   * Only K and V (backward: V, K, then the transposed operands of dK / dV) live in LDS: 36-49 KB per head, i.e. 3-4
//     workgroups of 4-5 waves per CU, so one head's staging loads and slab stores overlap the MFMAs of the others.
//   * dK = dS^T.Q and dV = P^T.dO contract over the query index, i.e. across waves: dS (then P) is written once to LDS
//     row-major and read back as the A operand of key-tile-owning waves; two passes share one buffer.
//
// LDS strides: an operand read with ds_read_b128 along the contraction index uses a row stride = 8 (mod 16) floats
// (the 16-lane groups of a b128 read are {0-3,12-15,20-27} ...: stride/4 = 2 (mod 4) makes their 16 four-bank slots
// distinct); an operand read with ds_read_b32 at rows 4g + r uses stride = 4 (mod 8) (rows 4 apart land 16 banks apart).
//
// Eligibility (host side; otherwise the older paths run): fp32 slabs, head_dim % 4 == 0 and <= 64, Nk <= 128,
// Nq <= 256, 16-byte aligned q / k / v / dO rows.
#include "mmx_common.h"
#include "attention_args.h"

#include <type_traits>

namespace mmx {

namespace {

// -DMMX_HEAD_TIMELINE (probe builds only, tools/probe_head_timeline.py): every wave of the backward records s_memtime at its phase
// boundaries and lane 0 writes the differences (shader cycles) over row 0 of its head's dV -- the output is garbage in such a build.
#ifdef MMX_HEAD_TIMELINE
#define MMX_TL_DECL long long tl_[10]; int tl_n_ = 0
#define MMX_TL_MARK() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tl_[tl_n_++] = __builtin_readcyclecounter(); } while (0)
#else
#define MMX_TL_DECL
#define MMX_TL_MARK() do { } while (0)
#endif

__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// 16-byte store that only assumes 4-byte alignment (rows of an odd-Nk slab)
__device__ __forceinline__ void stg4_u(float* p, f32x4 v) { reinterpret_cast<f32x4_u*>(p)->v = v; }

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48); every row ends with the same bits.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the
// upper half of the first with the lower half of the second; with both operands = x the two results hold the two
// partners of every lane.
template <bool MAX>
__device__ __forceinline__ float rows4_allreduce(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float u = __uint_as_float(r[0]), v = __uint_as_float(r[1]);
    x = MAX ? fmaxf(u, v) : u + v;
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    u = __uint_as_float(s[0]);
    v = __uint_as_float(s[1]);
    return MAX ? fmaxf(u, v) : u + v;
}

// Rows [0, rows) x [0, D) of TWO equally shaped strided global matrices -> LDS [rows_cap][LS0 / LS1], zero padded.
// All loads of a pass are in flight before the first ds_write and every load is unconditional (clamped address).
template <int DP, int UNR>
__device__ __forceinline__ void stage_pair(float* l0, int LS0, const float* g0, int64_t sn0, float* l1, int LS1,
                                           const float* g1, int64_t sn1, int rows, int rows_cap, int D, int tid,
                                           int nthreads) {
    constexpr int C4 = DP / 4;
    const int total = rows_cap * C4;
    for (int base = tid; base < total; base += nthreads * UNR) {
        f32x4 v0[UNR], v1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = min(base + u * nthreads, total - 1);
            const int r = idx / C4, c = (idx - r * C4) * 4;
            const bool real = r < rows && c < D;
            v0[u] = *reinterpret_cast<const f32x4*>(g0 + (real ? static_cast<int64_t>(r) * sn0 + c : 0));
            v1[u] = *reinterpret_cast<const f32x4*>(g1 + (real ? static_cast<int64_t>(r) * sn1 + c : 0));
            if (!real) v0[u] = v1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int r = idx / C4, c = (idx - r * C4) * 4;
                *reinterpret_cast<f32x4*>(l0 + r * LS0 + c) = v0[u];
                *reinterpret_cast<f32x4*>(l1 + r * LS1 + c) = v1[u];
            }
        }
    }
}

template <int DP, int UNR>
__device__ __forceinline__ void stage_one(float* l0, int LS0, const float* g0, int64_t sn0, int rows, int rows_cap, int D,
                                          int tid, int nthreads) {
    constexpr int C4 = DP / 4;
    const int total = rows_cap * C4;
    for (int base = tid; base < total; base += nthreads * UNR) {
        f32x4 v0[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = min(base + u * nthreads, total - 1);
            const int r = idx / C4, c = (idx - r * C4) * 4;
            const bool real = r < rows && c < D;
            v0[u] = *reinterpret_cast<const f32x4*>(g0 + (real ? static_cast<int64_t>(r) * sn0 + c : 0));
            if (!real) v0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * nthreads;
            if (idx < total) {
                const int r = idx / C4, c = (idx - r * C4) * 4;
                *reinterpret_cast<f32x4*>(l0 + r * LS0 + c) = v0[u];
            }
        }
    }
}

// This wave's 16 rows of a [N, D] strided matrix as MFMA B-operand registers: lane l gets row (row0 + l&15), columns
// 16kk + 4(l>>4) .. +3 (zero beyond N / D), times mul.
template <int DP>
__device__ __forceinline__ void load_rows16(f32x4 (&reg)[DP / 16], const float* base, int64_t sn, int row, int N, int D,
                                            int g, float mul) {
    const bool rv = row < N;
    const float* p = base + (rv ? static_cast<int64_t>(row) * sn : 0);
#pragma unroll
    for (int kk = 0; kk < DP / 16; ++kk) {
        const int d = kk * 16 + 4 * g;
        const bool ok = rv && d < D;
        f32x4 v = *reinterpret_cast<const f32x4*>(p + (ok ? d : 0));
        reg[kk] = ok ? v * mul : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// four consecutive elements of a slab row (any 4-byte alignment), zero beyond n.  The 16-byte load is UNCONDITIONAL (a
// lane without a full chunk reads the row start instead): a load under a lane-divergent branch would make hipcc drain
// vmcnt per chunk; only the one ragged chunk of a row takes the scalar branch.
__device__ __forceinline__ f32x4 load_chunk(const float* row, int k0, int n, bool row_valid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const bool full = row_valid && k0 + 3 < n;
    if (n >= 4) {                                   // wave-uniform
        const f32x4 t = ldg4_u(row + (full ? k0 : 0));
        if (full) v = t;
    }
    if (row_valid && !full && k0 < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (k0 + r < n) v[r] = row[k0 + r];
    }
    return v;
}

__device__ __forceinline__ void store_chunk(float* row, int k0, int n, bool row_valid, f32x4 v) {
    if (row_valid) {
        if (k0 + 3 < n) {
            stg4_u(row + k0, v);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r < n) row[k0 + r] = v[r];
        }
    }
}


// acc[t] += X[16t + c16][.] . breg[.] over the head dimension: the A operand tiles (one ds_read_b128 per (kk, t)) are
// fetched ONE kk-step ahead of the MFMAs that consume them, and consecutive MFMAs go to different accumulators
// (the 16x16x4 fp32 MFMA has 40 cycles of dependent latency against 32 of issue).
// NA <= NTK: only the FIRST NA tiles are computed (the others are known to be masked out / multiplied by exact zeros: see live_tiles)
template <int DP, int NTK, int LS, int NA = NTK>
__device__ __forceinline__ void tiles_kd(f32x4 (&acc)[NTK], const float* Xs, const f32x4 (&breg)[DP / 16], int c16, int g) {
    constexpr int KK = DP / 16;
    f32x4 cur[NA], nxt[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) cur[t] = *reinterpret_cast<const f32x4*>(Xs + (t * 16 + c16) * LS + 4 * g);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        if (kk + 1 < KK) {
#pragma unroll
            for (int t = 0; t < NA; ++t)
                nxt[t] = *reinterpret_cast<const f32x4*>(Xs + (t * 16 + c16) * LS + (kk + 1) * 16 + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE the MFMAs (hipcc sinks LDS reads to their use)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NA; ++t) acc[t] = mfma16x16x4(cur[t][i], breg[kk][i], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < KK) {
#pragma unroll
            for (int t = 0; t < NA; ++t) cur[t] = nxt[t];
        }
    }
}

// out[td] += (A . Y)^T with the A operand straight from accumulator-layout registers: k-step (t, r) pairs areg[t][r] with row
// 16t + 4g + r of Y (LDS, stride LS, ds_read_b32 at columns 16td + c16); the Y operands are fetched one k-step ahead.
// The registers go in as the MFMA's B operand and Y as its A operand, i.e. the tile comes out TRANSPOSED: lane (c16, g) holds
// columns 16td + 4g .. + 3 of ITS OWN row c16 -- four consecutive floats of one output row, one 16-byte store
// (store_rows16) instead of four 4-byte stores to four rows.  Same products, same contraction order.
template <int DP, int NTK, int LS, int NA = NTK>
__device__ __forceinline__ void tiles_from_regs(f32x4 (&out)[DP / 16], const f32x4 (&areg)[NTK], const float* Ys, int c16, int g) {
    constexpr int KK = DP / 16;
    float cur[KK], nxt[KK];
    const float* y0 = Ys + 4 * g * LS + c16;
#pragma unroll
    for (int td = 0; td < KK; ++td) cur[td] = y0[td * 16];
#pragma unroll
    for (int s = 0; s < NA * 4; ++s) {
        if (s + 1 < NA * 4) {
            const float* yn = y0 + (((s + 1) >> 2) * 16 + ((s + 1) & 3)) * LS;
#pragma unroll
            for (int td = 0; td < KK; ++td) nxt[td] = yn[td * 16];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int td = 0; td < KK; ++td) out[td] = mfma16x16x4(cur[td], areg[s >> 2][s & 3], out[td]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NA * 4) {
#pragma unroll
            for (int td = 0; td < KK; ++td) cur[td] = nxt[td];
        }
    }
}

// Masked-tile skip (round 6).  A 16-key tile whose probabilities are exact zeros for all 16 query rows of a wave -- the keys above the
// diagonal of a causally masked tower (CLIP's text tower: 10 of 25 tiles at 77 tokens), the padding keys of a padded batch -- contributes
// exact zeros to O = P.V, dQ = dS.K, dK = dS^T.Q and dV = P^T.dO (finite operands), and its scores end at -inf whatever K.Q^T says.  A wave
// finds its LAST tile that is not like that (from the mask chunks in the forward, from the P chunks in the backward; wave-uniform) and runs
// the products of the tiles up to it only; the instantiation for that count is picked by a wave-uniform switch.  Same bits as the full
// products; dP = dO.V^T, which the reference exposes as `attn_grad`, stays dense.  a.tile_skip = 0 switches it off (A / B runs).
template <int NTK, typename F>
__device__ __forceinline__ void with_tile_count(int na, F&& f) {
    switch (na) {
#define MMX_NA_CASE(K) case K: if constexpr (K < NTK) { f(std::integral_constant<int, K>{}); break; }
        MMX_NA_CASE(1) MMX_NA_CASE(2) MMX_NA_CASE(3) MMX_NA_CASE(4) MMX_NA_CASE(5) MMX_NA_CASE(6) MMX_NA_CASE(7)
#undef MMX_NA_CASE
        default: f(std::integral_constant<int, NTK>{});
    }
}

// Transposed-layout tiles (tiles_from_regs, phase C of the backward) -> global: row `row` of a strided [N, D] matrix, columns
// 16td + 4g .. + 3 per tile, times mul.  D % 4 == 0 (eligibility), so a chunk is either whole or beyond D.
template <int DP>
__device__ __forceinline__ void store_rows16(float* base, int64_t sn, int row, bool row_valid, int D, int g,
                                             const f32x4 (&acc)[DP / 16], float mul) {
    if (!row_valid) return;
    float* p = base + static_cast<int64_t>(row) * sn + 4 * g;
#pragma unroll
    for (int td = 0; td < DP / 16; ++td)
        if (td * 16 + 4 * g < D) stg4_u(p + td * 16, acc[td] * mul);
}

// bf16 gradient I/O (IOH): the upstream gradient dO arrives as bf16 and dQ / dK / dV leave as bf16 (round to nearest even) -- the
// bf16 gradient stream of a bf16 body (clip_model.backward_tape) through a SHORT tower: same exact-fp32 arithmetic in between, the two
// conversion passes around the kernel (dO -> fp32, dq | dk | dv -> bf16: 33 launches and 4.5 ms per cfg-5 step) are gone.
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
template <int DP>
__device__ __forceinline__ void load_rows16_bf16(f32x4 (&reg)[DP / 16], const unsigned short* base, int64_t sn, int row, int N,
                                                 int D, int g) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bool rv = row < N;
    const unsigned short* p = base + (rv ? static_cast<int64_t>(row) * sn : 0);
#pragma unroll
    for (int kk = 0; kk < DP / 16; ++kk) {
        const int d = kk * 16 + 4 * g;
        const bool ok = rv && d < D;
        const u32x2 w = *reinterpret_cast<const u32x2*>(p + (ok ? d : 0));
        reg[kk] = ok ? f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16),
                             __uint_as_float(w[1] & 0xffff0000u)}
                     : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int DP>
__device__ __forceinline__ void store_rows16_bf16(unsigned short* base, int64_t sn, int row, bool row_valid, int D, int g,
                                                  const f32x4 (&acc)[DP / 16], float mul) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if (!row_valid) return;
    unsigned short* p = base + static_cast<int64_t>(row) * sn + 4 * g;
#pragma unroll
    for (int td = 0; td < DP / 16; ++td)
        if (td * 16 + 4 * g < D) {
            const f32x4 v = acc[td] * mul;
            *reinterpret_cast<u32x2*>(p + td * 16) = u32x2{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        }
}

// Dispatch "round" r of this workgroup (how many workgroups the hardware has probably placed on the same CU before it):
// round r > 0 waits r * units * 64 cycles before issuing its first load, so that a CU's co-resident workgroups are in
// different phases (loading / MFMA / storing) instead of marching through them in lockstep.  A pure timing hint.
__device__ __forceinline__ void stagger(int units) {
    if (units <= 0) return;
    const int linear = blockIdx.y * gridDim.x + blockIdx.x;
    const int round = linear / 256;
    for (int i = 0; i < round * units; ++i) __builtin_amdgcn_s_sleep(1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------- forward
// grid (H, B), blockDim = 64 * ceil(Nq / 16): wave w owns query rows [16w, 16w + 16).
template <int DP, int NTK>
__global__ __launch_bounds__(1024) void attn_fwd_head_kernel(const AttnFwdArgs a) {
    constexpr int LSK = DP + 8, LSV = DP + 4, KK = DP / 16, NPk = NTK * 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                 // [NPk][LSK]  A operand of S^T (ds_read_b128 along d)
    float* Vs = Ks + NPk * LSK;       // [NPk][LSV]  B operand of O   (ds_read_b32 at rows 4g + r)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int h = blockIdx.x, b = blockIdx.y;
    const int c16 = lane & 15, g = lane >> 4;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int q = wave * 16 + c16;
    const bool qv = q < a.Nq;

    stagger(a.debug >> 8);
    const int skip = a.debug & 0xff;     // profiling only: 1 no loads, 2 no S MFMA, 4 no P store, 8 no PV MFMA, 16 no O store
    f32x4 qreg[KK];
    if (!(skip & 1)) {
        load_rows16<DP>(qreg, a.q + b * a.qs.sb + h * a.qs.sh, a.qs.sn, q, a.Nq, a.D, g, q_first ? a.scale : 1.f);
        stage_pair<DP, 4>(Ks, LSK, a.k + b * a.ks.sb + h * a.ks.sh, a.ks.sn, Vs, LSV, a.v + b * a.vs.sb + h * a.vs.sh,
                          a.vs.sn, a.Nk, NPk, a.D, tid, nthreads);
    } else {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qreg[kk] = f32x4{0.01f * lane, 0.02f, 0.03f, 0.04f};
    }
    // additive mask chunks of this lane's (q, keys): in flight while the S^T MFMAs run
    f32x4 mk[NTK];
    {
        const float* mrow = a.mask ? a.mask + b * a.mask_sb + static_cast<int64_t>(qv ? q : 0) * a.mask_sq : nullptr;
#pragma unroll
        for (int t = 0; t < NTK; ++t)
            mk[t] = mrow ? load_chunk(mrow, t * 16 + 4 * g, a.Nk, true) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();

    // S^T tiles: acc[t][r] = S[q][key = 16t + 4g + r]
    f32x4 acc[NTK];
#pragma unroll
    for (int t = 0; t < NTK; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tiles beyond `na` are masked with -inf for every valid row of this wave (with_tile_count): their scores are -inf without the MFMAs
    int na = NTK;
    if (a.mask && a.tile_skip) {
        na = 1;
#pragma unroll
        for (int t = 1; t < NTK; ++t) {
            bool dead = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) dead = dead && (mk[t][r] == -__builtin_inff() || t * 16 + 4 * g + r >= a.Nk);
            if (!__all(dead || !qv)) na = t + 1;
        }
    }
    if (!(skip & 2))
        with_tile_count<NTK>(na, [&](auto n) { tiles_kd<DP, NTK, LSK, decltype(n)::value>(acc, Ks, qreg, c16, g); });

    // scale / mask / softmax, all in registers (the mask chunks were requested before the S MFMAs)
    float m = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < NTK; ++t) {
        const int key0 = t * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = acc[t][r];
            if (!q_first) s = s / a.scale;
            s += mk[t][r];
            if (key0 + r >= a.Nk) s = -__builtin_inff();
            acc[t][r] = s;
            m = fmaxf(m, s);
        }
    }
    m = rows4_allreduce<true>(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = exp_fast(acc[t][r] - m);
            acc[t][r] = e;
            sum += e;
        }
    sum = rows4_allreduce<false>(sum);
    const float inv_sum = 1.0f / sum;        // one division per row; e * (1 / sum) is within 1 ulp of e / sum
    float* prow = a.probs + ((static_cast<int64_t>(b) * a.H + h) * a.Nq + (qv ? q : 0)) * a.Nk;
#pragma unroll
    for (int t = 0; t < NTK; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = acc[t][r] * inv_sum;
        store_chunk(prow, t * 16 + 4 * g, a.Nk, qv && !(skip & 4), acc[t]);
    }

    // O = P.V with P straight from the accumulators: k-step (t, r) pairs P[q][16t + 4g + r] with V[16t + 4g + r][d]
    f32x4 oacc[KK];
#pragma unroll
    for (int td = 0; td < KK; ++td) oacc[td] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(skip & 8))
        with_tile_count<NTK>(na, [&](auto n) { tiles_from_regs<DP, NTK, LSV, decltype(n)::value>(oacc, acc, Vs, c16, g); });
    store_rows16<DP>(a.o + b * a.os.sb + h * a.os.sh, a.os.sn, q, qv && (!(skip & 16) || oacc[0][0] == 12345.f), a.D, g, oacc, 1.f);
}

// ------------------------------------------------------------------------------------------------------- backward
// (NTK >= 7 keeps 100+ live registers per lane: those instantiations are capped at 8 waves so that they get 256 VGPRs)
template <int DP, int NTK, bool IOH = false>
__global__ __launch_bounds__(NTK >= 7 ? 512 : 1024) void attn_bwd_head_kernel(const AttnBwdArgs a) {
    constexpr int LSA = DP + 8, LSB = DP + 4, KK = DP / 16, NPk = NTK * 16, SS = NPk + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int NTQ = nthreads >> 6, NPq = NTQ * 16;
    float* Vs = smem;                 // [NPk][LSA]   phase A: A operand of dP^T = V.dO^T
    float* Ks = Vs + NPk * LSA;       // [NPk][LSB]   phase B: B operand of dQ = dS.K
    float* Ts = smem;                 // [NPq][SS]    phase C: dS (pass 1) / P (pass 2), row-major [q][key]
    float* Bs = Ts + NPq * SS;        // [NPq][LSB]   phase C: Q' (pass 1) / dO (pass 2)
    // [NTQ] bit t set: P tile (this strip, key tile t) is not all zero -- behind whichever of the two layouts above is the longer one
    constexpr int kAB = NPk * (LSA + LSB);
    const int kC = NPq * (SS + LSB);
    unsigned* live_tab = reinterpret_cast<unsigned*>(smem + (kAB > kC ? kAB : kC));

    const int h = blockIdx.x, b = blockIdx.y;
    const int c16 = lane & 15, g = lane >> 4;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int q = wave * 16 + c16;
    const bool qv = q < a.Nq;
    const int64_t head = static_cast<int64_t>(b) * a.H + h;

    stagger(a.debug >> 8);
    MMX_TL_DECL;
    MMX_TL_MARK();                                                   // 0: start
    // this wave's rows of dO (and Q') and its chunks of P: global -> registers, issued before the LDS staging
    f32x4 doreg[KK], qreg[KK], preg[NTK];
    if constexpr (IOH)
        load_rows16_bf16<DP>(doreg, reinterpret_cast<const unsigned short*>(a.dout) + b * a.os.sb + h * a.os.sh, a.os.sn, q, a.Nq,
                             a.D, g);
    else
        load_rows16<DP>(doreg, a.dout + b * a.os.sb + h * a.os.sh, a.os.sn, q, a.Nq, a.D, g, 1.f);
    const float* prow = a.probs + b * a.probs_sb + (static_cast<int64_t>(h) * a.Nq + (qv ? q : 0)) * a.Nk;
#pragma unroll
    for (int t = 0; t < NTK; ++t) preg[t] = load_chunk(prow, t * 16 + 4 * g, a.Nk, qv);
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    if (a.need_dqkv) {
        load_rows16<DP>(qreg, a.q + b * a.qs.sb + h * a.qs.sh, a.qs.sn, q, a.Nq, a.D, g, q_first ? a.scale : 1.f);
        stage_pair<DP, 4>(Vs, LSA, vb, a.vs.sn, Ks, LSB, a.k + b * a.ks.sb + h * a.ks.sh, a.ks.sn, a.Nk, NPk, a.D, tid,
                          nthreads);
    } else {
        stage_one<DP, 4>(Vs, LSA, vb, a.vs.sn, a.Nk, NPk, a.D, tid, nthreads);
    }
    MMX_TL_MARK();                                                   // 1: V / K staged (this wave's part)
    lds_barrier();
    MMX_TL_MARK();                                                   // 2: after the first barrier

    // ---- phase A: dP^T tiles, dP -> slab, delta, dS (all in registers: acc[t][r] <-> [q][key = 16t + 4g + r])
    f32x4 acc[NTK];
#pragma unroll
    for (int t = 0; t < NTK; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    tiles_kd<DP, NTK, LSA>(acc, Vs, doreg, c16, g);
    MMX_TL_MARK();                                                   // 3: dP tiles done (waits for dO rows)
    float* dprow = a.dprobs + (head * a.Nq + (qv ? q : 0)) * a.Nk;
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < NTK; ++t) {
        store_chunk(dprow, t * 16 + 4 * g, a.Nk, qv, acc[t]);      // the captured attention gradient
#pragma unroll
        for (int r = 0; r < 4; ++r) dot += acc[t][r] * preg[t][r];
    }
    if (!a.need_dqkv) return;
    // key tiles whose probabilities are exact zeros for every row of this strip (with_tile_count): bit t of `live`, wave-uniform; published
    // for phase C (the barriers in front of its first read order the store)
    unsigned live = (1u << NTK) - 1u;
    if (a.tile_skip) {
        live = 0u;
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            const bool zero = preg[t][0] == 0.f && preg[t][1] == 0.f && preg[t][2] == 0.f && preg[t][3] == 0.f;
            if (!__all(zero)) live |= 1u << t;
        }
    }
    if (lane == 0) live_tab[wave] = live;
    const int na = live ? 32 - __builtin_clz(live) : 1;
    dot = rows4_allreduce<false>(dot);
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float ds = preg[t][r] * (acc[t][r] - dot);
            if (!q_first) ds = ds / a.scale;
            acc[t][r] = ds;
        }

    MMX_TL_MARK();                                                   // 4: dP stored, delta reduced, dS (waits for P chunks)
    // ---- phase B: dQ = dS.K, dS straight from the accumulators
    {
        f32x4 dq[KK];
#pragma unroll
        for (int td = 0; td < KK; ++td) dq[td] = f32x4{0.f, 0.f, 0.f, 0.f};
        with_tile_count<NTK>(na, [&](auto n) { tiles_from_regs<DP, NTK, LSB, decltype(n)::value>(dq, acc, Ks, c16, g); });
        if constexpr (IOH)
            store_rows16_bf16<DP>(reinterpret_cast<unsigned short*>(a.dq) + b * a.dqs.sb + h * a.dqs.sh, a.dqs.sn, q, qv, a.D, g, dq,
                                  q_first ? a.scale : 1.f);
        else
            store_rows16<DP>(a.dq + b * a.dqs.sb + h * a.dqs.sh, a.dqs.sn, q, qv, a.D, g, dq, q_first ? a.scale : 1.f);
    }

    MMX_TL_MARK();                                                   // 5: dQ done and stored
    // ---- phase C: dK = dS^T.Q' (pass 0), dV = P^T.dO (pass 1); contraction over q = across waves, through LDS
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        lds_barrier();                                           // previous readers of this LDS region are done
        MMX_TL_MARK();                                               // 6 / 8: barrier passed
#pragma unroll
        for (int t = 0; t < NTK; ++t)
            *reinterpret_cast<f32x4*>(Ts + q * SS + t * 16 + 4 * g) = pass == 0 ? acc[t] : preg[t];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            *reinterpret_cast<f32x4*>(Bs + q * LSB + kk * 16 + 4 * g) = pass == 0 ? qreg[kk] : doreg[kk];
        lds_barrier();
        float* outb = pass == 0 ? a.dk + b * a.dks.sb + h * a.dks.sh : a.dv + b * a.dvs.sb + h * a.dvs.sh;
        unsigned short* outh = pass == 0 ? reinterpret_cast<unsigned short*>(a.dk) + b * a.dks.sb + h * a.dks.sh
                                         : reinterpret_cast<unsigned short*>(a.dv) + b * a.dvs.sb + h * a.dvs.sh;
        const int64_t osn = pass == 0 ? a.dks.sn : a.dvs.sn;
        for (int kt = wave; kt < NTK; kt += NTQ) {
            // out^T tile (keys 16kt .. + 15 as the MFMA's N index): k-step s = (tq, r) pairs row 16tq + 4g + r of Ts (column
            // 16kt + c16) with the same row of Bs; the operands of step s + 1 are fetched before the MFMAs of step s
            f32x4 o[KK];
#pragma unroll
            for (int td = 0; td < KK; ++td) o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* t0 = Ts + 4 * g * SS + kt * 16 + c16;
            const float* b0 = Bs + 4 * g * LSB + c16;
            // query strips whose P tile at this key tile is all zero contribute exact zeros (dS = P (dP - delta) is zero with it): run from
            // the first live strip to the last one (a causal tower: strips kt ... NTQ - 1)
            int tq0 = NTQ, tq1 = 0;
            for (int tq = 0; tq < NTQ; ++tq)
                if ((live_tab[tq] >> kt) & 1u) { tq0 = min(tq0, tq); tq1 = tq + 1; }
            tq0 = __builtin_amdgcn_readfirstlane(min(tq0, tq1));
            tq1 = __builtin_amdgcn_readfirstlane(tq1);
            const int s0 = tq0 * 4, steps = tq1 * 4;
            const int row0 = tq0 * 16;
            float a_cur = t0[row0 * SS], a_nxt, b_cur[KK], b_nxt[KK];
#pragma unroll
            for (int td = 0; td < KK; ++td) b_cur[td] = b0[row0 * LSB + td * 16];
#pragma unroll 4
            for (int s = s0; s < steps; ++s) {
                const int sn = s + 1 < steps ? s + 1 : s;
                const int rown = (sn >> 2) * 16 + (sn & 3);
                a_nxt = t0[rown * SS];
#pragma unroll
                for (int td = 0; td < KK; ++td) b_nxt[td] = b0[rown * LSB + td * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int td = 0; td < KK; ++td) o[td] = mfma16x16x4(b_cur[td], a_cur, o[td]);
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_nxt;
#pragma unroll
                for (int td = 0; td < KK; ++td) b_cur[td] = b_nxt[td];
            }
            const int key = kt * 16 + c16;
            if constexpr (IOH) store_rows16_bf16<DP>(outh, osn, key, key < a.Nk, a.D, g, o, 1.f);
            else store_rows16<DP>(outb, osn, key, key < a.Nk, a.D, g, o, 1.f);
        }
        MMX_TL_MARK();                                               // 7 / 9: pass done
    }
#ifdef MMX_HEAD_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const long long tend = __builtin_readcyclecounter();
        __syncthreads();
        if (lane == 0 && wave < 5) {
            float* out = a.dv + b * a.dvs.sb + h * a.dvs.sh + wave * 12;
            for (int i = 1; i < 10; ++i) out[i - 1] = static_cast<float>(tl_[i] - tl_[0]);
            out[9] = static_cast<float>(tend - tl_[0]);
            out[10] = static_cast<float>(tl_[0] & 0xffffff);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------- host side
static bool aligned16(const void* p, int64_t s0, int64_t s1, int64_t s2) {
    return ((reinterpret_cast<uintptr_t>(p) | static_cast<uintptr_t>(s0 * 4) | static_cast<uintptr_t>(s1 * 4) |
             static_cast<uintptr_t>(s2 * 4)) & 15u) == 0;
}

static int g_attn_head = 1;   // 0: skip these kernels (tests / A-B profiling run the older paths)
static int g_attn_head_tile_skip = 1;   // option "attn_head_tile_skip": masked-tile skip (with_tile_count); 0 for A / B runs
void attn_head_tile_skip(int on) { g_attn_head_tile_skip = on ? 1 : 0; }
static int g_attn_head_stagger = 0;
void attn_head_enable(int on) { g_attn_head = on & 1; g_attn_head_stagger = on >> 8; }   // bits 8..15 fwd phase skips, 16.. stagger

static size_t fwd_head_lds(int DP, int NTK) { return sizeof(float) * NTK * 16 * (2 * DP + 12); }

static size_t bwd_head_lds(int DP, int NTK, int NTQ) {
    const size_t ab = static_cast<size_t>(NTK) * 16 * (2 * DP + 12);
    const size_t c = static_cast<size_t>(NTQ) * 16 * (NTK * 16 + 4 + DP + 4);
    return sizeof(float) * ((ab > c ? ab : c) + static_cast<size_t>(NTQ));      // + the strips' live-tile words
}

template <typename K, typename A>
static int launch_head(K kern, const A& args, int threads, size_t lds, hipStream_t s, const char* name) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    kern<<<dim3(args.H, args.B), threads, lds, s>>>(args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MMX_OK;
}

#define MMX_HEAD_CASE(KERN, DPV, N)                                                                            \
    case N:                                                                                                    \
        return launch_head(KERN<DPV, N>, a, threads, lds, s, #KERN)

template <int DP>
static int fwd_head_dispatch(const AttnFwdArgs& a, int NTK, int threads, size_t lds, hipStream_t s) {
    switch (NTK) {
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 1);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 2);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 3);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 4);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 5);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 6);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 7);
        MMX_HEAD_CASE(attn_fwd_head_kernel, DP, 8);
    }
    return MMX_ENOTSUP;
}

#define MMX_HEAD_CASE_IO(DPV, N, IO)                                                                          \
    case N:                                                                                                    \
        return launch_head(attn_bwd_head_kernel<DPV, N, IO>, a, threads, lds, s, "attn_bwd_head_kernel")

template <int DP, bool IOH>
static int bwd_head_dispatch(const AttnBwdArgs& a, int NTK, int threads, size_t lds, hipStream_t s) {
    switch (NTK) {
        MMX_HEAD_CASE_IO(DP, 1, IOH);
        MMX_HEAD_CASE_IO(DP, 2, IOH);
        MMX_HEAD_CASE_IO(DP, 3, IOH);
        MMX_HEAD_CASE_IO(DP, 4, IOH);
        MMX_HEAD_CASE_IO(DP, 5, IOH);
        MMX_HEAD_CASE_IO(DP, 6, IOH);
        MMX_HEAD_CASE_IO(DP, 7, IOH);
        MMX_HEAD_CASE_IO(DP, 8, IOH);
    }
    return MMX_ENOTSUP;
}

// returns 1 if the kernel was launched (rc in *rc_out), 0 if the shape is not eligible
int attn_fwd_head_try(const AttnFwdArgs& a_in, hipStream_t s, int* rc_out) {
    AttnFwdArgs a = a_in;
    a.debug = g_attn_head_stagger;
    a.tile_skip = g_attn_head_tile_skip;
    if (!g_attn_head || a.slab_dt != MMX_F32 || a.D % 4 || a.D > 64 || a.Nk > 128 || a.Nq > 256 || a.Nq < 1 || a.Nk < 1)
        return 0;
    if (!aligned16(a.q, a.qs.sb, a.qs.sh, a.qs.sn) || !aligned16(a.k, a.ks.sb, a.ks.sh, a.ks.sn) ||
        !aligned16(a.v, a.vs.sb, a.vs.sh, a.vs.sn))
        return 0;
    const int DP = a.D <= 32 ? 32 : 64, NTK = (a.Nk + 15) / 16, threads = 64 * ((a.Nq + 15) / 16);
    const size_t lds = fwd_head_lds(DP, NTK);
    *rc_out = DP == 32 ? fwd_head_dispatch<32>(a, NTK, threads, lds, s) : fwd_head_dispatch<64>(a, NTK, threads, lds, s);
    return 1;
}

int attn_bwd_head_try(const AttnBwdArgs& a_in, hipStream_t s, int* rc_out) {
    AttnBwdArgs a = a_in;
    a.debug = g_attn_head_stagger;
    a.tile_skip = g_attn_head_tile_skip;
    if (!g_attn_head || a.slab_dt != MMX_F32 || a.D % 4 || a.D > 64 || a.Nk > 128 || a.Nq > 256 || a.Nq < 1 || a.Nk < 1)
        return 0;
    // (bf16 gradient I/O: 4-element chunks are 8 bytes -- aligned16 on half the byte strides is the 8-byte test)
    const auto io_ok = [&](const void* p, const Strides& st) {
        return a.io_bf16 ? ((reinterpret_cast<uintptr_t>(p) | static_cast<uintptr_t>(st.sb * 2) | static_cast<uintptr_t>(st.sh * 2) |
                             static_cast<uintptr_t>(st.sn * 2)) & 7u) == 0
                         : aligned16(p, st.sb, st.sh, st.sn);
    };
    if (!aligned16(a.v, a.vs.sb, a.vs.sh, a.vs.sn) || !io_ok(a.dout, a.os)) return 0;
    if (a.need_dqkv && (!aligned16(a.q, a.qs.sb, a.qs.sh, a.qs.sn) || !aligned16(a.k, a.ks.sb, a.ks.sh, a.ks.sn)))
        return 0;
    if (a.io_bf16 && a.need_dqkv && (!io_ok(a.dq, a.dqs) || !io_ok(a.dk, a.dks) || !io_ok(a.dv, a.dvs))) return 0;
    const int DP = a.D <= 32 ? 32 : 64, NTK = (a.Nk + 15) / 16, NTQ = (a.Nq + 15) / 16, threads = 64 * NTQ;
    const size_t lds = bwd_head_lds(DP, NTK, NTQ);
    if (lds > 160 * 1024 || (NTK >= 7 && NTQ > 8)) return 0;
    if (a.io_bf16)
        *rc_out = DP == 32 ? bwd_head_dispatch<32, true>(a, NTK, threads, lds, s) : bwd_head_dispatch<64, true>(a, NTK, threads, lds, s);
    else
        *rc_out = DP == 32 ? bwd_head_dispatch<32, false>(a, NTK, threads, lds, s) : bwd_head_dispatch<64, false>(a, NTK, threads, lds, s);
    return 1;
}

}  // namespace mmx
