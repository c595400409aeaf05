// Second generation of the bf16-matrix-core attention BACKWARD for long sequences (MMX_ATTN_MMA_BF16; BASELINE config 5:
// ViT-L/14@336, 577 tokens x 64).  attention_stream.hip's bf16 mode kept the fp32 LDS tiles of the exact-fp32 kernels and
// rounded every operand in registers; PMC showed those kernels latency-bound at 2 waves / SIMD (33-54 % of the wave cycles
// parked between the two barriers of a tile, profiles/r02_cfg5_probe.txt).  Here
//   * the streamed tiles live in LDS as bf16 -- half the ds traffic, an MFMA operand is ONE ds_read_b128 (or two b64),
//     no cvt in the inner loop;
//   * operands that are contracted over the tile's ROW index (K in dQ = dS.K; dO, Q in dV = P^T.dO, dK = dS^T.Q) are staged
//     TRANSPOSED ([d][row]): a thread fetches a 4 x 4 block (4 consecutive rows x 4 consecutive d), so both the row-major and
//     the transposed copy are written with ds_write_b64 -- no 2-byte scatter;
//   * the score-shaped tile never goes through LDS: the products are oriented so that the MFMA's C layout IS the next MFMA's
//     operand layout (the k-order of a contraction is free: slot (g, j) <-> index 16 (j >> 2) + 4 g + (j & 3) of a tile pair);
//       query side:  dP^T[key][q] = V . dO^T   (A = V tile rows, B = the wave's dO rows in registers)
//                    dQ^T[d][q]  += K^T . dS^T (A = K^T tile, B = dS^T straight from the accumulators)
//       key side:    dP[q][key]   = dO . V^T   (A = dO tile rows, B = the wave's V rows in registers)
//                    dV^T, dK^T: dV[key][d] += P^T . dO, dK[key][d] += dS^T . Q  (A = P^T / dS^T from the accumulators, B = dO^T / Q^T tiles)
//     on the query side a lane then owns 4 CONSECUTIVE keys of one query row: P is one 8-byte load per 16 keys instead of four
//     2-byte ones, and the row-relevancy reduction over the 16 queries of a wave is a DPP row reduction;
//   * tiles are double-buffered: ONE barrier per tile, the next tile's global loads are in flight (raw registers) during the
//     current tile's MFMAs and are written to the other buffer at the top of the next iteration.
// Softmax-side arithmetic (delta, dS), accumulation and the relevancy reduction stay fp32, as in the first generation.
#include "mmx_common.h"
#include "attention_args.h"

#include <type_traits>

namespace mmx {
namespace {

constexpr int kR = 64;        // query rows (query side) / keys (key side) per workgroup = 4 waves x 16
constexpr int kT = 64;        // rows of the streamed operand per step
constexpr int kTh = 256;
constexpr int kDP = 64;       // padded head dim (D <= 64, D % 8 == 0)
// LDS layouts, bank-conflict free for every access below under the lane-group rules of MI355X_MICROARCH.md (searched by
// brute force over row strides / swizzles; the first version -- 72-element rows, no swizzle -- spent 22 % of its wave cycles
// in SQ_LDS_BANK_CONFLICT: the transposed ds_write_b64 of 16 lanes hit 2 bank positions):
//   row-major tile  [row][d]:  80-element (160 B) rows, no swizzle   (ds_write_b64 by (row, 4 d), ds_read_b128 by (row i, 8 g))
//   transposed tile [d][row]:  80-element rows; the 4-row group index of a row is XOR-ed with (d / 4) & 15
//                              (ds_write_b64 of 4 consecutive rows by 16 lanes of different d; ds_read_b64 by (d = i, group g))
// (Unpadded 64-element rows with XOR swizzles are conflict-free too and would let THREE workgroups fit a CU, but the
//  kernels need 162-184 VGPRs: under __launch_bounds__(256, 3) they spill 60-128 registers and run 4 x slower.)
constexpr int kLR = kDP + 16;
constexpr int kLT = kT + 16;

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

int g_attn_bf16_v2 = 1;

__device__ __forceinline__ unsigned pk2(float lo, float hi) {      // two fp32 -> packed bf16 pair, round to nearest even
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r;
    r[0] = static_cast<__bf16>(lo);
    r[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4v v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
    return as_bf16x8(u32x4v{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])});
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- a [64 rows x D] operand block in flight: thread f owns rows 4 (f / 16) .. + 3, columns 4 (f % 16) .. + 3 (a 4 x 4 block).
// HALF: the operand is bf16 in global memory (8-byte loads), else fp32 (16-byte loads).  Raw words only: no ALU at fetch time.
template <bool HALF>
struct Block4x4 {
    typename std::conditional<HALF, u32x2v, f32x4>::type raw[4];
    int row0;
};
template <bool HALF, typename T>
__device__ __forceinline__ void block_fetch(Block4x4<HALF>& blk, const T* base, int64_t sn, int row0, int rows_total, int D,
                                            int tid) {
    const int r4 = 4 * (tid >> 4), c = 4 * (tid & 15);
    blk.row0 = row0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = row0 + r4 + e;
        const bool ok = row < rows_total && c < D;
        const T* src = base + (ok ? static_cast<int64_t>(row) * sn + c : 0);     // clamped, unconditional
        if constexpr (HALF) blk.raw[e] = *reinterpret_cast<const u32x2v*>(src);
        else blk.raw[e] = *reinterpret_cast<const f32x4*>(src);
    }
}
// the block as 4 rows of 2 packed-bf16 dwords (columns c, c+1 | c+2, c+3), masked and scaled
template <bool HALF>
__device__ __forceinline__ void block_rows(u32x2v (&rows)[4], const Block4x4<HALF>& blk, float mul, int rows_total, int D,
                                           int tid) {
    const int r4 = 4 * (tid >> 4), c = 4 * (tid & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = blk.row0 + r4 + e < rows_total && c < D;
        u32x2v v;
        if constexpr (HALF) v = blk.raw[e];                          // (mul == 1 for the bf16 operand: dO)
        else v = u32x2v{pk2(blk.raw[e][0] * mul, blk.raw[e][1] * mul), pk2(blk.raw[e][2] * mul, blk.raw[e][3] * mul)};
        rows[e] = ok ? v : u32x2v{0u, 0u};
    }
}
__device__ __forceinline__ void store_row_major(bf16_t* tile, const u32x2v (&rows)[4], int tid) {
    const int r4 = 4 * (tid >> 4), c = 4 * (tid & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x2v*>(tile + (r4 + e) * kLR + c) = rows[e];
}
// transposed copy: element (row r4 + e, column c + dd) -> tile[c + dd][r4 + e]; 4 consecutive rows = one 8-byte store
__device__ __forceinline__ void store_transposed(bf16_t* tile, const u32x2v (&rows)[4], int tid) {
    const int c = 4 * (tid & 15);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        const int w = dd >> 1;
        const unsigned sel = (dd & 1) ? 0x07060302u : 0x05040100u;     // high / low halves of (row e+1, row e)
        const unsigned lo = __builtin_amdgcn_perm(rows[1][w], rows[0][w], sel);
        const unsigned hi = __builtin_amdgcn_perm(rows[3][w], rows[2][w], sel);
        *reinterpret_cast<u32x2v*>(tile + (c + dd) * kLT + 4 * ((tid >> 4) ^ (tid & 15))) = u32x2v{lo, hi};   // (c + dd) / 4 == tid & 15
    }
}

// the wave's 16 rows of an operand as MFMA operand registers: lane (i, g) holds row `row`, d = 32 pr + 8 g .. + 7
template <bool HALF, typename T>
__device__ __forceinline__ void load_rows8(bf16x8 (&op)[kDP / 32], const T* base, int64_t sn, int row, bool row_ok, int D,
                                           int g, float mul) {
#pragma unroll
    for (int pr = 0; pr < kDP / 32; ++pr) {
        const int d0 = 32 * pr + 8 * g;
        const bool ok = row_ok && d0 < D;
        const T* src = base + static_cast<int64_t>(row) * sn + (d0 < D ? d0 : 0);
        u32x4v v;
        if constexpr (HALF) {
            v = *reinterpret_cast<const u32x4v*>(src);
        } else {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            v = u32x4v{pk2(lo[0] * mul, lo[1] * mul), pk2(lo[2] * mul, lo[3] * mul), pk2(hi[0] * mul, hi[1] * mul),
                       pk2(hi[2] * mul, hi[3] * mul)};
        }
        op[pr] = as_bf16x8(ok ? v : u32x4v{0u, 0u, 0u, 0u});
    }
}

// acc[t] (t = the four 16-row sub-tiles of a row-major LDS tile) = (tile rows 16 t + i) x (register rows), contraction over d.
// TILE_IS_A: C[m = tile row][n = register row] (else swapped).
template <bool TILE_IS_A>
__device__ __forceinline__ void tile_x_regs(f32x4 (&acc)[4], const bf16_t* tile, const bf16x8 (&reg)[kDP / 32], int i, int g) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < kDP / 32; ++pr) {
        bf16x8 op[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) op[t] = *reinterpret_cast<const bf16x8*>(tile + (16 * t + i) * kLR + 32 * pr + 8 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = TILE_IS_A ? mfma16x16x32_bf16(op[t], reg[pr], acc[t]) : mfma16x16x32_bf16(reg[pr], op[t], acc[t]);
    }
}

// one operand of the "contract over the tile's row index" products, from a TRANSPOSED tile [d][row]: row-slots (g, j) of
// tile pair p (rows 32 p + 16 (j >> 2) + 4 g + (j & 3)) for d = 16 dt + i
__device__ __forceinline__ bf16x8 transposed_operand(const bf16_t* tile, int dt, int p, int i, int g) {
    const int sw = (4 * dt + (i >> 2)) & 15;                          // (d / 4) & 15 of d = 16 dt + i
    const bf16_t* row = tile + (16 * dt + i) * kLT;
    const u32x2v lo = *reinterpret_cast<const u32x2v*>(row + 4 * ((8 * p + g) ^ sw));
    const u32x2v hi = *reinterpret_cast<const u32x2v*>(row + 4 * ((8 * p + 4 + g) ^ sw));
    return as_bf16x8(u32x4v{lo[0], lo[1], hi[0], hi[1]});
}

// ===================================================================================================== query side
// dP (-> capture slab), delta = rowsum(dO * O), dS, dQ = dS.K.  (The row-relevancy reduction lives on the KEY side: there
// a lane owns a key and 16 query rows, so the sum over queries is an in-lane sum -- here it was 16 DPP row reductions per
// tile, a third of the kernel's VALU work.)
template <int DT, bool IOH>
__global__ __launch_bounds__(kTh, 2) void attn_bwd_q_bf16_kernel(const AttnBwdArgs a) {
    typedef typename slab_elem<DT>::type slab_t;
    constexpr int NB = kDP / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* Vt = reinterpret_cast<bf16_t*>(smem_raw);                 // [2][kT][kLR]   V rows (row-major)
    bf16_t* Kt = Vt + 2 * kT * kLR;                                   // [2][kDP][kLT]  K transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int nrt = (a.Nq + kR - 1) / kR;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int rt = wg % nrt, h = (wg / nrt) % a.H, b = wg / (nrt * a.H);
    const int q = rt * kR + wave * 16 + i;                            // this lane's query row (columns of the C tiles)
    const bool q_ok = q < a.Nq;
    const int qc = min(q, a.Nq - 1);
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    const bool need_ds = a.need_dqkv != 0;

    // the wave's dO rows as the B operand of dP^T, and delta = rowsum(dO * O) of this lane's row
    bf16x8 dob[kDP / 32];
    float delta = 0.f;
    {
        const float* ob = a.o + b * a.oos.sb + h * a.oos.sh + static_cast<int64_t>(qc) * a.oos.sn;
        float part = 0.f;
        if constexpr (IOH) {
            const bf16_t* src = reinterpret_cast<const bf16_t*>(a.dout) + b * a.os.sb + h * a.os.sh;
            load_rows8<true>(dob, src, a.os.sn, qc, q_ok, a.D, g, 1.f);
#pragma unroll
            for (int pr = 0; pr < kDP / 32; ++pr) {
                const int d0 = 32 * pr + 8 * g;
                if (d0 < a.D) {
                    const u32x4v w = __builtin_bit_cast(u32x4v, dob[pr]);
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(ob + d0), o1 = *reinterpret_cast<const f32x4*>(ob + d0 + 4);
                    part += bflo(w[0]) * o0[0] + bfhi(w[0]) * o0[1] + bflo(w[1]) * o0[2] + bfhi(w[1]) * o0[3] +
                            bflo(w[2]) * o1[0] + bfhi(w[2]) * o1[1] + bflo(w[3]) * o1[2] + bfhi(w[3]) * o1[3];
                }
            }
        } else {
            const float* src = a.dout + b * a.os.sb + h * a.os.sh;
            load_rows8<false>(dob, src, a.os.sn, qc, q_ok, a.D, g, 1.f);
#pragma unroll
            for (int pr = 0; pr < kDP / 32; ++pr) {
                const int d0 = 32 * pr + 8 * g;
                if (d0 < a.D) {
                    const float* dsrc = src + static_cast<int64_t>(qc) * a.os.sn + d0;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(dsrc), x1 = *reinterpret_cast<const f32x4*>(dsrc + 4);
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(ob + d0), o1 = *reinterpret_cast<const f32x4*>(ob + d0 + 4);
                    part += x0[0] * o0[0] + x0[1] * o0[1] + x0[2] * o0[2] + x0[3] * o0[3] + x1[0] * o1[0] + x1[1] * o1[1] +
                            x1[2] * o1[2] + x1[3] * o1[3];
                }
            }
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        delta = q_ok ? part : 0.f;
        if (need_ds && g == 0 && q_ok) a.delta[head * a.Nq + q] = delta;
    }
    const slab_t* prow = reinterpret_cast<const slab_t*>(a.probs) + b * a.probs_sb + (static_cast<int64_t>(h) * a.Nq + qc) * a.Nk;
    const int64_t prow_idx = b * a.probs_sb + (static_cast<int64_t>(h) * a.Nq + qc) * a.Nk;     // element index of the row start
    slab_t* dprow = (a.dprobs && q_ok) ? reinterpret_cast<slab_t*>(a.dprobs) + (head * a.Nq + q) * a.Nk : nullptr;
    const float ds_mul = q_first ? 1.f : 1.f / a.scale;
    const int ntiles = (a.Nk + kT - 1) / kT;

    f32x4 qacc[NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) qacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    Block4x4<false> vreg, kreg;
    auto fetch = [&](int kt) {
        block_fetch<false>(vreg, vb, a.vs.sn, kt * kT, a.Nk, a.D, tid);
        if (need_ds) block_fetch<false>(kreg, kb, a.ks.sn, kt * kT, a.Nk, a.D, tid);
    };
    auto stage = [&](int buf) {
        u32x2v rows[4];
        block_rows<false>(rows, vreg, 1.f, a.Nk, a.D, tid);
        store_row_major(Vt + buf * kT * kLR, rows, tid);
        if (need_ds) {
            block_rows<false>(rows, kreg, 1.f, a.Nk, a.D, tid);
            store_transposed(Kt + buf * kDP * kLT, rows, tid);
        }
    };

    // everything that follows dP^T of a tile: p[t][r], dpT[t][r] <-> key 16 t + 4 g + r of this lane's query row
    auto tile_tail = [&](const f32x4 (&p)[4], const f32x4 (&dpT)[4], int kt, bool edge) {
        const bf16_t* Kcur = Kt + (kt & 1) * kDP * kLT;
        if (dprow) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * kT + 16 * t + 4 * g + r;
                    if (!edge || key < a.Nk) slab_store<DT>(dprow + key, dpT[t][r]);
                }
        }
        if (need_ds) {
            f32x4 ds[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ds[t] = p[t] * (dpT[t] - delta) * ds_mul;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const bf16x8 dsb = pack8(ds[2 * pp], ds[2 * pp + 1]);
#pragma unroll
                for (int dt = 0; dt < NB; ++dt)
                    qacc[dt] = mfma16x16x32_bf16(transposed_operand(Kcur, dt, pp, i, g), dsb, qacc[dt]);
            }
        }
    };
    // Interior tiles: the lane's 4 consecutive probabilities per 16 keys are ONE (possibly 2-byte aligned) load.  They are
    // fetched ONE TILE AHEAD as raw words (p_cur was issued during the previous iteration, before that iteration's K / V
    // prefetch; converted here): a load consumed inside the iteration that issues it waits behind the prefetch it was queued
    // with -- vmcnt retires in order -- and that exposed a full memory round trip per tile (3 us of a 3.2 us tile step).
    // The last tile may run past Nk: clamped element loads, issued and consumed in place (one tile of ~10).
    // 16-bit slabs: load4_stream_raw fetches the three ALIGNED dwords from element (idx & ~1), i.e. up to 2 elements past
    // the 4 it needs.  For the last row of the slab and Nk % 64 == 1 (577) that window would end past the tensor: a chunk
    // whose window does not fit (idx + 6 > slab elements; a handful of lanes of ONE workgroup) is assembled from clamped
    // element loads instead, into the same raw layout -- nothing is ever read outside the caller's slab.
    const bool want_p = need_ds;
    stream_raw<DT> p_cur[4], p_nxt[4];
    const int64_t slab_elems = (a.probs_sb == 0 ? 1 : static_cast<int64_t>(a.B)) * a.H * a.Nq * a.Nk;
    auto p_issue = [&](stream_raw<DT> (&raw)[4], int kt) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t idx = prow_idx + kt * kT + 16 * t + 4 * g;
            if constexpr (DT == MMX_F32) {
                raw[t] = load4_stream_raw<DT>(a.probs, idx);
            } else {
                if (idx + 6 <= slab_elems) {
                    raw[t] = load4_stream_raw<DT>(a.probs, idx);
                } else {                                                   // the slab's last few elements: element loads
                    const unsigned short* pe = reinterpret_cast<const unsigned short*>(a.probs);
                    const int64_t e0 = idx & ~static_cast<int64_t>(1);
                    unsigned w[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int64_t lo = e0 + 2 * j, hi = lo + 1;
                        const unsigned a0 = lo < slab_elems ? pe[lo] : 0u, a1 = hi < slab_elems ? pe[hi] : 0u;
                        w[j] = a0 | (a1 << 16);
                    }
                    raw[t].v = u32x3{w[0], w[1], w[2]};
                }
            }
        }
    };
    auto tile_compute = [&](int kt, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        const bf16_t* Vcur = Vt + (kt & 1) * kT * kLR;
        f32x4 p[4], dpT[4];
        if (!want_p) {
#pragma unroll
            for (int t = 0; t < 4; ++t) p[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            tile_x_regs<true>(dpT, Vcur, dob, i, g);
        } else if constexpr (!EDGE) {
            tile_x_regs<true>(dpT, Vcur, dob, i, g);
#pragma unroll
            for (int t = 0; t < 4; ++t) p[t] = stream_cvt<DT>(p_cur[t], prow_idx + kt * kT + 16 * t + 4 * g);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * kT + 16 * t + 4 * g + r;
                    const float v = slab_load<DT>(prow + min(key, a.Nk - 1));
                    p[t][r] = key < a.Nk ? v : 0.f;
                }
            tile_x_regs<true>(dpT, Vcur, dob, i, g);
        }
        tile_tail(p, dpT, kt, EDGE);
    };

    fetch(0);
    stage(0);
    if (ntiles > 1) {
        if (want_p) p_issue(p_cur, 0);
        fetch(1);
    }
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) {
            stage((kt + 1) & 1);                                   // the tile fetched during the previous iteration
            if (want_p && kt + 2 < ntiles) p_issue(p_nxt, kt + 1);  // (tile kt + 1 is interior)
            if (kt + 2 < ntiles) fetch(kt + 2);
        }
        if (kt + 1 < ntiles) tile_compute(kt, std::false_type{}); else tile_compute(kt, std::true_type{});
        if (want_p && kt + 2 < ntiles) {
#pragma unroll
            for (int t = 0; t < 4; ++t) p_cur[t] = p_nxt[t];
        }
        lds_barrier();
    }
    if (!need_ds || !q_ok) return;
    // dQ^T accumulators: lane (q = column i), rows d = 16 dt + 4 g + r: 4 consecutive d of one query row
    const float mul = q_first ? a.scale : 1.f;
    const int64_t off = b * a.dqs.sb + h * a.dqs.sh + static_cast<int64_t>(q) * a.dqs.sn;
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        if (d0 < a.D) {
            const f32x4 v = qacc[dt] * mul;
            if constexpr (IOH)
                *reinterpret_cast<u32x2v*>(reinterpret_cast<bf16_t*>(a.dq) + off + d0) = u32x2v{pk2(v[0], v[1]), pk2(v[2], v[3])};
            else
                *reinterpret_cast<f32x4*>(a.dq + off + d0) = v;
        }
    }
}

// ===================================================================================================== key side
// per 64 keys (16 per wave): dP recomputed, dV = P^T.dO, dK = dS^T.Q (DKV), and / or the row-relevancy product of this head
// (REL):  rel_part[b][h][key] = sum_q rel_v[b][q] * clamp(dP * P, 0)[q][key] -- the lane owns the key, its 16 query rows of a
// tile are registers, so the sum over ALL queries is one running register, two cross-row shuffles at the very end and ONE
// partial row per head (fixed order: deterministic).  DKV = false (the lowest explained layer) runs only that.
template <int DT, bool IOH, bool REL, bool DKV>
__global__ __launch_bounds__(kTh, 2) void attn_bwd_kv_bf16_kernel(const AttnBwdArgs a) {
    typedef typename slab_elem<DT>::type slab_t;
    constexpr int NB = kDP / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* dOr = reinterpret_cast<bf16_t*>(smem_raw);                // [2][kT][kLR]   dO rows (row-major)
    bf16_t* dOt = dOr + 2 * kT * kLR;                                 // [2][kDP][kLT]  dO transposed
    bf16_t* Qt = dOt + 2 * kDP * kLT;                                 // [2][kDP][kLT]  Q transposed (pre-scaled)
    float* dl = reinterpret_cast<float*>(Qt + 2 * kDP * kLT);         // [2][kT]        delta of the staged query rows
    float* vl = dl + 2 * kT;                                          // [2][kT]        REL: rel_v of the staged query rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int nkt = (a.Nk + kR - 1) / kR;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int h = (wg / nkt) % a.H, b = wg / (nkt * a.H);
    const int kw = (wg % nkt) * kR + wave * 16;                       // first key of this wave
    const int key = kw + i;
    const bool key_ok = key < a.Nk;
    const int keyc = min(key, a.Nk - 1);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    const slab_t* pcol = reinterpret_cast<const slab_t*>(a.probs) + b * a.probs_sb + static_cast<int64_t>(h) * a.Nq * a.Nk + keyc;
    const float ds_mul = q_first ? 1.f : 1.f / a.scale;

    bf16x8 vop[kDP / 32];                                              // this wave's V rows: B operand of dP = dO . V^T
    load_rows8<false>(vop, a.v + b * a.vs.sb + h * a.vs.sh, a.vs.sn, keyc, key_ok, a.D, g, 1.f);

    f32x4 kacc[NB], vacc[NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) kacc[dt] = vacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (a.Nq + kT - 1) / kT;
    Block4x4<false> qreg;
    Block4x4<IOH> doreg;
    float dlreg = 0.f, vlreg = 0.f, racc = 0.f;
    auto fetch = [&](int qt) {
        if constexpr (DKV) block_fetch<false>(qreg, qb, a.qs.sn, qt * kT, a.Nq, a.D, tid);
        if constexpr (IOH)
            block_fetch<true>(doreg, reinterpret_cast<const bf16_t*>(a.dout) + b * a.os.sb + h * a.os.sh, a.os.sn, qt * kT, a.Nq,
                              a.D, tid);
        else
            block_fetch<false>(doreg, a.dout + b * a.os.sb + h * a.os.sh, a.os.sn, qt * kT, a.Nq, a.D, tid);
        if (tid < kT) {
            const int row = min(qt * kT + tid, a.Nq - 1);
            if constexpr (DKV) dlreg = a.delta[head * a.Nq + row];
            if constexpr (REL) vlreg = a.rel_v[static_cast<int64_t>(b) * a.Nq + row];
        }
    };
    auto stage = [&](int buf) {
        u32x2v rows[4];
        block_rows<IOH>(rows, doreg, 1.f, a.Nq, a.D, tid);
        store_row_major(dOr + buf * kT * kLR, rows, tid);
        if constexpr (DKV) {
            store_transposed(dOt + buf * kDP * kLT, rows, tid);
            block_rows<false>(rows, qreg, q_first ? a.scale : 1.f, a.Nq, a.D, tid);
            store_transposed(Qt + buf * kDP * kLT, rows, tid);
        }
        if (tid < kT) {
            if constexpr (DKV) dl[buf * kT + tid] = dlreg;
            if constexpr (REL) vl[buf * kT + tid] = vlreg;      // (rows past Nq: p is masked to 0 below)
        }
    };
    // this lane's probability column segments of a query tile (lane = key, 16 query rows): raw words, fetched one tile
    // ahead like the query side's (and ahead of the Q / dO prefetch in issue order)
    slab_t p_cur[4][4], p_nxt[4][4];
    auto p_issue = [&](slab_t (&raw)[4][4], int qt) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                raw[t][r] = pcol[static_cast<int64_t>(min(qt * kT + 16 * t + 4 * g + r, a.Nq - 1)) * a.Nk];
    };
    fetch(0);
    stage(0);
    p_issue(p_cur, 0);
    if (ntiles > 1) fetch(1);
    __syncthreads();
    for (int qt = 0; qt < ntiles; ++qt) {
        const int cur = qt & 1;
        if (qt + 1 < ntiles) {
            stage(cur ^ 1);
            p_issue(p_nxt, qt + 1);
            if (qt + 2 < ntiles) fetch(qt + 2);
        }
        float p[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                if constexpr (DT == MMX_F32) v = p_cur[t][r];
                else if constexpr (DT == MMX_BF16) v = bf16_bits_to_f32(p_cur[t][r]);
                else v = f16_bits_to_f32(p_cur[t][r]);
                p[t][r] = (key_ok && qt * kT + 16 * t + 4 * g + r < a.Nq) ? v : 0.f;
            }
        f32x4 dp[4];                                                   // dp[t][r] = dP[16 t + 4 g + r][key i]
        tile_x_regs<true>(dp, dOr + cur * kT * kLR, vop, i, g);
        if constexpr (REL) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(vl + cur * kT + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) racc += vv[r] * relu_nan(p[t][r] * dp[t][r]);
            }
        }
        const bf16_t* dOtc = dOt + cur * kDP * kLT;
        const bf16_t* Qtc = Qt + cur * kDP * kLT;
        if constexpr (DKV) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            f32x4 ds[2], pv[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * pp + hh;
                const f32x4 dlv = *reinterpret_cast<const f32x4*>(dl + cur * kT + 16 * t + 4 * g);
                pv[hh] = f32x4{p[t][0], p[t][1], p[t][2], p[t][3]};
                ds[hh] = pv[hh] * (dp[t] - dlv) * ds_mul;
            }
            const bf16x8 p_op = pack8(pv[0], pv[1]), ds_op = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < NB; ++dt) {
                vacc[dt] = mfma16x16x32_bf16(p_op, transposed_operand(dOtc, dt, pp, i, g), vacc[dt]);
                kacc[dt] = mfma16x16x32_bf16(ds_op, transposed_operand(Qtc, dt, pp, i, g), kacc[dt]);
            }
        }
        }
        if (qt + 1 < ntiles) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) p_cur[t][r] = p_nxt[t][r];
        }
        lds_barrier();
    }
    if constexpr (REL) {
        racc += __shfl_xor(racc, 16);
        racc += __shfl_xor(racc, 32);                              // the 4 row groups of the tile rows: all queries of key i
        if (g == 0 && key_ok) a.rel_part[head * a.Nk + key] = racc;
    }
    if constexpr (!DKV) return;
    // accumulators: lane (d = 16 dt + i), rows key = kw + 4 g + r
    const int64_t dk0 = b * a.dks.sb + h * a.dks.sh, dv0 = b * a.dvs.sb + h * a.dvs.sh;
    const bool odd = i & 1;
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) {
        const int d = 16 * dt + i;
        if constexpr (IOH) {
            // pair lanes i / i ^ 1: the even lane writes rows r = 0, 2, the odd lane rows 1, 3, as (d, d + 1) dwords
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const int r = 2 * rp + (odd ? 1 : 0);
                const int j = kw + 4 * g + r, c0 = d - (odd ? 1 : 0);
                const float km = odd ? kacc[dt][2 * rp + 1] : kacc[dt][2 * rp];
                const float vm = odd ? vacc[dt][2 * rp + 1] : vacc[dt][2 * rp];
                const float ko = __int_as_float(__builtin_amdgcn_update_dpp(
                    0, __float_as_int(odd ? kacc[dt][2 * rp] : kacc[dt][2 * rp + 1]), 0xB1, 0xF, 0xF, false));
                const float vo = __int_as_float(__builtin_amdgcn_update_dpp(
                    0, __float_as_int(odd ? vacc[dt][2 * rp] : vacc[dt][2 * rp + 1]), 0xB1, 0xF, 0xF, false));
                if (j < a.Nk && c0 < a.D) {
                    bf16_t* dk = reinterpret_cast<bf16_t*>(a.dk) + dk0 + static_cast<int64_t>(j) * a.dks.sn + c0;
                    bf16_t* dv = reinterpret_cast<bf16_t*>(a.dv) + dv0 + static_cast<int64_t>(j) * a.dvs.sn + c0;
                    *reinterpret_cast<unsigned*>(dk) = odd ? pk2(ko, km) : pk2(km, ko);
                    *reinterpret_cast<unsigned*>(dv) = odd ? pk2(vo, vm) : pk2(vm, vo);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kw + 4 * g + r;
                if (j < a.Nk && d < a.D) {
                    a.dk[dk0 + static_cast<int64_t>(j) * a.dks.sn + d] = kacc[dt][r];
                    a.dv[dv0 + static_cast<int64_t>(j) * a.dvs.sn + d] = vacc[dt][r];
                }
            }
        }
    }
}

constexpr size_t kQLds = sizeof(bf16_t) * (2 * kT * kLR + 2 * kDP * kLT);
constexpr size_t kKvLds = sizeof(bf16_t) * (2 * kT * kLR + 4 * kDP * kLT) + sizeof(float) * 4 * kT;

template <typename K>
int launch_v2(K kern, const AttnBwdArgs& a, dim3 grid, size_t lds, hipStream_t s, const char* name) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    kern<<<grid, kTh, lds, s>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MMX_OK;
}

template <int DT, bool IOH>
int launch_dt_io(const AttnBwdArgs& a, dim3 gq, dim3 gk, hipStream_t s) {
    int rc = MMX_OK;
    // the query side: dQ (and delta for the key side), the dP slab when one is wanted
    if (a.need_dqkv || a.dprobs) rc = launch_v2(attn_bwd_q_bf16_kernel<DT, IOH>, a, gq, kQLds, s, "attn_bwd_q_bf16_kernel");
    if (rc) return rc;
    if (a.rel_v) {
        rc = a.need_dqkv ? launch_v2(attn_bwd_kv_bf16_kernel<DT, IOH, true, true>, a, gk, kKvLds, s, "attn_bwd_kv_bf16_kernel<rel>")
                         : launch_v2(attn_bwd_kv_bf16_kernel<DT, IOH, true, false>, a, gk, kKvLds, s, "attn_bwd_kv_bf16_kernel<rel only>");
        if (rc) return rc;
        return rel_row_update(a.rel_v, a.rel_part, a.rel_out, a.B, a.H, a.Nk, 1.0f / a.H, s);     // one partial row per head
    }
    if (a.need_dqkv) rc = launch_v2(attn_bwd_kv_bf16_kernel<DT, IOH, false, true>, a, gk, kKvLds, s, "attn_bwd_kv_bf16_kernel");
    return rc;
}

template <int DT>
int launch_dt(const AttnBwdArgs& a, dim3 gq, dim3 gk, hipStream_t s) {
    return a.io_bf16 ? launch_dt_io<DT, true>(a, gq, gk, s) : launch_dt_io<DT, false>(a, gq, gk, s);
}

bool al16(const void* p, const Strides& s, int elem) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && (s.sb * elem) % 16 == 0 && (s.sh * elem) % 16 == 0 && (s.sn * elem) % 16 == 0;
}

}  // namespace

void attn_bf16_v2_enable(int on) { g_attn_bf16_v2 = on & 1; }

// returns 1 if the second-generation bf16 kernels were launched (rc in *rc_out), 0 if the call is not eligible:
// bf16-MFMA mode with the forward's O at hand, head_dim % 8 == 0 (<= 64), 16-byte aligned operand rows.
int attn_bwd_bf16_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out) {
    if (!g_attn_bf16_v2 || !a.mma_bf16 || !a.o || a.D % 8 || a.D > kDP) return 0;
    if (!al16(a.v, a.vs, 4) || !al16(a.o, a.oos, 4) || !al16(a.dout, a.os, a.io_bf16 ? 2 : 4)) return 0;
    if (a.need_dqkv) {
        if (!al16(a.q, a.qs, 4) || !al16(a.k, a.ks, 4)) return 0;
        const int e = a.io_bf16 ? 2 : 4;
        if (!al16(a.dq, a.dqs, e) || reinterpret_cast<uintptr_t>(a.dk) % 4 || reinterpret_cast<uintptr_t>(a.dv) % 4 ||
            (a.dks.sn * e) % 4 || (a.dvs.sn * e) % 4 || (a.dks.sh * e) % 4 || (a.dvs.sh * e) % 4 || (a.dks.sb * e) % 4 ||
            (a.dvs.sb * e) % 4)
            return 0;
    }
    dim3 gq(((a.Nq + kR - 1) / kR) * a.H * a.B), gk(((a.Nk + kR - 1) / kR) * a.H * a.B);
    switch (a.slab_dt) {
        case MMX_F32: *rc_out = launch_dt<MMX_F32>(a, gq, gk, s); break;
        case MMX_F16: *rc_out = launch_dt<MMX_F16>(a, gq, gk, s); break;
        case MMX_BF16: *rc_out = launch_dt<MMX_BF16>(a, gq, gk, s); break;
        default: return 0;
    }
    return 1;
}

}  // namespace mmx
