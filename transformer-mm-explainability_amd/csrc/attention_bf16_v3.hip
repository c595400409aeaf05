// Third generation of the bf16-matrix-core attention BACKWARD, for the configuration BASELINE config 5 runs (CLIP
// ViT-L/14@336, 577 tokens x 64, batch 128 per GPU): SHARED forward (q / k / v / P of ONE image, batch stride 0), bf16 gradient
// stream (dO in, dq / dk / dv out), row-relevancy mode (no dP slab), q * d^-0.5 first.  Everything else stays on the
// second generation (attention_bf16.hip), whose tile arithmetic this file keeps.
//
// What the second generation paid per workgroup and per tile although the operands are the same for all B samples
// (profiles/r02_cfg5_probe.txt: 21-26 % of the wave cycles in VALU at 2 waves / SIMD, 36-51 % parked):
//   * K, V (query side) and Q (key side) were fetched as fp32 and rounded / transposed in registers by every workgroup --
//     B x (N / 64) times per head;
//   * the probabilities were 12-byte unaligned loads + funnel shifts (query side) and SIXTEEN 2-byte loads with a clamped
//     64-bit address each (key side: a lane owns a key, its 16 query rows are a strided column of the slab).
// Here two small PREP kernels run once per call on the shared operands (a few MB, L2-resident afterwards):
//   Vb  [H][Np][64]  V rows, bf16                     KbT / QbT  K / (scale q) transposed, bf16, staging-blocked (prep_qkv_kernel)
//   Pq / PT  P and its transpose, zero padded to Np = 64 ceil(N / 64), in OPERAND-BLOCKED order (p_block_offset below): the 8
//            words a lane needs of a (16 rows x 32 columns) half tile are 16 contiguous bytes and a wave's 64 lanes are 1 KB
//            contiguous.  (Row-major images made every 8-byte lane load of a wave a gather over 16 cache lines, 32 bytes used
//            of each: dense loads took 14 % off both kernels in the round-4 timing ablation, profiles/r04_cfg5_ablation.txt.)
// so that in the main kernels every LDS tile is a raw 8-byte copy of its global image (no ALU, no masks: the padding is
// zero), a lane's probabilities of a tile are four ALIGNED 8-byte loads on both sides, and on the key side those raw words
// ARE the bf16 A operand of dV = P^T . dO.  The padded row images also end the over-read of the 16-bit slab the second
// generation needed slack for: no load of this file touches the caller's slab outside the prep kernel,
// which reads it element-wise.  The tile loop is written in two 32-row halves so that only half of the score-shaped
// registers are live at a time (the second generation's 162-184 VGPRs capped it at 2 waves per SIMD).
#include "mmx_common.h"
#include "attention_args.h"

#include <cstdlib>
#include <type_traits>

namespace mmx {
namespace {

constexpr int kT = 64;        // rows of the streamed operand per step
constexpr int kD = 64;        // head dim (exactly)
constexpr int kLR = kD + 16;  // row-major LDS tile [row][d]: 80-element rows (conflict-free, see attention_bf16.hip)
constexpr int kLT = kT + 16;  // transposed LDS tile [d][row]: 80-element rows, 4-row group index XOR (d / 4) & 15

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

int g_attn_bf16_v3 = 3;       // 3 (default, round 6): query side as below, key side attn_bwd_kv_v4_kernel (32 keys per wave, 32x32x16 MFMA,
                              // transpose reads) | 2: the third-generation pair |
                              // 0: off (the call falls through to the second generation, attention_bf16.hip: the A / B arm of
                              // test_bf16_backward_third_generation_equals_second), non-zero: 8-wave workgroups, 4 waves / SIMD on both kernels
                              // (measured and removed in round 5: 4-wave workgroups; a fourth-generation key side with several key
                              // blocks per wave -- 882-980 us vs 780 us per layer pair, profiles/r04_cfg5_probe.txt)

struct V3Images {
    const bf16_t *Vb, *KbT, *QbT, *Pq, *PT;
    int Np;
    const bf16_t *Qb, *PT32;      // fourth-generation key side: (scale q) rows, row-major like Vb; P^T blocked for 32-key waves
};

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
__device__ __forceinline__ f32x4 unpack4(u32x2v w) { return f32x4{bflo(w[0]), bfhi(w[0]), bflo(w[1]), bfhi(w[1])}; }

// 4 rows x 4 columns of bf16 (rows[e] = columns c, c+1 | c+2, c+3 of row e) -> column dd as 4 consecutive rows (8 bytes)
__device__ __forceinline__ u32x2v column_of(const u32x2v (&rows)[4], int dd) {
    const int w = dd >> 1;
    const unsigned sel = (dd & 1) ? 0x07060302u : 0x05040100u;
    return u32x2v{__builtin_amdgcn_perm(rows[1][w], rows[0][w], sel), __builtin_amdgcn_perm(rows[3][w], rows[2][w], sel)};
}

// ======================================================================================================= prep kernels
// which = 0: Vb (row-major), 1: KbT, 2: QbT (times scale).  One workgroup per (64-row tile, head, which); fp32 in, 16-byte
// aligned rows (checked by the caller), rows >= N are written as zeros.
__global__ __launch_bounds__(256) void prep_qkv_kernel(const AttnBwdArgs a, bf16_t* Vb, bf16_t* KbT, bf16_t* QbT, bf16_t* Qb, int Np) {
    const int tid = threadIdx.x, row0 = blockIdx.x * kT, h = blockIdx.y, which = blockIdx.z;
    const int r4 = 4 * (tid >> 4), c = 4 * (tid & 15);
    const float* base = which == 0 ? a.v + h * a.vs.sh : which == 1 ? a.k + h * a.ks.sh : a.q + h * a.qs.sh;
    const int64_t sn = which == 0 ? a.vs.sn : which == 1 ? a.ks.sn : a.qs.sn;
    const float mul = which >= 2 ? a.scale : 1.f;                      // which == 3: (scale q) rows, row-major (4th-generation key side)
    u32x2v rows[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = row0 + r4 + e;
        rows[e] = u32x2v{0u, 0u};
        if (row < a.Nk) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(row) * sn + c);
            rows[e] = u32x2v{pk2(x[0] * mul, x[1] * mul), pk2(x[2] * mul, x[3] * mul)};
        }
    }
    if (which == 0 || which == 3) {
        bf16_t* dst = which == 0 ? Vb : Qb;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *reinterpret_cast<u32x2v*>(dst + (static_cast<int64_t>(h) * Np + row0 + r4 + e) * kD + c) = rows[e];
    } else {
        bf16_t* out = which == 1 ? KbT : QbT;
        // transposed images, staging-blocked: the 4 (d) x 4 (rows) block a staging thread moves per tile is 32 contiguous bytes
        // at [h][tile][thread][dd][4 rows] -- a wave's fetch is 2 KB contiguous (as [d][row] it was a 16-line gather per load)
        u32x4v* dst = reinterpret_cast<u32x4v*>(out + ((static_cast<int64_t>(h) * (Np / kT) + blockIdx.x) * 256 + tid) * 16);
        const u32x2v c0 = column_of(rows, 0), c1 = column_of(rows, 1), c2 = column_of(rows, 2), c3 = column_of(rows, 3);
        dst[0] = u32x4v{c0[0], c0[1], c1[0], c1[1]};
        dst[1] = u32x4v{c2[0], c2[1], c3[0], c3[1]};
    }
}

// Operand-blocked image of a matrix M [Np][Np] of head h: block (rb = row / 16, ct = column / 64, pp) holds, for lane = 16 g + i,
// the 8 elements M[16 rb + i][64 ct + 16 (2 pp + hh) + 4 g + r] in the order (hh, r) -- lane (i, g)'s words of the two 16 x 16
// sub-tiles of half pp, i.e. one bf16x8 MFMA operand.  Offsets in elements.
__host__ __device__ __forceinline__ int64_t p_block_offset(int h, int rb, int ct, int Np) {
    return ((static_cast<int64_t>(h) * (Np / 16) + rb) * (Np / kT) + ct) * (2 * 64 * 8);
}

// Fourth-generation key side (32 keys per wave, v_mfma_f32_32x32x16_bf16): block (kb = key / 32, qt = query / 64, pp) holds, for
// lane = 32 hi + i, the 16 elements P[64 qt + 32 pp + (r & 3) + 8 (r >> 2) + 4 hi][32 kb + i], r = 0 .. 15 -- the lane's entries of
// the 32 x 32 accumulator tile dP[query][key], and at the same time (r = 8 m + j) the two bf16x8 A operands of dV = P^T . dO.
__host__ __device__ __forceinline__ int64_t p32_block_offset(int h, int kb, int qt, int Np) {
    return ((static_cast<int64_t>(h) * (Np / 32) + kb) * (Np / kT) + qt) * (2 * 64 * 16);
}

// P slab [H][N][N] (16-bit elements, any alignment) -> Pq (blocked image of P, zero padded) and PT (blocked image of P^T; `wide`:
// in the 32-key form p32_block_offset describes)
__global__ __launch_bounds__(256) void prep_p_kernel(const bf16_t* __restrict__ P, bf16_t* __restrict__ Pq, bf16_t* __restrict__ PT,
                                                     int N, int Np, int wide) {
    __shared__ bf16_t tile[kT][kT + 2];
    const int tid = threadIdx.x, k0 = blockIdx.x * kT, q0 = blockIdx.y * kT, h = blockIdx.z;
    const bf16_t* src = P + static_cast<int64_t>(h) * N * N;
    for (int idx = tid; idx < kT * kT; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        bf16_t v = 0;
        if (q0 + r < N && k0 + c < N) v = src[static_cast<int64_t>(q0 + r) * N + k0 + c];
        tile[r][c] = v;
    }
    __syncthreads();
    // 4 row blocks x 2 halves x 64 lanes = 512 operand words of 16 bytes per image: two per thread and image
#pragma unroll
    for (int e = tid; e < 512; e += 256) {
        const int rbl = e >> 7, pp = (e >> 6) & 1, lane = e & 63, i = lane & 15, g = lane >> 4;
        bf16_t wq[8], wt[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rbl + i, col = 16 * (2 * pp + hh) + 4 * g + r;
                wq[4 * hh + r] = tile[row][col];
                wt[4 * hh + r] = tile[col][row];
            }
        const int64_t oq = p_block_offset(h, q0 / 16 + rbl, k0 / kT, Np) + (pp * 64 + lane) * 8;
        const int64_t ot = p_block_offset(h, k0 / 16 + rbl, q0 / kT, Np) + (pp * 64 + lane) * 8;
        *reinterpret_cast<u32x4v*>(Pq + oq) = *reinterpret_cast<const u32x4v*>(wq);
        if (!wide) *reinterpret_cast<u32x4v*>(PT + ot) = *reinterpret_cast<const u32x4v*>(wt);
    }
    if (wide) {                                                    // 2 key blocks x 2 halves x 64 lanes: one 32-byte record per thread
        const int kbl = tid >> 7, pp = (tid >> 6) & 1, lane = tid & 63, i = lane & 31, hi = lane >> 5;
        bf16_t w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = tile[32 * pp + (r & 3) + 8 * (r >> 2) + 4 * hi][32 * kbl + i];
        u32x4v* dst = reinterpret_cast<u32x4v*>(PT + p32_block_offset(h, k0 / 32 + kbl, q0 / kT, Np) + (pp * 64 + lane) * 16);
        dst[0] = *reinterpret_cast<const u32x4v*>(w);
        dst[1] = *reinterpret_cast<const u32x4v*>(w + 8);
    }
}

// ======================================================================================================= shared pieces
// the wave's 16 rows of a bf16 operand as MFMA operand registers: lane (i, g) holds row `row`, d = 32 pr + 8 g .. + 7
__device__ __forceinline__ void load_rows8(bf16x8 (&op)[kD / 32], const bf16_t* base, int64_t sn, int row, bool row_ok, int g) {
#pragma unroll
    for (int pr = 0; pr < kD / 32; ++pr) {
        const u32x4v v = *reinterpret_cast<const u32x4v*>(base + static_cast<int64_t>(row) * sn + 32 * pr + 8 * g);
        op[pr] = as_bf16x8(row_ok ? v : u32x4v{0u, 0u, 0u, 0u});
    }
}

// one operand of the "contract over the tile's row index" products, from a TRANSPOSED tile [d][row]: row-slots (g, j) of
// tile pair p (rows 32 p + 16 (j >> 2) + 4 g + (j & 3)) for d = 16 dt + i.  The swizzled 4-row group index
// (8 p + 4 hi + g) ^ ((4 dt + (i >> 2)) & 15) splits into a per-lane part (g ^ (i >> 2), bits 0-1) and a compile-time part
// (dt ^ (2 p + hi), bits 2-3), so ONE lane address (transposed_lane_base) + immediate offsets serve all 16 reads of a tile
// (written with the XOR on the whole index the compiler kept 16-32 address registers live across the loop).
__device__ __forceinline__ int transposed_lane_base(int i, int g) { return i * kLT + 4 * (g ^ (i >> 2)); }
__device__ __forceinline__ bf16x8 transposed_operand(const bf16_t* tile_lane, int dt, int p) {
    // two SEPARATE ds_read_b64 (2 LDS cycles each) instead of the ds_read2_b64 hipcc merges them into (8 cycles: each of its two
    // accesses is serviced as 4 x 16 lanes) -- and no register shuffle when the pieces come out in descending address order
    // (volatile keeps the two loads apart; the explicit LDS address space keeps them ds_read -- a volatile generic load is a flat_load)
    typedef const volatile __attribute__((address_space(3))) u32x2v* lds_b64_ptr;
    const u32x2v lo = *(lds_b64_ptr)(tile_lane + 16 * dt * kLT + 16 * (dt ^ (2 * p)));
    const u32x2v hi = *(lds_b64_ptr)(tile_lane + 16 * dt * kLT + 16 * (dt ^ (2 * p + 1)));
    return as_bf16x8(u32x4v{lo[0], lo[1], hi[0], hi[1]});
}

// a [64 x 64] bf16 tile of a row-major image (row stride `sn` elements): thread st (0..255) owns a 4 x 4 block
struct RawBlock { u32x2v raw[4]; };
__device__ __forceinline__ void fetch_row_major(RawBlock& blk, const bf16_t* img, int64_t sn, int row0, int st) {
    const int r4 = 4 * (st >> 4), c = 4 * (st & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) blk.raw[e] = *reinterpret_cast<const u32x2v*>(img + static_cast<int64_t>(row0 + r4 + e) * sn + c);
}
__device__ __forceinline__ void store_row_major(bf16_t* tile, const u32x2v (&rows)[4], int st) {
    const int r4 = 4 * (st >> 4), c = 4 * (st & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x2v*>(tile + (r4 + e) * kLR + c) = rows[e];
}
// the same block of a TRANSPOSED operand: raw[dd] = rows row0 + r4 .. + 3 of d = c + dd, from the staging-blocked image
// [tile][thread][dd][4 rows] of one head (prep_qkv_kernel): 32 contiguous bytes per thread.  (Rounds 3-4 kept a [d][row] image:
// four 8-byte loads per thread, each a 16-line gather per wave -- and, until round 4, FLAT loads with an s_waitcnt vmcnt(0) in
// front of each, because the wave-uniform base was rebuilt through an integer -> pointer cast; profiles/r04_cfg5_probe.txt.)
__device__ __forceinline__ void fetch_transposed(RawBlock& blk, const bf16_t* imgT, int Np, int row0, int st) {
    const u32x4v* src = reinterpret_cast<const u32x4v*>(imgT + (static_cast<int64_t>(row0 / kT) * 256 + st) * 16);
    const u32x4v lo = src[0], hi = src[1];
    blk.raw[0] = u32x2v{lo[0], lo[1]};
    blk.raw[1] = u32x2v{lo[2], lo[3]};
    blk.raw[2] = u32x2v{hi[0], hi[1]};
    blk.raw[3] = u32x2v{hi[2], hi[3]};
}
__device__ __forceinline__ void store_transposed_raw(bf16_t* tile, const u32x2v (&cols)[4], int st) {
    const int c = 4 * (st & 15);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
        *reinterpret_cast<u32x2v*>(tile + (c + dd) * kLT + 4 * ((st >> 4) ^ (st & 15))) = cols[dd];
}

// ===================================================================================================== query side
// delta = rowsum(dO * O), dP^T = V . dO^T, dS, dQ = dS . K  (orientation and accumulator layouts: attention_bf16.hip)
//
// A workgroup owns 16 NW query rows of NS consecutive SAMPLES of one head: the V / K tiles, the probabilities and every LDS
// operand read of a tile are the same for all samples (shared forward), so they are fetched ONCE and used NS times -- the
// shared operands were 3.3 GB of L2 -> CU traffic per launch at one sample per workgroup (4.7 TB/s, i.e. what bounded it).
// Loads run TWO tiles ahead of their use (two register sets, loop unrolled by two): with one set the fetch of tile t + 2 could
// only be issued after the wait for tile t + 1, so at most one tile per workgroup was ever in flight.
template <int NW, int NS, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 3) void attn_bwd_q_v3_kernel(const AttnBwdArgs a, const V3Images im) {
    constexpr int NB = kD / 16, R = 16 * NW, NOP = NW == 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* Vt = reinterpret_cast<bf16_t*>(smem_raw);                 // [2][kT][kLR]   V rows (row-major)
    bf16_t* Kt = Vt + 2 * kT * kLR;                                   // [2][kD][kLT]   K transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int st = tid & 255;
    const bool stage_v = NW == 4 || tid < 256;
    const int nrt = (a.Nq + R - 1) / R, nbg = (a.B + NS - 1) / NS;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    // row tile fastest, then the sample group, the head slowest: the workgroups an XCD runs at a time (a contiguous id range)
    // share one or two heads' K / V / P images (L2-resident, ~1 MB per head)
    const int rt = wg % nrt, b0 = ((wg / nrt) % nbg) * NS, h = wg / (nrt * nbg);
    const int q = rt * R + wave * 16 + i;
    const bool q_ok = q < a.Nq;
    const int qc = min(q, a.Nq - 1);
    const bool wave_live = __builtin_amdgcn_readfirstlane(rt * R + wave * 16) < a.Nq;

    // per sample: the wave's dO rows as the B operand of dP^T, and delta = rowsum(dO * O) of this lane's row
    bf16x8 dob[NS][kD / 32];
    float delta[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int b = min(b0 + s, a.B - 1);                            // (a ragged last group recomputes the last sample; not stored)
        const bf16_t* src = reinterpret_cast<const bf16_t*>(a.dout) + b * a.os.sb + h * a.os.sh;
        load_rows8(dob[s], src, a.os.sn, qc, q_ok, g);
        const float* ob = a.o + h * a.oos.sh + static_cast<int64_t>(qc) * a.oos.sn;
        float part = 0.f;
#pragma unroll
        for (int pr = 0; pr < kD / 32; ++pr) {
            const int d0 = 32 * pr + 8 * g;
            const u32x4v w = __builtin_bit_cast(u32x4v, dob[s][pr]);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(ob + d0), o1 = *reinterpret_cast<const f32x4*>(ob + d0 + 4);
            part += bflo(w[0]) * o0[0] + bfhi(w[0]) * o0[1] + bflo(w[1]) * o0[2] + bfhi(w[1]) * o0[3] +
                    bflo(w[2]) * o1[0] + bfhi(w[2]) * o1[1] + bflo(w[3]) * o1[2] + bfhi(w[3]) * o1[3];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        delta[s] = q_ok ? part : 0.f;
        if (g == 0 && q_ok && b0 + s < a.B) a.delta[(static_cast<int64_t>(b) * a.H + h) * a.Nq + q] = delta[s];
    }
    // this wave's row block of the blocked P image (key tile kt: + 1024 kt elements); a block past the image repeats the last one
    const bf16_t* prow = im.Pq + p_block_offset(h, min(rt * (R / 16) + wave, im.Np / 16 - 1), 0, im.Np) + 8 * lane;
    const bf16_t* vimg = im.Vb + static_cast<int64_t>(h) * im.Np * kD;
    const bf16_t* kimg = im.KbT + static_cast<int64_t>(h) * kD * im.Np;
    const int ntiles = (a.Nk + kT - 1) / kT;
    const int tlane = transposed_lane_base(i, g);

    f32x4 qacc[NS][NB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int dt = 0; dt < NB; ++dt) qacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // NW == 4: every thread stages a block of V and one of K; NW == 8: waves 0-3 stage V, waves 4-7 K (ONE block per set)
    RawBlock rs[2][NOP];
    // [half pp]: the words of sub-tiles 2 pp, 2 pp + 1.  The products read p_cur, which is only ever COPIED (top of `body`) out of
    // p_nxt, the destination of the loads: with the words loaded straight into the registers the MFMAs read (rounds 3-5: two sets,
    // two tiles ahead) hipcc's wait-count pass -- which merges the pending-load state of the prologue, the early exits and the
    // exec-masked staging branches -- put s_waitcnt vmcnt(1) / vmcnt(0) INSIDE the product block, i.e. every tile waited for the
    // operand fetch it had just issued (round 6, found in the ISA: the query side's 48 % SQ_WAIT_ANY of profiles/r05_cfg5_probe.txt)
    u32x4v p_cur[2], p_nxt[2];
    auto fetch = [&](int kt, auto set) {
        constexpr int S = decltype(set)::value;
        if constexpr (NW == 4) {
            fetch_row_major(rs[S][0], vimg, kD, kt * kT, st);
            fetch_transposed(rs[S][1], kimg, im.Np, kt * kT, st);
        } else {
            if (stage_v) fetch_row_major(rs[S][0], vimg, kD, kt * kT, st);
            else fetch_transposed(rs[S][0], kimg, im.Np, kt * kT, st);
        }
    };
    auto stage = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
        if constexpr (NW == 4) {
            store_row_major(Vt + buf * kT * kLR, rs[S][0].raw, st);
            store_transposed_raw(Kt + buf * kD * kLT, rs[S][1].raw, st);
        } else {
            if (stage_v) store_row_major(Vt + buf * kT * kLR, rs[S][0].raw, st);
            else store_transposed_raw(Kt + buf * kD * kLT, rs[S][0].raw, st);
        }
    };
    auto p_issue = [&](u32x4v (&raw)[2], int kt) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) raw[pp] = *reinterpret_cast<const u32x4v*>(prow + kt * 1024 + pp * 512);
    };
    // tile kt: LDS buffer / register set / probability set kt & 1 (= PAR)
    auto body = [&](int kt, auto par) {
        constexpr int PAR = decltype(par)::value;
        p_cur[0] = p_nxt[0];
        p_cur[1] = p_nxt[1];
        if (kt + 1 < ntiles) p_issue(p_nxt, kt + 1);                   // consumed one iteration from now
        if (ABL != 2 && ABL != 6 && kt + 2 < ntiles) fetch(kt + 2, par);   // this set's tile kt went to LDS one iteration ago
        if (ABL != 2 && ABL != 6 && kt + 1 < ntiles) stage(1 - PAR, std::integral_constant<int, 1 - PAR>{});   // waits for THAT set only
        const bf16_t* Vcur = Vt + PAR * kT * kLR;
        const bf16_t* Kcur = Kt + PAR * kD * kLT + tlane;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                               // keys 32 pp .. + 31 of the tile
            // a half tile with no real key in it (577 = 9 x 64 + 1: the second half of the last tile) has P = 0: nothing to add;
            // a wave whose 16 query rows all lie past Nq (577 = 4 x 128 + 65: three of the last workgroup's eight waves) only stages
            if ((pp == 1 && kt * kT + 32 >= a.Nk) || !wave_live) break;
            bf16x8 dsb[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 dpT[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int pr = 0; pr < kD / 32; ++pr)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const bf16x8 op = *reinterpret_cast<const bf16x8*>(Vcur + (16 * (2 * pp + hh) + i) * kLR + 32 * pr + 8 * g);
                        dpT[hh] = mfma16x16x32_bf16(op, dob[s][pr], dpT[hh]);
                    }
                // dS[q][key] = P * (dP - delta) for keys 16 t + 4 g + r, t = 2 pp + hh (scale_mode Q_FIRST: no further factor)
                dsb[s] = pack8(unpack4(u32x2v{p_cur[pp][0], p_cur[pp][1]}) * (dpT[0] - delta[s]),
                               unpack4(u32x2v{p_cur[pp][2], p_cur[pp][3]}) * (dpT[1] - delta[s]));
                // the V operands are RE-READ from LDS for the next sample (16 registers the budget of 4 waves / SIMD does not
                // have): the clobber keeps the compiler from carrying them over
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int dt = 0; dt < NB; ++dt) {
                const bf16x8 kop = transposed_operand(Kcur, dt, pp);
#pragma unroll
                for (int s = 0; s < NS; ++s) qacc[s][dt] = mfma16x16x32_bf16(kop, dsb[s], qacc[s][dt]);
            }
            __builtin_amdgcn_sched_barrier(0);                        // keep the two halves apart: bounds the live registers
        }
        lds_barrier();
    };

    fetch(0, std::integral_constant<int, 0>{});
    if (ntiles > 1) fetch(1, std::integral_constant<int, 1>{});
    p_issue(p_nxt, 0);
    stage(0, std::integral_constant<int, 0>{});
    __syncthreads();
    for (int kt = 0; kt < ntiles; kt += 2) {
        body(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < ntiles) body(kt + 1, std::integral_constant<int, 1>{});
    }
    if (!q_ok) return;
    // dQ^T accumulators: lane (q = column i), rows d = 16 dt + 4 g + r: 4 consecutive d of one query row
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (b0 + s >= a.B) break;
        const int64_t off = (b0 + s) * a.dqs.sb + h * a.dqs.sh + static_cast<int64_t>(q) * a.dqs.sn;
#pragma unroll
        for (int dt = 0; dt < NB; ++dt) {
            const f32x4 v = qacc[s][dt] * a.scale;
            *reinterpret_cast<u32x2v*>(reinterpret_cast<bf16_t*>(a.dq) + off + 16 * dt + 4 * g) = u32x2v{pk2(v[0], v[1]), pk2(v[2], v[3])};
        }
    }
}

// ===================================================================================================== key side
// per 16 NW keys: dP recomputed, dV = P^T . dO, dK = dS^T . Q (DKV), and the row-relevancy partial of this head
// rel_part[b][h][key] = sum_q rel_v[b][q] * clamp(dP * P, 0)[q][key]
template <int NW, bool DKV, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 3) void attn_bwd_kv_v3_kernel(const AttnBwdArgs a, const V3Images im) {
    constexpr int NB = kD / 16, R = 16 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* dOr = reinterpret_cast<bf16_t*>(smem_raw);                // [2][kT][kLR]   dO rows (row-major)
    bf16_t* dOt = dOr + 2 * kT * kLR;                                 // [2][kD][kLT]   dO transposed
    bf16_t* Qt = dOt + 2 * kD * kLT;                                  // [2][kD][kLT]   (scale q) transposed
    float* dl = reinterpret_cast<float*>(Qt + 2 * kD * kLT);          // [2][kT]        delta of the staged query rows
    float* vl = dl + 2 * kT;                                          // [2][kT]        rel_v of the staged query rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int st = tid & 255;
    const bool stage_do = NW == 4 || tid < 256, stage_q = DKV && (NW == 4 || tid >= 256);
    const int nkt = (a.Nk + R - 1) / R;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int b = (wg / nkt) % a.B, h = wg / (nkt * a.B);          // key tile fastest, then the sample, the head slowest (see the query side)
    const int kw = (wg % nkt) * R + wave * 16;                        // first key of this wave
    const int key = kw + i;
    const bool key_ok = key < a.Nk;
    const bool wave_live = __builtin_amdgcn_readfirstlane(kw) < a.Nk;
    const int keyc = min(key, a.Nk - 1);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    // this wave's key block of the blocked P^T image (query tile qt: + 1024 qt elements); a block past the image repeats the last
    const bf16_t* pcol = im.PT + p_block_offset(h, min(kw / 16, im.Np / 16 - 1), 0, im.Np) + 8 * lane;
    const bf16_t* qimg = im.QbT + static_cast<int64_t>(h) * kD * im.Np;
    const bf16_t* dobase = reinterpret_cast<const bf16_t*>(a.dout) + b * a.os.sb + h * a.os.sh;

    bf16x8 vop[kD / 32];                                              // this wave's V rows: B operand of dP = dO . V^T
    load_rows8(vop, im.Vb + static_cast<int64_t>(h) * im.Np * kD, kD, keyc, key_ok, g);

    f32x4 kacc[NB], vacc[NB];
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) kacc[dt] = vacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (a.Nq + kT - 1) / kT;
    const int tlane = transposed_lane_base(i, g);
    RawBlock doreg, qreg_;
    RawBlock& qreg = NW == 4 ? qreg_ : doreg;                        // NW == 8: waves 4-7 stage Q, ONE register block per thread
    int do_row0 = 0;
    float dlreg = 0.f, vlreg = 0.f, racc = 0.f;
    auto fetch = [&](int qt) {
        if (stage_do) {                                               // per-sample operand: clamped rows, masked at the store
            const int r4 = 4 * (st >> 4), c = 4 * (st & 15);
            do_row0 = qt * kT;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                doreg.raw[e] = *reinterpret_cast<const u32x2v*>(dobase + static_cast<int64_t>(min(qt * kT + r4 + e, a.Nq - 1)) * a.os.sn + c);
        }
        if (stage_q) fetch_transposed(qreg, qimg, im.Np, qt * kT, st);
        if (tid < kT) {
            const int row = min(qt * kT + tid, a.Nq - 1);
            if constexpr (DKV) dlreg = a.delta[head * a.Nq + row];
            vlreg = a.rel_v[static_cast<int64_t>(b) * a.Nq + row];
        }
    };
    auto stage = [&](int buf) {
        if (stage_do) {
            const int r4 = 4 * (st >> 4);
            u32x2v rows[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rows[e] = do_row0 + r4 + e < a.Nq ? doreg.raw[e] : u32x2v{0u, 0u};
            store_row_major(dOr + buf * kT * kLR, rows, st);
            if constexpr (DKV) {
                u32x2v cols[4];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) cols[dd] = column_of(rows, dd);
                store_transposed_raw(dOt + buf * kD * kLT, cols, st);
            }
        }
        if (stage_q) store_transposed_raw(Qt + buf * kD * kLT, qreg.raw, st);
        if (tid < kT) {
            if constexpr (DKV) dl[buf * kT + tid] = dlreg;
            vl[buf * kT + tid] = vlreg;                                // (rows past Nq: p is zero there)
        }
    };
    u32x4v p_cur[2], p_nxt[2];                                        // [half pp]: the words of sub-tiles 2 pp, 2 pp + 1
    auto p_issue = [&](u32x4v (&raw)[2], int qt) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) raw[pp] = *reinterpret_cast<const u32x4v*>(pcol + qt * 1024 + pp * 512);
    };
    fetch(0);
    stage(0);
    p_issue(p_cur, 0);
    if (ntiles > 1) fetch(1);
    __syncthreads();
    for (int qt = 0; qt < ntiles; ++qt) {
        const int cur = qt & 1;
        if (qt + 1 < ntiles) {
            if (ABL != 2 && ABL != 6) stage(cur ^ 1);
            p_issue(p_nxt, qt + 1);
            if (ABL != 2 && ABL != 6 && qt + 2 < ntiles) fetch(qt + 2);
        }
        const bf16_t* dOrc = dOr + cur * kT * kLR;
        const bf16_t* dOtc = dOt + cur * kD * kLT + tlane;
        const bf16_t* Qtc = Qt + cur * kD * kLT + tlane;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                               // query rows 32 pp .. + 31 of the tile
            // a half tile with no real query row in it (577 = 9 x 64 + 1: the second half of the last tile): P = 0 and dO = 0 there;
            // a wave whose 16 keys all lie past Nk only stages
            if ((pp == 1 && qt * kT + 32 >= a.Nq) || !wave_live) break;
            f32x4 dp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // dp[hh][r] = dP[16 t + 4 g + r][key i]
#pragma unroll
            for (int pr = 0; pr < kD / 32; ++pr)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const bf16x8 op = *reinterpret_cast<const bf16x8*>(dOrc + (16 * (2 * pp + hh) + i) * kLR + 32 * pr + 8 * g);
                    dp[hh] = mfma16x16x32_bf16(op, vop[pr], dp[hh]);
                }
            f32x4 ds[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * pp + hh;
                const f32x4 p = unpack4(u32x2v{p_cur[pp][2 * hh], p_cur[pp][2 * hh + 1]});
                const f32x4 vv = *reinterpret_cast<const f32x4*>(vl + cur * kT + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ABL == 3 || ABL == 7) racc += dp[hh][r];
                    else racc += vv[r] * relu_nan(p[r] * dp[hh][r]);
                }
                if constexpr (DKV) {
                    const f32x4 dlv = *reinterpret_cast<const f32x4*>(dl + cur * kT + 16 * t + 4 * g);
                    ds[hh] = p * (dp[hh] - dlv);
                }
            }
            if constexpr (DKV) {
                // the raw probability words of the two sub-tiles ARE the bf16 A operand of dV = P^T . dO
                // (the two 8-byte pieces of a transposed operand are read with two SEPARATE ds_read_b64: no merge into a half-rate
                // ds_read2_b64, no register shuffle)
                const bf16x8 p_op = as_bf16x8(p_cur[pp]);
                const bf16x8 ds_op = ABL == 4 || ABL == 7 ? p_op : pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < NB; ++dt) {
                    vacc[dt] = mfma16x16x32_bf16(p_op, transposed_operand(dOtc, dt, pp), vacc[dt]);
                    kacc[dt] = mfma16x16x32_bf16(ds_op, transposed_operand(Qtc, dt, pp), kacc[dt]);
                }
            }
            // pin this half's share of the relevancy sum here: left alone the compiler sinks all 16 multiply / clamp / fma
            // triples of a tile behind the second half and keeps both halves' p, dP and rel_v registers live until then
            asm volatile("" : "+v"(racc));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qt + 1 < ntiles) {
#pragma unroll
            for (int t = 0; t < 2; ++t) p_cur[t] = p_nxt[t];
        }
        lds_barrier();
    }
    racc += __shfl_xor(racc, 16);
    racc += __shfl_xor(racc, 32);                                  // the 4 row groups of the tile rows: all queries of key i
    if (g == 0 && key_ok) a.rel_part[head * a.Nk + key] = racc;
    if constexpr (!DKV) return;
    // accumulators: lane (d = 16 dt + i), rows key = kw + 4 g + r; lanes i / i ^ 1 pair up so that every store is 4 bytes
    const int64_t dk0 = b * a.dks.sb + h * a.dks.sh, dv0 = b * a.dvs.sb + h * a.dvs.sh;
    const bool odd = i & 1;
#pragma unroll
    for (int dt = 0; dt < NB; ++dt) {
        const int d = 16 * dt + i;
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
            if (j < a.Nk) {
                bf16_t* dk = reinterpret_cast<bf16_t*>(a.dk) + dk0 + static_cast<int64_t>(j) * a.dks.sn + c0;
                bf16_t* dv = reinterpret_cast<bf16_t*>(a.dv) + dv0 + static_cast<int64_t>(j) * a.dvs.sn + c0;
                *reinterpret_cast<unsigned*>(dk) = odd ? pk2(ko, km) : pk2(km, ko);
                *reinterpret_cast<unsigned*>(dv) = odd ? pk2(vo, vm) : pk2(vm, vo);
            }
        }
    }
}

// ===================================================================================================== key side, 4th generation
// The same three products per query tile (dP recomputed, dV = P^T . dO, dK = dS^T . Q) on v_mfma_f32_32x32x16_bf16 with 32 keys
// per wave: an LDS operand (a [32 x 16] slice of the dO / Q tile) now feeds a 32 x 32 accumulator tile instead of 16 x 16, i.e.
// HALF the LDS operand bytes per FLOP -- the third generation's key side spent more LDS-array cycles per tile (16 waves x 160) than
// matrix-pipe cycles (profiles/r05_cfg5_probe.txt: 30 % of the wave cycles issuing, 38 % waiting on an instruction pipe).  The
// "contract over the query index" operands are read with ds_read_b64_tr_b16 straight from the ROW-MAJOR dO / Q tiles (a 16-lane
// group reads a 4-row x 16-column block and lane c gets column c's four rows): no transposed LDS images (the third generation
// staged dO twice and Q pre-transposed), no v_perm transposes in the staging waves.  Lane maps (tools/hip/tr16_probe.hip checks
// them on the hardware): A[i = l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][n = l & 31], C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// 4 waves = 128 keys per workgroup, <= 256 registers: two workgroups per CU.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
// What was measured around this kernel in round 6 and NOT kept (in-process A / B on one MI355X, rocprofv3 kernel trace of
// tools/probe_attn_v3.py, profiles/r06_cfg5_probe.txt): unpadded 128-byte rows with XOR-swizzled 16-byte slots (conflict-free for both
// read patterns: SQ_LDS_BANK_CONFLICT 10.5 % -> 2.5 % of the wave cycles, time +/- 0), clamp(x, 0) as (x + |x|) / 2, dS as
// fma(-p, delta, p dP), and a half-tile software pipeline over three LDS buffers (dP of the next half issued ahead of the VALU work of
// the current one, the barrier between the halves): 487-540 us against this form's 455 us and the third generation's 500 us -- the
// kernel is bound by VALU issue (~12 VALU instructions per 32-cycle MFMA: SQ_ACTIVE_INST_VALU 29 % of the wave cycles at two waves
// per SIMD), not by the matrix pipe or the LDS, and none of those moves VALU instructions off the critical path.
// B operand "rows of the tile are the contraction index" for k-slots (hi, j) <-> tile row 32 pp + 16 m + 8 (j >> 2) + 4 hi + (j & 3),
// column n = 32 db + (l & 31): two transpose reads of 4 rows x 16 columns per 16-lane group
__device__ __forceinline__ int tr_lane_base(int lane) {
    const int a = lane & 15, gq = lane >> 4;
    return (4 * (gq >> 1) + (a >> 2)) * kLR + 16 * (gq & 1) + 4 * (a & 3);
}
__device__ __forceinline__ bf16x8 tr_operand(const bf16_t* tile_lane, int pp, int m, int db) {
    typedef __attribute__((address_space(3))) bf16x4v* lds_tr_ptr;
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_ptr)(tile_lane + (32 * pp + 16 * m) * kLR + 32 * db));
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr_ptr)(tile_lane + (32 * pp + 16 * m + 8) * kLR + 32 * db));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <bool DKV>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_v4_kernel(const AttnBwdArgs a, const V3Images im) {
    constexpr int R = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* dOr = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Qr = dOr + 2 * kT * kLR;
    float* dl = reinterpret_cast<float*>(Qr + 2 * kT * kLR);
    float* vl = dl + 2 * kT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
    const int nkt = (a.Nk + R - 1) / R;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int b = (wg / nkt) % a.B, h = wg / (nkt * a.B);
    const int kw = (wg % nkt) * R + wave * 32;
    const int key = kw + i;
    const bool key_ok = key < a.Nk;
    const bool wave_live = __builtin_amdgcn_readfirstlane(kw) < a.Nk;
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    const bf16_t* pcol = im.PT32 + p32_block_offset(h, min(kw / 32, im.Np / 32 - 1), 0, im.Np) + 16 * lane;
    const bf16_t* qimg = im.Qb + static_cast<int64_t>(h) * im.Np * kD;
    const bf16_t* dobase = reinterpret_cast<const bf16_t*>(a.dout) + b * a.os.sb + h * a.os.sh;
    bf16x8 vop[kD / 16];
    {
        const bf16_t* vrow = im.Vb + (static_cast<int64_t>(h) * im.Np + min(key, im.Np - 1)) * kD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < kD / 16; ++ks) vop[ks] = as_bf16x8(*reinterpret_cast<const u32x4v*>(vrow + 16 * ks));
    }
    f32x16 kacc[2], vacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) kacc[db][r] = vacc[db][r] = 0.f;
    const int ntiles = (a.Nq + kT - 1) / kT;
    const int srow = tid >> 2, sseg = 16 * (tid & 3);
    u32x4v doreg[2], qreg[2];
    bool do_ok = false;
    float dlreg = 0.f, vlreg = 0.f, racc = 0.f;
    auto fetch = [&](int qt) {
        const int row = qt * kT + srow;
        do_ok = row < a.Nq;
        const u32x4v* src = reinterpret_cast<const u32x4v*>(dobase + static_cast<int64_t>(min(row, a.Nq - 1)) * a.os.sn + sseg);
        doreg[0] = src[0];
        doreg[1] = src[1];
        if constexpr (DKV) {
            const u32x4v* qs = reinterpret_cast<const u32x4v*>(qimg + static_cast<int64_t>(row) * kD + sseg);   // padded image: row < Np
            qreg[0] = qs[0];
            qreg[1] = qs[1];
        }
        if (tid < kT) {
            const int r = min(qt * kT + tid, a.Nq - 1);
            if constexpr (DKV) dlreg = a.delta[head * a.Nq + r];
            vlreg = a.rel_v[static_cast<int64_t>(b) * a.Nq + r];
        }
    };
    auto stage = [&](int buf) {
        const u32x4v z = {0u, 0u, 0u, 0u};
        u32x4v* dd = reinterpret_cast<u32x4v*>(dOr + buf * kT * kLR + srow * kLR + sseg);
        dd[0] = do_ok ? doreg[0] : z;
        dd[1] = do_ok ? doreg[1] : z;
        if constexpr (DKV) {
            u32x4v* qd = reinterpret_cast<u32x4v*>(Qr + buf * kT * kLR + srow * kLR + sseg);
            qd[0] = qreg[0];
            qd[1] = qreg[1];
        }
        if (tid < kT) {
            if constexpr (DKV) dl[buf * kT + tid] = dlreg;
            vl[buf * kT + tid] = vlreg;
        }
    };
    u32x4v p_cur[2][2], p_nxt[2][2];
    auto p_issue = [&](u32x4v (&raw)[2][2], int qt) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const u32x4v* src = reinterpret_cast<const u32x4v*>(pcol + qt * 2048 + pp * 1024);
            raw[pp][0] = src[0];
            raw[pp][1] = src[1];
        }
    };
    const int trl = tr_lane_base(lane);
    fetch(0);
    stage(0);
    p_issue(p_nxt, 0);
    if (ntiles > 1) fetch(1);
    __syncthreads();
    for (int qt = 0; qt < ntiles; ++qt) {
        const int cur = qt & 1;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            p_cur[pp][0] = p_nxt[pp][0];
            p_cur[pp][1] = p_nxt[pp][1];
        }
        if (qt + 1 < ntiles) {
            stage(cur ^ 1);
            p_issue(p_nxt, qt + 1);
            if (qt + 2 < ntiles) fetch(qt + 2);
        }
        const bf16_t* dOrc = dOr + cur * kT * kLR;
        const bf16_t* Qrc = Qr + cur * kT * kLR;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            if ((pp == 1 && qt * kT + 32 >= a.Nq) || !wave_live) break;
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < kD / 16; ++ks) {
                const bf16x8 op = *reinterpret_cast<const bf16x8*>(dOrc + (32 * pp + i) * kLR + 16 * ks + 8 * hi);
                dp = mfma32x32x16_bf16(op, vop[ks], dp);
            }
            f32x4 ds[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const u32x4v w = p_cur[pp][rg >> 1];
                const f32x4 p = unpack4(u32x2v{w[2 * (rg & 1)], w[2 * (rg & 1) + 1]});
                const f32x4 vv = *reinterpret_cast<const f32x4*>(vl + cur * kT + 32 * pp + 8 * rg + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) racc += vv[e] * relu_nan(p[e] * dp[4 * rg + e]);
                if constexpr (DKV) {
                    const f32x4 dlv = *reinterpret_cast<const f32x4*>(dl + cur * kT + 32 * pp + 8 * rg + 4 * hi);
                    ds[rg] = p * (f32x4{dp[4 * rg], dp[4 * rg + 1], dp[4 * rg + 2], dp[4 * rg + 3]} - dlv);
                }
            }
#pragma unroll
            for (int m = 0; m < (DKV ? 2 : 0); ++m) {
                const bf16x8 p_op = as_bf16x8(p_cur[pp][m]);
                const bf16x8 ds_op = pack8(ds[2 * m], ds[2 * m + 1]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    vacc[db] = mfma32x32x16_bf16(p_op, tr_operand(dOrc + trl, pp, m, db), vacc[db]);
                    kacc[db] = mfma32x32x16_bf16(ds_op, tr_operand(Qrc + trl, pp, m, db), kacc[db]);
                }
            }
            asm volatile("" : "+v"(racc));
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    }
    racc += __shfl_xor(racc, 32);
    if (hi == 0 && key_ok) a.rel_part[head * a.Nk + key] = racc;
    if constexpr (!DKV) return;
    const int64_t dk0 = b * a.dks.sb + h * a.dks.sh, dv0 = b * a.dvs.sb + h * a.dvs.sh;
    const bool odd = i & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int d = 32 * db + i;
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const int r = 2 * rp + (odd ? 1 : 0);
            const int j = kw + (r & 3) + 8 * (r >> 2) + 4 * hi, c0 = d - (odd ? 1 : 0);
            const float km = odd ? kacc[db][2 * rp + 1] : kacc[db][2 * rp];
            const float vm = odd ? vacc[db][2 * rp + 1] : vacc[db][2 * rp];
            const float ko = __int_as_float(__builtin_amdgcn_update_dpp(
                0, __float_as_int(odd ? kacc[db][2 * rp] : kacc[db][2 * rp + 1]), 0xB1, 0xF, 0xF, false));
            const float vo = __int_as_float(__builtin_amdgcn_update_dpp(
                0, __float_as_int(odd ? vacc[db][2 * rp] : vacc[db][2 * rp + 1]), 0xB1, 0xF, 0xF, false));
            if (j < a.Nk) {
                bf16_t* dk = reinterpret_cast<bf16_t*>(a.dk) + dk0 + static_cast<int64_t>(j) * a.dks.sn + c0;
                bf16_t* dv = reinterpret_cast<bf16_t*>(a.dv) + dv0 + static_cast<int64_t>(j) * a.dvs.sn + c0;
                *reinterpret_cast<unsigned*>(dk) = odd ? pk2(ko, km) : pk2(km, ko);
                *reinterpret_cast<unsigned*>(dv) = odd ? pk2(vo, vm) : pk2(vm, vo);
            }
        }
    }
}

constexpr size_t kQLds = sizeof(bf16_t) * (2 * kT * kLR + 2 * kD * kLT);
constexpr size_t kKvLds = sizeof(bf16_t) * (2 * kT * kLR + 4 * kD * kLT) + sizeof(float) * 4 * kT;

template <typename K>
int launch_v3(K kern, const AttnBwdArgs& a, const V3Images& im, dim3 grid, int threads, size_t lds, hipStream_t s, const char* name) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    kern<<<grid, threads, lds, s>>>(a, im);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MMX_OK;
}

template <int NW>
int run_v3(const AttnBwdArgs& a, const V3Images& im, hipStream_t s) {
    constexpr int R = 16 * NW;
    constexpr int NS = 2;                                                // samples per query-side workgroup
    dim3 gq(((a.Nq + R - 1) / R) * a.H * ((a.B + NS - 1) / NS)), gk(((a.Nk + R - 1) / R) * a.H * a.B);
    int rc = MMX_OK;
    const bool wide = g_attn_bf16_v3 >= 3;                              // fourth-generation key side (32 keys per wave)
    const dim3 gk4(((a.Nk + 127) / 128) * a.H * a.B);
    if (wide) {
        constexpr size_t lds4 = sizeof(bf16_t) * 4 * kT * kLR + sizeof(float) * 4 * kT;
        if (a.need_dqkv) {
            rc = launch_v3(attn_bwd_q_v3_kernel<NW, NS>, a, im, gq, 64 * NW, kQLds, s, "attn_bwd_q_v3_kernel");
            if (rc) return rc;
            rc = launch_v3(attn_bwd_kv_v4_kernel<true>, a, im, gk4, 256, lds4, s, "attn_bwd_kv_v4_kernel");
        } else {
            rc = launch_v3(attn_bwd_kv_v4_kernel<false>, a, im, gk4, 256, lds4, s, "attn_bwd_kv_v4_kernel<rel only>");
        }
        if (rc) return rc;
        return rel_row_update(a.rel_v, a.rel_part, a.rel_out, a.B, a.H, a.Nk, 1.0f / a.H, s);
    }
    if (a.need_dqkv) {
#ifdef MMX_ATTN_ABLATE
        static const int abl = getenv("MMX_ATTN_ABLATE") ? atoi(getenv("MMX_ATTN_ABLATE")) : 0;
        auto qk = abl == 1 ? attn_bwd_q_v3_kernel<NW, NS, 1> : abl == 2 ? attn_bwd_q_v3_kernel<NW, NS, 2>
                : abl == 6 ? attn_bwd_q_v3_kernel<NW, NS, 6> : attn_bwd_q_v3_kernel<NW, NS, 0>;
        rc = launch_v3(qk, a, im, gq, 64 * NW, kQLds, s, "attn_bwd_q_v3_kernel");
        if (rc) return rc;
        auto kern = abl == 1 ? attn_bwd_kv_v3_kernel<NW, true, 1> : abl == 2 ? attn_bwd_kv_v3_kernel<NW, true, 2>
                  : abl == 3 ? attn_bwd_kv_v3_kernel<NW, true, 3> : abl == 4 ? attn_bwd_kv_v3_kernel<NW, true, 4>
                  : abl == 6 ? attn_bwd_kv_v3_kernel<NW, true, 6> : abl == 7 ? attn_bwd_kv_v3_kernel<NW, true, 7>
                                                                                 : attn_bwd_kv_v3_kernel<NW, true, 0>;
        rc = launch_v3(kern, a, im, gk, 64 * NW, kKvLds, s, "attn_bwd_kv_v3_kernel");
#else
        rc = launch_v3(attn_bwd_q_v3_kernel<NW, NS>, a, im, gq, 64 * NW, kQLds, s, "attn_bwd_q_v3_kernel");
        if (rc) return rc;
        rc = launch_v3(attn_bwd_kv_v3_kernel<NW, true>, a, im, gk, 64 * NW, kKvLds, s, "attn_bwd_kv_v3_kernel");
#endif
    } else {
        rc = launch_v3(attn_bwd_kv_v3_kernel<NW, false>, a, im, gk, 64 * NW, kKvLds, s, "attn_bwd_kv_v3_kernel<rel only>");
    }
    if (rc) return rc;
    return rel_row_update(a.rel_v, a.rel_part, a.rel_out, a.B, a.H, a.Nk, 1.0f / a.H, s);     // one partial row per head
}

bool al16(const void* p, const Strides& s, int elem) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && (s.sh * elem) % 16 == 0 && (s.sn * elem) % 16 == 0;
}

}  // namespace

void attn_bf16_v3_enable(int mode) { g_attn_bf16_v3 = mode; }

// bytes of the prep images (independent of the batch): appended to the row-relevancy workspace
size_t attn_bwd_bf16_v3_prep_bytes(int H, int N) {
    const size_t Np = (static_cast<size_t>(N) + kT - 1) / kT * kT;
    return sizeof(bf16_t) * (4 * H * Np * kD + 2 * H * Np * Np) + 256;       // Vb, KbT, QbT, Qb + Pq, PT (or PT32: the same size)
}

// returns 1 if the third-generation kernels were launched (rc in *rc_out), 0 if the call is not eligible (see the header
// comment): shared forward, bf16 slab + bf16 gradient stream, row-relevancy mode, head_dim 64, q * scale first.
int attn_bwd_bf16_v3_try(const AttnBwdArgs& a, void* prep, size_t prep_bytes, hipStream_t s, int* rc_out) {
    if (!g_attn_bf16_v3 || !a.mma_bf16 || !a.io_bf16 || !a.rel_v || !a.o || a.dprobs) return 0;
    if (a.D != kD || a.slab_dt != MMX_BF16 || a.scale_mode != MMX_SCALE_Q_FIRST || a.Nq != a.Nk) return 0;
    if (a.probs_sb != 0 || a.vs.sb != 0 || a.oos.sb != 0 || (a.need_dqkv && (a.qs.sb != 0 || a.ks.sb != 0))) return 0;
    if (!prep || prep_bytes < attn_bwd_bf16_v3_prep_bytes(a.H, a.Nk)) return 0;
    if (!al16(a.v, a.vs, 4) || !al16(a.o, a.oos, 4) || !al16(a.dout, a.os, 2) || (a.os.sb * 2) % 16) return 0;
    if (a.need_dqkv) {
        if (!al16(a.q, a.qs, 4) || !al16(a.k, a.ks, 4)) return 0;
        if (!al16(a.dq, a.dqs, 2) || (a.dqs.sb * 2) % 8 || reinterpret_cast<uintptr_t>(a.dk) % 4 || reinterpret_cast<uintptr_t>(a.dv) % 4 ||
            (a.dks.sn * 2) % 4 || (a.dvs.sn * 2) % 4 || (a.dks.sh * 2) % 4 || (a.dvs.sh * 2) % 4 || (a.dks.sb * 2) % 4 ||
            (a.dvs.sb * 2) % 4)
            return 0;
    }
    const int Np = (a.Nk + kT - 1) / kT * kT;
    bf16_t* base = reinterpret_cast<bf16_t*>((reinterpret_cast<uintptr_t>(prep) + 255) / 256 * 256);
    bf16_t* Vb = base;
    bf16_t* KbT = Vb + static_cast<size_t>(a.H) * Np * kD;
    bf16_t* QbT = KbT + static_cast<size_t>(a.H) * Np * kD;
    bf16_t* Qb = QbT + static_cast<size_t>(a.H) * Np * kD;
    bf16_t* Pq = Qb + static_cast<size_t>(a.H) * Np * kD;
    bf16_t* PT = Pq + static_cast<size_t>(a.H) * Np * Np;
    const int wide = g_attn_bf16_v3 >= 3 ? 1 : 0;
    prep_qkv_kernel<<<dim3(Np / kT, a.H, a.need_dqkv ? 4 : 1), 256, 0, s>>>(a, Vb, KbT, QbT, Qb, Np);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *rc_out = hip_fail(e, "prep_qkv_kernel"); return 1; }
    prep_p_kernel<<<dim3(Np / kT, Np / kT, a.H), 256, 0, s>>>(reinterpret_cast<const bf16_t*>(a.probs), Pq, PT, a.Nk, Np, wide);
    e = hipGetLastError();
    if (e != hipSuccess) { *rc_out = hip_fail(e, "prep_p_kernel"); return 1; }
    const V3Images im{Vb, KbT, QbT, Pq, PT, Np, Qb, PT};
    *rc_out = run_v3<8>(a, im, s);
    return 1;
}

}  // namespace mmx
