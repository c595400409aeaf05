// Relevancy-propagation kernels for gfx950 (MI355X): rule 5 head reduction, the fused all-layer
// self-attention chain (rules 5+6), exact-fp32 MFMA batched matmul, eq. 8-9 normalisation, rollout prep.
//
// Reference sites replaced by each kernel are cited in include/mmx_relevancy.h; DESIGN.md has the
// layouts, the algorithmic-byte model and the roofline each kernel is bound by.
#include "mmx_common.h"

#include <type_traits>

namespace mmx {

// =====================================================================================================
// K_avg_heads: A_bar[b] = (1/H) sum_h clamp(G[b,h] * A[b,h], 0).   Pure HBM stream: A and G read once.
// grid = (ceil(NN/4 / 256), B), block 256; thread owns 4 consecutive positions and walks the heads in
// order (deterministic, same summation order as a sequential mean over dim h).
// =====================================================================================================
// R16: the reference's half-precision chain (notebook cell 6:20,43 with an fp16 model): the product grad * attn and the head mean
// are each rounded to fp16 (what torch does for fp16 tensors: fp32 arithmetic inside an op, one rounding of its result).
__device__ __forceinline__ float round_f16(float x) { return static_cast<float>(static_cast<_Float16>(x)); }
__device__ __forceinline__ f32x4 round_f16(f32x4 v) { return f32x4{round_f16(v[0]), round_f16(v[1]), round_f16(v[2]), round_f16(v[3])}; }

template <int DT, bool R16 = false>
__global__ __launch_bounds__(256) void avg_heads_kernel(const void* __restrict__ attn,
                                                        const void* __restrict__ grad,
                                                        float* __restrict__ out, int H, int64_t NN,
                                                        int64_t attn_bstride) {
    const int b = blockIdx.y;
    const int64_t p = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (p >= NN) return;
    const int64_t base = static_cast<int64_t>(b) * H * NN + p;
    const int64_t base_a = static_cast<int64_t>(b) * attn_bstride + p;   // attn_bstride = 0: one forward shared by the batch
    const float fH = static_cast<float>(H);
    // fast path: one aligned load per (head, array).  16-bit slabs of odd N^2 start every second head on a 2-byte
    // boundary; load4_stream fetches the three aligned dwords around the chunk instead of four 2-byte loads (it
    // over-reads two elements, hence p + 5 < NN; the last chunk of a slab takes the element-wise path)
    if (p + 5 < NN) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const f32x4 a = load4_stream<DT>(attn, base_a + h * NN);
            const f32x4 g = load4_stream<DT>(grad, base + h * NN);
            const f32x4 x = R16 ? round_f16(g * a) : g * a;
            s[0] += relu_nan(x[0]); s[1] += relu_nan(x[1]);
            s[2] += relu_nan(x[2]); s[3] += relu_nan(x[3]);
        }
        float* o = out + static_cast<int64_t>(b) * NN + p;
        const f32x4 m = R16 ? round_f16(s / fH) : s / fH;
        o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[3];
    } else {
        for (int e = 0; p + e < NN; ++e) {
            float s = 0.f;
            for (int h = 0; h < H; ++h) {
                const float x = load1_as_f32<DT>(grad, base + h * NN + e) * load1_as_f32<DT>(attn, base_a + h * NN + e);
                s += relu_nan(R16 ? round_f16(x) : x);
            }
            out[static_cast<int64_t>(b) * NN + p + e] = R16 ? round_f16(s / fH) : s / fH;
        }
    }
}

// =====================================================================================================
// K_self_chain_fused: one workgroup (16 waves) per sample runs the WHOLE chain
//     R <- I;  for every layer l:  A_bar_l = mean_h clamp(G_l*A_l, 0);  R <- R + A_bar_l . R
// in one launch.  Roles (wave-uniform):
//   waves [NT, 16)  "stream" waves: read A_l / G_l head slabs with 16-B loads, reduce over heads in
//                    registers, write A_bar_l into one of two LDS buffers (row stride NP+4 floats).
//   waves [0, NT)   "matrix" waves: wave w owns the 16-column slab w of R in REGISTERS in the MFMA
//                    C/D layout (row = 16t + 4*(lane>>4) + r, col = 16w + (lane&15)).  Because the
//                    k-order of a dot product is free, k is visited as (t, r, lane>>4): the B operand
//                    of v_mfma_f32_16x16x4_f32 is then exactly the register R[t][r] the lane already
//                    holds -- R never moves; the A operand (A_bar) is one ds_read_b128 per 4 MFMAs.
// One s_barrier per layer: stream waves publish A_bar_l, then immediately start streaming layer l+1
// into the other buffer while the matrix waves multiply.  HBM traffic = A and G once + R out.
// NT = ceil(N/16) <= 8 (N <= 128); larger N takes the split path (avg_heads + bmm).
// =====================================================================================================
struct ChainArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    const float* R_init;
    float* R_out;
    // layer-group split (G > 1): workgroup (b, g) multiplies only its contiguous layer group; the partial products
    // P_g go to `parts` [B][G][N*N] and the last arriver of a sample (ticket on counters[b]) combines them.
    int G;
    float* parts;
    unsigned* counters;
    int debug;  // profiling only: bit0 = return before the hand-off/combine, bit2 = matrix waves skip the MFMAs
    int64_t attn_bstride;  // batch stride of the attention slabs in elements (H*N*N, or 0: one forward shared by the batch)
    int pipe;  // option "self_chain_pipe": software-pipelined stream waves (fp32 slabs)
    int nt;    // option "self_chain_nt": cache policy of the pipelined slab loads (0 default policy, 1 = nt on the read-once slabs)
};

constexpr int kChainThreads = 1024;

// THREADS / U / EQ: workgroup width, heads per load batch and work split of the head reduction.
//   EQ = false (1024 threads, U = 4): NT matrix waves + (16 - NT) stream waves; the matrix waves only reduce the chunks
//        left over after the stream waves' full passes.
//   EQ = true: every wave reduces an equal share of the chunks (the matrix waves do theirs after their MFMAs).  With
//        fewer threads per workgroup the VGPR budget per lane grows (1024 threads: 128, 768: 170, 512: 256), which
//        lets a lane keep U = 8..12 heads x 2 arrays = 16..24 16-byte loads in flight instead of 8 (verified in the
//        ISA).  Measured on MI355X (profiles/r01_chain_probe.txt): every such variant is SLOWER than the default
//        (text tower 75.7 us vs 77.8 / 84.6 / 82.3 / 90.4 us for 1024-EQ / 768-U8 / 768-U12 / 512-U8) -- a CU
//        already streams at ~8 of its ~10 B/cycle HBM ceiling, more bytes in flight per lane do not raise it and
//        fewer waves lose issue overlap.  Only the default is instantiated.
//   R16 = true: the half-precision chain of the reference's fp16 mode (round_f16 above): grad * attn, the head mean, A_bar . R and
//        R + A_bar . R are each rounded to fp16, sums run in fp32 -- R is carried as fp32 registers holding fp16 values.  One
//        workgroup per sample (the layer-group split re-associates the products), the plain (not software-pipelined) stream loop.
template <int NT, int DT, int THREADS = kChainThreads, int U = 4, bool EQ = false, bool R16 = false>
__global__ __launch_bounds__(THREADS) void self_chain_fused_kernel(const ChainArgs a) {
    constexpr int NP = NT * 16;
    constexpr int S = NP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * NP * S floats

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int G = a.G;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int N = a.N, H = a.H;
    const int per = (a.n_layers + G - 1) / G;
    const int l0 = min(a.n_layers, g * per), l1 = min(a.n_layers, l0 + per);
    const int L = l1 - l0;
    const int64_t NN = static_cast<int64_t>(N) * N;

    for (int i = tid; i < 2 * NP * S; i += THREADS) smem[i] = 0.f;  // pads must read as 0
    __syncthreads();

    const int col = wave * 16 + (lane & 15);
    const int rq = (lane >> 4) * 4;
    f32x4 Rold[NT], Rnew[NT];

    // ---- head reduction of one 4-element chunk of A_bar_l (used by the stream waves and, for the remainder pass, by
    // the otherwise idle matrix waves)
    constexpr int LT = THREADS - NT * 64;         // stream lanes
    constexpr int ML = NT * 64;                   // matrix lanes
    const float fH = static_cast<float>(H);
    const int64_t sample = static_cast<int64_t>(b) * H * NN;
    const int64_t sampleA = static_cast<int64_t>(b) * a.attn_bstride;
    const int nchunks = static_cast<int>((NN + 3) >> 2);
    // Work split: a chunk costs one batch of global round trips whatever the number of busy lanes, so the pass count
    // of the stream waves is what matters.  When the chunks left over after the stream waves' full passes fit the
    // matrix lanes (text tower: 1483 = 2 x 704 + 75), the matrix waves reduce them between their MFMAs and the stream
    // waves save a whole, almost empty, pass per layer.
    const int full = nchunks / LT;
    const bool split = !EQ && full >= 1 && nchunks - full * LT <= ML;
    const int stream_end = split ? full * LT : nchunks;
    constexpr int SSTRIDE = EQ ? THREADS : LT;    // chunk stride of a stream lane
    auto reduce_chunk = [&](int c, const void* A, const void* Gr, float* Ab) {
        const int64_t p = static_cast<int64_t>(c) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (p + 3 < NN) {
#pragma unroll U  // at 1024 threads (128 VGPRs per lane) U = 4 is the optimum: 6 / 8 spill (r01_chain_probe.txt)
            for (int h = 0; h < H; ++h) {
                const f32x4 av = load4_as_f32<DT>(A, sampleA + h * NN + p);
                const f32x4 gv = load4_as_f32<DT>(Gr, sample + h * NN + p);
                const f32x4 x = R16 ? round_f16(gv * av) : gv * av;
                s[0] += relu_nan(x[0]); s[1] += relu_nan(x[1]);
                s[2] += relu_nan(x[2]); s[3] += relu_nan(x[3]);
            }
        } else {
            for (int e = 0; p + e < NN; ++e)
                for (int h = 0; h < H; ++h) {
                    const float x = load1_as_f32<DT>(Gr, sample + h * NN + p + e) * load1_as_f32<DT>(A, sampleA + h * NN + p + e);
                    s[e] += relu_nan(R16 ? round_f16(x) : x);
                }
        }
        int row = static_cast<int>(p / N);
        int cc = static_cast<int>(p - static_cast<int64_t>(row) * N);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p + e < NN) Ab[row * S + cc] = R16 ? round_f16(s[e] / fH) : s[e] / fH;
            if (++cc == N) { cc = 0; ++row; }
        }
    };

    if (wave < NT) {
        // ------------------------------------------------------------------ matrix waves
        auto remainder_pass = [&](int l) {   // this lane's share of A_bar_l: the remainder chunks, or an equal share (EQ)
            if (l >= L) return;
            if (EQ) {
                for (int c = LT + tid; c < nchunks; c += THREADS)
                    reduce_chunk(c, a.attn[l0 + l], a.grad[l0 + l], smem + (l & 1) * NP * S);
            } else {
                const int c = stream_end + tid;
                if (split && c < nchunks) reduce_chunk(c, a.attn[l0 + l], a.grad[l0 + l], smem + (l & 1) * NP * S);
            }
        };
        remainder_pass(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                float v = 0.f;
                if (row < N && col < N)
                    v = (a.R_init && g == 0) ? a.R_init[b * NN + static_cast<int64_t>(row) * N + col]
                                             : (row == col ? 1.f : 0.f);
                Rold[t][r] = v;
            }
        for (int l = 0; l < L; ++l) {
            __syncthreads();  // A_bar_l is in buffer l&1
            if (a.debug & 4) { remainder_pass(l + 1); continue; }
            const float* Ab = smem + (l & 1) * NP * S + (lane & 15) * S + rq;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + ti * 16 * S + t * 16);
                    acc = mfma16x16x4(av[0], Rold[t][0], acc);
                    acc = mfma16x16x4(av[1], Rold[t][1], acc);
                    acc = mfma16x16x4(av[2], Rold[t][2], acc);
                    acc = mfma16x16x4(av[3], Rold[t][3], acc);
                }
                Rnew[ti] = R16 ? round_f16(Rold[ti] + round_f16(acc))   // torch.bmm rounds its result, the sum is rounded again
                               : Rold[ti] + acc;                        // R + (A_bar . R): same association as the reference
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) Rold[t] = Rnew[t];
            remainder_pass(l + 1);   // into the other buffer; published by the next barrier
        }
        float* dst = (G == 1) ? a.R_out + b * NN : a.parts + (static_cast<int64_t>(b) * G + g) * NN;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                if (row < N && col < N) {
                    if (G == 1) dst[static_cast<int64_t>(row) * N + col] = Rold[t][r];
                    else  // write-through: the hand-off below needs no L2 write-back fence
                        __hip_atomic_store(dst + static_cast<int64_t>(row) * N + col, Rold[t][r], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
            }
    } else {
        // ------------------------------------------------------------------ stream waves
        const int lt = tid - NT * 64;
        if (!R16 && DT == MMX_F32 && a.pipe && static_cast<int64_t>(stream_end) * 4 <= NN && stream_end > 0 &&
            static_cast<int64_t>(H) * NN * 4 < (1ll << 31)) {
            // Software-pipelined form (fp32 slabs, option "self_chain_pipe"): the pass is a flat sequence of load batches
            // (layer, chunk of this lane, 4 heads x 2 arrays); batch i + 1 is ISSUED before batch i is reduced, so 8 16-byte
            // loads per lane are always in flight -- also across the chunk's LDS write and across the per-layer barrier (the
            // plain loop below drains its loads at every chunk and restarts cold after every barrier).  Raw buffer loads: one
            // wave-uniform resource per (layer, array), a scalar head offset, ONE lane offset register for all 16 loads of two
            // batches, which is what lets two batches fit the 128-VGPR budget of the 1024-thread workgroup.
            // Chunk-to-lane mapping: a stream wave owns blocks of 64 * CPL consecutive chunks and a batch is 4 / CPL heads x CPL
            // chunks x 2 arrays, so that one batch reads CPL KB CONTIGUOUS per (head, array) instead of 1 KB from each of 8
            // streams (CPL = 1).  CPL is as large as still leaves every stream wave a block (text tower, 77 tokens: 1408 chunks
            // = 11 waves x 128 -> CPL = 2).  Heads are summed in ascending order whatever CPL: bit-identical results.
            auto run = [&](auto cpl_tag, auto aux_tag) {
                constexpr int CPL = decltype(cpl_tag)::value, HPB = 4 / CPL;
                // cache policy of the slab loads (buffer aux bits: 2 = nt).  The gradient slab is read exactly once per launch;
                // the probability slab as well unless the batch shares one forward (attn_bstride == 0: every sample re-reads it
                // from L2, so it keeps the default policy).
                constexpr int AUXG = decltype(aux_tag)::value, AUXA = decltype(aux_tag)::value & 1 ? 0 : decltype(aux_tag)::value;
                constexpr int NW = LT / 64;
                const int ws = wave - NT;                                   // this stream wave
                const int nk = (stream_end + LT * CPL - 1) / (LT * CPL);   // rounds (lanes past the end redo the last chunk)
                const int HB = (H + HPB - 1) / HPB;
                const int total = L * nk * HB;
                const int hstride = static_cast<int>(NN) * 4;
                auto issue = [&](int it, u32x4 (&av)[4], u32x4 (&gv)[4]) {
                    const int hb = it % HB, k = (it / HB) % nk, l = it / (HB * nk);
                    const int base = (ws + k * NW) * (64 * CPL) + lane;
                    const auto rA = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(a.attn[l0 + l]) + sampleA * 4)), 0, 0x7fffffff,
                        kRawBufferFlags);
                    const auto rG = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(a.grad[l0 + l]) + sample * 4)), 0, 0x7fffffff,
                        kRawBufferFlags);
#pragma unroll
                    for (int u = 0; u < HPB; ++u) {
                        const int hoff = min(hb * HPB + u, H - 1) * hstride;        // clamped: no conditional load
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) {
                            const unsigned voff = static_cast<unsigned>(min(base + jj * 64, stream_end - 1)) * 16u;
                            av[u * CPL + jj] = __builtin_amdgcn_raw_buffer_load_b128(rA, voff, hoff, AUXA & ~1);
                            gv[u * CPL + jj] = __builtin_amdgcn_raw_buffer_load_b128(rG, voff, hoff, AUXG & ~1);
                        }
                    }
                };
                f32x4 s[CPL];
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) s[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                auto consume = [&](int it, bool valid, const u32x4 (&av)[4], const u32x4 (&gv)[4]) {
                    const int hb = it % HB, k = (it / HB) % nk, l = it / (HB * nk);
#pragma unroll
                    for (int u = 0; u < HPB; ++u) {
                        // heads in order: the same sum as the plain loop.  Every loaded register is USED unconditionally (a
                        // head beyond H is a clamped duplicate, weighted 0): behind a branch the compiler would have to assume
                        // the skipped loads still pending and drain vmcnt before the next batch may overwrite their registers.
                        const float w = (valid && hb * HPB + u < H) ? 1.f : 0.f;
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) {
                            const f32x4 x = __builtin_bit_cast(f32x4, gv[u * CPL + jj]) * __builtin_bit_cast(f32x4, av[u * CPL + jj]);
                            s[jj][0] += relu_nan(x[0]) * w; s[jj][1] += relu_nan(x[1]) * w;
                            s[jj][2] += relu_nan(x[2]) * w; s[jj][3] += relu_nan(x[3]) * w;
                        }
                    }
                    if (valid && hb == HB - 1) {
                        float* Ab = smem + (l & 1) * NP * S;
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) {
                            const int c = (ws + k * NW) * (64 * CPL) + jj * 64 + lane;
                            if (c < stream_end) {
                                const int p = c * 4;
                                int row = p / N, cc = p - row * N;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    Ab[row * S + cc] = s[jj][e] / fH;
                                    if (++cc == N) { cc = 0; ++row; }
                                }
                            }
                            s[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        if (k == nk - 1) __syncthreads();  // publish A_bar_l (pairs with the matrix waves' barrier of layer l)
                    }
                };
                // Two register sets; every issue is UNCONDITIONAL (past the end: the last batch again, consumed with weight 0)
                // so that the number of loads in flight is the same on every path -- with a conditional issue the compiler
                // merges the two paths' wait counts and drains everything before each reduction.
                u32x4 a0[4], g0[4], a1[4], g1[4];
                if (total > 0) issue(0, a0, g0);
                for (int it = 0; it < total; it += 2) {
                    issue(min(it + 1, total - 1), a1, g1);
                    consume(it, true, a0, g0);
                    issue(min(it + 2, total - 1), a0, g0);
                    consume(min(it + 1, total - 1), it + 1 < total, a1, g1);
                }
            };
            constexpr int NWs = LT / 64;
            // aux tag: 0 default policy | 2 nt on both slabs | 3 nt on the gradient slab only (bit 0 = "probabilities are shared")
            auto run_cpl = [&](auto aux_tag) {
                if (a.pipe >= 4 && stream_end >= NWs * 256) run(std::integral_constant<int, 4>{}, aux_tag);
                else if (a.pipe >= 2 && stream_end >= NWs * 128) run(std::integral_constant<int, 2>{}, aux_tag);
                else run(std::integral_constant<int, 1>{}, aux_tag);
            };
            if (!a.nt) run_cpl(std::integral_constant<int, 0>{});
            else if (a.attn_bstride == 0) run_cpl(std::integral_constant<int, 3>{});
            else run_cpl(std::integral_constant<int, 2>{});
        } else
        for (int l = 0; l < L; ++l) {
            float* Ab = smem + (l & 1) * NP * S;
            const void* A = a.attn[l0 + l];
            const void* Gr = a.grad[l0 + l];
            for (int c = lt; c < stream_end; c += SSTRIDE) reduce_chunk(c, A, Gr, Ab);
            __syncthreads();  // publish A_bar_l (pairs with the matrix waves' barrier of layer l)
        }
    }
    if (G == 1 || (a.debug & 1)) return;

    // ------------------------------------------------------------------ hand-off + combine (G > 1)
    // R = P_{G-1} . ... . P_1 . P_0 (P_0 already includes R_init).  Mathematically the sequential chain; the
    // products are re-associated at the group boundaries (rounding-level difference, tests bound it at 1e-5).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    unsigned* ticket_lds = reinterpret_cast<unsigned*>(smem + 2 * NP * S - 4);  // inside the (zero) row padding
    if (tid == 0)
        *ticket_lds = __hip_atomic_fetch_add(a.counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *ticket_lds;
    if (ticket != static_cast<unsigned>(G - 1)) return;
    if (tid == 0) {
        *ticket_lds = 0u;  // restore the zero padding
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const float* part = a.parts + static_cast<int64_t>(b) * G * NN;
    if (wave < NT) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                Rold[t][r] = (row < N && col < N) ? part[static_cast<int64_t>(row) * N + col] : 0.f;
            }
    }
    constexpr int CE = (NP * NP + THREADS - 1) / THREADS;  // elements of a partial product per thread
    float pre[CE];
    auto prefetch = [&](int gg) {
        const float* P = part + gg * NN;
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            const int idx = tid + i * THREADS;
            pre[i] = (idx < N * N) ? P[idx] : 0.f;
        }
    };
    prefetch(1);
    for (int gg = 1; gg < G; ++gg) {
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            const int idx = tid + i * THREADS;
            if (idx < N * N) {
                const int row = idx / N, cc = idx - row * N;
                smem[row * S + cc] = pre[i];
            }
        }
        lds_barrier();
        if (gg + 1 < G) prefetch(gg + 1);  // in flight while the MFMAs run
        if (wave < NT) {
            const float* Ab = smem + (lane & 15) * S + rq;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + ti * 16 * S + t * 16);
                    acc = mfma16x16x4(av[0], Rold[t][0], acc);
                    acc = mfma16x16x4(av[1], Rold[t][1], acc);
                    acc = mfma16x16x4(av[2], Rold[t][2], acc);
                    acc = mfma16x16x4(av[3], Rold[t][3], acc);
                }
                Rnew[ti] = acc;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) Rold[t] = Rnew[t];
        }
        lds_barrier();
    }
    if (wave < NT) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                if (row < N && col < N) a.R_out[b * NN + static_cast<int64_t>(row) * N + col] = Rold[t][r];
            }
    }
}

// =====================================================================================================
// K_bmm_f32: C[b] = (Cin ? Cin[b] : 0) + op(A[b]) . B[b] on v_mfma_f32_16x16x4_f32 (exact fp32).
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, 32x32 each = 2x2 MFMA tiles), BK = 32.
// The next K-slab's global loads are issued into registers before the MFMAs of the current one (the problems here
// are small -- rule 10 at DETR size is [100 x 950] . [950 x 950] = 30 workgroups -- so the loop is latency-, not
// bandwidth-bound, and a workgroup has to cover its own load latency).  LDS tiles are k-major with row stride 80
// floats: a wave's ds_read_b32 of [k = lane>>4][m = lane&15] hits 32 distinct banks per half-wave.  Guarded scalar
// global loads: any M, N, K, any alignment.
// =====================================================================================================
constexpr int kBmmBK = 32;

// T x T output tile per 256-thread workgroup, 4 waves as 2 x 2, each (T/2) x (T/2) = (T/32)^2 MFMA tiles.
// T = 64 for large problems; T = 32 when 64 x 64 tiles would leave most CUs with at most one workgroup (nothing to
// hide the global-load latency behind): 4x the workgroups for the same work.
template <int T>
__global__ __launch_bounds__(256) void bmm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      const float* Cin, float* C, int M, int N, int K,
                                                      int trans_a, int64_t sa, int64_t sb, int64_t sc,
                                                      int nan_to_zero, int cin_is_row) {
    constexpr int W = T / 32;              // MFMA tiles per wave and dimension
    constexpr int E = kBmmBK * T / 256;    // elements of each operand per thread and slab
    constexpr int LA = T + 17, LB = T + 16;   // odd A stride: the k-fastest stores of a row-major A spread over the banks
    __shared__ float As[kBmmBK][LA];
    __shared__ float Bs[kBmmBK][LB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // 1-D grid, XCD-aware: workgroups are dealt round-robin to the 8 XCDs, so in dispatch order the tiles of ONE batch entry
    // would land on all eight L2s and every one of them would fetch that entry's A and B (measured at N = 577: FETCH_SIZE
    // 2.8x the operands, profiles/r02_chain_split_roofline.txt).  xcd_contiguous_id gives each XCD one contiguous range of
    // (batch, m tile, n tile) triples: an entry's operands live in one L2.
    const int tiles_n = (N + T - 1) / T, tiles_m = (M + T - 1) / T;
    const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int bx = wg % tiles_n, by = (wg / tiles_n) % tiles_m, bz = wg / (tiles_n * tiles_m);
    const int m0 = by * T, n0 = bx * T;
    const float* Ab = A + static_cast<int64_t>(bz) * sa;
    const float* Bb = B + static_cast<int64_t>(bz) * sb;
    f32x4 acc[W][W];
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ra[E], rb[E];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * 256;
            int m, k;
            if (trans_a) { k = idx / T; m = idx % T; } else { m = idx / kBmmBK; k = idx % kBmmBK; }
            const int gm = m0 + m, gk = k0 + k;
            const bool ok = gm < M && gk < K;
            const int64_t off = trans_a ? static_cast<int64_t>(gk) * M + gm : static_cast<int64_t>(gm) * K + gk;
            const float va = Ab[ok ? off : 0];            // unconditional (clamped) load: no per-element vmcnt(0)
            ra[e] = ok ? va : 0.f;
            const int kb = idx / T, nb = idx % T;
            const int gkb = k0 + kb, gn = n0 + nb;
            const bool okb = gkb < K && gn < N;
            const float vb = Bb[okb ? static_cast<int64_t>(gkb) * N + gn : 0];
            rb[e] = okb ? vb : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kBmmBK) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = tid + e * 256;
            if (trans_a) As[idx / T][idx % T] = ra[e]; else As[idx % kBmmBK][idx / kBmmBK] = ra[e];
            Bs[idx / T][idx % T] = rb[e];
        }
        lds_barrier();
        if (k0 + kBmmBK < K) fetch(k0 + kBmmBK);          // in flight across the MFMAs and the next (LDS-only) barrier
#pragma unroll
        for (int ks = 0; ks < kBmmBK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            float av[W], bv[W];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                av[i] = As[kk][wr * (T / 2) + 16 * i + (lane & 15)];
                bv[i] = Bs[kk][wc * (T / 2) + 16 * i + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < W; ++i)
#pragma unroll
                for (int j = 0; j < W; ++j) acc[i][j] = mfma16x16x4(av[i], bv[j], acc[i][j]);
        }
        lds_barrier();
    }
    const int64_t cbase = static_cast<int64_t>(bz) * sc;
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wr * (T / 2) + i * 16 + (lane >> 4) * 4 + r;
                const int gn = n0 + wc * (T / 2) + j * 16 + (lane & 15);
                if (gm < M && gn < N) {
                    const int64_t off = cbase + static_cast<int64_t>(gm) * N + gn;
                    float v = acc[i][j][r];
                    if (Cin) v = (cin_is_row ? Cin[gn] : Cin[off]) + v;      // cin_is_row: a bias row shared by every row
                    if (nan_to_zero && v != v) v = 0.f;
                    C[off] = v;
                }
            }
}

static int g_bmm_tiles = 1;   // option "bmm_tiles": 1 (default) large plain products on bmm_f32_tiles.hip | 0: always the general kernel below

static void launch_bmm(const float* A, const float* B, const float* Cin, float* C, int batch, int M, int N, int K,
                       int trans_a, int64_t sa, int64_t sb, int64_t sc, int nan_to_zero, hipStream_t s, int cin_is_row = 0) {
    if (g_bmm_tiles && bmm_f32_tiles_try(A, B, Cin, C, batch, M, N, K, trans_a, sa, sb, sc, nan_to_zero, cin_is_row, s)) return;
    const int64_t wgs64 = static_cast<int64_t>((N + 63) / 64) * ((M + 63) / 64) * batch;
    // (a 128 x 128 tiling was measured SLOWER at the long-sequence chain shapes -- profiles/r02_bmm_probe.txt: 384 vs 281 us at
    // [32 x 577 x 577]^2 -- and removed in round 5)
    if (wgs64 >= 1024) {
        bmm_f32_kernel<64><<<dim3(static_cast<unsigned>(wgs64)), 256, 0, s>>>(A, B, Cin, C, M, N, K, trans_a, sa,
                                                                                    sb, sc, nan_to_zero, cin_is_row);
    } else {
        bmm_f32_kernel<32><<<dim3(static_cast<unsigned>(static_cast<int64_t>((N + 31) / 32) * ((M + 31) / 32) * batch)), 256, 0, s>>>(A, B, Cin, C, M, N, K, trans_a, sa,
                                                                                    sb, sc, nan_to_zero, cin_is_row);
    }
}

// =====================================================================================================
// Row kernels: one wave per row.
//   mode 0  handle_residual (eq. 8-9): out = (R - I)/rowsum(R - I) + I ; also min_i(R[i,i] - 1)
//   mode 1  rollout prep, normalised : out = (A + I)/rowsum(A + I)
//   mode 2  rollout prep, plain      : out = A + I
// =====================================================================================================
__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void fill_scalar_kernel(float* p, float v) { *p = v; }

__global__ __launch_bounds__(256) void row_normalise_kernel(const float* __restrict__ R, float* __restrict__ out,
                                                            int rows_total, int N, int mode, float* diag_min) {
    const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);  // global row over batch*N
    const int lane = threadIdx.x & 63;
    if (row_g >= rows_total) return;
    const int i = row_g % N;
    const float* r = R + static_cast<int64_t>(row_g) * N;
    float* o = out + static_cast<int64_t>(row_g) * N;
    const float dsub = (mode == 0) ? -1.f : 1.f;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += r[j] + (j == i ? dsub : 0.f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (mode == 0 && diag_min && lane == 0) atomic_min_float(diag_min, r[i] - 1.f);
    for (int j = lane; j < N; j += 64) {
        float v = r[j] + (j == i ? dsub : 0.f);
        if (mode != 2) v = v / s;
        if (mode == 0 && j == i) v += 1.f;
        o[j] = v;
    }
}

}  // namespace mmx

// =====================================================================================================
// C-ABI entry points (see include/mmx_relevancy.h)
// =====================================================================================================
using namespace mmx;

static int avg_heads_launch(const void* attn_dev, const void* grad_dev, void* out_dev, int B, int H, int Nq, int Nk,
                            int dtype, int64_t attn_bstride, void* stream) {
    MMX_CHECK_ARG(attn_dev && grad_dev && out_dev, "mmx_avg_heads: null pointer");
    MMX_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "mmx_avg_heads: non-positive size B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    const int64_t NN = static_cast<int64_t>(Nq) * Nk;
    dim3 grid(static_cast<unsigned>((((NN + 3) >> 2) + 255) / 256), B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* out = static_cast<float*>(out_dev);
    switch (dtype) {
        case MMX_F32: avg_heads_kernel<MMX_F32><<<grid, 256, 0, s>>>(attn_dev, grad_dev, out, H, NN, attn_bstride); break;
        case MMX_F16: avg_heads_kernel<MMX_F16><<<grid, 256, 0, s>>>(attn_dev, grad_dev, out, H, NN, attn_bstride); break;
        case MMX_BF16: avg_heads_kernel<MMX_BF16><<<grid, 256, 0, s>>>(attn_dev, grad_dev, out, H, NN, attn_bstride); break;
        default: set_error("mmx_avg_heads: unsupported dtype %d", dtype); return MMX_EINVAL;
    }
    MMX_LAUNCH_CHECK("avg_heads_kernel");
    return MMX_OK;
}

extern "C" int mmx_avg_heads(const void* attn_dev, const void* grad_dev, void* out_dev, int B, int H, int Nq,
                             int Nk, int dtype, void* stream) {
    return avg_heads_launch(attn_dev, grad_dev, out_dev, B, H, Nq, Nk, dtype, static_cast<int64_t>(H) * Nq * Nk, stream);
}

extern "C" int mmx_avg_heads_ex(const void* attn_dev, const void* grad_dev, void* out_dev, int B, int H, int Nq,
                                int Nk, int dtype, int64_t attn_batch_stride, void* stream) {
    const int64_t full = static_cast<int64_t>(H) * Nq * Nk;
    if (attn_batch_stride < 0) attn_batch_stride = full;
    MMX_CHECK_ARG(attn_batch_stride == 0 || attn_batch_stride == full,
                  "mmx_avg_heads_ex: attn_batch_stride must be 0 (shared forward) or H*Nq*Nk");
    return avg_heads_launch(attn_dev, grad_dev, out_dev, B, H, Nq, Nk, dtype, attn_batch_stride, stream);
}

extern "C" int mmx_bmm_f32(const void* A_dev, const void* B_dev, const void* Cin_dev, void* C_dev, int batch, int M,
                           int N, int K, int trans_a, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                           int nan_to_zero, void* stream) {
    MMX_CHECK_ARG(A_dev && B_dev && C_dev, "mmx_bmm_f32: null pointer");
    MMX_CHECK_ARG(batch > 0 && M > 0 && N > 0 && K > 0, "mmx_bmm_f32: non-positive size");
    MMX_CHECK_ARG(batch <= 65535, "mmx_bmm_f32: batch %d > 65535", batch);
    launch_bmm(static_cast<const float*>(A_dev), static_cast<const float*>(B_dev), static_cast<const float*>(Cin_dev),
               static_cast<float*>(C_dev), batch, M, N, K, trans_a, stride_a, stride_b, stride_c, nan_to_zero,
               static_cast<hipStream_t>(stream));
    MMX_LAUNCH_CHECK("bmm_f32_kernel");
    return MMX_OK;
}

extern "C" int mmx_linear_f32(const void* x_dev, const void* wt_dev, const void* bias_dev, void* out_dev, int M, int N, int K,
                              void* stream) {
    MMX_CHECK_ARG(x_dev && wt_dev && out_dev, "mmx_linear_f32: null pointer");
    MMX_CHECK_ARG(M > 0 && N > 0 && K > 0, "mmx_linear_f32: non-positive size");
    launch_bmm(static_cast<const float*>(x_dev), static_cast<const float*>(wt_dev), static_cast<const float*>(bias_dev),
               static_cast<float*>(out_dev), 1, M, N, K, 0, 0, 0, 0, 0, static_cast<hipStream_t>(stream), 1);
    MMX_LAUNCH_CHECK("bmm_f32_kernel (linear)");
    return MMX_OK;
}

static int launch_rows(const void* R, void* out, int batch, int N, int mode, float* diag_min, hipStream_t s) {
    const int rows = batch * N;
    if (diag_min) fill_scalar_kernel<<<1, 1, 0, s>>>(diag_min, __builtin_inff());
    row_normalise_kernel<<<(rows + 3) / 4, 256, 0, s>>>(static_cast<const float*>(R), static_cast<float*>(out), rows,
                                                        N, mode, diag_min);
    MMX_LAUNCH_CHECK("row_normalise_kernel");
    return MMX_OK;
}

extern "C" int mmx_handle_residual(const void* R_dev, void* out_dev, int batch, int N, void* diag_min_dev,
                                   void* stream) {
    MMX_CHECK_ARG(R_dev && out_dev, "mmx_handle_residual: null pointer");
    MMX_CHECK_ARG(batch > 0 && N > 0, "mmx_handle_residual: non-positive size");
    return launch_rows(R_dev, out_dev, batch, N, 0, static_cast<float*>(diag_min_dev),
                       static_cast<hipStream_t>(stream));
}

// ----------------------------------------------------------------------------------------- chain on vectors
// One ROW / COLUMN of the chain instead of the matrix (DETR rows-only rules, detr_explainability._rows_only_rules):
//   matvec:  out[b] = base[b] + A[b] . y[b]       (R 1 carried bottom-up: the row sums eq. 8-9 divides by)
//   vecmat:  out[b] = base[b] + x[b] . A[b]       (a row of R carried top-down)
// `base` is separate from the multiplied vector so that a caller can carry the DEVIATION from the start vector
// (e <- e + A (1 + e) for R 1 - 1, d <- d + (v + d) A for v R - v): subtracting the start vector at the end instead would
// cancel 2-3 digits, because R - I is small against I.
// A: [B, N, N] fp32 (the head-averaged map of a layer).  Both read A once: N^2 bytes instead of the 2 N^3 flops of
// R <- R + A.R.  matvec: one wave per row.  vecmat: a workgroup reduces a 32-row chunk for all columns into a partial
// row, a second pass adds the chunks in a fixed order (deterministic; no atomics).
namespace mmx {

__global__ __launch_bounds__(256) void chain_matvec_kernel(const float* __restrict__ A, const float* __restrict__ y,
                                                           const float* __restrict__ base, float* __restrict__ out,
                                                           int rows_total, int N) {
    const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row_g >= rows_total) return;
    const int b = row_g / N;
    const float* a = A + static_cast<int64_t>(row_g) * N;
    const float* yb = y + static_cast<int64_t>(b) * N;
    float s = 0.f;
    const int n4 = N >> 2;
    for (int j = lane; j < n4; j += 64) {
        const f32x4 av = ldg4_u(a + 4 * j), yv = ldg4_u(yb + 4 * j);
        s += av[0] * yv[0] + av[1] * yv[1] + av[2] * yv[2] + av[3] * yv[3];
    }
    for (int j = 4 * n4 + lane; j < N; j += 64) s += a[j] * yb[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row_g] = base[row_g] + s;
}

constexpr int kVecmatRows = 32;

__global__ __launch_bounds__(256) void chain_vecmat_partial_kernel(const float* __restrict__ A, const float* __restrict__ x,
                                                                   float* __restrict__ part, int N, int chunks) {
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int r0 = chunk * kVecmatRows, r1 = min(N, r0 + kVecmatRows);
    const float* xb = x + static_cast<int64_t>(b) * N;
    const float* Ab = A + static_cast<int64_t>(b) * N * N;
    float* pb = part + (static_cast<int64_t>(b) * chunks + chunk) * N;
    for (int c = threadIdx.x * 4; c < N; c += 256 * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < N) {
            for (int r = r0; r < r1; ++r) acc += ldg4_u(Ab + static_cast<int64_t>(r) * N + c) * xb[r];
            pb[c] = acc[0]; pb[c + 1] = acc[1]; pb[c + 2] = acc[2]; pb[c + 3] = acc[3];
        } else {
            for (int e = 0; c + e < N; ++e) {
                float sacc = 0.f;
                for (int r = r0; r < r1; ++r) sacc += Ab[static_cast<int64_t>(r) * N + c + e] * xb[r];
                pb[c + e] = sacc;
            }
        }
    }
}

__global__ __launch_bounds__(256) void chain_vecmat_reduce_kernel(const float* __restrict__ part, const float* __restrict__ base,
                                                                  float* __restrict__ out, int N, int chunks) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= N) return;
    const float* pb = part + static_cast<int64_t>(b) * chunks * N + c;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += pb[static_cast<int64_t>(k) * N];
    out[static_cast<int64_t>(b) * N + c] = base[static_cast<int64_t>(b) * N + c] + s;
}

// One row of the chain through one layer WITHOUT materialising A_bar:  out[b] = base[b] + x[b] . mean_h clamp(G[b, h] * A[b, h], 0).
// A workgroup owns 4 rows of the slabs (wave w: row 4 chunk + w; lane: 4 consecutive columns, all heads' 2 x 16-byte loads in flight
// at once), scales its head mean by x[b, row] and the four rows meet in LDS: one [N] partial per workgroup, summed over the N / 4
// workgroups by a second small launch (in workgroup order: deterministic).  Replaces avg_heads + the two chain_vecmat launches of
// the row-vector chain (ViT-B/16 at one image: 10.7 + 17.4 + 4.9 us per layer, latency-bound at 7 workgroups).
constexpr int kAhvRows = 4;

template <int DT>
__global__ __launch_bounds__(256) void avg_heads_vecmat_partial_kernel(const void* __restrict__ attn, const void* __restrict__ grad,
                                                                       const float* __restrict__ x, float* __restrict__ part, int H,
                                                                       int N, int64_t attn_bstride, int chunks) {
    __shared__ f32x4 red[kAhvRows][64];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int rl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = chunk * kAhvRows + rl;
    const int64_t NN = static_cast<int64_t>(N) * N;
    const int n4 = (N + 3) >> 2;
    const float fH = static_cast<float>(H);
    const float xr = row < N ? x[static_cast<int64_t>(b) * N + row] : 0.f;
    for (int cg0 = 0; cg0 < n4; cg0 += 64) {
        const int cg = cg0 + lane, c = cg * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < N && cg < n4) {
            const int64_t p = static_cast<int64_t>(row) * N + c;
            const int64_t base_g = static_cast<int64_t>(b) * H * NN + p, base_a = static_cast<int64_t>(b) * attn_bstride + p;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (c + 3 < N && p + 5 < NN) {          // a whole chunk (p + 5: the 16-bit loads over-read two elements)
#pragma unroll 4
                for (int h = 0; h < H; ++h) {
                    const f32x4 a = load4_stream<DT>(attn, base_a + h * NN);
                    const f32x4 g = load4_stream<DT>(grad, base_g + h * NN);
                    const f32x4 t = g * a;
                    s[0] += relu_nan(t[0]); s[1] += relu_nan(t[1]); s[2] += relu_nan(t[2]); s[3] += relu_nan(t[3]);
                }
            } else {
                for (int e = 0; e < 4 && c + e < N; ++e)
                    for (int h = 0; h < H; ++h)
                        s[e] += relu_nan(load1_as_f32<DT>(grad, base_g + h * NN + e) * load1_as_f32<DT>(attn, base_a + h * NN + e));
            }
            v = f32x4{s[0] / fH * xr, s[1] / fH * xr, s[2] / fH * xr, s[3] / fH * xr};
        }
        red[rl][lane] = v;
        __syncthreads();
        if (rl == 0 && cg < n4) {
            const f32x4 t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
            *reinterpret_cast<f32x4*>(part + (static_cast<int64_t>(b) * chunks + chunk) * (n4 * 4) + c) = t;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void avg_heads_vecmat_reduce_kernel(const float* __restrict__ part, const float* __restrict__ base,
                                                                      float* __restrict__ out, int N, int ld, int chunks) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= N) return;
    const float* pb = part + static_cast<int64_t>(b) * chunks * ld + c;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += pb[static_cast<int64_t>(k) * ld];
    out[static_cast<int64_t>(b) * N + c] = base[static_cast<int64_t>(b) * N + c] + s;
}

}  // namespace mmx

extern "C" size_t mmx_avg_heads_vecmat_workspace_bytes(int B, int N) {
    const size_t chunks = (static_cast<size_t>(N) + mmx::kAhvRows - 1) / mmx::kAhvRows;
    return sizeof(float) * static_cast<size_t>(B) * chunks * (((static_cast<size_t>(N) + 3) >> 2) * 4);
}

extern "C" int mmx_avg_heads_vecmat(const void* attn_dev, const void* grad_dev, const void* x_dev, const void* base_dev, void* out_dev,
                                    int B, int H, int N, int dtype, int64_t attn_batch_stride, void* workspace_dev,
                                    size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(attn_dev && grad_dev && x_dev && base_dev && out_dev && B > 0 && H > 0 && N > 0 && B <= 65535,
                  "mmx_avg_heads_vecmat: bad argument");
    const int64_t full = static_cast<int64_t>(H) * N * N;
    if (attn_batch_stride < 0) attn_batch_stride = full;
    MMX_CHECK_ARG(attn_batch_stride == 0 || attn_batch_stride == full,
                  "mmx_avg_heads_vecmat: attn_batch_stride must be 0 (shared forward) or H*N*N");
    MMX_CHECK_ARG(x_dev != out_dev, "mmx_avg_heads_vecmat: out may not alias x");
    if (!workspace_dev || workspace_bytes < mmx_avg_heads_vecmat_workspace_bytes(B, N) ||
        (reinterpret_cast<uintptr_t>(workspace_dev) & 15u)) {
        mmx::set_error("mmx_avg_heads_vecmat: workspace %zu < %zu (or not 16-byte aligned)", workspace_bytes,
                       mmx_avg_heads_vecmat_workspace_bytes(B, N));
        return MMX_EWORKSPACE;
    }
    const int chunks = (N + mmx::kAhvRows - 1) / mmx::kAhvRows, ld = ((N + 3) >> 2) * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace_dev);
    const float* x = static_cast<const float*>(x_dev);
    dim3 grid(chunks, B);
    switch (dtype) {
        case MMX_F32: mmx::avg_heads_vecmat_partial_kernel<MMX_F32><<<grid, 256, 0, s>>>(attn_dev, grad_dev, x, part, H, N, attn_batch_stride, chunks); break;
        case MMX_F16: mmx::avg_heads_vecmat_partial_kernel<MMX_F16><<<grid, 256, 0, s>>>(attn_dev, grad_dev, x, part, H, N, attn_batch_stride, chunks); break;
        case MMX_BF16: mmx::avg_heads_vecmat_partial_kernel<MMX_BF16><<<grid, 256, 0, s>>>(attn_dev, grad_dev, x, part, H, N, attn_batch_stride, chunks); break;
        default: mmx::set_error("mmx_avg_heads_vecmat: unsupported dtype %d", dtype); return MMX_EINVAL;
    }
    MMX_LAUNCH_CHECK("avg_heads_vecmat_partial_kernel");
    mmx::avg_heads_vecmat_reduce_kernel<<<dim3((N + 255) / 256, B), 256, 0, s>>>(part, static_cast<const float*>(base_dev),
                                                                                 static_cast<float*>(out_dev), N, ld, chunks);
    MMX_LAUNCH_CHECK("avg_heads_vecmat_reduce_kernel");
    return MMX_OK;
}

extern "C" int mmx_chain_matvec(const void* A_dev, const void* y_dev, const void* base_dev, void* out_dev, int B, int N,
                                void* stream) {
    MMX_CHECK_ARG(A_dev && y_dev && base_dev && out_dev && B > 0 && N > 0 && y_dev != out_dev, "mmx_chain_matvec: bad argument");
    const int rows = B * N;
    mmx::chain_matvec_kernel<<<(rows + 3) / 4, 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(A_dev), static_cast<const float*>(y_dev), static_cast<const float*>(base_dev),
        static_cast<float*>(out_dev), rows, N);
    MMX_LAUNCH_CHECK("chain_matvec_kernel");
    return MMX_OK;
}

extern "C" size_t mmx_chain_vecmat_workspace_bytes(int B, int N) {
    const size_t chunks = (static_cast<size_t>(N) + mmx::kVecmatRows - 1) / mmx::kVecmatRows;
    return sizeof(float) * static_cast<size_t>(B) * chunks * N;
}

extern "C" int mmx_chain_vecmat(const void* A_dev, const void* x_dev, const void* base_dev, void* out_dev, int B, int N,
                                void* workspace_dev, size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(A_dev && x_dev && base_dev && out_dev && B > 0 && N > 0, "mmx_chain_vecmat: bad argument");
    if (!workspace_dev || workspace_bytes < mmx_chain_vecmat_workspace_bytes(B, N)) {
        mmx::set_error("mmx_chain_vecmat: workspace %zu < %zu", workspace_bytes, mmx_chain_vecmat_workspace_bytes(B, N));
        return MMX_EWORKSPACE;
    }
    const int chunks = (N + mmx::kVecmatRows - 1) / mmx::kVecmatRows;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace_dev);
    mmx::chain_vecmat_partial_kernel<<<dim3(chunks, B), 256, 0, s>>>(static_cast<const float*>(A_dev),
                                                                    static_cast<const float*>(x_dev), part, N, chunks);
    MMX_LAUNCH_CHECK("chain_vecmat_partial_kernel");
    mmx::chain_vecmat_reduce_kernel<<<dim3((N + 255) / 256, B), 256, 0, s>>>(part, static_cast<const float*>(base_dev),
                                                                             static_cast<float*>(out_dev), N, chunks);
    MMX_LAUNCH_CHECK("chain_vecmat_reduce_kernel");
    return MMX_OK;
}

// ----------------------------------------------------------------------------------------- rules 10/11
__global__ void copy_scrub_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int nan_to_zero) {
    const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = in[i];
        out[i] = (nan_to_zero && v != v) ? 0.0f : v;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

extern "C" size_t mmx_mm_rules_workspace_bytes(int Ns, int Nq) {
    // Rn_ss [Ns,Ns] + Rn_qq [Nq,Nq] + tmp [Ns,Nq]
    return align256(sizeof(float) * Ns * Ns) + align256(sizeof(float) * Nq * Nq) + align256(sizeof(float) * Ns * Nq);
}

extern "C" int mmx_mm_attention_rules(const void* R_ss_dev, const void* R_qq_dev, const void* R_qs_dev,
                                      const void* cam_sq_dev, void* R_sq_add_dev, void* R_ss_add_dev, int Ns, int Nq,
                                      unsigned flags, void* diag_min_dev, void* workspace_dev,
                                      size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(R_ss_dev && R_qq_dev && cam_sq_dev && R_sq_add_dev, "mmx_mm_attention_rules: null pointer");
    MMX_CHECK_ARG(Ns > 0 && Nq > 0, "mmx_mm_attention_rules: non-positive size");
    MMX_CHECK_ARG((R_qs_dev == nullptr) == (R_ss_add_dev == nullptr),
                  "mmx_mm_attention_rules: R_qs and R_ss_add must be given together (rule 11)");
    if (workspace_bytes < mmx_mm_rules_workspace_bytes(Ns, Nq) || !workspace_dev) {
        set_error("mmx_mm_attention_rules: workspace %zu < %zu", workspace_bytes, mmx_mm_rules_workspace_bytes(Ns, Nq));
        return MMX_EWORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace_dev);
    float* Rn_ss = reinterpret_cast<float*>(ws);
    float* Rn_qq = reinterpret_cast<float*>(ws + align256(sizeof(float) * Ns * Ns));
    float* tmp = reinterpret_cast<float*>(ws + align256(sizeof(float) * Ns * Ns) + align256(sizeof(float) * Nq * Nq));
    const int nan0 = (flags & MMX_MM_NAN_TO_ZERO) ? 1 : 0;
    int rc;
    const float* ss = static_cast<const float*>(R_ss_dev);
    const float* qq = static_cast<const float*>(R_qq_dev);
    if (flags & MMX_MM_NORMALIZE) {
        // The reference normalises (and asserts diag >= 0) BEFORE it looks at apply_self_in_rule_10
        // (DETR/modules/ExplanationGenerator.py:36-38, lxmert/.../ExplanationGenerator.py:35-37), so the check word is
        // produced in both branches.  Both residual normalisations share diag_min (min over both diagonals).
        float* dm = static_cast<float*>(diag_min_dev);
        if (dm) fill_scalar_kernel<<<1, 1, 0, s>>>(dm, __builtin_inff());
        if ((flags & MMX_MM_SELF_IN_RULE10) || dm) {
            row_normalise_kernel<<<(Ns + 3) / 4, 256, 0, s>>>(ss, Rn_ss, Ns, Ns, 0, dm);
            row_normalise_kernel<<<(Nq + 3) / 4, 256, 0, s>>>(qq, Rn_qq, Nq, Nq, 0, dm);
            MMX_LAUNCH_CHECK("row_normalise_kernel");
        }
        ss = Rn_ss;
        qq = Rn_qq;
    }
    if (flags & MMX_MM_SELF_IN_RULE10) {
        // tmp = cam_sq . Rn_qq ; R_sq_add = Rn_ss^T . tmp
        rc = mmx_bmm_f32(cam_sq_dev, qq, nullptr, tmp, 1, Ns, Nq, Nq, 0, 0, 0, 0, 0, stream);
        if (rc) return rc;
        rc = mmx_bmm_f32(ss, tmp, nullptr, R_sq_add_dev, 1, Ns, Nq, Ns, 1, 0, 0, 0, nan0, stream);
        if (rc) return rc;
    } else {
        // the reference returns cam_sq itself; the DETR flavour scrubs its NaNs too (:40-42)
        const long n = static_cast<long>(Ns) * Nq;
        copy_scrub_kernel<<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(
            static_cast<const float*>(cam_sq_dev), static_cast<float*>(R_sq_add_dev), n, nan0);
        MMX_LAUNCH_CHECK("copy_scrub_kernel");
    }
    if (R_qs_dev) {
        rc = mmx_bmm_f32(cam_sq_dev, R_qs_dev, nullptr, R_ss_add_dev, 1, Ns, Ns, Nq, 0, 0, 0, 0, 0, stream);
        if (rc) return rc;
    }
    return MMX_OK;
}

// ----------------------------------------------------------------------------------------- rollout
extern "C" size_t mmx_rollout_workspace_bytes(int B, int N) {
    return 2 * align256(sizeof(float) * static_cast<size_t>(B) * N * N);
}

extern "C" int mmx_rollout_chain(const void* const* layers, int n_layers, int B, int N, int normalize, void* out_dev,
                                 void* workspace_dev, size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(layers && out_dev, "mmx_rollout_chain: null pointer");
    MMX_CHECK_ARG(n_layers > 0 && B > 0 && N > 0, "mmx_rollout_chain: non-positive size");
    if (workspace_bytes < mmx_rollout_workspace_bytes(B, N) || !workspace_dev) {
        set_error("mmx_rollout_chain: workspace %zu < %zu", workspace_bytes, mmx_rollout_workspace_bytes(B, N));
        return MMX_EWORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t mat = align256(sizeof(float) * static_cast<size_t>(B) * N * N);
    float* aug = static_cast<float*>(workspace_dev);
    float* pong = reinterpret_cast<float*>(static_cast<char*>(workspace_dev) + mat);
    float* out = static_cast<float*>(out_dev);
    const int mode = normalize ? 1 : 2;
    const int64_t nn = static_cast<int64_t>(N) * N;
    // joint = aug_0; joint = aug_i . joint.  Ping-pong so that the last product lands in `out`.
    float* cur = ((n_layers - 1) % 2 == 0) ? out : pong;
    int rc = launch_rows(layers[0], cur, B, N, mode, nullptr, s);
    if (rc) return rc;
    for (int i = 1; i < n_layers; ++i) {
        float* nxt = (cur == out) ? pong : out;
        rc = launch_rows(layers[i], aug, B, N, mode, nullptr, s);
        if (rc) return rc;
        rc = mmx_bmm_f32(aug, cur, nullptr, nxt, B, N, N, N, 0, nn, nn, nn, 0, stream);
        if (rc) return rc;
        cur = nxt;
    }
    return MMX_OK;
}

// ----------------------------------------------------------------------------------------- self chain
static int nt_for(int N) { return (N + 15) / 16; }
// option "debug_flags" (profiling only), ONE meaning per bit whatever kernel the dispatcher picks:
//   1  return before the hand-off / combine (fused + groups kernels; the cols kernel has neither and ignores it)
//   4  matrix waves skip the MFMAs (all three kernels)          8  groups kernel: take the ticket, skip the combine
//   16 cols kernel: no block rotation
//   32 / 64  long-sequence layer kernel (relevancy_chain_rows.hip): no head reduction / return after the head reduction
static int g_debug_flags = 0;
static int cols_debug_flags(int g) { return ((g & 4) ? 1 : 0) | ((g & 16) ? 2 : 0); }   // -> relevancy_chain_cols.hip's own bits
static int g_chain_pipe = 4;     // option "self_chain_pipe": software-pipelined stream waves of the fused chain (fp32 slabs)
static int g_chain_nt = 1;       // option "self_chain_nt": nt cache policy on the read-once slab loads of the pipelined stream waves
                                 // (default since round 4: text tower 85.8 -> 82.0 us, image 65.5 -> 63.8 us inside the replayed step)
static int g_chain_groups = 0;  // layer groups per sample of the per-sample kernel: 0 auto, 1 = strict sequential order
static int g_chain_rows = 0;    // option "self_chain_rows": 0 (default) avg_heads_kernel + the tiled product, two launches per layer | 1: N > 128, one
                                // right-hand side: a layer is ONE launch (relevancy_chain_rows.hip: head reduction of a 16-row block into LDS +
                                // that block row of the product).  Built in round 6, parity-green, and NOT faster: 27.6 vs 24.0 ms for the 24
                                // layers of cfg 5's slab variant (profiles/r06_chain_rows_probe.txt has the phase split and why)
static int g_chain_algo = 0;  // 0 auto = 4: layer groups with barrier-free stream waves where they apply (fp32 slabs, G > 1), else this file's
                              // per-sample kernel | 1: this file's kernel everywhere | 5: relevancy_chain_cols.hip wherever it applies

extern "C" int mmx_set_option(const char* key, int value) {
    if (key && strcmp(key, "self_chain_cols_c") == 0 && value >= 0 && value <= 8) {
        chain_cols_options(value, -1);
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_cols_nb") == 0 && value >= 0 && value <= 8) {
        chain_cols_options(-1, value);
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_algo") == 0 && (value == 0 || value == 1 || value == 4 || value == 5)) {
        g_chain_algo = value;
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_pipe") == 0 && value >= 0 && value <= 4) {
        g_chain_pipe = value;            // 0 off | 1 one chunk per lane and head | 2 / 4: up to that many contiguous chunks
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_rows") == 0 && value >= 0 && value <= 1) {
        g_chain_rows = value;
        return MMX_OK;
    }
    if (key && strcmp(key, "bmm_tiles") == 0 && value >= 0 && value <= 1) {
        g_bmm_tiles = value;
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_nt") == 0 && value >= 0 && value <= 1) {
        g_chain_nt = value;
        return MMX_OK;
    }
    if (key && strcmp(key, "self_chain_groups") == 0 && value >= 0 && value <= 8) {
        g_chain_groups = value;
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_head_tile_skip") == 0) {
        if (value != 0 && value != 1) { set_error("mmx_set_option(attn_head_tile_skip): 0 or 1"); return MMX_EINVAL; }
        attn_head_tile_skip(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_head") == 0) {
        attn_head_enable(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_bf16_v3") == 0 && value >= 0 && value <= 3) {
        attn_bf16_v3_enable(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_bf16_v2") == 0) {
        attn_bf16_v2_enable(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_stream") == 0) {
        attn_stream_enable(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "attn_fwd_split") == 0) {
        attn_fwd_split_enable(value);
        return MMX_OK;
    }
    if (key && strcmp(key, "debug_flags") == 0) {
        g_debug_flags = value;
        return MMX_OK;
    }
    set_error("mmx_set_option: unknown option/value %s=%d", key ? key : "(null)", value);
    return MMX_EINVAL;
}

static int fused_groups(int n_layers, int B, int H, int N);
// One group, fp32 slabs, N >= 40: relevancy_chain_cols.hip with one workgroup per sample (the fused kernel's single-group form with
// barrier-free stream waves and a ring of A_bar images; same bits).  Option self_chain_algo = 5 takes it for every shape it supports,
// 1 never.
static bool use_cols(int n_layers, int B, int H, int N, int M, int dtype) {
    if (g_chain_algo == 1 || dtype != MMX_F32 || M != 0 || !self_chain_cols_applies(n_layers, B, H, N)) return false;
    if (g_chain_algo == 5) return true;
    return N >= 40 && fused_groups(n_layers, B, H, N) == 1;
}

static int fused_groups(int n_layers, int B, int H, int N) {
    if (n_layers < 2) return 1;
    int G = g_chain_groups;
    if (G == 0) {
        // auto: split only when a sample streams enough bytes to pay for the hand-off (>= 1 MB).  Then the FEWEST groups that put a
        // workgroup on ~70 % of the CUs: every further group adds an exact-fp32 product to the serial combine on ONE CU (~2.9 us each
        // at 77 tokens) while the stream rate is flat from ~190 workgroups on; never more workgroups than CUs (a second round of
        // workgroups costs far more than idle CUs: B = 96 / 128 at CLIP's text shape: G = 4 141 / 155 us, G = 2 96 / 120 us); at
        // most 4.  profiles/r05_chain_groups_probe.txt (B = 16 ... 128); rounds 1-4 used 4 groups up to B = 128.
        const double sample_bytes = 8.0 * n_layers * H * N * N;
        if (sample_bytes < 1e6) {
            G = 1;
        } else {
            const int cus = device_cu_count();
            G = (7 * cus + 10 * B - 1) / (10 * B);
            while (G > 1 && B * G > cus) --G;
            if (G > 4) G = 4;
        }
    }
    return G < n_layers ? G : n_layers;
}

static size_t group_counter_bytes(int B) { return align256(sizeof(unsigned) * static_cast<size_t>(B)); }

extern "C" size_t mmx_self_chain_workspace_bytes(int n_layers, int B, int H, int N, int M, int dtype) {
    (void)dtype;
    if (nt_for(N) <= 8 && M == 0) {
        if (use_cols(n_layers, B, H, N, M, dtype)) return 0;
        const int G = fused_groups(n_layers, B, H, N);
        if (G == 1) return 0;  // strict-order per-sample kernel needs no scratch
        return group_counter_bytes(B) + align256(sizeof(float) * static_cast<size_t>(B) * G * N * N);
    }
    const size_t mat = align256(sizeof(float) * static_cast<size_t>(B) * N * N);
    const size_t sq = M > 0 ? align256(sizeof(float) * static_cast<size_t>(B) * N * M) : 0;
    return 2 * mat + sq;  // A_bar + R ping-pong [+ R_sq ping-pong]
}

template <int NT>
static int launch_fused(const ChainArgs& args, int dtype, hipStream_t s, bool half_chain = false) {
    constexpr int NP = NT * 16;
    const size_t lds = sizeof(float) * 2 * NP * (NP + 4);
    void (*kern)(const ChainArgs) = nullptr;
    switch (dtype) {
        case MMX_F32: kern = half_chain ? self_chain_fused_kernel<NT, MMX_F32, kChainThreads, 4, false, true>
                                        : self_chain_fused_kernel<NT, MMX_F32>; break;
        case MMX_F16: kern = half_chain ? self_chain_fused_kernel<NT, MMX_F16, kChainThreads, 4, false, true>
                                        : self_chain_fused_kernel<NT, MMX_F16>; break;
        case MMX_BF16: if (!half_chain) { kern = self_chain_fused_kernel<NT, MMX_BF16>; break; }
            [[fallthrough]];
        default: set_error("self_chain: unsupported dtype %d%s", dtype, half_chain ? " for the fp16 chain" : ""); return MMX_EINVAL;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    if (args.G > 1) {
        int zrc = zero_async(args.counters, sizeof(unsigned) * args.B, s);
        if (zrc) return zrc;
    }
    kern<<<args.B * args.G, kChainThreads, lds, s>>>(args);
    MMX_LAUNCH_CHECK("self_chain_fused_kernel");
    return MMX_OK;
}

extern "C" int mmx_relevancy_self_chain(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                        int B, int H, int N, int dtype, const void* R_init_dev, void* R_out_dev,
                                        const void* Rsq_init_dev, void* Rsq_out_dev, int M, void* workspace_dev,
                                        size_t workspace_bytes, void* stream) {
    return mmx_relevancy_self_chain_ex(attn_layers, grad_layers, n_layers, B, H, N, dtype, -1, R_init_dev, R_out_dev,
                                       Rsq_init_dev, Rsq_out_dev, M, workspace_dev, workspace_bytes, stream);
}

extern "C" int mmx_relevancy_self_chain_ex(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                           int B, int H, int N, int dtype, int64_t attn_batch_stride,
                                           const void* R_init_dev, void* R_out_dev, const void* Rsq_init_dev,
                                           void* Rsq_out_dev, int M, void* workspace_dev, size_t workspace_bytes,
                                           void* stream) {
    return mmx_relevancy_self_chain_flags(attn_layers, grad_layers, n_layers, B, H, N, dtype, attn_batch_stride, R_init_dev, R_out_dev,
                                          Rsq_init_dev, Rsq_out_dev, M, 0u, workspace_dev, workspace_bytes, stream);
}

extern "C" int mmx_relevancy_self_chain_flags(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                              int B, int H, int N, int dtype, int64_t attn_batch_stride,
                                              const void* R_init_dev, void* R_out_dev, const void* Rsq_init_dev,
                                              void* Rsq_out_dev, int M, unsigned flags, void* workspace_dev, size_t workspace_bytes,
                                              void* stream) {
    MMX_CHECK_ARG((flags & ~MMX_CHAIN_CAUSAL) == 0u, "mmx_relevancy_self_chain_flags: unknown flag bits %#x", flags);
    const int stream_policy = g_chain_nt | ((flags & MMX_CHAIN_CAUSAL) ? 2 : 0);   // -> the stream waves of the fp32 chain kernels
    const int64_t full_stride = static_cast<int64_t>(H) * N * N;
    if (attn_batch_stride < 0) attn_batch_stride = full_stride;
    MMX_CHECK_ARG(attn_batch_stride == 0 || attn_batch_stride == full_stride,
                  "mmx_relevancy_self_chain_ex: attn_batch_stride must be 0 (shared forward) or H*N*N");
    MMX_CHECK_ARG(attn_layers && grad_layers && R_out_dev, "mmx_relevancy_self_chain: null pointer");
    MMX_CHECK_ARG(n_layers >= 0 && n_layers <= MMX_MAX_LAYERS, "mmx_relevancy_self_chain: n_layers %d not in [0, %d]",
                  n_layers, MMX_MAX_LAYERS);
    MMX_CHECK_ARG(B > 0 && H > 0 && N > 0 && M >= 0, "mmx_relevancy_self_chain: non-positive size");
    MMX_CHECK_ARG((M > 0) == (Rsq_out_dev != nullptr), "mmx_relevancy_self_chain: R_sq output and M must agree");
    MMX_CHECK_ARG(dtype == MMX_F32 || dtype == MMX_F16 || dtype == MMX_BF16, "mmx_relevancy_self_chain: dtype %d", dtype);
    for (int l = 0; l < n_layers; ++l)
        MMX_CHECK_ARG(attn_layers[l] && grad_layers[l], "mmx_relevancy_self_chain: null layer pointer %d", l);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = nt_for(N);

    if (use_cols(n_layers, B, H, N, M, dtype))
        return self_chain_cols_launch(attn_layers, grad_layers, n_layers, B, H, N, attn_batch_stride, R_init_dev, R_out_dev, stream_policy,
                                      cols_debug_flags(g_debug_flags), s);
    if (nt <= 8 && M == 0) {
        ChainArgs args;
        memset(&args, 0, sizeof(args));
        for (int l = 0; l < n_layers; ++l) { args.attn[l] = attn_layers[l]; args.grad[l] = grad_layers[l]; }
        args.n_layers = n_layers; args.B = B; args.H = H; args.N = N;
        args.R_init = static_cast<const float*>(R_init_dev);
        args.R_out = static_cast<float*>(R_out_dev);
        args.G = fused_groups(n_layers, B, H, N);
        args.debug = g_debug_flags;
        args.attn_bstride = attn_batch_stride;
        args.nt = g_chain_nt;
        args.pipe = N * N >= 1600 ? g_chain_pipe : 0;   // (below ~40 tokens the pipeline's bookkeeping costs more than it hides)
        if (args.G > 1) {
            const size_t need = mmx_self_chain_workspace_bytes(n_layers, B, H, N, M, dtype);
            if (workspace_bytes < need || !workspace_dev) {
                set_error("mmx_relevancy_self_chain: workspace %zu < %zu", workspace_bytes, need);
                return MMX_EWORKSPACE;
            }
            args.counters = static_cast<unsigned*>(workspace_dev);
            args.parts = reinterpret_cast<float*>(static_cast<char*>(workspace_dev) + group_counter_bytes(B));
            // fp32 slabs: the kernel with barrier-free stream waves (relevancy_chain_groups.hip) unless algo 1 asks for this file's
            if (g_chain_algo != 1 && dtype == MMX_F32 && self_chain_groups_applies(n_layers, args.G, H, N))
                return self_chain_groups_launch(attn_layers, grad_layers, n_layers, B, H, N, args.G, attn_batch_stride, R_init_dev,
                                                R_out_dev, args.counters, args.parts, stream_policy, g_debug_flags, s);
        }
        switch (nt) {
            case 1: return launch_fused<1>(args, dtype, s);
            case 2: return launch_fused<2>(args, dtype, s);
            case 3: return launch_fused<3>(args, dtype, s);
            case 4: return launch_fused<4>(args, dtype, s);
            case 5: return launch_fused<5>(args, dtype, s);
            case 6: return launch_fused<6>(args, dtype, s);
            case 7: return launch_fused<7>(args, dtype, s);
            default: return launch_fused<8>(args, dtype, s);
        }
    }

    // ---- split path: per layer  A_bar = avg_heads(A_l, G_l);  R' = R + A_bar . R  [; R_sq' = R_sq + A_bar . R_sq]
    const size_t need = mmx_self_chain_workspace_bytes(n_layers, B, H, N, M, dtype);
    if (workspace_bytes < need || !workspace_dev) {
        set_error("mmx_relevancy_self_chain: workspace %zu < %zu", workspace_bytes, need);
        return MMX_EWORKSPACE;
    }
    const int64_t nn = static_cast<int64_t>(N) * N, nm = static_cast<int64_t>(N) * M;
    const size_t mat = align256(sizeof(float) * static_cast<size_t>(B) * N * N);
    char* ws = static_cast<char*>(workspace_dev);
    float* abar = reinterpret_cast<float*>(ws);
    float* Rpong = reinterpret_cast<float*>(ws + mat);
    float* SQpong = reinterpret_cast<float*>(ws + 2 * mat);
    float* Rout = static_cast<float*>(R_out_dev);
    float* SQout = static_cast<float*>(Rsq_out_dev);

    // state starts in the buffer that makes the LAST product land in the caller's output
    float* Rcur = (n_layers % 2 == 0) ? Rout : Rpong;
    float* SQcur = (n_layers % 2 == 0) ? SQout : SQpong;
    hipError_t e;
    const bool rows_path = g_chain_rows && M == 0 && chain_rows_layer_applies(N);
    if (R_init_dev) {
        e = hipMemcpyAsync(Rcur, R_init_dev, sizeof(float) * B * nn, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(R_init)");
    } else if (!rows_path || n_layers == 0) {
        int rc = identity_async(Rcur, B, N, s);
        if (rc) return rc;
    }
    if (rows_path) {
        // one launch per layer: the head reduction lands in LDS and is multiplied from there (relevancy_chain_rows.hip); the first
        // layer of a chain that starts at the identity is R = I + A_bar (no product, the identity is never materialised)
        for (int l = 0; l < n_layers; ++l) {
            float* Rnxt = (Rcur == Rout) ? Rpong : Rout;
            const float* Rin = (l == 0 && !R_init_dev) ? nullptr : Rcur;
            int rc = chain_rows_layer_launch(attn_layers[l], grad_layers[l], Rin, Rnxt, B, H, N, dtype, attn_batch_stride, s, g_debug_flags);
            if (rc) return rc;
            Rcur = Rnxt;
        }
        return MMX_OK;
    }
    if (M > 0) {
        if (Rsq_init_dev) {
            e = hipMemcpyAsync(SQcur, Rsq_init_dev, sizeof(float) * B * nm, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(R_sq_init)");
        } else {
            int rc = zero_async(SQcur, sizeof(float) * B * nm, s);
            if (rc) return rc;
        }
    }
    // (Round 5 measured the head reductions on a side stream one layer ahead of the products -- A_bar_{l+1} does not depend on R_l,
    // one kernel is an HBM stream, the other an MFMA loop -- and found NO gain at 197 / 577 / 950 tokens, with or without capping
    // the product's workgroups per CU: profiles/r05_chain_split_roofline.txt.  One stream, two launches per layer.)
    for (int l = 0; l < n_layers; ++l) {
        int rc = avg_heads_launch(attn_layers[l], grad_layers[l], abar, B, H, N, N, dtype, attn_batch_stride, stream);
        if (rc) return rc;
        float* Rnxt = (Rcur == Rout) ? Rpong : Rout;
        rc = mmx_bmm_f32(abar, Rcur, Rcur, Rnxt, B, N, N, N, 0, nn, nn, nn, 0, stream);
        if (rc) return rc;
        Rcur = Rnxt;
        if (M > 0) {
            float* SQnxt = (SQcur == SQout) ? SQpong : SQout;
            rc = mmx_bmm_f32(abar, SQcur, SQcur, SQnxt, B, N, M, N, 0, nn, nm, nm, 0, stream);
            if (rc) return rc;
            SQcur = SQnxt;
        }
    }
    return MMX_OK;
}

// R' = round_f16(R + round_f16(P)): the two roundings of `R = R + torch.bmm(cam, R)` on fp16 tensors
__global__ __launch_bounds__(256) void half_chain_add_kernel(const float* __restrict__ R, const float* __restrict__ P,
                                                             float* __restrict__ out, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = round_f16(R[i] + round_f16(P[i]));
}

// The reference's HALF-PRECISION chain (CLIP_explainability.ipynb cell 6:20-32, 43-55 on a model after `convert_weights`,
// CLIP/clip/model.py:381-402: R is created in the dtype of the fp16 attention probabilities): same rule, every tensor-level result
// rounded to fp16.  R_out: fp32 storage of fp16-representable values.  N <= 128: the fused kernel, one workgroup per sample;
// larger N: avg_heads + bmm + a rounding add per layer (workspace: mmx_self_chain_workspace_bytes(..., M = 0)).
extern "C" int mmx_relevancy_self_chain_half(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                             int B, int H, int N, int dtype, int64_t attn_batch_stride, void* R_out_dev,
                                             void* workspace_dev, size_t workspace_bytes, void* stream) {
    const int64_t full_stride = static_cast<int64_t>(H) * N * N;
    if (attn_batch_stride < 0) attn_batch_stride = full_stride;
    MMX_CHECK_ARG(attn_batch_stride == 0 || attn_batch_stride == full_stride,
                  "mmx_relevancy_self_chain_half: attn_batch_stride must be 0 (shared forward) or H*N*N");
    MMX_CHECK_ARG(attn_layers && grad_layers && R_out_dev, "mmx_relevancy_self_chain_half: null pointer");
    MMX_CHECK_ARG(n_layers >= 0 && n_layers <= MMX_MAX_LAYERS, "mmx_relevancy_self_chain_half: n_layers %d not in [0, %d]",
                  n_layers, MMX_MAX_LAYERS);
    MMX_CHECK_ARG(B > 0 && H > 0 && N > 0, "mmx_relevancy_self_chain_half: non-positive size");
    MMX_CHECK_ARG(dtype == MMX_F32 || dtype == MMX_F16, "mmx_relevancy_self_chain_half: slabs must be fp32 or fp16, got %d", dtype);
    for (int l = 0; l < n_layers; ++l)
        MMX_CHECK_ARG(attn_layers[l] && grad_layers[l], "mmx_relevancy_self_chain_half: null layer pointer %d", l);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = nt_for(N);
    if (nt <= 8) {
        ChainArgs args;
        memset(&args, 0, sizeof(args));
        for (int l = 0; l < n_layers; ++l) { args.attn[l] = attn_layers[l]; args.grad[l] = grad_layers[l]; }
        args.n_layers = n_layers; args.B = B; args.H = H; args.N = N;
        args.R_out = static_cast<float*>(R_out_dev);
        args.G = 1;
        args.attn_bstride = attn_batch_stride;
        switch (nt) {
            case 1: return launch_fused<1>(args, dtype, s, true);
            case 2: return launch_fused<2>(args, dtype, s, true);
            case 3: return launch_fused<3>(args, dtype, s, true);
            case 4: return launch_fused<4>(args, dtype, s, true);
            case 5: return launch_fused<5>(args, dtype, s, true);
            case 6: return launch_fused<6>(args, dtype, s, true);
            case 7: return launch_fused<7>(args, dtype, s, true);
            default: return launch_fused<8>(args, dtype, s, true);
        }
    }
    const size_t need = mmx_self_chain_workspace_bytes(n_layers, B, H, N, 0, dtype);
    if (workspace_bytes < need || !workspace_dev) {
        set_error("mmx_relevancy_self_chain_half: workspace %zu < %zu", workspace_bytes, need);
        return MMX_EWORKSPACE;
    }
    const int64_t nn = static_cast<int64_t>(N) * N;
    const size_t mat = align256(sizeof(float) * static_cast<size_t>(B) * N * N);
    char* ws = static_cast<char*>(workspace_dev);
    float* abar = reinterpret_cast<float*>(ws);
    float* prod = reinterpret_cast<float*>(ws + mat);
    float* R = static_cast<float*>(R_out_dev);
    int rc = identity_async(R, B, N, s);
    if (rc) return rc;
    dim3 grid(static_cast<unsigned>((((nn + 3) >> 2) + 255) / 256), B);
    for (int l = 0; l < n_layers; ++l) {
        if (dtype == MMX_F32)
            avg_heads_kernel<MMX_F32, true><<<grid, 256, 0, s>>>(attn_layers[l], grad_layers[l], abar, H, nn, attn_batch_stride);
        else
            avg_heads_kernel<MMX_F16, true><<<grid, 256, 0, s>>>(attn_layers[l], grad_layers[l], abar, H, nn, attn_batch_stride);
        MMX_LAUNCH_CHECK("avg_heads_kernel (fp16 chain)");
        rc = mmx_bmm_f32(abar, R, nullptr, prod, B, N, N, N, 0, nn, nn, nn, 0, stream);
        if (rc) return rc;
        const int64_t n = static_cast<int64_t>(B) * nn;
        half_chain_add_kernel<<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(R, prod, R, n);
        MMX_LAUNCH_CHECK("half_chain_add_kernel");
    }
    return MMX_OK;
}
