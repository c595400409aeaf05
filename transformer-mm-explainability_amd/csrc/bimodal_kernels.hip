// Single-launch bi-modal relevancy schedule (LXMERT-style two-stream model): rules 5, 6, 7, 10, 11 + eq. 8-9.
//
// Replaces GeneratorOurs.generate_ours' rule schedule (lxmert/lxmert/src/ExplanationGenerator.py:131-211, helpers
// :18-54,61-129): 9 language + 5 vision self-attention layers, 5 cross layers with both cross directions and two
// self-attention blocks each -- 38 rule applications, every one of which is 4-8 small ATen launches in the reference
// and was 3-6 launches in our multi-launch path.  At these sizes (T <= 48 text tokens, I = 36 boxes) the problem is
// launch/latency-bound, so the whole schedule runs in ONE launch (two phases, below); HBM traffic = the captured slabs once
// (~2 MB per sample).  The round 1-3 form (one workgroup per sample, everything in LDS, products on the fp32 VALU: 362 us at
// B = 32) was removed in round 5; the kernel below took over all of its shapes in round 4 (109 us).
#include "mmx_common.h"
#include <stdlib.h>

namespace mmx {

constexpr int kBmMax = 48;       // max tokens per modality
constexpr int kBmMaxLayers = 16;

__device__ __forceinline__ void atomic_min_float_bm(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// =====================================================================================================
// lxmert_schedule_v2_kernel (round 4): the schedule in two phases.
//   phase 1  workgroup (b, w) head-averages the blocks k = w, w + W, ... of sample b (rule 5: 16-byte loads at 4-byte
//            alignment, 12 in flight per lane, heads in order) and writes A_bar_k -- 1 / (2H) of the bytes read -- to an
//            L2-resident scratch with write-through stores; every CU streams.
//   publish  per-wave vmcnt(0), barrier, one relaxed agent-scope ticket per workgroup (cdna guide G16, write-through form).
//   phase 2  the LAST arriver of a sample (no spinning, placement independent) acquires once and runs the 38 rule
//            applications with all state in LDS: every product is exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) on 16 x 16 tiles
//            dealt to the 8 waves, the A_bar blocks are prefetched from the scratch (registers -> a three-slot LDS ring),
//            eq. 8-9 is computed ONCE per cross layer (both directions read the same two normalised matrices) with 4 lanes
//            per row, the diag word is reduced in LDS and leaves as one global atomic.
// mode 0: one launch (grid B x W, tickets).  mode 1 / 2: the two phases as two launches (grid B x nblk without LDS, then grid
// B): no tickets, no acquire; every block of every sample is its own workgroup in phase 1.
// Block order (storage = consumption): lang[0..nl), vis[0..nv), then per cross layer xlc, xic, xls, xis (the last layer has no
// xic / xis).  Same arithmetic as the reference's schedule; products are summed in MFMA k-order instead of sequentially.
// =====================================================================================================
constexpr int kV2BmThreads = 512;
constexpr int kV2BmWaves = kV2BmThreads / 64;
constexpr int kV2MaxBlk = 2 * kBmMaxLayers + 4 * kBmMaxLayers;

struct BimodalV2Args {
    const float* A[kV2MaxBlk];     // block k's probability / gradient slabs, in consumption order
    const float* G[kV2MaxBlk];
    int n_lang, n_vis, n_x;
    int B, H, T, I;
    unsigned flags;
    float *R_tt, *R_ti, *R_ii, *R_it;
    float* diag_min;
    const int* text_len;
    float* abar;          // [B][nblk][bs]  A_bar blocks, row stride = the block's padded key count
    unsigned* counters;   // [B] arrival tickets (zeroed before the launch)
    int nblk, bs, W, D16;
    int mode;             // 0 one launch | 1 phase 1 only | 2 phase 2 only
    int debug;            // profiling only (env MMX_BM_DEBUG): bit0 = stop after phase 1, bit1 = phase 2 without the MFMA tiles,
                          // bit2 = no A_bar staging inside the loop, bit3 = no add_mat, bit4 = no eq. 8-9, bit5 = no barriers
};

struct M2 {               // an LDS matrix [D16][ld]
    float* p;
    int ld;
    __device__ __forceinline__ float& at(int i, int j) const { return p[i * ld + j]; }
};

struct BlkInfo { int pq, pk, nq, nk, kind, layer; };
enum { BK_LANG = 0, BK_VIS, BK_XLC, BK_XIC, BK_XLS, BK_XIS };

__device__ __forceinline__ void bm_dims(BlkInfo& r, int PT, int I, int T) {
    const bool q_lang = r.kind == BK_LANG || r.kind == BK_XLC || r.kind == BK_XLS;
    const bool k_lang = r.kind == BK_LANG || r.kind == BK_XIC || r.kind == BK_XLS;
    r.pq = q_lang ? PT : I; r.nq = q_lang ? T : I;
    r.pk = k_lang ? PT : I; r.nk = k_lang ? T : I;
}

// kind / layer / padded + real dims of block k (pure arithmetic on scalars: nothing here indexes the argument struct)
__device__ __forceinline__ BlkInfo bm_block(int k, int n_lang, int n_vis, int n_x, int PT, int I, int T) {
    BlkInfo r;
    if (k < n_lang) { r.kind = BK_LANG; r.layer = k; }
    else if (k < n_lang + n_vis) { r.kind = BK_VIS; r.layer = k - n_lang; }
    else {
        const int q = k - n_lang - n_vis, full = 4 * (n_x - 1);
        if (q < full) { r.layer = q >> 2; r.kind = BK_XLC + (q & 3); }
        else { r.layer = n_x - 1; r.kind = (q - full) == 0 ? BK_XLC : BK_XLS; }
    }
    bm_dims(r, PT, I, T);
    return r;
}

// one 16 x 16 tile of A[i0.., :K] . B[:K, j0..]; TA: A is stored K x M.  The A operand is masked beyond K, so only B's rows
// beyond K have to be finite (they are zero: no matrix is ever written outside its valid block).  The contraction runs in
// groups of 16 (four MFMA k-steps): the eight operand reads of a group are issued before its first MFMA, so the LDS latency is
// paid once per group instead of once per k-step (K <= 48: at most three groups; rows up to round16(K) <= D16 exist).
template <bool TA>
__device__ __forceinline__ f32x4 bm_tile(M2 A, M2 B, int i0, int j0, int K, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const float* ap = TA ? A.p + lk * A.ld + i0 + li : A.p + (i0 + li) * A.ld + lk;
    const float* bp = B.p + lk * B.ld + j0 + li;
    const int astep = TA ? 4 * A.ld : 4, bstep = 4 * B.ld;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            av[s] = ap[s * astep];
            bv[s] = bp[s * bstep];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16x16x4((k0 + 4 * s + lk < K) ? av[s] : 0.f, bv[s], acc);
        ap += 4 * astep; bp += 4 * bstep;
    }
    return acc;
}

__global__ __launch_bounds__(kV2BmThreads) void lxmert_schedule_v2_kernel(const BimodalV2Args v) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BimodalV2Args& a = v;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / v.W, w = blockIdx.x - b * v.W;
    const int PT = a.T, I = a.I, H = a.H;
    const int T = a.text_len ? min(max(a.text_len[b], 1), PT) : PT;
    const float fH = static_cast<float>(H);

    if (v.mode != 2) {
        // -------------------------------------------------------------- phase 1: rule 5 for this workgroup's blocks
        for (int k = w; k < v.nblk; k += v.W) {
            const BlkInfo bi = bm_block(k, v.n_lang, v.n_vis, v.n_x, PT, I, T);
            const float* __restrict__ Ak = v.A[k];
            const float* __restrict__ Gk = v.G[k];
            const int hs = bi.pq * bi.pk;
            const int64_t sample = static_cast<int64_t>(b) * H * hs;
            const int64_t slab_end = static_cast<int64_t>(a.B) * H * hs;
            float* dst = v.abar + (static_cast<int64_t>(b) * v.nblk + k) * v.bs;
            for (int c = tid; c * 4 < hs; c += kV2BmThreads) {
                const int p = c * 4;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                if (sample + static_cast<int64_t>(H - 1) * hs + p + 3 < slab_end) {   // 16-byte loads stay inside the slab
                    int h = 0;
                    for (; h + 6 <= H; h += 6) {
                        f32x4 av[6], gv[6];
#pragma unroll
                        for (int u = 0; u < 6; ++u) {
                            av[u] = ldg4_u(Ak + sample + static_cast<int64_t>(h + u) * hs + p);
                            gv[u] = ldg4_u(Gk + sample + static_cast<int64_t>(h + u) * hs + p);
                        }
#pragma unroll
                        for (int u = 0; u < 6; ++u) {
                            const f32x4 x = gv[u] * av[u];
                            s[0] += relu_nan(x[0]); s[1] += relu_nan(x[1]); s[2] += relu_nan(x[2]); s[3] += relu_nan(x[3]);
                        }
                    }
                    for (; h < H; ++h) {
                        const f32x4 x = ldg4_u(Gk + sample + static_cast<int64_t>(h) * hs + p) *
                                        ldg4_u(Ak + sample + static_cast<int64_t>(h) * hs + p);
                        s[0] += relu_nan(x[0]); s[1] += relu_nan(x[1]); s[2] += relu_nan(x[2]); s[3] += relu_nan(x[3]);
                    }
                } else {
                    for (int e = 0; e < 4 && p + e < hs; ++e)
                        for (int h = 0; h < H; ++h)
                            s[e] += relu_nan(Gk[sample + static_cast<int64_t>(h) * hs + p + e] *
                                             Ak[sample + static_cast<int64_t>(h) * hs + p + e]);
                }
                // write-through (sc1) 8-byte stores: the publish below then needs no L2 write-back fence.  Elements of the
                // chunk beyond the head slab (hs % 4 != 0) are garbage of the NEXT head and never read (phase 2 masks by hs).
                const f32x2 lo = {s[0] / fH, s[1] / fH}, hi = {s[2] / fH, s[3] / fH};
                unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst + p);
                __hip_atomic_store(d8, __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d8 + 1, __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (v.mode == 1) return;
        // -------------------------------------------------------------- publish + ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        unsigned* ticket_lds = reinterpret_cast<unsigned*>(smem);
        if (tid == 0)
            *ticket_lds = __hip_atomic_fetch_add(v.counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *ticket_lds;
        if (ticket != static_cast<unsigned>(v.W - 1) || (v.debug & 1)) return;
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }

    // ------------------------------------------------------------------ phase 2: the schedule of sample b
    // Written for a SMALL instruction footprint: one workgroup per CU executes this path, so every instruction line it
    // touches is an L2 -> I-cache miss; every rule application runs through ONE copy of each loop, with the operands picked at
    // run time (a first version with everything unrolled / inlined per block kind was ~12 500 instructions).
    const int D16 = v.D16, LD = D16 + 4, MS = D16 * LD;
    float* const base = smem;
    auto mat = [&](int i) { return M2{base + i * MS, LD}; };
    // 0 R_tt  1 R_ii  2 R_ti  3 R_it  4 N_t  5 N_i  6..8 cam ring  9 tmp  10 park_sq (R_ti additions / self R_sq update)
    // 11 park_ss (R_tt additions / self R_ss update)
    const M2 R_tt = mat(0), R_ii = mat(1), R_ti = mat(2), R_it = mat(3), N_t = mat(4), N_i = mat(5), tmp = mat(9),
             park_sq = mat(10), park_ss = mat(11);
    const bool normalize = a.flags & MMX_MM_NORMALIZE, self10 = a.flags & MMX_MM_SELF_IN_RULE10;
    const bool no_mfma = v.debug & 2;
#define lds_barrier() do { if (!(v.debug & 32)) mmx::lds_barrier(); } while (0)

    float* const diag_lds = smem + 12 * MS;                 // + inf .. the smallest diag(R - I) any eq. 8-9 of this sample saw
    signed char* const blk_kind = reinterpret_cast<signed char*>(smem + 12 * MS + 4);   // [nblk] kind | layer << 3, built once
    for (int e = tid; e < 12 * MS; e += kV2BmThreads) smem[e] = 0.f;
    if (tid == 0) *diag_lds = __builtin_inff();
    if (tid < v.nblk) {
        const BlkInfo bi = bm_block(tid, v.n_lang, v.n_vis, v.n_x, PT, I, T);
        blk_kind[tid] = static_cast<signed char>(bi.kind | (bi.layer << 3));
    }
    __syncthreads();
    for (int i = tid; i < D16; i += kV2BmThreads) {
        if (i < T) R_tt.at(i, i) = 1.f;
        if (i < I) R_ii.at(i, i) = 1.f;
    }
    // block descriptor from the LDS table (a handful of instructions instead of bm_block's compare chain, three times per block)
    auto blk = [&](int k) {
        BlkInfo r;
        const int kl = blk_kind[k];
        r.kind = kl & 7; r.layer = kl >> 3;
        bm_dims(r, PT, I, T);
        return r;
    };
    // (row, col) of this lane's two chunks inside a block whose padded key count is PT / I: computed once (no division per block)
    int st_row[2][2], st_col[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e0 = (tid + u * kV2BmThreads) * 4;
        st_row[u][0] = e0 / PT; st_col[u][0] = e0 - st_row[u][0] * PT;
        st_row[u][1] = e0 / I;  st_col[u][1] = e0 - st_row[u][1] * I;
    }

    const float* scratch = v.abar + static_cast<int64_t>(b) * v.nblk * v.bs;
    f32x4 pre[2];
    auto issue = [&](int k) {     // global -> registers: block k's A_bar, <= 2 chunks per lane (48 x 48 / 4 / 512)
        if (k >= v.nblk) return;
        const BlkInfo bi = blk(k);
        const int hs = bi.pq * bi.pk;
        const f32x4* src = reinterpret_cast<const f32x4*>(scratch + static_cast<int64_t>(k) * v.bs);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = tid + u * kV2BmThreads;
            pre[u] = (c * 4 < hs) ? src[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](int k) {     // registers -> LDS ring slot k % 3, masked to the sample's real block
        if (k >= v.nblk) return;
        const BlkInfo bi = blk(k);
        const int hs = bi.pq * bi.pk;
        const M2 cam = mat(6 + k % 3);
        const int w_ = bi.pk == PT ? 0 : 1;       // (PT == I: both variants are identical)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = tid + u * kV2BmThreads;
            if (c * 4 < hs) {
                int row = st_row[u][w_], col = st_col[u][w_];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c * 4 + e < hs) cam.at(row, col) = (row < bi.nq && col < bi.nk) ? pre[u][e] : 0.f;
                    if (++col == bi.pk) { col = 0; ++row; }
                }
            }
        }
    };
    const int li = lane & 15, lk4 = (lane >> 4) * 4;
    // dst[i0.., j0..] (+)= acc for the valid part of a tile
    auto put = [&](M2 dst, const f32x4& acc, int i0, int j0, int M, int N, bool add) {
        const int col = j0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk4 + r;
            if (row < M && col < N) dst.at(row, col) = add ? dst.at(row, col) + acc[r] : acc[r];
        }
    };
    // tiles of [A . B0 | A . B1] (A: M x K; B0: K x N0 -> D0, B1: K x N1 -> D1), dealt to the waves; TA: A stored K x M
    auto products = [&](bool ta, M2 A, int M, int K, M2 B0, int N0, M2 D0, bool add0, M2 B1, int N1, M2 D1, bool add1) {
        const int tr = (M + 15) >> 4, t0 = (N0 + 15) >> 4, t1 = (N1 + 15) >> 4, per = t0 + t1, total = tr * per;
        for (int t = wave; t < total; t += kV2BmWaves) {
            const int ti = t / per, tj = t - ti * per;
            const bool first = tj < t0;
            const M2 Bm = first ? B0 : B1, Dm = first ? D0 : D1;
            const int j0 = (first ? tj : tj - t0) * 16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!no_mfma) acc = ta ? bm_tile<true>(A, Bm, ti * 16, j0, K, lane) : bm_tile<false>(A, Bm, ti * 16, j0, K, lane);
            put(Dm, acc, ti * 16, j0, M, first ? N0 : N1, first ? add0 : add1);
        }
    };
    auto add_mat = [&](M2 dst, M2 src, int M, int N) {
        if (v.debug & 8) return;
        for (int e = tid; e < M * N; e += kV2BmThreads) {
            const int i = e / N, j = e - i * N;
            dst.at(i, j) += src.at(i, j);
        }
    };
    // eq. 8-9 of R_tt -> N_t and R_ii -> N_i (4 lanes per row), or plain copies without normalisation
    auto residuals = [&]() {
        const int g = tid >> 2, part = tid & 3;
        const bool is_t = g < T;
        const int i = is_t ? g : g - T, n = is_t ? T : I;
        const bool active = is_t || i < I;
        const M2 R = is_t ? R_tt : R_ii, N = is_t ? N_t : N_i;
        float s = 0.f;
        if (active && normalize)
            for (int j = part; j < n; j += 4) s += R.at(i, j) - (j == i ? 1.f : 0.f);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (active) {
            for (int j = part; j < n; j += 4) {
                const float d = (j == i) ? 1.f : 0.f;
                N.at(i, j) = normalize ? (R.at(i, j) - d) / s + d : R.at(i, j);
            }
            // the workgroup's running minimum lives in LDS (LDS atomics on the sign-split integer image of the float); ONE global
            // atomic leaves at the very end.  NaN sticks: as 0xffc00000 it is below every float in this ordering.
            if (normalize && part == 0 && a.diag_min) {
                const float dv = R.at(i, i) - 1.f;
                if (dv != dv) atomicMax(reinterpret_cast<unsigned int*>(diag_lds), 0xffc00000u);
                else if (dv >= 0.f) atomicMin(reinterpret_cast<int*>(diag_lds), __float_as_int(dv));
                else atomicMax(reinterpret_cast<unsigned int*>(diag_lds), __float_as_uint(dv));
            }
        }
    };

    issue(0);
    stage(0);
    issue(1);
    stage(1);
    issue(2);
    __syncthreads();
    bool crossed = false;        // R_ti / R_it are still zero before the first cross layer: rule 7 adds nothing
    for (int k = 0; k < v.nblk; ++k) {
        if (!(v.debug & 4)) {
            stage(k + 2);        // ring slot (k + 2) % 3 was last read by block k - 1, which ended with a barrier
            issue(k + 3);        // ... and is staged one block later
        }
        const BlkInfo bi = blk(k);
        const M2 cam = mat(6 + k % 3);
        const bool lang_side = bi.kind == BK_LANG || bi.kind == BK_XLS || bi.kind == BK_XLC;
        const M2 Rss = lang_side ? R_tt : R_ii, Rsq = lang_side ? R_ti : R_it, Rqs = lang_side ? R_it : R_ti;
        const int ns = lang_side ? T : I, nq = lang_side ? I : T;
        if (bi.kind == BK_XLC || bi.kind == BK_XIC) {
            // rules 10 + 11 (queries s, keys q) from the PRE-update state: sq = Ss^T . (cam . Qq) (or cam itself), ss = cam . R_qs.
            // The language direction parks its pair (park_sq, park_ss) unless it is the last layer; the image direction adds
            // directly (nothing reads R_ii / R_it again inside this cross layer) and then lands the parked pair.
            const bool direct = bi.kind == BK_XIC || bi.layer == a.n_x - 1;
            const M2 Ss = lang_side ? N_t : N_i, Qq = lang_side ? N_i : N_t;
            if (bi.kind == BK_XLC) {
                if (!(v.debug & 16)) residuals();
                lds_barrier();
            }
            products(false, cam, ns, nq, Qq, self10 ? nq : 0, tmp, false, Rqs, ns, direct ? Rss : park_ss, direct);
            lds_barrier();
            if (self10) products(true, Ss, ns, ns, tmp, nq, direct ? Rsq : park_sq, direct, tmp, 0, tmp, false);
            else {
                const M2 dst = direct ? Rsq : park_sq;
                for (int e = tid; e < ns * nq; e += kV2BmThreads) {
                    const int i = e / nq, j = e - i * nq;
                    dst.at(i, j) = direct ? dst.at(i, j) + cam.at(i, j) : cam.at(i, j);
                }
            }
            if (bi.kind == BK_XIC) {
                lds_barrier();   // every read of R_ti (the R_qs operand above) is done: the parked language pair may land
                add_mat(R_ti, park_sq, T, I);
                add_mat(R_tt, park_ss, T, T);
            }
            crossed = true;
            lds_barrier();
        } else {
            // rules 6 + 7: R_ss += cam . R_ss ; R_sq += cam . R_sq, both from the old state (updates wait in the park buffers)
            products(false, cam, ns, ns, Rss, ns, park_ss, false, Rsq, crossed ? nq : 0, park_sq, false);
            lds_barrier();
            add_mat(Rss, park_ss, ns, ns);
            if (crossed) add_mat(Rsq, park_sq, ns, nq);
            lds_barrier();
        }
    }
#undef lds_barrier
    if (tid == 0) R_tt.at(0, 0) = 0.f;   // disregard the [CLS] token itself (:210)
    __syncthreads();
    if (tid == 0 && a.diag_min && normalize) {
        const float dv = *diag_lds;
        if (dv != dv) atomicMax(reinterpret_cast<unsigned int*>(a.diag_min), 0xffc00000u);
        else atomic_min_float_bm(a.diag_min, dv);
    }

    for (int e = tid; e < PT * PT; e += kV2BmThreads) {
        const int i = e / PT, j = e - i * PT;
        a.R_tt[static_cast<int64_t>(b) * PT * PT + e] = (i < T && j < T) ? R_tt.at(i, j) : 0.f;
    }
    for (int e = tid; e < PT * I; e += kV2BmThreads) {
        const int i = e / I, j = e - i * I;
        a.R_ti[static_cast<int64_t>(b) * PT * I + e] = (i < T) ? R_ti.at(i, j) : 0.f;
    }
    if (a.R_ii)
        for (int e = tid; e < I * I; e += kV2BmThreads) a.R_ii[static_cast<int64_t>(b) * I * I + e] = R_ii.at(e / I, e % I);
    if (a.R_it)
        for (int e = tid; e < I * PT; e += kV2BmThreads) {
            const int i = e / PT, j = e - i * PT;
            a.R_it[static_cast<int64_t>(b) * I * PT + e] = (j < T) ? R_it.at(i, j) : 0.f;
        }
}

// diag word <- +inf and the arrival tickets <- 0 in one small launch (no memset nodes: see mmx_common.h)
__global__ void bm_v2_reset_kernel(float* diag, unsigned* counters, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) counters[i] = 0u;
    if (i == 0 && diag) *diag = __builtin_inff();
}


}  // namespace mmx

using namespace mmx;

static inline size_t bm_align256(size_t n) { return (n + 255) & ~static_cast<size_t>(255); }
static inline int bm_nblk(int n_lang, int n_vis, int n_x) { return n_lang + n_vis + 4 * n_x - 2; }
static inline int bm_block_floats(int T, int I) {
    const int m = T > I ? T : I;
    return ((m * m + 3) / 4) * 4;
}
static int g_bm_split = 0;   // MMX_BM_SPLIT=1 (read per call, profiling knob): the two phases as two launches

extern "C" size_t mmx_lxmert_schedule_workspace_bytes(int n_lang, int n_vis, int n_x, int B, int T, int I) {
    if (n_lang < 0 || n_vis < 0 || n_x < 1 || B < 1 || T < 1 || I < 1) return 0;
    return bm_align256(sizeof(unsigned) * static_cast<size_t>(B)) +
           bm_align256(sizeof(float) * static_cast<size_t>(B) * bm_nblk(n_lang, n_vis, n_x) * bm_block_floats(T, I));
}

extern "C" int mmx_lxmert_schedule(const void* const* lang_attn, const void* const* lang_grad, int n_lang,
                                      const void* const* vis_attn, const void* const* vis_grad, int n_vis,
                                      const void* const* x_lang_cross_attn, const void* const* x_lang_cross_grad,
                                      const void* const* x_img_cross_attn, const void* const* x_img_cross_grad,
                                      const void* const* x_lang_self_attn, const void* const* x_lang_self_grad,
                                      const void* const* x_img_self_attn, const void* const* x_img_self_grad, int n_x,
                                      int B, int H, int T, int I, unsigned flags, const void* text_len_dev, void* R_tt_dev,
                                      void* R_ti_dev, void* R_ii_dev, void* R_it_dev, void* diag_min_dev, void* workspace_dev,
                                      size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(R_tt_dev && R_ti_dev, "mmx_lxmert_schedule: null output");
    MMX_CHECK_ARG(B > 0 && H > 0 && T > 0 && I > 0 && n_x >= 1, "mmx_lxmert_schedule: non-positive size");
    MMX_CHECK_ARG(n_lang >= 0 && n_vis >= 0 && n_lang <= kBmMaxLayers && n_vis <= kBmMaxLayers && n_x <= kBmMaxLayers,
                  "mmx_lxmert_schedule: at most %d layers per group", kBmMaxLayers);
    if (T > kBmMax || I > kBmMax) {
        set_error("mmx_lxmert_schedule: T=%d / I=%d exceed the LDS-resident limit %d (use the per-rule entry points)", T, I, kBmMax);
        return MMX_ENOTSUP;
    }
    const size_t need = mmx_lxmert_schedule_workspace_bytes(n_lang, n_vis, n_x, B, T, I);
    if (!workspace_dev || workspace_bytes < need) {
        set_error("mmx_lxmert_schedule: workspace of %zu bytes needed (mmx_lxmert_schedule_workspace_bytes), got %zu", need,
                  workspace_bytes);
        return MMX_EWORKSPACE;
    }
    BimodalV2Args v;
    memset(&v, 0, sizeof(v));
    BimodalV2Args& a = v;
    int nb = 0;
    bool ok = true;
    auto push = [&](const void* const* sa, const void* const* sg, int l) {
        if (!sa || !sg || !sa[l] || !sg[l]) { ok = false; return; }
        v.A[nb] = static_cast<const float*>(sa[l]);
        v.G[nb] = static_cast<const float*>(sg[l]);
        ++nb;
    };
    for (int l = 0; l < n_lang; ++l) push(lang_attn, lang_grad, l);
    for (int l = 0; l < n_vis; ++l) push(vis_attn, vis_grad, l);
    for (int x = 0; x < n_x; ++x) {      // consumption order: xlc, xic, xls, xis (the last layer has no image side)
        push(x_lang_cross_attn, x_lang_cross_grad, x);
        if (x < n_x - 1) push(x_img_cross_attn, x_img_cross_grad, x);
        push(x_lang_self_attn, x_lang_self_grad, x);
        if (x < n_x - 1) push(x_img_self_attn, x_img_self_grad, x);
    }
    MMX_CHECK_ARG(ok, "mmx_lxmert_schedule: null layer pointer");
    a.n_lang = n_lang; a.n_vis = n_vis; a.n_x = n_x; a.B = B; a.H = H; a.T = T; a.I = I; a.flags = flags;
    a.R_tt = static_cast<float*>(R_tt_dev); a.R_ti = static_cast<float*>(R_ti_dev);
    a.R_ii = static_cast<float*>(R_ii_dev); a.R_it = static_cast<float*>(R_it_dev);
    a.diag_min = static_cast<float*>(diag_min_dev);
    a.text_len = static_cast<const int*>(text_len_dev);
    v.nblk = bm_nblk(n_lang, n_vis, n_x);
    v.bs = bm_block_floats(T, I);
    v.counters = static_cast<unsigned*>(workspace_dev);
    v.abar = reinterpret_cast<float*>(static_cast<char*>(workspace_dev) + bm_align256(sizeof(unsigned) * static_cast<size_t>(B)));
    const int m = T > I ? T : I;
    v.D16 = ((m + 15) / 16) * 16;
    {
        const char* dbg = getenv("MMX_BM_DEBUG");
        v.debug = dbg ? atoi(dbg) : 0;
        const char* sp = getenv("MMX_BM_SPLIT");
        if (sp) g_bm_split = atoi(sp);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    bm_v2_reset_kernel<<<(B + 255) / 256, 256, 0, s>>>(a.diag_min, v.counters, B);
    const size_t lds = sizeof(float) * (12 * static_cast<size_t>(v.D16) * (v.D16 + 4) + 4) + kV2MaxBlk;   // + diag word + block table
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lxmert_schedule_v2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (g_bm_split) {
        v.mode = 1; v.W = v.nblk;
        lxmert_schedule_v2_kernel<<<B * v.nblk, kV2BmThreads, 0, s>>>(v);
        MMX_LAUNCH_CHECK("lxmert_schedule_v2_kernel (phase 1)");
        v.mode = 2; v.W = 1;
        lxmert_schedule_v2_kernel<<<B, kV2BmThreads, lds, s>>>(v);
        MMX_LAUNCH_CHECK("lxmert_schedule_v2_kernel (phase 2)");
        return MMX_OK;
    }
    // one launch: workgroups per sample so that the CUs (256 on MI355X) are filled once when the batch allows it (each workgroup is 512
    // threads and holds phase 2's LDS, i.e. one per CU)
    int W = device_cu_count() / B;
    if (W < 1) W = 1;
    if (W > v.nblk) W = v.nblk;
    v.W = W;
    v.mode = 0;
    lxmert_schedule_v2_kernel<<<B * W, kV2BmThreads, lds, s>>>(v);
    MMX_LAUNCH_CHECK("lxmert_schedule_v2_kernel");
    return MMX_OK;
}
