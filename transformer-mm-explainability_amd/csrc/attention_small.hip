// Whole-head-resident attention-capture kernels for short sequences (CLIP 50/77 tokens, LXMERT, BERT-sized N).
//
// The tiled kernels in attention_kernels.hip re-read K/V once per 16-query tile (rocprofv3 FETCH_SIZE: 3.7x the
// algorithmic bytes at N = 77) and run latency-bound 16-row tiles.  Here ONE workgroup owns a (batch, head):
// Q, K, V (and dO, P, dP for backward) are staged into LDS exactly once with 16-B loads, every matrix product is a
// loop of v_mfma_f32_16x16x4_f32 tiles dealt round-robin to the waves, and P / dP leave the chip once, straight
// into the capture slabs.  MFMA operands are read as ds_read_b128 along the contraction index wherever that index
// is contiguous in LDS (the k-order inside a dot product is free: k is visited as (t, r, lane>>4)).
//
// Eligibility (checked on the host, otherwise the tiled kernels run): head_dim % 4 == 0, 16-B aligned rows,
// LDS footprint <= 160 KiB (forward: N <= ~112; backward: N <= 80 at d = 64).
#include "mmx_common.h"
#include "attention_args.h"

namespace mmx {

__device__ __forceinline__ int ceil16(int x) { return (x + 15) & ~15; }

// One staging job: rows [0, rows) x [0, D) of a strided global matrix -> LDS [rows_cap][DP+4], zero padded, x mul.
struct StageJob {
    float* lds;
    const float* base;
    int64_t sn;
    int rows, rows_cap;
    float mul;
};

// Stage up to NJ matrices in ONE pass: the flattened 16-B chunk list of all jobs is walked with UNR loads in
// flight per lane before the first ds_write (a plain per-matrix loop serialises ~15 global round trips per
// workgroup, which was 80 % of the first version's run time).
template <int DP, int NJ, int UNR>
__device__ __forceinline__ void stage_jobs(const StageJob (&jobs)[NJ], int D, int tid, int nthreads) {
    constexpr int LS = DP + 4;
    constexpr int C4 = DP / 4;
    int start[NJ + 1];
    start[0] = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) start[j + 1] = start[j] + jobs[j].rows_cap * C4;
    const int total = start[NJ];
    for (int base_idx = tid; base_idx < total; base_idx += nthreads * UNR) {
        f32x4 v[UNR];
        float* dst[UNR];
        float mul[UNR];
        // Every load is UNCONDITIONAL (invalid chunks read a clamped, always-valid address and are zeroed by `mul`):
        // a load guarded by a divergent branch makes hipcc wait vmcnt(0) per element (cdna guide, trap (c)).
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = min(base_idx + u * nthreads, total - 1);
            // select the job with compile-time indices only: a runtime-indexed jobs[j] would live in scratch memory
            float* lds = jobs[0].lds;
            const float* gbase = jobs[0].base;
            int64_t sn = jobs[0].sn;
            int rows = jobs[0].rows, first = 0;
            float jmul = jobs[0].mul;
#pragma unroll
            for (int k = 1; k < NJ; ++k)
                if (idx >= start[k]) {
                    lds = jobs[k].lds; gbase = jobs[k].base; sn = jobs[k].sn; rows = jobs[k].rows; jmul = jobs[k].mul;
                    first = start[k];
                }
            const int local = idx - first;
            const int r = local / C4, c = (local - r * C4) * 4;
            const bool live = (base_idx + u * nthreads < total);
            const bool real = live && r < rows && c < D;
            dst[u] = live ? lds + r * LS + c : nullptr;
            mul[u] = real ? jmul : 0.f;
            const float* src = gbase + (real ? static_cast<int64_t>(r) * sn + c : 0);
            v[u] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (dst[u]) *reinterpret_cast<f32x4*>(dst[u]) = mul[u] == 0.f ? f32x4{0.f, 0.f, 0.f, 0.f} : v[u] * mul[u];
    }
}

// D(16x16) += A.B^T-style product where BOTH operands are contiguous along the contraction index in LDS:
// A[i][k] at pa + i*lda + k, B[j][k] at pb + j*ldb + k, k in [0, K16*16).
__device__ __forceinline__ f32x4 tile_kk(const float* pa, int lda, const float* pb, int ldb, int K16, int i_a, int kq) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* a = pa + i_a * lda + kq;
    const float* b = pb + i_a * ldb + kq;
    for (int t = 0; t < K16; ++t) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + t * 16);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + t * 16);
        acc = mfma16x16x4(av[0], bv[0], acc);
        acc = mfma16x16x4(av[1], bv[1], acc);
        acc = mfma16x16x4(av[2], bv[2], acc);
        acc = mfma16x16x4(av[3], bv[3], acc);
    }
    return acc;
}

// A contiguous along k (A[i][k] at pa + i*lda + k), B row-major over k (B[k][j] at pb + k*ldb + j)
__device__ __forceinline__ f32x4 tile_kn(const float* pa, int lda, const float* pb, int ldb, int K16, int i_a, int kq) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* a = pa + i_a * lda + kq;
    const float* b = pb + kq * ldb + i_a;
    for (int t = 0; t < K16; ++t) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + t * 16);
        const float* bt = b + t * 16 * ldb;
        acc = mfma16x16x4(av[0], bt[0], acc);
        acc = mfma16x16x4(av[1], bt[ldb], acc);
        acc = mfma16x16x4(av[2], bt[2 * ldb], acc);
        acc = mfma16x16x4(av[3], bt[3 * ldb], acc);
    }
    return acc;
}

// both operands row-major over k: A^T stored as At[k][i] at pa + k*lda + i, B[k][j] at pb + k*ldb + j
__device__ __forceinline__ f32x4 tile_nn(const float* pa, int lda, const float* pb, int ldb, int K16, int i_a, int kq) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* a = pa + kq * lda + i_a;
    const float* b = pb + kq * ldb + i_a;
    for (int t = 0; t < K16; ++t) {
        const float* at = a + t * 16 * lda;
        const float* bt = b + t * 16 * ldb;
        acc = mfma16x16x4(at[0], bt[0], acc);
        acc = mfma16x16x4(at[lda], bt[ldb], acc);
        acc = mfma16x16x4(at[2 * lda], bt[2 * ldb], acc);
        acc = mfma16x16x4(at[3 * lda], bt[3 * ldb], acc);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------- forward
template <int DP>
__global__ __launch_bounds__(512) void attn_fwd_small_kernel(const AttnFwdArgs a) {
    constexpr int LS = DP + 4;
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NPq = ceil16(a.Nq), NPk = ceil16(a.Nk), SS = NPk + 4;
    float* Qs = smem;
    float* Ks = Qs + NPq * LS;
    float* Vs = Ks + NPk * LS;
    float* Ss = Vs + NPk * LS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int i_a = lane & 15, kq = (lane >> 4) * 4;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;

    {
        const StageJob jobs[3] = {{Qs, qb, a.qs.sn, a.Nq, NPq, q_first ? a.scale : 1.f},
                                  {Ks, kb, a.ks.sn, a.Nk, NPk, 1.f},
                                  {Vs, vb, a.vs.sn, a.Nk, NPk, 1.f}};
        if (!(a.debug & 8)) stage_jobs<DP, 3, 8>(jobs, a.D, tid, 512);
    }
    __syncthreads();

    // S = Q.K^T (+ mask)
    const int ntq = NPq >> 4, ntk = NPk >> 4;
    for (int tile = wave; tile < ntq * ntk && !(a.debug & 1); tile += NW) {
        const int ti = tile / ntk, tj = tile - ti * ntk;
        const f32x4 acc = tile_kk(Qs + ti * 16 * LS, LS, Ks + tj * 16 * LS, LS, DP / 16, i_a, kq);
        const int key = tj * 16 + i_a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + kq + r;
            float s = acc[r];
            if (!q_first) s = s / a.scale;
            if (a.mask && key < a.Nk && row < a.Nq)
                s += a.mask[b * a.mask_sb + static_cast<int64_t>(row) * a.mask_sq + key];
            Ss[row * SS + key] = s;
        }
    }
    __syncthreads();

    // row softmax, 4 rows per wave at a time (16 lanes per row); P -> LDS (zero padded) and -> the capture slab
    float* pbase = a.probs + (static_cast<int64_t>(b) * a.H + h) * a.Nq * a.Nk;
    for (int row = wave * 4 + (lane >> 4); row < NPq && !(a.debug & 2); row += NW * 4) {
        float* srow = Ss + row * SS;
        if (row >= a.Nq) {
            for (int j = i_a; j < NPk; j += 16) srow[j] = 0.f;
            continue;
        }
        float m = -__builtin_inff();
        for (int j = i_a; j < a.Nk; j += 16) m = fmaxf(m, srow[j]);
        m = group16_max(m);
        float sum = 0.f;
        for (int j = i_a; j < a.Nk; j += 16) {
            const float e = expf(srow[j] - m);
            srow[j] = e;
            sum += e;
        }
        sum = group16_sum(sum);
        float* prow = pbase + static_cast<int64_t>(row) * a.Nk;
        for (int j = i_a; j < NPk; j += 16) {
            float p = 0.f;
            if (j < a.Nk) {
                p = srow[j] / sum;
                prow[j] = p;
            }
            srow[j] = p;
        }
    }
    __syncthreads();

    // O = P.V
    float* ob = a.o + b * a.os.sb + h * a.os.sh;
    constexpr int ntd = DP / 16;
    for (int tile = wave; tile < ntq * ntd && !(a.debug & 4); tile += NW) {
        const int ti = tile / ntd, td = tile - ti * ntd;
        const f32x4 acc = tile_kn(Ss + ti * 16 * SS, SS, Vs + td * 16, LS, ntk, i_a, kq);
        const int d = td * 16 + i_a;
        if (d < a.D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + kq + r;
                if (row < a.Nq) ob[static_cast<int64_t>(row) * a.os.sn + d] = acc[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
constexpr int kBwdThreads = 1024;   // 16 waves: the dQ/dK/dV tile pool (48-60 tiles) is the longest phase

template <int DP>
__global__ __launch_bounds__(kBwdThreads) void attn_bwd_small_kernel(const AttnBwdArgs a) {
    constexpr int LS = DP + 4;
    constexpr int NW = kBwdThreads / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NPq = ceil16(a.Nq), NPk = ceil16(a.Nk), SS = NPk + 4;
    float* Qs = smem;
    float* dOs = Qs + NPq * LS;
    float* Ks = dOs + NPq * LS;
    float* Vs = Ks + NPk * LS;
    float* Ps = Vs + NPk * LS;      // [NPq][SS]
    float* dSs = Ps + NPq * SS;     // [NPq][SS]  dP, then dS

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int i_a = lane & 15, kq = (lane >> 4) * 4;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    const float* pg = a.probs + b * a.probs_sb + static_cast<int64_t>(h) * a.Nq * a.Nk;
    float* dpg = a.dprobs + head * a.Nq * a.Nk;

    {
        const float* qp = a.need_dqkv ? a.q + b * a.qs.sb + h * a.qs.sh : a.v;   // never read when rows_cap == 0
        const float* kp = a.need_dqkv ? a.k + b * a.ks.sb + h * a.ks.sh : a.v;
        // without dq/dk/dv the Q and K jobs stage zero rows (rows = 0 -> no loads)
        const StageJob jobs[4] = {{dOs, a.dout + b * a.os.sb + h * a.os.sh, a.os.sn, a.Nq, NPq, 1.f},
                                  {Vs, a.v + b * a.vs.sb + h * a.vs.sh, a.vs.sn, a.Nk, NPk, 1.f},
                                  {Qs, qp, a.qs.sn, a.need_dqkv ? a.Nq : 0, a.need_dqkv ? NPq : 0, q_first ? a.scale : 1.f},
                                  {Ks, kp, a.ks.sn, a.need_dqkv ? a.Nk : 0, a.need_dqkv ? NPk : 0, 1.f}};
        stage_jobs<DP, 4, 8>(jobs, a.D, tid, kBwdThreads);
    }
    {   // P rows (Nk floats each, any alignment) with 8 loads in flight per lane
        const int total = NPq * NPk;
        for (int base_idx = tid; base_idx < total; base_idx += kBwdThreads * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base_idx + u * kBwdThreads;
                const int r = idx / NPk, j = idx - r * NPk;
                const bool real = idx < total && r < a.Nq && j < a.Nk;
                const float x = pg[real ? static_cast<int64_t>(r) * a.Nk + j : 0];   // unconditional load
                v[u] = real ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base_idx + u * kBwdThreads;
                if (idx < total) {
                    const int r = idx / NPk, j = idx - r * NPk;
                    Ps[r * SS + j] = v[u];
                }
            }
        }
    }
    __syncthreads();

    // dP = dO.V^T
    const int ntq = NPq >> 4, ntk = NPk >> 4;
    for (int tile = wave; tile < ntq * ntk; tile += NW) {
        const int ti = tile / ntk, tj = tile - ti * ntk;
        const f32x4 acc = tile_kk(dOs + ti * 16 * LS, LS, Vs + tj * 16 * LS, LS, DP / 16, i_a, kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) dSs[(ti * 16 + kq + r) * SS + tj * 16 + i_a] = acc[r];
    }
    __syncthreads();

    // rows (4 per wave at a time): dP -> capture slab; delta = rowsum(dP*P); dS = P*(dP - delta)
    for (int row = wave * 4 + (lane >> 4); row < NPq; row += NW * 4) {
        float* drow = dSs + row * SS;
        if (row >= a.Nq) {
            for (int j = i_a; j < NPk; j += 16) drow[j] = 0.f;
            continue;
        }
        const float* prow = Ps + row * SS;
        float dot = 0.f;
        for (int j = i_a; j < a.Nk; j += 16) {
            const float dp = drow[j];
            dpg[static_cast<int64_t>(row) * a.Nk + j] = dp;  // the captured attention gradient
            dot += dp * prow[j];
        }
        if (!a.need_dqkv) continue;
        dot = group16_sum(dot);
        for (int j = i_a; j < NPk; j += 16) {
            float ds = 0.f;
            if (j < a.Nk) {
                ds = prow[j] * (drow[j] - dot);
                if (!q_first) ds = ds / a.scale;
            }
            drow[j] = ds;
        }
    }
    if (!a.need_dqkv) return;
    __syncthreads();

    // dQ = dS.K (x scale) | dK = dS^T.Q | dV = P^T.dO  -- one pool of independent MFMA tiles
    constexpr int ntd = DP / 16;
    const int n_dq = ntq * ntd, n_dkv = ntk * ntd;
    float* dqb = a.dq + b * a.dqs.sb + h * a.dqs.sh;
    float* dkb = a.dk + b * a.dks.sb + h * a.dks.sh;
    float* dvb = a.dv + b * a.dvs.sb + h * a.dvs.sh;
    for (int tile = wave; tile < n_dq + 2 * n_dkv; tile += NW) {
        if (tile < n_dq) {
            const int ti = tile / ntd, td = tile - ti * ntd;
            const f32x4 acc = tile_kn(dSs + ti * 16 * SS, SS, Ks + td * 16, LS, ntk, i_a, kq);
            const int d = td * 16 + i_a;
            const float mul = q_first ? a.scale : 1.f;
            if (d < a.D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ti * 16 + kq + r;
                    if (row < a.Nq) dqb[static_cast<int64_t>(row) * a.dqs.sn + d] = acc[r] * mul;
                }
            }
        } else {
            const int t2 = tile - n_dq;
            const bool is_k = t2 < n_dkv;
            const int t3 = is_k ? t2 : t2 - n_dkv;
            const int tj = t3 / ntd, td = t3 - tj * ntd;
            const f32x4 acc = is_k ? tile_nn(dSs + tj * 16, SS, Qs + td * 16, LS, ntq, i_a, kq)
                                   : tile_nn(Ps + tj * 16, SS, dOs + td * 16, LS, ntq, i_a, kq);
            const int d = td * 16 + i_a;
            float* dst = is_k ? dkb : dvb;
            const int64_t sn = is_k ? a.dks.sn : a.dvs.sn;
            if (d < a.D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = tj * 16 + kq + r;
                    if (j < a.Nk) dst[static_cast<int64_t>(j) * sn + d] = acc[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
static bool aligned16(const void* p, int64_t s0, int64_t s1, int64_t s2) {
    return ((reinterpret_cast<uintptr_t>(p) | static_cast<uintptr_t>(s0 * 4) | static_cast<uintptr_t>(s1 * 4) |
             static_cast<uintptr_t>(s2 * 4)) & 15u) == 0;
}

static size_t fwd_small_lds(int DP, int Nq, int Nk) {
    const int NPq = (Nq + 15) & ~15, NPk = (Nk + 15) & ~15;
    return sizeof(float) * (static_cast<size_t>(NPq + 2 * NPk) * (DP + 4) + static_cast<size_t>(NPq) * (NPk + 4));
}

static size_t bwd_small_lds(int DP, int Nq, int Nk) {
    const int NPq = (Nq + 15) & ~15, NPk = (Nk + 15) & ~15;
    return sizeof(float) * (static_cast<size_t>(2 * NPq + 2 * NPk) * (DP + 4) + 2 * static_cast<size_t>(NPq) * (NPk + 4));
}

static int g_attn_small = 1;  // 0 forces the tiled kernels (tests / A-B profiling); bits 8.. = phase-skip debug flags
static int g_attn_debug = 0;
void attn_small_enable(int on) { g_attn_small = on & 1; g_attn_debug = on >> 8; }

template <typename K, typename A>
static int launch_small(K kern, const A& args, int threads, size_t lds, hipStream_t s, const char* name) {
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

// returns 1 if the small kernel was launched (rc in *rc_out), 0 if the shape is not eligible
int attn_fwd_small_try(AttnFwdArgs& a, hipStream_t s, int* rc_out) {
    a.debug = g_attn_debug;
    if (!g_attn_small || a.D % 4 || a.D > 64) return 0;
    const int DP = a.D <= 32 ? 32 : 64;
    const size_t lds = fwd_small_lds(DP, a.Nq, a.Nk);
    if (lds > 160 * 1024) return 0;
    if (!aligned16(a.q, a.qs.sb, a.qs.sh, a.qs.sn) || !aligned16(a.k, a.ks.sb, a.ks.sh, a.ks.sn) ||
        !aligned16(a.v, a.vs.sb, a.vs.sh, a.vs.sn))
        return 0;
    *rc_out = DP == 32 ? launch_small(attn_fwd_small_kernel<32>, a, 512, lds, s, "attn_fwd_small_kernel<32>")
                       : launch_small(attn_fwd_small_kernel<64>, a, 512, lds, s, "attn_fwd_small_kernel<64>");
    return 1;
}

int attn_bwd_small_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out) {
    if (!g_attn_small || a.D % 4 || a.D > 64) return 0;
    const int DP = a.D <= 32 ? 32 : 64;
    const size_t lds = bwd_small_lds(DP, a.Nq, a.Nk);
    if (lds > 160 * 1024) return 0;
    if (!aligned16(a.v, a.vs.sb, a.vs.sh, a.vs.sn) || !aligned16(a.dout, a.os.sb, a.os.sh, a.os.sn)) return 0;
    if (a.need_dqkv && (!aligned16(a.q, a.qs.sb, a.qs.sh, a.qs.sn) || !aligned16(a.k, a.ks.sb, a.ks.sh, a.ks.sn)))
        return 0;
    *rc_out = DP == 32 ? launch_small(attn_bwd_small_kernel<32>, a, kBwdThreads, lds, s, "attn_bwd_small_kernel<32>")
                       : launch_small(attn_bwd_small_kernel<64>, a, kBwdThreads, lds, s, "attn_bwd_small_kernel<64>");
    return 1;
}

}  // namespace mmx
