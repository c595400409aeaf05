// Attention-capture op for gfx950: the softmax probabilities P and their gradient dP are the PRODUCT
// here (they feed the relevancy rules), so unlike flash attention both are materialised -- straight into
// the caller's preallocated capture slabs, replacing the reference's save_attn / save_attn_gradients
// Python hooks (sites cited in include/mmx_relevancy.h).
//
// Tiling: one workgroup (4 waves) per (batch, head, 16-query tile) for forward and the dQ half of
// backward, per (batch, head, 16-key tile) for the dK/dV half.  All matrix products run on the exact-fp32
// MFMA v_mfma_f32_16x16x4_f32; the full score row block [16 x Nk] lives in LDS so softmax is one pass.
#include "mmx_common.h"
#include "attention_args.h"

namespace mmx {

constexpr int kTQ = 16;   // query rows per workgroup
constexpr int kTK = 64;   // keys staged per step

__device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// stage `rows` x D floats (row r at base + r*sn) into LDS tile [rows_cap][DP+2], zero padded, times `mul`
template <int DP>
__device__ __forceinline__ void stage_tile(float* lds, const float* base, int64_t sn, int row0, int rows_valid,
                                           int rows_cap, int D, float mul, int tid, int nthreads) {
    constexpr int LS = DP + 2;
    for (int idx = tid; idx < rows_cap * DP; idx += nthreads) {
        const int r = idx / DP, d = idx - r * DP;
        float v = 0.f;
        if (r < rows_valid && d < D) v = base[static_cast<int64_t>(row0 + r) * sn + d] * mul;
        lds[r * LS + d] = v;
    }
}

// ---------------------------------------------------------------------------------------------- forward
template <int DP>
__global__ __launch_bounds__(256) void attn_capture_fwd_kernel(const AttnFwdArgs a) {
    constexpr int LS = DP + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NKP = round_up(a.Nk, kTK);
    const int SS = NKP + 2;
    float* Qs = smem;                    // [16][LS]
    float* KVs = Qs + kTQ * LS;          // [64][LS]
    float* Ss = KVs + kTK * LS;          // [16][SS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * kTQ, h = blockIdx.y, b = blockIdx.z;
    const int qv = min(kTQ, a.Nq - q0);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);

    stage_tile<DP>(Qs, qb, a.qs.sn, q0, qv, kTQ, a.D, q_first ? a.scale : 1.f, tid, 256);

    // ---- phase 1: S = Q.K^T (+ mask) into the LDS row block
    const int i_a = lane & 15, kk = lane >> 4;
    for (int kt = 0; kt < NKP; kt += kTK) {
        __syncthreads();
        stage_tile<DP>(KVs, kb, a.ks.sn, kt, min(kTK, a.Nk - kt), kTK, a.D, 1.f, tid, 256);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DP / 4; ++ks)
            acc = mfma16x16x4(Qs[i_a * LS + 4 * ks + kk], KVs[(wave * 16 + i_a) * LS + 4 * ks + kk], acc);
        const int key = kt + wave * 16 + i_a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            float s = acc[r];
            if (!q_first) s = s / a.scale;  // scale = sqrt(d) divisor in MMX_SCALE_SCORES mode
            if (a.mask && key < a.Nk && row < qv)
                s += a.mask[b * a.mask_sb + static_cast<int64_t>(q0 + row) * a.mask_sq + key];
            Ss[row * SS + key] = s;
        }
    }
    __syncthreads();

    // ---- phase 2: row softmax (wave w owns rows 4w..4w+3), P -> LDS and -> the capture slab
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        if (row >= qv) break;
        float* srow = Ss + row * SS;
        float m = -__builtin_inff();
        for (int j = lane; j < a.Nk; j += 64) m = fmaxf(m, srow[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float sum = 0.f;
        for (int j = lane; j < a.Nk; j += 64) {
            const float e = expf(srow[j] - m);
            srow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        float* prow = a.probs + ((static_cast<int64_t>(b) * a.H + h) * a.Nq + q0 + row) * a.Nk;
        for (int j = lane; j < NKP; j += 64) {
            float p = 0.f;
            if (j < a.Nk) {
                p = srow[j] / sum;
                prow[j] = p;
            }
            srow[j] = p;  // zero the key padding so phase 3 can run over whole 64-key tiles
        }
    }
    // rows >= qv of Ss hold finite junk only if they were written: zero them for the MFMA below
    for (int idx = tid; idx < (kTQ - qv) * SS; idx += 256) Ss[qv * SS + idx] = 0.f;

    // ---- phase 3: O = P.V ; wave w owns output columns 16w..16w+15
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < NKP; kt += kTK) {
        __syncthreads();
        stage_tile<DP>(KVs, vb, a.vs.sn, kt, min(kTK, a.Nk - kt), kTK, a.D, 1.f, tid, 256);
        __syncthreads();
        if (wave * 16 < DP) {
#pragma unroll
            for (int ks = 0; ks < kTK / 4; ++ks)
                acc = mfma16x16x4(Ss[i_a * SS + kt + 4 * ks + kk], KVs[(4 * ks + kk) * LS + wave * 16 + i_a], acc);
        }
    }
    const int d = wave * 16 + i_a;
    if (d < a.D) {
        float* ob = a.o + b * a.os.sb + h * a.os.sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            if (row < qv) ob[static_cast<int64_t>(q0 + row) * a.os.sn + d] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
// Kernel A, per 16-query tile: dP = dO.V^T -> capture slab; delta; dS = P*(dP - delta); dQ = dS.K
template <int DP>
__global__ __launch_bounds__(256) void attn_capture_bwd_q_kernel(const AttnBwdArgs a) {
    constexpr int LS = DP + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NKP = round_up(a.Nk, kTK);
    const int SS = NKP + 2;
    float* dOs = smem;                   // [16][LS]
    float* KVs = dOs + kTQ * LS;         // [64][LS]
    float* dPs = KVs + kTK * LS;         // [16][SS]  dP then dS

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * kTQ, h = blockIdx.y, b = blockIdx.z;
    const int qv = min(kTQ, a.Nq - q0);
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const float* dob = a.dout + b * a.os.sb + h * a.os.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int i_a = lane & 15, kk = lane >> 4;

    stage_tile<DP>(dOs, dob, a.os.sn, q0, qv, kTQ, a.D, 1.f, tid, 256);

    for (int kt = 0; kt < NKP; kt += kTK) {
        __syncthreads();
        stage_tile<DP>(KVs, vb, a.vs.sn, kt, min(kTK, a.Nk - kt), kTK, a.D, 1.f, tid, 256);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DP / 4; ++ks)
            acc = mfma16x16x4(dOs[i_a * LS + 4 * ks + kk], KVs[(wave * 16 + i_a) * LS + 4 * ks + kk], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) dPs[(kk * 4 + r) * SS + kt + wave * 16 + i_a] = acc[r];
    }
    __syncthreads();

    const int64_t head = static_cast<int64_t>(b) * a.H + h;
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        float* drow = dPs + row * SS;
        if (row >= qv) {
            for (int j = lane; j < NKP; j += 64) drow[j] = 0.f;
            continue;
        }
        const int64_t goff = (head * a.Nq + q0 + row) * a.Nk;
        const float* prow_g = a.probs + b * a.probs_sb + (static_cast<int64_t>(h) * a.Nq + q0 + row) * a.Nk;
        float dot = 0.f;
        for (int j = lane; j < a.Nk; j += 64) {
            const float dp = drow[j];
            a.dprobs[goff + j] = dp;  // the captured attention gradient
            dot += dp * prow_g[j];
        }
        if (!a.need_dqkv) continue;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        if (lane == 0) a.delta[head * a.Nq + q0 + row] = dot;
        for (int j = lane; j < NKP; j += 64) {
            float ds = 0.f;
            if (j < a.Nk) {
                ds = prow_g[j] * (drow[j] - dot);
                if (!q_first) ds = ds / a.scale;
            }
            drow[j] = ds;
        }
    }
    if (!a.need_dqkv) return;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < NKP; kt += kTK) {
        __syncthreads();
        stage_tile<DP>(KVs, kb, a.ks.sn, kt, min(kTK, a.Nk - kt), kTK, a.D, 1.f, tid, 256);
        __syncthreads();
        if (wave * 16 < DP) {
#pragma unroll
            for (int ks = 0; ks < kTK / 4; ++ks)
                acc = mfma16x16x4(dPs[i_a * SS + kt + 4 * ks + kk], KVs[(4 * ks + kk) * LS + wave * 16 + i_a], acc);
        }
    }
    const int d = wave * 16 + i_a;
    if (d < a.D) {
        float* dqb = a.dq + b * a.dqs.sb + h * a.dqs.sh;
        const float mul = q_first ? a.scale : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            if (row < qv) dqb[static_cast<int64_t>(q0 + row) * a.dqs.sn + d] = acc[r] * mul;
        }
    }
}

// Kernel B, per 16-key tile: dV = P^T.dO ; dK = dS^T.Q  (dS recomputed from the two capture slabs + delta)
template <int DP>
__global__ __launch_bounds__(256) void attn_capture_bwd_kv_kernel(const AttnBwdArgs a) {
    constexpr int LS = DP + 2;
    constexpr int PS = 18;
    __shared__ float Qs[kTK * LS];
    __shared__ float dOs[kTK * LS];
    __shared__ float Ps[kTK * PS];
    __shared__ float dSs[kTK * PS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int jv = min(16, a.Nk - j0);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* dob = a.dout + b * a.os.sb + h * a.os.sh;
    const bool q_first = (a.scale_mode == MMX_SCALE_Q_FIRST);
    const int i_a = lane & 15, kk = lane >> 4;
    const int64_t head = static_cast<int64_t>(b) * a.H + h;

    f32x4 accK = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < a.Nq; qt += kTK) {
        const int rows = min(kTK, a.Nq - qt);
        __syncthreads();
        stage_tile<DP>(Qs, qb, a.qs.sn, qt, rows, kTK, a.D, q_first ? a.scale : 1.f, tid, 256);
        stage_tile<DP>(dOs, dob, a.os.sn, qt, rows, kTK, a.D, 1.f, tid, 256);
        for (int idx = tid; idx < kTK * 16; idx += 256) {
            const int r = idx >> 4, j = idx & 15;
            float p = 0.f, ds = 0.f;
            if (r < rows && j < jv) {
                const int64_t g = (head * a.Nq + qt + r) * a.Nk + j0 + j;
                p = a.probs[b * a.probs_sb + (static_cast<int64_t>(h) * a.Nq + qt + r) * a.Nk + j0 + j];
                ds = p * (a.dprobs[g] - a.delta[head * a.Nq + qt + r]);
                if (!q_first) ds = ds / a.scale;
            }
            Ps[r * PS + j] = p;
            dSs[r * PS + j] = ds;
        }
        __syncthreads();
        if (wave * 16 < DP) {
#pragma unroll
            for (int ks = 0; ks < kTK / 4; ++ks) {
                const int r = 4 * ks + kk;
                accV = mfma16x16x4(Ps[r * PS + i_a], dOs[r * LS + wave * 16 + i_a], accV);
                accK = mfma16x16x4(dSs[r * PS + i_a], Qs[r * LS + wave * 16 + i_a], accK);
            }
        }
    }
    const int d = wave * 16 + i_a;
    if (d < a.D) {
        float* dkb = a.dk + b * a.dks.sb + h * a.dks.sh;
        float* dvb = a.dv + b * a.dvs.sb + h * a.dvs.sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = kk * 4 + r;
            if (j < jv) {
                dkb[static_cast<int64_t>(j0 + j) * a.dks.sn + d] = accK[r];
                dvb[static_cast<int64_t>(j0 + j) * a.dvs.sn + d] = accV[r];
            }
        }
    }
}

static size_t attn_lds_bytes(int DP, int Nk) {
    const int NKP = (Nk + kTK - 1) / kTK * kTK;
    return sizeof(float) * (static_cast<size_t>(kTQ + kTK) * (DP + 2) + static_cast<size_t>(kTQ) * (NKP + 2));
}

template <typename K, typename A>
static int launch_dyn(K kern, const A& args, dim3 grid, size_t lds, hipStream_t s, const char* name) {
    if (lds > 160 * 1024) {
        set_error("%s: Nk too large for the LDS row block (%zu bytes > 160 KiB)", name, lds);
        return MMX_ENOTSUP;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    kern<<<grid, 256, lds, s>>>(args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    return MMX_OK;
}

}  // namespace mmx

using namespace mmx;

static int check_attn_dims(const char* fn, int B, int H, int Nq, int Nk, int D, int scale_mode) {
    MMX_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0 && D > 0, "%s: non-positive size", fn);
    MMX_CHECK_ARG(B <= 65535 && H <= 65535, "%s: B/H exceed the grid limit 65535", fn);
    MMX_CHECK_ARG(scale_mode == MMX_SCALE_Q_FIRST || scale_mode == MMX_SCALE_SCORES, "%s: scale_mode %d", fn, scale_mode);
    if (D > 64) {
        set_error("%s: head_dim %d > 64 not supported", fn, D);
        return MMX_ENOTSUP;
    }
    return MMX_OK;
}

extern "C" int mmx_attn_capture_fwd(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb, int64_t q_sh,
                                    int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn, int64_t v_sb, int64_t v_sh,
                                    int64_t v_sn, const void* mask_dev, int64_t mask_sb, int64_t mask_sq,
                                    void* probs_dev, void* o_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B,
                                    int H, int Nq, int Nk, int D, float scale, int scale_mode, void* stream) {
    return mmx_attn_capture_fwd_ex(q_dev, k_dev, v_dev, q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, mask_dev,
                                   mask_sb, mask_sq, probs_dev, MMX_F32, o_dev, o_sb, o_sh, o_sn, B, H, Nq, Nk, D, scale,
                                   scale_mode, stream);
}

extern "C" int mmx_attn_capture_fwd_ex(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb,
                                       int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn, int64_t v_sb,
                                       int64_t v_sh, int64_t v_sn, const void* mask_dev, int64_t mask_sb, int64_t mask_sq,
                                       void* probs_dev, int slab_dtype, void* o_dev, int64_t o_sb, int64_t o_sh,
                                       int64_t o_sn, int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                                       void* stream) {
    MMX_CHECK_ARG(q_dev && k_dev && v_dev && probs_dev && o_dev, "mmx_attn_capture_fwd: null pointer");
    const int mma_bf16 = (slab_dtype & MMX_ATTN_MMA_BF16) ? 1 : 0;
    slab_dtype &= ~MMX_ATTN_MMA_BF16;
    MMX_CHECK_ARG(slab_dtype == MMX_F32 || slab_dtype == MMX_F16 || slab_dtype == MMX_BF16,
                  "mmx_attn_capture_fwd: slab dtype %d", slab_dtype);
    int rc = check_attn_dims("mmx_attn_capture_fwd", B, H, Nq, Nk, D, scale_mode);
    if (rc) return rc;
    AttnFwdArgs a;
    a.q = static_cast<const float*>(q_dev); a.k = static_cast<const float*>(k_dev); a.v = static_cast<const float*>(v_dev);
    a.qs = {q_sb, q_sh, q_sn}; a.ks = {k_sb, k_sh, k_sn}; a.vs = {v_sb, v_sh, v_sn};
    a.mask = static_cast<const float*>(mask_dev); a.mask_sb = mask_sb; a.mask_sq = mask_sq;
    a.probs = static_cast<float*>(probs_dev); a.o = static_cast<float*>(o_dev); a.os = {o_sb, o_sh, o_sn};
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale; a.scale_mode = scale_mode; a.debug = 0;
    a.slab_dt = slab_dtype;
    a.mma_bf16 = mma_bf16;
    dim3 grid((Nq + kTQ - 1) / kTQ, H, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (slab_dtype != MMX_F32 || mma_bf16) {   // half-precision slabs / bf16 MFMA: the streaming kernels only
        if (attn_fwd_stream_try(a, s, &rc)) return rc;
        set_error("mmx_attn_capture_fwd: fp16 / bf16 capture slabs and MMX_ATTN_MMA_BF16 need head_dim %% 4 == 0 "
                  "(<= 64) and 16-byte aligned q/k/v views");
        return MMX_ENOTSUP;
    }
    if (attn_fwd_head_try(a, s, &rc)) return rc;    // short sequences: a wave owns 16 query rows, scores in registers
    if (attn_fwd_stream_try(a, s, &rc)) return rc;  // long sequences: K/V streamed, nothing of size Nk on chip
    if (D <= 32) return launch_dyn(attn_capture_fwd_kernel<32>, a, grid, attn_lds_bytes(32, Nk), s, "attn_capture_fwd_kernel<32>");
    return launch_dyn(attn_capture_fwd_kernel<64>, a, grid, attn_lds_bytes(64, Nk), s, "attn_capture_fwd_kernel<64>");
}

extern "C" size_t mmx_attn_capture_bwd_workspace_bytes(int B, int H, int Nq) {
    return sizeof(float) * static_cast<size_t>(B) * H * Nq;
}

extern "C" int mmx_attn_capture_bwd(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb, int64_t q_sh,
                                    int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn, int64_t v_sb, int64_t v_sh,
                                    int64_t v_sn, const void* probs_dev, int64_t probs_sb, const void* do_dev, int64_t o_sb, int64_t o_sh,
                                    int64_t o_sn, void* dprobs_dev, void* dq_dev, void* dk_dev, void* dv_dev,
                                    int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, int64_t dk_sb, int64_t dk_sh,
                                    int64_t dk_sn, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn, int B, int H, int Nq,
                                    int Nk, int D, float scale, int scale_mode, int need_dqkv, void* workspace_dev,
                                    size_t workspace_bytes, void* stream) {
    return mmx_attn_capture_bwd_ex(q_dev, k_dev, v_dev, q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, probs_dev,
                                   probs_sb, MMX_F32, do_dev, o_sb, o_sh, o_sn, nullptr, 0, 0, 0, dprobs_dev, dq_dev, dk_dev, dv_dev, dq_sb,
                                   dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn, B, H, Nq, Nk, D, scale,
                                   scale_mode, need_dqkv, workspace_dev, workspace_bytes, stream);
}

namespace mmx {
namespace {
// v_out[b][k] = v_in[b][k] + (1 / H) sum_j part[b][j][k]   (row-relevancy mode: the rows j are (head, query tile))
__global__ __launch_bounds__(256) void rel_row_update_kernel(const float* __restrict__ v_in, const float* __restrict__ part,
                                                             float* __restrict__ v_out, int J, int N, float inv_h) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int64_t b = blockIdx.y;
    const float* p = part + b * J * N + k;
    float sum = 0.f;
    for (int j = 0; j < J; ++j) sum += p[static_cast<int64_t>(j) * N];
    v_out[b * N + k] = v_in[b * N + k] + sum * inv_h;
}
}  // namespace
}  // namespace mmx

namespace mmx {
int rel_row_update(const float* v_in, const float* part, float* out, int B, int J, int N, float inv_h, hipStream_t s) {
    rel_row_update_kernel<<<dim3((N + 255) / 256, B), 256, 0, s>>>(v_in, part, out, J, N, inv_h);
    MMX_LAUNCH_CHECK("rel_row_update_kernel");
    return MMX_OK;
}
}  // namespace mmx

static size_t rowrel_delta_bytes(int B, int H, int Nq) {
    return (sizeof(float) * static_cast<size_t>(B) * H * Nq + 255) / 256 * 256;
}

static size_t rowrel_part_bytes(int B, int H, int Nq, int Nk) {
    const size_t nrt = (static_cast<size_t>(Nq) + 63) / 64;
    return (sizeof(float) * static_cast<size_t>(B) * H * nrt * Nk + 255) / 256 * 256;
}

extern "C" size_t mmx_attn_capture_bwd_rowrel_workspace_bytes(int B, int H, int Nq, int Nk) {
    // delta | partial relevancy rows | bf16 images of the shared operands (third-generation kernels, attention_bf16_v3.hip)
    return rowrel_delta_bytes(B, H, Nq) + rowrel_part_bytes(B, H, Nq, Nk) + mmx::attn_bwd_bf16_v3_prep_bytes(H, Nk);
}

static int attn_bwd_impl(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb,
                                       int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn, int64_t v_sb,
                                       int64_t v_sh, int64_t v_sn, const void* probs_dev, int64_t probs_sb, int slab_dtype,
                                       const void* do_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn, const void* fwd_o_dev,
                                       int64_t fo_sb, int64_t fo_sh, int64_t fo_sn, void* dprobs_dev,
                                       void* dq_dev, void* dk_dev, void* dv_dev, int64_t dq_sb, int64_t dq_sh,
                                       int64_t dq_sn, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn, int64_t dv_sb,
                                       int64_t dv_sh, int64_t dv_sn, int B, int H, int Nq, int Nk, int D, float scale,
                                       int scale_mode, int need_dqkv, void* workspace_dev, size_t workspace_bytes,
                                       void* stream, const void* rel_in_dev, void* rel_out_dev) {
    const bool rel = rel_in_dev != nullptr;
    MMX_CHECK_ARG(v_dev && probs_dev && do_dev && (dprobs_dev || rel), "mmx_attn_capture_bwd: null pointer");
    const int io_bf16 = (slab_dtype & MMX_ATTN_IO_BF16) ? 1 : 0;
    // (MMX_ATTN_IO_BF16 without MMX_ATTN_MMA_BF16: exact-fp32 arithmetic on a bf16 gradient stream -- the whole-head kernels only)
    if (rel) {
        MMX_CHECK_ARG(rel_out_dev && Nq == Nk, "mmx_attn_capture_bwd_rowrel: needs rel_out and self-attention (Nq == Nk)");
        MMX_CHECK_ARG(slab_dtype & MMX_ATTN_MMA_BF16, "mmx_attn_capture_bwd_rowrel: MMX_ATTN_MMA_BF16 kernels only");
        if (!workspace_dev || workspace_bytes < mmx_attn_capture_bwd_rowrel_workspace_bytes(B, H, Nq, Nk)) {
            set_error("mmx_attn_capture_bwd_rowrel: workspace %zu < %zu", workspace_bytes,
                      mmx_attn_capture_bwd_rowrel_workspace_bytes(B, H, Nq, Nk));
            return MMX_EWORKSPACE;
        }
    }
    const int mma_bf16 = (slab_dtype & MMX_ATTN_MMA_BF16) ? 1 : 0;
    slab_dtype &= ~(MMX_ATTN_MMA_BF16 | MMX_ATTN_IO_BF16);
    MMX_CHECK_ARG(slab_dtype == MMX_F32 || slab_dtype == MMX_F16 || slab_dtype == MMX_BF16,
                  "mmx_attn_capture_bwd: slab dtype %d", slab_dtype);
    int rc = check_attn_dims("mmx_attn_capture_bwd", B, H, Nq, Nk, D, scale_mode);
    if (rc) return rc;
    if (need_dqkv) {
        MMX_CHECK_ARG(q_dev && k_dev && dq_dev && dk_dev && dv_dev, "mmx_attn_capture_bwd: null q/k/dq/dk/dv");
        if (!workspace_dev || workspace_bytes < mmx_attn_capture_bwd_workspace_bytes(B, H, Nq)) {
            set_error("mmx_attn_capture_bwd: workspace %zu < %zu", workspace_bytes,
                      mmx_attn_capture_bwd_workspace_bytes(B, H, Nq));
            return MMX_EWORKSPACE;
        }
    }
    AttnBwdArgs a;
    a.q = static_cast<const float*>(q_dev); a.k = static_cast<const float*>(k_dev); a.v = static_cast<const float*>(v_dev);
    a.qs = {q_sb, q_sh, q_sn}; a.ks = {k_sb, k_sh, k_sn}; a.vs = {v_sb, v_sh, v_sn};
    a.probs = static_cast<const float*>(probs_dev); a.probs_sb = probs_sb;
    a.dout = static_cast<const float*>(do_dev); a.os = {o_sb, o_sh, o_sn};
    a.o = static_cast<const float*>(fwd_o_dev); a.oos = {fo_sb, fo_sh, fo_sn};
    if (reinterpret_cast<uintptr_t>(fwd_o_dev) % 16 || fo_sb % 4 || fo_sh % 4 || fo_sn % 4) a.o = nullptr;   // a hint only
    a.dprobs = static_cast<float*>(dprobs_dev);
    a.dq = static_cast<float*>(dq_dev); a.dk = static_cast<float*>(dk_dev); a.dv = static_cast<float*>(dv_dev);
    a.dqs = {dq_sb, dq_sh, dq_sn}; a.dks = {dk_sb, dk_sh, dk_sn}; a.dvs = {dv_sb, dv_sh, dv_sn};
    a.delta = static_cast<float*>(workspace_dev);
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale; a.scale_mode = scale_mode; a.need_dqkv = need_dqkv;
    a.slab_dt = slab_dtype;
    a.mma_bf16 = mma_bf16;
    a.io_bf16 = io_bf16;
    a.rel_v = static_cast<const float*>(rel_in_dev);
    a.rel_part = rel ? reinterpret_cast<float*>(static_cast<char*>(workspace_dev) + rowrel_delta_bytes(B, H, Nq)) : nullptr;
    a.rel_out = static_cast<float*>(rel_out_dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rel) {
        const size_t used = rowrel_delta_bytes(B, H, Nq) + rowrel_part_bytes(B, H, Nq, Nk);
        if (attn_bwd_bf16_v3_try(a, static_cast<char*>(workspace_dev) + used, workspace_bytes - used, s, &rc)) return rc;
        if (!attn_bwd_stream_try(a, s, &rc)) {
            set_error("mmx_attn_capture_bwd_rowrel: needs head_dim %% 4 == 0 (<= 64) and 16-byte aligned views");
            return MMX_ENOTSUP;
        }
        return rc;      // (the path that ran has also launched rel_row_update over the partial rows it made)
    }
    if (slab_dtype != MMX_F32 || mma_bf16) {
        if (attn_bwd_stream_try(a, s, &rc)) return rc;
        set_error("mmx_attn_capture_bwd: fp16 / bf16 capture slabs and MMX_ATTN_MMA_BF16 need head_dim %% 4 == 0 "
                  "(<= 64) and 16-byte aligned views");
        return MMX_ENOTSUP;
    }
    if (attn_bwd_head_try(a, s, &rc)) return rc;    // short sequences: a wave owns 16 query rows, scores in registers
    if (io_bf16) {
        set_error("mmx_attn_capture_bwd: MMX_ATTN_IO_BF16 without MMX_ATTN_MMA_BF16 is served by the whole-head kernels only "
                  "(fp32 slabs, Nk <= 128, Nq <= 256, head_dim %% 4 == 0 and <= 64, 8-byte aligned gradient rows)");
        return MMX_ENOTSUP;
    }
    if (attn_bwd_stream_try(a, s, &rc)) return rc;  // long sequences
    dim3 gq((Nq + kTQ - 1) / kTQ, H, B), gk((Nk + 15) / 16, H, B);
    if (D <= 32) {
        rc = launch_dyn(attn_capture_bwd_q_kernel<32>, a, gq, attn_lds_bytes(32, Nk), s, "attn_capture_bwd_q_kernel<32>");
        if (rc || !need_dqkv) return rc;
        attn_capture_bwd_kv_kernel<32><<<gk, 256, 0, s>>>(a);
    } else {
        rc = launch_dyn(attn_capture_bwd_q_kernel<64>, a, gq, attn_lds_bytes(64, Nk), s, "attn_capture_bwd_q_kernel<64>");
        if (rc || !need_dqkv) return rc;
        attn_capture_bwd_kv_kernel<64><<<gk, 256, 0, s>>>(a);
    }
    MMX_LAUNCH_CHECK("attn_capture_bwd_kv_kernel");
    return MMX_OK;
}

extern "C" int mmx_attn_capture_bwd_ex(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb,
                                       int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn, int64_t v_sb,
                                       int64_t v_sh, int64_t v_sn, const void* probs_dev, int64_t probs_sb, int slab_dtype,
                                       const void* do_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn, const void* fwd_o_dev,
                                       int64_t fo_sb, int64_t fo_sh, int64_t fo_sn, void* dprobs_dev,
                                       void* dq_dev, void* dk_dev, void* dv_dev, int64_t dq_sb, int64_t dq_sh,
                                       int64_t dq_sn, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn, int64_t dv_sb,
                                       int64_t dv_sh, int64_t dv_sn, int B, int H, int Nq, int Nk, int D, float scale,
                                       int scale_mode, int need_dqkv, void* workspace_dev, size_t workspace_bytes,
                                       void* stream) {
    return attn_bwd_impl(q_dev, k_dev, v_dev, q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, probs_dev, probs_sb,
                         slab_dtype, do_dev, o_sb, o_sh, o_sn, fwd_o_dev, fo_sb, fo_sh, fo_sn, dprobs_dev, dq_dev, dk_dev,
                         dv_dev, dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn, B, H, Nq, Nk, D, scale,
                         scale_mode, need_dqkv, workspace_dev, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int mmx_attn_capture_bwd_rowrel(const void* q_dev, const void* k_dev, const void* v_dev, int64_t q_sb,
                                           int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                           int64_t v_sb, int64_t v_sh, int64_t v_sn, const void* probs_dev,
                                           int64_t probs_sb, int slab_dtype, const void* do_dev, int64_t o_sb,
                                           int64_t o_sh, int64_t o_sn, const void* fwd_o_dev, int64_t fo_sb,
                                           int64_t fo_sh, int64_t fo_sn, void* dprobs_dev, void* dq_dev, void* dk_dev,
                                           void* dv_dev, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, int64_t dk_sb,
                                           int64_t dk_sh, int64_t dk_sn, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                                           int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                                           int need_dqkv, const void* rel_in_dev, void* rel_out_dev,
                                           void* workspace_dev, size_t workspace_bytes, void* stream) {
    MMX_CHECK_ARG(rel_in_dev && rel_out_dev, "mmx_attn_capture_bwd_rowrel: null relevancy row");
    return attn_bwd_impl(q_dev, k_dev, v_dev, q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, probs_dev, probs_sb,
                         slab_dtype, do_dev, o_sb, o_sh, o_sn, fwd_o_dev, fo_sb, fo_sh, fo_sn, dprobs_dev, dq_dev, dk_dev,
                         dv_dev, dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn, B, H, Nq, Nk, D, scale,
                         scale_mode, need_dqkv, workspace_dev, workspace_bytes, stream, rel_in_dev, rel_out_dev);
}
