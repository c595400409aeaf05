// LRP relevance through the attention core, gfx950 -- SURVEY.md section 8 row f4.
//
// Replaces the two `einsum` relprops inside the reference's MultiheadAttention.relprop (DETR/modules/layers.py:770-781;
// einsum = RelPropSimple, layers.py:54-66, both halved; Softmax / Dropout pass relevance through, layers.py:170-186), which
// the reference runs as autograd-in-autograd (re-run the einsum on saved inputs + torch.autograd.grad, twice):
//
//   S      = safe_divide(cam_O, O)                       O = P.V (the forward's output, per head)
//   cam_P  = P * (S.V^T) / 2          -> the caller's attn_cam slab (what save_attn_cam stores, layers.py:776)
//   cam_V  = V * (P^T.S) / 2
//   S1     = safe_divide(cam_P, Z)                       Z = (scale q).k^T, the pre-softmax scores
//   cam_Q  = (scale q) * (S1.k) / 2
//   cam_K  = k * (S1^T.(scale q)) / 2
//
// Same tiling as the capture op's first-generation kernels (attention_kernels.hip): the query side owns 16 query rows and
// walks the keys in 64-key tiles (cam_P, cam_Q); the key side owns 16 keys and walks the queries in 64-row tiles (cam_K,
// cam_V; it re-reads cam_P from the slab the query side wrote, so it must be launched after it on the same stream).  All
// products are exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32); Z is recomputed from q and k, never stored.
//
// Phases (mmx_attn_relprop_phase): BertSelfAttention.relprop (VisualBERT/.../BERT_ours.py:345-395) puts one more rule
// between the two matmul relprops -- Add.relprop of [scores / sqrt(d), attention_mask], whose rescale needs sums over the
// whole tensor -- so the core can be run in halves: MMX_LRP_VALUES (cam_P, cam_V from cam_O) and MMX_LRP_SCORES
// (cam_Q, cam_K from a caller-supplied relevance of the scores, `cam_scores`).  Both bits without `cam_scores` = the
// fused form above (LxmertAttention.relprop, lxmert_lrp.py:422-461, never sees its mask: its attention_mask slot stays None).
#include "mmx_common.h"
#include "attention_args.h"

namespace mmx {
namespace {

constexpr int kT16 = 16;   // rows a workgroup owns
constexpr int kT64 = 64;   // rows of the streamed operand staged per step

// DETR/modules/layers.py:11-14 in fp32
__device__ __forceinline__ float safe_divide(float a, float b) {
    float den = fmaxf(b, 1e-9f) + fminf(b, 1e-9f);
    if (den == 0.f) den += 1e-9f;
    return (b != 0.f) ? a / den : a / den * 0.f;
}

struct LrpArgs {
    const float *q, *k, *v, *o, *cam_o;
    Strides qs, ks, vs, os, cos;
    const float* probs;      // [B, H, Nq, Nk]
    float* cam_probs;        // [B, H, Nq, Nk]
    float *cam_q, *cam_k, *cam_v;
    Strides cqs, cks, cvs;
    int B, H, Nq, Nk, D;
    float scale; int scale_mode;
    const float* cam_scores; // [B, H, Nq, Nk] relevance of the pre-softmax scores, or null: cam_P itself
    int phase;               // MMX_LRP_VALUES | MMX_LRP_SCORES
};

// rows x D tile (row r at base + (row0 + r) * sn) -> LDS [rows_cap][DP + 2], zero padded, times mul
template <int DP>
__device__ __forceinline__ void stage(float* lds, const float* base, int64_t sn, int row0, int rows_valid, int rows_cap,
                                      int D, float mul, int tid) {
    constexpr int LS = DP + 2;
    for (int idx = tid; idx < rows_cap * DP; idx += 256) {
        const int r = idx / DP, d = idx - r * DP;
        float x = 0.f;
        if (r < rows_valid && d < D) x = base[static_cast<int64_t>(row0 + r) * sn + d] * mul;
        lds[r * LS + d] = x;
    }
}

// S = safe_divide(cam_O, O) for a tile of query rows
template <int DP>
__device__ __forceinline__ void stage_ratio(float* lds, const float* cam, int64_t cam_sn, const float* o, int64_t o_sn,
                                            int row0, int rows_valid, int rows_cap, int D, int tid) {
    constexpr int LS = DP + 2;
    for (int idx = tid; idx < rows_cap * DP; idx += 256) {
        const int r = idx / DP, d = idx - r * DP;
        float x = 0.f;
        if (r < rows_valid && d < D)
            x = safe_divide(cam[static_cast<int64_t>(row0 + r) * cam_sn + d], o[static_cast<int64_t>(row0 + r) * o_sn + d]);
        lds[r * LS + d] = x;
    }
}

// ------------------------------------------------------------------------------------------ query side: cam_P, cam_Q
template <int DP>
__global__ __launch_bounds__(256) void attn_lrp_q_kernel(const LrpArgs a) {
    constexpr int LS = DP + 2, TS = kT64 + 2;
    __shared__ float Qs[kT16 * LS], Ss[kT16 * LS], Ks[kT64 * LS], Vs[kT64 * LS], S1s[kT16 * TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i_a = lane & 15, kk = lane >> 4;
    const int q0 = blockIdx.x * kT16, h = blockIdx.y, b = blockIdx.z;
    const int qv = min(kT16, a.Nq - q0);
    const float qmul = a.scale_mode == MMX_SCALE_Q_FIRST ? a.scale : 1.f;   // MMX_SCALE_SCORES: Z is the raw product
    stage<DP>(Qs, a.q + b * a.qs.sb + h * a.qs.sh, a.qs.sn, q0, qv, kT16, a.D, qmul, tid);
    stage_ratio<DP>(Ss, a.cam_o + b * a.cos.sb + h * a.cos.sh, a.cos.sn, a.o + b * a.os.sb + h * a.os.sh, a.os.sn, q0, qv,
                    kT16, a.D, tid);
    const float* kb = a.k + b * a.ks.sb + h * a.ks.sh;
    const float* vb = a.v + b * a.vs.sb + h * a.vs.sh;
    const int64_t slab = (static_cast<int64_t>(b) * a.H + h) * a.Nq * a.Nk;
    const bool values = a.phase & MMX_LRP_VALUES, scores = a.phase & MMX_LRP_SCORES;
    f32x4 acc_q = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < a.Nk; kt += kT64) {
        __syncthreads();                                                  // previous tile's readers are done
        const int kv = min(kT64, a.Nk - kt);
        stage<DP>(Ks, kb, a.ks.sn, kt, kv, kT64, a.D, 1.f, tid);
        stage<DP>(Vs, vb, a.vs.sn, kt, kv, kT64, a.D, 1.f, tid);
        __syncthreads();
        // wave w: keys 16w .. 16w+15 of the tile.  z[row][key] and dp[row][key], row = 4 kk + r, key = 16 w + i_a
        f32x4 z = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DP / 4; ++ks) {
            z = mfma16x16x4(Qs[i_a * LS + 4 * ks + kk], Ks[(wave * 16 + i_a) * LS + 4 * ks + kk], z);
            dp = mfma16x16x4(Ss[i_a * LS + 4 * ks + kk], Vs[(wave * 16 + i_a) * LS + 4 * ks + kk], dp);
        }
        const int key = kt + wave * 16 + i_a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            float s1 = 0.f;
            if (row < qv && key < a.Nk) {
                const int64_t idx = slab + static_cast<int64_t>(q0 + row) * a.Nk + key;
                float camp = 0.f;
                if (values) {
                    camp = a.probs[idx] * dp[r] * 0.5f;
                    a.cam_probs[idx] = camp;
                }
                if (scores) s1 = safe_divide(a.cam_scores ? a.cam_scores[idx] : camp, z[r]);
            }
            S1s[row * TS + wave * 16 + i_a] = s1;
        }
        __syncthreads();
        // C_q[row][d] += sum_key S1[row][key] k[key][d]; wave w owns d = 16w .. 16w+15
        if (scores && wave * 16 < DP) {
#pragma unroll
            for (int ks = 0; ks < kT64 / 4; ++ks)
                acc_q = mfma16x16x4(S1s[i_a * TS + 4 * ks + kk], Ks[(4 * ks + kk) * LS + wave * 16 + i_a], acc_q);
        }
    }
    const int d = wave * 16 + i_a;
    if (scores && d < a.D) {
        float* out = a.cam_q + b * a.cqs.sb + h * a.cqs.sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            if (row < qv) out[static_cast<int64_t>(q0 + row) * a.cqs.sn + d] = Qs[row * LS + d] * acc_q[r] * 0.5f;
        }
    }
}

// ------------------------------------------------------------------------------------------ key side: cam_K, cam_V
template <int DP>
__global__ __launch_bounds__(256) void attn_lrp_kv_kernel(const LrpArgs a) {
    constexpr int LS = DP + 2, TS = kT64 + 2;
    __shared__ float Ks[kT16 * LS], Vs[kT16 * LS], Qs[kT64 * LS], Ss[kT64 * LS], S1t[kT16 * TS], Pt[kT16 * TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i_a = lane & 15, kk = lane >> 4;
    const int k0 = blockIdx.x * kT16, h = blockIdx.y, b = blockIdx.z;
    const int kv = min(kT16, a.Nk - k0);
    const float qmul = a.scale_mode == MMX_SCALE_Q_FIRST ? a.scale : 1.f;   // MMX_SCALE_SCORES: Z is the raw product
    stage<DP>(Ks, a.k + b * a.ks.sb + h * a.ks.sh, a.ks.sn, k0, kv, kT16, a.D, 1.f, tid);
    stage<DP>(Vs, a.v + b * a.vs.sb + h * a.vs.sh, a.vs.sn, k0, kv, kT16, a.D, 1.f, tid);
    const float* qb = a.q + b * a.qs.sb + h * a.qs.sh;
    const float* ob = a.o + b * a.os.sb + h * a.os.sh;
    const float* cb = a.cam_o + b * a.cos.sb + h * a.cos.sh;
    const int64_t slab = (static_cast<int64_t>(b) * a.H + h) * a.Nq * a.Nk;
    const bool values = a.phase & MMX_LRP_VALUES, scores = a.phase & MMX_LRP_SCORES;
    const float* cam_s = a.cam_scores ? a.cam_scores : a.cam_probs;        // fused form: what the query side just wrote
    f32x4 acc_k = {0.f, 0.f, 0.f, 0.f}, acc_v = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < a.Nq; qt += kT64) {
        __syncthreads();
        const int qv = min(kT64, a.Nq - qt);
        stage<DP>(Qs, qb, a.qs.sn, qt, qv, kT64, a.D, qmul, tid);
        stage_ratio<DP>(Ss, cb, a.cos.sn, ob, a.os.sn, qt, qv, kT64, a.D, tid);
        __syncthreads();
        // wave w: queries 16w .. 16w+15 of the tile.  z^T[key][query]: key = 4 kk + r, query = 16 w + i_a
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DP / 4; ++ks)
            z = mfma16x16x4(Ks[i_a * LS + 4 * ks + kk], Qs[(wave * 16 + i_a) * LS + 4 * ks + kk], z);
        const int query = qt + wave * 16 + i_a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kk * 4 + r;
            float s1 = 0.f, p = 0.f;
            if (key < kv && query < a.Nq) {
                const int64_t idx = slab + static_cast<int64_t>(query) * a.Nk + k0 + key;
                if (values) p = a.probs[idx];
                if (scores) s1 = safe_divide(cam_s[idx], z[r]);
            }
            S1t[key * TS + wave * 16 + i_a] = s1;
            Pt[key * TS + wave * 16 + i_a] = p;
        }
        __syncthreads();
        // C_k[key][d] += sum_query S1[query][key] q[query][d];  C_v[key][d] += sum_query P[query][key] S[query][d]
        if (wave * 16 < DP) {
#pragma unroll
            for (int ks = 0; ks < kT64 / 4; ++ks) {
                acc_k = mfma16x16x4(S1t[i_a * TS + 4 * ks + kk], Qs[(4 * ks + kk) * LS + wave * 16 + i_a], acc_k);
                acc_v = mfma16x16x4(Pt[i_a * TS + 4 * ks + kk], Ss[(4 * ks + kk) * LS + wave * 16 + i_a], acc_v);
            }
        }
    }
    const int d = wave * 16 + i_a;
    if (d < a.D) {
        float* outk = a.cam_k + b * a.cks.sb + h * a.cks.sh;
        float* outv = a.cam_v + b * a.cvs.sb + h * a.cvs.sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kk * 4 + r;
            if (key < kv) {
                if (scores) outk[static_cast<int64_t>(k0 + key) * a.cks.sn + d] = Ks[key * LS + d] * acc_k[r] * 0.5f;
                if (values) outv[static_cast<int64_t>(k0 + key) * a.cvs.sn + d] = Vs[key * LS + d] * acc_v[r] * 0.5f;
            }
        }
    }
}

template <int DP>
int launch(const LrpArgs& a, hipStream_t s) {
    dim3 gq((a.Nq + kT16 - 1) / kT16, a.H, a.B), gk((a.Nk + kT16 - 1) / kT16, a.H, a.B);
    hipLaunchKernelGGL(attn_lrp_q_kernel<DP>, gq, dim3(256), 0, s, a);
    MMX_LAUNCH_CHECK("attn_lrp_q_kernel");
    hipLaunchKernelGGL(attn_lrp_kv_kernel<DP>, gk, dim3(256), 0, s, a);
    MMX_LAUNCH_CHECK("attn_lrp_kv_kernel");
    return MMX_OK;
}

}  // namespace
}  // namespace mmx

using namespace mmx;

extern "C" int mmx_attn_relprop_phase(const void* q_dev, const void* k_dev, const void* v_dev, const void* o_dev,
                                      const void* cam_o_dev,
                                      int64_t q_sb, int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                      int64_t v_sb, int64_t v_sh, int64_t v_sn, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                                      int64_t co_sb, int64_t co_sh, int64_t co_sn,
                                      const void* probs_dev, void* cam_probs_dev, void* cam_q_dev, void* cam_k_dev, void* cam_v_dev,
                                      int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                                      int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                                      int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                                      const void* cam_scores_dev, int phase, void* stream) {
    const bool values = phase & MMX_LRP_VALUES, scores = phase & MMX_LRP_SCORES;
    MMX_CHECK_ARG((values || scores) && !(phase & ~(MMX_LRP_VALUES | MMX_LRP_SCORES)), "mmx_attn_relprop: bad phase %d", phase);
    MMX_CHECK_ARG(q_dev && k_dev, "mmx_attn_relprop: null q / k");
    MMX_CHECK_ARG(!values || (v_dev && o_dev && cam_o_dev && probs_dev && cam_probs_dev && cam_v_dev),
                  "mmx_attn_relprop: the values phase needs v, o, cam_o, probs, cam_probs, cam_v");
    MMX_CHECK_ARG(!scores || (cam_q_dev && cam_k_dev && (values || cam_scores_dev)),
                  "mmx_attn_relprop: the scores phase needs cam_q, cam_k and (without the values phase) cam_scores");
    MMX_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0 && D > 0, "mmx_attn_relprop: bad sizes B=%d H=%d Nq=%d Nk=%d D=%d", B, H,
                  Nq, Nk, D);
    MMX_CHECK_ARG(scale_mode == MMX_SCALE_Q_FIRST || scale_mode == MMX_SCALE_SCORES, "mmx_attn_relprop: bad scale_mode %d",
                  scale_mode);
    MMX_CHECK_ARG(H <= 65535 && B <= 65535, "mmx_attn_relprop: H / B beyond the grid limits");
    if (D > 64) {
        set_error("mmx_attn_relprop: head_dim %d > 64 is not supported", D);
        return MMX_ENOTSUP;
    }
    LrpArgs a;
    // (a phase that does not run never dereferences its operands; q stands in so that the staging loops stay uniform)
    a.q = static_cast<const float*>(q_dev); a.k = static_cast<const float*>(k_dev);
    a.v = static_cast<const float*>(values ? v_dev : k_dev);
    a.o = static_cast<const float*>(values ? o_dev : q_dev); a.cam_o = static_cast<const float*>(values ? cam_o_dev : q_dev);
    a.qs = {q_sb, q_sh, q_sn}; a.ks = {k_sb, k_sh, k_sn};
    a.vs = values ? Strides{v_sb, v_sh, v_sn} : a.ks;
    a.os = values ? Strides{o_sb, o_sh, o_sn} : a.qs;
    a.cos = values ? Strides{co_sb, co_sh, co_sn} : a.qs;
    a.probs = static_cast<const float*>(probs_dev); a.cam_probs = static_cast<float*>(cam_probs_dev);
    a.cam_q = static_cast<float*>(cam_q_dev); a.cam_k = static_cast<float*>(cam_k_dev); a.cam_v = static_cast<float*>(cam_v_dev);
    a.cqs = {cq_sb, cq_sh, cq_sn}; a.cks = {ck_sb, ck_sh, ck_sn}; a.cvs = {cv_sb, cv_sh, cv_sn};
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale; a.scale_mode = scale_mode;
    a.cam_scores = static_cast<const float*>(cam_scores_dev); a.phase = phase;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return D <= 32 ? launch<32>(a, s) : launch<64>(a, s);
}

extern "C" int mmx_attn_relprop(const void* q_dev, const void* k_dev, const void* v_dev, const void* o_dev, const void* cam_o_dev,
                                int64_t q_sb, int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                int64_t v_sb, int64_t v_sh, int64_t v_sn, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                                int64_t co_sb, int64_t co_sh, int64_t co_sn,
                                const void* probs_dev, void* cam_probs_dev, void* cam_q_dev, void* cam_k_dev, void* cam_v_dev,
                                int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                                int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                                int B, int H, int Nq, int Nk, int D, float scale, int scale_mode, void* stream) {
    return mmx_attn_relprop_phase(q_dev, k_dev, v_dev, o_dev, cam_o_dev, q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn,
                                  o_sb, o_sh, o_sn, co_sb, co_sh, co_sn, probs_dev, cam_probs_dev, cam_q_dev, cam_k_dev, cam_v_dev,
                                  cq_sb, cq_sh, cq_sn, ck_sb, ck_sh, ck_sn, cv_sb, cv_sh, cv_sn, B, H, Nq, Nk, D, scale,
                                  scale_mode, nullptr, MMX_LRP_VALUES | MMX_LRP_SCORES, stream);
}
