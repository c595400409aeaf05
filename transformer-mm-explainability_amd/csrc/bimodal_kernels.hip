// Single-launch bi-modal relevancy schedule (LXMERT-style two-stream model): rules 5, 6, 7, 10, 11 + eq. 8-9.
//
// Replaces GeneratorOurs.generate_ours' rule schedule (lxmert/lxmert/src/ExplanationGenerator.py:131-211, helpers
// :18-54,61-129): 9 language + 5 vision self-attention layers, 5 cross layers with both cross directions and two
// self-attention blocks each -- 38 rule applications, every one of which is 4-8 small ATen launches in the reference
// and was 3-6 launches in our multi-launch path.  At these sizes (T <= 48 text tokens, I = 36 boxes) the problem is
// launch/latency-bound, so the whole schedule runs in ONE launch: one workgroup per sample keeps R_tt, R_ii, R_ti,
// R_it and all temporaries in LDS and streams each layer's [H, Nq, Nk] attention / gradient slabs exactly once.
// Matrices are <= 48 x 48: products run on the fp32 VALU with the k index innermost and sequential (an MFMA tile
// padded to 16 would do 2-3x the work and serialise on one wave); HBM traffic = the captured slabs once (~2 MB).
#include "mmx_common.h"

namespace mmx {

constexpr int kBmMax = 48;       // max tokens per modality
constexpr int kBmLd = kBmMax + 1;  // LDS row stride (odd: conflict-free column walks)
constexpr int kBmThreads = 256;
constexpr int kBmMaxLayers = 16;

struct BimodalArgs {
    const float* lang_a[kBmMaxLayers]; const float* lang_g[kBmMaxLayers];   // [B, H, T, T]
    const float* vis_a[kBmMaxLayers];  const float* vis_g[kBmMaxLayers];    // [B, H, I, I]
    const float* xlc_a[kBmMaxLayers];  const float* xlc_g[kBmMaxLayers];    // lang cross  [B, H, T, I]
    const float* xic_a[kBmMaxLayers];  const float* xic_g[kBmMaxLayers];    // img cross   [B, H, I, T]
    const float* xls_a[kBmMaxLayers];  const float* xls_g[kBmMaxLayers];    // lang self   [B, H, T, T]
    const float* xis_a[kBmMaxLayers];  const float* xis_g[kBmMaxLayers];    // img self    [B, H, I, I]
    int n_lang, n_vis, n_x;
    int B, H, T, I;
    unsigned flags;                 // MMX_MM_NORMALIZE | MMX_MM_SELF_IN_RULE10
    float *R_tt, *R_ti, *R_ii, *R_it;  // outputs [B,T,T] [B,T,I] [B,I,I] [B,I,T]
    float* diag_min;                // min over every handle_residual call of diag(R - I), or null
    const int* text_len;            // [B] real question length of every sample (<= T, the padded slab size), or null (= T)
};

struct Mat {      // an LDS matrix [rows][kBmLd]
    float* p;
    __device__ __forceinline__ float& at(int i, int j) const { return p[i * kBmLd + j]; }
};

// rule 5: cam[i][j] = (1/H) sum_h max(G*A, 0); heads in order, 4 (A, G) pairs in flight.  The slab of a head is
// [pq][pk] (padded sizes); only its leading nq x nk block is read (per-sample question lengths inside one padded batch).
__device__ __forceinline__ void avg_heads_lds(Mat cam, const float* A, const float* G, int64_t sample, int H, int nq,
                                              int nk, int pq, int pk, int tid) {
    const int nn = nq * nk;
    const int64_t hs = static_cast<int64_t>(pq) * pk;
    const float fH = static_cast<float>(H);
    for (int e = tid; e < nn; e += kBmThreads) {
        const int i = e / nk, j = e - i * nk;
        const int64_t off = sample + static_cast<int64_t>(i) * pk + j;
        float s = 0.f;
        int h = 0;
        for (; h + 4 <= H; h += 4) {
            float a[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = A[off + (h + u) * hs];
                g[u] = G[off + (h + u) * hs];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += relu_nan(g[u] * a[u]);
        }
        for (; h < H; ++h) s += relu_nan(G[off + h * hs] * A[off + h * hs]);
        cam.at(i, j) = s / fH;
    }
}

// out[i][j] = sum_k A[i][k] * B[k][j]   (A: M x K, B: K x N); transA: A is stored K x M
template <bool TRANS_A>
__device__ __forceinline__ void matmul_lds(Mat out, Mat A, Mat B, int M, int K, int N, int tid) {
    for (int e = tid; e < M * N; e += kBmThreads) {
        const int i = e / N, j = e - i * N;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += (TRANS_A ? A.at(k, i) : A.at(i, k)) * B.at(k, j);
        out.at(i, j) = acc;
    }
}

__device__ __forceinline__ void add_into(Mat dst, Mat src, int M, int N, int tid) {
    for (int e = tid; e < M * N; e += kBmThreads) {
        const int i = e / N, j = e - i * N;
        dst.at(i, j) = dst.at(i, j) + src.at(i, j);
    }
}

__device__ __forceinline__ void copy_mat(Mat dst, Mat src, int M, int N, int tid) {
    for (int e = tid; e < M * N; e += kBmThreads) {
        const int i = e / N, j = e - i * N;
        dst.at(i, j) = src.at(i, j);
    }
}

__device__ __forceinline__ void atomic_min_float_bm(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// eq. 8-9: out = (R - I) / rowsum(R - I) + I ; one thread per row (n <= 48)
__device__ __forceinline__ void handle_residual_lds(Mat out, Mat R, int n, float* diag_min, int tid) {
    if (tid < n) {
        const int i = tid;
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += R.at(i, j) - (j == i ? 1.f : 0.f);
        for (int j = 0; j < n; ++j) {
            const float d = (j == i) ? 1.f : 0.f;
            out.at(i, j) = (R.at(i, j) - d) / s + d;
        }
        if (diag_min) atomic_min_float_bm(diag_min, R.at(i, i) - 1.f);
    }
}

__global__ __launch_bounds__(kBmThreads) void lxmert_schedule_kernel(const BimodalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MS = kBmMax * kBmLd;
    Mat R_tt{smem}, R_ii{smem + MS}, R_ti{smem + 2 * MS}, R_it{smem + 3 * MS};
    Mat N_a{smem + 4 * MS}, N_b{smem + 5 * MS};       // normalised self matrices
    Mat cam{smem + 6 * MS}, tmp{smem + 7 * MS};
    Mat add0{smem + 8 * MS}, add1{smem + 9 * MS}, add2{smem + 10 * MS}, add3{smem + 11 * MS};

    const int tid = threadIdx.x, b = blockIdx.x;
    const int PT = a.T, I = a.I, H = a.H;                       // PT: padded question length = slab size
    const int T = a.text_len ? min(max(a.text_len[b], 1), PT) : PT;   // this sample's real question length
    const bool normalize = a.flags & MMX_MM_NORMALIZE, self10 = a.flags & MMX_MM_SELF_IN_RULE10;
    const int64_t s_tt = static_cast<int64_t>(b) * H * PT * PT, s_ii = static_cast<int64_t>(b) * H * I * I;
    const int64_t s_ti = static_cast<int64_t>(b) * H * PT * I;

    for (int e = tid; e < kBmMax * kBmMax; e += kBmThreads) {
        const int i = e / kBmMax, j = e - i * kBmMax;
        R_tt.at(i, j) = (i == j && i < T) ? 1.f : 0.f;
        R_ii.at(i, j) = (i == j && i < I) ? 1.f : 0.f;
        R_ti.at(i, j) = 0.f;
        R_it.at(i, j) = 0.f;
    }
    __syncthreads();

    // rules 6 + 7 for one self-attention block: R_ss += cam.R_ss ; R_sq += cam.R_sq  (both from the old state)
    auto self_block = [&](const float* A, const float* G, int64_t sample, Mat R_ss, Mat R_sq, int ns, int nq, int ps) {
        avg_heads_lds(cam, A, G, sample, H, ns, ns, ps, ps, tid);
        __syncthreads();
        matmul_lds<false>(add0, cam, R_ss, ns, ns, ns, tid);
        matmul_lds<false>(add1, cam, R_sq, ns, ns, nq, tid);
        __syncthreads();
        add_into(R_ss, add0, ns, ns, tid);
        add_into(R_sq, add1, ns, nq, tid);
        __syncthreads();
    };
    // rules 10 + 11 for one cross-attention (queries s, keys q):
    //   sq_add = Rn_ss^T.(cam_sq.Rn_qq) (or cam_sq) ; ss_add = cam_sq.R_qs      -- written to (o_sq, o_ss), NOT applied
    auto cross_block = [&](const float* A, const float* G, int64_t sample, Mat R_ss, Mat R_qq, Mat R_qs, int ns, int nq,
                           int ps, int pq, Mat o_sq, Mat o_ss) {
        avg_heads_lds(cam, A, G, sample, H, ns, nq, ps, pq, tid);
        if (self10 && normalize) {
            handle_residual_lds(N_a, R_ss, ns, a.diag_min, tid);
            if (tid >= 64 && tid < 64 + nq) handle_residual_lds(N_b, R_qq, nq, a.diag_min, tid - 64);
        }
        __syncthreads();
        if (self10) {
            Mat Ss = normalize ? N_a : R_ss, Qq = normalize ? N_b : R_qq;
            matmul_lds<false>(tmp, cam, Qq, ns, nq, nq, tid);
            matmul_lds<false>(o_ss, cam, R_qs, ns, nq, ns, tid);
            __syncthreads();
            matmul_lds<true>(o_sq, Ss, tmp, ns, ns, nq, tid);
        } else {
            copy_mat(o_sq, cam, ns, nq, tid);
            matmul_lds<false>(o_ss, cam, R_qs, ns, nq, ns, tid);
        }
        __syncthreads();
    };

    for (int l = 0; l < a.n_lang; ++l) self_block(a.lang_a[l], a.lang_g[l], s_tt, R_tt, R_ti, T, I, PT);
    for (int l = 0; l < a.n_vis; ++l) self_block(a.vis_a[l], a.vis_g[l], s_ii, R_ii, R_it, I, T, I);
    for (int x = 0; x < a.n_x; ++x) {
        const bool last = (x == a.n_x - 1);
        // both directions are computed from the pre-update state, then added (reference :181-189)
        cross_block(a.xlc_a[x], a.xlc_g[x], s_ti, R_tt, R_ii, R_it, T, I, PT, I, add2, add3);      // (R_ti_add, R_tt_add)
        if (!last) {
            // the second direction needs its own outputs: reuse add0/add1 (free outside self_block)
            cross_block(a.xic_a[x], a.xic_g[x], s_ti, R_ii, R_tt, R_ti, I, T, I, PT, add0, add1);  // (R_it_add, R_ii_add)
            add_into(R_it, add0, I, T, tid);
            add_into(R_ii, add1, I, I, tid);
        }
        add_into(R_ti, add2, T, I, tid);
        add_into(R_tt, add3, T, T, tid);
        __syncthreads();
        self_block(a.xls_a[x], a.xls_g[x], s_tt, R_tt, R_ti, T, I, PT);
        if (!last) self_block(a.xis_a[x], a.xis_g[x], s_ii, R_ii, R_it, I, T, I);
    }
    if (tid == 0) R_tt.at(0, 0) = 0.f;   // disregard the [CLS] token itself (:210)
    __syncthreads();

    // outputs are [PT]-padded; rows / columns beyond this sample's question length are zero
    for (int e = tid; e < PT * PT; e += kBmThreads) {
        const int i = e / PT, j = e - i * PT;
        a.R_tt[static_cast<int64_t>(b) * PT * PT + e] = (i < T && j < T) ? R_tt.at(i, j) : 0.f;
    }
    for (int e = tid; e < PT * I; e += kBmThreads) {
        const int i = e / I, j = e - i * I;
        a.R_ti[static_cast<int64_t>(b) * PT * I + e] = (i < T) ? R_ti.at(i, j) : 0.f;
    }
    if (a.R_ii)
        for (int e = tid; e < I * I; e += kBmThreads) a.R_ii[static_cast<int64_t>(b) * I * I + e] = R_ii.at(e / I, e % I);
    if (a.R_it)
        for (int e = tid; e < I * PT; e += kBmThreads) {
            const int i = e / PT, j = e - i * PT;
            a.R_it[static_cast<int64_t>(b) * I * PT + e] = (j < T) ? R_it.at(i, j) : 0.f;
        }
}

__global__ void bm_fill_scalar_kernel(float* p, float v) { *p = v; }

}  // namespace mmx

using namespace mmx;

extern "C" int mmx_lxmert_schedule_ex(const void* const* lang_attn, const void* const* lang_grad, int n_lang,
                                      const void* const* vis_attn, const void* const* vis_grad, int n_vis,
                                      const void* const* x_lang_cross_attn, const void* const* x_lang_cross_grad,
                                      const void* const* x_img_cross_attn, const void* const* x_img_cross_grad,
                                      const void* const* x_lang_self_attn, const void* const* x_lang_self_grad,
                                      const void* const* x_img_self_attn, const void* const* x_img_self_grad, int n_x,
                                      int B, int H, int T, int I, unsigned flags, const void* text_len_dev, void* R_tt_dev,
                                      void* R_ti_dev, void* R_ii_dev, void* R_it_dev, void* diag_min_dev, void* stream);

extern "C" int mmx_lxmert_schedule(const void* const* lang_attn, const void* const* lang_grad, int n_lang,
                                   const void* const* vis_attn, const void* const* vis_grad, int n_vis,
                                   const void* const* x_lang_cross_attn, const void* const* x_lang_cross_grad,
                                   const void* const* x_img_cross_attn, const void* const* x_img_cross_grad,
                                   const void* const* x_lang_self_attn, const void* const* x_lang_self_grad,
                                   const void* const* x_img_self_attn, const void* const* x_img_self_grad, int n_x,
                                   int B, int H, int T, int I, unsigned flags, void* R_tt_dev, void* R_ti_dev,
                                   void* R_ii_dev, void* R_it_dev, void* diag_min_dev, void* stream) {
    return mmx_lxmert_schedule_ex(lang_attn, lang_grad, n_lang, vis_attn, vis_grad, n_vis, x_lang_cross_attn,
                                  x_lang_cross_grad, x_img_cross_attn, x_img_cross_grad, x_lang_self_attn, x_lang_self_grad,
                                  x_img_self_attn, x_img_self_grad, n_x, B, H, T, I, flags, nullptr, R_tt_dev, R_ti_dev,
                                  R_ii_dev, R_it_dev, diag_min_dev, stream);
}

extern "C" int mmx_lxmert_schedule_ex(const void* const* lang_attn, const void* const* lang_grad, int n_lang,
                                   const void* const* vis_attn, const void* const* vis_grad, int n_vis,
                                   const void* const* x_lang_cross_attn, const void* const* x_lang_cross_grad,
                                   const void* const* x_img_cross_attn, const void* const* x_img_cross_grad,
                                   const void* const* x_lang_self_attn, const void* const* x_lang_self_grad,
                                   const void* const* x_img_self_attn, const void* const* x_img_self_grad, int n_x,
                                   int B, int H, int T, int I, unsigned flags, const void* text_len_dev, void* R_tt_dev,
                                   void* R_ti_dev, void* R_ii_dev, void* R_it_dev, void* diag_min_dev, void* stream) {
    MMX_CHECK_ARG(R_tt_dev && R_ti_dev, "mmx_lxmert_schedule: null output");
    MMX_CHECK_ARG(B > 0 && H > 0 && T > 0 && I > 0 && n_x >= 1, "mmx_lxmert_schedule: non-positive size");
    MMX_CHECK_ARG(n_lang >= 0 && n_vis >= 0 && n_lang <= kBmMaxLayers && n_vis <= kBmMaxLayers && n_x <= kBmMaxLayers,
                  "mmx_lxmert_schedule: at most %d layers per group", kBmMaxLayers);
    if (T > kBmMax || I > kBmMax) {
        set_error("mmx_lxmert_schedule: T=%d / I=%d exceed the LDS-resident limit %d (use the per-rule entry points)", T, I, kBmMax);
        return MMX_ENOTSUP;
    }
    BimodalArgs a;
    memset(&a, 0, sizeof(a));
    auto fill = [](const float** dst_a, const float** dst_g, const void* const* sa, const void* const* sg, int n) -> bool {
        for (int l = 0; l < n; ++l) {
            if (!sa || !sg || !sa[l] || !sg[l]) return false;
            dst_a[l] = static_cast<const float*>(sa[l]);
            dst_g[l] = static_cast<const float*>(sg[l]);
        }
        return true;
    };
    bool ok = fill(a.lang_a, a.lang_g, lang_attn, lang_grad, n_lang) && fill(a.vis_a, a.vis_g, vis_attn, vis_grad, n_vis) &&
              fill(a.xlc_a, a.xlc_g, x_lang_cross_attn, x_lang_cross_grad, n_x) &&
              fill(a.xls_a, a.xls_g, x_lang_self_attn, x_lang_self_grad, n_x) &&
              fill(a.xic_a, a.xic_g, x_img_cross_attn, x_img_cross_grad, n_x - 1) &&
              fill(a.xis_a, a.xis_g, x_img_self_attn, x_img_self_grad, n_x - 1);
    MMX_CHECK_ARG(ok, "mmx_lxmert_schedule: null layer pointer");
    a.n_lang = n_lang; a.n_vis = n_vis; a.n_x = n_x; a.B = B; a.H = H; a.T = T; a.I = I; a.flags = flags;
    a.R_tt = static_cast<float*>(R_tt_dev); a.R_ti = static_cast<float*>(R_ti_dev);
    a.R_ii = static_cast<float*>(R_ii_dev); a.R_it = static_cast<float*>(R_it_dev);
    a.diag_min = static_cast<float*>(diag_min_dev);
    a.text_len = static_cast<const int*>(text_len_dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.diag_min) bm_fill_scalar_kernel<<<1, 1, 0, s>>>(a.diag_min, __builtin_inff());
    const size_t lds = sizeof(float) * 12 * kBmMax * kBmLd;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lxmert_schedule_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    lxmert_schedule_kernel<<<B, kBmThreads, lds, s>>>(a);
    MMX_LAUNCH_CHECK("lxmert_schedule_kernel");
    return MMX_OK;
}
