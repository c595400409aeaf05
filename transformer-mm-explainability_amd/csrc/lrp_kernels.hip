// Fused elementwise halves of the alpha-beta LRP layer rules.
//
// The reference evaluates every `relprop` as autograd-in-autograd (`Linear.relprop`: 4 F.linear + 4 autograd.grad,
// DETR/modules/layers.py:409-437); round 3 wrote the rules in closed form with torch ops (lrp.py): 4 library GEMMs + ~35 ATen
// elementwise / reduction launches per Linear, ~45 per Add, ~12 per Clone input -- an LRP pass of the DETR head was ~7000
// launches of 2-3 us and ran 5.6x the no-LRP pass, bound by the host (profiles/r04_lrp_probe.txt).  Here each rule is
// 2 GEMMs (on K- / N-concatenated sign-split operands) plus 2-4 launches:
//   Linear   XX = [max(X,0) | min(X,0)]                                         lrp_split_signs_kernel
//            Z  = XX . [max(W,0) | min(W,0)]^T                                  (library GEMM, K = 2 in)
//            S  = safe_divide(R, Z)                                             lrp_safe_divide_kernel
//            Y  = S . [max(W,0) | min(W,0)]                                     (library GEMM, N = 2 in)
//            out = XX_p * Y_p + XX_n * Y_n  (+ block partial sums of out and R) lrp_linear_combine_kernel
//            out *= safe_divide(sum R, sum out)   (DETR flavour only)           lrp_scale_ratio_kernel
//   Add      S = safe_divide(R, a + b); ra = a S; rb = b S (+ partial sums)     lrp_add_split_kernel
//            ra *= safe_divide(|sa| / (|sa| + |sb|) * sum R, sa), likewise rb   lrp_add_scale_kernel
//   Clone    out = X * sum_i safe_divide(R_i, X)                                lrp_clone_kernel
// Sums are two-stage and deterministic (fixed grid, partials added in index order).  safe_divide is the reference's
// (DETR/modules/layers.py:11-14): a / (b + 1e-9), 0 where b == 0 (a denominator that cancels to exactly 0 becomes 1e-9).
#include "mmx_common.h"

namespace mmx {

constexpr int kLrpThreads = 256;
constexpr int kLrpMaxBlocks = 1024;

__device__ __forceinline__ float lrp_safe_divide(float a, float b) {
    // den = clamp(b, min=1e-9) + clamp(b, max=1e-9) = b + 1e-9 (one of the two clamps is the constant); den == 0 -> 1e-9
    float den = (b < 1e-9f ? 1e-9f : b) + (b > 1e-9f ? 1e-9f : b);
    den = den + (den == 0.f ? 1e-9f : 0.f);
    return a / den * (b != 0.f ? 1.f : 0.f);
}

// block-wide sum of up to 3 values, result valid on thread 0 (fixed order: lanes by xor-shuffle tree, waves in index order)
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* lds) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < kLrpThreads / 64; ++w) s += lds[w * NV + i];
            v[i] = s;
        }
    }
}

__global__ __launch_bounds__(kLrpThreads) void lrp_split_signs_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                      int64_t rows, int n) {
    const int64_t total = rows * n;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads) {
        const int64_t r = e / n;
        const int j = static_cast<int>(e - r * n);
        const float v = x[e];
        out[r * 2 * n + j] = v < 0.f ? 0.f : v;           // clamp(min=0): NaN stays NaN
        out[r * 2 * n + n + j] = v > 0.f ? 0.f : v;       // clamp(max=0)
    }
}

__global__ __launch_bounds__(kLrpThreads) void lrp_safe_divide_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                      float* __restrict__ out, int64_t n) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads)
        out[e] = lrp_safe_divide(a[e], b[e]);
}

// out[r, j] = XX[r, j] * Y[r, j] + XX[r, n + j] * Y[r, n + j]; partial[0][block] = sum out, partial[1][block] = sum R (if asked)
__global__ __launch_bounds__(kLrpThreads) void lrp_linear_combine_kernel(const float* __restrict__ XX, const float* __restrict__ Y,
                                                                         float* __restrict__ out, int64_t rows, int n,
                                                                         const float* __restrict__ R, int64_t r_numel,
                                                                         float* __restrict__ partial) {
    __shared__ float lds[2 * kLrpThreads / 64];
    const int64_t total = rows * n;
    float acc[2] = {0.f, 0.f};
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads) {
        const int64_t r = e / n;
        const int j = static_cast<int>(e - r * n);
        const int64_t base = r * 2 * n + j;
        const float v = __fadd_rn(__fmul_rn(XX[base], Y[base]), __fmul_rn(XX[base + n], Y[base + n]));   // as torch: mul, mul, add
        out[e] = v;
        acc[0] += v;
    }
    if (partial) {
        for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < r_numel;
             e += static_cast<int64_t>(gridDim.x) * kLrpThreads)
            acc[1] += R[e];
        block_sum<2>(acc, lds);
        if (threadIdx.x == 0) {
            partial[blockIdx.x] = acc[0];
            partial[gridDim.x + blockIdx.x] = acc[1];
        }
    }
}

// sum of NV interleaved-by-array partial lists (`nparts` entries each, list i at p[i * stride + j * step]) by a whole block, the
// same value in every block and on every run: lane t adds entries t, t + 256, ... in order, then the fixed-order block tree
template <int NV>
__device__ __forceinline__ void partials_sum(const float* p, int nparts, int64_t stride, int step, float (&v)[NV], float* lds,
                                             float* bcast) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (int j = threadIdx.x; j < nparts; j += kLrpThreads)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += p[i * stride + static_cast<int64_t>(j) * step];
    block_sum<NV>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) bcast[i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = bcast[i];
}

// out *= safe_divide(sum R, sum out)
__global__ __launch_bounds__(kLrpThreads) void lrp_scale_ratio_kernel(float* __restrict__ out, int64_t total,
                                                                      const float* __restrict__ partial, int nparts) {
    __shared__ float lds[2 * kLrpThreads / 64], bc[2];
    float v[2];
    partials_sum<2>(partial, nparts, nparts, 1, v, lds, bc);      // v[0] = sum out, v[1] = sum R
    const float k = lrp_safe_divide(v[1], v[0]);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads)
        out[e] *= k;
}

// Add rule, first half: per sample (blockIdx.y) S = safe_divide(R, a + b), ra = a S, rb = b S; partial[sample][block][3]
__global__ __launch_bounds__(kLrpThreads) void lrp_add_split_kernel(const float* __restrict__ R, const float* __restrict__ a,
                                                                    const float* __restrict__ b, float* __restrict__ ra,
                                                                    float* __restrict__ rb, int64_t per, float* __restrict__ partial) {
    __shared__ float lds[3 * kLrpThreads / 64];
    const int64_t off = static_cast<int64_t>(blockIdx.y) * per;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < per;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads) {
        const float r = R[off + e], av = a[off + e], bv = b[off + e];
        const float s = lrp_safe_divide(r, av + bv);
        const float x = av * s, y = bv * s;
        ra[off + e] = x;
        rb[off + e] = y;
        acc[0] += x; acc[1] += y; acc[2] += r;
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        float* p = partial + (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 3;
        p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2];
    }
}

// Add rule, second half (layers.py:207-220): ra *= safe_divide(|sa| / (|sa| + |sb|) * total, sa), rb likewise
__global__ __launch_bounds__(kLrpThreads) void lrp_add_scale_kernel(float* __restrict__ ra, float* __restrict__ rb, int64_t per,
                                                                    const float* __restrict__ partial, int nparts) {
    __shared__ float lds[3 * kLrpThreads / 64], bc[3];
    const int64_t off = static_cast<int64_t>(blockIdx.y) * per;
    float v[3];
    partials_sum<3>(partial + static_cast<int64_t>(blockIdx.y) * nparts * 3, nparts, 1, 3, v, lds, bc);   // sa, sb, total
    const float den = fabsf(v[0]) + fabsf(v[1]);
    const float fa = lrp_safe_divide(fabsf(v[0]), den) * v[2], fb = lrp_safe_divide(fabsf(v[1]), den) * v[2];
    const float xa = lrp_safe_divide(fa, v[0]), xb = lrp_safe_divide(fb, v[1]);
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < per;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads) {
        ra[off + e] *= xa;
        rb[off + e] *= xb;
    }
}

// MultiheadAttention.relprop's closing branch (DETR/modules/layers.py:791-799): when the value stream carried relevance INTO its
// projection rule (cam_v_pre not all zero) but none came out (cam_v_post all zero: decoder layer 0, value = 0), the head-level
// relevance total = sum(cam_o) is handed to the query / key streams by their shares:
//   cam_k *= safe_divide(|ks| / (|ks| + |qs|) * total, ks),  cam_q likewise.       "all zero" = min == max == 0 (a NaN is not).
// stage 1: partial[0..4][block] = #nonzero(v_pre), #nonzero(v_post), sum k, sum q, sum o;  stage 2: the in-place scaling.
struct MhaRescaleArgs {
    const float *v_pre, *v_post, *cam_o;
    float *cam_k, *cam_q;
    int64_t n_vpre, n_vpost, n_k, n_q, n_o;
    float* partial;
};
__global__ __launch_bounds__(kLrpThreads) void lrp_mha_partials_kernel(const MhaRescaleArgs a) {
    __shared__ float lds[5 * kLrpThreads / 64];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x, st = static_cast<int64_t>(gridDim.x) * kLrpThreads;
    for (int64_t e = t0; e < a.n_vpre; e += st) v[0] += (a.v_pre[e] == 0.f) ? 0.f : 1.f;
    for (int64_t e = t0; e < a.n_vpost; e += st) v[1] += (a.v_post[e] == 0.f) ? 0.f : 1.f;
    for (int64_t e = t0; e < a.n_k; e += st) v[2] += a.cam_k[e];
    for (int64_t e = t0; e < a.n_q; e += st) v[3] += a.cam_q[e];
    for (int64_t e = t0; e < a.n_o; e += st) v[4] += a.cam_o[e];
    block_sum<5>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < 5; ++i) a.partial[i * gridDim.x + blockIdx.x] = v[i];
}
__global__ __launch_bounds__(kLrpThreads) void lrp_mha_apply_kernel(const MhaRescaleArgs a, int nparts) {
    __shared__ float lds[5 * kLrpThreads / 64], bc[5];
    float v[5];
    partials_sum<5>(a.partial, nparts, nparts, 1, v, lds, bc);
    if (!(v[1] == 0.f && v[0] != 0.f)) return;                      // rescale = all_zero(v_post) & ~all_zero(v_pre)
    const float den = fabsf(v[2]) + fabsf(v[3]);
    const float kf = lrp_safe_divide(fabsf(v[2]), den) * v[4], qf = lrp_safe_divide(fabsf(v[3]), den) * v[4];
    const float xk = lrp_safe_divide(kf, v[2]), xq = lrp_safe_divide(qf, v[3]);
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x, st = static_cast<int64_t>(gridDim.x) * kLrpThreads;
    for (int64_t e = t0; e < a.n_k; e += st) a.cam_k[e] *= xk;
    for (int64_t e = t0; e < a.n_q; e += st) a.cam_q[e] *= xq;
}

// Clone rule: out = X * (safe_divide(R_0, X) + safe_divide(R_1, X) + ...)   (summed in list order, like the reference's loop)
constexpr int kLrpMaxClones = 8;
struct CloneArgs { const float* R[kLrpMaxClones]; int nr; };
__global__ __launch_bounds__(kLrpThreads) void lrp_clone_kernel(const CloneArgs a, const float* __restrict__ X,
                                                                float* __restrict__ out, int64_t n) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * kLrpThreads + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * kLrpThreads) {
        const float x = X[e];
        float c = lrp_safe_divide(a.R[0][e], x);
#pragma unroll
        for (int i = 1; i < kLrpMaxClones; ++i)
            if (i < a.nr) c += lrp_safe_divide(a.R[i][e], x);
        out[e] = x * c;
    }
}

static inline int lrp_grid(int64_t n) {
    int64_t g = (n + kLrpThreads - 1) / kLrpThreads;
    return static_cast<int>(g < 1 ? 1 : (g > kLrpMaxBlocks ? kLrpMaxBlocks : g));
}

}  // namespace mmx

using namespace mmx;

extern "C" int mmx_lrp_split_signs(const void* x_dev, void* out_dev, int64_t rows, int n, void* stream) {
    MMX_CHECK_ARG(x_dev && out_dev && rows > 0 && n > 0, "mmx_lrp_split_signs: bad argument");
    lrp_split_signs_kernel<<<lrp_grid(rows * n), kLrpThreads, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(x_dev), static_cast<float*>(out_dev), rows, n);
    MMX_LAUNCH_CHECK("lrp_split_signs_kernel");
    return MMX_OK;
}

extern "C" int mmx_lrp_safe_divide(const void* a_dev, const void* b_dev, void* out_dev, int64_t n, void* stream) {
    MMX_CHECK_ARG(a_dev && b_dev && out_dev && n > 0, "mmx_lrp_safe_divide: bad argument");
    lrp_safe_divide_kernel<<<lrp_grid(n), kLrpThreads, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(a_dev), static_cast<const float*>(b_dev), static_cast<float*>(out_dev), n);
    MMX_LAUNCH_CHECK("lrp_safe_divide_kernel");
    return MMX_OK;
}

extern "C" size_t mmx_lrp_workspace_bytes(void) { return sizeof(float) * 3 * kLrpMaxBlocks * 64; }

extern "C" int mmx_lrp_linear_combine(const void* xx_dev, const void* y_dev, void* out_dev, int64_t rows, int n,
                                      const void* r_dev, int64_t r_numel, void* workspace_dev, void* stream) {
    MMX_CHECK_ARG(xx_dev && y_dev && out_dev && rows > 0 && n > 0, "mmx_lrp_linear_combine: bad argument");
    MMX_CHECK_ARG(!r_dev || (workspace_dev && r_numel > 0), "mmx_lrp_linear_combine: normalisation needs R and a workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = lrp_grid(rows * n);
    float* partial = r_dev ? static_cast<float*>(workspace_dev) : nullptr;
    lrp_linear_combine_kernel<<<grid, kLrpThreads, 0, s>>>(static_cast<const float*>(xx_dev), static_cast<const float*>(y_dev),
                                                          static_cast<float*>(out_dev), rows, n, static_cast<const float*>(r_dev),
                                                          r_numel, partial);
    MMX_LAUNCH_CHECK("lrp_linear_combine_kernel");
    if (r_dev) {
        lrp_scale_ratio_kernel<<<grid, kLrpThreads, 0, s>>>(static_cast<float*>(out_dev), rows * n, partial, grid);
        MMX_LAUNCH_CHECK("lrp_scale_ratio_kernel");
    }
    return MMX_OK;
}

extern "C" int mmx_lrp_add_relprop(const void* r_dev, const void* a_dev, const void* b_dev, void* ra_dev, void* rb_dev,
                                   int batch, int64_t per, void* workspace_dev, void* stream) {
    MMX_CHECK_ARG(r_dev && a_dev && b_dev && ra_dev && rb_dev && workspace_dev && batch > 0 && batch <= 64 && per > 0,
                  "mmx_lrp_add_relprop: bad argument (batch <= 64)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = lrp_grid(per);
    float* partial = static_cast<float*>(workspace_dev);
    lrp_add_split_kernel<<<dim3(gx, batch), kLrpThreads, 0, s>>>(static_cast<const float*>(r_dev), static_cast<const float*>(a_dev),
                                                                static_cast<const float*>(b_dev), static_cast<float*>(ra_dev),
                                                                static_cast<float*>(rb_dev), per, partial);
    MMX_LAUNCH_CHECK("lrp_add_split_kernel");
    lrp_add_scale_kernel<<<dim3(gx, batch), kLrpThreads, 0, s>>>(static_cast<float*>(ra_dev), static_cast<float*>(rb_dev), per,
                                                                partial, gx);
    MMX_LAUNCH_CHECK("lrp_add_scale_kernel");
    return MMX_OK;
}

extern "C" int mmx_lrp_clone_relprop(const void* const* r_list, int n_r, const void* x_dev, void* out_dev, int64_t n, void* stream) {
    MMX_CHECK_ARG(r_list && n_r >= 1 && n_r <= kLrpMaxClones && x_dev && out_dev && n > 0,
                  "mmx_lrp_clone_relprop: bad argument (1..%d relevance tensors)", kLrpMaxClones);
    CloneArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n_r; ++i) {
        MMX_CHECK_ARG(r_list[i], "mmx_lrp_clone_relprop: null relevance pointer");
        a.R[i] = static_cast<const float*>(r_list[i]);
    }
    a.nr = n_r;
    lrp_clone_kernel<<<lrp_grid(n), kLrpThreads, 0, static_cast<hipStream_t>(stream)>>>(a, static_cast<const float*>(x_dev),
                                                                                         static_cast<float*>(out_dev), n);
    MMX_LAUNCH_CHECK("lrp_clone_kernel");
    return MMX_OK;
}

extern "C" int mmx_lrp_mha_rescale(const void* v_pre_dev, int64_t n_vpre, const void* v_post_dev, int64_t n_vpost, void* cam_k_dev,
                                   int64_t n_k, void* cam_q_dev, int64_t n_q, const void* cam_o_dev, int64_t n_o,
                                   void* workspace_dev, void* stream) {
    MMX_CHECK_ARG(v_pre_dev && v_post_dev && cam_k_dev && cam_q_dev && cam_o_dev && workspace_dev, "mmx_lrp_mha_rescale: null pointer");
    MMX_CHECK_ARG(n_vpre > 0 && n_vpost > 0 && n_k > 0 && n_q > 0 && n_o > 0, "mmx_lrp_mha_rescale: empty tensor");
    MhaRescaleArgs a;
    a.v_pre = static_cast<const float*>(v_pre_dev); a.v_post = static_cast<const float*>(v_post_dev);
    a.cam_o = static_cast<const float*>(cam_o_dev);
    a.cam_k = static_cast<float*>(cam_k_dev); a.cam_q = static_cast<float*>(cam_q_dev);
    a.n_vpre = n_vpre; a.n_vpost = n_vpost; a.n_k = n_k; a.n_q = n_q; a.n_o = n_o;
    a.partial = static_cast<float*>(workspace_dev);
    int64_t big = n_vpre;
    for (int64_t x : {n_vpost, n_k, n_q, n_o}) big = x > big ? x : big;
    int grid = lrp_grid(big);
    if (grid > 256) grid = 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    lrp_mha_partials_kernel<<<grid, kLrpThreads, 0, s>>>(a);
    MMX_LAUNCH_CHECK("lrp_mha_partials_kernel");
    lrp_mha_apply_kernel<<<grid, kLrpThreads, 0, s>>>(a, grid);
    MMX_LAUNCH_CHECK("lrp_mha_apply_kernel");
    return MMX_OK;
}
