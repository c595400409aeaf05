// Fused QuickGELU (x * sigmoid(1.702 x)) forward / backward for the CLIP body's MLP (CLIP/clip/model.py:162-164).
// PyTorch runs the activation as 3 elementwise kernels forward and 5 backward over a [B*N, 4E] fp32 tensor
// (40 MB per layer at batch 64); on an HBM-bound op that is 7x the necessary traffic.  One pass each here:
// forward reads x, writes y; backward reads x and dy, writes dx (sigmoid recomputed, nothing saved but x).
#include "mmx_common.h"

namespace mmx {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pack(float a, float b) {        // round to nearest even (v_cvt_pk_bf16_f32)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r;
    r[0] = static_cast<__bf16>(a);
    r[1] = static_cast<__bf16>(b);
    return __builtin_bit_cast(unsigned, r);
}


__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + expf(-z)); }

__global__ __launch_bounds__(256) void quick_gelu_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n4,
                                                             const float* xt, float* yt, int tail) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
        const f32x4 v = x[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * sigmoid_f(1.702f * v[e]);
        y[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < tail) yt[threadIdx.x] = xt[threadIdx.x] * sigmoid_f(1.702f * xt[threadIdx.x]);
}

// the same with a bf16 result: the activation only feeds the next GEMM of a bf16 body (no conversion pass)
__global__ __launch_bounds__(256) void quick_gelu_fwd_bf16_kernel(const f32x4* __restrict__ x, u32x2* __restrict__ y, int64_t n4) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
        const f32x4 v = x[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * sigmoid_f(1.702f * v[e]);
        y[i] = u32x2{bf_pack(o[0], o[1]), bf_pack(o[2], o[3])};
    }
}

// x_n4: number of 16-byte groups of x; x_n4 < n4 broadcasts x over the leading (batch) dimension of dy -- the shared-forward
// backward has ONE activation tensor for B upstream gradients.
__global__ __launch_bounds__(256) void quick_gelu_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy,
                                                             f32x4* __restrict__ dx, int64_t n4, int64_t x_n4, const float* xt,
                                                             const float* dyt, float* dxt, int tail) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
        const f32x4 v = x[x_n4 == n4 ? i : i % x_n4];
        const f32x4 g = dy[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = sigmoid_f(1.702f * v[e]);
            o[e] = g[e] * (s + 1.702f * v[e] * s * (1.f - s));
        }
        dx[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < tail) {
        const float v = xt[threadIdx.x], s = sigmoid_f(1.702f * v);
        dxt[threadIdx.x] = dyt[threadIdx.x] * (s + 1.702f * v * s * (1.f - s));
    }
}

}  // namespace mmx

using namespace mmx;

static int gelu_grid(int64_t n4) {
    const int64_t blocks = (n4 + 255) / 256;
    return static_cast<int>(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

extern "C" int mmx_quick_gelu_fwd(const void* x_dev, void* y_dev, int64_t n, void* stream) {
    MMX_CHECK_ARG(x_dev && y_dev && n > 0, "mmx_quick_gelu_fwd: bad argument");
    MMX_CHECK_ARG(((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(y_dev)) & 15u) == 0,
                  "mmx_quick_gelu_fwd: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const float* x = static_cast<const float*>(x_dev);
    float* y = static_cast<float*>(y_dev);
    quick_gelu_fwd_kernel<<<gelu_grid(n4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), n4, x + n4 * 4, y + n4 * 4, static_cast<int>(n - n4 * 4));
    MMX_LAUNCH_CHECK("quick_gelu_fwd_kernel");
    return MMX_OK;
}

extern "C" int mmx_quick_gelu_bwd_bcast(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, int64_t x_n,
                                        void* stream) {
    MMX_CHECK_ARG(x_dev && dy_dev && dx_dev && n > 0 && x_n > 0, "mmx_quick_gelu_bwd_bcast: bad argument");
    MMX_CHECK_ARG(n % x_n == 0 && x_n % 4 == 0, "mmx_quick_gelu_bwd_bcast: n=%ld must be a multiple of x_n=%ld, x_n %% 4 == 0",
                  static_cast<long>(n), static_cast<long>(x_n));
    MMX_CHECK_ARG(((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(dy_dev) |
                    reinterpret_cast<uintptr_t>(dx_dev)) & 15u) == 0, "mmx_quick_gelu_bwd_bcast: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    quick_gelu_bwd_kernel<<<gelu_grid(n4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const f32x4*>(x_dev), static_cast<const f32x4*>(dy_dev), static_cast<f32x4*>(dx_dev), n4, x_n / 4, nullptr,
        nullptr, nullptr, 0);
    MMX_LAUNCH_CHECK("quick_gelu_bwd_kernel");
    return MMX_OK;
}

extern "C" int mmx_quick_gelu_bwd(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, void* stream) {
    MMX_CHECK_ARG(x_dev && dy_dev && dx_dev && n > 0, "mmx_quick_gelu_bwd: bad argument");
    MMX_CHECK_ARG(((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(dy_dev) |
                    reinterpret_cast<uintptr_t>(dx_dev)) & 15u) == 0, "mmx_quick_gelu_bwd: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const float* x = static_cast<const float*>(x_dev);
    const float* dy = static_cast<const float*>(dy_dev);
    float* dx = static_cast<float*>(dx_dev);
    quick_gelu_bwd_kernel<<<gelu_grid(n4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(dy), reinterpret_cast<f32x4*>(dx), n4, n4, x + n4 * 4,
        dy + n4 * 4, dx + n4 * 4, static_cast<int>(n - n4 * 4));
    MMX_LAUNCH_CHECK("quick_gelu_bwd_kernel");
    return MMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm input-gradient fused with the residual add, with the forward statistics SHARED across the batch:
//     dx[r] = d_res[r] + rstd[m] * (g - mean(g) - xhat[m] * mean(g * xhat[m])),   g = dy[r] * gamma,  m = r % x_rows
// Used by the hand-written batched backward of the shared image tower (clip_model.Transformer.backward_shared): the
// forward ran once (x_rows = N rows of x / mean / rstd), the backward has B*N rows.  ATen's native_layer_norm_backward
// wants the input replicated per sample (an expand().contiguous() copy per call) and leaves the residual add to a
// second elementwise kernel.  One wave per row, 16-B accesses; E % 4 == 0.
// ---------------------------------------------------------------------------------------------------------------------
namespace mmx {

__global__ __launch_bounds__(256) void layernorm_bwd_add_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* d_res,
                                                                float* __restrict__ dx, int64_t rows, int x_rows, int E) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int m = static_cast<int>(r % x_rows);
    const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + r * E);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<int64_t>(m) * E);
    const f32x4* gm = reinterpret_cast<const f32x4*>(gamma);
    const float mu = mean[m], rs = rstd[m];
    const int n4 = E >> 2;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4 g = dyr[i] * gm[i];
        const f32x4 xh = (xr[i] - mu) * rs;
        a += g[0] + g[1] + g[2] + g[3];
        b += g[0] * xh[0] + g[1] * xh[1] + g[2] * xh[2] + g[3] * xh[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    a /= static_cast<float>(E);
    b /= static_cast<float>(E);
    const f32x4* dr = d_res ? reinterpret_cast<const f32x4*>(d_res + r * E) : nullptr;
    f32x4* out = reinterpret_cast<f32x4*>(dx + r * E);
    for (int i = lane; i < n4; i += 64) {
        const f32x4 g = dyr[i] * gm[i];
        const f32x4 xh = (xr[i] - mu) * rs;
        f32x4 o = (g - a - xh * b) * rs;
        if (dr) o = o + dr[i];
        out[i] = o;
    }
}

}  // namespace mmx

extern "C" int mmx_layernorm_bwd_add(const void* dy_dev, const void* x_dev, const void* mean_dev, const void* rstd_dev,
                                     const void* gamma_dev, const void* d_res_dev, void* dx_dev, int64_t rows, int x_rows,
                                     int E, void* stream) {
    MMX_CHECK_ARG(dy_dev && x_dev && mean_dev && rstd_dev && gamma_dev && dx_dev, "mmx_layernorm_bwd_add: null pointer");
    MMX_CHECK_ARG(rows > 0 && x_rows > 0 && E > 0 && E % 4 == 0, "mmx_layernorm_bwd_add: rows=%ld x_rows=%d E=%d (E %% 4 must be 0)",
                  static_cast<long>(rows), x_rows, E);
    mmx::layernorm_bwd_add_kernel<<<static_cast<unsigned>((rows + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(dy_dev), static_cast<const float*>(x_dev), static_cast<const float*>(mean_dev),
        static_cast<const float*>(rstd_dev), static_cast<const float*>(gamma_dev), static_cast<const float*>(d_res_dev),
        static_cast<float*>(dx_dev), rows, x_rows, E);
    MMX_LAUNCH_CHECK("layernorm_bwd_add_kernel");
    return MMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Residual add fused with the LayerNorm that follows it (pre-LN blocks: x1 = x + attn_out; h = LN(x1), CLIP/clip/model.py:
// 195-197):  sum = x + y (written out: it is the next residual), h = (sum - mean) * rstd * gamma + beta, plus the row
// statistics the hand-written backward needs.  y == NULL: plain LayerNorm of x with statistics.  ATen runs the add and the
// LayerNorm as two kernels (3 reads + 2 writes of the [rows, E] activation); this is 2 reads + 2 writes in one pass.  One
// wave per row, the row lives in registers (E <= 64 * 4 * NV); two-pass mean / variance like ATen's row-wise moments.
// ---------------------------------------------------------------------------------------------------------------------
namespace mmx {

template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ sum_out, float* __restrict__ h_out,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                int64_t rows, int E, float eps, unsigned short* __restrict__ h_bf16) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, n4 = E >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + r * E);
    const f32x4* yr = y ? reinterpret_cast<const f32x4*>(y + r * E) : nullptr;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            v[j] = xr[i];
            if (yr) v[j] = v[j] + yr[i];
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / static_cast<float>(E);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (lane + 64 * j < n4) {
            const f32x4 d = v[j] - mu;
            q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rs = rsqrtf(q / static_cast<float>(E) + eps);
    if (lane == 0) {
        mean_out[r] = mu;
        rstd_out[r] = rs;
    }
    const f32x4* gm = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* bt = reinterpret_cast<const f32x4*>(beta);
    f32x4* so = (sum_out && yr) ? reinterpret_cast<f32x4*>(sum_out + r * E) : nullptr;
    f32x4* ho = h_bf16 ? nullptr : reinterpret_cast<f32x4*>(h_out + r * E);
    u32x2* hb = h_bf16 ? reinterpret_cast<u32x2*>(h_bf16 + r * E) : nullptr;   // bf16 body: h only feeds the next GEMM
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < n4) {
            if (so) so[i] = v[j];
            const f32x4 h = (v[j] - mu) * rs * gm[i] + bt[i];
            if (hb) hb[i] = u32x2{bf_pack(h[0], h[1]), bf_pack(h[2], h[3])};
            else ho[i] = h;
        }
    }
}

}  // namespace mmx

extern "C" int mmx_add_layernorm_fwd_ex(const void* x_dev, const void* y_dev, const void* gamma_dev, const void* beta_dev,
                                        void* sum_dev, void* h_dev, void* mean_dev, void* rstd_dev, int64_t rows, int E,
                                        float eps, int h_dtype, void* stream) {
    MMX_CHECK_ARG(x_dev && gamma_dev && beta_dev && h_dev && mean_dev && rstd_dev, "mmx_add_layernorm_fwd: null pointer");
    MMX_CHECK_ARG(!y_dev || sum_dev, "mmx_add_layernorm_fwd: the sum x + y needs an output buffer");
    MMX_CHECK_ARG(rows > 0 && E > 0 && E % 4 == 0 && E <= 4096, "mmx_add_layernorm_fwd: rows=%ld E=%d (E %% 4 == 0, E <= 4096)",
                  static_cast<long>(rows), E);
    MMX_CHECK_ARG(h_dtype == MMX_F32 || h_dtype == MMX_BF16, "mmx_add_layernorm_fwd: h is fp32 or bf16");
    const unsigned grid = static_cast<unsigned>((rows + 3) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *x = static_cast<const float*>(x_dev), *y = static_cast<const float*>(y_dev);
    const float *g = static_cast<const float*>(gamma_dev), *b = static_cast<const float*>(beta_dev);
    float *so = static_cast<float*>(sum_dev), *ho = h_dtype == MMX_F32 ? static_cast<float*>(h_dev) : nullptr;
    unsigned short* hb = h_dtype == MMX_BF16 ? static_cast<unsigned short*>(h_dev) : nullptr;
    float *mo = static_cast<float*>(mean_dev), *ro = static_cast<float*>(rstd_dev);
    if (E <= 256) mmx::add_layernorm_fwd_kernel<1><<<grid, 256, 0, s>>>(x, y, g, b, so, ho, mo, ro, rows, E, eps, hb);
    else if (E <= 512) mmx::add_layernorm_fwd_kernel<2><<<grid, 256, 0, s>>>(x, y, g, b, so, ho, mo, ro, rows, E, eps, hb);
    else if (E <= 1024) mmx::add_layernorm_fwd_kernel<4><<<grid, 256, 0, s>>>(x, y, g, b, so, ho, mo, ro, rows, E, eps, hb);
    else if (E <= 2048) mmx::add_layernorm_fwd_kernel<8><<<grid, 256, 0, s>>>(x, y, g, b, so, ho, mo, ro, rows, E, eps, hb);
    else mmx::add_layernorm_fwd_kernel<16><<<grid, 256, 0, s>>>(x, y, g, b, so, ho, mo, ro, rows, E, eps, hb);
    MMX_LAUNCH_CHECK("add_layernorm_fwd_kernel");
    return MMX_OK;
}

extern "C" int mmx_add_layernorm_fwd(const void* x_dev, const void* y_dev, const void* gamma_dev, const void* beta_dev,
                                     void* sum_dev, void* h_dev, void* mean_dev, void* rstd_dev, int64_t rows, int E,
                                     float eps, void* stream) {
    return mmx_add_layernorm_fwd_ex(x_dev, y_dev, gamma_dev, beta_dev, sum_dev, h_dev, mean_dev, rstd_dev, rows, E, eps, MMX_F32,
                                    stream);
}

extern "C" int mmx_quick_gelu_fwd_bf16(const void* x_dev, void* y_dev, int64_t n, void* stream) {
    MMX_CHECK_ARG(x_dev && y_dev && n > 0 && n % 4 == 0, "mmx_quick_gelu_fwd_bf16: bad argument (n %% 4 == 0)");
    MMX_CHECK_ARG((reinterpret_cast<uintptr_t>(x_dev) & 15u) == 0 && (reinterpret_cast<uintptr_t>(y_dev) & 7u) == 0,
                  "mmx_quick_gelu_fwd_bf16: x must be 16-byte, y 8-byte aligned");
    mmx::quick_gelu_fwd_bf16_kernel<<<gelu_grid(n / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const mmx::f32x4*>(x_dev), static_cast<mmx::u32x2*>(y_dev), n / 4);
    MMX_LAUNCH_CHECK("quick_gelu_fwd_bf16_kernel");
    return MMX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 gradient stream of a bf16 body (BASELINE config 5): the upstream gradients between the backward's GEMMs are bf16
// (what the bf16 GEMMs produce and consume -- no conversion passes), the residual gradient stream stays fp32.
//   * quick_gelu_bwd_bcast_bf16:  dx(bf16) = dy(bf16) * QuickGELU'(x),  x fp32 broadcast over the batch;
//   * layernorm_bwd_add_bf16:     dx = d_res + LN'(dy), dy bf16, written as fp32 (the next residual) AND bf16 (the
//                                 operand of the next GEMM).
// ---------------------------------------------------------------------------------------------------------------------
namespace mmx {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__global__ __launch_bounds__(256) void quick_gelu_bwd_bf16_kernel(const f32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                                  u32x4* __restrict__ dx, int64_t n8, int64_t x_n8) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n8; i += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t xi = (x_n8 == n8 ? i : i % x_n8) * 2;
        const f32x4 v0 = x[xi], v1 = x[xi + 1];
        const u32x4 g = dy[i];
        float gv[8], xv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) { gv[2 * e] = bf_lo(g[e]); gv[2 * e + 1] = bf_hi(g[e]); }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = sigmoid_f(1.702f * xv[e]);
            o[e] = gv[e] * (sg + 1.702f * xv[e] * sg * (1.f - sg));
        }
        dx[i] = u32x4{bf_pack(o[0], o[1]), bf_pack(o[2], o[3]), bf_pack(o[4], o[5]), bf_pack(o[6], o[7])};
    }
}

// Shared-forward mode (x is ONE sample's pre-activation, dy carries `batch` upstream gradients): a thread keeps its 8 columns,
// forms QuickGELU'(x) for them once and sweeps `per` samples with it -- no transcendental and no index arithmetic per element,
// the pass is the bf16 read + write of dy / dx.
__global__ __launch_bounds__(256) void quick_gelu_bwd_bf16_sweep_kernel(const f32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                                        u32x4* __restrict__ dx, int64_t x_n8, int batch, int per) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (p >= x_n8) return;
    const f32x4 v0 = x[2 * p], v1 = x[2 * p + 1];
    const float xv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float sg = sigmoid_f(1.702f * xv[e]);
        d[e] = sg + 1.702f * xv[e] * sg * (1.f - sg);
    }
    const int b0 = blockIdx.y * per, b1 = min(batch, b0 + per);
    const u32x4* src = dy + static_cast<int64_t>(b0) * x_n8 + p;
    u32x4* dst = dx + static_cast<int64_t>(b0) * x_n8 + p;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        u32x4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = __builtin_nontemporal_load(src + u * x_n8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf_pack(bf_lo(g[u][e]) * d[2 * e], bf_hi(g[u][e]) * d[2 * e + 1]);
            __builtin_nontemporal_store(o, dst + u * x_n8);
        }
        src += 4 * x_n8;
        dst += 4 * x_n8;
    }
    for (; b < b1; ++b) {
        const u32x4 g = *src;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bf_pack(bf_lo(g[e]) * d[2 * e], bf_hi(g[e]) * d[2 * e + 1]);
        *dst = o;
        src += x_n8;
        dst += x_n8;
    }
}

// E = 256 * ITER: the row lives in registers between the two passes (one round of independent loads per wave instead of two
// dependent sweeps); same summation order as the generic kernel below.
template <int ITER>
__global__ __launch_bounds__(256) void layernorm_bwd_add_bf16_row_kernel(const unsigned short* __restrict__ dy,
                                                                         const float* __restrict__ x, const float* __restrict__ mean,
                                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                         const float* d_res, float* __restrict__ dx,
                                                                         unsigned short* __restrict__ dx_h, int64_t rows, int x_rows) {
    constexpr int E = 256 * ITER;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int m = static_cast<int>(r % x_rows);
    const u32x2* dyr = reinterpret_cast<const u32x2*>(dy + r * E);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<int64_t>(m) * E);
    const f32x4* gm = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* dr = d_res ? reinterpret_cast<const f32x4*>(d_res + r * E) : nullptr;
    u32x2 raw[ITER];
    f32x4 xh[ITER], res[ITER], g[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) raw[k] = __builtin_nontemporal_load(dyr + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < ITER; ++k) xh[k] = xr[lane + 64 * k];
    if (dr) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) res[k] = __builtin_nontemporal_load(dr + lane + 64 * k);
    }
    const float mu = mean[m], rs = rstd[m];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        g[k] = f32x4{bf_lo(raw[k][0]), bf_hi(raw[k][0]), bf_lo(raw[k][1]), bf_hi(raw[k][1])} * gm[lane + 64 * k];
        xh[k] = (xh[k] - mu) * rs;
        a += g[k][0] + g[k][1] + g[k][2] + g[k][3];
        b += g[k][0] * xh[k][0] + g[k][1] * xh[k][1] + g[k][2] * xh[k][2] + g[k][3] * xh[k][3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    a /= static_cast<float>(E);
    b /= static_cast<float>(E);
    f32x4* out = dx ? reinterpret_cast<f32x4*>(dx + r * E) : nullptr;
    u32x2* out_h = dx_h ? reinterpret_cast<u32x2*>(dx_h + r * E) : nullptr;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        f32x4 o = (g[k] - a - xh[k] * b) * rs;
        if (dr) o = o + res[k];
        if (out) out[lane + 64 * k] = o;
        if (out_h) out_h[lane + 64 * k] = u32x2{bf_pack(o[0], o[1]), bf_pack(o[2], o[3])};
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_add_bf16_kernel(const unsigned short* __restrict__ dy,
                                                                     const float* __restrict__ x, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                     const float* d_res, float* __restrict__ dx,
                                                                     unsigned short* __restrict__ dx_h, int64_t rows, int x_rows,
                                                                     int E) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int m = static_cast<int>(r % x_rows);
    const u32x2* dyr = reinterpret_cast<const u32x2*>(dy + r * E);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + static_cast<int64_t>(m) * E);
    const f32x4* gm = reinterpret_cast<const f32x4*>(gamma);
    const float mu = mean[m], rs = rstd[m];
    const int n4 = E >> 2;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const u32x2 raw = dyr[i];
        const f32x4 g = f32x4{bf_lo(raw[0]), bf_hi(raw[0]), bf_lo(raw[1]), bf_hi(raw[1])} * gm[i];
        const f32x4 xh = (xr[i] - mu) * rs;
        a += g[0] + g[1] + g[2] + g[3];
        b += g[0] * xh[0] + g[1] * xh[1] + g[2] * xh[2] + g[3] * xh[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    a /= static_cast<float>(E);
    b /= static_cast<float>(E);
    const f32x4* dr = d_res ? reinterpret_cast<const f32x4*>(d_res + r * E) : nullptr;
    f32x4* out = dx ? reinterpret_cast<f32x4*>(dx + r * E) : nullptr;
    u32x2* out_h = dx_h ? reinterpret_cast<u32x2*>(dx_h + r * E) : nullptr;
    for (int i = lane; i < n4; i += 64) {
        const u32x2 raw = dyr[i];
        const f32x4 g = f32x4{bf_lo(raw[0]), bf_hi(raw[0]), bf_lo(raw[1]), bf_hi(raw[1])} * gm[i];
        const f32x4 xh = (xr[i] - mu) * rs;
        f32x4 o = (g - a - xh * b) * rs;
        if (dr) o = o + dr[i];
        if (out) out[i] = o;
        if (out_h) out_h[i] = u32x2{bf_pack(o[0], o[1]), bf_pack(o[2], o[3])};
    }
}

// One row per sample scattered into a zeroed dense [B, N, E] tensor / added to an existing one: the top block of a CLIP tower
// carries a gradient on ONE token per sample (class / EOT token); torch's zeros + index_put pair is two launches of 19 + 27 us.
__global__ __launch_bounds__(256) void rows_to_dense_kernel(const f32x4* __restrict__ vals, const long long* __restrict__ rows,
                                                            f32x4* __restrict__ out, int N, int E4) {
    const int b = blockIdx.y;
    const long long row = rows[b];
    const int64_t per = static_cast<int64_t>(N) * E4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < per; i += static_cast<int64_t>(gridDim.x) * 256) {
        const int n = static_cast<int>(i / E4), e = static_cast<int>(i - static_cast<int64_t>(n) * E4);
        out[b * per + i] = n == row ? vals[static_cast<int64_t>(b) * E4 + e] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__global__ __launch_bounds__(256) void rows_add_kernel(f32x4* __restrict__ dense, const long long* __restrict__ rows,
                                                       const f32x4* __restrict__ vals, int N, int E4) {
    const int b = blockIdx.y;
    const long long row = rows[b];
    if (row < 0 || row >= N) return;                 // an index outside the sequence writes nothing (never out of bounds)
    f32x4* dst = dense + (static_cast<int64_t>(b) * N + row) * E4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < E4; e += gridDim.x * 256) dst[e] = dst[e] + vals[static_cast<int64_t>(b) * E4 + e];
}

}  // namespace mmx

extern "C" int mmx_rows_to_dense(const void* vals_dev, const void* rows_dev, void* out_dev, int B, int N, int E, void* stream) {
    MMX_CHECK_ARG(vals_dev && rows_dev && out_dev && B > 0 && N > 0 && E > 0 && E % 4 == 0 && B <= 65535,
                  "mmx_rows_to_dense: bad argument (B=%d N=%d E=%d, E %% 4 must be 0)", B, N, E);
    const int64_t per = static_cast<int64_t>(N) * (E / 4);
    const dim3 grid(static_cast<unsigned>(per / 256 + 1 > 64 ? 64 : per / 256 + 1), B);
    mmx::rows_to_dense_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const f32x4*>(vals_dev), static_cast<const long long*>(rows_dev), static_cast<f32x4*>(out_dev), N, E / 4);
    MMX_LAUNCH_CHECK("rows_to_dense_kernel");
    return MMX_OK;
}

extern "C" int mmx_rows_add(void* dense_dev, const void* rows_dev, const void* vals_dev, int B, int N, int E, void* stream) {
    MMX_CHECK_ARG(dense_dev && rows_dev && vals_dev && B > 0 && N > 0 && E > 0 && E % 4 == 0 && B <= 65535,
                  "mmx_rows_add: bad argument (B=%d N=%d E=%d, E %% 4 must be 0)", B, N, E);
    mmx::rows_add_kernel<<<dim3((E / 4 + 255) / 256, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<f32x4*>(dense_dev), static_cast<const long long*>(rows_dev), static_cast<const f32x4*>(vals_dev), N, E / 4);
    MMX_LAUNCH_CHECK("rows_add_kernel");
    return MMX_OK;
}

extern "C" int mmx_quick_gelu_bwd_bcast_bf16(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, int64_t x_n,
                                             void* stream) {
    MMX_CHECK_ARG(x_dev && dy_dev && dx_dev && n > 0 && x_n > 0, "mmx_quick_gelu_bwd_bcast_bf16: bad argument");
    MMX_CHECK_ARG(n % x_n == 0 && x_n % 8 == 0, "mmx_quick_gelu_bwd_bcast_bf16: n=%ld must be a multiple of x_n=%ld, x_n %% 8 == 0",
                  static_cast<long>(n), static_cast<long>(x_n));
    MMX_CHECK_ARG(((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(dy_dev) |
                    reinterpret_cast<uintptr_t>(dx_dev)) & 15u) == 0, "mmx_quick_gelu_bwd_bcast_bf16: pointers must be 16-byte aligned");
    const int64_t n8 = n / 8;
    if (n / x_n >= 4 && n / x_n <= 65535 * 8) {       // shared-forward mode: derivative once per column, swept over the samples
        const int batch = static_cast<int>(n / x_n), per = 8;
        const dim3 grid(static_cast<unsigned>((x_n / 8 + 255) / 256), static_cast<unsigned>((batch + per - 1) / per));
        quick_gelu_bwd_bf16_sweep_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
            static_cast<const f32x4*>(x_dev), static_cast<const u32x4*>(dy_dev), static_cast<u32x4*>(dx_dev), x_n / 8, batch, per);
        MMX_LAUNCH_CHECK("quick_gelu_bwd_bf16_sweep_kernel");
        return MMX_OK;
    }
    quick_gelu_bwd_bf16_kernel<<<gelu_grid(n8), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const f32x4*>(x_dev), static_cast<const u32x4*>(dy_dev), static_cast<u32x4*>(dx_dev), n8, x_n / 8);
    MMX_LAUNCH_CHECK("quick_gelu_bwd_bf16_kernel");
    return MMX_OK;
}

extern "C" int mmx_layernorm_bwd_add_bf16(const void* dy_dev, const void* x_dev, const void* mean_dev, const void* rstd_dev,
                                          const void* gamma_dev, const void* d_res_dev, void* dx_dev, void* dx_bf16_dev,
                                          int64_t rows, int x_rows, int E, void* stream) {
    MMX_CHECK_ARG(dy_dev && x_dev && mean_dev && rstd_dev && gamma_dev && (dx_dev || dx_bf16_dev),
                  "mmx_layernorm_bwd_add_bf16: null pointer");
    MMX_CHECK_ARG(rows > 0 && x_rows > 0 && E > 0 && E % 4 == 0, "mmx_layernorm_bwd_add_bf16: rows=%ld x_rows=%d E=%d (E %% 4 must be 0)",
                  static_cast<long>(rows), x_rows, E);
    const bool aligned = ((reinterpret_cast<uintptr_t>(dy_dev) | reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(gamma_dev) |
                           reinterpret_cast<uintptr_t>(d_res_dev) | reinterpret_cast<uintptr_t>(dx_dev) |
                           reinterpret_cast<uintptr_t>(dx_bf16_dev)) & 15u) == 0;
#define MMX_LN_ROW(ITER)                                                                                                          \
    if (aligned && E == 256 * ITER) {                                                                                            \
        mmx::layernorm_bwd_add_bf16_row_kernel<ITER><<<static_cast<unsigned>((rows + 3) / 4), 256, 0,                             \
                                                       static_cast<hipStream_t>(stream)>>>(                                      \
            static_cast<const unsigned short*>(dy_dev), static_cast<const float*>(x_dev), static_cast<const float*>(mean_dev),   \
            static_cast<const float*>(rstd_dev), static_cast<const float*>(gamma_dev), static_cast<const float*>(d_res_dev),     \
            static_cast<float*>(dx_dev), static_cast<unsigned short*>(dx_bf16_dev), rows, x_rows);                               \
        MMX_LAUNCH_CHECK("layernorm_bwd_add_bf16_row_kernel");                                                                   \
        return MMX_OK;                                                                                                           \
    }
    MMX_LN_ROW(2) MMX_LN_ROW(3) MMX_LN_ROW(4) MMX_LN_ROW(5)
#undef MMX_LN_ROW
    mmx::layernorm_bwd_add_bf16_kernel<<<static_cast<unsigned>((rows + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const unsigned short*>(dy_dev), static_cast<const float*>(x_dev), static_cast<const float*>(mean_dev),
        static_cast<const float*>(rstd_dev), static_cast<const float*>(gamma_dev), static_cast<const float*>(d_res_dev),
        static_cast<float*>(dx_dev), static_cast<unsigned short*>(dx_bf16_dev), rows, x_rows, E);
    MMX_LAUNCH_CHECK("layernorm_bwd_add_bf16_kernel");
    return MMX_OK;
}
