// On-device post-processing of relevancy maps: the reference does these per map on the
// host (D2H copy + numpy / cv2), i.e. one device synchronisation per map inside the evaluator loops.
//
//  * heatmap: bilinear upsample of the g x g patch map to S x S (torch.nn.functional.interpolate, mode='bilinear',
//    align_corners=False) followed by min-max normalisation -- CLIP_explainability.ipynb cell 7:14-18,
//    Transformer_MM_explainability_ViT.ipynb cell 8:25-28.
//  * otsu: min-max to [0, 255], truncation to uint8, Otsu threshold, binary mask (255 / 0) --
//    DETR/mask_generator.py:116-121 (`cv2.threshold(..., THRESH_BINARY + THRESH_OTSU)`).
// One workgroup per map; everything stays on the GPU and on the stream.
#include "mmx_common.h"

namespace mmx {

__device__ __forceinline__ float wave_min(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fminf(x, __shfl_xor(x, off));
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
    return x;
}

// workgroup-wide min & max through LDS (red[0..31] min, red[32..63] max); returns them to every thread
__device__ __forceinline__ void block_minmax(float& lo, float& hi, float* red, int tid, int nthreads) {
    lo = wave_min(lo);
    hi = wave_max(hi);
    const int wave = tid >> 6, nw = nthreads >> 6;
    if ((tid & 63) == 0) { red[wave] = lo; red[32 + wave] = hi; }
    __syncthreads();
    lo = red[0]; hi = red[32];
    for (int w = 1; w < nw; ++w) { lo = fminf(lo, red[w]); hi = fmaxf(hi, red[32 + w]); }
    __syncthreads();
}

// ATen's upsample_bilinear2d, align_corners = False: src = max(0, (dst + 0.5) * in/out - 0.5)
__device__ __forceinline__ float bilinear_at(const float* src, int g, float scale, int y, int x) {
    float fy = (y + 0.5f) * scale - 0.5f, fx = (x + 0.5f) * scale - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
    const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * src[y0 * g + x0] + lx * src[y0 * g + x1]) + ly * (hx * src[y1 * g + x0] + lx * src[y1 * g + x1]);
}

__global__ __launch_bounds__(1024) void heatmap_kernel(const float* __restrict__ in, float* __restrict__ out, int g,
                                                       int S) {
    __shared__ float src[64 * 64];
    __shared__ float red[64];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < g * g; i += 1024) src[i] = in[static_cast<int64_t>(b) * g * g + i];
    __syncthreads();
    const float scale = static_cast<float>(g) / static_cast<float>(S);
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int p = tid; p < S * S; p += 1024) {
        const float v = bilinear_at(src, g, scale, p / S, p % S);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, red, tid, 1024);
    const float range = hi - lo;
    float* o = out + static_cast<int64_t>(b) * S * S;
    for (int p = tid; p < S * S; p += 1024) o[p] = (bilinear_at(src, g, scale, p / S, p % S) - lo) / range;
}

__global__ __launch_bounds__(256) void otsu_kernel(const float* __restrict__ cam, float* __restrict__ masks,
                                                   int* __restrict__ thresholds, int n) {
    __shared__ float red[64];
    __shared__ unsigned hist[256];
    __shared__ int thr;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* c = cam + static_cast<int64_t>(b) * n;
    hist[tid] = 0;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int i = tid; i < n; i += 256) { lo = fminf(lo, c[i]); hi = fmaxf(hi, c[i]); }
    block_minmax(lo, hi, red, tid, 256);
    const float range = hi - lo;
    for (int i = tid; i < n; i += 256) {
        const float v = (c[i] - lo) / range * 255.f;                 // mask_generator.py:116
        const int q = static_cast<int>(v);                            // numpy astype(uint8): truncation
        atomicAdd(&hist[q < 0 ? 0 : (q > 255 ? 255 : q)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        // OpenCV getThreshVal_Otsu_8u (imgproc/thresh.cpp): double precision, first maximum of the between-class variance
        const double scale = 1.0 / n;
        double mu = 0;
        for (int i = 0; i < 256; ++i) mu += static_cast<double>(i) * hist[i];
        mu *= scale;
        double mu1 = 0, q1 = 0, max_sigma = 0;
        int max_val = 0;
        for (int i = 0; i < 256; ++i) {
            const double p_i = hist[i] * scale;
            mu1 *= q1;
            q1 += p_i;
            const double q2 = 1.0 - q1;
            if (fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1.0 - 1.1920928955078125e-07) continue;
            mu1 = (mu1 + i * p_i) / q1;
            const double mu2 = (mu - q1 * mu1) / q2;
            const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
            if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
        }
        thr = max_val;
        if (thresholds) thresholds[b] = max_val;
    }
    __syncthreads();
    const int t = thr;
    for (int i = tid; i < n; i += 256) {
        const int q = static_cast<int>((c[i] - lo) / range * 255.f);
        masks[static_cast<int64_t>(b) * n + i] = (q > t) ? 255.f : 0.f;   // THRESH_BINARY: src > thresh
    }
}

}  // namespace mmx

using namespace mmx;

extern "C" int mmx_heatmap_bilinear_minmax(const void* in_dev, void* out_dev, int B, int g, int S, void* stream) {
    MMX_CHECK_ARG(in_dev && out_dev, "mmx_heatmap_bilinear_minmax: null pointer");
    MMX_CHECK_ARG(B > 0 && g > 0 && S > 0, "mmx_heatmap_bilinear_minmax: non-positive size");
    if (g > 64) {
        set_error("mmx_heatmap_bilinear_minmax: patch grid %d > 64 not supported", g);
        return MMX_ENOTSUP;
    }
    heatmap_kernel<<<B, 1024, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const float*>(in_dev),
                                                                   static_cast<float*>(out_dev), g, S);
    MMX_LAUNCH_CHECK("heatmap_kernel");
    return MMX_OK;
}

extern "C" int mmx_otsu_masks(const void* cam_dev, void* masks_dev, void* thresholds_dev, int K, int n, void* stream) {
    MMX_CHECK_ARG(cam_dev && masks_dev, "mmx_otsu_masks: null pointer");
    MMX_CHECK_ARG(K > 0 && n > 0, "mmx_otsu_masks: non-positive size");
    otsu_kernel<<<K, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const float*>(cam_dev),
                                                                static_cast<float*>(masks_dev),
                                                                static_cast<int*>(thresholds_dev), n);
    MMX_LAUNCH_CHECK("otsu_kernel");
    return MMX_OK;
}
