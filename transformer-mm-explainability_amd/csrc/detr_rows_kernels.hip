// K2-DETR: the decoder half of DETR's rule schedule for callers that read ROWS of R_q_i (SURVEY.md section 2a K2;
// DETR/modules/ExplanationGenerator.py:27-43, 120-140 -- rules 6, 7 and 10 with eq. 8-9 and the NaN policy of :42).
//
// Generator.generate_ours returns row t of R_q_i (one per kept query).  That row is
//     sum_l  u_l . N(R_qq^(l))^T . C_l . N(R_ii)           u_l = e_t^T (I + B_L) ... (I + B_(l+1))
// with B_l / C_l the head-averaged decoder self- / cross-attention maps (rule 5), R_qq^(l) = (I + B_l) ... (I + B_1) and
// N(.) eq. 8-9.  Everything left of C_l lives in the 100-query space; C_l is the [Q x Ni] map whose slabs are the bytes of
// this step (K x H x Q x Ni x 4 per layer and operand: 30 MB at K = 10, Ni = 950).  Four launches for ALL decoder layers
// replace, per layer, avg_heads (self) + baddbmm + row_normalise + bmm + avg_heads (cross) + bmm + isnan / any / where / add +
// baddbmm (torch launches in the round-2 rows-only route):
//   0. detr_self_heads_kernel        B_l = rule 5 of every decoder self-attention map, chip-wide (round 6)
//   1. detr_decoder_vectors_kernel   one workgroup per sample, R_qq and B_l in LDS (products on the exact-fp32 MFMA): bottom-up B_l, R_qq^(l), N(R_qq^(l)) (+ the
//                                    diag >= 0 word of handle_residual and the NaN flag of the DETR policy), then top-down
//                                    w_l = u_l . N(R_qq^(l))^T and u_(l-1) = u_l (I + B_l)
//   2. detr_cross_rows_kernel        z_l = w_l . C_l WITHOUT materialising C_l: every (sample, layer, 64-column tile)
//                                    workgroup streams its slab columns once, rule 5 and the weighted row sum in registers
//   3. detr_rows_finish_kernel       s = sum_l clean_l ? z_l : 0   (clean_l: no NaN in N(R_qq^(l)) nor in C_l -- the
//                                    reference zeroes the NaNs of the rule-10 addition, whose row is NaN iff one of them is)
// Summation orders differ from the matrix route (fp32 rounding level; tests bound both against the oracle at 1e-5).
#include "mmx_common.h"

namespace mmx {
namespace {

constexpr int kMaxQ = 128;
constexpr int kVecThreads = 1024;
constexpr int kMaxDecLayers = 16;

struct DetrRowsArgs {
    const float* self_a[kMaxDecLayers]; const float* self_g[kMaxDecLayers];     // [K | 1, H, Q, Q]
    const float* cross_a[kMaxDecLayers]; const float* cross_g[kMaxDecLayers];   // [K | 1, H, Q, Ni]
    int L, K, H, Q, Ni;
    int64_t self_a_bs, cross_a_bs;       // batch stride of the probability slabs in elements (0: ONE forward shared by all samples)
    const long long* targets;            // [K] explained query of every sample
    float* Bq;                           // [K][L][Q][Q]   scratch: head-averaged self maps
    float* Rhat;                         // [K][L][Q][Q]   scratch: N(R_qq^(l))
    float* w;                            // [K][L][Q]
    float* z;                            // [K][L][Ni]
    int* flags;                          // [2][K][L]      NaN seen in N(R_qq^(l)) / in C_l
    float* diag_k;                       // [K]            min over layers of diag(R_qq^(l) - I)
    float* s_out;                        // [K][Ni]
    float* diag_min;                     // [1] or null
};

// LDS: R and B as [QP][LD] (QP = 16 ceil(Q / 16) rows, zero padded: the MFMA tiles read whole 16 x 4 blocks), LD = QP + 1.
// min that CARRIES a NaN (fminf drops it): the reference's ``assert diag.min() >= 0`` fails on a NaN diagonal, and so does the
// word the Python wrapper reads (``NaN >= 0`` is false) -- ops.handle_residual's atomic_min_float keeps NaN bits the same way.
__device__ __forceinline__ float min_keep_nan(float m, float v) { return (m != m || v != v) ? __builtin_nanf("") : fminf(m, v); }

// Rule 5 of the decoder SELF-attention maps, every (sample, layer) at once: B_l = mean_h clamp(G_l * A_l, 0) into the Bq scratch.  The maps do not
// depend on the chain, so they are reduced chip-wide here (K x L x ceil(Q^2 / 1024) workgroups, heads in ascending order: the same sums as
// avg_heads_kernel) instead of inside the one-workgroup-per-sample chain below, where the 2 H dependent slab reads per element were 50 of its
// 64 us per layer (round 6).
__global__ __launch_bounds__(256) void detr_self_heads_kernel(const DetrRowsArgs a) {
    const int l = blockIdx.y, k = blockIdx.z, QQ = a.Q * a.Q;
    const int e0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= QQ) return;
    const float* A = a.self_a[l] + static_cast<int64_t>(k) * a.self_a_bs;
    const float* G = a.self_g[l] + static_cast<int64_t>(k) * a.H * QQ;
    float* Bg = a.Bq + (static_cast<int64_t>(k) * a.L + l) * QQ;
    const float fH = static_cast<float>(a.H);
    if (e0 + 3 < QQ) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < a.H; ++h) {
            const f32x4 av = ldg4_u(A + static_cast<int64_t>(h) * QQ + e0), gv = ldg4_u(G + static_cast<int64_t>(h) * QQ + e0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] += relu_nan(gv[r] * av[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Bg[e0 + r] = sum[r] / fH;
    } else {
        for (int e = e0; e < QQ; ++e) {
            float sum = 0.f;
            for (int h = 0; h < a.H; ++h) sum += relu_nan(G[static_cast<int64_t>(h) * QQ + e] * A[static_cast<int64_t>(h) * QQ + e]);
            Bg[e] = sum / fH;
        }
    }
}

__global__ __launch_bounds__(kVecThreads) void detr_decoder_vectors_kernel(const DetrRowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = a.Q, NTQ = (Q + 15) / 16, QP = NTQ * 16, LD = QP + 1, QQ = Q * Q;
    float* R = smem;                     // [QP][LD]
    float* Bm = R + QP * LD;             // [QP][LD]
    float* u = Bm + QP * LD;             // [2][kMaxQ]
    float* red = u + 2 * kMaxQ;          // [kMaxQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = blockIdx.x;
    const int i_a = lane & 15, kk = lane >> 4;
    for (int e = tid; e < QP * LD; e += kVecThreads) {
        const int i = e / LD, j = e - i * LD;
        R[e] = (i == j && i < Q) ? 1.f : 0.f;
        Bm[e] = 0.f;
    }
    if (tid < 2 * a.L) a.flags[(tid / a.L) * a.K * a.L + k * a.L + tid % a.L] = 0;
    float dmin = __builtin_inff();
    __syncthreads();
    // ---- bottom-up: B_l (rule 5), R_qq <- R_qq + B_l R_qq (rule 6), N(R_qq) (eq. 8-9)
    for (int l = 0; l < a.L; ++l) {
        const float* Bg = a.Bq + (static_cast<int64_t>(k) * a.L + l) * QQ;          // B_l: detr_self_heads_kernel
        for (int e = tid; e < QQ; e += kVecThreads) Bm[(e / Q) * LD + e % Q] = Bg[e];
        __syncthreads();
        // R_new = R + B R on the exact-fp32 MFMA: 16 x 16 output tiles dealt round-robin to the 16 waves, both operands in LDS
        // (A = B_l [i][c], B = R [c][j], zero padded); the new tiles stay in registers until every wave has read the old R
        constexpr int kMaxTiles = (kMaxQ / 16) * (kMaxQ / 16) / (kVecThreads / 64);       // 4 tiles per wave at Q = 128
        f32x4 acc[kMaxTiles];
        int nt = 0;
        for (int tile = wave; tile < NTQ * NTQ; tile += kVecThreads / 64, ++nt) {
            const int ti = tile / NTQ, tj = tile - ti * NTQ;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < QP / 4; ++ks)
                c = mfma16x16x4(Bm[(ti * 16 + i_a) * LD + 4 * ks + kk], R[(4 * ks + kk) * LD + tj * 16 + i_a], c);
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] += R[(ti * 16 + kk * 4 + r) * LD + tj * 16 + i_a];
            if (nt < kMaxTiles) acc[nt] = c;
        }
        __syncthreads();
        nt = 0;
        for (int tile = wave; tile < NTQ * NTQ; tile += kVecThreads / 64, ++nt) {
            const int ti = tile / NTQ, tj = tile - ti * NTQ;
            if (nt < kMaxTiles) {
#pragma unroll
                for (int r = 0; r < 4; ++r) R[(ti * 16 + kk * 4 + r) * LD + tj * 16 + i_a] = acc[nt][r];
            }
        }
        __syncthreads();
        float* Hg = a.Rhat + (static_cast<int64_t>(k) * a.L + l) * QQ;
        {   // row sums of R - I: 8 lanes per row, 16 columns at a time each
            const int row = tid >> 3, part = tid & 7;
            float s = 0.f;
            if (row < Q)
                for (int j = part; j < Q; j += 8) s += R[row * LD + j] - (j == row ? 1.f : 0.f);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            if (row < Q && part == 0) {
                red[row] = s;
                dmin = min_keep_nan(dmin, R[row * LD + row] - 1.f);
            }
        }
        __syncthreads();
        bool nan_seen = false;
        for (int e = tid; e < QQ; e += kVecThreads) {
            const int i = e / Q, j = e % Q;
            const float d = (i == j) ? 1.f : 0.f;
            const float v = (R[i * LD + j] - d) / red[i] + d;
            nan_seen |= (v != v);
            Hg[e] = v;
        }
        if (nan_seen) atomicOr(&a.flags[k * a.L + l], 1);
        __syncthreads();
    }
    // diag(R - I) >= 0 contract of handle_residual: the smallest value any layer saw (threads 8 row own row's running minimum)
    if ((tid & 7) == 0 && (tid >> 3) < Q) red[tid >> 3] = dmin;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
        for (int i = 1; i < Q; ++i) m = min_keep_nan(m, red[i]);
        a.diag_k[k] = m;
    }
    // ---- top-down: w_l = u_l N(R_qq^(l))^T, u_(l-1) = u_l (I + B_l); 8 lanes per output element
    const int t = static_cast<int>(a.targets[k]);
    if (tid < Q) u[tid] = (tid == t) ? 1.f : 0.f;
    int cur = 0;
    for (int l = a.L - 1; l >= 0; --l) {
        __syncthreads();                                            // u[cur] complete; R / Bm free (and this workgroup's global writes done)
        const float* Hg = a.Rhat + (static_cast<int64_t>(k) * a.L + l) * QQ;
        const float* Bg = a.Bq + (static_cast<int64_t>(k) * a.L + l) * QQ;
        for (int e = tid; e < QQ; e += kVecThreads) {
            R[(e / Q) * LD + e % Q] = Hg[e];
            Bm[(e / Q) * LD + e % Q] = Bg[e];
        }
        __syncthreads();
        const int row = tid >> 3, part = tid & 7;
        float s = 0.f, v = 0.f;
        if (row < Q)
            for (int j = part; j < Q; j += 8) {
                const float uj = u[cur * kMaxQ + j];
                s += uj * R[row * LD + j];                           // w[q] = sum_j u[j] N[q][j]
                v += uj * Bm[j * LD + row];                          // (u B)[q] = sum_c u[c] B[c][q]
            }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        if (row < Q && part == 0) {
            a.w[(static_cast<int64_t>(k) * a.L + l) * Q + row] = s;
            u[(cur ^ 1) * kMaxQ + row] = u[cur * kMaxQ + row] + v;
        }
        cur ^= 1;
    }
}

constexpr int kCrossCols = 64;   // columns per workgroup: 16 lanes x 4

__global__ __launch_bounds__(256) void detr_cross_rows_kernel(const DetrRowsArgs a) {
    __shared__ float part[16][kCrossCols + 4];
    const int tid = threadIdx.x, cl = tid & 15, qg = tid >> 4;
    const int n0 = blockIdx.x * kCrossCols + 4 * cl, l = blockIdx.y, k = blockIdx.z;
    const int Ni = a.Ni, Q = a.Q;
    const int64_t hs = static_cast<int64_t>(Q) * Ni;
    const float* A = a.cross_a[l] + static_cast<int64_t>(k) * a.cross_a_bs;
    const float* G = a.cross_g[l] + static_cast<int64_t>(k) * a.H * hs;
    const float* w = a.w + (static_cast<int64_t>(k) * a.L + l) * Q;
    const float fH = static_cast<float>(a.H);
    const bool full = n0 + 3 < Ni, any = n0 < Ni;
    f32x4 zacc = {0.f, 0.f, 0.f, 0.f};
    bool nan_seen = false;
    for (int q = qg; q < Q; q += 16) {
        const float wq = w[q];
        const int64_t off = static_cast<int64_t>(q) * Ni + n0;
        f32x4 cam = {0.f, 0.f, 0.f, 0.f};
        if (full) {
            // heads in ascending order (the reference's sum), four at a time: the eight 16-byte loads of a batch are requested before
            // the first is used (one load pair per dependent iteration left the slabs at 1.4 TB/s); a padding head re-reads head H - 1
            // and is selected away
            for (int h0 = 0; h0 < a.H; h0 += 4) {
                f32x4 av[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t ho = static_cast<int64_t>(min(h0 + u, a.H - 1)) * hs + off;
                    av[u] = ldg4_u(A + ho);
                    gv[u] = ldg4_u(G + ho);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool on = h0 + u < a.H;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cam[r] += on ? relu_nan(gv[u][r] * av[u][r]) : 0.f;
                }
            }
        } else if (any) {
            for (int h = 0; h < a.H; ++h)
                for (int r = 0; r < 4 && n0 + r < Ni; ++r) cam[r] += relu_nan(G[h * hs + off + r] * A[h * hs + off + r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = cam[r] / fH;                            // (as avg_heads_kernel: sum / H)
            nan_seen |= (c != c);
            zacc[r] += wq * c;
        }
    }
    *reinterpret_cast<f32x4*>(&part[qg][4 * cl]) = zacc;
    if (nan_seen) atomicOr(&a.flags[(a.K + k) * a.L + l], 1);
    __syncthreads();
    if (qg == 0 && any) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 16; ++j) s += *reinterpret_cast<const f32x4*>(&part[j][4 * cl]);       // fixed order: deterministic
        float* out = a.z + (static_cast<int64_t>(k) * a.L + l) * Ni + n0;
        for (int r = 0; r < 4 && n0 + r < Ni; ++r) out[r] = s[r];
    }
}

__global__ __launch_bounds__(256) void detr_rows_finish_kernel(const DetrRowsArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (n < a.Ni) {
        float s = 0.f;
        for (int l = a.L - 1; l >= 0; --l) {                       // the order the rows-only route adds them in (top-down)
            const bool clean = !(a.flags[k * a.L + l] | a.flags[(a.K + k) * a.L + l]);
            const float zl = a.z[(static_cast<int64_t>(k) * a.L + l) * a.Ni + n];
            s += clean ? zl : 0.f;
        }
        a.s_out[static_cast<int64_t>(k) * a.Ni + n] = s;
    }
    if (a.diag_min && blockIdx.x == 0 && k == 0 && threadIdx.x == 0) {
        float m = a.diag_k[0];
        for (int i = 1; i < a.K; ++i) m = min_keep_nan(m, a.diag_k[i]);
        *a.diag_min = m;
    }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace
}  // namespace mmx

using namespace mmx;

extern "C" size_t mmx_detr_decoder_rows_workspace_bytes(int n_layers, int K, int Q, int Ni) {
    const size_t kl = static_cast<size_t>(K) * n_layers;
    return 2 * align256(sizeof(float) * kl * Q * Q) + align256(sizeof(float) * kl * Q) + align256(sizeof(float) * kl * Ni) +
           align256(sizeof(int) * 2 * kl) + align256(sizeof(float) * K);
}

extern "C" int mmx_detr_decoder_rows(const void* const* self_attn, const void* const* self_grad, const void* const* cross_attn,
                                     const void* const* cross_grad, int n_layers, int K, int H, int Q, int Ni,
                                     int64_t self_attn_bstride, int64_t cross_attn_bstride, const void* targets_dev,
                                     void* s_out_dev, void* diag_min_dev, void* workspace_dev, size_t workspace_bytes,
                                     void* stream) {
    MMX_CHECK_ARG(self_attn && self_grad && cross_attn && cross_grad && targets_dev && s_out_dev,
                  "mmx_detr_decoder_rows: null pointer");
    MMX_CHECK_ARG(n_layers >= 1 && n_layers <= kMaxDecLayers, "mmx_detr_decoder_rows: %d layers (1..%d)", n_layers, kMaxDecLayers);
    MMX_CHECK_ARG(K >= 1 && K <= 65535 && H >= 1 && Ni >= 1, "mmx_detr_decoder_rows: bad sizes K=%d H=%d Ni=%d", K, H, Ni);
    if (Q < 1 || Q > kMaxQ) {
        set_error("mmx_detr_decoder_rows: %d queries (1..%d: R_qq and B_l live in LDS)", Q, kMaxQ);
        return MMX_ENOTSUP;
    }
    const size_t need = mmx_detr_decoder_rows_workspace_bytes(n_layers, K, Q, Ni);
    if (!workspace_dev || workspace_bytes < need) {
        set_error("mmx_detr_decoder_rows: workspace %zu < %zu", workspace_bytes, need);
        return MMX_EWORKSPACE;
    }
    DetrRowsArgs a;
    for (int l = 0; l < n_layers; ++l) {
        MMX_CHECK_ARG(self_attn[l] && self_grad[l] && cross_attn[l] && cross_grad[l], "mmx_detr_decoder_rows: null layer %d", l);
        a.self_a[l] = static_cast<const float*>(self_attn[l]); a.self_g[l] = static_cast<const float*>(self_grad[l]);
        a.cross_a[l] = static_cast<const float*>(cross_attn[l]); a.cross_g[l] = static_cast<const float*>(cross_grad[l]);
    }
    a.L = n_layers; a.K = K; a.H = H; a.Q = Q; a.Ni = Ni;
    a.self_a_bs = self_attn_bstride; a.cross_a_bs = cross_attn_bstride;
    a.targets = static_cast<const long long*>(targets_dev);
    const size_t kl = static_cast<size_t>(K) * n_layers;
    char* p = static_cast<char*>(workspace_dev);
    a.Bq = reinterpret_cast<float*>(p); p += align256(sizeof(float) * kl * Q * Q);
    a.Rhat = reinterpret_cast<float*>(p); p += align256(sizeof(float) * kl * Q * Q);
    a.w = reinterpret_cast<float*>(p); p += align256(sizeof(float) * kl * Q);
    a.z = reinterpret_cast<float*>(p); p += align256(sizeof(float) * kl * Ni);
    a.flags = reinterpret_cast<int*>(p); p += align256(sizeof(int) * 2 * kl);
    a.diag_k = reinterpret_cast<float*>(p);
    a.s_out = static_cast<float*>(s_out_dev);
    a.diag_min = static_cast<float*>(diag_min_dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t QP = (static_cast<size_t>(Q) + 15) / 16 * 16;
    const size_t lds = sizeof(float) * (2 * QP * (QP + 1) + 3 * kMaxQ);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(detr_decoder_vectors_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    detr_self_heads_kernel<<<dim3((Q * Q + 1023) / 1024, n_layers, K), 256, 0, s>>>(a);
    MMX_LAUNCH_CHECK("detr_self_heads_kernel");
    detr_decoder_vectors_kernel<<<dim3(K), kVecThreads, lds, s>>>(a);
    MMX_LAUNCH_CHECK("detr_decoder_vectors_kernel");
    detr_cross_rows_kernel<<<dim3((Ni + kCrossCols - 1) / kCrossCols, n_layers, K), 256, 0, s>>>(a);
    MMX_LAUNCH_CHECK("detr_cross_rows_kernel");
    detr_rows_finish_kernel<<<dim3((Ni + 255) / 256, K), 256, 0, s>>>(a);
    MMX_LAUNCH_CHECK("detr_rows_finish_kernel");
    return MMX_OK;
}
