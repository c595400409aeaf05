// K1c  self_chain_cols_kernel: the self-attention relevancy chain (rules 5 + 6, all layers in one launch) in strict layer order, one
// workgroup per sample by default -- what runs for fp32 slabs when the batch alone fills the chip (no layer groups), and the
// experiment VERDICT r04 asked for (split by COLUMNS of R over several workgroups of a sample).
//
//     R <- R + A_bar_l . R   (l = 0 .. L-1, strict order),   A_bar_l = mean_h clamp(G_l * A_l, 0)
//
// is independent per column of R.  A sample's C workgroups each own every C-th 16-column slab of R over ALL layers (MFMA accumulators
// of one wave per slab: the K1 trick), and each of them reduces the FULL A_bar_l of every layer itself.  With C > 1 the workgroups
// of a sample are placed on one XCD (ids congruent mod 8), so a slab line is fetched from HBM by whichever of them asks first and
// re-served from that XCD's L2 to the others; workgroup c starts every layer at block c * NBLK / C so that each of them is the first
// to ask for a different part.  There are no partial products, no hand-off, no counters in global memory, no scratch: the result is
// the sequential chain, bit-identical to self_chain_fused_kernel with one group, at any batch size.
//
// Measured (profiles/r05_chain_cols_probe.txt): C = 2..4 at B = 64 takes 107-137 us against 73-74 us of the layer-group kernel -- a CU
// ingests ~43 GB/s at most, from L2 as from HBM, and every workgroup has to ingest the sample's whole 4.58 MB.  C = 1 however beats
// the fused kernel's single-group form at every batch (B = 64 118 vs 145 us, B = 160 139 vs 162 us, B = 256 202 vs 207 us at CLIP's
// text shape) because its stream waves never meet a barrier: it is the default for fp32 slabs whenever one group is chosen.
//
// Reference sites: CLIP_explainability.ipynb cell 6:22-32 / 45-55, CLIP/example.py:22-30, ViT notebook cell 7:28-33,
// VisualBERT/.../ExplanationGenerator.py:86-93 (include/mmx_relevancy.h, mmx_relevancy_self_chain).
//
// Inside a workgroup (1024 threads): `nown` matrix waves (one per owned slab) + 16 - nown stream waves.  Stream waves are the
// barrier-free register pipelines of relevancy_chain_groups.hip (64-chunk blocks of (layer, block) items in layer order, heads
// summed in ascending order), scattering into a ring of NB LDS images of A_bar (slot l % NB) and counting arrivals per 16-row tile;
// a matrix wave follows tile by tile.  Back-pressure on the ring: a stream wave about to write layer l waits until every matrix wave
// has finished layer l - NB (progress words in LDS) -- the matrix waves need 1.3 us per layer against ~5 us of streaming, so it
// does not wait in practice.
#include "chain_stream.h"

namespace mmx {

struct ColsArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    int C;               // workgroups per sample (<= NT)
    int NB;              // LDS images of A_bar (ring)
    int nchunks;         // ceil(N*N / 4)
    unsigned row_magic;  // ceil(2^32 / N): row = (p * magic) >> 32, exact for p < N*N + 8
    const float* R_init;
    float* R_out;
    int64_t attn_bstride;
    int nt;
    int causal;          // MMX_CHAIN_CAUSAL: chunks entirely above the diagonal are not requested (chain_stream.h)
    int debug;           // profiling only: bit0 = matrix waves skip the MFMAs, bit1 = no block rotation
};

constexpr int kColsThreads = 1024;
constexpr int kColsWaves = kColsThreads / 64;

template <int NT>
__global__ __launch_bounds__(kColsThreads) void self_chain_cols_kernel(const ColsArgs a) {
    constexpr int NP = NT * 16;
    constexpr int S = NP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // NB images (NP x S floats) + NB x NT arrival counters + 16 progress words

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int C = a.C, NB = a.NB;
    // the C workgroups of a sample on ONE XCD (workgroup w runs on XCD w % 8).  A placement hint only.
    int b, c;
    {
        const int w = blockIdx.x, full = (a.B >> 3) * 8 * C;
        if (w < full) { b = (w & 7) + 8 * ((w >> 3) / C); c = (w >> 3) % C; }
        else { b = (a.B >> 3) * 8 + (w - full) / C; c = (w - full) % C; }
    }
    const int N = a.N, H = a.H, L = a.n_layers;
    const int NN = N * N;
    const int nown = (NT - c + C - 1) / C;                                     // slabs c, c + C, ... < NT
    unsigned* lds_cnt = reinterpret_cast<unsigned*>(smem + NB * NP * S);       // [NB][NT] elements landed per (slot, 16-row tile), cumulative
    unsigned* prog = lds_cnt + NB * NT;                                        // [16] layers finished per matrix wave

    {   // pads must read as 0; counters = 0
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int n4 = (NB * NP * S + NB * NT + 16 + 3) >> 2;            // (rounded up: the allocation has the slack)
        for (int i = tid; i < n4; i += kColsThreads) reinterpret_cast<f32x4*>(smem)[i] = z;
    }
    __syncthreads();

    const int NBLK = (a.nchunks + 63) >> 6;            // 64-chunk blocks per layer
    const int rot = (a.debug & 2) ? 0 : (c * NBLK) / C;

    if (wave >= nown) {
        // =============================================================================================== stream waves (chain_stream.h)
        const ChainStreamGeom gm{a.attn, a.grad, 0, NBLK, rot, a.nchunks, H, NN, b, a.B, a.attn_bstride, (a.nt && C == 1) ? 1 : 0, a.causal, N, a.row_magic};
        // (with several workgroups per sample the partners re-read every line from L2: default cache policy)
        chain_stream_wave(gm, wave - nown, kColsWaves - nown, L * NBLK, lane, [&](int l, int cidx, f32x4 mean) {
            const int slot = l % NB;
            if (l >= NB) {
                // ring back-pressure: every matrix wave must have left layer l - NB (bounded like every wait here)
                const unsigned need = static_cast<unsigned>(l - NB + 1);
                int turns = 0;
                while (!__all(lane >= nown ||
                              __hip_atomic_load(prog + (lane & 15), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) &&
                       ++turns < (1 << 24))
                    __builtin_amdgcn_s_sleep(1);
                if (turns >= (1 << 24) && lane == 0) prog[15] = 1u;        // (word 15 is no wave's progress: "a wait gave up")
            }
            // arrival counts are cumulative over the uses of the slot
            chain_stream_deliver(smem + slot * NP * S, S, lds_cnt + slot * NT, cidx, lane, a.nchunks, N, NN, a.row_magic, mean);
        });
    } else {
        // =============================================================================================== matrix waves
        const int slab = c + wave * C;
        const int col = slab * 16 + (lane & 15);
        const int rq = (lane >> 4) * 4;
        f32x4 Rold[NT], Rnew[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                float v = 0.f;
                if (row < N && col < N)
                    v = a.R_init ? a.R_init[static_cast<int64_t>(b) * NN + row * N + col] : (row == col ? 1.f : 0.f);
                Rold[t][r] = v;
            }
        bool poisoned = false;
        for (int l = 0; l < L; ++l) {
            const int slot = l % NB;
            const unsigned uses = static_cast<unsigned>(l / NB + 1);
            const float* Ab = smem + slot * NP * S + (lane & 15) * S + rq;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const unsigned expect = uses * static_cast<unsigned>(max(0, min(N, ti * 16 + 16) - ti * 16) * N);
                int turns = 0;
                while (__hip_atomic_load(lds_cnt + slot * NT + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expect &&
                       ++turns < (1 << 24))
                    __builtin_amdgcn_s_sleep(1);
                if (turns >= (1 << 24)) poisoned = true;        // never seen; if it happens the result says so (NaN), it does not lie
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (!(a.debug & 1)) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + ti * 16 * S + t * 16);
                        acc = mfma16x16x4(av[0], Rold[t][0], acc);
                        acc = mfma16x16x4(av[1], Rold[t][1], acc);
                        acc = mfma16x16x4(av[2], Rold[t][2], acc);
                        acc = mfma16x16x4(av[3], Rold[t][3], acc);
                    }
                }
                Rnew[ti] = Rold[ti] + acc;                  // R + (A_bar . R): same association as the reference
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) Rold[t] = Rnew[t];
            // this wave's reads of the slot are done (its MFMAs have their operands): release it to the stream waves
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog + wave, static_cast<unsigned>(l + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (__hip_atomic_load(prog + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) poisoned = true;
        float* dst = a.R_out + static_cast<int64_t>(b) * NN;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                if (row < N && col < N) dst[row * N + col] = poisoned ? __builtin_nanf("") : Rold[t][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------------------ host side
static size_t cols_lds_bytes(int nt, int nb) {
    return sizeof(float) * (static_cast<size_t>(nb) * nt * 16 * (nt * 16 + 4) + static_cast<size_t>(nb) * nt + 20);
}

static int g_cols_c = 0, g_cols_nb = 0;    // 0 = auto
void chain_cols_options(int c, int nb) {
    if (c >= 0) g_cols_c = c;
    if (nb >= 0) g_cols_nb = nb;
}

// workgroups per sample / ring depth
static void cols_plan(int n_layers, int B, int N, int* C, int* NB) {
    const int nt = (N + 15) / 16;
    // One workgroup per sample unless asked: a CU ingests at most ~43 GB/s whether the lines come from HBM or from its XCD's L2
    // (profiles/r05_chain_cols_probe.txt: 4.58 MB per sample = 107-137 us with C = 2..4 at B = 64), so several workgroups that each
    // reduce the whole sample never beat the layer-group kernel; C > 1 stays as the measured experiment.
    (void)B;
    int c = g_cols_c ? g_cols_c : 1;
    if (c > nt) c = nt;
    int nb = g_cols_nb ? g_cols_nb : 6;
    while (nb > 1 && cols_lds_bytes(nt, nb) > 160 * 1024) --nb;
    if (nb > n_layers && n_layers > 0) nb = n_layers;
    *C = c;
    *NB = nb;
}

bool self_chain_cols_applies(int n_layers, int B, int H, int N) {
    const int nt = (N + 15) / 16;
    return n_layers >= 1 && nt <= 8 && cols_lds_bytes(nt, 1) <= 160 * 1024 && static_cast<size_t>(H) * N * N * 4 < (1ull << 31);
}

template <int NT>
static int cols_launch(const ColsArgs& r, hipStream_t s) {
    const size_t lds = cols_lds_bytes(NT, r.NB);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(self_chain_cols_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    self_chain_cols_kernel<NT><<<r.B * r.C, kColsThreads, lds, s>>>(r);
    MMX_LAUNCH_CHECK("self_chain_cols_kernel");
    return MMX_OK;
}

int self_chain_cols_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N,
                           int64_t attn_bstride, const void* R_init, void* R_out, int nt_policy, int debug, hipStream_t s) {
    ColsArgs r;
    memset(&r, 0, sizeof(r));
    for (int l = 0; l < n_layers; ++l) { r.attn[l] = attn_layers[l]; r.grad[l] = grad_layers[l]; }
    r.n_layers = n_layers; r.B = B; r.H = H; r.N = N;
    cols_plan(n_layers, B, N, &r.C, &r.NB);
    r.nchunks = (N * N + 3) / 4;
    r.row_magic = static_cast<unsigned>((0x100000000ull + N - 1) / N);
    r.R_init = static_cast<const float*>(R_init);
    r.R_out = static_cast<float*>(R_out);
    r.attn_bstride = attn_bstride;
    r.nt = nt_policy & 1;              // `nt_policy`: bit 0 = nt loads on the read-once slabs, bit 1 = MMX_CHAIN_CAUSAL
    r.causal = (nt_policy >> 1) & 1;
    r.debug = debug;
    switch ((N + 15) / 16) {
        case 1: return cols_launch<1>(r, s);
        case 2: return cols_launch<2>(r, s);
        case 3: return cols_launch<3>(r, s);
        case 4: return cols_launch<4>(r, s);
        case 5: return cols_launch<5>(r, s);
        case 6: return cols_launch<6>(r, s);
        case 7: return cols_launch<7>(r, s);
        default: return cols_launch<8>(r, s);
    }
}

}  // namespace mmx
