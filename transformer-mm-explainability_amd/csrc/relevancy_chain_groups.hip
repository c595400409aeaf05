// K1g  self_chain_groups_kernel: the layer-group form of the self-attention relevancy chain (rules 5 + 6, one launch for all
// layers) with BARRIER-FREE stream waves -- round 5's successor of self_chain_fused_kernel for fp32 slabs and G > 1.
//
//     workgroup (b, g):  P_g = prod_{l in group g} (I + A_bar_l)   (group 0 starts from R_init),   A_bar_l = mean_h clamp(G_l * A_l, 0)
//     last arriver of sample b:  R = P_{G-1} . ... . P_1 . P_0
//
// Reference sites: CLIP_explainability.ipynb cell 6:22-32 / 45-55, CLIP/example.py:22-30, ViT notebook cell 7:28-33,
// VisualBERT/.../ExplanationGenerator.py:86-93 (include/mmx_relevancy.h, mmx_relevancy_self_chain).
//
// What changed against self_chain_fused_kernel (relevancy_kernels.hip), which made stream and matrix waves meet at one s_barrier per
// layer and let the stream waves walk a whole layer as one chunk loop: round 5's relay experiments (a strict-order kernel fed through
// L2, since removed: profiles/r05_chain_relay_probe.txt)
// showed that stream waves which never wait -- each with its own register software pipeline over 64-chunk blocks -- reach 0.58 of the
// HBM peak against 0.50-0.53 for the coupled form.  Here:
//   * every A_bar_l of the group has its OWN LDS buffer (NP x (NP + 4) floats each), so a stream wave never has to wait
//     for a buffer: it takes the blocks ws, ws + NWs, ... of the group's (layer, 64-chunk block) list in layer order, reduces the heads
//     IN ORDER (two register sets of 4 heads x 2 arrays: 16 x 16 B per lane in flight, raw buffer loads, nt policy, resources that end
//     at the tensor end), scatters its 64 chunks into the layer's LDS image and adds its chunk count to the layer's LDS counter;
//   * a matrix wave (16-column slab of R in MFMA accumulators, the K1 trick) spins on that counter and multiplies: no s_barrier
//     between the two roles anywhere;
//   * hand-off and combine as before (write-through partial products, one ticket per sample, the last arriver multiplies), i.e.
//     the products are re-associated at the group boundaries exactly as in self_chain_fused_kernel (same 1e-5 bound, same tests).
#include "chain_stream.h"

namespace mmx {

struct GroupsArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    int G;               // layer groups per sample
    int nchunks;         // ceil(N*N / 4)
    unsigned row_magic;  // ceil(2^32 / N): row = (p * magic) >> 32, exact for p < N*N + 8
    const float* R_init;
    float* R_out;
    float* parts;        // [B][G][N*N] partial products (scratch)
    unsigned* counters;  // [B] arrival tickets
    int64_t attn_bstride;
    int nt;
    int causal;          // MMX_CHAIN_CAUSAL: chunks entirely above the diagonal are not requested (chain_stream.h)
    int debug;           // profiling only: bit0 = return before the hand-off / combine, bit2 = matrix waves skip the MFMAs,
                         // bit3 = return after the ticket (no combine)
};

constexpr int kGroupsThreads = 1024;
constexpr int kGroupsWaves = kGroupsThreads / 64;

template <int NT>
__global__ __launch_bounds__(kGroupsThreads) void self_chain_groups_kernel(const GroupsArgs a) {
    constexpr int NP = NT * 16;
    constexpr int S = NP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // A_bar images (NP x S floats each) + per x NT counters + ticket

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int G = a.G;
    // the G workgroups of a sample on ONE XCD (workgroup w runs on XCD w % 8): the partial products then meet in that XCD's L2.
    // A placement hint only -- nothing below depends on it.
    int b, g;
    {
        const int w = blockIdx.x, full = (a.B >> 3) * 8 * G;
        if (w < full) { b = (w & 7) + 8 * ((w >> 3) / G); g = (w >> 3) % G; }
        else { b = (a.B >> 3) * 8 + (w - full) / G; g = (w - full) % G; }
    }
    const int N = a.N, H = a.H;
    const int per = (a.n_layers + G - 1) / G;
    const int l0 = min(a.n_layers, g * per), l1 = min(a.n_layers, l0 + per);
    const int L = l1 - l0;
    const int NN = N * N;
    const int images = per > 3 ? per : 3;                                      // (the combine uses three)
    unsigned* lds_cnt = reinterpret_cast<unsigned*>(smem + images * NP * S);   // [per][NT] elements of A_bar_l landed per 16-row tile
    unsigned* ticket_lds = lds_cnt + per * NT;

    {   // pads must read as 0; counters = 0
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int n4 = (images * NP * S + per * NT + 1 + 3) >> 2;        // (NP * S is a multiple of 4; the allocation is rounded up)
        for (int i = tid; i < n4; i += kGroupsThreads) reinterpret_cast<f32x4*>(smem)[i] = z;
    }
    __syncthreads();

    const int col = wave * 16 + (lane & 15);
    const int rq = (lane >> 4) * 4;
    f32x4 Rold[NT], Rnew[NT];
    // MMX_CHAIN_CAUSAL and R starts as the identity: every A_bar_l, every R and every partial product is LOWER TRIANGULAR, so the
    // 16 x 16 tile (ti, tj) of a product is sum over tj <= t <= ti only -- 35 of the 125 tile products at 77 tokens.  The skipped
    // ones multiply exact zeros (finite operands): same bits up to the sign of a zero.
    const bool tri = a.causal && !a.R_init;

    if (wave >= NT) {
        // =============================================================================================== stream waves (chain_stream.h)
        const int NBLK = (a.nchunks + 63) >> 6;            // 64-chunk blocks per layer
        const ChainStreamGeom gm{a.attn, a.grad, l0, NBLK, 0, a.nchunks, H, NN, b, a.B, a.attn_bstride, a.nt, a.causal, N, a.row_magic};
        chain_stream_wave(gm, wave - NT, kGroupsWaves - NT, L * NBLK, lane, [&](int lg, int cidx, f32x4 mean) {
            chain_stream_deliver(smem + lg * NP * S, S, lds_cnt + lg * NT, cidx, lane, a.nchunks, N, NN, a.row_magic, mean);
        });
    } else {
        // =============================================================================================== matrix waves
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                float v = 0.f;
                if (row < N && col < N)
                    v = (a.R_init && g == 0) ? a.R_init[static_cast<int64_t>(b) * NN + row * N + col] : (row == col ? 1.f : 0.f);
                Rold[t][r] = v;
            }
        bool poisoned = false;
        for (int l = 0; l < L; ++l) {
            const float* Ab = smem + l * NP * S + (lane & 15) * S + rq;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                // rows [16 ti, 16 ti + 16) of A_bar_l are all this wave's tile ti needs: the product runs behind the stream waves tile
                // by tile, only the last tile of the group's last layer is exposed.  (Bounded like every wait here: the stream
                // waves of this workgroup deliver in microseconds or something is broken.)
                const unsigned expect = static_cast<unsigned>(max(0, min(N, ti * 16 + 16) - ti * 16) * N);
                int turns = 0;
                while (__hip_atomic_load(lds_cnt + l * NT + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expect &&
                       ++turns < (1 << 24))
                    __builtin_amdgcn_s_sleep(1);
                if (turns >= (1 << 24)) poisoned = true;        // never seen; if it happens the result says so (NaN), it does not lie
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (!(a.debug & 4)) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (tri && (t > ti || t < wave)) continue;      // (wave = this wave's column slab tj; wave-uniform)
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
        }
        float* dst = a.parts + (static_cast<int64_t>(b) * G + g) * NN;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                if (row < N && col < N)      // write-through: the hand-off below needs no L2 write-back fence
                    __hip_atomic_store(dst + row * N + col, poisoned ? __builtin_nanf("") : Rold[t][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    if (a.debug & 1) return;

    // ------------------------------------------------------------------ hand-off + combine
    // R = P_{G-1} . ... . P_1 . P_0 (P_0 already includes R_init): the sequential chain, re-associated at the group boundaries
    // (rounding-level difference, tests bound it at 1e-5).  Placement independent, no spinning: the LAST arriver of a sample multiplies.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0)
        *ticket_lds = __hip_atomic_fetch_add(a.counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *ticket_lds;
    if (ticket != static_cast<unsigned>(G - 1) || (a.debug & 8)) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (skipping it when all groups share an XCD gains nothing: measured)
    __syncthreads();
    // All 16 waves multiply: the NT x NT output tiles of P_gg . X are dealt round-robin (4 NT^3 exact-fp32 MFMAs spread over the four
    // matrix pipes instead of NT column-slab waves with two on one pipe); X ping-pongs between two LDS images; up to four partial
    // products are fetched at once (one memory round trip, not one per product).  The images' pads are still zero: the stream
    // waves wrote only elements of the N x N matrix, and so does everything below.
    const float* part = a.parts + static_cast<int64_t>(b) * G * NN;
    float* Pimg = smem;                    // [NP][S]  left operand P_gg
    float* X0 = smem + NP * S;             // [NP][S]  right operand, ping
    float* X1 = smem + 2 * NP * S;         // [NP][S]  pong
    constexpr int CE = (NP * NP + kGroupsThreads - 1) / kGroupsThreads;
    float pre[4][CE];
    auto prefetch = [&](int gg, float (&dst)[CE]) {
        const float* P = part + gg * NN;
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            const int idx = tid + i * kGroupsThreads;
            dst[i] = (idx < NN) ? P[idx] : 0.f;
        }
    };
    auto stash = [&](float* img, const float (&src)[CE]) {
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            const int idx = tid + i * kGroupsThreads;
            if (idx < NN) {
                const int row = static_cast<int>(__umulhi(static_cast<unsigned>(idx), a.row_magic));
                img[row * S + idx - row * N] = src[i];
            }
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q < G) prefetch(q, pre[q]);
    stash(X0, pre[0]);
    const int li = lane & 15;
#pragma unroll
    for (int gg = 1; gg < 8; ++gg) {       // G <= 8 (option range)
        if (gg >= G) break;
        stash(Pimg, pre[gg & 3]);
        lds_barrier();
        if (gg + 3 < G) prefetch(gg + 3, pre[(gg + 3) & 3]);        // (slot of P_{gg-1}: stashed an iteration ago)
        const float* Xc = (gg & 1) ? X0 : X1;
        float* Xn = (gg & 1) ? X1 : X0;
        for (int u = wave; u < NT * NT; u += kGroupsWaves) {
            const int ti = u / NT, tj = u - ti * NT;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* ap = Pimg + (ti * 16 + li) * S + rq;
            const float* bp = Xc + rq * S + tj * 16 + li;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (tri && (t > ti || t < tj)) continue;                // (tiles above the diagonal come out as the zeros they are)
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + t * 16);
                acc = mfma16x16x4(av[0], bp[(t * 16 + 0) * S], acc);
                acc = mfma16x16x4(av[1], bp[(t * 16 + 1) * S], acc);
                acc = mfma16x16x4(av[2], bp[(t * 16 + 2) * S], acc);
                acc = mfma16x16x4(av[3], bp[(t * 16 + 3) * S], acc);
            }
            float* xn = Xn + (ti * 16 + rq) * S + tj * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) xn[r * S] = acc[r];
        }
        lds_barrier();
    }
    const float* Xf = (G & 1) ? X0 : X1;
    for (int idx = tid; idx < NN; idx += kGroupsThreads) {
        const int row = static_cast<int>(__umulhi(static_cast<unsigned>(idx), a.row_magic));
        a.R_out[static_cast<int64_t>(b) * NN + idx] = Xf[row * S + idx - row * N];
    }
}

// ------------------------------------------------------------------------------------------------------------ host side
static size_t groups_lds_bytes(int nt, int per) {
    const int images = per > 3 ? per : 3;            // the combine uses three images
    return sizeof(float) * (static_cast<size_t>(images) * nt * 16 * (nt * 16 + 4) + static_cast<size_t>(per) * nt + 8);
}

// fp32 slabs, G > 1, every A_bar image of a group resident in LDS
bool self_chain_groups_applies(int n_layers, int G, int H, int N) {
    const int nt = (N + 15) / 16;
    if (nt > 8 || G < 2 || n_layers < G) return false;
    if (static_cast<size_t>(H) * N * N * 4 >= (1ull << 31)) return false;      // 32-bit offsets inside one buffer resource
    return groups_lds_bytes(nt, (n_layers + G - 1) / G) <= 160 * 1024;
}

template <int NT>
static int groups_launch(const GroupsArgs& r, hipStream_t s) {
    const size_t lds = groups_lds_bytes(NT, (r.n_layers + r.G - 1) / r.G);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(self_chain_groups_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    self_chain_groups_kernel<NT><<<r.B * r.G, kGroupsThreads, lds, s>>>(r);
    MMX_LAUNCH_CHECK("self_chain_groups_kernel");
    return MMX_OK;
}

int self_chain_groups_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N, int G,
                             int64_t attn_bstride, const void* R_init, void* R_out, unsigned* counters, float* parts, int nt_policy,
                             int debug, hipStream_t s) {
    if (static_cast<size_t>(H) * N * N * 4 >= (1ull << 31)) { set_error("self_chain_groups: head slabs beyond 2 GiB"); return MMX_ENOTSUP; }
    GroupsArgs r;
    memset(&r, 0, sizeof(r));
    for (int l = 0; l < n_layers; ++l) { r.attn[l] = attn_layers[l]; r.grad[l] = grad_layers[l]; }
    r.n_layers = n_layers; r.B = B; r.H = H; r.N = N; r.G = G;
    r.nchunks = (N * N + 3) / 4;
    r.row_magic = static_cast<unsigned>((0x100000000ull + N - 1) / N);
    r.R_init = static_cast<const float*>(R_init);
    r.R_out = static_cast<float*>(R_out);
    r.parts = parts;
    r.counters = counters;
    r.attn_bstride = attn_bstride;
    r.nt = nt_policy & 1;              // `nt_policy`: bit 0 = nt loads on the read-once slabs, bit 1 = MMX_CHAIN_CAUSAL
    r.causal = (nt_policy >> 1) & 1;
    r.debug = debug;
    int zrc = zero_async(counters, sizeof(unsigned) * B, s);
    if (zrc) return zrc;
    switch ((N + 15) / 16) {
        case 1: return groups_launch<1>(r, s);
        case 2: return groups_launch<2>(r, s);
        case 3: return groups_launch<3>(r, s);
        case 4: return groups_launch<4>(r, s);
        case 5: return groups_launch<5>(r, s);
        case 6: return groups_launch<6>(r, s);
        case 7: return groups_launch<7>(r, s);
        default: return groups_launch<8>(r, s);
    }
}

}  // namespace mmx
