// K1-big: the all-layer relevancy chain  R <- R + A_bar_l . R  (rules 5 + 6) for N > 128 in ONE launch.
//
// Replaces, for the long-sequence bodies (ViT-B/16 N = 197, ViT-L/14@336 N = 577, DETR encoder N = 850..1050), the split
// path of relevancy_kernels.hip (per layer: avg_heads launch + bmm launch, A_bar and R round-tripping through HBM) and
// with it the reference's per-layer loops: ViT notebook cell 7:27-33, CLIP_explainability.ipynb cell 6:22-32,
// DETR/modules/ExplanationGenerator.py:110-118.
//
// R of one sample no longer fits a CU (N = 577: 1.3 MB), so a sample is owned by a TEAM of T = ceil(ceil(N/16) / 4)
// workgroups of 4 waves; wave w of member m owns the 16-column slab 4m + w of R:
//
//   * R_old lives in REGISTERS for the whole layer in the MFMA C/D layout (row = 16t + 4(lane>>4) + r, col = lane&15).
//     As in K1 the contraction index is visited as (t, r, lane>>4), so the register R_old[t][r] IS the B operand of
//     v_mfma_f32_16x16x4_f32 -- R never goes through LDS.  R_new cannot be held next to it (N = 950 is 240 registers
//     per copy), so every finished 16 x 16 tile R_new = R_old + A_bar . R_old is stored to the output matrix -- the
//     wave's own columns, nobody else reads them -- and the slab is re-read (L2) into the registers for the next layer.
//   * phase 1 of a layer: the T members reduce A_bar_l = mean_h clamp(G . A, 0) cooperatively, each streaming 1/T of the
//     [H, N, N] slabs ONCE (16-byte loads, heads in order), into a per-team fp32 scratch image that stays in L2;
//   * team barrier: ONE monotonic counter per team (plain stores -> barrier -> lane-0 agent-scope release -> relaxed
//     add; consumer: one relaxed poll loop -> agent-scope acquire -> barrier: cdna guide G16).  The grid is PERSISTENT and
//     sized by the occupancy query, so every member of every team is resident; teams loop over samples.  Spins are
//     bounded (a status word records a timeout instead of hanging the GPU).
//   * phase 2: every member streams the full A_bar_l image through LDS in 16-row tiles (next tile in flight in
//     registers while the MFMAs of the current one run) and multiplies it with its R slab.
//
// HBM traffic = A and G once (+ R out); A_bar (N^2 fp32 per sample and layer) is written once and re-read T times from L2.
// Team members are taken from ONE XCD when the grid allows it (block b runs on XCD b % 8) so that those re-reads stay
// in that XCD's L2 -- a speed hint only, correctness never depends on placement.
#include "mmx_common.h"

#include <type_traits>

namespace mmx {

struct ChainBigArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    int T, nteams, xcd_map;
    const float* R_init;
    float* R_out;
    float* abar;          // [nteams][2][N][NS]
    unsigned* counters;   // [nteams] monotonic arrival counters (zeroed by the host before the launch)
    unsigned* status;     // [1] != 0: a bounded spin timed out (results are garbage, the GPU did not hang)
    int64_t attn_bstride;
    int debug;   // profiling only: 1 no phase-1 streaming, 2 no team wait, 4 no MFMAs, 8 no tile staging, 16 no R tile load / store, 32 no slab load
};

constexpr int kBigThreads = 256;

template <int NTC, int DT>
__global__ __launch_bounds__(kBigThreads, (NTC <= 40 ? 2 : 1)) void self_chain_big_kernel(const ChainBigArgs a) {
    constexpr int NS = NTC * 16;          // padded row length of the A_bar image
    constexpr int S = NS + 8;             // LDS row stride: = 8 (mod 16), conflict-free ds_read_b128 of the A operand
    constexpr int CPT = NS / 64;          // 16-byte chunks of a 16-row tile per thread (16 * NS / 4 / 256)
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 tiles of 16 x S

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int N = a.N, H = a.H, T = a.T;
    const int NT = (N + 15) >> 4;
    const int64_t NN = static_cast<int64_t>(N) * N;
    int team, member;
    if (a.xcd_map) {            // members of a team = consecutive blocks of ONE XCD (block b runs on XCD b % 8)
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
        team = x + 8 * (k / T);
        member = k % T;
    } else {
        team = blockIdx.x / T;
        member = blockIdx.x - team * T;
    }
    const int ct = member * 4 + wave;                 // column tile of R this wave owns
    const bool active = ct < NT;
    const int col = ct * 16 + c16;
    const bool colv = active && col < N;
    float* abar0 = a.abar + static_cast<int64_t>(team) * 2 * N * NS;
    unsigned* counter = a.counters + team;
    const float fH = static_cast<float>(H);
    const int nchunks = static_cast<int>((NN + 3) >> 2);
    const int per = (nchunks + T - 1) / T;
    const int c_lo = min(nchunks, member * per), c_hi = min(nchunks, c_lo + per);
    unsigned gl = 0;                                   // layers finished by this team (all samples)

    f32x4 Rold[NTC];
    for (int b = team; b < a.B; b += a.nteams) {
        float* Rb = a.R_out + static_cast<int64_t>(b) * NN;
        const float* Rin = a.R_init ? a.R_init + static_cast<int64_t>(b) * NN : nullptr;
        // Slab loader.  Row base (16t + r) * N is wave-uniform (scalar address arithmetic), the lane part 4g * N + col is
        // ONE 32-bit register.  `gq` is laundered through an empty asm at every use site: otherwise hipcc hoists the
        // NTC * 4 per-element predicates and 64-bit addresses out of the sample / layer loops and spills them (KBs of
        // scratch per lane).  mode 0: identity, 1: R_init (plain loads), 2: the output matrix (nt loads are served by L2:
        // the wave re-reads its own stores of the previous layer; an agent-scope atomic load would do too, but hipcc waits
        // for each of those individually).
        auto load_slab = [&](auto mode_c) {
            constexpr int mode = decltype(mode_c)::value;
            int gq = 4 * g;
            asm volatile("" : "+v"(gq));
            const int voff = gq * N + col;
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t * 16 < N) {                                   // wave-uniform
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rb = t * 16 + r;                   // row = rb + gq
                        const bool ok = colv && gq < N - rb;
                        float x = 0.f;
                        if (mode == 0) x = (rb + gq == col) ? 1.f : 0.f;
                        else if (mode == 1) x = (Rin + static_cast<int64_t>(rb) * N)[ok ? voff : 0];
                        else x = __builtin_nontemporal_load(Rb + static_cast<int64_t>(rb) * N + (ok ? voff : 0));
                        v[r] = ok ? x : 0.f;
                    }
                }
                Rold[t] = v;
                __builtin_amdgcn_sched_barrier(0);   // one tile's addresses at a time (no pile-up of 64-bit address registers)
            }
        };
        const int64_t sample = static_cast<int64_t>(b) * H * NN;
        const int64_t sampleA = static_cast<int64_t>(b) * a.attn_bstride;

        for (int l = 0; l < a.n_layers; ++l, ++gl) {
            float* Ag = abar0 + static_cast<int64_t>(gl & 1u) * N * NS;
            // ---------------------------------------------------------------- phase 1: this member's share of A_bar_l
            {
                const void* A = a.attn[l];
                const void* Gr = a.grad[l];
                // two chunks per pass and 8 heads per load batch: 32 loads in flight per lane (a CU has only 4-8 waves
                // here, against K1's 11 stream waves; with 8 loads per lane the stream ran at ~8 GB/s per CU)
                auto write_chunk = [&](int64_t p, const f32x4& sv) {
                    int row = static_cast<int>(p / N);
                    int cc = static_cast<int>(p - static_cast<int64_t>(row) * N);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (p + e < NN) Ag[static_cast<int64_t>(row) * NS + cc] = sv[e] / fH;
                        if (++cc == N) { cc = 0; ++row; }
                    }
                };
                auto slow_chunk = [&](int64_t p) {
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
                    for (int e = 0; p + e < NN && e < 4; ++e)
                        for (int h = 0; h < H; ++h)
                            sv[e] += relu_nan(load1_as_f32<DT>(Gr, sample + h * NN + p + e) *
                                              load1_as_f32<DT>(A, sampleA + h * NN + p + e));
                    write_chunk(p, sv);
                };
                for (int c = c_lo + tid; c < c_hi && !(a.debug & 1); c += 2 * kBigThreads) {
                    const bool two = c + kBigThreads < c_hi;
                    const int64_t p0 = static_cast<int64_t>(c) * 4;
                    const int64_t p1 = static_cast<int64_t>(two ? c + kBigThreads : c) * 4;
                    if (p0 + 5 < NN && p1 + 5 < NN) {       // (+5: the 16-bit loader over-reads two elements)
                        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
                        constexpr int UB = 8;                      // heads per load batch: 4 * UB loads in flight per lane
                        for (int h0 = 0; h0 < H; h0 += UB) {
                            stream_raw<DT> ra0[UB], rg0[UB], ra1[UB], rg1[UB];
#pragma unroll
                            for (int u = 0; u < UB; ++u) {          // heads in order; a batch past H re-reads head H-1 (unused)
                                const int64_t hh = static_cast<int64_t>(min(h0 + u, H - 1)) * NN;
                                ra0[u] = load4_stream_raw<DT>(A, sampleA + hh + p0);
                                rg0[u] = load4_stream_raw<DT>(Gr, sample + hh + p0);
                                ra1[u] = load4_stream_raw<DT>(A, sampleA + hh + p1);
                                rg1[u] = load4_stream_raw<DT>(Gr, sample + hh + p1);
                            }
                            __builtin_amdgcn_sched_barrier(0);      // every load of the batch is issued before the first use
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                if (h0 + u < H) {
                                    const int64_t hh = static_cast<int64_t>(h0 + u) * NN;
                                    const f32x4 x0 = stream_cvt<DT>(rg0[u], sample + hh + p0) * stream_cvt<DT>(ra0[u], sampleA + hh + p0);
                                    const f32x4 x1 = stream_cvt<DT>(rg1[u], sample + hh + p1) * stream_cvt<DT>(ra1[u], sampleA + hh + p1);
                                    s0[0] += relu_nan(x0[0]); s0[1] += relu_nan(x0[1]);
                                    s0[2] += relu_nan(x0[2]); s0[3] += relu_nan(x0[3]);
                                    s1[0] += relu_nan(x1[0]); s1[1] += relu_nan(x1[1]);
                                    s1[2] += relu_nan(x1[2]); s1[3] += relu_nan(x1[3]);
                                }
                            }
                        }
                        write_chunk(p0, s0);
                        if (two) write_chunk(p1, s1);
                    } else {
                        slow_chunk(p0);
                        if (two) slow_chunk(p1);
                    }
                }
            }
            // ---------------------------------------------------------------- team barrier: all of A_bar_l is published
            __syncthreads();                                   // every wave's stores are issued and drained (vmcnt(0))
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (gl + 1u) * static_cast<unsigned>(T);
                unsigned spins = 0;
                while (!(a.debug & 2) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 24)) {                // ~1 s: a member never arrived
                        __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();

            // ---------------------------------------------------------------- phase 2: R_new = R_old + A_bar_l . R_old
            constexpr int HC = CPT / 2;                         // the next tile travels in two halves (register budget)
            f32x4 pre[HC];
            auto fetch_half = [&](int ti, int half) {          // 16 rows of the A_bar image -> registers (zero padded)
#pragma unroll
                for (int k = 0; k < HC; ++k) {
                    const int idx = tid + (half * HC + k) * kBigThreads;
                    const int row = idx / (NS / 4), c4 = (idx - row * (NS / 4)) * 4;
                    const int grow = ti * 16 + row;
                    const bool ok = grow < N && c4 < N;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Ag + (ok ? static_cast<int64_t>(grow) * NS + c4 : 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!ok || c4 + e >= N) v[e] = 0.f;    // columns >= N of the image are never written
                    pre[k] = v;
                }
            };
            auto put_half = [&](float* dst, int half) {
#pragma unroll
                for (int k = 0; k < HC; ++k) {
                    const int idx = tid + (half * HC + k) * kBigThreads;
                    const int row = idx / (NS / 4), c4 = (idx - row * (NS / 4)) * 4;
                    *reinterpret_cast<f32x4*>(dst + row * S + c4) = pre[k];
                }
            };
            // R_old of this layer: identity / R_init, or what this wave stored during the previous layer (its stores were
            // drained by the barriers above).  Loaded HERE, after the streaming phase, so that phase 1 has the whole
            // register file for loads in flight.
            if (a.debug & 32) load_slab(std::integral_constant<int, 0>{});
            else if (l > 0) load_slab(std::integral_constant<int, 2>{});
            else if (Rin) load_slab(std::integral_constant<int, 1>{});
            else load_slab(std::integral_constant<int, 0>{});
            fetch_half(0, 0);
            put_half(smem, 0);
            fetch_half(0, 1);
            put_half(smem, 1);
            lds_barrier();
            for (int ti = 0; ti < NT; ++ti) {
                const float* Ab = smem + (ti & 1) * 16 * S + c16 * S + 4 * g;
                float* nxt = smem + ((ti + 1) & 1) * 16 * S;
                const bool stage = ti + 1 < NT && !(a.debug & 8);
                if (stage) fetch_half(ti + 1, 0);        // in flight while the first half of the MFMAs runs
                if (active) {
                    // R_old tile (ti) of this wave's slab: the register copy cannot be indexed at run time; read it
                    // back from the output matrix (layer 0: identity / R_init), where the previous layer stored it
                    f32x4 old;
                    int gq = 4 * g;
                    asm volatile("" : "+v"(gq));
                    const int voff = gq * N + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rb = ti * 16 + r;
                        const bool ok = colv && gq < N - rb;
                        float x;
                        if (a.debug & 16) x = 0.5f;
                        else if (l == 0) x = Rin ? (Rin + static_cast<int64_t>(rb) * N)[ok ? voff : 0] : ((rb + gq == col) ? 1.f : 0.f);
                        else x = __builtin_nontemporal_load(Rb + static_cast<int64_t>(rb) * N + (ok ? voff : 0));
                        old[r] = ok ? x : 0.f;
                    }
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    auto mfma_range = [&](int t_lo, int t_hi) {
#pragma unroll
                        for (int t = 0; t < NTC; t += 2) {
                            if (t < t_lo || t >= t_hi) continue;
                            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + t * 16);
                            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + (t + 1) * 16);
                            acc0 = mfma16x16x4(a0[0], Rold[t][0], acc0);
                            acc1 = mfma16x16x4(a1[0], Rold[t + 1][0], acc1);
                            acc0 = mfma16x16x4(a0[1], Rold[t][1], acc0);
                            acc1 = mfma16x16x4(a1[1], Rold[t + 1][1], acc1);
                            acc0 = mfma16x16x4(a0[2], Rold[t][2], acc0);
                            acc1 = mfma16x16x4(a1[2], Rold[t + 1][2], acc1);
                            acc0 = mfma16x16x4(a0[3], Rold[t][3], acc0);
                            acc1 = mfma16x16x4(a1[3], Rold[t + 1][3], acc1);
                        }
                    };
                    constexpr int TH = (NTC / 2) & ~1;
                    if (!(a.debug & 4)) mfma_range(0, TH);
                    if (stage) {
                        put_half(nxt, 0);
                        fetch_half(ti + 1, 1);
                    }
                    if (!(a.debug & 4)) mfma_range(TH, NTC);
                    const f32x4 out = old + (acc0 + acc1);      // R + (A_bar . R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rb = ti * 16 + r;
                        if (colv && gq < N - rb && (!(a.debug & 16) || out[r] == 12345.f)) (Rb + static_cast<int64_t>(rb) * N)[voff] = out[r];
                    }
                }
                if (!active && stage) {                  // idle waves still carry their share of the staging
                    put_half(nxt, 0);
                    fetch_half(ti + 1, 1);
                }
                if (stage) put_half(nxt, 1);
                lds_barrier();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
// 2: run this kernel whenever the shape is eligible; 0 / 1: the per-layer split path.  Measured on MI355X
// (profiles/r02_chain_big_probe.txt) the split path -- two full-chip, high-occupancy kernels per layer -- is 1.7-2x FASTER
// at every shape tried (ViT-B/16, ViT-L/14@336 at B = 16..128, DETR encoder), so the one-launch kernel is opt-in.
static int g_chain_big = 1;
static int g_chain_big_debug = 0;
void chain_big_enable(int on) { g_chain_big = on & 3; g_chain_big_debug = on >> 8; }

static int ntc_for(int N) {
    const int nt = (N + 15) / 16;
    const int steps[] = {16, 24, 32, 40, 48, 64, 72};
    for (int s : steps)
        if (nt <= s) return s;
    return 0;
}

static size_t align256b(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// upper bound on resident workgroups of this kernel on an MI355X (256 CUs x 2): the workspace is sized with it so that
// the query needs no device
constexpr int kBigMaxResident = 512;

size_t self_chain_big_workspace(int B, int N) {
    const int ntc = ntc_for(N);
    if (!ntc) return 0;
    const int T = ((N + 15) / 16 + 3) / 4;
    int nteams = kBigMaxResident / T;
    if (nteams > B) nteams = B;
    if (nteams < 1) nteams = 1;
    return 256 + align256b(sizeof(unsigned) * nteams) +
           sizeof(float) * static_cast<size_t>(nteams) * 2 * N * (static_cast<size_t>(ntc) * 16);
}

template <int NTC>
static int launch_big(ChainBigArgs& args, int dtype, void* workspace, hipStream_t s) {
    void (*kern)(const ChainBigArgs) = nullptr;
    switch (dtype) {
        case MMX_F32: kern = self_chain_big_kernel<NTC, MMX_F32>; break;
        case MMX_F16: kern = self_chain_big_kernel<NTC, MMX_F16>; break;
        case MMX_BF16: kern = self_chain_big_kernel<NTC, MMX_BF16>; break;
        default: set_error("self_chain: unsupported dtype %d", dtype); return MMX_EINVAL;
    }
    const size_t lds = sizeof(float) * 2 * 16 * (NTC * 16 + 8);
    hipError_t e;
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    // persistent grid: every workgroup must be resident (team members wait for each other)
    int occ = 0, dev = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), kBigThreads, lds);
    if (e != hipSuccess) return hip_fail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(MultiprocessorCount)");
    if (occ > 2) occ = 2;                                   // register budget of the kernel (launch bounds): never more
    int capacity = occ * cus;
    if (capacity > kBigMaxResident) capacity = kBigMaxResident;
    int nteams = capacity / args.T;
    if (nteams > args.B) nteams = args.B;
    if (nteams < 1) {
        set_error("self_chain_big: a team of %d workgroups does not fit the device (%d resident)", args.T, capacity);
        return MMX_ENOTSUP;
    }
    args.xcd_map = 0;
    if (nteams >= 8) {
        nteams &= ~7;
        args.xcd_map = 1;
    }
    args.nteams = nteams;
    char* ws = static_cast<char*>(workspace);
    args.status = reinterpret_cast<unsigned*>(ws);
    args.counters = reinterpret_cast<unsigned*>(ws + 256);
    args.abar = reinterpret_cast<float*>(ws + 256 + align256b(sizeof(unsigned) * nteams));
    int zrc = zero_async(ws, 256 + align256b(sizeof(unsigned) * nteams), s);
    if (zrc) return zrc;
    kern<<<nteams * args.T, kBigThreads, lds, s>>>(args);
    MMX_LAUNCH_CHECK("self_chain_big_kernel");
    return MMX_OK;
}

// returns 1 if the fused long-sequence kernel was launched (rc in *rc_out), 0 if this shape takes the split path
int self_chain_big_try(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N,
                       int dtype, int64_t attn_bstride, const void* R_init, void* R_out, void* workspace,
                       size_t workspace_bytes, hipStream_t s, int* rc_out) {
    const int ntc = ntc_for(N);
    if (!g_chain_big || !ntc || N <= 128 || n_layers < 1) return 0;
    const int T = ((N + 15) / 16 + 3) / 4;
    if (g_chain_big != 2) return 0;
    const size_t need = self_chain_big_workspace(B, N);
    if (workspace_bytes < need || !workspace) {
        set_error("mmx_relevancy_self_chain: workspace %zu < %zu", workspace_bytes, need);
        *rc_out = MMX_EWORKSPACE;
        return 1;
    }
    ChainBigArgs args;
    memset(&args, 0, sizeof(args));
    for (int l = 0; l < n_layers; ++l) { args.attn[l] = attn_layers[l]; args.grad[l] = grad_layers[l]; }
    args.n_layers = n_layers; args.B = B; args.H = H; args.N = N; args.T = T;
    args.R_init = static_cast<const float*>(R_init);
    args.R_out = static_cast<float*>(R_out);
    args.attn_bstride = attn_bstride;
    args.debug = g_chain_big_debug;
    switch (ntc) {
        case 16: *rc_out = launch_big<16>(args, dtype, workspace, s); break;
        case 24: *rc_out = launch_big<24>(args, dtype, workspace, s); break;
        case 32: *rc_out = launch_big<32>(args, dtype, workspace, s); break;
        case 40: *rc_out = launch_big<40>(args, dtype, workspace, s); break;
        case 48: *rc_out = launch_big<48>(args, dtype, workspace, s); break;
        case 64: *rc_out = launch_big<64>(args, dtype, workspace, s); break;
        default: *rc_out = launch_big<72>(args, dtype, workspace, s); break;
    }
    return 1;
}

}  // namespace mmx
