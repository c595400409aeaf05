// K1r  self_chain_relay_kernel: the self-attention relevancy chain (rules 5 + 6) of ALL layers in ONE launch, strict layer
// order, with the head reduction of a sample spread over Q "streamer" workgroups that feed one "chain" workgroup.
//
//     R <- R_init or I;   for l = 0 .. L-1:   A_bar_l = mean_h clamp(G_l * A_l, 0);   R <- R + A_bar_l . R
//
// Reference sites: CLIP_explainability.ipynb cell 6:22-32 / 45-55, CLIP/example.py:22-30, ViT notebook cell 7:28-33,
// VisualBERT/.../ExplanationGenerator.py:86-93 (include/mmx_relevancy.h, mmx_relevancy_self_chain).
//
// Why this shape.  The chain is an HBM stream (A and G read once, 2.7 FLOP/B at CLIP's text
// tower) followed, per layer, by a tiny product that needs the WHOLE A_bar_l.  One workgroup per sample leaves 3/4 of the
// chip idle at B = 64; the round-1..4 kernel therefore cut the LAYERS of a sample into 4 groups and multiplied the partial
// products at the end -- a re-association of the reference's product and an exposed tail.  Here the
// POSITIONS of every layer are cut instead: streamer q of sample b reduces positions [q, q+1) * NN/Q of every layer and
// publishes its piece of A_bar_l (write-through stores, one arrival counter per (sample, layer)); the chain workgroup
// of the sample follows the counters layer by layer with R in MFMA accumulators (the K1 trick: the B operand of
// v_mfma_f32_16x16x4_f32 is the accumulator register the lane already holds).  Same summation order as the sequential
// per-sample kernel: BIT-IDENTICAL to it (tests/test_gpu_ops.py::test_self_chain_relay_bit_identical).
//
// Streamer = LDS-DMA ring.  One loader wave issues global_load_lds_dwordx4 (nt) for whole (head, array) fragments --
// up to 7 KiB CONTIGUOUS per fragment, against 1-2 KiB per wave instruction from 8 interleaved streams in the
// register-pipelined kernel -- into a ring of D slots; CW consumer waves read a landed slot with ds_read_b128, multiply,
// clamp and add the heads IN ORDER into one f32x4 per lane.  One s_barrier per round; the DMA stays in flight across it.
// Fragments are fetched from the 16-byte-aligned address below their first element (head slabs of odd N^2 start on
// 4/8/12-byte offsets); the consumers undo the shift with a second b128 read and a wave-uniform select.
//
// Roles are taken at run time: every workgroup of a sample draws a ticket; tickets 0 .. Q-1 stream piece `ticket`, the
// LAST one to start (ticket Q) runs the chain.  The chain workgroup therefore only ever waits for workgroups that are
// already running and that wait for nothing themselves: no assumption about dispatch order, placement or residency.
#include "mmx_common.h"

namespace mmx {

struct RelayArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    int Q;         // streamers per sample
    int nchunks;   // ceil(N*N / 4): 4-element chunks of one head slab
    int cpq;       // chunks per streamer
    int SB, sbc;   // sub-blocks of a streamer's share, chunks per sub-block (<= 64 * CW)
    int CW;        // consumer waves
    int IPF;       // DMA instructions (1 KiB each) per (head, array) fragment = ceil((sbc + 1) / 64)
    int HPR;       // heads per round (divides H)
    int D;         // ring slots
    int threads;
    unsigned row_magic;   // ceil(2^32 / N)
    const float* R_init;
    float* R_out;
    float* abar;       // [B][L][nchunks * 4] dense A_bar pieces (scratch)
    unsigned* start;   // [B] role tickets
    unsigned* done;    // [B][L] arrivals of A_bar_l pieces
    int64_t attn_bstride;   // H*N*N, or 0: one forward shared by the batch
    int nt;      // nt cache policy on the read-once slab stream
    int debug;   // profiling only: 1 = chain workgroups exit at once (stream + publish only), 2 = streamers publish without streaming
};

constexpr int kRelayMaxThreads = 512;
constexpr int kRelayHeader = 16;   // bytes in front of the ring / A_bar image: the role ticket

__device__ __forceinline__ void wait_vmcnt(int n) {
    // s_waitcnt takes an immediate: a jump table over the 6-bit field
    switch (n) {
#define MMX_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        MMX_W(0) MMX_W(1) MMX_W(2) MMX_W(3) MMX_W(4) MMX_W(5) MMX_W(6) MMX_W(7) MMX_W(8) MMX_W(9) MMX_W(10) MMX_W(11) MMX_W(12)
        MMX_W(13) MMX_W(14) MMX_W(15) MMX_W(16) MMX_W(17) MMX_W(18) MMX_W(19) MMX_W(20) MMX_W(21) MMX_W(22) MMX_W(23) MMX_W(24)
        MMX_W(25) MMX_W(26) MMX_W(27) MMX_W(28) MMX_W(29) MMX_W(30) MMX_W(31) MMX_W(32) MMX_W(33) MMX_W(34) MMX_W(35) MMX_W(36)
        MMX_W(37) MMX_W(38) MMX_W(39) MMX_W(40) MMX_W(41) MMX_W(42) MMX_W(43) MMX_W(44) MMX_W(45) MMX_W(46) MMX_W(47) MMX_W(48)
        MMX_W(49) MMX_W(50) MMX_W(51) MMX_W(52) MMX_W(53) MMX_W(54) MMX_W(55) MMX_W(56) MMX_W(57) MMX_W(58) MMX_W(59) MMX_W(60)
        MMX_W(61) MMX_W(62)
#undef MMX_W
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// 4 floats from dword `d` of an LDS fragment: one b128 read when d is a multiple of 4 (s == 0, wave-uniform), else four b32 reads
// (the fragment was fetched from the aligned address below its first element: head slabs of odd N^2 start at 4 / 8 / 12 bytes)
__device__ __forceinline__ f32x4 lds_read4(const unsigned char* frag_base, int j, int s) {
    if (s == 0) return reinterpret_cast<const f32x4*>(frag_base)[j];
    const float* f = reinterpret_cast<const float*>(frag_base) + 4 * j + s;
    return f32x4{f[0], f[1], f[2], f[3]};
}

// One (head, array) fragment: IPF LDS-DMA instructions of 1 KiB each from the 16-byte-aligned address `base`; every instruction runs
// with all 64 lanes (lanes past the fragment repeat its last unit), so the instruction count per round is exact -- the counted vmcnt
// waits of the loader rely on it.  Raw buffer form: the address is one SGPR resource + a lane offset (2 VALU per instruction).
template <int AUX>
__device__ __forceinline__ void dma_fragment(const char* base, unsigned nunits, unsigned char* dst, int ipf, int lane) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(base)), 0, 0x7fffffff, kRawBufferFlags);
    for (int i = 0; i < ipf; ++i) {
        const unsigned u = min(static_cast<unsigned>(i * 64 + lane), nunits - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 1024), 16, u * 16u, 0, 0, AUX);
    }
}

template <int NT>
__global__ __launch_bounds__(kRelayMaxThreads, 4) void self_chain_relay_kernel(const RelayArgs a) {
    constexpr int NP = NT * 16;
    constexpr int S = NP + 4;
    constexpr bool PRE = NT <= 6;   // A_bar double-buffered in LDS (<= 77 KB): the next layer's piece loads fly under the MFMAs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* ticket_lds = reinterpret_cast<unsigned*>(smem_raw);
    unsigned char* body = smem_raw + kRelayHeader;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int N = a.N, H = a.H, L = a.n_layers, Q = a.Q;
    const int NN = N * N;

    // workgroup -> sample.  Streamer candidates come first (ids < groups*8*Q), chain candidates last, and a sample's
    // workgroups share id % 8, i.e. an XCD and its L2 (a speed hint only: the hand-off is agent-scope).
    const int groups = (a.B + 7) >> 3;
    int b;
    {
        const int id = blockIdx.x, nstream = groups * 8 * Q;
        if (id < nstream) { const int j = id >> 3; b = (j / Q) * 8 + (id & 7); }
        else b = id - nstream;
    }
    if (b >= a.B) return;

    if (tid == 0) *ticket_lds = __hip_atomic_fetch_add(a.start + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(static_cast<int>(*ticket_lds));
    const int64_t Lb = static_cast<int64_t>(b) * L;

    if (ticket < Q) {
        // =============================================================================================== streamer
        const int cq0 = ticket * a.cpq, cq1 = min(a.nchunks, cq0 + a.cpq);
        const int HPR = a.HPR, HB = H / HPR, SB = a.SB, IPF = a.IPF, D = a.D;
        const int R = (cq0 < cq1 && !(a.debug & 2)) ? L * SB * HB : 0;
        const int IPR = HPR * 2 * IPF;                     // DMA instructions per round
        const int frag = IPF * 1024, slot_bytes = HPR * 2 * frag;
        if (wave > a.CW) return;
        if (R == 0) {   // nothing to reduce (more streamers than chunks): the arrivals are still owed
            if (wave == 0)
                for (int l = lane; l < L; l += 64)
                    __hip_atomic_fetch_add(a.done + Lb + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const uintptr_t offG = static_cast<uintptr_t>(b) * H * NN * 4, offA = static_cast<uintptr_t>(b) * a.attn_bstride * 4;
        // Round r = (layer l, sub-block sb, head block hb), hb fastest.  Both roles walk the rounds with counters (no
        // division per round: a round is ~0.4 us of streaming, a few dependent integer divisions are a third of that).
        if (wave == 0) {
            // ------------------------------------------------------------------------------------------- loader
            int il = 0, isb = 0, ihb = 0, islot = 0;              // the NEXT fill to issue
            uintptr_t pA = 0, pG = 0;
            auto layer_ptrs = [&](int l) {
                pA = reinterpret_cast<uintptr_t>(a.attn[l]) + offA;
                pG = reinterpret_cast<uintptr_t>(a.grad[l]) + offG;
            };
            layer_ptrs(0);
            // the gradient slab is read exactly once per launch; the probability slab as well unless the batch shares one
            // forward (then every sample re-reads it from L2: default policy)
            const bool ntG = a.nt != 0, ntA = a.nt != 0 && a.attn_bstride != 0;
            auto issue = [&]() {
                // (a shorter last share can leave its last sub-block empty: fetch its first chunk instead -- nobody reads it --
                // so that every address stays inside the slab)
                const int cs = min(cq0 + isb * a.sbc, cq1 - 1), ce = max(cs + 1, min(cq1, cq0 + (isb + 1) * a.sbc));
                const unsigned nvalid = static_cast<unsigned>(min(4 * (ce - cs), NN - 4 * cs));       // >= 1: cs < nchunks
                unsigned char* dst = body + islot * slot_bytes;
                unsigned hoff = static_cast<unsigned>(ihb * HPR) * NN + 4u * cs;
                for (int hh = 0; hh < HPR; ++hh, hoff += NN) {
                    {
                        const unsigned al = static_cast<unsigned>(pA >> 2) & 3u, E = al + hoff;      // elements above a 16-byte boundary
                        const char* src = reinterpret_cast<const char*>(pA - 4u * al) + static_cast<size_t>(E >> 2) * 16;
                        const unsigned nunits = ((E & 3u) + nvalid + 3u) >> 2;
                        if (ntA) dma_fragment<2>(src, nunits, dst, IPF, lane); else dma_fragment<0>(src, nunits, dst, IPF, lane);
                        dst += frag;
                    }
                    {
                        const unsigned al = static_cast<unsigned>(pG >> 2) & 3u, E = al + hoff;
                        const char* src = reinterpret_cast<const char*>(pG - 4u * al) + static_cast<size_t>(E >> 2) * 16;
                        const unsigned nunits = ((E & 3u) + nvalid + 3u) >> 2;
                        if (ntG) dma_fragment<2>(src, nunits, dst, IPF, lane); else dma_fragment<0>(src, nunits, dst, IPF, lane);
                        dst += frag;
                    }
                }
                if (++ihb == HB) { ihb = 0; if (++isb == SB) { isb = 0; if (++il < L) layer_ptrs(il); } }
                if (++islot == D) islot = 0;
            };
            const int pro = min(D - 1, R);
            for (int r = 0; r < pro; ++r) issue();
            for (int r = 0; r < R; ++r) {
                const int issued = min(r + D - 1, R);
                wait_vmcnt((issued - 1 - r) * IPR);                 // fill r has landed
                asm volatile("s_barrier" ::: "memory");             // consumers: slot r is yours; slot r-1 is free again
                if (r + D - 1 < R) issue();
            }
            asm volatile("s_barrier" ::: "memory");                 // pairs with the consumers' closing barrier
        } else {
            // ----------------------------------------------------------------------------------------- consumers
            const int j = (wave - 1) * 64 + lane;                   // this lane's chunk inside the sub-block
            const float fH = static_cast<float>(H);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.abar), 0, 0x7fffffff, kRawBufferFlags);
            int publish = -1;                                       // layer whose pieces are stored and wait for their arrival tick
            int l = 0, sb = 0, hb = 0, slot_i = 0;
            unsigned alA = 0, alG = 0;
            auto layer_align = [&](int ll) {
                alA = static_cast<unsigned>((reinterpret_cast<uintptr_t>(a.attn[ll]) + offA) >> 2) & 3u;
                alG = static_cast<unsigned>((reinterpret_cast<uintptr_t>(a.grad[ll]) + offG) >> 2) & 3u;
            };
            layer_align(0);
            for (int r = 0; r < R; ++r) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (publish >= 0) {
                    // every consumer wave drained its stores before the barrier it has just passed
                    if (wave == 1 && lane == 0)
                        __hip_atomic_fetch_add(a.done + Lb + publish, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    publish = -1;
                }
                const int cs = cq0 + sb * a.sbc, ce = min(cq1, cs + a.sbc);
                const bool mine = j < ce - cs;
                const unsigned char* slot = body + slot_i * slot_bytes;
                if (mine) {
                    unsigned off = static_cast<unsigned>(hb * HPR) * NN + 4u * cs;
                    for (int hh = 0; hh < HPR; ++hh, off += NN) {
                        const f32x4 av = lds_read4(slot + (hh * 2) * frag, j, static_cast<int>((alA + off) & 3u));
                        const f32x4 gv = lds_read4(slot + (hh * 2 + 1) * frag, j, static_cast<int>((alG + off) & 3u));
                        const f32x4 x = gv * av;
                        acc[0] += relu_nan(x[0]); acc[1] += relu_nan(x[1]);
                        acc[2] += relu_nan(x[2]); acc[3] += relu_nan(x[3]);
                    }
                }
                const bool last_hb = hb == HB - 1, last_sb = sb == SB - 1;
                if (last_hb) {
                    if (mine) {
                        const f32x4 m = acc / fH;
                        // write-through (sc1): the arrival tick below then needs no L2 write-back fence (cdna guide G16, R1 form)
                        const int64_t byte = ((Lb + l) * a.nchunks + cs + j) * 16;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rs, static_cast<unsigned>(byte), 0, 16);
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (last_sb) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of A_bar_l are in L2 / memory
                        publish = l;
                    }
                }
                if (++hb == HB) { hb = 0; if (++sb == SB) { sb = 0; if (++l < L) layer_align(l); } }
                if (++slot_i == D) slot_i = 0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (publish >= 0 && wave == 1 && lane == 0)
                __hip_atomic_fetch_add(a.done + Lb + publish, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    // =================================================================================================== chain
    if (wave >= NT || (a.debug & 1)) return;
    float* Ab0 = reinterpret_cast<float*>(body);
    constexpr int NBUF = PRE ? 2 : 1;
    constexpr int CT = NT * 64;                                         // chain threads
    for (int i = tid; i < NBUF * NP * S; i += CT) Ab0[i] = 0.f;       // the pads must read as 0
    const int col = wave * 16 + (lane & 15);
    const int rq = (lane >> 4) * 4;
    f32x4 Rold[NT], Rnew[NT];
    if (a.R_init) {
        const float* R0 = a.R_init + static_cast<int64_t>(b) * NN;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + rq + r;
                Rold[t][r] = (row < N && col < N) ? R0[row * N + col] : 0.f;
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Rold[t][r] = (t * 16 + rq + r == col && col < N) ? 1.f : 0.f;
    }
    constexpr int CH = NT;   // 16-byte pieces of A_bar per chain lane: ceil(ceil(N*N/4) / (64 NT)) <= NT
    u32x4 pre[CH];
    auto ready = [&](int l) -> bool {
        const unsigned v = __hip_atomic_load(a.done + Lb + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane(v) >= static_cast<unsigned>(Q);
    };
    auto wait_ready = [&](int l) {
        while (!ready(l)) __builtin_amdgcn_s_sleep(1);
    };
    auto fetch = [&](int l) {   // sc1 loads: the pieces were stored write-through by other CUs, possibly of another XCD
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.abar) + (Lb + l) * a.nchunks * 4, 0, 0x7fffffff, kRawBufferFlags);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = min(tid + i * CT, a.nchunks - 1);
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, static_cast<unsigned>(k) * 16u, 0, 16);
        }
    };
    const unsigned magic = a.row_magic;                                // ceil(2^32 / N): row = (p * magic) >> 32, exact for p < N*N + 8
    auto stash = [&](int buf) {
        float* Ab = Ab0 + buf * NP * S;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = tid + i * CT;
            if (k < a.nchunks) {
                const f32x4 v = __builtin_bit_cast(f32x4, pre[i]);
                const int p = k * 4;
                int row = static_cast<int>(__umulhi(static_cast<unsigned>(p), magic)), cc = p - row * N;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p + e < NN) Ab[row * S + cc] = v[e];
                    if (++cc == N) { cc = 0; ++row; }
                }
            }
        }
    };
    auto chain_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    chain_barrier();            // zero fill visible
    if (L > 0) { wait_ready(0); fetch(0); stash(0); }
    chain_barrier();
    constexpr int TI = NT <= 5 ? 2 : 1;   // output tiles in flight: two accumulator chains cover the 40-cycle MFMA latency (registers permitting)
    for (int l = 0; l < L; ++l) {
        const int buf = PRE ? (l & 1) : 0;
        bool fetched = false;
        if (PRE && l + 1 < L && ready(l + 1)) { fetch(l + 1); fetched = true; }
        const float* Ab = Ab0 + buf * NP * S + (lane & 15) * S + rq;
#pragma unroll
        for (int ti = 0; ti < NT; ti += TI) {
            f32x4 acc[TI];
#pragma unroll
            for (int q = 0; q < TI; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int q = 0; q < TI; ++q) {
                    if (ti + q < NT) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + (ti + q) * 16 * S + t * 16);
                        acc[q] = mfma16x16x4(av[0], Rold[t][0], acc[q]);
                        acc[q] = mfma16x16x4(av[1], Rold[t][1], acc[q]);
                        acc[q] = mfma16x16x4(av[2], Rold[t][2], acc[q]);
                        acc[q] = mfma16x16x4(av[3], Rold[t][3], acc[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < TI; ++q)
                if (ti + q < NT) Rnew[ti + q] = Rold[ti + q] + acc[q];      // R + (A_bar . R): the reference's association
            // keep the scheduler from hoisting all NT*NT operand reads (100+ VGPRs) above the first MFMA: two workgroups share a
            // CU only under 128 registers per lane
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) Rold[t] = Rnew[t];
        if (l + 1 < L) {
            if (!PRE) chain_barrier();          // single buffer: every wave is done reading A_bar_l
            if (!fetched) { wait_ready(l + 1); fetch(l + 1); }
            stash(PRE ? ((l + 1) & 1) : 0);
            chain_barrier();
        }
    }
    float* dst = a.R_out + static_cast<int64_t>(b) * NN;
    int rq2 = rq, col2 = col;
    asm volatile("" : "+v"(rq2), "+v"(col2));   // recompute the store indices here instead of keeping 4 NT of them (and their predicates) live
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = t * 16 + rq2 + r;
            if (row < N && col2 < N) dst[row * N + col2] = Rold[t][r];
        }
}

// ------------------------------------------------------------------------------------------------------------ host side
static int g_relay_q = 0;     // option "self_chain_relay_q": streamers per sample (0 = auto: fill the CUs once)
static int g_relay_d = 0;     // option "self_chain_relay_d": ring slots (0 = auto)
void chain_relay_options(int q, int d) { if (q >= 0) g_relay_q = q; if (d >= 0) g_relay_d = d; }

static size_t relay_counter_bytes(int B, int L) { return ((sizeof(unsigned) * static_cast<size_t>(B) * (1 + L)) + 255) & ~static_cast<size_t>(255); }

// Launch geometry for a shape; returns false when the relay form does not apply (the per-sample kernel then runs)
static bool relay_plan(int n_layers, int B, int H, int N, RelayArgs* out, size_t* lds_out) {
    const int nt = (N + 15) / 16;
    if (nt > 8 || n_layers < 1) return false;
    RelayArgs r;
    memset(&r, 0, sizeof(r));
    r.n_layers = n_layers; r.B = B; r.H = H; r.N = N;
    r.nchunks = (N * N + 3) / 4;
    int Q = g_relay_q;
    if (Q <= 0) {
        const int cus = device_cu_count();
        Q = (cus + B / 2) / B;                       // streamers fill the CUs once
    }
    if (Q > 16) Q = 16;
    if (Q > r.nchunks / 64) Q = r.nchunks / 64;      // a streamer's share is at least one wave of chunks
    if (Q < 1) Q = 1;
    r.Q = Q;
    r.cpq = (r.nchunks + Q - 1) / Q;
    r.SB = (r.cpq + 447) / 448;
    r.sbc = (r.cpq + r.SB - 1) / r.SB;
    r.CW = (r.sbc + 63) / 64;
    r.IPF = (r.sbc + 1 + 63) / 64;
    const int frag = r.IPF * 1024;
    r.HPR = 1;
    for (int h = 1; h <= H; ++h)
        if (H % h == 0 && h * 2 * frag <= 16 * 1024 && h * 2 * r.IPF <= 21) r.HPR = h;   // <= 16 KiB slots, >= 3 fills under the vmcnt field
    const int slot = r.HPR * 2 * frag, ipr = r.HPR * 2 * r.IPF;
    int D = g_relay_d > 0 ? g_relay_d : 6;
    const int ring_budget = 72 * 1024;
    if (D > ring_budget / slot) D = ring_budget / slot;
    if (D > 1 + 63 / ipr) D = 1 + 63 / ipr;          // (D - 2) * ipr outstanding instructions must fit the 6-bit vmcnt field ...
    while (D > 2 && (D - 1) * ipr > 63) --D;         // ... and so must the D - 1 fills of the prologue
    if (D < 2) return false;
    r.D = D;
    const int NP = nt * 16, S = NP + 4;
    const size_t chain_lds = sizeof(float) * (nt <= 6 ? 2 : 1) * NP * S;
    const size_t ring_lds = static_cast<size_t>(D) * slot;
    *lds_out = kRelayHeader + (chain_lds > ring_lds ? chain_lds : ring_lds);
    const int waves = (1 + r.CW) > nt ? (1 + r.CW) : nt;
    r.threads = waves * 64;
    r.row_magic = static_cast<unsigned>((0x100000000ull + N - 1) / N);
    if (r.threads > kRelayMaxThreads) return false;
    *out = r;
    return true;
}

bool self_chain_relay_applies(int n_layers, int B, int H, int N) {
    RelayArgs r; size_t lds;
    return relay_plan(n_layers, B, H, N, &r, &lds);
}

size_t self_chain_relay_workspace(int n_layers, int B, int H, int N) {
    RelayArgs r; size_t lds;
    if (!relay_plan(n_layers, B, H, N, &r, &lds)) return 0;
    return relay_counter_bytes(B, n_layers) + sizeof(float) * static_cast<size_t>(B) * n_layers * r.nchunks * 4;
}

template <int NT>
static int relay_launch(const RelayArgs& r, size_t lds, hipStream_t s) {
    auto kern = self_chain_relay_kernel<NT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    const int groups = (r.B + 7) / 8;
    const int grid = groups * 8 * r.Q + r.B;
    kern<<<grid, r.threads, lds, s>>>(r);
    MMX_LAUNCH_CHECK("self_chain_relay_kernel");
    return MMX_OK;
}

int self_chain_relay_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N,
                            int64_t attn_bstride, const void* R_init, void* R_out, void* workspace, size_t workspace_bytes,
                            int nt_policy, int debug, hipStream_t s) {
    RelayArgs r; size_t lds;
    if (!relay_plan(n_layers, B, H, N, &r, &lds)) { set_error("self_chain_relay: shape not supported"); return MMX_ENOTSUP; }
    const size_t need = self_chain_relay_workspace(n_layers, B, H, N);
    if (workspace_bytes < need || !workspace) {
        set_error("mmx_relevancy_self_chain: workspace %zu < %zu", workspace_bytes, need);
        return MMX_EWORKSPACE;
    }
    if (static_cast<size_t>(B) * n_layers * r.nchunks * 16 >= (1ull << 31)) { set_error("self_chain_relay: A_bar scratch beyond 2 GiB"); return MMX_ENOTSUP; }
    for (int l = 0; l < n_layers; ++l) { r.attn[l] = attn_layers[l]; r.grad[l] = grad_layers[l]; }
    r.R_init = static_cast<const float*>(R_init);
    r.R_out = static_cast<float*>(R_out);
    r.start = static_cast<unsigned*>(workspace);
    r.done = r.start + B;
    r.abar = reinterpret_cast<float*>(static_cast<char*>(workspace) + relay_counter_bytes(B, n_layers));
    r.attn_bstride = attn_bstride;
    r.nt = nt_policy;
    r.debug = debug;
    int zrc = zero_async(r.start, sizeof(unsigned) * static_cast<size_t>(B) * (1 + n_layers), s);
    if (zrc) return zrc;
    switch ((N + 15) / 16) {
        case 1: return relay_launch<1>(r, lds, s);
        case 2: return relay_launch<2>(r, lds, s);
        case 3: return relay_launch<3>(r, lds, s);
        case 4: return relay_launch<4>(r, lds, s);
        case 5: return relay_launch<5>(r, lds, s);
        case 6: return relay_launch<6>(r, lds, s);
        case 7: return relay_launch<7>(r, lds, s);
        default: return relay_launch<8>(r, lds, s);
    }
}

}  // namespace mmx
