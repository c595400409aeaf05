// K1r  self_chain_relay_kernel: the self-attention relevancy chain (rules 5 + 6) of ALL layers in ONE launch, in strict layer
// order, with the head reduction of a sample spread over Q workgroups -- no barrier, no LDS, no partial products.
//
//     R <- R_init or I;   for l = 0 .. L-1:   A_bar_l = mean_h clamp(G_l * A_l, 0);   R <- R + A_bar_l . R
//
// Reference sites: CLIP_explainability.ipynb cell 6:22-32 / 45-55, CLIP/example.py:22-30, ViT notebook cell 7:28-33,
// VisualBERT/.../ExplanationGenerator.py:86-93 (include/mmx_relevancy.h, mmx_relevancy_self_chain).
//
// The chain is an HBM stream (A and G read once: 2.7 FLOP/B at CLIP's text tower) followed, per layer, by a small product that
// needs the WHOLE A_bar_l.  One workgroup per sample leaves 3/4 of the chip idle at B = 64; the round-1..4 kernel cut the LAYERS
// of a sample into 4 groups and multiplied the partial products at the end -- a re-association of the reference's product and an
// exposed tail.  Here the POSITIONS are cut instead, and every wave is on its own:
//
//   stream waves   workgroup (b, q) owns positions [q, q + 1) * NN / Q of every layer of sample b.  A stream wave owns 64 of
//                  those 4-element chunks of ONE layer of the current round (a round = as many consecutive layers as the stream
//                  waves of a workgroup can cover), walks the heads IN ORDER with a register software pipeline of raw buffer
//                  loads (two sets of 4 heads x 2 arrays: 16 x 16 B per lane in flight -- the VGPR file holds 3x the bytes in
//                  flight that the LDS could: an LDS-DMA ring was built first and streamed at 3.1 TB/s, profiles/
//                  r05_chain_relay_probe.txt), writes its 64 chunks of A_bar_l to an L2-resident scratch with write-through
//                  stores, drains them and adds its chunk count to the arrival counter of (b, l).  Stream waves never wait.
//   chain waves    the product is independent per COLUMN block of R: chain wave (b, w) keeps the 16-column slab w of R in MFMA
//                  accumulators for the whole launch (the K1 trick: the B operand of v_mfma_f32_16x16x4_f32 is the accumulator
//                  register the lane already holds), waits for counter (b, l) to reach the chunk count of a layer, streams the
//                  A operand -- A_bar_l in 16-byte row pieces -- from the scratch through a 12-deep register ring and runs
//                  R[:, w] <- R[:, w] + A_bar_l . R[:, w].  The NT chain waves of a sample form a workgroup of their OWN, on a CU
//                  that runs no stream waves: a CU full of stream waves keeps ~200 KB of loads queued, and every request of a
//                  chain wave on that CU -- a counter poll, a piece of A_bar -- waits its turn behind them (~10 us a round trip:
//                  measured, the chain then only catches up after the streaming has ended).
//
// Same head order, same MFMA k order as the sequential per-sample kernel: BIT-IDENTICAL to it
// (tests/test_gpu_ops.py::test_self_chain_relay_bit_identical).  The launch is sized so that every workgroup is resident
// (B * (Q + 1) <= CUs, one 1024-thread workgroup per CU): chain workgroups only ever wait for stream workgroups, which wait for nothing.
#include "mmx_common.h"

#include <type_traits>

namespace mmx {

struct RelayArgs {
    const void* attn[MMX_MAX_LAYERS];
    const void* grad[MMX_MAX_LAYERS];
    int n_layers, B, H, N;
    int Q;         // workgroups per sample
    int nchunks;   // ceil(N*N / 4): 4-element chunks of one head slab
    int cpq;       // chunks per workgroup share
    int CWc;       // (unused: chain waves have workgroups of their own)
    int WPL;       // stream waves per (layer, share) = ceil(cpq / 64)
    int V;         // layers per round
    int BPW;       // 64-chunk blocks per stream wave and round
    int NB;        // A_bar ring slots of a chain workgroup in LDS (1..6)
    const float* R_init;
    float* R_out;
    float* abar;       // [B][L][nchunks * 4] (+ one row of slack) dense A_bar (scratch)
    unsigned* done;    // [B][L] chunks of A_bar_l that have arrived
    int64_t attn_bstride;   // H*N*N, or 0: one forward shared by the batch
    int nt;      // nt cache policy on the read-once slab stream
    int debug;   // profiling only: 1 = chain waves exit at once (stream + publish only), 2 = stream waves publish without streaming
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

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

constexpr int kRelayThreads = 1024;
constexpr int kRelayWaves = kRelayThreads / 64;

template <int NT>
__global__ __launch_bounds__(kRelayThreads) void self_chain_relay_kernel(const RelayArgs a) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int N = a.N, H = a.H, L = a.n_layers, Q = a.Q;
    const int NN = N * N;
    // workgroup -> (sample, role): the B * Q stream workgroups come first, then one chain workgroup per sample; a sample's
    // workgroups share id % 8, i.e. an XCD and its L2 (a speed hint only)
    const int groups = (a.B + 7) >> 3;
    const bool chain_wg = static_cast<int>(blockIdx.x) >= groups * 8 * Q;
    const int j = (chain_wg ? static_cast<int>(blockIdx.x) - groups * 8 * Q : static_cast<int>(blockIdx.x)) >> 3;
    const int b = (chain_wg ? j : j / Q) * 8 + (blockIdx.x & 7), q = chain_wg ? 0 : j % Q;
    if (b >= a.B) return;
    const int64_t Lb = static_cast<int64_t>(b) * L;

    if (!chain_wg) {
        // =============================================================================================== stream wave
        const int ws = wave, NWs = kRelayWaves;
        const int cq0 = q * a.cpq, cq1 = min(a.nchunks, cq0 + a.cpq);
        const int WPL = a.WPL, V = a.V, BPW = a.BPW;
        const int rounds = (L + V - 1) / V;
        const int HB = (H + 3) >> 2;                       // batches of 4 heads
        const int hstride = NN * 4;                        // bytes between heads
        const float fH = static_cast<float>(H);
        const int64_t sampleG = static_cast<int64_t>(b) * H * NN * 4, sampleA = static_cast<int64_t>(b) * a.attn_bstride * 4;
        const auto rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.abar), 0, 0x7fffffff, kRawBufferFlags);
        if (cq0 >= cq1 || (a.debug & 2)) {
            // nothing to reduce in this share (more workgroups than chunks); profiling (2): the chain must still see full counters
            if (ws == 0 && q == 0 && (a.debug & 2))
                for (int l = lane; l < L; l += 64)
                    __hip_atomic_fetch_add(a.done + Lb + l, static_cast<unsigned>(a.nchunks), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (ws >= V * WPL) return;                         // (more stream waves than blocks in a round: nothing for this one, ever)
        const int64_t restG = static_cast<int64_t>(a.B - b) * H * NN * 4;
        const int bytesG = static_cast<int>(restG < 0x7fffffff ? restG : 0x7fffffff), bytesA = a.attn_bstride ? bytesG : H * NN * 4;
        // item = one 64-chunk block of one layer; a wave's items: (round r, block slot s) -> block ws + s * NWs of the round's
        // V * WPL blocks -> layer r * V + blk / WPL, chunks cq0 + 64 * (blk % WPL) + lane
        struct Item { int l, c; bool wave_on, lane_on; };
        auto item_of = [&](int it) -> Item {
            const int r = it / BPW, s = it - r * BPW;
            const int blk = ws + s * NWs, dl = blk / WPL;
            const int l = r * V + dl, c = cq0 + (blk - dl * WPL) * 64 + lane;
            const bool on = dl < V && l < L;
            return Item{min(l, L - 1), min(c, cq1 - 1), on, on && c < cq1};
        };
        const int items = rounds * BPW;
        const int total = items * HB;
        // Software pipeline over the flat batch sequence k = item * HB + hb (4 heads x 2 arrays = 8 x 16 B per batch): batch k + 1
        // is ISSUED before batch k is reduced.  Every issue is unconditional (a batch past the end, a head past H or a lane past
        // the share repeats a valid address and is reduced with weight 0): with conditional issues the compiler merges the wait
        // counts of the two paths and drains everything before each reduction.
        auto issue = [&](int k, u32x4 (&av)[4], u32x4 (&gv)[4], auto aux_tag) {
            constexpr int AUXG = decltype(aux_tag)::value & 2, AUXA = (decltype(aux_tag)::value & 1) ? 0 : AUXG;
            const int it = k / HB, hb = k - it * HB;
            const Item im = item_of(it);
            const int lu = __builtin_amdgcn_readfirstlane(im.l);
            // the resources end exactly at the end of the tensors and the whole offset is in the VGPR: the 16-byte load of the last,
            // partial chunk of the last head of the last sample (N^2 % 4 != 0) reads its out-of-range dwords as zero instead of
            // touching memory behind the slab
            const auto rA = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(a.attn[lu]) + sampleA)), 0, bytesA, kRawBufferFlags);
            const auto rG = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(a.grad[lu]) + sampleG)), 0, bytesG, kRawBufferFlags);
            const unsigned voff = static_cast<unsigned>(im.c) * 16u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned off = voff + static_cast<unsigned>(min(hb * 4 + u, H - 1) * hstride);
                av[u] = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, AUXA);
                gv[u] = __builtin_amdgcn_raw_buffer_load_b128(rG, off, 0, AUXG);
            }
        };
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        auto consume = [&](int k, bool live, const u32x4 (&av)[4], const u32x4 (&gv)[4]) {
            const int it = k / HB, hb = k - it * HB;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = (live && hb * 4 + u < H) ? 1.f : 0.f;      // heads in ascending order: the sequential sum
                const f32x4 x = __builtin_bit_cast(f32x4, gv[u]) * __builtin_bit_cast(f32x4, av[u]);
                s[0] += relu_nan(x[0]) * w; s[1] += relu_nan(x[1]) * w;
                s[2] += relu_nan(x[2]) * w; s[3] += relu_nan(x[3]) * w;
            }
            if (live && hb == HB - 1) {
                const Item im = item_of(it);
                if (im.wave_on) {
                    if (im.lane_on) {
                        const f32x4 m = s / fH;
                        // write-through (sc1): the arrival tick below then needs no L2 write-back fence (cdna guide G16, R1 form)
                        const int64_t byte = ((Lb + im.l) * a.nchunks + im.c) * 16;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rS, static_cast<unsigned>(byte), 0, 16);
                    }
                    // this wave's pieces of A_bar_l must be in L2 / memory before the tick: loads and stores may retire out of
                    // order with respect to each other, so the wave drains (the prefetched batch is needed next anyway)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long m = __ballot(im.lane_on);
                    if (lane == 0 && m)
                        __hip_atomic_fetch_add(a.done + Lb + im.l, static_cast<unsigned>(__popcll(m)), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                }
                s = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto run = [&](auto aux_tag) {
            u32x4 a0[4], g0[4], a1[4], g1[4];
            issue(0, a0, g0, aux_tag);
            for (int k = 0; k < total; k += 2) {
                issue(min(k + 1, total - 1), a1, g1, aux_tag);
                consume(k, true, a0, g0);
                issue(min(k + 2, total - 1), a0, g0, aux_tag);
                consume(min(k + 1, total - 1), k + 1 < total, a1, g1);
            }
        };
        // aux tag: 0 default policy | 2 nt on both slabs | 3 nt on the gradient slab only (the batch shares the probabilities)
        if (!a.nt) run(std::integral_constant<int, 0>{});
        else if (a.attn_bstride == 0) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 2>{});
        return;
    }

    // =================================================================================================== chain workgroup
    // CW = min(8, NT^2) waves; the NT x NT output tiles of A_bar_l . R are dealt round-robin to them, so that the exact-fp32 MFMAs of a
    // layer (4 NT^3 of 32 cycles) spread evenly over the four matrix pipes of the CU (5 column-slab waves would put two on one pipe).
    // R lives in LDS (two buffers of NP rows x NP + 4 floats, zero padded: ping-pong per layer), A_bar_l in a ring of NB dense copies.
    constexpr int NP = NT * 16, SR = NP + 4, NTILE = NT * NT;
    constexpr int CW = NTILE < 8 ? NTILE : 8;
    if (wave >= CW || (a.debug & 1)) return;                 // (the other waves of the 1024-thread footprint leave: barriers below count CW waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char relay_smem[];
    const int NB = a.NB;                                     // A_bar ring slots
    const int CPI = (a.nchunks + 63) >> 6;                   // 1 KiB copy instructions per layer
    const int buf_bytes = CPI * 1024;
    unsigned char* abuf = relay_smem;
    volatile unsigned* poll_lds = reinterpret_cast<volatile unsigned*>(abuf + NB * buf_bytes);     // 256 B: where a counter poll lands
    volatile int* avail_lds = reinterpret_cast<volatile int*>(abuf + NB * buf_bytes + 256);        // wave 0 -> all: layer l has landed
    float* Rbuf = reinterpret_cast<float*>(abuf + NB * buf_bytes + 512);                           // 2 x NP x SR floats
    const int ctid = wave * 64 + lane;
    for (int i = ctid; i < 2 * NP * SR; i += CW * 64) Rbuf[i] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = ctid; i < NN; i += CW * 64) {
        const int row = i / N, cc = i - row * N;
        Rbuf[row * SR + cc] = a.R_init ? a.R_init[static_cast<int64_t>(b) * NN + i] : (row == cc ? 1.f : 0.f);
    }
    // from here on wave 0 counts its own vector memory operations (the R_init loads above have been consumed by the LDS stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Wave 0 is also the LOADER of its workgroup.  It keeps one counter poll (lane i polls the counter of layer avail + i: a round of
    // the stream waves completes several layers at once) and up to NB - 2 layer copies in flight, all of them LDS-DMA
    // (global_load_lds_dword / _dwordx4 sc1: A_bar, dense, 16 bytes x nchunks, lane-linear into the ring): no register is a destination
    // of anything in flight, and since vector memory operations of one kind retire in issue order, "operation X has landed" is
    // s_waitcnt vmcnt(operations issued after X).  All of it is inline asm: the compiler neither counts nor drains these operations (a
    // compiler-visible poll would be waited for with vmcnt(0), i.e. together with every copy behind it).  One s_barrier per layer: the
    // loader has seen layer l land; everybody has finished layer l - 1 (its ring slot may be refilled, its R buffer is complete).
    // MFMA operands, k visited as (t, r, lane >> 4) like the per-sample kernel: A = A_bar[16 ti + (lane & 15)][16 t + 4 (lane >> 4) + r]
    // (dense rows of N floats; only the LAST tile row / column reaches past N: one row mask and four column masks, as BIT masks -- what
    // lies behind a row is the next row, behind the matrix anything), B = R[16 t + 4 (lane >> 4) + r][16 tj + (lane & 15)].
    const unsigned expect = static_cast<unsigned>(a.nchunks);
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    unsigned cmask[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cmask[e] = ((NT - 1) * 16 + g4 + e < N) ? 0xffffffffu : 0u;
    const unsigned rmask = ((NT - 1) * 16 + li < N) ? 0xffffffffu : 0u;
    const int lane_row = min(li, N - 1) * N + g4, last_row = min((NT - 1) * 16 + li, N - 1) * N + g4;
    int avail = 0, copied = 0;                               // loader: layers known complete / copy issued (prefixes)
    int issued = 0;                                          // loader: vector memory operations issued so far
    int copy_seq[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // `issued` right after the copy into ring slot i
    int poll_seq = 0, poll_base = 0;
    bool poll_out = false;
    auto wait_for = [&](int seq) { wait_vmcnt(min(issued - seq, 63)); };
    // loader step: harvest / issue the poll, issue the copies the ring has room for (slots of layers < busy_from are free)
    auto loader_step = [&](int busy_from) {
        if (poll_out) {
            wait_for(poll_seq);
            poll_out = false;
            const unsigned long long okm = __ballot(poll_lds[lane] >= expect);
            while (avail < L && avail - poll_base < 64 && ((okm >> (avail - poll_base)) & 1ull)) ++avail;
        }
        if (avail < L) {
            poll_base = avail;
            const unsigned* pp = a.done + Lb + min(avail + lane, L - 1);
            const unsigned ldst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(abuf + NB * buf_bytes))));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(pp), "s"(ldst) : "memory");
            poll_seq = ++issued;
            poll_out = true;
        }
        while (copied < avail && copied - busy_from < NB) {
            const char* src = reinterpret_cast<const char*>(a.abar + (Lb + copied) * a.nchunks * 4);
            const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(abuf + (copied % NB) * buf_bytes)));
            for (int i = 0; i < CPI; ++i) {
                const char* gsrc = src + static_cast<size_t>(min(i * 64 + lane, a.nchunks - 1)) * 16;
                const unsigned ldst = __builtin_amdgcn_readfirstlane(lds0 + i * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
            }
            issued += CPI;
            copy_seq[copied % NB] = issued;
            ++copied;
        }
    };
    for (int l = 0; l < L; ++l) {
        if (wave == 0) {
            // (copies may go into every ring slot except the one of layer l - 1, which the other waves may still be reading)
            const int busy_from = l > 0 ? l - 1 : 0;
            // every wait in this kernel is bounded: ~seconds of polling without the stream waves delivering means something outside
            // this launch is wrong -- give up (the result is then poisoned below) instead of hanging the device
            int turns = 0;
            loader_step(busy_from);
            while (copied <= l && ++turns < (1 << 21)) {
                __builtin_amdgcn_s_sleep(4);
                loader_step(busy_from);
            }
            if (copied > l) wait_for(copy_seq[l % NB]);      // layer l has landed (younger copies and the poll stay in flight)
            if (lane == 0) *avail_lds = copied > l ? l + 1 : -1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (*avail_lds < 0) break;
        const float* Ab = reinterpret_cast<const float*>(abuf + (l % NB) * buf_bytes);
        const float* Rc = Rbuf + (l & 1) * NP * SR;
        float* Rn = Rbuf + ((l + 1) & 1) * NP * SR;
        for (int u = wave; u < NTILE; u += CW) {
            const int ti = u / NT, tj = u - ti * NT;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* rowp = Ab + (ti == NT - 1 ? last_row : lane_row + ti * 16 * N);
            const float* bp = Rc + g4 * SR + tj * 16 + li;
            const bool last_ti = ti == NT - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x4 raw;
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(rowp[t * 16 + e]);
                if (t == NT - 1) { raw[0] &= cmask[0]; raw[1] &= cmask[1]; raw[2] &= cmask[2]; raw[3] &= cmask[3]; }
                if (last_ti) { raw[0] &= rmask; raw[1] &= rmask; raw[2] &= rmask; raw[3] &= rmask; }
                const f32x4 av = __builtin_bit_cast(f32x4, raw);
                acc = mfma16x16x4(av[0], bp[(t * 16 + 0) * SR], acc);
                acc = mfma16x16x4(av[1], bp[(t * 16 + 1) * SR], acc);
                acc = mfma16x16x4(av[2], bp[(t * 16 + 2) * SR], acc);
                acc = mfma16x16x4(av[3], bp[(t * 16 + 3) * SR], acc);
            }
            // R + (A_bar . R): the reference's association; lane holds rows 16 ti + 4 (lane >> 4) + r, column 16 tj + (lane & 15)
            const float* ro = Rc + (ti * 16 + g4) * SR + tj * 16 + li;
            float* rn = Rn + (ti * 16 + g4) * SR + tj * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) rn[r * SR] = ro[r * SR] + acc[r];
        }
    }
    if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool failed = *avail_lds < 0;
    const float* Rf = Rbuf + (L & 1) * NP * SR;
    float* dst = a.R_out + static_cast<int64_t>(b) * NN;
    for (int i = ctid; i < NN; i += CW * 64) {
        const int row = i / N, cc = i - row * N;
        dst[i] = failed ? __builtin_nanf("") : Rf[row * SR + cc];
    }
}

// ------------------------------------------------------------------------------------------------------------ host side
static int g_relay_q = 0;     // option "self_chain_relay_q": workgroups per sample (0 = auto: fill the CUs once)
void chain_relay_options(int q, int d) { if (q >= 0) g_relay_q = q; (void)d; }

static size_t relay_counter_bytes(int B, int L) { return ((sizeof(unsigned) * static_cast<size_t>(B) * L) + 255) & ~static_cast<size_t>(255); }

// Launch geometry for a shape; returns false when the relay form does not apply (the per-sample kernel then runs)
static bool relay_plan(int n_layers, int B, int H, int N, RelayArgs* out) {
    const int nt = (N + 15) / 16;
    if (nt > 8 || n_layers < 1 || B < 1) return false;
    RelayArgs r;
    memset(&r, 0, sizeof(r));
    r.n_layers = n_layers; r.B = B; r.H = H; r.N = N;
    r.nchunks = (N * N + 3) / 4;
    const int cus = device_cu_count();
    const int groups8 = (B + 7) / 8 * 8;                           // samples are dealt to the XCDs in groups of 8
    // every workgroup resident, one 1024-thread workgroup per CU: B * Q stream workgroups + B chain workgroups (which wait for them)
    int Q = cus / groups8 - 1;
    if (Q < 1) return false;
    if (g_relay_q > 0 && g_relay_q < Q) Q = g_relay_q;
    if (Q > 8) Q = 8;
    if (Q > r.nchunks / 64) Q = r.nchunks / 64;                    // a share is at least one wave of chunks
    if (Q < 1) Q = 1;
    r.Q = Q;
    r.cpq = (r.nchunks + Q - 1) / Q;
    r.CWc = 0;
    const int nws = kRelayWaves;
    r.WPL = (r.cpq + 63) / 64;
    r.V = nws / r.WPL;
    if (r.V < 1) r.V = 1;
    if (r.V > n_layers) r.V = n_layers;
    r.BPW = (r.V * r.WPL + nws - 1) / nws;
    // the chain workgroup's LDS: A_bar ring (dense copies of a layer, >= 2 slots: the one being multiplied + one to copy into) + two R
    // buffers of NP x (NP + 4) floats; beyond ~100 tokens that no longer fits a CU and the per-sample kernel runs
    const size_t per = static_cast<size_t>((r.nchunks + 63) / 64) * 1024;
    const size_t rb = 2 * static_cast<size_t>(nt * 16) * (nt * 16 + 4) * sizeof(float);
    if (rb + 2 * per + 512 > 160 * 1024) return false;
    r.NB = static_cast<int>((160 * 1024 - 512 - rb) / per);
    if (r.NB > 6) r.NB = 6;
    *out = r;
    return true;
}

bool self_chain_relay_applies(int n_layers, int B, int H, int N) {
    RelayArgs r;
    return relay_plan(n_layers, B, H, N, &r);
}

size_t self_chain_relay_workspace(int n_layers, int B, int H, int N) {
    RelayArgs r;
    if (!relay_plan(n_layers, B, H, N, &r)) return 0;
    // + slack: the chain waves' 16-byte pieces of the last rows reach past the dense N x N matrix
    return relay_counter_bytes(B, n_layers) + sizeof(float) * (static_cast<size_t>(B) * n_layers * r.nchunks * 4 + 256);
}

template <int NT>
static int relay_launch(const RelayArgs& r, hipStream_t s) {
    const int groups = (r.B + 7) / 8;
    const size_t lds = static_cast<size_t>(r.NB) * ((r.nchunks + 63) / 64) * 1024 + 512 +
                       2 * static_cast<size_t>(NT * 16) * (NT * 16 + 4) * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(self_chain_relay_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    self_chain_relay_kernel<NT><<<groups * 8 * (r.Q + 1), kRelayThreads, lds, s>>>(r);
    MMX_LAUNCH_CHECK("self_chain_relay_kernel");
    return MMX_OK;
}

int self_chain_relay_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N,
                            int64_t attn_bstride, const void* R_init, void* R_out, void* workspace, size_t workspace_bytes,
                            int nt_policy, int debug, hipStream_t s) {
    RelayArgs r;
    if (!relay_plan(n_layers, B, H, N, &r)) { set_error("self_chain_relay: shape not supported"); return MMX_ENOTSUP; }
    const size_t need = self_chain_relay_workspace(n_layers, B, H, N);
    if (workspace_bytes < need || !workspace) {
        set_error("mmx_relevancy_self_chain: workspace %zu < %zu", workspace_bytes, need);
        return MMX_EWORKSPACE;
    }
    if (static_cast<size_t>(B) * n_layers * r.nchunks * 16 >= (1ull << 31) || static_cast<size_t>(H) * N * N * 4 >= (1ull << 31)) {
        set_error("self_chain_relay: slabs beyond the 2 GiB offset range of one buffer resource");
        return MMX_ENOTSUP;
    }
    for (int l = 0; l < n_layers; ++l) { r.attn[l] = attn_layers[l]; r.grad[l] = grad_layers[l]; }
    r.R_init = static_cast<const float*>(R_init);
    r.R_out = static_cast<float*>(R_out);
    r.done = static_cast<unsigned*>(workspace);
    r.abar = reinterpret_cast<float*>(static_cast<char*>(workspace) + relay_counter_bytes(B, n_layers));
    r.attn_bstride = attn_bstride;
    r.nt = nt_policy;
    r.debug = debug;
    int zrc = zero_async(r.done, sizeof(unsigned) * static_cast<size_t>(B) * n_layers, s);
    if (zrc) return zrc;
    switch ((N + 15) / 16) {
        case 1: return relay_launch<1>(r, s);
        case 2: return relay_launch<2>(r, s);
        case 3: return relay_launch<3>(r, s);
        case 4: return relay_launch<4>(r, s);
        case 5: return relay_launch<5>(r, s);
        case 6: return relay_launch<6>(r, s);
        case 7: return relay_launch<7>(r, s);
        default: return relay_launch<8>(r, s);
    }
}

}  // namespace mmx
