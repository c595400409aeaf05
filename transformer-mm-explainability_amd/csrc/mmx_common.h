// Shared device/host helpers for libmmx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mmx_relevancy.h"

namespace mmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16-byte global load that only assumes 4-byte alignment: head slabs of odd N (77^2 floats) start on
// 4/8/12-byte offsets.  gfx950 global_load_dwordx4 needs dword alignment only.
struct __attribute__((packed, aligned(4))) f32x4_u { f32x4 v; };
__device__ __forceinline__ f32x4 ldg4_u(const float* p) { return reinterpret_cast<const f32x4_u*>(p)->v; }

// A pointer the caller knows to be wave-uniform, pinned to SGPRs (a buffer resource must live in SGPRs: a base the compiler
// cannot prove uniform would be wrapped in a waterfall loop).
__device__ __forceinline__ const char* sgpr_ptr(const void* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
    return reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRawBufferFlags = 0x00020000;   // word 3 of a raw (untyped, stride 0) buffer resource on gfx9 / CDNA: DATA_FORMAT = 32

struct __attribute__((packed, aligned(2))) u16x4_u { unsigned short v[4]; };

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(static_cast<unsigned int>(b) << 16);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) {
    return __half2float(__ushort_as_half(b));
}

// load 4 consecutive captured values (any alignment >= element size) as fp32
template <int DT>
__device__ __forceinline__ f32x4 load4_as_f32(const void* base, int64_t idx);
template <>
__device__ __forceinline__ f32x4 load4_as_f32<MMX_F32>(const void* base, int64_t idx) {
    return ldg4_u(static_cast<const float*>(base) + idx);
}
template <>
__device__ __forceinline__ f32x4 load4_as_f32<MMX_BF16>(const void* base, int64_t idx) {
    u16x4_u r = *reinterpret_cast<const u16x4_u*>(static_cast<const unsigned short*>(base) + idx);
    f32x4 o;
    o[0] = bf16_bits_to_f32(r.v[0]); o[1] = bf16_bits_to_f32(r.v[1]);
    o[2] = bf16_bits_to_f32(r.v[2]); o[3] = bf16_bits_to_f32(r.v[3]);
    return o;
}
template <>
__device__ __forceinline__ f32x4 load4_as_f32<MMX_F16>(const void* base, int64_t idx) {
    u16x4_u r = *reinterpret_cast<const u16x4_u*>(static_cast<const unsigned short*>(base) + idx);
    f32x4 o;
    o[0] = f16_bits_to_f32(r.v[0]); o[1] = f16_bits_to_f32(r.v[1]);
    o[2] = f16_bits_to_f32(r.v[2]); o[3] = f16_bits_to_f32(r.v[3]);
    return o;
}
// Same for a latency-sensitive streaming loop (few waves per CU): 16-bit slabs whose element index is ODD are only 2-byte
// aligned, which the generic helper above has to fetch as four 2-byte loads (a slab of odd N^2 starts every second head
// on such an offset).  Here the three ALIGNED dwords that cover the chunk are loaded instead (one dwordx3).
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) u32x3_u { u32x3 v; };
// Split into the LOAD (raw registers, no dependent instruction and NO branch: a caller can issue a whole batch before
// the first use) and the CONVERSION.  16-bit slabs: always the three aligned dwords from element (idx & ~1) -- for an
// even idx the third one is an over-read of two elements, so the caller must guarantee idx + 5 is still inside the
// tensor -- and the chunk is cut out with a funnel shift by 0 or 2 bytes.
template <int DT> struct stream_raw { u32x3 v; };
template <> struct stream_raw<MMX_F32> { f32x4 v; };
template <int DT>
__device__ __forceinline__ stream_raw<DT> load4_stream_raw(const void* base, int64_t idx) {
    stream_raw<DT> r;
    if constexpr (DT == MMX_F32) {
        r.v = ldg4_u(static_cast<const float*>(base) + idx);
    } else {
        const unsigned short* p = static_cast<const unsigned short*>(base) + (idx & ~static_cast<int64_t>(1));
        r.v = reinterpret_cast<const u32x3_u*>(p)->v;
    }
    return r;
}
template <int DT>
__device__ __forceinline__ f32x4 stream_cvt(const stream_raw<DT>& r, int64_t idx) {
    if constexpr (DT == MMX_F32) {
        return r.v;
    } else {
        const unsigned sh = (idx & 1) ? 2u : 0u;
        const unsigned lo = __builtin_amdgcn_alignbyte(r.v[1], r.v[0], sh);
        const unsigned hi = __builtin_amdgcn_alignbyte(r.v[2], r.v[1], sh);
        f32x4 o;
        if constexpr (DT == MMX_BF16) {
            o[0] = __uint_as_float(lo << 16); o[1] = __uint_as_float(lo & 0xffff0000u);
            o[2] = __uint_as_float(hi << 16); o[3] = __uint_as_float(hi & 0xffff0000u);
        } else {
            o[0] = f16_bits_to_f32(static_cast<unsigned short>(lo)); o[1] = f16_bits_to_f32(static_cast<unsigned short>(lo >> 16));
            o[2] = f16_bits_to_f32(static_cast<unsigned short>(hi)); o[3] = f16_bits_to_f32(static_cast<unsigned short>(hi >> 16));
        }
        return o;
    }
}
template <int DT>
__device__ __forceinline__ f32x4 load4_stream(const void* base, int64_t idx) {
    return stream_cvt<DT>(load4_stream_raw<DT>(base, idx), idx);
}

template <int DT>
__device__ __forceinline__ float load1_as_f32(const void* base, int64_t idx);
template <>
__device__ __forceinline__ float load1_as_f32<MMX_F32>(const void* base, int64_t idx) {
    return static_cast<const float*>(base)[idx];
}
template <>
__device__ __forceinline__ float load1_as_f32<MMX_BF16>(const void* base, int64_t idx) {
    return bf16_bits_to_f32(static_cast<const unsigned short*>(base)[idx]);
}
template <>
__device__ __forceinline__ float load1_as_f32<MMX_F16>(const void* base, int64_t idx) {
    return f16_bits_to_f32(static_cast<const unsigned short*>(base)[idx]);
}

// Storage type of a capture slab element and its conversions (round to nearest even on store)
template <int DT> struct slab_elem { typedef unsigned short type; };
template <> struct slab_elem<MMX_F32> { typedef float type; };
template <int DT>
__device__ __forceinline__ float slab_load(const typename slab_elem<DT>::type* p) {
    if constexpr (DT == MMX_F32) return *p;
    else if constexpr (DT == MMX_BF16) return bf16_bits_to_f32(*p);
    else return f16_bits_to_f32(*p);
}
template <int DT>
__device__ __forceinline__ void slab_store(typename slab_elem<DT>::type* p, float v) {
    if constexpr (DT == MMX_F32) {
        *p = v;
    } else if constexpr (DT == MMX_BF16) {
        const unsigned u = __float_as_uint(v);
        *p = (v != v) ? static_cast<unsigned short>(0x7FC0)
                      : static_cast<unsigned short>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    } else {
        *p = __half_as_ushort(__float2half(v));
    }
}

// clamp(x, min=0) with torch semantics: NaN propagates (fmaxf would drop it)
__device__ __forceinline__ float relu_nan(float x) { return (x < 0.0f) ? 0.0f : x; }

// exact-fp32 MFMA: D(16x16) += A(16x4) . B(4x16).
// lane l: a = A[l & 15][l >> 4], b = B[l >> 4][l & 15]; acc[r] = D[(l >> 4) * 4 + r][l & 15].
__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// bf16 MFMA, fp32 accumulate: D(16x16) += A(16x32) . B(32x16).  lane l holds A[l & 15][8 (l >> 4) + j] and
// B[8 (l >> 4) + j][l & 15], j = 0..7; acc as for the fp32 form.  Both operands index the contraction the same way, so
// any 8 values per lane may stand for slots j as long as A and B agree (the kernels pick the 8 that one b128 pair holds).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack_bf16(f32x4 lo, f32x4 hi) {     // round to nearest even (v_cvt_pk_bf16_f32)
    bf16x8 r;
    r[0] = static_cast<__bf16>(lo[0]); r[1] = static_cast<__bf16>(lo[1]); r[2] = static_cast<__bf16>(lo[2]);
    r[3] = static_cast<__bf16>(lo[3]); r[4] = static_cast<__bf16>(hi[0]); r[5] = static_cast<__bf16>(hi[1]);
    r[6] = static_cast<__bf16>(hi[2]); r[7] = static_cast<__bf16>(hi[3]);
    return r;
}
__device__ __forceinline__ f32x4 mfma16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (its workgroup fence covers
// global memory), which would serialise every in-flight global prefetch behind the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-lane all-reduce on DPP row rotations (row_ror:8/4/2/1; a DPP "row" is 16 lanes): a wave works on 4 rows of the
// score matrix at once (lane>>4 selects the row).  VALU-only -- __shfl_xor would go through ds_bpermute (LDS pipe,
// ~10x the latency), and these reductions sit on the softmax critical path.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group16_max(float x) {
    x = fmaxf(x, dpp_row<0x128>(x)); x = fmaxf(x, dpp_row<0x124>(x));
    x = fmaxf(x, dpp_row<0x122>(x)); x = fmaxf(x, dpp_row<0x121>(x));
    return x;
}
__device__ __forceinline__ float group16_sum(float x) {
    x += dpp_row<0x128>(x); x += dpp_row<0x124>(x); x += dpp_row<0x122>(x); x += dpp_row<0x121>(x);
    return x;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2 (linear workgroup id w runs on XCD w % 8).
// Map w to a logical id such that every XCD owns ONE contiguous range of logical ids: neighbours in logical order --
// e.g. the query tiles of one attention head, which all stream the same K / V -- then share an L2 instead of filling
// eight.  A bijection on [0, total) for any total; placement is a performance hint only (nothing depends on it).
__device__ __forceinline__ int xcd_contiguous_id(int w, int total) {
    constexpr int X = 8;
    const int xcd = w % X, local = w / X, per = total / X, rem = total % X;
    return xcd < rem ? xcd * (per + 1) + local : rem * (per + 1) + (xcd - rem) * per + local;
}

inline size_t dtype_size(int dt) { return dt == MMX_F32 ? 4 : 2; }

void set_error(const char* fmt, ...);
void attn_head_enable(int on);
void attn_head_tile_skip(int on);
// relevancy_chain_cols.hip: the chain split by columns of R over the workgroups of a sample, strict layer order (K1c)
bool self_chain_cols_applies(int n_layers, int B, int H, int N);
int self_chain_cols_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N,
                           int64_t attn_bstride, const void* R_init, void* R_out, int nt_policy, int debug, hipStream_t s);
void chain_cols_options(int c, int nb);
// relevancy_chain_groups.hip: the layer-group chain with barrier-free stream waves (K1g)
bool self_chain_groups_applies(int n_layers, int G, int H, int N);
int self_chain_groups_launch(const void* const* attn_layers, const void* const* grad_layers, int n_layers, int B, int H, int N, int G,
                             int64_t attn_bstride, const void* R_init, void* R_out, unsigned* counters, float* parts, int nt_policy,
                             int debug, hipStream_t s);
void attn_stream_enable(int on);
void attn_fwd_split_enable(int on);
void attn_bf16_v2_enable(int on);
void attn_bf16_v3_enable(int mode);
int hip_fail(hipError_t e, const char* what);
// Fill kernels instead of hipMemsetAsync: a memset NODE of a captured hipGraph was observed (ROCm 7.2, gfx950) to replay
// with a corrupted 64-bit pattern once other work had run between capture and replay (every second fp32 word garbage);
// kernel nodes carry their arguments by value and do not have that problem.  ``bytes`` must be a multiple of 4.
int zero_async(void* dst, size_t bytes, hipStream_t s);
// bmm_f32_tiles.hip: the large exact-fp32 products (32 x 32 x 2 MFMA, 64 x 64 tiles); false = not its shape
// relevancy_chain_rows.hip: one layer R_out = R_in + mean_h clamp(G * A, 0) . R_in of the long-sequence chain in ONE launch
bool chain_rows_layer_applies(int N);
int chain_rows_layer_launch(const void* attn, const void* grad, const float* R_in, float* R_out, int B, int H, int N, int dtype,
                            int64_t attn_bstride, hipStream_t s, int debug = 0);
bool bmm_f32_tiles_try(const float* A, const float* B, const float* Cin, float* C, int batch, int M, int N, int K, int trans_a,
                       int64_t sa, int64_t sb, int64_t sc, int nan_to_zero, int cin_is_row, hipStream_t s);
int device_cu_count();   // compute units of the current device, cached
int identity_async(float* R, int batch, int N, hipStream_t s);   // R[b] = I  (N x N, contiguous)

}  // namespace mmx

#define MMX_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            mmx::set_error(__VA_ARGS__);    \
            return MMX_EINVAL;              \
        }                                   \
    } while (0)

#define MMX_LAUNCH_CHECK(what)                                  \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return mmx::hip_fail(e__, what); \
    } while (0)
