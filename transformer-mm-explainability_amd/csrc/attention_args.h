// Argument blocks shared by the tiled (attention_kernels.hip) and whole-head (attention_small.hip) kernels.
#pragma once
#include "mmx_common.h"

namespace mmx {

struct Strides { int64_t sb, sh, sn; };

struct AttnFwdArgs {
    const float *q, *k, *v;
    Strides qs, ks, vs;
    const float* mask; int64_t mask_sb, mask_sq;
    float* probs; float* o; Strides os;
    int B, H, Nq, Nk, D;
    float scale; int scale_mode;
    int debug;  // profiling only (attention_small.hip)
    int slab_dt;  // MMX_F32 | MMX_F16 | MMX_BF16: element type behind `probs` (non-fp32: streaming kernels only)
    int mma_bf16; // 1: products on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32 accumulate, fp32 softmax)
    int tile_skip = 0;  // whole-head kernels: skip the products of key tiles that are masked out for a whole wave (attention_head.hip)
};

struct AttnBwdArgs {
    const float *q, *k, *v;
    Strides qs, ks, vs;
    const float* probs; int64_t probs_sb;  // batch stride of P in elements (0: one P shared by the whole batch)
    const float* dout; Strides os;
    const float* o; Strides oos;   // optional forward output O (same logical shape as dout): delta = rowsum(dO * O) without a sweep
    float* dprobs;
    float *dq, *dk, *dv;
    Strides dqs, dks, dvs;
    float* delta;  // [B, H, Nq] workspace: rowsum(dP * P)
    int B, H, Nq, Nk, D;
    float scale; int scale_mode; int need_dqkv;
    int slab_dt;  // element type behind `probs` and `dprobs` (see AttnFwdArgs)
    int debug;    // tuning knob of attention_head.hip (stagger)
    int mma_bf16; // see AttnFwdArgs (streaming kernels only)
    // Row-relevancy mode (mmx_attn_capture_bwd_rowrel, streaming bf16-MFMA kernels only): instead of (or besides) storing
    // dP, every query-side workgroup reduces  part[k] = sum_{q in its 64 rows} rel_v[b][q] * clamp(dP * P, 0)[q][k]  of
    // its head into rel_part[b][h * nrt + row_tile][k]; rel_v == nullptr: off.
    const float* rel_v;
    float* rel_part;
    float* rel_out;   // rel_v + (1 / H) sum of the partial rows: written by the path that ran (it knows how many rows it made)
    // MMX_ATTN_IO_BF16 (bf16-MFMA streaming kernels only): `dout` is bf16 and dq / dk / dv are written as bf16 (the
    // gradient stream between the bf16 GEMMs of a bf16 body); strides stay in elements.  q / k / v / o / delta: fp32.
    int io_bf16;
    int tile_skip = 0;  // whole-head kernels: skip the products of all-zero probability tiles (attention_head.hip, with_tile_count)
};

int attn_fwd_head_try(const AttnFwdArgs& a, hipStream_t s, int* rc_out);    // attention_head.hip (register-resident)
int attn_bwd_head_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out);
int attn_fwd_stream_try(const AttnFwdArgs& a, hipStream_t s, int* rc_out);   // attention_stream.hip
int attn_bwd_stream_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out);
// out[b][k] = v_in[b][k] + inv_h * sum_{j < J} part[b][j][k]   (row-relevancy mode, second pass; attention_kernels.hip)
int rel_row_update(const float* v_in, const float* part, float* out, int B, int J, int N, float inv_h, hipStream_t s);
int attn_bwd_bf16_try(const AttnBwdArgs& a, hipStream_t s, int* rc_out);     // attention_bf16.hip (2nd-generation bf16 MFMA backward)
void attn_bf16_v2_enable(int on);
// attention_bf16_v3.hip: third generation, shared-forward + bf16 gradient stream + row-relevancy mode only (BASELINE config 5)
int attn_bwd_bf16_v3_try(const AttnBwdArgs& a, void* prep, size_t prep_bytes, hipStream_t s, int* rc_out);
size_t attn_bwd_bf16_v3_prep_bytes(int H, int N);
void attn_bf16_v3_enable(int mode);

}  // namespace mmx
