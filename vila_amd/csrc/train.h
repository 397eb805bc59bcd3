// Launchers of the backward / optimizer kernels (train.hip, attn_bwd.hip).
#pragma once
#include "common.h"

struct AttnBwdArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* d_o;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int64_t q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride, do_tok_stride, dq_tok_stride, dk_tok_stride, dv_tok_stride;
    int q_head_stride, k_head_stride, v_head_stride, o_head_stride, do_head_stride, dq_head_stride, dk_head_stride, dv_head_stride;
    const int32_t* cu_seqlens; int n_seq, total_tokens, max_seqlen;
    int n_q_heads, n_kv_heads, head_dim, causal; float scale;
    const float* lse;     // [Hq][T] natural-log LSE from the forward
    float* delta;         // [Hq][T] workspace
};
int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s, int parts = 7);   // parts: 1 delta | 2 dQ | 4 dK/dV
int launch_attn_bwd_dma(const AttnBwdArgs& a, hipStream_t s, int parts);       // round-3 dQ / dK-dV kernels (attn_bwd_dma.hip)

int launch_transpose(const bf16_t* in, bf16_t* out, int R, int C, int64_t ldi, int64_t ldo, hipStream_t s);
int launch_act_fwd(const bf16_t* z, bf16_t* y, int64_t n, int act, hipStream_t s);
int launch_act_bwd(const bf16_t* z, const bf16_t* dy, bf16_t* dz, int64_t n, int act, hipStream_t s);
int launch_silu_mul_fwd(const bf16_t* g, const bf16_t* u, bf16_t* a, int64_t n, hipStream_t s);
int launch_silu_mul_bwd(const bf16_t* g, const bf16_t* u, const bf16_t* da, bf16_t* dg, bf16_t* du, int64_t n, hipStream_t s);
int launch_add(const bf16_t* a, const bf16_t* b, bf16_t* y, int64_t n, hipStream_t s);
int launch_grad_accum(float* acc, const bf16_t* g, bf16_t* out, int64_t n, int mode, hipStream_t s);
int launch_colsum(const bf16_t* x, bf16_t* out, float* scratch, int R, int C, int64_t ld, int accumulate, int period, hipStream_t s);
int launch_norm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* dy, bf16_t* dx, bf16_t* dw, bf16_t* db, float* scratch,
                    int rows, int cols, float eps, int rms, int accumulate, hipStream_t s);
int launch_ce(const float* logits, const int64_t* labels, bf16_t* dlogits, float* loss, float* row_loss, int rows, int V, int64_t ldl, float scale, hipStream_t s);
int launch_scatter_add_rows(const bf16_t* src, bf16_t* dst, const int32_t* rows, int n, int H, hipStream_t s);
int launch_depth_to_space(const bf16_t* dy, bf16_t* dx, int B, int g, int C, int k, hipStream_t s);
int launch_rope_bwd(bf16_t* dqkv, const float* cs, const float* sn, int S, int nq, int nkv, int hd, hipStream_t s);
int launch_adamw(float* master, float* m, float* v, const bf16_t* grad, bf16_t* param, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, int step, float grad_scale, hipStream_t s);
int launch_adamw_lean(float* master, float* m, float* v, const bf16_t* grad, bf16_t* param, int64_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, float grad_scale, hipStream_t s);
#define SUMSQ_PARTS 2048
int launch_sumsq(const bf16_t* x, int64_t n, float* out, float* scratch, hipStream_t s);
size_t colsum_scratch_floats(int R, int C);
size_t norm_bwd_scratch_floats(int rows, int cols);
