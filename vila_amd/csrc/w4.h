// W4A16 decode GEMV arguments (gemv_w4.hip).
#pragma once
#include "common.h"

struct GemvW4Args {
    const bf16_t* x; const bf16_t* norm_w; float eps;
    const uint32_t* Wq; const uint32_t* Wsz;       // tile-major packed nibbles / {scale, zero} bf16 pairs (layout: gemv_w4.hip header)
    const bf16_t* bias; const bf16_t* residual; bf16_t* y;
    int N, K, mode;                                // N = outputs. 0 plain, 1 gate/up (2N interleaved rows), 3 fused QKV + RoPE + KV append,
                                                   // 4 plain with x = the merge of the decode attention's per-slice partials (o_proj; hd = 128)
    bf16_t* q_out; bf16_t* kcache; bf16_t* vcache; const int32_t* pos_ptr; const float* rope_cs; int nq, nkv, hd, max_ctx;
    const float* part_o; const float* part_ml; int n_splits, split_keys;     // mode 4: [n_splits][K] un-normalised partial O, [n_splits][K/128][2] (m, l)
};
int launch_gemv_w4(const GemvW4Args& a, hipStream_t s);
