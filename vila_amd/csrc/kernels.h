// Internal launcher declarations shared by the translation units of libvila_hip.so.
#pragma once
#include "common.h"

// stream handle of the C-ABI -> hipStream_t, with the calling thread switched to the stream's device (api.hip)
hipStream_t vila_stream_enter(void* stream);

enum { EPI_NONE = 0, EPI_GELU_TANH = 1, EPI_GELU_ERF = 2, EPI_GATEUP = 3 };

// C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]);  bf16 in, fp32 accumulate, bf16 (or fp32) out.
// EPI_GATEUP: C[M,N] = silu(A.W^T) * (A.W2^T)   (N = intermediate size)
struct GemmArgs {
    const bf16_t* A = nullptr; int64_t lda = 0;
    const bf16_t* W = nullptr; int64_t ldw = 0;
    const bf16_t* W2 = nullptr;
    const bf16_t* bias = nullptr;
    const bf16_t* residual = nullptr; int64_t ldr = 0;
    int res_mod = 0;                            // > 0: the residual has res_mod rows and row m reads residual[m % res_mod] (position embeddings of a batch)
    void* C = nullptr; int64_t ldc = 0; int out_f32 = 0;
    int M = 0, N = 0, K = 0; int epi = EPI_NONE;
    float* ws = nullptr; size_t ws_bytes = 0;   // optional fp32 workspace: enables split-K for under-filled grids
    // operand storage (gemm256_kernel.h): 0 = contraction-contiguous X[rows][K] (forward layout), 1 = contraction-major X[K][rows]
    // (dgrad reads W[N,K] with b_cm; wgrad reads dY[T,N] and X[T,K] with a_cm and b_cm); lda / ldw are the STORED leading dimensions
    int a_cm = 0, b_cm = 0;
    // optional fused follow-up normalisation of the OUTPUT rows (the next block's LayerNorm / RMSNorm): honoured only where the launcher takes
    // the split-K path (its reduce kernel holds whole rows) — then *norm_done = 1 and norm_out[M][N] = norm(C) with C as it was just stored
    const bf16_t* norm_w = nullptr; const bf16_t* norm_b = nullptr; float norm_eps = 0.f; int norm_rms = 0;
    bf16_t* norm_out = nullptr; int* norm_done = nullptr;
    // optional fused follow-up of a q/k/v projection (prefill): bias -> bf16 -> RoPE on the q and k heads -> K / V rows into the cache, i.e. what
    // rope_kv_kernel (elementwise.hip) does to C afterwards.  Honoured only on the split-K path (its reduce then does both) — *rope_done = 1
    const float* rope_cs = nullptr; const float* rope_sn = nullptr; const int32_t* rope_pos = nullptr; const int32_t* rope_seq = nullptr;
    bf16_t* rope_kc = nullptr; bf16_t* rope_vc = nullptr; int rope_nq = 0, rope_nkv = 0, rope_hd = 0, rope_max_ctx = 0; int* rope_done = nullptr;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
// a complete, applicable RoPE + KV follow-up request (the output row IS the fused q|k|v row, no residual, heads of a multiple of 16)
static inline bool gemm_rope_offer(const GemmArgs& a) {
    return a.rope_done != nullptr && a.rope_cs != nullptr && a.rope_sn != nullptr && a.rope_pos != nullptr && a.rope_hd > 0 && a.rope_hd % 16 == 0 &&
           a.N == (a.rope_nq + 2 * a.rope_nkv) * a.rope_hd && a.residual == nullptr && a.epi == EPI_NONE && !a.out_f32 && a.ldc % 8 == 0;
}

// ---- normalisation / elementwise (elementwise.hip) ----
int launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int cols, float eps, hipStream_t s);
int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, float eps, hipStream_t s);
// im2col for Conv2d(k=s=P, valid): pixels [B,C,H,W] bf16 -> patches [B*gh*gw, Kp] (K = C*P*P zero-padded to Kp)
int launch_im2col(const bf16_t* px, bf16_t* out, int B, int C, int H, int W, int P, int Kp, hipStream_t s);
int launch_pad_rows(const bf16_t* in, bf16_t* out, int rows, int K, int Kp, hipStream_t s);
int launch_add_pos(bf16_t* x, const bf16_t* pos, int B, int N, int D, hipStream_t s);
// space-to-depth (flat_square k x k with zero padding) : x [B, g*g, C] -> y [B, gd*gd, k*k*C]
int launch_space_to_depth(const bf16_t* x, bf16_t* y, int B, int g, int C, int k, hipStream_t s);
// rope table: positions [S] i32 -> cos/sin [S][hd/2] stored as bf16-rounded fp32 (HF casts cos/sin to the act dtype)
int launch_rope_table(const int32_t* pos, float* cs, float* sn, int S, int hd, float theta, hipStream_t s);
// applies bias-added q,k RoPE in place on the fused qkv buffer [S][q+2kv] and scatters k,v into the cache
int launch_rope_kv(bf16_t* qkv, const float* cs, const float* sn, const int32_t* pos, const int32_t* seq_of_tok,
                   bf16_t* kcache, bf16_t* vcache, int S, int nq, int nkv, int hd, int max_ctx, hipStream_t s);
int launch_embed_gather(const bf16_t* table, const int64_t* ids, bf16_t* out, int n, int H, int64_t vocab, hipStream_t s);
int launch_copy_rows(const bf16_t* src, bf16_t* dst, const int32_t* src_row, const int32_t* dst_row, int n, int H, hipStream_t s);
int launch_argmax(const float* logits, int V, int64_t* out, float* tmpv, int* tmpi, hipStream_t s);
// stochastic token choice (sample.hip): temperature -> top-k (1..64) -> top-p -> draw; counter = device scalar mixed into the RNG
size_t sample_workspace_bytes();
int launch_sample(const float* logits, int n, float temperature, int top_k, float top_p, uint64_t seed, const uint64_t* seed_dev, const int32_t* counter, int64_t* out,
                  void* workspace, float* prob_out, hipStream_t s);
// dynamic_s2 merge (s2.hip): tower output -> projector input, desc = device [n_blocks][6] {tile_base, bh, bw, i, j, single}
int launch_s2_merge(const bf16_t* feats, bf16_t* out, const int32_t* desc, int n_blocks, int g, int C, int n_scales, const int* splits,
                    hipStream_t s);
// adjoint of the merge: dy [n_blocks][g*g][n_scales*C] -> dx [n_tiles][g*g][C]; tdesc = device [n_tiles][8] {first block of the image,
// bh, bw, scale, tile row, tile col, single, 0}
int launch_s2_merge_bwd(const bf16_t* dy, bf16_t* dx, const int32_t* tdesc, int n_tiles, int g, int C, int n_scales, const int* splits,
                        hipStream_t s);

// video token assembly (video.hip): temporal / spatial mean pooling + start / end token rows per pooled frame
int launch_video_pool(const bf16_t* feats, bf16_t* out, int nt, int nl, int C, int pt, int ph, int pw, const bf16_t* start_rows, int n_start,
                      const bf16_t* end_rows, int n_end, hipStream_t s);
int launch_video_pool_bwd(const bf16_t* dpooled, bf16_t* dfeats, int nt, int nl, int C, int pt, int ph, int pw, int accumulate, hipStream_t s);

// ---- W8A8 (gemm_i8.hip): Y = epi((Xq . Wq^T) * sx[m] * sw[n] + bias) (+ residual), int8 operands, int32 accumulate, bf16 out ----
int launch_gemm_i8(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const float* sx, const float* sw, const bf16_t* bias,
                   const bf16_t* residual, int64_t ldr, bf16_t* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s);
int launch_quant_rows_i8(const bf16_t* x, int8_t* q, float* scale, int rows, int cols, hipStream_t s);

// ---- attention (attn.hip) ----
struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride;   // elements between consecutive tokens
    int q_head_stride, k_head_stride, v_head_stride, o_head_stride;  // elements between heads
    const int32_t* cu_seqlens;   // [n_seq+1] device, or null => one sequence of total_tokens
    int n_seq, total_tokens, max_seqlen;
    int n_q_heads, n_kv_heads, head_dim;
    int causal; float scale;
    float* lse;                  // optional [n_q_heads][total_tokens] fp32 (natural-log LSE) for backward
};
int launch_attn_fwd(const AttnArgs& a, hipStream_t s);

// ---- decode (gemv.hip) ----
// Chained decode kernels (round 4): kernel i of a token is launched BEFORE kernel i-1 has finished (alternating streams: i-2 -> i is stream
// order), issues the weight loads that do not depend on activations, then waits until kernel i-1's blocks have all counted themselves done.
// ctr == nullptr: a plain kernel (no wait, no count, plain stores).  gemv_common.h: chain_wait / chain_done / the coherent load + store forms.
#define CHAIN_FLAGS 64            // replicated "go" words per kernel, 256 B apart (one memory channel each): pollers spread over them
#define CHAIN_STRIDE 64           // u32 words between two polled words (256 B)
#define CHAIN_WORDS ((CHAIN_FLAGS + 1) * CHAIN_STRIDE)      // per kernel: word 0 = arrival count, words (1 + f) * CHAIN_STRIDE = flag f
struct ChainLink {
    uint32_t* ctr = nullptr;        // [n_kernels][CHAIN_WORDS] of this token (zeroed by the prologue kernel)
    uint32_t* err = nullptr;        // set to 1 if a wait gave up (bounded spin: a hang becomes a reported error)
    int wait_idx = -1;              // kernel to wait for (-1: none)
    uint32_t wait_target = 0;       // unused by the flag form (kept for the error report)
    int done_idx = -1;              // this kernel's slot (>= 0: outputs are stored write-through, sc1); the last of `done_blocks` raises the flags
    uint32_t done_blocks = 0;       // = this kernel's grid size
    int pre_sleep_us = 0;           // the predecessor's predicted run time: polls concentrate around it (gemv_common.h chain_wait)
};
struct GemvArgs {
    const bf16_t* x;          // [K] activations (bf16)
    const bf16_t* norm_w;     // optional RMSNorm gain fused in front (null = none)
    float eps;
    const bf16_t* W;          // [N][K]
    const bf16_t* W2;         // gate/up mode: up rows
    const bf16_t* bias;       // [N] optional
    const bf16_t* residual;   // [N] optional, added after
    bf16_t* y;                // [N] bf16 out (or null)
    float* y_f32;             // [N] fp32 out (logits) (or null)
    int N, K; int mode;       // 0 plain, 1 gate/up silu-mul, 2 plain with x = merge of the decode-attention partials
    const float* part_o; const float* part_ml; const int32_t* pos_ptr; int n_splits;   // mode 2
    int split_keys;           // mode 2: keys per partial slice (0 = the 64-key slices of attn_decode_partial)
    int grid_cap;             // mode 2: upper bound of the grid (0 = 256 blocks)
    int max_bpc;              // blocks per CU cap (0 = 4): a chained kernel and its neighbour each take at most half a CU
    ChainLink chain;
};
int launch_gemv(const GemvArgs& a, hipStream_t s, int* grid_out = nullptr);
struct QkvDecodeArgs {
    const bf16_t* x; const bf16_t* norm_w; float eps;
    const bf16_t* Wqkv; const bf16_t* bqkv;   // fused [q+2kv][K], [q+2kv]
    bf16_t* q_out;                             // [nq*hd]
    bf16_t* kcache; bf16_t* vcache;            // this layer's [nkv][max_ctx][hd]
    const int32_t* pos_ptr;                    // device scalar: position of the new token (= current context length)
    const float* rope_cs;                      // [hd] cos | sin of that position (decode_prologue_kernel)
    int K, nq, nkv, hd, max_ctx;
    int max_bpc;              // blocks per CU cap (0 = 4)
    ChainLink chain;
};
int launch_qkv_decode(const QkvDecodeArgs& a, hipStream_t s, int* grid_out = nullptr);
struct AttnDecodeArgs {
    const bf16_t* q; const bf16_t* kcache; const bf16_t* vcache;
    bf16_t* o;                       // [nq*hd] merged output (second launch); null = partials only
    float* part_o; float* part_ml;   // workspace: [n_splits][nq][hd], [n_splits][nq][2]
    const int32_t* pos_ptr;          // context length BEFORE this token; keys 0..pos inclusive are attended
    int nq, nkv, hd, max_ctx, n_splits; float scale;
    int force_split;                 // 1: always use the split-KV + merge pair (default: single-launch per-head kernel when max_ctx <= 2048)
    int split256;                    // 1: per-head blocks over 256-key slices, partials only (merged in the o_proj GEMV prologue)
    // batched decode (decode_batch.hip): row = blockIdx.z reads q + row * q_row_stride, cache slot row (+ row * slot_stride), pos_ptr[row]
    int64_t q_row_stride, o_row_stride, slot_stride;
};
int launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s, int* grid_out = nullptr);
int launch_attn_decode_rows(const AttnDecodeArgs& a, int n_rows, int64_t q_row_stride, int64_t o_row_stride, int64_t slot_stride, hipStream_t s);
// batched decode step (decode_batch.hip)
struct BDecodeArgs {
    const void* embed; const void* norm_w; const void* lm_head;
    int hidden, inter, n_layers, q_heads, kv_heads, head_dim, vocab; float rms_eps, rope_theta;
};
struct BLayer { const void *ln1_w, *wqkv, *bqkv, *wo, *ln2_w, *w_gate, *w_up, *w_down; };
size_t bdecode_workspace_bytes(int H, int F, int QS, int hd, int n);
int bdecode_step(const BDecodeArgs& m, const BLayer* layers, bf16_t* kcache, bf16_t* vcache, int max_ctx, int n_slots, int n, int32_t* pos, int64_t* token,
                 int64_t* out_ids, int32_t* n_out, int max_out, float* logits, void* workspace, size_t workspace_bytes, hipStream_t s);
int launch_decode_prologue(const bf16_t* table, const int64_t* tok, bf16_t* out, int H, int64_t vocab, const int32_t* pos, float* rope_cs,
                           int hd, float theta, hipStream_t s, uint32_t* chain_ctr = nullptr, int n_chain = 0);
int launch_decode_advance(int32_t* pos, const int64_t* tok, int64_t* out_ids, int32_t* n_out, int max_out, hipStream_t s);
