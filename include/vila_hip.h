/* vila_hip.h — C ABI of libvila_hip.so: the MI355X (gfx950) implementation of NVILA's forward hot path.
 *
 * The reference (NVlabs/VILA) has no FFI/plugin API: its seams are Python module contracts.
 * Every entry point below names the reference interface it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer in the caller's current HIP context unless marked [host];
 *   - the caller (PyTorch) owns weights, activations, KV cache and workspace; the library allocates nothing
 *     and keeps no device memory between calls;
 *   - all work is enqueued asynchronously on `stream`; nothing synchronises the device;
 *   - layouts are the reference's: nn.Linear.weight [out,in] row-major bf16, activations [tokens, channels] bf16,
 *     conv weight [out, C, P, P];
 *   - return 0 on success, <0 on error; vila_last_error() gives a thread-local message.  The Python shim turns a
 *     non-zero status into ValueError/RuntimeError with the reference's wording where one exists.
 *   - re-entrant per (device, stream); no thread-local HIP state is assumed (backward runs on autograd threads).
 */
#ifndef VILA_HIP_H
#define VILA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vila_stream_t; /* hipStream_t */

const char* vila_last_error(void);
int vila_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Vision tower — replaces SiglipVisionTower / VisionTower.forward + feature_select
 *   llava/model/multimodal_encoder/vision_encoder.py:44-52,133-177  (hidden_states[select_layer], "cls_patch")
 *   llava/model/multimodal_encoder/siglip/modeling_siglip.py:320-329 (embeddings), :728-764 (encoder layer),
 *   :389-439 (attention), :711-715 (MLP)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int hidden, inter, heads, image, patch, channels;
    int n_layers_run;   /* layers actually executed = index of hidden_states[select_layer] (26 for so400m, -2) */
    float ln_eps;
} VilaVitShape;
typedef struct {
    const void *ln1_w, *ln1_b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} VilaVitLayer;
typedef struct {
    VilaVitShape shape;
    const void* patch_w;          /* [hidden, C, P, P] */
    const void* patch_b;          /* [hidden] */
    const void* pos_emb;          /* [ (image/patch)^2, hidden ] */
    const VilaVitLayer* layers;   /* [host] array of n_layers_run entries */
} VilaVitWeights;
size_t vila_vit_workspace_bytes(const VilaVitShape* s, int n_images);
/* pixels [B, C, image, image] bf16 NCHW  ->  out [B, (image/patch)^2, hidden] bf16 */
int vila_vit_forward(const VilaVitWeights* w, const void* pixels, int n_images, void* out,
                     void* workspace, size_t workspace_bytes, vila_stream_t stream);

/* W8A8 vision tower (BASELINE configs[4]; the reference's quantised numbers come from the external TinyChat
 * backend, README.md:87 — no code in-tree, so the format is defined here): the q/k/v (fused), out_proj, fc1 and fc2 linears of every
 * encoder layer run int8 x int8 -> int32 on the matrix cores with per-output-channel weight scales (w = wq * ws[n], symmetric,
 * ws[n] = max|W[n,:]| / 127) and per-token dynamic activation scales; everything else as vila_vit_forward (same call contract). */
typedef struct {
    const void *wqkv_q, *wo_q, *fc1_q, *fc2_q;        /* int8 [3*hidden][hidden], [hidden][hidden], [inter][hidden], [hidden][inter] */
    const float *wqkv_s, *wo_s, *fc1_s, *fc2_s;       /* fp32 per output row */
} VilaVitLayerW8;
size_t vila_vit_w8a8_workspace_bytes(const VilaVitShape* s, int n_images);
int vila_vit_forward_w8a8(const VilaVitWeights* w, const VilaVitLayerW8* qlayers /*[host]*/, const void* pixels, int n_images, void* out,
                          void* workspace, size_t workspace_bytes, vila_stream_t stream);
/* operator level: per-token dynamic int8 quantisation (scale[r] = max|x_r| / 127) and the W8A8 GEMM
 * C[M,N] = epi((Aq . Wq^T) * sx[m] * sw[n] + bias[n]) (+ residual), epi in {VILA_EPI_NONE, VILA_EPI_GELU_TANH}; K %% 16 == 0 */
int vila_quant_rows_i8(const void* x_bf16, void* q_i8, float* scale, int rows, int cols, vila_stream_t stream);
int vila_gemm_w8a8(const void* Aq, int64_t lda, const void* Wq, int64_t ldw, const float* sx, const float* sw, const void* bias,
                   const void* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int epi, vila_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * mm_projector — replaces MultimodalProjector.forward (llava/model/multimodal_projector/base_projector.py:248-252)
 * for mlp_downsample (:145-153), mlp_downsample_2x2_fix (:155-162), mlp_downsample_3x3_fix (:163-174), including the
 * space-to-depth blocks flat_square / flat_square_2x2 / flat_square_3x3 (:58-71, :84-97, :110-123).
 * ------------------------------------------------------------------------------------------------------------ */
enum { VILA_PROJ_MLP_DOWNSAMPLE = 0, VILA_PROJ_MLP_DOWNSAMPLE_2X2_FIX = 1, VILA_PROJ_MLP_DOWNSAMPLE_3X3_FIX = 2 };
typedef struct {
    int kind;
    int in_dim;    /* vision hidden C */
    int out_dim;   /* LLM hidden */
    const void *ln1_w, *ln1_b;   /* layers.1 */
    const void *fc1_w, *fc1_b;   /* layers.2 */
    const void *ln2_w, *ln2_b;   /* layers.4 (3x3 only) */
    const void *fc2_w, *fc2_b;   /* layers.4 (2x2) / layers.5 (3x3) */
    const void *fc3_w, *fc3_b;   /* layers.7 (3x3 only) */
} VilaProjWeights;
size_t vila_proj_workspace_bytes(const VilaProjWeights* w, int n_images, int n_tokens);
int vila_proj_out_tokens(int kind, int n_tokens);
/* feat [B, N, C] bf16 (N a perfect square) -> out [B, N', out_dim] bf16 */
int vila_proj_forward(const VilaProjWeights* w, const void* feat, int n_images, int n_tokens, void* out,
                      void* workspace, size_t workspace_bytes, vila_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Embedding + splice — replaces llm.model.embed_tokens (llava_arch.py:429, encoders/image/basic.py:22-27) and the
 * per-token python splice loop of LlavaMetaForCausalLM._embed (llava_arch.py:457-479): dst[dst_row[i]] = src[src_row[i]].
 * The caller computes the row maps (integer work on the ids, no per-token host sync).
 * ------------------------------------------------------------------------------------------------------------ */
int vila_embed_tokens(const void* table, int64_t vocab, int hidden, const int64_t* ids, int n, void* out, vila_stream_t stream);
int vila_copy_rows(const void* src, void* dst, const int32_t* src_row /*nullable*/, const int32_t* dst_row /*nullable*/,
                   int n, int hidden, vila_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * LLM — replaces HF Qwen2ForCausalLM.forward / .generate as called from
 *   llava/model/language_model/llava_llama.py:134-141 (forward(inputs_embeds, attention_mask, position_ids, labels))
 *   llava/model/llava_arch.py:833 (llm.generate(inputs_embeds=..., attention_mask=...))
 * (third-party transformers==4.46.0, pyproject.toml:17; arithmetic per SURVEY.md Appendix B).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int hidden, inter, n_layers, q_heads, kv_heads, head_dim, vocab;
    float rms_eps, rope_theta;
} VilaLlmShape;
typedef struct {
    const void *ln1_w, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *ln2_w, *w_gate, *w_up, *w_down;
} VilaLlmLayer;
typedef struct {
    VilaLlmShape shape;
    const void* embed;            /* [vocab, hidden] */
    const VilaLlmLayer* layers;   /* [host] */
    const void* norm_w;
    const void* lm_head;          /* [vocab, hidden] (== embed when tied) */
} VilaLlmWeights;
typedef struct {
    void* k;        /* [n_layers][n_slots][kv_heads][max_ctx][head_dim] bf16 */
    void* v;
    int max_ctx, n_slots;
} VilaKvCache;

size_t vila_llm_prefill_workspace_bytes(const VilaLlmShape* s, int total_tokens);
/* Prefill / teacher-forced forward over a packed token stream.
 *   embeds      [total_tokens, hidden] bf16 (spliced image+text embeddings)
 *   positions   [total_tokens] i32   (restart per sequence, llava_arch.py:751)
 *   cu_seqlens  [n_seq+1] i32 or NULL (one sequence)  — flash-attn varlen semantics of model/utils/packing.py:12-21
 *   seq_of_tok  [total_tokens] i32 or NULL: KV-cache slot per token
 *   cache       nullable: K/V are appended at `positions`
 *   last_rows   [n_last] i32 or NULL: rows whose logits are wanted -> last_logits [n_last, vocab] fp32
 *   all_logits  nullable [total_tokens, vocab] fp32 (training / parity)
 *   final_hidden nullable [total_tokens, hidden] bf16: output of model.norm
 *   layer_hidden nullable [(n_layers+1), total_tokens, hidden] bf16: residual stream taps (parity tests)
 */
int vila_llm_prefill(const VilaLlmWeights* w, const void* embeds, const int32_t* positions, const int32_t* cu_seqlens,
                     int n_seq, int total_tokens, int max_seqlen, const int32_t* seq_of_tok, const VilaKvCache* cache,
                     const int32_t* last_rows, int n_last, float* last_logits, float* all_logits, void* final_hidden,
                     void* layer_hidden, void* workspace, size_t workspace_bytes, vila_stream_t stream);

/* Greedy decode state for one sequence (slot 0 of the cache); everything lives on the device so a step can be
 * replayed from a hipGraph with no host round trip (replaces the python loop of GenerationMixin greedy search). */
typedef struct {
    int32_t* pos;        /* scalar: number of tokens already in the cache = position of the next token */
    int64_t* token;      /* scalar: token to feed (in) / token chosen by argmax (out) */
    int64_t* out_ids;    /* [max_out] generated ids, appended every step */
    int32_t* n_out;      /* scalar */
    int max_out;
    float* logits;       /* [vocab] fp32 logits of the last step (kept for parity tests) */
} VilaDecodeState;
size_t vila_llm_decode_workspace_bytes(const VilaLlmShape* s, int max_ctx);
/* number of kernel launches one decode step enqueues (what a captured hipGraph replays per token) */
int vila_llm_decode_launches(const VilaLlmShape* s, int max_ctx);
int vila_llm_decode_step(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st,
                         void* workspace, size_t workspace_bytes, vila_stream_t stream);
/* Word 0 of the decode workspace is an error flag: the caller zeroes the first 256 bytes of a new workspace once.  Only the opt-in CHAINED
 * step writes it (vila_hip_tuning.h vila_decode_force_chain / VILA_DECODE_CHAIN=1: each kernel starts streaming its weights while its predecessor
 * finishes and waits on a device-side arrival count before it reads an activation; the waits are bounded and a give-up sets the word — measured
 * slower than the plain step and off by default).  This call synchronises `stream`, returns the word and clears it: 0 = every token so far is valid. */
int vila_llm_decode_chain_error(void* workspace, vila_stream_t stream);

/* generate(do_sample=True): what HF GenerationMixin.sample does after the forward — logits / temperature -> TopK -> TopP -> softmax ->
 * multinomial (server.py:101-102,185-187 sets temperature 0.2 / top_p 0.9; GenerationConfig's default top_k = 50 applies).  The whole
 * choice runs on the device (three short launches for top_k in 1..64; one exact radix-selection launch for any other k) so that a sampled step
 * replays from a hipGraph; the uniform number is splitmix64(seed, position of the token).  Parity with torch.multinomial is distributional,
 * not bitwise. */
typedef struct {
    float temperature;   /* > 0 */
    int top_k;           /* >= 0; 0 = no top-k filter (HF), k >= n = the same */
    float top_p;         /* (0, 1] */
    uint64_t seed;
    const uint64_t* seed_dev;   /* optional device scalar: when non-NULL the kernel reads the seed from it (a captured decode graph then
                                   serves every sampled request: the host updates 8 bytes instead of re-capturing) and `seed` is ignored */
} VilaSampling;
size_t vila_sample_workspace_bytes(void);
/* logits [n] fp32 -> *out; counter: device scalar mixed into the RNG (nullable); dist_out (nullable): the distribution actually sampled from —
 * top_k in 1..64: [64] probabilities followed by [64] int32 token ids (descending probability, -1 = unused slot); any other top_k: [n] dense
 * probabilities */
int vila_sample_f32(const float* logits, int n, const VilaSampling* sp, const int32_t* counter, int64_t* out, void* workspace,
                    float* dist_out, vila_stream_t stream);
int vila_llm_decode_step_sample(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st,
                                void* workspace, size_t workspace_bytes, const VilaSampling* sp, vila_stream_t stream);

/* Batched greedy decode step (serving: llava_arch.py:823-833 called with a batch, server.py:171-290 serving concurrent requests): ONE pass
 * over the weights advances n <= 16 sequences.  Row i lives in KV-cache slot i (cache->n_slots >= n, filled by vila_llm_prefill with
 * seq_of_tok = slot), consumes token[i] at position pos[i] and leaves its next token in token[i] / out_ids[i][n_out[i]++], pos[i]++.
 * Rows that hit EOS keep stepping (the host ignores what they produce).  head_dim 128, caches up to 2048 positions. */
typedef struct {
    int n;               /* sequences */
    int32_t* pos;        /* [n] */
    int64_t* token;      /* [n] */
    int64_t* out_ids;    /* [n][max_out] */
    int32_t* n_out;      /* [n] */
    int max_out;
    float* logits;       /* [n][vocab] fp32 logits of the last step */
} VilaDecodeBatch;
size_t vila_llm_decode_batch_workspace_bytes(const VilaLlmShape* s, int n);
int vila_llm_decode_step_batch(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeBatch* st,
                               void* workspace, size_t workspace_bytes, vila_stream_t stream);

/* hipGraph helpers: capture whatever is enqueued on `stream` between begin/end, replay it later. */
int vila_graph_begin(vila_stream_t stream);
int vila_graph_end(vila_stream_t stream, void** graph_exec_out);
int vila_graph_launch(void* graph_exec, vila_stream_t stream);
int vila_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the parity tests and by autograd wrappers)
 * ------------------------------------------------------------------------------------------------------------ */
enum { VILA_EPI_NONE = 0, VILA_EPI_GELU_TANH = 1, VILA_EPI_GELU_ERF = 2, VILA_EPI_GATEUP = 3 };
/* C[M,N] = epi(A[M,K] W[N,K]^T + bias[N]) + residual[M,N];  EPI_GATEUP: C = silu(A W^T) * (A W2^T) */
int vila_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* W2, const void* bias,
                   const void* residual, int64_t ldr, void* C, int64_t ldc, int out_f32, int M, int N, int K, int epi,
                   vila_stream_t stream);
/* same, with an fp32 workspace (>= splits*M*N*4 bytes): lets under-filled grids (small M x N, long K) run split-K */
int vila_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const void* W2, const void* bias,
                      const void* residual, int64_t ldr, void* C, int64_t ldc, int out_f32, int M, int N, int K, int epi,
                      void* ws, size_t ws_bytes, vila_stream_t stream);
/* Backward GEMMs on the forward tensors AS THEY LIE (replaces the transposed operand copies autograd's mm backward / cuBLAS NT-TN
 * variants stand for): C[M,N] = A.B^T (+bias)(+residual), bf16 out.  a_cm / b_cm = 1: that operand is stored contraction-major,
 * X[K][rows] with rows contiguous (lda / ldw = its stored leading dimension), instead of [rows][K]:
 *   dgrad  dX[T,K] = dY[T,N] . W[N,K]     : A = dY (a_cm 0), B = W   (b_cm 1, ldw = K),  contraction N
 *   wgrad  dW[N,K] = dY[T,N]^T . X[T,K]   : A = dY (a_cm 1, lda = N), B = X (b_cm 1, ldw = K), contraction T (any T; zero-filled tail)
 * A contraction-major operand needs rows %% 8 == 0.  ws (nullable): fp32 workspace enabling split-K on under-filled grids. */
int vila_gemm_bf16_t(const void* A, int64_t lda, int a_cm, const void* W, int64_t ldw, int b_cm, const void* bias,
                     const void* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, void* ws, size_t ws_bytes,
                     vila_stream_t stream);
/* (process-global tuning / test switches of the kernels live in include/vila_hip_tuning.h: they are NOT part of the drop-in boundary) */
int vila_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps, vila_stream_t stream);
int vila_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int cols, float eps, vila_stream_t stream);
int vila_space_to_depth_bf16(const void* x, void* y, int n_images, int grid, int channels, int k, vila_stream_t stream);
/* q,k,v,o: [tokens][heads][head_dim] views given by element strides; cu_seqlens NULL => n_seq sequences of max_seqlen */
int vila_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t q_tok_stride, int64_t k_tok_stride,
                       int64_t v_tok_stride, int64_t o_tok_stride, int q_head_stride, int k_head_stride, int v_head_stride,
                       int o_head_stride, const int32_t* cu_seqlens, int n_seq, int total_tokens, int max_seqlen,
                       int n_q_heads, int n_kv_heads, int head_dim, int causal, float scale, float* lse, vila_stream_t stream);
int vila_gemv_bf16(const void* x, const void* norm_w, float eps, const void* W, const void* W2, const void* bias,
                   const void* residual, void* y_bf16, float* y_f32, int N, int K, int mode, vila_stream_t stream);
int vila_argmax_f32(const float* logits, int n, int64_t* out, void* workspace /* >= 4 KiB */, vila_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------
 * SFT step: backward and optimizer operators.  They replace what autograd + cuBLAS +
 * flash_attn backward + torch.optim.AdamW do under HF Trainer.training_step (llava/train/transformer_normalize_monkey_patch.py:183-249).
 * dgrad / wgrad are vila_gemm_bf16 on transposed operands (vila_transpose_bf16).
 * ------------------------------------------------------------------------------------------------------------ */
int vila_transpose_bf16(const void* in, void* out, int rows, int cols, int64_t ld_in, int64_t ld_out, vila_stream_t stream);
int vila_act_fwd_bf16(const void* z, void* y, int64_t n, int act /*1 tanh-GELU, 2 erf-GELU*/, vila_stream_t stream);
int vila_act_bwd_bf16(const void* z, const void* dy, void* dz, int64_t n, int act, vila_stream_t stream);
int vila_silu_mul_fwd_bf16(const void* gate, const void* up, void* act, int64_t n, vila_stream_t stream);
int vila_silu_mul_bwd_bf16(const void* gate, const void* up, const void* dact, void* dgate, void* dup, int64_t n, vila_stream_t stream);
int vila_add_bf16(const void* a, const void* b, void* y, int64_t n, vila_stream_t stream);
/* Gradient accumulation across the micro-batches of one update (`--gradient_accumulation_steps`; HF Trainer adds each micro-batch's
 * backward into the fp32 / param-dtype .grad, transformer_normalize_monkey_patch.py:236-249): the running sum `acc` is fp32.
 * mode 0: acc = g;  mode 1: acc += g;  mode 2: out = bf16(acc + g) (acc untouched, out may alias g).  g, out: bf16[n]; n % 8 == 0. */
int vila_grad_accum_f32(float* acc, const void* g, void* out, int64_t n, int mode, vila_stream_t stream);
/* Reductions of the SFT step are DETERMINISTIC (round 6): every block writes fp32 partials into the caller's scratch and a second kernel adds
 * them in a fixed order — no atomics, so two identical steps produce identical bits (a resumed run equals the uninterrupted one). */
size_t vila_colsum_scratch_floats(int rows, int cols);
size_t vila_norm_bwd_scratch_floats(int rows, int cols);
#define VILA_SUMSQ_SCRATCH_FLOATS 2048
/* out[c] (+)= sum_r x[r][c] (scratch: vila_colsum_scratch_floats fp32); period > 0: out[p][c] = sum over rows r == p (mod period) (position-embedding gradient, no scratch) */
int vila_colsum_bf16(const void* x, void* out, float* scratch, int rows, int cols, int64_t ld, int accumulate, int period, vila_stream_t stream);
/* LayerNorm (rms=0) / RMSNorm (rms=1) backward; scratch = vila_norm_bwd_scratch_floats(rows, cols) fp32 */
int vila_norm_bwd_bf16(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, float* scratch, int rows, int cols,
                       float eps, int rms, int accumulate, vila_stream_t stream);
/* softmax-CE over fp32 logits rows: loss += sum_i (lse_i - z_i[label_i]) * scale ; dlogits = (softmax - onehot) * scale (bf16);
 * row_loss = scratch of `rows` floats (the per-row losses, summed in a fixed order) */
int vila_ce_loss_f32(const float* logits, const int64_t* labels, void* dlogits, float* loss, float* row_loss, int rows, int vocab, int64_t ld_logits,
                     float scale, vila_stream_t stream);
/* dst[rows[i]] += src[i] (embedding gradient); repeated ids are summed in fp32 in ascending i by the first occurrence's block and rounded once */
int vila_scatter_add_rows_bf16(const void* src, void* dst, const int32_t* rows, int n, int hidden, vila_stream_t stream);
int vila_depth_to_space_bf16(const void* dy, void* dx, int n_images, int grid, int channels, int k, vila_stream_t stream);
int vila_im2col_bf16(const void* pixels, void* out, int n_images, int channels, int H, int W, int patch, int k_padded, vila_stream_t stream);
int vila_rope_table_f32(const int32_t* positions, float* cos_out, float* sin_out, int n_tokens, int head_dim, float theta, vila_stream_t stream);
int vila_rope_fwd_bf16(void* qkv, const float* cos_t, const float* sin_t, const int32_t* positions, int n_tokens, int q_heads, int kv_heads,
                       int head_dim, vila_stream_t stream);
int vila_rope_bwd_bf16(void* dqkv, const float* cos_t, const float* sin_t, int n_tokens, int q_heads, int kv_heads, int head_dim, vila_stream_t stream);
/* flash-attention backward; tok_strides / head_strides: [8] host arrays for q,k,v,o,do,dq,dk,dv (elements) */
int vila_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                       const int64_t* tok_strides, const int32_t* head_strides, const int32_t* cu_seqlens, int n_seq, int total_tokens,
                       int max_seqlen, int n_q_heads, int n_kv_heads, int head_dim, int causal, float scale, const float* lse, float* delta,
                       vila_stream_t stream);
/* the same in separately launchable parts (bit mask): 1 = delta = rowsum(dO o O) (needed by the other two), 2 = dQ, 4 = dK / dV.  The parts
 * share no output: once delta is done a trainer runs dQ and dK / dV on different streams (vila_amd/train.py, VILA_SFT_ATTN_STREAM) */
int vila_attn_bwd_bf16_parts(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                             const int64_t* tok_strides, const int32_t* head_strides, const int32_t* cu_seqlens, int n_seq, int total_tokens,
                             int max_seqlen, int n_q_heads, int n_kv_heads, int head_dim, int causal, float scale, const float* lse, float* delta,
                             int parts, vila_stream_t stream);
/* torch.optim.AdamW semantics on flat buffers: fp32 master/m/v, bf16 grad in (times grad_scale), bf16 param out */
int vila_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, vila_stream_t stream);
/* the same update from a kernel that allocates <= 32 VGPRs per lane, so that it can be co-resident with a 256x256 GEMM block on a side
 * stream (per-bucket optimizer overlapped with the backward of the layers below); results identical to vila_adamw_step */
int vila_adamw_step_lean(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, vila_stream_t stream);
/* *out += sum x^2 (global gradient norm); scratch = VILA_SUMSQ_SCRATCH_FLOATS floats */
int vila_sumsq_bf16(const void* x, int64_t n, float* out, float* scratch, vila_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------
 * The whole SFT forward + backward as ONE call — replaces what autograd does under the reference's
 * patched `Trainer.training_step` (llava/train/transformer_normalize_monkey_patch.py:183-249): LlavaLlamaModel.forward
 * (llava_llama.py:94-159) on a packed batch, loss = sum CE * loss_scale (loss_scale = 1 / GLOBAL num_items, :261-268), gradients of
 * every parameter of the LLM, the mm_projector and the vision tower.
 *   - weights / gradients: the inference structs; `*_grad` mirrors the weight struct with pointers to the bf16 gradient tensors of
 *     the same shapes (q/k/v fused like the weights).  The caller ZEROES the gradient buffers before the call (embedding rows are
 *     scatter-added; parameters the path never touches keep zero).  Every other gradient tensor is overwritten.
 *   - VilaSftBatch: the integer work of `_embed` / `repack_multimodal_data` (llava_arch.py:412-490, 744-800) planned on the host
 *     (vila_amd/host.py), all index arrays on the device
 *   - workspace: saved activations + temporaries (vila_sft_workspace_bytes; about 4 GB per 769-token sample at NVILA-8B), no
 *     re-computation
 *   - cb: called on the HOST, on the calling thread, right after the last kernel that writes a gradient bucket has been enqueued on
 *     `stream` (buckets in backward order: LM_HEAD (untied head only), FINAL_NORM, LLM_LAYER n-1..0, EMBED, PROJECTOR, VIT_LAYER n-1..0,
 *     VIT_EMBED).  That is where a data-parallel caller records an event and starts the bucket's all-reduce / optimizer on its own
 *     streams; nullable.
 * ------------------------------------------------------------------------------------------------------------ */
enum { VILA_BUCKET_LM_HEAD = 0, VILA_BUCKET_FINAL_NORM = 1, VILA_BUCKET_LLM_LAYER = 2, VILA_BUCKET_EMBED = 3, VILA_BUCKET_PROJECTOR = 4,
       VILA_BUCKET_VIT_LAYER = 5, VILA_BUCKET_VIT_EMBED = 6 };
typedef void (*VilaGradReadyCb)(void* arg, int bucket, int index);
typedef struct {
    const void* pixels; int n_images;                 /* [n_images, C, image, image] bf16 */
    int total_tokens;                                  /* T: tokens of the packed row */
    const int32_t* txt_src; const int32_t* txt_dst; int n_txt;       /* embed_tokens row (token id) -> packed row */
    const int32_t* feat_src; const int32_t* feat_dst; int n_feat;    /* mm_projector output row -> packed row */
    const int32_t* nl_src; const int32_t* nl_dst; int n_nl;          /* the "\n" end token of every image (ids) -> packed row */
    const int32_t* positions;                          /* [T] restarted per sample (llava_arch.py:751) */
    const int32_t* cu_seqlens; int n_seq; int max_seqlen;            /* flash-attn varlen description (model/utils/packing.py:12-21) */
    const int32_t* target_rows; const int64_t* targets; int n_targets;  /* packed rows that predict a label, and the labels */
    float loss_scale;
    /* dynamic_s2 recipe (scripts/NVILA/stage1_9tile.sh:19-22; llava_arch.py:298-390): s2_n_blocks > 0 => `pixels` holds the n_images TILES
     * of every scale of every image, the tower output goes through vila_s2_merge_bf16 (s2_desc, device [s2_n_blocks][6]) into the
     * projector (in_dim = n_scales * tower hidden, s2_n_blocks inputs), feat_src indexes the projector's [s2_n_blocks * tokens] rows
     * (the final chessboard merge is folded into it by the host plan), and the backward runs vila_s2_merge_bwd_bf16 (s2_tile_desc,
     * device [n_images][8]).  s2_n_blocks = 0: the plain single-scale path (the fields are ignored). */
    const int32_t* s2_desc; const int32_t* s2_tile_desc; int s2_n_blocks; int s2_n_scales; int32_t s2_splits[4];
    /* pooling video encoder (TSPVideoEncoder, encoders/video/tsp.py:28-52).  n_pools > 0 => behind the projector's rows the step keeps
     * n_media_rows - (projector rows) POOLED rows: pools = HOST array [n_pools][7] = {first projector input of the video's frames, n_frames,
     * pool_t, pool_h, pool_w, first row of this (video, pool size)'s pooled rows in the buffer, their count}; forward = vila_video_pool_bf16
     * per entry, backward = vila_video_pool_bwd_bf16 accumulated onto the frames' projector rows.  feat_src then indexes the whole buffer
     * [projector rows | pooled rows].  n_pools = 0: feat_src indexes the projector rows only (n_media_rows is ignored). */
    const int32_t* pools; int n_pools; int n_media_rows;
} VilaSftBatch;
size_t vila_sft_workspace_bytes(const VilaVitWeights* vit, const VilaProjWeights* proj, const VilaLlmWeights* llm, const VilaSftBatch* batch);
int vila_sft_fwd_bwd(const VilaVitWeights* vit, const VilaVitWeights* vit_grad, const VilaProjWeights* proj, const VilaProjWeights* proj_grad,
                     const VilaLlmWeights* llm, const VilaLlmWeights* llm_grad, const VilaSftBatch* batch, float* loss_out /* device */,
                     void* workspace, size_t workspace_bytes, VilaGradReadyCb cb, void* cb_arg, vila_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * dynamic_s2 multi-scale path — replaces merge_features_for_dynamic_s2 + split_chessboard +
 * rearrange of LlavaMetaModel.encode_images (llava/model/llava_arch.py:298-379):
 *   feats [n_tiles, g*g, C] (tower output for every tile of every image, scales in ascending order per image)
 *   desc  device [n_blocks][6] = {first tile of the image, bh | obh << 16, bw | obw << 16, block row i, block col j, single (block_sizes None)}
 *         bh x bw = tiles of the image's last scale; obh x obw = its OUTPUT blocks: 0 in the high halves = the same grid
 *         (s2_resize_output_to_scale_idx = -1, every shipped recipe); s x s for an earlier scale of s x s tiles (llava_arch.py:340-358)
 *   out   [n_blocks, g*g, n_scales*C] = the projector input.  splits[k] = scales[k] / scales[0] for k < n_scales-1 [host].
 * ------------------------------------------------------------------------------------------------------------ */
int vila_s2_merge_bf16(const void* feats, void* out, const int32_t* desc, int n_blocks, int grid, int channels, int n_scales,
                       const int32_t* splits, vila_stream_t stream);
/* Backward of the above for the SFT step of the dynamic_s2 recipe (autograd through llava_arch.py:298-379: F.interpolate(mode="area")
 * backward = dy / |window| broadcast into each scale's chessboard, merge / split_chessboard adjoints), as one gather:
 *   dy [n_blocks, g*g, n_scales*C] -> dx [n_tiles, g*g, C] (every element written, no accumulation)
 *   tile_desc device [n_tiles][8] = {first output block of the tile's image, bh | obh << 16, bw | obw << 16, scale index, tile row, tile col, single, 0} */
int vila_s2_merge_bwd_bf16(const void* dy, void* dx, const int32_t* tile_desc, int n_tiles, int grid, int channels, int n_scales,
                           const int32_t* splits, vila_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------
 * Video encoders — replaces `pool` + `_process_features` of TSPVideoEncoder
 * (llava/model/encoders/video/tsp.py:10-11,28-52) and `_process_features` of BasicVideoEncoder (video/basic.py:30-41):
 *   feats [n_frames, grid*grid, C] bf16 (projected frames of ONE video) ->
 *   out   [(n_frames/pool_t) * (n_start + (grid/pool_h)*(grid/pool_w) + n_end), C]: per pooled frame the start-token rows, the mean over
 *   every (pool_t, pool_h, pool_w) window (fp32 accumulate), the end-token rows.  pool = (1,1,1) is BasicVideoEncoder.
 *   A ragged split (a pooled dimension not divisible by its pool size) is an error, as the reference's view() raises.
 * ------------------------------------------------------------------------------------------------------------ */
int vila_video_pool_bf16(const void* feats, void* out, int n_frames, int grid, int channels, int pool_t, int pool_h, int pool_w,
                         const void* start_rows, int n_start, const void* end_rows, int n_end, vila_stream_t stream);
/* Adjoint of the pooling for the SFT step (row a13 over a7; the `mean` backward of tsp.py:10-11 under autograd):
 *   dpooled [(n_frames/pool_t) * (grid/pool_h)(grid/pool_w), C] (gradient of the pooled FEATURE rows only) ->
 *   dfeats  [n_frames, grid*grid, C]: dfeats[t][h][w] = dpooled[t/pool_t][h/pool_h][w/pool_w] / (pool_t pool_h pool_w);
 *   accumulate != 0 adds into dfeats (the second and later pool sizes of one video). */
int vila_video_pool_bwd_bf16(const void* dpooled, void* dfeats, int n_frames, int grid, int channels, int pool_t, int pool_h, int pool_w,
                             int accumulate, vila_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * W4A16 decode (BASELINE configs[4]).  The reference's W4A16 backend is the external TinyChat
 * (README.md:87), with no code in-tree: the packed format is defined in vila_amd/csrc/gemv_w4.hip and produced by
 * vila_amd/quant.py; parity is against a CPU dequantise-then-fp32 oracle of the same quantised weights.
 *   w = (q - zero) * scale, uint4 q, groups of 128 along K.  Tile-major HBM layout (16-row tiles, one 1-KB wave load per group):
 *   Wq  [N/16][K/128][64 lanes][4] u32   lane = 16*g + n holds k = 128*group + 32*g + (0..31) of row n; per u32 nibble p<4 = element 2p,
 *                                        nibble p+4 = element 2p+1
 *   Wsz [N/16][K/128][16] u32            {bf16 scale, bf16 128 + zero}
 *   Row order inside a matrix: o/down as the bf16 weight; gate/up interleaved (row 2i = gate i, 2i+1 = up i); q and k heads of
 *   the fused qkv interleaved so RoPE partners are neighbours (row 2i = element i, 2i+1 = element i + head_dim/2), v heads as is.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const void *qkv_q, *qkv_sz, *o_q, *o_sz, *gateup_q, *gateup_sz, *down_q, *down_sz;
} VilaLlmLayerW4;
/* N outputs.  mode 0: y = W x (+bias)(+residual); mode 1: W holds 2N interleaved gate/up rows, y = silu(Wg x) * (Wu x);
 * optional fused RMSNorm on x.  K % 128 != 0 -> -1 */
int vila_gemv_w4_bf16(const void* x, const void* norm_w, float eps, const void* Wq, const void* Wsz,
                      const void* bias, const void* residual, void* y, int N, int K, int mode, vila_stream_t stream);
/* same contract as vila_llm_decode_step; `w` still supplies embed, norms, q/k/v biases and the bf16 lm_head */
int vila_llm_decode_step_w4(const VilaLlmWeights* w, const VilaLlmLayerW4* qlayers /*[host]*/, const VilaKvCache* cache,
                            const VilaDecodeState* st, void* workspace, size_t workspace_bytes, vila_stream_t stream);
/* the same step with a stochastic pick (generate(do_sample=True) on the quantised decoder): temperature -> top-k -> top-p -> draw */
int vila_llm_decode_step_w4_sample(const VilaLlmWeights* w, const VilaLlmLayerW4* qlayers /*[host]*/, const VilaKvCache* cache,
                            const VilaDecodeState* st, void* workspace, size_t workspace_bytes, const VilaSampling* sampling, vila_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
