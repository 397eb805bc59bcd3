// C-ABI of libvila_hip.so (include/vila_hip.h): model-level chaining of the kernels on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "../../include/vila_hip.h"
#include "../../include/vila_hip_tuning.h"
#include "kernels.h"
#include "decode_persist.h"
#include "train.h"
#include "w4.h"

static thread_local char g_err[512] = "";
void vila_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vila_last_error(void) { return g_err; }
extern "C" int vila_abi_version(void) { return 1; }

int gemm256_tiles_m_of(int M);      // gemm256.hip

namespace {
struct Arena {
    char* base; size_t size, off;
    Arena(void* p, size_t n) : base((char*)p), size(n), off(0) {}
    template <typename T> T* take(size_t count) {
        off = align_up(off, 256);
        T* r = (T*)(base + off);
        off += count * sizeof(T);
        return r;
    }
    bool ok() const { return off <= size; }
};
}  // namespace
hipStream_t vila_stream_enter(void* s);      // (kernels.h) the same for the other translation units
namespace {
// Every entry point takes its device from the stream handle (backward runs on autograd worker threads whose thread-local HIP
// device need not be the caller's): a non-default stream names its device, and the calling thread is switched to it before anything is
// launched.  The lookup is cached per thread and stream, so the steady state costs one compare.
inline hipStream_t S(vila_stream_t s) {
    hipStream_t st = (hipStream_t)s;
    if (st != nullptr) {
        static thread_local hipStream_t last = nullptr;
        static thread_local int last_dev = -1;
        if (st != last) {
            hipDevice_t dev = 0;
            if (hipStreamGetDevice(st, &dev) == hipSuccess) { last = st; last_dev = (int)dev; }
            else (void)hipGetLastError();
        }
        if (st == last && last_dev >= 0) {
            int cur = -1;
            if (hipGetDevice(&cur) == hipSuccess && cur != last_dev) (void)hipSetDevice(last_dev);
        }
    }
    return st;
}
}  // namespace
hipStream_t vila_stream_enter(void* s) { return S((vila_stream_t)s); }
namespace {
inline const bf16_t* B(const void* p) { return (const bf16_t*)p; }
inline bf16_t* B(void* p) { return (bf16_t*)p; }

// the NEXT block's normalisation, offered to a GEMM whose split-K reduce can take it along (kernels.h GemmArgs::norm_*)
struct NextNorm { const void* w = nullptr; const void* b = nullptr; float eps = 0.f; int rms = 0; bf16_t* out = nullptr; int* done = nullptr; };
// the q/k/v projection's follow-up (bias -> RoPE -> K / V into the cache), offered to its GEMM in the same way (kernels.h GemmArgs::rope_*)
struct NextRope { const float* cs = nullptr; const float* sn = nullptr; const int32_t* pos = nullptr; const int32_t* seq = nullptr; bf16_t* kc = nullptr; bf16_t* vc = nullptr;
                  int nq = 0, nkv = 0, hd = 0, max_ctx = 0; int* done = nullptr; };
int gemm(const bf16_t* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const bf16_t* res, int64_t ldr,
         void* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s, const void* W2 = nullptr, int out_f32 = 0,
         float* ws = nullptr, size_t ws_bytes = 0, int res_mod = 0, const NextNorm* nn = nullptr, const NextRope* nr = nullptr) {
    GemmArgs g;
    if (nr != nullptr) { g.rope_cs = nr->cs; g.rope_sn = nr->sn; g.rope_pos = nr->pos; g.rope_seq = nr->seq; g.rope_kc = nr->kc; g.rope_vc = nr->vc;
                         g.rope_nq = nr->nq; g.rope_nkv = nr->nkv; g.rope_hd = nr->hd; g.rope_max_ctx = nr->max_ctx; g.rope_done = nr->done; }
    if (nn != nullptr) { g.norm_w = B(nn->w); g.norm_b = B(nn->b); g.norm_eps = nn->eps; g.norm_rms = nn->rms; g.norm_out = nn->out; g.norm_done = nn->done; }
    g.ws = ws; g.ws_bytes = ws_bytes; g.res_mod = res_mod;
    g.A = A; g.lda = lda; g.W = B(W); g.ldw = ldw; g.W2 = B(W2); g.bias = B(bias); g.residual = res; g.ldr = ldr;
    g.C = C; g.ldc = ldc; g.out_f32 = out_f32; g.M = M; g.N = N; g.K = K; g.epi = epi;
    return launch_gemm(g, s);
}
}  // namespace

// =================================================================================================
// Vision tower
// =================================================================================================
static inline int vit_kp(const VilaVitShape* s) { return (int)align_up((size_t)s->channels * s->patch * s->patch, 8); }

// fc2 (K = 4304) of one or two images is 20-40 tiles of 256^2: its weights arrive cold from HBM and a tile's K loop runs at memory latency,
// so it is sliced over K (8 slices of one image: 45.6 -> 33.7 us, tools/gemm_bench precold); beyond 2048 rows the tiles fill the chip
static size_t vit_splitk_rows(size_t M) { return M <= 2048 ? M : 0; }
extern "C" size_t vila_vit_workspace_bytes(const VilaVitShape* s, int n_images) {
    const size_t g = s->image / s->patch, M = (size_t)n_images * g * g, D = s->hidden, F = s->inter, Kp = vit_kp(s);
    size_t b = 0;
    b += align_up(M * Kp * 2, 256) + align_up(D * Kp * 2, 256);     // patches, padded conv weight
    b += 2 * align_up(M * D * 2, 256);                               // x, h
    b += align_up(M * 3 * D * 2, 256);                               // qkv
    b += align_up(M * F * 2, 256);                                   // mlp hidden
    b += align_up(vit_splitk_rows(M) * D * 4 * 8, 256);              // fp32 K-slices of fc2 when its 256^2 tiles cannot fill the chip
    return b + 4096;
}

extern "C" int vila_vit_forward(const VilaVitWeights* w, const void* pixels, int n_images, void* out,
                                void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    const VilaVitShape& sh = w->shape;
    hipStream_t s = S(stream);
    VILA_REQUIRE(n_images > 0, "vit: n_images must be positive");
    VILA_REQUIRE(sh.image % sh.patch == 0, "vit: image size %d is not a multiple of patch size %d", sh.image, sh.patch);
    VILA_REQUIRE(sh.hidden % sh.heads == 0, "embed_dim must be divisible by num_heads (got `embed_dim`: %d and `num_heads`: %d).", sh.hidden, sh.heads);
    VILA_REQUIRE(workspace_bytes >= vila_vit_workspace_bytes(&sh, n_images), "vit: workspace too small");
    const int g = sh.image / sh.patch, N = g * g, M = n_images * N, D = sh.hidden, F = sh.inter, hd = D / sh.heads;
    const int Kc = sh.channels * sh.patch * sh.patch, Kp = vit_kp(&sh);
    Arena a(workspace, workspace_bytes);
    bf16_t* patches = a.take<bf16_t>((size_t)M * Kp);
    bf16_t* wpad = a.take<bf16_t>((size_t)D * Kp);
    bf16_t* x = a.take<bf16_t>((size_t)M * D);
    bf16_t* h = a.take<bf16_t>((size_t)M * D);
    bf16_t* qkv = a.take<bf16_t>((size_t)M * 3 * D);
    bf16_t* f = a.take<bf16_t>((size_t)M * F);
    const size_t sk_bytes = vit_splitk_rows(M) * D * 4 * 8;
    float* skws = sk_bytes ? a.take<float>(sk_bytes / 4) : nullptr;
    VILA_REQUIRE(a.ok(), "vit: workspace arena overflow");

    // a2: patch embed = im2col + ONE GEMM over all images (+bias) with the position embedding as a periodic residual operand
    // (row m of the batch reads pos_emb[m % N]): 64 frames are one launch with one tail, not 64
    VILA_TRY(launch_im2col(B(pixels), patches, n_images, sh.channels, sh.image, sh.image, sh.patch, Kp, s));
    VILA_TRY(launch_pad_rows(B(w->patch_w), wpad, D, Kc, Kp, s));
    bf16_t* x0 = (sh.n_layers_run == 0) ? B(out) : x;
    VILA_TRY(gemm(patches, Kp, wpad, Kp, w->patch_b, B(w->pos_emb), D, x0, D, M, D, Kp, EPI_NONE, s, nullptr, 0, nullptr, 0, N));

    int ln1_done = 0;                                           // the previous layer's fc2 reduce already wrote layer_norm1(x) into h
    for (int l = 0; l < sh.n_layers_run; ++l) {
        const VilaVitLayer& L = w->layers[l];
        bf16_t* xo = (l == sh.n_layers_run - 1) ? B(out) : x;   // last layer writes straight into `out`
        if (!ln1_done) VILA_TRY(launch_layernorm(x, B(L.ln1_w), B(L.ln1_b), h, M, D, sh.ln_eps, s));
        ln1_done = 0;
        const bool fused = (B(L.wk) == B(L.wq) + (size_t)D * D) && (B(L.wv) == B(L.wk) + (size_t)D * D) &&
                           (B(L.bk) == B(L.bq) + D) && (B(L.bv) == B(L.bk) + D);
        if (fused) {
            VILA_TRY(gemm(h, D, L.wq, D, L.bq, nullptr, 0, qkv, 3 * D, M, 3 * D, D, EPI_NONE, s));
        } else {
            VILA_TRY(gemm(h, D, L.wq, D, L.bq, nullptr, 0, qkv, 3 * D, M, D, D, EPI_NONE, s));
            VILA_TRY(gemm(h, D, L.wk, D, L.bk, nullptr, 0, qkv + D, 3 * D, M, D, D, EPI_NONE, s));
            VILA_TRY(gemm(h, D, L.wv, D, L.bv, nullptr, 0, qkv + 2 * D, 3 * D, M, D, D, EPI_NONE, s));
        }
        AttnArgs at{};
        at.q = qkv; at.k = qkv + D; at.v = qkv + 2 * D; at.o = h;
        at.q_tok_stride = at.k_tok_stride = at.v_tok_stride = 3 * D; at.o_tok_stride = D;
        at.q_head_stride = at.k_head_stride = at.v_head_stride = at.o_head_stride = hd;
        at.cu_seqlens = nullptr; at.n_seq = n_images; at.total_tokens = M; at.max_seqlen = N;
        at.n_q_heads = at.n_kv_heads = sh.heads; at.head_dim = hd; at.causal = 0; at.scale = 1.0f / sqrtf((float)hd);
        at.lse = nullptr;
        VILA_TRY(launch_attn_fwd(at, s));
        VILA_TRY(gemm(h, D, L.wo, D, L.bo, x, D, x, D, M, D, D, EPI_NONE, s));               // x += out_proj(attn)
        VILA_TRY(launch_layernorm(x, B(L.ln2_w), B(L.ln2_b), h, M, D, sh.ln_eps, s));
        VILA_TRY(gemm(h, D, L.fc1_w, D, L.fc1_b, nullptr, 0, f, F, M, F, D, EPI_GELU_TANH, s));
        // x += fc2(gelu(fc1)); where fc2 is K-sliced its reduce takes the NEXT layer's layer_norm1 along (h is free: fc1 has consumed it)
        NextNorm nn;
        if (l + 1 < sh.n_layers_run) { nn.w = w->layers[l + 1].ln1_w; nn.b = w->layers[l + 1].ln1_b; nn.eps = sh.ln_eps; nn.rms = 0; nn.out = h; nn.done = &ln1_done; }
        VILA_TRY(gemm(f, F, L.fc2_w, F, L.fc2_b, x, D, xo, D, M, D, F, EPI_NONE, s, nullptr, 0, skws, sk_bytes, 0, l + 1 < sh.n_layers_run ? &nn : nullptr));
    }
    return 0;
}

// W8A8 vision tower (SURVEY.md §8f row 3 / BASELINE configs[4]): the four linears of every encoder layer run int8 x int8 on the matrix
// cores (per-output-channel weight scales, per-token dynamic activation scales); patch embedding, LayerNorms, attention, biases,
// residual stream stay bf16.  Same call contract as vila_vit_forward.
extern "C" size_t vila_vit_w8a8_workspace_bytes(const VilaVitShape* s, int n_images) {
    const size_t g = s->image / s->patch, M = (size_t)n_images * g * g, F = s->inter, D = s->hidden;
    return vila_vit_workspace_bytes(s, n_images) + align_up(M * (F > D ? F : D), 256) + align_up(M * 4, 256) + 1024;
}
extern "C" int vila_vit_forward_w8a8(const VilaVitWeights* w, const VilaVitLayerW8* ql, const void* pixels, int n_images, void* out,
                                     void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    const VilaVitShape& sh = w->shape;
    hipStream_t s = S(stream);
    VILA_REQUIRE(n_images > 0 && ql != nullptr, "vit_w8a8: n_images must be positive and the int8 layers given");
    VILA_REQUIRE(sh.image % sh.patch == 0 && sh.hidden % sh.heads == 0, "vit_w8a8: bad shape");
    VILA_REQUIRE(sh.hidden % 16 == 0 && sh.inter % 16 == 0, "vit_w8a8: hidden (%d) and intermediate (%d) sizes must be multiples of 16", sh.hidden, sh.inter);
    VILA_REQUIRE(workspace_bytes >= vila_vit_w8a8_workspace_bytes(&sh, n_images), "vit_w8a8: workspace too small");
    const int g = sh.image / sh.patch, N = g * g, M = n_images * N, D = sh.hidden, F = sh.inter, hd = D / sh.heads;
    const int Kc = sh.channels * sh.patch * sh.patch, Kp = vit_kp(&sh);
    Arena a(workspace, workspace_bytes);
    bf16_t* patches = a.take<bf16_t>((size_t)M * Kp);
    bf16_t* wpad = a.take<bf16_t>((size_t)D * Kp);
    bf16_t* x = a.take<bf16_t>((size_t)M * D);
    bf16_t* h = a.take<bf16_t>((size_t)M * D);
    bf16_t* qkv = a.take<bf16_t>((size_t)M * 3 * D);
    bf16_t* f = a.take<bf16_t>((size_t)M * F);
    int8_t* xq = a.take<int8_t>((size_t)M * (F > D ? F : D));
    float* sx = a.take<float>((size_t)M);
    VILA_REQUIRE(a.ok(), "vit_w8a8: workspace arena overflow");
    VILA_TRY(launch_im2col(B(pixels), patches, n_images, sh.channels, sh.image, sh.image, sh.patch, Kp, s));
    VILA_TRY(launch_pad_rows(B(w->patch_w), wpad, D, Kc, Kp, s));
    bf16_t* x0 = (sh.n_layers_run == 0) ? B(out) : x;
    VILA_TRY(gemm(patches, Kp, wpad, Kp, w->patch_b, B(w->pos_emb), D, x0, D, M, D, Kp, EPI_NONE, s, nullptr, 0, nullptr, 0, N));
    for (int l = 0; l < sh.n_layers_run; ++l) {
        const VilaVitLayer& L = w->layers[l];
        const VilaVitLayerW8& Q = ql[l];
        bf16_t* xo = (l == sh.n_layers_run - 1) ? B(out) : x;
        VILA_TRY(launch_layernorm(x, B(L.ln1_w), B(L.ln1_b), h, M, D, sh.ln_eps, s));
        VILA_TRY(launch_quant_rows_i8(h, xq, sx, M, D, s));
        VILA_TRY(launch_gemm_i8(xq, D, (const int8_t*)Q.wqkv_q, D, sx, Q.wqkv_s, B(L.bq), nullptr, 0, qkv, 3 * D, M, 3 * D, D, EPI_NONE, s));
        AttnArgs at{};
        at.q = qkv; at.k = qkv + D; at.v = qkv + 2 * D; at.o = h;
        at.q_tok_stride = at.k_tok_stride = at.v_tok_stride = 3 * D; at.o_tok_stride = D;
        at.q_head_stride = at.k_head_stride = at.v_head_stride = at.o_head_stride = hd;
        at.cu_seqlens = nullptr; at.n_seq = n_images; at.total_tokens = M; at.max_seqlen = N;
        at.n_q_heads = at.n_kv_heads = sh.heads; at.head_dim = hd; at.causal = 0; at.scale = 1.0f / sqrtf((float)hd);
        at.lse = nullptr;
        VILA_TRY(launch_attn_fwd(at, s));
        VILA_TRY(launch_quant_rows_i8(h, xq, sx, M, D, s));
        VILA_TRY(launch_gemm_i8(xq, D, (const int8_t*)Q.wo_q, D, sx, Q.wo_s, B(L.bo), x, D, x, D, M, D, D, EPI_NONE, s));            // x += out_proj(attn)
        VILA_TRY(launch_layernorm(x, B(L.ln2_w), B(L.ln2_b), h, M, D, sh.ln_eps, s));
        VILA_TRY(launch_quant_rows_i8(h, xq, sx, M, D, s));
        VILA_TRY(launch_gemm_i8(xq, D, (const int8_t*)Q.fc1_q, D, sx, Q.fc1_s, B(L.fc1_b), nullptr, 0, f, F, M, F, D, EPI_GELU_TANH, s));
        VILA_TRY(launch_quant_rows_i8(f, xq, sx, M, F, s));
        VILA_TRY(launch_gemm_i8(xq, F, (const int8_t*)Q.fc2_q, F, sx, Q.fc2_s, B(L.fc2_b), x, D, xo, D, M, D, F, EPI_NONE, s));      // x += fc2(gelu(fc1))
    }
    return 0;
}
extern "C" int vila_quant_rows_i8(const void* x, void* q, float* scale, int rows, int cols, vila_stream_t stream) {
    return launch_quant_rows_i8(B(x), (int8_t*)q, scale, rows, cols, S(stream));
}
extern "C" int vila_gemm_w8a8(const void* Aq, int64_t lda, const void* Wq, int64_t ldw, const float* sx, const float* sw, const void* bias,
                              const void* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int epi, vila_stream_t stream) {
    return launch_gemm_i8((const int8_t*)Aq, lda, (const int8_t*)Wq, ldw, sx, sw, B(bias), B(residual), ldr, B(C), ldc, M, N, K, epi, S(stream));
}

// =================================================================================================
// Projector
// =================================================================================================
extern "C" int vila_proj_out_tokens(int kind, int n_tokens) {
    const int g = (int)(sqrt((double)n_tokens) + 0.5);
    const int k = (kind == VILA_PROJ_MLP_DOWNSAMPLE_3X3_FIX) ? 3 : 2;
    const int gd = (g + k - 1) / k;
    return gd * gd;
}
extern "C" size_t vila_proj_workspace_bytes(const VilaProjWeights* w, int n_images, int n_tokens) {
    const int k = (w->kind == VILA_PROJ_MLP_DOWNSAMPLE_3X3_FIX) ? 3 : 2;
    const size_t T = (size_t)n_images * vila_proj_out_tokens(w->kind, n_tokens);
    const size_t C = w->in_dim, H = w->out_dim;
    return align_up(T * k * k * C * 2, 256) + align_up(T * 3 * C * 2, 256) + 2 * align_up(T * H * 2, 256) + 4096;
}
extern "C" int vila_proj_forward(const VilaProjWeights* w, const void* feat, int n_images, int n_tokens, void* out,
                                 void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    hipStream_t s = S(stream);
    const int g = (int)(sqrt((double)n_tokens) + 0.5);
    VILA_REQUIRE(g * g == n_tokens, "projector: token count %d is not a perfect square", n_tokens);
    VILA_REQUIRE(w->kind >= 0 && w->kind <= 2, "Unknown projector type: %d", w->kind);
    VILA_REQUIRE(workspace_bytes >= vila_proj_workspace_bytes(w, n_images, n_tokens), "projector: workspace too small");
    const int k = (w->kind == VILA_PROJ_MLP_DOWNSAMPLE_3X3_FIX) ? 3 : 2;
    const int T = n_images * vila_proj_out_tokens(w->kind, n_tokens);
    const int C = w->in_dim, H = w->out_dim, C1 = k * k * C;
    Arena a(workspace, workspace_bytes);
    bf16_t* y = a.take<bf16_t>((size_t)T * C1);
    bf16_t* mid = a.take<bf16_t>((size_t)T * 3 * C);
    bf16_t* h1 = a.take<bf16_t>((size_t)T * H);
    VILA_REQUIRE(a.ok(), "projector: workspace arena overflow");
    VILA_TRY(launch_space_to_depth(B(feat), y, n_images, g, C, k, s));
    VILA_TRY(launch_layernorm(y, B(w->ln1_w), B(w->ln1_b), y, T, C1, 1e-5f, s));
    if (k == 2) {
        VILA_TRY(gemm(y, C1, w->fc1_w, C1, w->fc1_b, nullptr, 0, h1, H, T, H, C1, EPI_GELU_ERF, s));
        VILA_TRY(gemm(h1, H, w->fc2_w, H, w->fc2_b, nullptr, 0, out, H, T, H, H, EPI_NONE, s));
    } else {
        VILA_TRY(gemm(y, C1, w->fc1_w, C1, w->fc1_b, nullptr, 0, mid, 3 * C, T, 3 * C, C1, EPI_GELU_ERF, s));
        VILA_TRY(launch_layernorm(mid, B(w->ln2_w), B(w->ln2_b), mid, T, 3 * C, 1e-5f, s));
        VILA_TRY(gemm(mid, 3 * C, w->fc2_w, 3 * C, w->fc2_b, nullptr, 0, h1, H, T, H, 3 * C, EPI_GELU_ERF, s));
        VILA_TRY(gemm(h1, H, w->fc3_w, H, w->fc3_b, nullptr, 0, out, H, T, H, H, EPI_NONE, s));
    }
    return 0;
}

// =================================================================================================
// Embedding / splice
// =================================================================================================
extern "C" int vila_embed_tokens(const void* table, int64_t vocab, int hidden, const int64_t* ids, int n, void* out, vila_stream_t stream) {
    return launch_embed_gather(B(table), ids, B(out), n, hidden, vocab, S(stream));
}
extern "C" int vila_copy_rows(const void* src, void* dst, const int32_t* src_row, const int32_t* dst_row, int n, int hidden, vila_stream_t stream) {
    return launch_copy_rows(B(src), B(dst), src_row, dst_row, n, hidden, S(stream));
}

// =================================================================================================
// LLM prefill
// =================================================================================================
// VILA_PREFILL_OPROJ_SPLITK=0: o_proj never takes the K-sliced path for the sake of the fused post-attention norm (A/B switch)
static bool prefill_oproj_norm();
static bool prefill_oproj_norm_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_PREFILL_OPROJ_SPLITK"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
// tuning / test hook (vila_hip_tuning.h): -1 = the environment's choice (default on), 0 / 1 = off / on
static int g_prefill_qkv_rope = -1, g_prefill_oproj_norm = -1;
extern "C" void vila_prefill_force_fusions(int qkv_rope, int oproj_norm) { g_prefill_qkv_rope = qkv_rope; g_prefill_oproj_norm = oproj_norm; }
static bool prefill_oproj_norm() { return g_prefill_oproj_norm >= 0 ? g_prefill_oproj_norm != 0 : prefill_oproj_norm_env(); }
static bool prefill_qkv_rope() {
    if (g_prefill_qkv_rope >= 0) return g_prefill_qkv_rope != 0;
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_PREFILL_QKV_SPLITK"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
extern "C" size_t vila_llm_prefill_workspace_bytes(const VilaLlmShape* s, int T) {
    const size_t H = s->hidden, F = s->inter, QKV = (size_t)(s->q_heads + 2 * s->kv_heads) * s->head_dim;
    size_t b = 0;
    b += 2 * align_up((size_t)T * H * 2, 256);                 // x, h
    b += align_up((size_t)T * QKV * 2, 256);                   // qkv
    b += align_up((size_t)T * F * 2, 256);                     // act
    b += 2 * align_up((size_t)T * (s->head_dim / 2) * 4, 256); // rope cos/sin
    b += align_up((size_t)(T > 8 ? T : 8) * H * 2, 256);       // gathered last rows / final norm (the pruned last layer keeps 2 x n_last <= 8 rows there)
    b += align_up((size_t)8 * T * H * 4, 256);                 // split-K fp32 slabs (down_proj: up to 8 slices; tail round of gate/up) at small T
    return b + 8192;
}

extern "C" int vila_llm_prefill(const VilaLlmWeights* w, const void* embeds, const int32_t* positions, const int32_t* cu_seqlens,
                                int n_seq, int T, int max_seqlen, const int32_t* seq_of_tok, const VilaKvCache* cache,
                                const int32_t* last_rows, int n_last, float* last_logits, float* all_logits, void* final_hidden,
                                void* layer_hidden, void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    const VilaLlmShape& sh = w->shape;
    hipStream_t s = S(stream);
    VILA_REQUIRE(T > 0 && n_seq > 0, "llm_prefill: empty input");
    VILA_REQUIRE(sh.q_heads % sh.kv_heads == 0, "llm: q heads must be a multiple of kv heads");
    VILA_REQUIRE(cu_seqlens != nullptr || n_seq == 1, "llm_prefill: n_seq > 1 needs cu_seqlens");
    VILA_REQUIRE(workspace_bytes >= vila_llm_prefill_workspace_bytes(&sh, T), "llm_prefill: workspace too small");
    const int H = sh.hidden, F = sh.inter, hd = sh.head_dim, QS = sh.q_heads * hd, KS = sh.kv_heads * hd, QKV = QS + 2 * KS;
    Arena a(workspace, workspace_bytes);
    bf16_t* x = a.take<bf16_t>((size_t)T * H);
    bf16_t* h = a.take<bf16_t>((size_t)T * H);
    bf16_t* qkv = a.take<bf16_t>((size_t)T * QKV);
    bf16_t* act = a.take<bf16_t>((size_t)T * F);
    float* cs = a.take<float>((size_t)T * hd / 2);
    float* sn = a.take<float>((size_t)T * hd / 2);
    bf16_t* lastbuf = a.take<bf16_t>((size_t)(T > 8 ? T : 8) * H);
    float* skws = a.take<float>((size_t)8 * T * H);
    const size_t skws_bytes = (size_t)8 * T * H * 4;
    VILA_REQUIRE(a.ok(), "llm_prefill: workspace arena overflow");
    if (cache != nullptr) VILA_REQUIRE(max_seqlen <= cache->max_ctx, "llm_prefill: sequence (%d) longer than the KV cache (%d)", max_seqlen, cache->max_ctx);

    VILA_HIP(hipMemcpyAsync(x, embeds, (size_t)T * H * 2, hipMemcpyDeviceToDevice, s));
    VILA_TRY(launch_rope_table(positions, cs, sn, T, hd, sh.rope_theta, s));
    bf16_t* taps = B(layer_hidden);
    if (taps) VILA_HIP(hipMemcpyAsync(taps, x, (size_t)T * H * 2, hipMemcpyDeviceToDevice, s));

    // only last-row logits wanted (generation): the last layer is finished for those rows alone (see below)
    const bool prune_last = final_hidden == nullptr && all_logits == nullptr && taps == nullptr && cache != nullptr && last_logits != nullptr &&
                            last_rows != nullptr && n_last >= 1 && n_last <= 4 && H % 8 == 0 && F % 8 == 0 && QS == H;
    // 1..16 leftover rows (T = 256 k + r) ride in the last row tile of the 256^2 GEMMs as an extra fragment (gemm256_kernel.h, EX); only when
    // that policy is switched off (VILA_GEMM_EX=0) do 1..4 leftover rows of the MLP go through the decode GEMVs as in rounds 1 / 2
    const bool ex_rows = gemm256_tiles_m_of(T) < cdiv(T, 256);
    const int tail_rows = (!ex_rows && T > 256 && T % 256 >= 1 && T % 256 <= 4 && H % 8 == 0 && F % 8 == 0) ? T % 256 : 0;
    const int Tg = T - tail_rows;     // rows of the gate/up and down GEMMs; the rest via GEMV
    int ln1_done = 0;                                           // the previous layer's down-proj reduce already wrote input_layernorm(x) into h
    for (int l = 0; l < sh.n_layers; ++l) {
        const VilaLlmLayer& L = w->layers[l];
        if (!ln1_done) VILA_TRY(launch_rmsnorm(x, B(L.ln1_w), h, T, H, sh.rms_eps, s));
        ln1_done = 0;
        const bool fused = (B(L.wk) == B(L.wq) + (size_t)QS * H) && (B(L.wv) == B(L.wk) + (size_t)KS * H) &&
                           (B(L.bk) == B(L.bq) + QS) && (B(L.bv) == B(L.bk) + KS);
        bf16_t* kc = nullptr; bf16_t* vc = nullptr; int max_ctx = 0;
        if (cache != nullptr) {
            const size_t per_layer = (size_t)cache->n_slots * sh.kv_heads * cache->max_ctx * hd;
            kc = B(cache->k) + l * per_layer; vc = B(cache->v) + l * per_layer; max_ctx = cache->max_ctx;
        }
        int rope_done = 0;
        if (fused) {
            // one GEMM for q | k | v; where its grid is K-sliced (S = 769: 54 tiles x 4 slices) the reduce adds the bias, rotates q and k and
            // writes K / V into the cache (VILA_PREFILL_QKV_SPLITK=0: ring GEMM + rope_kv_kernel as in rounds 1-5)
            NextRope nr;
            nr.cs = cs; nr.sn = sn; nr.pos = positions; nr.seq = seq_of_tok; nr.kc = kc; nr.vc = vc; nr.nq = sh.q_heads; nr.nkv = sh.kv_heads; nr.hd = hd;
            nr.max_ctx = max_ctx; nr.done = &rope_done;
            VILA_TRY(gemm(h, H, L.wq, H, L.bq, nullptr, 0, qkv, QKV, T, QKV, H, EPI_NONE, s, nullptr, 0, skws, skws_bytes, 0, nullptr, prefill_qkv_rope() ? &nr : nullptr));
        } else {
            VILA_TRY(gemm(h, H, L.wq, H, L.bq, nullptr, 0, qkv, QKV, T, QS, H, EPI_NONE, s));
            VILA_TRY(gemm(h, H, L.wk, H, L.bk, nullptr, 0, qkv + QS, QKV, T, KS, H, EPI_NONE, s));
            VILA_TRY(gemm(h, H, L.wv, H, L.bv, nullptr, 0, qkv + QS + KS, QKV, T, KS, H, EPI_NONE, s));
        }
        if (!rope_done) VILA_TRY(launch_rope_kv(qkv, cs, sn, positions, seq_of_tok, kc, vc, T, sh.q_heads, sh.kv_heads, hd, max_ctx, s));
        AttnArgs at{};
        at.q = qkv; at.k = qkv + QS; at.v = qkv + QS + KS; at.o = h;
        at.q_tok_stride = at.k_tok_stride = at.v_tok_stride = QKV; at.o_tok_stride = QS;
        at.q_head_stride = at.k_head_stride = at.v_head_stride = at.o_head_stride = hd;
        at.cu_seqlens = cu_seqlens; at.n_seq = n_seq; at.total_tokens = T; at.max_seqlen = (cu_seqlens ? max_seqlen : T);
        at.n_q_heads = sh.q_heads; at.n_kv_heads = sh.kv_heads; at.head_dim = hd; at.causal = 1;
        at.scale = 1.0f / sqrtf((float)hd); at.lse = nullptr;
        VILA_REQUIRE(QS == H, "llm: q_heads*head_dim (%d) must equal hidden (%d) for the in-place attention buffer", QS, H);
        VILA_TRY(launch_attn_fwd(at, s));
        if (l == sh.n_layers - 1 && prune_last) {
            // Generation prefill: after the last layer's K / V are in the cache only the rows whose logits are asked for feed anything.
            // Their o_proj, post-attention norm and MLP run as three decode GEMVs per row (weights stream once at HBM rate) instead of
            // four GEMMs over all T rows: same arithmetic order per row as the decode step, which is parity-tested against this path.
            VILA_TRY(launch_copy_rows(h, lastbuf, last_rows, nullptr, n_last, QS, s));             // attention output rows
            VILA_TRY(launch_copy_rows(x, lastbuf + (size_t)n_last * H, last_rows, nullptr, n_last, H, s));   // residual rows
            for (int r = 0; r < n_last; ++r) {
                bf16_t* ar = lastbuf + (size_t)r * H;                       // attention row  -> later the row's final hidden state
                bf16_t* xr = lastbuf + (size_t)(n_last + r) * H;            // residual row
                bf16_t* fr = act + (size_t)r * F;                           // silu(gate) * up of the row
                GemvArgs g0{};
                g0.x = ar; g0.W = B(L.wo); g0.residual = xr; g0.y = xr; g0.N = H; g0.K = QS; g0.mode = 0;
                VILA_TRY(launch_gemv(g0, s));
                GemvArgs g1{};
                g1.x = xr; g1.norm_w = B(L.ln2_w); g1.eps = sh.rms_eps; g1.W = B(L.w_gate); g1.W2 = B(L.w_up); g1.y = fr; g1.N = F; g1.K = H; g1.mode = 1;
                VILA_TRY(launch_gemv(g1, s));
                GemvArgs g2{};
                g2.x = fr; g2.W = B(L.w_down); g2.residual = xr; g2.y = ar; g2.N = H; g2.K = F; g2.mode = 0;
                VILA_TRY(launch_gemv(g2, s));
            }
            break;
        }
        // x += o_proj(attn).  The post-attention RMSNorm is offered to the GEMM: where its grid is K-sliced (S = 769: 42 tiles x 6 slices) the
        // reduce holds whole rows and writes h = norm(x) as well (h is free: the GEMM kernel that read it has finished when the reduce runs)
        int ln2_done = 0;
        NextNorm n2;
        n2.w = L.ln2_w; n2.eps = sh.rms_eps; n2.rms = 1; n2.out = h; n2.done = &ln2_done;
        VILA_TRY(gemm(h, QS, L.wo, QS, nullptr, x, H, x, H, T, H, QS, EPI_NONE, s, nullptr, 0, skws, skws_bytes, 0, prefill_oproj_norm() ? &n2 : nullptr));
        if (!ln2_done) VILA_TRY(launch_rmsnorm(x, B(L.ln2_w), h, T, H, sh.rms_eps, s));
        // MLP.  A prompt of 3 x 256 + 1 tokens (the benchmark's 769) would spend a whole extra row-tile round on ONE row in the two
        // big GEMMs; those 1-4 leftover rows go through the decode GEMV kernels instead (the weights stream once more at HBM rate:
        // 44 + 25 us per row against 94 + 40 us for the extra tile round)
        VILA_TRY(gemm(h, H, L.w_gate, H, nullptr, nullptr, 0, act, F, Tg, F, H, EPI_GATEUP, s, L.w_up, 0, skws, skws_bytes));  // silu(gate)*up
        for (int r = Tg; r < T; ++r) {
            GemvArgs g{};
            g.x = h + (size_t)r * H; g.W = B(L.w_gate); g.W2 = B(L.w_up); g.y = act + (size_t)r * F; g.N = F; g.K = H; g.mode = 1;
            VILA_TRY(launch_gemv(g, s));
        }
        // x += down(...); where the GEMM is K-sliced (S = 769: 56 tiles cannot fill the chip) its reduce takes the NEXT layer's input_layernorm
        // along (h is free: gate/up has consumed it).  Only when every row goes through the GEMM (no GEMV tail rows).
        NextNorm nn;
        const bool offer = l + 1 < sh.n_layers && Tg == T;
        if (offer) { nn.w = w->layers[l + 1].ln1_w; nn.eps = sh.rms_eps; nn.rms = 1; nn.out = h; nn.done = &ln1_done; }
        VILA_TRY(gemm(act, F, L.w_down, F, nullptr, x, H, x, H, Tg, H, F, EPI_NONE, s, nullptr, 0, skws, skws_bytes, 0, offer ? &nn : nullptr));  // x += down(...)
        for (int r = Tg; r < T; ++r) {
            GemvArgs g{};
            g.x = act + (size_t)r * F; g.W = B(L.w_down); g.residual = x + (size_t)r * H; g.y = x + (size_t)r * H; g.N = H; g.K = F; g.mode = 0;
            VILA_TRY(launch_gemv(g, s));
        }
        if (taps) VILA_HIP(hipMemcpyAsync(taps + (size_t)(l + 1) * T * H, x, (size_t)T * H * 2, hipMemcpyDeviceToDevice, s));
    }

    if (final_hidden != nullptr || all_logits != nullptr) {
        bf16_t* fh = final_hidden ? B(final_hidden) : h;
        VILA_TRY(launch_rmsnorm(x, B(w->norm_w), fh, T, H, sh.rms_eps, s));
        if (all_logits) VILA_TRY(gemm(fh, H, w->lm_head, H, nullptr, nullptr, 0, all_logits, sh.vocab, T, sh.vocab, H, EPI_NONE, s, nullptr, 1));
    }
    if (n_last > 0 && last_logits != nullptr) {
        VILA_REQUIRE(last_rows != nullptr, "llm_prefill: last_rows is NULL");
        if (!prune_last) VILA_TRY(launch_copy_rows(x, lastbuf, last_rows, nullptr, n_last, H, s));     // (pruned: lastbuf already holds the rows)
        if (n_last == 1) {
            GemvArgs g{};
            g.x = lastbuf; g.norm_w = B(w->norm_w); g.eps = sh.rms_eps; g.W = B(w->lm_head); g.y_f32 = last_logits;
            g.N = sh.vocab; g.K = H; g.mode = 0;
            VILA_TRY(launch_gemv(g, s));
        } else {
            VILA_TRY(launch_rmsnorm(lastbuf, B(w->norm_w), lastbuf, n_last, H, sh.rms_eps, s));
            VILA_TRY(gemm(lastbuf, H, w->lm_head, H, nullptr, nullptr, 0, last_logits, sh.vocab, n_last, sh.vocab, H, EPI_NONE, s, nullptr, 1));
        }
    }
    return 0;
}

// =================================================================================================
// LLM decode step (batch 1, greedy)
// =================================================================================================
static inline int dec_splits(int max_ctx) { return cdiv(max_ctx, 64); }
// kernel launches of one vila_llm_decode_step: prologue + per layer {qkv, attention (1 launch up to 2048 cached positions, else
// split-KV + merge), o_proj, gate/up, down} + lm_head + argmax x2 + advance
static int decode_persist_mode();
static bool decode_persist_ok(const VilaLlmShape& sh, int max_ctx) {
    return decode_persist_mode() != 0 && decode_persist_supported(sh.hidden, sh.inter, sh.q_heads, sh.kv_heads, sh.head_dim, sh.n_layers, max_ctx, sh.vocab);
}
extern "C" int vila_llm_decode_launches(const VilaLlmShape* s, int max_ctx) {
    if (decode_persist_ok(*s, max_ctx)) return 6;                 // prologue, the persistent layers kernel, lm_head, argmax x2, advance
    return 1 + s->n_layers * (max_ctx <= 2048 ? 5 : 6) + 4;      // a sampled step: + 1 (three selection launches instead of two argmax stages)
}
extern "C" size_t vila_llm_decode_workspace_bytes(const VilaLlmShape* s, int max_ctx) {
    const size_t H = s->hidden, F = s->inter, QS = (size_t)s->q_heads * s->head_dim;
    const size_t ns = dec_splits(max_ctx);
    size_t b = 0;
    b += 2 * align_up(H * 2, 256) + 2 * align_up(QS * 2, 256) + align_up(F * 2, 256);
    b += align_up(ns * QS * 4, 256) + align_up(ns * s->q_heads * 2 * 4, 256);
    b += 2 * align_up(256 * 4, 256) + align_up((size_t)s->head_dim * 4, 256);
    b += align_up(QS * 2, 256);
    b += align_up(sample_workspace_bytes(), 256);
    b += align_up((64 + (4 * (size_t)s->n_layers + 2) * CHAIN_WORDS) * 4, 256);      // chained step: error word + per-kernel count / flag words (first in the arena)
    return b + 4096;
}

// decode attention variant (vila_decode_force_attn): 2 (default) = per-head blocks over 256-key slices, merged in the prologue of the o_proj GEMV
// (512 blocks); 1 = the same with 256 o_proj blocks; 0 = one block per query head over the whole context + plain o_proj (round 1).
// Measured at context 785..913 (bench.py): 328.7 / 335.7 / 342.0 tok/s for 0 / 1 / 2.
static int g_decode_attn = -1;     // -1: not set yet -> environment VILA_DECODE_ATTN, else 2
extern "C" void vila_decode_force_attn(int mode) { g_decode_attn = mode; }
static int decode_attn_mode() {
    if (g_decode_attn < 0) { const char* e = getenv("VILA_DECODE_ATTN"); g_decode_attn = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }
    return g_decode_attn;
}
// The 28 layers of a token as ONE persistent launch (decode_persist.hip): VILA_DECODE_PERSIST=1 / vila_decode_force_persist(1).  OFF by default:
// bit-identical logits, and measured AT PARITY with the per-kernel step (336.7 vs 341.9 tok/s, profiles/r06_decode_persist_ab.log) — both spend
// ~7 us per all-to-all edge around phases that stream at 7.1-7.3 TB/s (profiles/r06_persist_trace.txt has the per-phase anatomy).
static int g_decode_persist = -1;
extern "C" void vila_decode_force_persist(int on) { g_decode_persist = on ? 1 : 0; }
// measurement hook: device buffer [n_blocks][n_layers * 5 + 1][12] of s_memrealtime stamps written by the next persistent launches (null: off)
extern "C" void vila_decode_persist_trace(void* buf, int n_blocks) { decode_persist_set_trace((unsigned long long*)buf, n_blocks); }
static int decode_persist_mode() {
    if (g_decode_persist < 0) { const char* e = getenv("VILA_DECODE_PERSIST"); g_decode_persist = (e && e[0] == '1') ? 1 : 0; }
    return g_decode_persist;
}
static int decode_step_impl(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st, void* workspace, size_t workspace_bytes,
                            const VilaSampling* sp, vila_stream_t stream);
extern "C" int vila_llm_decode_step(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st,
                                    void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    return decode_step_impl(w, cache, st, workspace, workspace_bytes, nullptr, stream);
}
// the same step with a stochastic pick (temperature / top-k / top-p) instead of argmax: generate(do_sample=True)
extern "C" int vila_llm_decode_step_sample(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st,
                                           void* workspace, size_t workspace_bytes, const VilaSampling* sp, vila_stream_t stream) {
    VILA_REQUIRE(sp != nullptr, "llm_decode_sample: sampling parameters are NULL");
    return decode_step_impl(w, cache, st, workspace, workspace_bytes, sp, stream);
}
extern "C" size_t vila_sample_workspace_bytes(void) { return sample_workspace_bytes(); }
extern "C" int vila_sample_f32(const float* logits, int n, const VilaSampling* sp, const int32_t* counter, int64_t* out, void* workspace,
                               float* dist_out, vila_stream_t stream) {
    VILA_REQUIRE(sp != nullptr && logits != nullptr && out != nullptr && workspace != nullptr, "sample: NULL argument");
    return launch_sample(logits, n, sp->temperature, sp->top_k, sp->top_p, sp->seed, sp->seed_dev, counter, out, workspace, dist_out, S(stream));
}
// ---- chained decode step (round 4) ------------------------------------------------------------------------------------------------------
// Three of a layer's five kernel boundaries are CHAINED: the successor is launched while its predecessor still runs, requests the weights it can
// (they do not depend on activations), and waits on the predecessor's device-side done counter before it touches an activation
// (gemv_common.h chain_wait / chain_done).  What a plain boundary costs — the predecessor's tail with the HBM pipe running dry, the launch gap,
// the successor's first memory round trip — is spent streaming the successor's weights instead.
//   layer l:   A qkv  ->  B attention  ->  C o_proj  =>  D gate/up  =>  E down  =>  A' qkv of layer l+1          ( => chained, -> plain )
// Two streams X / Y swap roles every layer:  X: A B C . E        Y: (event: B done) D . A' B' C' . E'      X: (event: B' done) D' ...
// so a chained kernel is launched when the kernel TWO before it has finished (stream order) and at most two kernels are in flight.
// Why it cannot deadlock (a waiting kernel must never keep the kernel it waits for off the chip): every chained kernel and the kernel it
// waits for are launched with <= 2 blocks per CU and <= 136 VGPRs — about half a CU each — so both are entirely resident whatever the dispatch order;
// the attention kernel (16-wave blocks that need a whole CU) is never beside a waiting kernel: D is held back by an event until B has finished.
// The waits are bounded all the same: a give-up is reported in workspace word 0 (vila_llm_decode_chain_error) instead of hanging the device.
// MEASURED (profiles/r04_decode_chain_ab.log, five sessions): 3.28 ms per token chained against 2.99 plain — the device-side hand-off costs what a
// kernel boundary costs on this chip, so the prefetched weights buy nothing.  The chain is therefore OFF by default; VILA_DECODE_CHAIN=1 /
// vila_decode_force_chain(1) selects it (parity-tested: tests/test_gpu_model.py::test_chained_decode_step_equals_the_plain_step).
static int g_decode_chain = -1;
extern "C" void vila_decode_force_chain(int on) { g_decode_chain = on ? 1 : 0; }
static int decode_chain_mode() {
    if (g_decode_chain < 0) { const char* e = getenv("VILA_DECODE_CHAIN"); g_decode_chain = (e && e[0] == '1') ? 1 : 0; }
    return g_decode_chain;
}
// VILA_DECODE_CHAIN_PRED (percent, default 95): how much of the predecessor's predicted run time (its bytes at 6.2 TB/s) a waiting kernel
// treats as the point around which its polls concentrate (gemv_common.h chain_wait)
static double chain_pred_scale() {
    static double v = -1.0;
    if (v < 0) { const char* e = getenv("VILA_DECODE_CHAIN_PRED"); v = (e && atoi(e) > 0) ? atoi(e) / 100.0 : 0.95; }
    return v;
}
struct ChainStreams { hipStream_t s2 = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static ChainStreams& chain_streams() {
    static thread_local ChainStreams per_dev[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    ChainStreams& c = per_dev[dev & 15];
    if (c.s2 == nullptr) {
        if (hipStreamCreateWithFlags(&c.s2, hipStreamNonBlocking) != hipSuccess) c.s2 = nullptr;
        (void)hipEventCreateWithFlags(&c.fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&c.join, hipEventDisableTiming);
    }
    return c;
}
// word 0 of the decode workspace: 1 if a chained kernel's bounded wait gave up since the last call of this function (then results are invalid)
extern "C" int vila_llm_decode_chain_error(void* workspace, vila_stream_t stream) {
    hipStream_t s = S(stream);
    uint32_t h = 0;
    VILA_HIP(hipMemcpyAsync(&h, workspace, 4, hipMemcpyDeviceToHost, s));
    VILA_HIP(hipStreamSynchronize(s));
    if (h != 0) VILA_HIP(hipMemsetAsync(workspace, 0, 4, s));
    return (int)h;
}

static int decode_step_impl(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeState* st, void* workspace, size_t workspace_bytes,
                            const VilaSampling* sp, vila_stream_t stream) {
    const VilaLlmShape& sh = w->shape;
    hipStream_t s = S(stream);
    VILA_REQUIRE(cache != nullptr && st != nullptr, "llm_decode: cache/state is NULL");
    VILA_REQUIRE(workspace_bytes >= vila_llm_decode_workspace_bytes(&sh, cache->max_ctx), "llm_decode: workspace too small");
    const int H = sh.hidden, F = sh.inter, hd = sh.head_dim, QS = sh.q_heads * hd, KS = sh.kv_heads * hd;
    const int ns = dec_splits(cache->max_ctx);
    Arena a(workspace, workspace_bytes);
    const int n_chain = 4 * sh.n_layers + 2;                    // chained kernels of a token: A, C, D, E per layer + lm_head (+ 1 spare)
    uint32_t* chain_mem = a.take<uint32_t>(64 + (size_t)n_chain * CHAIN_WORDS);    // [0] error word, [64..] per kernel: arrival count + CHAIN_FLAGS go-flags
    bf16_t* x = a.take<bf16_t>(H);
    bf16_t* x2 = a.take<bf16_t>(H);
    bf16_t* q = a.take<bf16_t>(QS);
    bf16_t* act = a.take<bf16_t>(F);
    float* part_o = a.take<float>((size_t)ns * QS);
    float* part_ml = a.take<float>((size_t)ns * sh.q_heads * 2);
    float* tv = a.take<float>(256);
    int* ti = a.take<int>(256);
    float* rope_cs = a.take<float>(hd);
    bf16_t* ao = a.take<bf16_t>(QS);
    void* smp_ws = a.take<char>(sample_workspace_bytes());
    VILA_REQUIRE(a.ok(), "llm_decode: workspace arena overflow");

    if (decode_persist_ok(sh, cache->max_ctx) && !decode_chain_mode() && cache->n_slots >= 1) {
        // ---- the persistent token: prologue (embedding row, RoPE row, barrier words) + ONE launch for 28 layers and the head ----
        DpArgs d{};
        const size_t per_layer = (size_t)cache->n_slots * sh.kv_heads * cache->max_ctx * hd;
        for (int l = 0; l < sh.n_layers; ++l) {
            const VilaLlmLayer& L = w->layers[l];
            const bool fused = (B(L.wk) == B(L.wq) + (size_t)QS * H) && (B(L.wv) == B(L.wk) + (size_t)KS * H) &&
                               (B(L.bk) == B(L.bq) + QS) && (B(L.bv) == B(L.bk) + KS);
            VILA_REQUIRE(fused, "llm_decode: q/k/v projection weights and biases must be views of one fused [q+2kv, hidden] buffer");
            d.layer[l] = DpLayer{B(L.ln1_w), B(L.wq), B(L.bq), B(L.wo), B(L.ln2_w), B(L.w_gate), B(L.w_up), B(L.w_down)};
        }
        d.norm_w = B(w->norm_w); d.lm_head = B(w->lm_head); d.logits = st->logits;
        d.kcache = B(cache->k); d.vcache = B(cache->v); d.kv_layer_stride = (int64_t)per_layer;
        d.pos_ptr = st->pos; d.rope_cs = rope_cs; d.x0 = x; d.x1 = x2; d.q = q; d.act = act; d.part_o = part_o; d.part_ml = part_ml;
        d.sync = chain_mem;
        d.n_layers = sh.n_layers; d.H = H; d.F = F; d.nq = sh.q_heads; d.nkv = sh.kv_heads; d.hd = hd; d.vocab = sh.vocab; d.max_ctx = cache->max_ctx;
        d.eps = sh.rms_eps; d.scale = 1.0f / sqrtf((float)hd);
        VILA_TRY(launch_decode_prologue(B(w->embed), st->token, x, H, sh.vocab, st->pos, rope_cs, hd, sh.rope_theta, s, chain_mem + 64, 1));
        VILA_TRY(launch_decode_persist(d, s));
        // the head as its own launch: final RMSNorm + lm_head rows -> fp32 logits (152 064 x 3584: 1.09 GB, 7 TB/s in gemv_kernel<0,7>).  The
        // last layer's residual stream is in `x` (an even number of buffer swaps per layer).
        GemvArgs lm{};
        lm.x = x; lm.norm_w = B(w->norm_w); lm.eps = sh.rms_eps; lm.W = B(w->lm_head); lm.y_f32 = st->logits; lm.N = sh.vocab; lm.K = H; lm.mode = 0;
        VILA_TRY(launch_gemv(lm, s));
        if (sp != nullptr) VILA_TRY(launch_sample(st->logits, sh.vocab, sp->temperature, sp->top_k, sp->top_p, sp->seed, sp->seed_dev, st->pos, st->token, smp_ws, nullptr, s));
        else VILA_TRY(launch_argmax(st->logits, sh.vocab, st->token, tv, ti, s));
        VILA_TRY(launch_decode_advance(st->pos, st->token, st->out_ids, st->n_out, st->max_out, s));
        return 0;
    }
    const bool split256 = decode_attn_mode() >= 1 && cache->max_ctx <= 2048 && hd == 128;
    ChainStreams* cs = nullptr;
    if (decode_chain_mode() && split256) {                      // (the long-context split-KV + merge pair stays unchained)
        cs = &chain_streams();
        if (cs->s2 == nullptr || cs->fork == nullptr || cs->join == nullptr) cs = nullptr;
    }
    const bool chained = cs != nullptr;
    uint32_t* ctr = chained ? chain_mem + 64 : nullptr;
    int k_idx = 0;                                              // index of the next kernel's done counter
    int prev_grid = 0;                                          // grid of the kernel launched last (what a chained successor waits for)
    // link of the next kernel: counts itself under its own index; `wait` = it is launched early and waits for the previous kernel's whole grid
    // `pred_bytes`: what the kernel waited for streams — its predicted run time (at 6 TB/s) is slept through before the first poll
    auto link = [&](ChainLink& c, bool wait, size_t pred_bytes) {
        if (!chained) return;
        c.ctr = ctr; c.err = chain_mem; c.done_idx = k_idx;
        c.wait_idx = (wait && k_idx > 0) ? k_idx - 1 : -1; c.wait_target = (uint32_t)prev_grid;
        const int us = (int)((double)pred_bytes / 6.2e6 * chain_pred_scale());
        c.pre_sleep_us = (wait && us > 0) ? us : 0;
        ++k_idx;
    };
    const size_t bytes_o = (size_t)H * QS * 2, bytes_gu = (size_t)2 * F * H * 2, bytes_dn = (size_t)H * F * 2;
    const int bpc = chained ? 2 : 0;                            // half a CU per chained kernel (0 = the launcher's default of 4)

    VILA_TRY(launch_decode_prologue(B(w->embed), st->token, x, H, sh.vocab, st->pos, rope_cs, hd, sh.rope_theta, s, ctr, chained ? n_chain : 0));
    hipStream_t X = s, Y = chained ? cs->s2 : s;                // X runs A B C E of this layer, Y its D
    if (chained) {
        VILA_HIP(hipEventRecord(cs->fork, s));
        VILA_HIP(hipStreamWaitEvent(cs->s2, cs->fork, 0));
    }
    bf16_t* cur = x; bf16_t* nxt = x2;
    for (int l = 0; l < sh.n_layers; ++l) {
        const VilaLlmLayer& L = w->layers[l];
        const size_t per_layer = (size_t)cache->n_slots * sh.kv_heads * cache->max_ctx * hd;
        bf16_t* kc = B(cache->k) + l * per_layer; bf16_t* vc = B(cache->v) + l * per_layer;
        const bool fused = (B(L.wk) == B(L.wq) + (size_t)QS * H) && (B(L.wv) == B(L.wk) + (size_t)KS * H) &&
                           (B(L.bk) == B(L.bq) + QS) && (B(L.bv) == B(L.bk) + KS);
        VILA_REQUIRE(fused, "llm_decode: q/k/v projection weights and biases must be views of one fused [q+2kv, hidden] buffer");
        QkvDecodeArgs qa{};
        qa.x = cur; qa.norm_w = B(L.ln1_w); qa.eps = sh.rms_eps; qa.Wqkv = B(L.wq); qa.bqkv = B(L.bq); qa.q_out = q;
        qa.kcache = kc; qa.vcache = vc; qa.pos_ptr = st->pos; qa.K = H; qa.nq = sh.q_heads; qa.nkv = sh.kv_heads; qa.hd = hd;
        qa.max_ctx = cache->max_ctx; qa.rope_cs = rope_cs; qa.max_bpc = bpc;
        link(qa.chain, l > 0, bytes_dn);                                  // A: chained behind the previous layer's E (layer 0: behind the prologue, stream order)
        VILA_TRY(launch_qkv_decode(qa, X, &prev_grid));
        AttnDecodeArgs ad{};
        ad.q = q; ad.kcache = kc; ad.vcache = vc; ad.o = ao; ad.part_o = part_o; ad.part_ml = part_ml; ad.pos_ptr = st->pos;
        ad.nq = sh.q_heads; ad.nkv = sh.kv_heads; ad.hd = hd; ad.max_ctx = cache->max_ctx; ad.n_splits = ns; ad.scale = 1.0f / sqrtf((float)hd);
        ad.split256 = split256 ? 1 : 0;
        VILA_TRY(launch_attn_decode(ad, X, nullptr));           // B: plain, behind A on the same stream
        if (chained) {
            VILA_HIP(hipEventRecord(cs->join, X));              // (event reuse: every record is consumed by the wait right below)
            VILA_HIP(hipStreamWaitEvent(Y, cs->join, 0));       // D may not sit on the CUs while B needs whole ones
        }
        GemvArgs o{};
        o.x = ao; o.W = B(L.wo); o.residual = cur; o.y = nxt; o.N = H; o.K = QS; o.mode = 0;
        if (split256) { o.mode = 2; o.part_o = part_o; o.part_ml = part_ml; o.pos_ptr = st->pos; o.n_splits = cdiv(cache->max_ctx, 256); o.split_keys = 256; o.grid_cap = decode_attn_mode() == 2 ? 512 : 256; }
        link(o.chain, false, 0);                                   // C: plain behind B, but it counts itself done for D
        VILA_TRY(launch_gemv(o, X, &prev_grid));
        GemvArgs gu{};
        gu.x = nxt; gu.norm_w = B(L.ln2_w); gu.eps = sh.rms_eps; gu.W = B(L.w_gate); gu.W2 = B(L.w_up); gu.y = act; gu.N = F; gu.K = H; gu.mode = 1;
        gu.max_bpc = bpc;
        link(gu.chain, true, bytes_o);                                   // D: launched when B is done, waits for C
        VILA_TRY(launch_gemv(gu, Y, &prev_grid));
        GemvArgs dn{};
        dn.x = act; dn.W = B(L.w_down); dn.residual = nxt; dn.y = cur; dn.N = H; dn.K = F; dn.mode = 0; dn.max_bpc = bpc;
        link(dn.chain, true, bytes_gu);                                   // E: launched when C is done (stream order on X), waits for D
        VILA_TRY(launch_gemv(dn, X, &prev_grid));
        if (chained) { hipStream_t t = X; X = Y; Y = t; }       // the next layer's A goes behind D (done before E can be) and waits for E
    }
    GemvArgs lm{};
    lm.x = cur; lm.norm_w = B(w->norm_w); lm.eps = sh.rms_eps; lm.W = B(w->lm_head); lm.y_f32 = st->logits; lm.N = sh.vocab; lm.K = H; lm.mode = 0;
    lm.max_bpc = bpc;
    link(lm.chain, true, bytes_dn);                                       // lm_head: like an A — behind the last D, waits for the last E
    VILA_TRY(launch_gemv(lm, X, &prev_grid));
    if (chained) {                                              // the token choice runs on the caller's stream behind BOTH streams
        VILA_HIP(hipEventRecord(cs->fork, cs->s2));
        VILA_HIP(hipStreamWaitEvent(s, cs->fork, 0));
    }
    if (sp != nullptr) VILA_TRY(launch_sample(st->logits, sh.vocab, sp->temperature, sp->top_k, sp->top_p, sp->seed, sp->seed_dev, st->pos, st->token, smp_ws, nullptr, s));
    else VILA_TRY(launch_argmax(st->logits, sh.vocab, st->token, tv, ti, s));
    VILA_TRY(launch_decode_advance(st->pos, st->token, st->out_ids, st->n_out, st->max_out, s));
    return 0;
}

// =================================================================================================
// Batched decode step (decode_batch.hip)
// =================================================================================================
extern "C" size_t vila_llm_decode_batch_workspace_bytes(const VilaLlmShape* s, int n) {
    return bdecode_workspace_bytes(s->hidden, s->inter, s->q_heads * s->head_dim, s->head_dim, n);
}
extern "C" int vila_llm_decode_step_batch(const VilaLlmWeights* w, const VilaKvCache* cache, const VilaDecodeBatch* st,
                                          void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    VILA_REQUIRE(w != nullptr && cache != nullptr && st != nullptr && workspace != nullptr, "llm_decode_batch: NULL argument");
    const VilaLlmShape& sh = w->shape;
    hipStream_t s = S(stream);
    const int hd = sh.head_dim, QS = sh.q_heads * hd, KS = sh.kv_heads * hd, H = sh.hidden;
    std::vector<BLayer> layers(sh.n_layers);
    for (int l = 0; l < sh.n_layers; ++l) {
        const VilaLlmLayer& L = w->layers[l];
        const bool fused = (B(L.wk) == B(L.wq) + (size_t)QS * H) && (B(L.wv) == B(L.wk) + (size_t)KS * H) &&
                           (B(L.bk) == B(L.bq) + QS) && (B(L.bv) == B(L.bk) + KS);
        VILA_REQUIRE(fused, "llm_decode_batch: q/k/v projection weights and biases must be views of one fused [q+2kv, hidden] buffer");
        layers[l] = BLayer{L.ln1_w, L.wq, L.bq, L.wo, L.ln2_w, L.w_gate, L.w_up, L.w_down};
    }
    BDecodeArgs m{w->embed, w->norm_w, w->lm_head, sh.hidden, sh.inter, sh.n_layers, sh.q_heads, sh.kv_heads, sh.head_dim, sh.vocab, sh.rms_eps, sh.rope_theta};
    return bdecode_step(m, layers.data(), B(cache->k), B(cache->v), cache->max_ctx, cache->n_slots, st->n, st->pos, st->token, st->out_ids, st->n_out,
                        st->max_out, st->logits, workspace, workspace_bytes, s);
}

// =================================================================================================
// hipGraph helpers
// =================================================================================================
extern "C" int vila_graph_begin(vila_stream_t stream) {
    VILA_HIP(hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int vila_graph_end(vila_stream_t stream, void** graph_exec_out) {
    hipGraph_t g = nullptr;
    VILA_HIP(hipStreamEndCapture(S(stream), &g));
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (err != hipSuccess) VILA_FAIL(-2, "hipGraphInstantiate failed: %s", hipGetErrorString(err));
    *graph_exec_out = (void*)e;
    return 0;
}
extern "C" int vila_graph_launch(void* graph_exec, vila_stream_t stream) {
    VILA_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, S(stream)));
    return 0;
}
extern "C" int vila_graph_destroy(void* graph_exec) {
    if (graph_exec) VILA_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

// =================================================================================================
// Operator-level exports
// =================================================================================================
extern "C" int vila_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* W2, const void* bias,
                              const void* residual, int64_t ldr, void* C, int64_t ldc, int out_f32, int M, int N, int K, int epi,
                              vila_stream_t stream) {
    return gemm(B(A), lda, W, ldw, bias, B(residual), ldr, C, ldc, M, N, K, epi, S(stream), W2, out_f32);
}
extern "C" int vila_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const void* W2, const void* bias,
                                 const void* residual, int64_t ldr, void* C, int64_t ldc, int out_f32, int M, int N, int K, int epi,
                                 void* ws, size_t ws_bytes, vila_stream_t stream) {
    return gemm(B(A), lda, W, ldw, bias, B(residual), ldr, C, ldc, M, N, K, epi, S(stream), W2, out_f32, (float*)ws, ws_bytes);
}
// C[M,N] = A . B^T (+bias)(+residual) with either operand stored contraction-major: dgrad (b_cm: B = W[K][N] as it lies) and wgrad
// (a_cm, b_cm: A = dY[K=tokens][M], B = X[K=tokens][N]) read the forward tensors in place, no transposed copies
extern "C" int vila_gemm_bf16_t(const void* A, int64_t lda, int a_cm, const void* W, int64_t ldw, int b_cm, const void* bias,
                                const void* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, void* ws, size_t ws_bytes,
                                vila_stream_t stream) {
    GemmArgs g;
    g.A = B(A); g.lda = lda; g.a_cm = a_cm ? 1 : 0; g.W = B(W); g.ldw = ldw; g.b_cm = b_cm ? 1 : 0; g.bias = B(bias);
    g.residual = B(residual); g.ldr = ldr; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.epi = EPI_NONE;
    g.ws = (float*)ws; g.ws_bytes = ws_bytes;
    return launch_gemm(g, S(stream));
}
extern "C" int vila_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps, vila_stream_t stream) {
    return launch_layernorm(B(x), B(w), B(b), B(y), rows, cols, eps, S(stream));
}
extern "C" int vila_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int cols, float eps, vila_stream_t stream) {
    return launch_rmsnorm(B(x), B(w), B(y), rows, cols, eps, S(stream));
}
extern "C" int vila_space_to_depth_bf16(const void* x, void* y, int n_images, int grid, int channels, int k, vila_stream_t stream) {
    return launch_space_to_depth(B(x), B(y), n_images, grid, channels, k, S(stream));
}
extern "C" int vila_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t q_tok_stride, int64_t k_tok_stride,
                                  int64_t v_tok_stride, int64_t o_tok_stride, int q_head_stride, int k_head_stride, int v_head_stride,
                                  int o_head_stride, const int32_t* cu_seqlens, int n_seq, int total_tokens, int max_seqlen,
                                  int n_q_heads, int n_kv_heads, int head_dim, int causal, float scale, float* lse, vila_stream_t stream) {
    AttnArgs at{};
    at.q = B(q); at.k = B(k); at.v = B(v); at.o = B(o);
    at.q_tok_stride = q_tok_stride; at.k_tok_stride = k_tok_stride; at.v_tok_stride = v_tok_stride; at.o_tok_stride = o_tok_stride;
    at.q_head_stride = q_head_stride; at.k_head_stride = k_head_stride; at.v_head_stride = v_head_stride; at.o_head_stride = o_head_stride;
    at.cu_seqlens = cu_seqlens; at.n_seq = n_seq; at.total_tokens = total_tokens; at.max_seqlen = max_seqlen;
    at.n_q_heads = n_q_heads; at.n_kv_heads = n_kv_heads; at.head_dim = head_dim; at.causal = causal; at.scale = scale; at.lse = lse;
    return launch_attn_fwd(at, S(stream));
}
extern "C" int vila_gemv_bf16(const void* x, const void* norm_w, float eps, const void* W, const void* W2, const void* bias,
                              const void* residual, void* y_bf16, float* y_f32, int N, int K, int mode, vila_stream_t stream) {
    GemvArgs g{};
    g.x = B(x); g.norm_w = B(norm_w); g.eps = eps; g.W = B(W); g.W2 = B(W2); g.bias = B(bias); g.residual = B(residual);
    g.y = B(y_bf16); g.y_f32 = y_f32; g.N = N; g.K = K; g.mode = mode;
    return launch_gemv(g, S(stream));
}
extern "C" int vila_argmax_f32(const float* logits, int n, int64_t* out, void* workspace, vila_stream_t stream) {
    float* tv = (float*)workspace;
    int* ti = (int*)((char*)workspace + 2048);
    return launch_argmax(logits, n, out, tv, ti, S(stream));
}

// =================================================================================================
// Training operator exports (backward kernels + optimizer); orchestrated by vila_amd/train.py
// =================================================================================================
extern "C" int vila_transpose_bf16(const void* in, void* out, int R, int C, int64_t ldi, int64_t ldo, vila_stream_t stream) {
    return launch_transpose(B(in), B(out), R, C, ldi, ldo, S(stream));
}
extern "C" int vila_act_fwd_bf16(const void* z, void* y, int64_t n, int act, vila_stream_t stream) { return launch_act_fwd(B(z), B(y), n, act, S(stream)); }
extern "C" int vila_act_bwd_bf16(const void* z, const void* dy, void* dz, int64_t n, int act, vila_stream_t stream) {
    return launch_act_bwd(B(z), B(dy), B(dz), n, act, S(stream));
}
extern "C" int vila_silu_mul_fwd_bf16(const void* g, const void* u, void* a, int64_t n, vila_stream_t stream) {
    return launch_silu_mul_fwd(B(g), B(u), B(a), n, S(stream));
}
extern "C" int vila_silu_mul_bwd_bf16(const void* g, const void* u, const void* da, void* dg, void* du, int64_t n, vila_stream_t stream) {
    return launch_silu_mul_bwd(B(g), B(u), B(da), B(dg), B(du), n, S(stream));
}
extern "C" int vila_add_bf16(const void* a, const void* b, void* y, int64_t n, vila_stream_t stream) { return launch_add(B(a), B(b), B(y), n, S(stream)); }
extern "C" int vila_grad_accum_f32(float* acc, const void* g, void* out, int64_t n, int mode, vila_stream_t stream) {
    return launch_grad_accum(acc, B(g), B(out), n, mode, S(stream));
}
extern "C" int vila_colsum_bf16(const void* x, void* out, float* scratch, int R, int C, int64_t ld, int accumulate, int period, vila_stream_t stream) {
    return launch_colsum(B(x), B(out), scratch, R, C, ld, accumulate, period, S(stream));
}
extern "C" int vila_norm_bwd_bf16(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, float* scratch, int rows, int cols,
                                  float eps, int rms, int accumulate, vila_stream_t stream) {
    return launch_norm_bwd(B(x), B(w), B(dy), B(dx), B(dw), B(db), scratch, rows, cols, eps, rms, accumulate, S(stream));
}
extern "C" int vila_ce_loss_f32(const float* logits, const int64_t* labels, void* dlogits, float* loss, float* row_loss, int rows, int V, int64_t ldl, float scale,
                                vila_stream_t stream) {
    return launch_ce(logits, labels, B(dlogits), loss, row_loss, rows, V, ldl, scale, S(stream));
}
extern "C" int vila_scatter_add_rows_bf16(const void* src, void* dst, const int32_t* rows, int n, int H, vila_stream_t stream) {
    return launch_scatter_add_rows(B(src), B(dst), rows, n, H, S(stream));
}
extern "C" int vila_depth_to_space_bf16(const void* dy, void* dx, int n_images, int grid, int channels, int k, vila_stream_t stream) {
    return launch_depth_to_space(B(dy), B(dx), n_images, grid, channels, k, S(stream));
}
extern "C" int vila_im2col_bf16(const void* pixels, void* out, int n_images, int channels, int H, int W, int P, int Kp, vila_stream_t stream) {
    return launch_im2col(B(pixels), B(out), n_images, channels, H, W, P, Kp, S(stream));
}
extern "C" int vila_rope_table_f32(const int32_t* positions, float* cos_out, float* sin_out, int S_, int head_dim, float theta, vila_stream_t stream) {
    return launch_rope_table(positions, cos_out, sin_out, S_, head_dim, theta, S(stream));
}
extern "C" int vila_rope_fwd_bf16(void* qkv, const float* cs, const float* sn, const int32_t* positions, int S_, int nq, int nkv, int hd,
                                  vila_stream_t stream) {
    return launch_rope_kv(B(qkv), cs, sn, positions, nullptr, nullptr, nullptr, S_, nq, nkv, hd, 0, S(stream));
}
extern "C" int vila_rope_bwd_bf16(void* dqkv, const float* cs, const float* sn, int S_, int nq, int nkv, int hd, vila_stream_t stream) {
    return launch_rope_bwd(B(dqkv), cs, sn, S_, nq, nkv, hd, S(stream));
}
extern "C" int vila_attn_bwd_bf16_parts(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                                  const int64_t* tok_strides /*[8] q,k,v,o,do,dq,dk,dv*/, const int32_t* head_strides /*[8]*/,
                                  const int32_t* cu_seqlens, int n_seq, int total_tokens, int max_seqlen, int n_q_heads, int n_kv_heads,
                                  int head_dim, int causal, float scale, const float* lse, float* delta, int parts, vila_stream_t stream) {
    AttnBwdArgs a{};
    a.q = B(q); a.k = B(k); a.v = B(v); a.o = B(o); a.d_o = B(d_o); a.dq = B(dq); a.dk = B(dk); a.dv = B(dv);
    a.q_tok_stride = tok_strides[0]; a.k_tok_stride = tok_strides[1]; a.v_tok_stride = tok_strides[2]; a.o_tok_stride = tok_strides[3];
    a.do_tok_stride = tok_strides[4]; a.dq_tok_stride = tok_strides[5]; a.dk_tok_stride = tok_strides[6]; a.dv_tok_stride = tok_strides[7];
    a.q_head_stride = head_strides[0]; a.k_head_stride = head_strides[1]; a.v_head_stride = head_strides[2]; a.o_head_stride = head_strides[3];
    a.do_head_stride = head_strides[4]; a.dq_head_stride = head_strides[5]; a.dk_head_stride = head_strides[6]; a.dv_head_stride = head_strides[7];
    for (int i = 0; i < 8; ++i) VILA_REQUIRE(tok_strides[i] % 8 == 0 && head_strides[i] % 8 == 0, "attn_bwd: strides must be multiples of 8 elements");
    a.cu_seqlens = cu_seqlens; a.n_seq = n_seq; a.total_tokens = total_tokens; a.max_seqlen = max_seqlen;
    a.n_q_heads = n_q_heads; a.n_kv_heads = n_kv_heads; a.head_dim = head_dim; a.causal = causal; a.scale = scale; a.lse = lse; a.delta = delta;
    return launch_attn_bwd(a, S(stream), parts);
}
extern "C" int vila_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o, void* dq, void* dk, void* dv,
                                  const int64_t* tok_strides, const int32_t* head_strides, const int32_t* cu_seqlens, int n_seq, int total_tokens,
                                  int max_seqlen, int n_q_heads, int n_kv_heads, int head_dim, int causal, float scale, const float* lse, float* delta,
                                  vila_stream_t stream) {
    return vila_attn_bwd_bf16_parts(q, k, v, o, d_o, dq, dk, dv, tok_strides, head_strides, cu_seqlens, n_seq, total_tokens, max_seqlen, n_q_heads,
                                    n_kv_heads, head_dim, causal, scale, lse, delta, 7, stream);
}
extern "C" int vila_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, vila_stream_t stream) {
    return launch_adamw(master, m, v, B(grad), B(param), n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, S(stream));
}
// same update, <= 32 VGPRs per lane: meant for a side stream, co-resident with the matrix kernels (see adamw_lean_kernel)
extern "C" int vila_adamw_step_lean(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, int step, float grad_scale, vila_stream_t stream) {
    return launch_adamw_lean(master, m, v, B(grad), B(param), n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, S(stream));
}
extern "C" int vila_sumsq_bf16(const void* x, int64_t n, float* out, float* scratch, vila_stream_t stream) { return launch_sumsq(B(x), n, out, scratch, S(stream)); }
extern "C" size_t vila_colsum_scratch_floats(int rows, int cols) { return colsum_scratch_floats(rows, cols); }
extern "C" size_t vila_norm_bwd_scratch_floats(int rows, int cols) { return norm_bwd_scratch_floats(rows, cols); }

// dynamic_s2: merge_chessboard + area interpolation + concat + split_chessboard in one gather
extern "C" int vila_s2_merge_bf16(const void* feats, void* out, const int32_t* desc, int n_blocks, int grid, int channels, int n_scales,
                                  const int32_t* splits /*[host] n_scales-1*/, vila_stream_t stream) {
    int sp[4] = {1, 1, 1, 1};
    for (int k = 0; k < n_scales - 1 && k < 4; ++k) sp[k] = splits[k];
    return launch_s2_merge(B(feats), B(out), desc, n_blocks, grid, channels, n_scales, sp, S(stream));
}

extern "C" int vila_s2_merge_bwd_bf16(const void* dy, void* dx, const int32_t* tile_desc, int n_tiles, int grid, int channels, int n_scales,
                                      const int32_t* splits /*[host] n_scales-1*/, vila_stream_t stream) {
    int sp[4] = {1, 1, 1, 1};
    for (int k = 0; k < n_scales - 1 && k < 4; ++k) sp[k] = splits[k];
    return launch_s2_merge_bwd(B(dy), B(dx), tile_desc, n_tiles, grid, channels, n_scales, sp, S(stream));
}

// video encoders: BasicVideoEncoder (pool 1,1,1) / TSPVideoEncoder token assembly in one launch
extern "C" int vila_video_pool_bf16(const void* feats, void* out, int n_frames, int grid, int channels, int pool_t, int pool_h, int pool_w,
                                    const void* start_rows, int n_start, const void* end_rows, int n_end, vila_stream_t stream) {
    return launch_video_pool(B(feats), B(out), n_frames, grid, channels, pool_t, pool_h, pool_w, B(start_rows), n_start, B(end_rows), n_end, S(stream));
}
extern "C" int vila_video_pool_bwd_bf16(const void* dpooled, void* dfeats, int n_frames, int grid, int channels, int pool_t, int pool_h, int pool_w,
                                        int accumulate, vila_stream_t stream) {
    return launch_video_pool_bwd(B(dpooled), B(dfeats), n_frames, grid, channels, pool_t, pool_h, pool_w, accumulate, S(stream));
}

// =================================================================================================
// W4A16 decode: int4 group-128 weights for the five decoder-layer projections, bf16 everything else
// =================================================================================================
extern "C" int vila_gemv_w4_bf16(const void* x, const void* norm_w, float eps, const void* Wq, const void* Wsz,
                                 const void* bias, const void* residual, void* y, int N, int K, int mode, vila_stream_t stream) {
    GemvW4Args g{};
    g.x = B(x); g.norm_w = B(norm_w); g.eps = eps; g.Wq = (const uint32_t*)Wq; g.Wsz = (const uint32_t*)Wsz;
    g.bias = B(bias); g.residual = B(residual); g.y = B(y); g.N = N; g.K = K; g.mode = mode;
    VILA_REQUIRE(mode == 0 || mode == 1, "vila_gemv_w4_bf16: mode must be 0 or 1");
    VILA_REQUIRE(x != nullptr && Wq != nullptr && Wsz != nullptr && y != nullptr, "vila_gemv_w4_bf16: NULL pointer");
    return launch_gemv_w4(g, S(stream));
}

static int decode_step_w4_impl(const VilaLlmWeights* w, const VilaLlmLayerW4* ql, const VilaKvCache* cache, const VilaDecodeState* st,
                               void* workspace, size_t workspace_bytes, const VilaSampling* sp, vila_stream_t stream);
extern "C" int vila_llm_decode_step_w4(const VilaLlmWeights* w, const VilaLlmLayerW4* ql, const VilaKvCache* cache, const VilaDecodeState* st,
                                       void* workspace, size_t workspace_bytes, vila_stream_t stream) {
    return decode_step_w4_impl(w, ql, cache, st, workspace, workspace_bytes, nullptr, stream);
}
// the W4A16 step with a stochastic pick (generate(do_sample=True) on a quantised decoder)
extern "C" int vila_llm_decode_step_w4_sample(const VilaLlmWeights* w, const VilaLlmLayerW4* ql, const VilaKvCache* cache, const VilaDecodeState* st,
                                              void* workspace, size_t workspace_bytes, const VilaSampling* sp, vila_stream_t stream) {
    VILA_REQUIRE(sp != nullptr, "llm_decode_w4_sample: sampling parameters are NULL");
    return decode_step_w4_impl(w, ql, cache, st, workspace, workspace_bytes, sp, stream);
}
static int decode_step_w4_impl(const VilaLlmWeights* w, const VilaLlmLayerW4* ql, const VilaKvCache* cache, const VilaDecodeState* st,
                               void* workspace, size_t workspace_bytes, const VilaSampling* sp, vila_stream_t stream) {
    const VilaLlmShape& sh = w->shape;
    hipStream_t s = S(stream);
    VILA_REQUIRE(cache != nullptr && st != nullptr && ql != nullptr, "llm_decode_w4: cache/state/weights is NULL");
    VILA_REQUIRE(workspace_bytes >= vila_llm_decode_workspace_bytes(&sh, cache->max_ctx), "llm_decode_w4: workspace too small");
    const int H = sh.hidden, F = sh.inter, hd = sh.head_dim, QS = sh.q_heads * hd;
    const int ns = dec_splits(cache->max_ctx);
    Arena a(workspace, workspace_bytes);
    (void)a.take<uint32_t>(64 + (4 * (size_t)sh.n_layers + 2) * CHAIN_WORDS);      // same layout as the bf16 step: word 0 = the chain error flag (unused here, stays 0)
    bf16_t* x = a.take<bf16_t>(H);
    bf16_t* x2 = a.take<bf16_t>(H);
    bf16_t* q = a.take<bf16_t>(QS);
    bf16_t* act = a.take<bf16_t>(F);
    float* part_o = a.take<float>((size_t)ns * QS);
    float* part_ml = a.take<float>((size_t)ns * sh.q_heads * 2);
    float* tv = a.take<float>(256);
    int* ti = a.take<int>(256);
    float* rope_cs = a.take<float>(hd);
    bf16_t* ao = a.take<bf16_t>(QS);
    void* smp_ws = a.take<char>(sample_workspace_bytes());
    VILA_REQUIRE(a.ok(), "llm_decode_w4: workspace arena overflow");
    const bool split256 = decode_attn_mode() >= 1 && cache->max_ctx <= 2048 && hd == 128 && QS <= 7 * 16 * 128;   // as the bf16 step (§4.3)
    VILA_TRY(launch_decode_prologue(B(w->embed), st->token, x, H, sh.vocab, st->pos, rope_cs, hd, sh.rope_theta, s));
    bf16_t* cur = x; bf16_t* nxt = x2;
    for (int l = 0; l < sh.n_layers; ++l) {
        const VilaLlmLayer& L = w->layers[l];
        const VilaLlmLayerW4& Q = ql[l];
        const size_t per_layer = (size_t)cache->n_slots * sh.kv_heads * cache->max_ctx * hd;
        bf16_t* kc = B(cache->k) + l * per_layer; bf16_t* vc = B(cache->v) + l * per_layer;
        GemvW4Args qa{};
        qa.x = cur; qa.norm_w = B(L.ln1_w); qa.eps = sh.rms_eps; qa.Wq = (const uint32_t*)Q.qkv_q; qa.Wsz = (const uint32_t*)Q.qkv_sz;
        qa.bias = B(L.bq); qa.K = H; qa.N = QS + 2 * sh.kv_heads * hd; qa.mode = 3; qa.q_out = q; qa.kcache = kc; qa.vcache = vc; qa.pos_ptr = st->pos;
        qa.rope_cs = rope_cs; qa.nq = sh.q_heads; qa.nkv = sh.kv_heads; qa.hd = hd; qa.max_ctx = cache->max_ctx;
        VILA_TRY(launch_gemv_w4(qa, s));
        AttnDecodeArgs ad{};
        ad.q = q; ad.kcache = kc; ad.vcache = vc; ad.o = ao; ad.part_o = part_o; ad.part_ml = part_ml; ad.pos_ptr = st->pos;
        ad.nq = sh.q_heads; ad.nkv = sh.kv_heads; ad.hd = hd; ad.max_ctx = cache->max_ctx; ad.n_splits = ns; ad.scale = 1.0f / sqrtf((float)hd);
        ad.split256 = split256 ? 1 : 0;                        // per-head blocks over 256-key slices; the slices meet in the o_proj kernel's prologue
        VILA_TRY(launch_attn_decode(ad, s));
        GemvW4Args o{};
        o.x = ao; o.Wq = (const uint32_t*)Q.o_q; o.Wsz = (const uint32_t*)Q.o_sz; o.residual = cur; o.y = nxt; o.N = H; o.K = QS; o.mode = 0;
        if (split256) { o.mode = 4; o.part_o = part_o; o.part_ml = part_ml; o.pos_ptr = st->pos; o.n_splits = cdiv(cache->max_ctx, 256); o.split_keys = 256; }
        VILA_TRY(launch_gemv_w4(o, s));
        GemvW4Args gu{};
        gu.x = nxt; gu.norm_w = B(L.ln2_w); gu.eps = sh.rms_eps; gu.Wq = (const uint32_t*)Q.gateup_q; gu.Wsz = (const uint32_t*)Q.gateup_sz;
        gu.y = act; gu.N = F; gu.K = H; gu.mode = 1;
        VILA_TRY(launch_gemv_w4(gu, s));
        GemvW4Args dn{};
        dn.x = act; dn.Wq = (const uint32_t*)Q.down_q; dn.Wsz = (const uint32_t*)Q.down_sz; dn.residual = nxt; dn.y = cur; dn.N = H; dn.K = F; dn.mode = 0;
        VILA_TRY(launch_gemv_w4(dn, s));
    }
    GemvArgs lm{};
    lm.x = cur; lm.norm_w = B(w->norm_w); lm.eps = sh.rms_eps; lm.W = B(w->lm_head); lm.y_f32 = st->logits; lm.N = sh.vocab; lm.K = H; lm.mode = 0;
    VILA_TRY(launch_gemv(lm, s));       // lm_head stays bf16 (as AWQ / TinyChat keep it fp16)
    if (sp != nullptr) VILA_TRY(launch_sample(st->logits, sh.vocab, sp->temperature, sp->top_k, sp->top_p, sp->seed, sp->seed_dev, st->pos, st->token, smp_ws, nullptr, s));
    else VILA_TRY(launch_argmax(st->logits, sh.vocab, st->token, tv, ti, s));
    VILA_TRY(launch_decode_advance(st->pos, st->token, st->out_ids, st->n_out, st->max_out, s));
    return 0;
}
