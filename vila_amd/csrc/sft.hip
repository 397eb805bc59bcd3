// vila_sft_fwd_bwd — one C-ABI call for the forward + backward of an SFT micro-batch.
//
// Replaces, for the step itself, what autograd does under HF `Trainer.training_step` as patched by the reference
// (llava/train/transformer_normalize_monkey_patch.py:183-249): `LlavaLlamaModel.forward` (llava_llama.py:94-159: `_embed` ->
// `repack_multimodal_data` -> `llm(..., labels)`), loss = sum CE / GLOBAL num_items (:261-268), backward through the LLM, the
// mm_projector and the vision tower (all three trainable: scripts/NVILA-Lite/sft.sh:25-27).  The integer work of `_embed` / repack
// (row maps, restarted positions, cu_seqlens, target rows) is planned on the host and handed in as index arrays (VilaSftBatch).
// The same sequence of kernels as vila_amd/train.py (which remains the Python mirror): explicit backward, no autograd graph, saved
// activations in the caller's workspace (no re-computation: 288 GB of HBM), dgrad / wgrad of the decoder and the head on the tensors as
// they lie (contraction-major GEMM operands; VILA_SFT_CM_VIT=0 sends the small ViT / projector GEMMs through transposed copies).
// After the LAST kernel touching a gradient bucket has been enqueued the host callback `cb(arg, bucket, index)` runs on the calling
// thread: that is where a caller records an event and starts the bucket's all-reduce / optimizer on its own streams (DDP-style overlap).
// Buckets arrive in backward order: LM_HEAD (untied only), FINAL_NORM, LLM_LAYER n-1 .. 0, EMBED, PROJECTOR, VIT_LAYER n-1 .. 0, VIT_EMBED.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/vila_hip.h"
#include "kernels.h"
#include "train.h"

namespace {
inline const bf16_t* B(const void* p) { return (const bf16_t*)p; }
inline bf16_t* B(void* p) { return (bf16_t*)p; }

struct Ws {                       // bump allocator over the caller's workspace; dry = size computation only (fake non-null base: the
    char* base; size_t off; bool dry;   // control flow tests pointers for NULL, so a dry take must not return NULL; never dereferenced)
    size_t peak = 0;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* r = (T*)(base + off);
        off += n * sizeof(T);
        if (off > peak) peak = off;
        return r;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }      // temporaries of a finished backward layer: all work is stream-ordered, so the space is reusable
};
struct Ctx { hipStream_t s; Ws* w; bool dry; float* gws; size_t gws_bytes; };
// fp32 slabs of the K-sliced GEMM launches (lm_head dgrad, tower shapes, tail tiles behind whole rounds)
constexpr size_t kSlabBytes = (size_t)128 << 20;

// VILA_SFT_DEBUG=1: name every launch on stderr before it is enqueued and wait for it (a device fault then points at its kernel)
static bool sft_debug() { static int v = -1; if (v < 0) { const char* e = getenv("VILA_SFT_DEBUG"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }
#define SFT_TRACE(what) do { if (!c.dry && sft_debug()) { (void)hipStreamSynchronize(c.s); fprintf(stderr, "sft[%s:%d] off=%zu %s\n", __func__, __LINE__, c.w->off, what); fflush(stderr); } } while (0)
// VILA_SFT_CM_VIT=0: tower / projector dgrad + wgrad through transposed copies (the round-1 path) instead of contraction-major operands
static bool vit_cm() { static int v = -1; if (v < 0) { const char* e = getenv("VILA_SFT_CM_VIT"); v = (e && e[0] == '0') ? 0 : 1; } return v == 1; }
#define RUN(call) do { if (!c.dry) { SFT_TRACE(#call); VILA_TRY(call); } } while (0)

int gemm(Ctx& c, const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const bf16_t* bias, const bf16_t* res, int64_t ldr, void* C, int64_t ldc,
         int M, int N, int K, int epi = EPI_NONE, int out_f32 = 0, int a_cm = 0, int b_cm = 0, int res_mod = 0, bool use_ws = false) {
    if (c.dry) return 0;
    if (sft_debug()) { (void)hipStreamSynchronize(c.s); fprintf(stderr, "sft gemm M=%d N=%d K=%d cm=%d%d epi=%d f32=%d res=%d ws=%d\n", M, N, K, a_cm, b_cm, epi, out_f32, res != nullptr, (int)use_ws); fflush(stderr); }
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.residual = res; g.ldr = ldr; g.C = C; g.ldc = ldc; g.out_f32 = out_f32;
    g.M = M; g.N = N; g.K = K; g.epi = epi; g.a_cm = a_cm; g.b_cm = b_cm; g.res_mod = res_mod;
    if (use_ws) { g.ws = c.gws; g.ws_bytes = c.gws_bytes; }
    return launch_gemm(g, c.s);
}

// x [M,K], w [N,K], dy [M,N] -> gw [N,K] (overwritten), gb [N] (nullable), dx [M,K] = dy . w (+ dx_res) (nullable)
int linear_bwd(Ctx& c, const bf16_t* x, const bf16_t* w, const bf16_t* dy, bf16_t* gw, bf16_t* gb, bf16_t* dx, const bf16_t* dx_res,
               int M, int N, int K, bool cm, bool use_ws = false) {
    if (cm && M >= 128 && N >= 128 && N % 8 == 0 && K % 8 == 0) {
        VILA_TRY(gemm(c, dy, N, x, K, nullptr, nullptr, 0, gw, K, N, K, M, EPI_NONE, 0, 1, 1, 0, use_ws));          // dW = dY^T X
        if (gb) { float* scr = c.w->take<float>(colsum_scratch_floats(M, N)); RUN(launch_colsum(dy, gb, scr, M, N, N, 0, 0, c.s)); }
        if (dx) VILA_TRY(gemm(c, dy, N, w, K, nullptr, dx_res, K, dx, K, M, K, N, EPI_NONE, 0, 0, 1, 0, use_ws));   // dX = dY W
        return 0;
    }
    const int Mp = (M + 63) / 64 * 64;
    bf16_t* dyt = c.w->take<bf16_t>((size_t)N * Mp);
    bf16_t* xt = c.w->take<bf16_t>((size_t)K * Mp);
    RUN(launch_transpose(dy, dyt, M, N, N, Mp, c.s));
    RUN(launch_transpose(x, xt, M, K, K, Mp, c.s));
    VILA_TRY(gemm(c, dyt, Mp, xt, Mp, nullptr, nullptr, 0, gw, K, N, K, Mp));
    if (gb) { float* scr = c.w->take<float>(colsum_scratch_floats(M, N)); RUN(launch_colsum(dy, gb, scr, M, N, N, 0, 0, c.s)); }
    if (dx) {
        const int Np = (N + 63) / 64 * 64;
        bf16_t* wt = c.w->take<bf16_t>((size_t)K * Np);
        if (!c.dry && sft_debug()) fprintf(stderr, "   w=%p wt=%p N=%d K=%d Np=%d\n", (const void*)w, (void*)wt, N, K, Np);
        RUN(launch_transpose(w, wt, N, K, K, Np, c.s));
        VILA_TRY(gemm(c, dy, N, wt, Np, nullptr, dx_res, K, dx, K, M, K, N));
    }
    return 0;
}

int norm_bwd(Ctx& c, const bf16_t* x, const bf16_t* w, const bf16_t* dy, bf16_t* dx, bf16_t* dw, bf16_t* db, int rows, int cols, float eps, int rms) {
    float* scr = c.w->take<float>(norm_bwd_scratch_floats(rows, cols));
    RUN(launch_norm_bwd(x, w, dy, dx, dw, db, scr, rows, cols, eps, rms, 0, c.s));
    return 0;
}

bool fused_qkv(const void* wq, const void* wk, const void* wv, const void* bq, const void* bk, const void* bv, size_t q_rows, size_t kv_rows, size_t cols) {
    return B(wk) == B(wq) + q_rows * cols && B(wv) == B(wk) + kv_rows * cols && (bq == nullptr || (B(bk) == B(bq) + q_rows && B(bv) == B(bk) + kv_rows));
}

struct VitSaved { bf16_t *x_in, *h1, *qkv, *a, *x_mid, *h2, *z1, *f; float* lse; };
struct LlmSaved { bf16_t *x_in, *h1, *qkv, *a, *x_mid, *h2, *g, *u, *act; float* lse; };

int run(const VilaVitWeights* vit, const VilaVitWeights* vg, const VilaProjWeights* pj, const VilaProjWeights* pg, const VilaLlmWeights* llm,
        const VilaLlmWeights* lg, const VilaSftBatch* b, float* loss_out, Ctx& c, VilaGradReadyCb cb, void* cb_arg) {
    Ws& a = *c.w;
    const VilaVitShape& vs = vit->shape;
    const VilaLlmShape& ls = llm->shape;
    const int n_img = b->n_images, T = b->total_tokens, H = ls.hidden, F = ls.inter, hd = ls.head_dim;
    const int QS = ls.q_heads * hd, KS = ls.kv_heads * hd, QKV = QS + 2 * KS;
    VILA_REQUIRE(T > 0 && b->n_seq > 0, "sft: empty batch");
    VILA_REQUIRE(QS == H, "sft: q_heads*head_dim (%d) must equal hidden (%d)", QS, H);
    auto ready = [&](int bucket, int index) { if (!c.dry && cb != nullptr) cb(cb_arg, bucket, index); };

    // ================= vision tower + projector forward =================
    const int g_ = vs.image / vs.patch, Nv = g_ * g_, Mv = n_img * Nv, D = vs.hidden, Fv = vs.inter, hdv = D / (vs.heads > 0 ? vs.heads : 1);
    const int Kc = vs.channels * vs.patch * vs.patch, Kp = (Kc + 7) / 8 * 8;
    const int kdown = (pj->kind == VILA_PROJ_MLP_DOWNSAMPLE_3X3_FIX) ? 3 : 2;
    // dynamic_s2: the projector sees s2_n_blocks inputs of n_scales * D channels (the merge sits between tower and projector)
    const bool s2 = b->s2_n_blocks > 0;
    const int n_pin = s2 ? b->s2_n_blocks : n_img;              // projector inputs ("images" of the projector)
    const int Dp = s2 ? b->s2_n_scales * D : D;                 // projector input channels
    const int gd = (g_ + kdown - 1) / kdown, Tm = gd * gd, C1 = kdown * kdown * Dp, Mp_ = n_pin * Tm;
    const int n_pools = n_img > 0 ? b->n_pools : 0;            // pooling video encoder: pooled rows behind the projector's rows
    const int Mbuf = n_pools > 0 ? b->n_media_rows : Mp_;
    VILA_REQUIRE(n_pools >= 0 && (n_pools == 0 || (b->pools != nullptr && Mbuf >= Mp_)), "sft: pools needs the host array and n_media_rows >= the projector's %d rows", Mp_);
    for (int i = 0; i < n_pools; ++i) {
        const int32_t* q = b->pools + 7 * i;
        VILA_REQUIRE(q[0] >= 0 && q[1] > 0 && q[0] + q[1] <= n_pin && q[2] > 0 && q[3] > 0 && q[4] > 0, "sft: pools[%d]: frames %d..%d of %d projector inputs, pool (%d, %d, %d)", i,
                     q[0], q[0] + q[1], n_pin, q[2], q[3], q[4]);
        VILA_REQUIRE(q[1] % q[2] == 0 && gd % q[3] == 0 && gd % q[4] == 0,
                     "shape '[%d, %d, %d]' is invalid for pooling by (%d, %d, %d): every pooled dimension must divide evenly", q[1], gd, gd, q[2], q[3], q[4]);
        VILA_REQUIRE(q[6] == (q[1] / q[2]) * (gd / q[3]) * (gd / q[4]) && q[5] >= Mp_ && q[5] + q[6] <= Mbuf, "sft: pools[%d]: rows %d..%d outside the pooled part %d..%d of the buffer",
                     i, q[5], q[5] + q[6], Mp_, Mbuf);
    }
    int s2_splits[4] = {1, 1, 1, 1};
    for (int k = 0; s2 && k < b->s2_n_scales - 1 && k < 4; ++k) s2_splits[k] = b->s2_splits[k];
    std::vector<VitSaved> vsv(vs.n_layers_run);
    bf16_t *patches = nullptr, *vx = nullptr, *p_y = nullptr, *p_yn = nullptr, *p_z1 = nullptr, *p_h1 = nullptr, *p_h1n = nullptr, *p_z2 = nullptr, *p_h2 = nullptr,
           *proj = nullptr;
    if (n_img > 0) {
        VILA_REQUIRE(pj->in_dim == Dp, "sft: projector in_dim %d != %d (tower hidden %d x %d scales)", pj->in_dim, Dp, D, s2 ? b->s2_n_scales : 1);
        VILA_REQUIRE(!s2 || (b->s2_desc != nullptr && b->s2_tile_desc != nullptr && b->s2_n_scales >= 1 && b->s2_n_scales <= 4),
                     "sft: dynamic_s2 needs s2_desc, s2_tile_desc and 1..4 scales");
        VILA_REQUIRE(b->pixels != nullptr, "sft: pixels are NULL");
        patches = a.take<bf16_t>((size_t)Mv * Kp);
        bf16_t* wpad = a.take<bf16_t>((size_t)D * Kp);
        vx = a.take<bf16_t>((size_t)Mv * D);
        RUN(launch_im2col(B(b->pixels), patches, n_img, vs.channels, vs.image, vs.image, vs.patch, Kp, c.s));
        RUN(launch_pad_rows(B(vit->patch_w), wpad, D, Kc, Kp, c.s));
        VILA_TRY(gemm(c, patches, Kp, wpad, Kp, B(vit->patch_b), B(vit->pos_emb), D, vx, D, Mv, D, Kp, EPI_NONE, 0, 0, 0, Nv));
        bf16_t* x = vx;
        for (int l = 0; l < vs.n_layers_run; ++l) {
            const VilaVitLayer& L = vit->layers[l];
            VILA_REQUIRE(fused_qkv(L.wq, L.wk, L.wv, L.bq, L.bk, L.bv, D, D, D), "sft: ViT q/k/v weights and biases must be views of one fused buffer");
            VitSaved& s = vsv[l];
            s.x_in = x;
            s.h1 = a.take<bf16_t>((size_t)Mv * D); s.qkv = a.take<bf16_t>((size_t)Mv * 3 * D); s.a = a.take<bf16_t>((size_t)Mv * D);
            s.lse = a.take<float>((size_t)vs.heads * Mv); s.x_mid = a.take<bf16_t>((size_t)Mv * D); s.h2 = a.take<bf16_t>((size_t)Mv * D);
            s.z1 = a.take<bf16_t>((size_t)Mv * Fv); s.f = a.take<bf16_t>((size_t)Mv * Fv);
            bf16_t* xo = a.take<bf16_t>((size_t)Mv * D);
            RUN(launch_layernorm(x, B(L.ln1_w), B(L.ln1_b), s.h1, Mv, D, vs.ln_eps, c.s));
            VILA_TRY(gemm(c, s.h1, D, B(L.wq), D, B(L.bq), nullptr, 0, s.qkv, 3 * D, Mv, 3 * D, D));
            AttnArgs at{};
            at.q = s.qkv; at.k = s.qkv + D; at.v = s.qkv + 2 * D; at.o = s.a;
            at.q_tok_stride = at.k_tok_stride = at.v_tok_stride = 3 * D; at.o_tok_stride = D;
            at.q_head_stride = at.k_head_stride = at.v_head_stride = at.o_head_stride = hdv;
            at.cu_seqlens = nullptr; at.n_seq = n_img; at.total_tokens = Mv; at.max_seqlen = Nv;
            at.n_q_heads = at.n_kv_heads = vs.heads; at.head_dim = hdv; at.causal = 0; at.scale = 1.0f / sqrtf((float)hdv); at.lse = s.lse;
            RUN(launch_attn_fwd(at, c.s));
            VILA_TRY(gemm(c, s.a, D, B(L.wo), D, B(L.bo), x, D, s.x_mid, D, Mv, D, D));
            RUN(launch_layernorm(s.x_mid, B(L.ln2_w), B(L.ln2_b), s.h2, Mv, D, vs.ln_eps, c.s));
            VILA_TRY(gemm(c, s.h2, D, B(L.fc1_w), D, B(L.fc1_b), nullptr, 0, s.z1, Fv, Mv, Fv, D));
            RUN(launch_act_fwd(s.z1, s.f, (int64_t)Mv * Fv, 1, c.s));
            VILA_TRY(gemm(c, s.f, Fv, B(L.fc2_w), Fv, B(L.fc2_b), s.x_mid, D, xo, D, Mv, D, Fv));
            x = xo;
        }
        // projector (base_projector.py:145-174): space-to-depth -> LN -> Linear -> GELU(erf) [-> LN -> Linear -> GELU] -> Linear
        if (s2) {                                             // merge_features_for_dynamic_s2 + split_chessboard + rearrange as one gather
            bf16_t* merged = a.take<bf16_t>((size_t)n_pin * Nv * Dp);
            RUN(launch_s2_merge(x, merged, b->s2_desc, n_pin, g_, D, b->s2_n_scales, s2_splits, c.s));
            x = merged;
        }
        p_y = a.take<bf16_t>((size_t)Mp_ * C1); p_yn = a.take<bf16_t>((size_t)Mp_ * C1);
        RUN(launch_space_to_depth(x, p_y, n_pin, g_, Dp, kdown, c.s));
        RUN(launch_layernorm(p_y, B(pj->ln1_w), B(pj->ln1_b), p_yn, Mp_, C1, 1e-5f, c.s));
        proj = a.take<bf16_t>((size_t)Mbuf * H);                      // [projector rows | pooled rows]
        if (kdown == 2) {
            p_z1 = a.take<bf16_t>((size_t)Mp_ * H); p_h1 = a.take<bf16_t>((size_t)Mp_ * H);
            VILA_TRY(gemm(c, p_yn, C1, B(pj->fc1_w), C1, B(pj->fc1_b), nullptr, 0, p_z1, H, Mp_, H, C1));
            RUN(launch_act_fwd(p_z1, p_h1, (int64_t)Mp_ * H, 2, c.s));
            VILA_TRY(gemm(c, p_h1, H, B(pj->fc2_w), H, B(pj->fc2_b), nullptr, 0, proj, H, Mp_, H, H));
        } else {
            const int C3 = 3 * Dp;
            p_z1 = a.take<bf16_t>((size_t)Mp_ * C3); p_h1 = a.take<bf16_t>((size_t)Mp_ * C3); p_h1n = a.take<bf16_t>((size_t)Mp_ * C3);
            p_z2 = a.take<bf16_t>((size_t)Mp_ * H); p_h2 = a.take<bf16_t>((size_t)Mp_ * H);
            VILA_TRY(gemm(c, p_yn, C1, B(pj->fc1_w), C1, B(pj->fc1_b), nullptr, 0, p_z1, C3, Mp_, C3, C1));
            RUN(launch_act_fwd(p_z1, p_h1, (int64_t)Mp_ * C3, 2, c.s));
            RUN(launch_layernorm(p_h1, B(pj->ln2_w), B(pj->ln2_b), p_h1n, Mp_, C3, 1e-5f, c.s));
            VILA_TRY(gemm(c, p_h1n, C3, B(pj->fc2_w), C3, B(pj->fc2_b), nullptr, 0, p_z2, H, Mp_, H, C3));
            RUN(launch_act_fwd(p_z2, p_h2, (int64_t)Mp_ * H, 2, c.s));
            VILA_TRY(gemm(c, p_h2, H, B(pj->fc3_w), H, B(pj->fc3_b), nullptr, 0, proj, H, Mp_, H, H));
        }
    }

    // pooled rows of a pooling video encoder (tsp.py:28-52), behind the projector's rows
    for (int i = 0; i < n_pools; ++i) {
        const int32_t* q = b->pools + 7 * i;
        RUN(launch_video_pool(proj + (size_t)q[0] * Tm * H, proj + (size_t)q[5] * H, q[1], gd, H, q[2], q[3], q[4], nullptr, 0, nullptr, 0, c.s));
    }

    // ================= splice into the packed row (llava_arch.py:412-490, 744-800: planned on the host) =================
    bf16_t* x0 = a.take<bf16_t>((size_t)T * H);
    const int32_t* nl_src = b->nl_src;
    if (!c.dry) VILA_HIP(hipMemsetAsync(x0, 0, (size_t)T * H * 2, c.s));
    RUN(launch_copy_rows(B(llm->embed), x0, b->txt_src, b->txt_dst, b->n_txt, H, c.s));
    if (n_img > 0 && b->n_feat > 0) RUN(launch_copy_rows(proj, x0, b->feat_src, b->feat_dst, b->n_feat, H, c.s));
    if (b->n_nl > 0) RUN(launch_copy_rows(B(llm->embed), x0, nl_src, b->nl_dst, b->n_nl, H, c.s));

    // ================= LLM forward (saved activations, varlen causal attention) =================
    float* cs = a.take<float>((size_t)T * hd / 2);
    float* sn = a.take<float>((size_t)T * hd / 2);
    RUN(launch_rope_table(b->positions, cs, sn, T, hd, ls.rope_theta, c.s));
    std::vector<LlmSaved> lsv(ls.n_layers);
    bf16_t* x = x0;
    for (int l = 0; l < ls.n_layers; ++l) {
        const VilaLlmLayer& L = llm->layers[l];
        VILA_REQUIRE(fused_qkv(L.wq, L.wk, L.wv, L.bq, L.bk, L.bv, QS, KS, H), "sft: q/k/v weights and biases must be views of one fused [q+2kv, hidden] buffer");
        LlmSaved& s = lsv[l];
        s.x_in = x;
        s.h1 = a.take<bf16_t>((size_t)T * H); s.qkv = a.take<bf16_t>((size_t)T * QKV); s.a = a.take<bf16_t>((size_t)T * QS);
        s.lse = a.take<float>((size_t)ls.q_heads * T); s.x_mid = a.take<bf16_t>((size_t)T * H); s.h2 = a.take<bf16_t>((size_t)T * H);
        s.g = a.take<bf16_t>((size_t)T * F); s.u = a.take<bf16_t>((size_t)T * F); s.act = a.take<bf16_t>((size_t)T * F);
        bf16_t* xo = a.take<bf16_t>((size_t)T * H);
        RUN(launch_rmsnorm(x, B(L.ln1_w), s.h1, T, H, ls.rms_eps, c.s));
        VILA_TRY(gemm(c, s.h1, H, B(L.wq), H, B(L.bq), nullptr, 0, s.qkv, QKV, T, QKV, H));
        RUN(launch_rope_kv(s.qkv, cs, sn, b->positions, nullptr, nullptr, nullptr, T, ls.q_heads, ls.kv_heads, hd, 0, c.s));
        AttnArgs at{};
        at.q = s.qkv; at.k = s.qkv + QS; at.v = s.qkv + QS + KS; at.o = s.a;
        at.q_tok_stride = at.k_tok_stride = at.v_tok_stride = QKV; at.o_tok_stride = QS;
        at.q_head_stride = at.k_head_stride = at.v_head_stride = at.o_head_stride = hd;
        at.cu_seqlens = b->cu_seqlens; at.n_seq = b->n_seq; at.total_tokens = T; at.max_seqlen = b->max_seqlen;
        at.n_q_heads = ls.q_heads; at.n_kv_heads = ls.kv_heads; at.head_dim = hd; at.causal = 1; at.scale = 1.0f / sqrtf((float)hd); at.lse = s.lse;
        RUN(launch_attn_fwd(at, c.s));
        VILA_TRY(gemm(c, s.a, QS, B(L.wo), QS, nullptr, x, H, s.x_mid, H, T, H, QS));
        RUN(launch_rmsnorm(s.x_mid, B(L.ln2_w), s.h2, T, H, ls.rms_eps, c.s));
        VILA_TRY(gemm(c, s.h2, H, B(L.w_gate), H, nullptr, nullptr, 0, s.g, F, T, F, H));
        VILA_TRY(gemm(c, s.h2, H, B(L.w_up), H, nullptr, nullptr, 0, s.u, F, T, F, H));
        RUN(launch_silu_mul_fwd(s.g, s.u, s.act, (int64_t)T * F, c.s));
        VILA_TRY(gemm(c, s.act, F, B(L.w_down), F, nullptr, s.x_mid, H, xo, H, T, H, F, EPI_NONE, 0, 0, 0, 0, true));
        x = xo;
    }
    bf16_t* x_out = x;
    bf16_t* hn = a.take<bf16_t>((size_t)T * H);
    RUN(launch_rmsnorm(x_out, B(llm->norm_w), hn, T, H, ls.rms_eps, c.s));

    // ================= loss on the rows that have a target; dL/dlogits; head gradients =================
    const int nt = b->n_targets;
    bf16_t* dhn = a.take<bf16_t>((size_t)T * H);
    if (!c.dry) VILA_HIP(hipMemsetAsync(dhn, 0, (size_t)T * H * 2, c.s));
    if (!c.dry) VILA_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), c.s));
    const bool tied = llm->lm_head == llm->embed;
    bf16_t* g_head = tied ? B((void*)lg->embed) : B((void*)lg->lm_head);
    if (nt > 0) {
        // lm_head + CE in chunks of SFT_LOGIT_CHUNK target rows: the fp32 logits are never materialised for the whole batch (SURVEY §7 step 7;
        // the reference does, llava_llama.py:134-149) — 256 x vocab x 4 B at a time, the bf16 dlogits of all rows feed ONE wgrad / dgrad pair
        constexpr int SFT_LOGIT_CHUNK = 256;
        const int lc = nt < SFT_LOGIT_CHUNK ? nt : SFT_LOGIT_CHUNK;
        bf16_t* hv = a.take<bf16_t>((size_t)nt * H);
        float* logits = a.take<float>((size_t)lc * ls.vocab);
        bf16_t* dlog = a.take<bf16_t>((size_t)nt * ls.vocab);
        bf16_t* dhv = a.take<bf16_t>((size_t)nt * H);
        float* row_loss = a.take<float>((size_t)lc);
        RUN(launch_copy_rows(hn, hv, b->target_rows, nullptr, nt, H, c.s));
        for (int r0 = 0; r0 < nt; r0 += lc) {
            const int rn = nt - r0 < lc ? nt - r0 : lc;
            VILA_TRY(gemm(c, hv + (size_t)r0 * H, H, B(llm->lm_head), H, nullptr, nullptr, 0, logits, ls.vocab, rn, ls.vocab, H, EPI_NONE, 1));
            RUN(launch_ce(logits, b->targets + r0, dlog + (size_t)r0 * ls.vocab, loss_out, row_loss, rn, ls.vocab, ls.vocab, b->loss_scale, c.s));
        }
        VILA_TRY(linear_bwd(c, hv, B(llm->lm_head), dlog, g_head, nullptr, dhv, nullptr, nt, ls.vocab, H, true, true));
        RUN(launch_copy_rows(dhv, dhn, nullptr, b->target_rows, nt, H, c.s));
    }
    if (!tied) ready(VILA_BUCKET_LM_HEAD, 0);
    bf16_t* dx = a.take<bf16_t>((size_t)T * H);
    VILA_TRY(norm_bwd(c, x_out, B(llm->norm_w), dhn, dx, B((void*)lg->norm_w), nullptr, T, H, ls.rms_eps, 1));
    ready(VILA_BUCKET_FINAL_NORM, 0);

    // ================= LLM backward =================
    bf16_t* dx_pp[2] = {a.take<bf16_t>((size_t)T * H), a.take<bf16_t>((size_t)T * H)};      // the dX chain ping-pongs between two buffers
    for (int l = ls.n_layers - 1; l >= 0; --l) {
        const VilaLlmLayer& L = llm->layers[l];
        const VilaLlmLayer& G = lg->layers[l];
        LlmSaved& s = lsv[l];
        const size_t layer_mark = a.mark();
        bf16_t* dact = a.take<bf16_t>((size_t)T * F);
        VILA_TRY(linear_bwd(c, s.act, B(L.w_down), dx, B((void*)G.w_down), nullptr, dact, nullptr, T, H, F, true, true));
        bf16_t* dg = a.take<bf16_t>((size_t)T * F);
        bf16_t* du = a.take<bf16_t>((size_t)T * F);
        RUN(launch_silu_mul_bwd(s.g, s.u, dact, dg, du, (int64_t)T * F, c.s));
        bf16_t* dh2a = a.take<bf16_t>((size_t)T * H);
        bf16_t* dh2 = a.take<bf16_t>((size_t)T * H);
        VILA_TRY(linear_bwd(c, s.h2, B(L.w_gate), dg, B((void*)G.w_gate), nullptr, dh2a, nullptr, T, F, H, true, true));
        VILA_TRY(linear_bwd(c, s.h2, B(L.w_up), du, B((void*)G.w_up), nullptr, dh2, dh2a, T, F, H, true, true));
        bf16_t* dxm = a.take<bf16_t>((size_t)T * H);
        VILA_TRY(norm_bwd(c, s.x_mid, B(L.ln2_w), dh2, dxm, B((void*)G.ln2_w), nullptr, T, H, ls.rms_eps, 1));
        bf16_t* dx_mid = a.take<bf16_t>((size_t)T * H);
        RUN(launch_add(dx, dxm, dx_mid, (int64_t)T * H, c.s));
        bf16_t* da = a.take<bf16_t>((size_t)T * QS);
        VILA_TRY(linear_bwd(c, s.a, B(L.wo), dx_mid, B((void*)G.wo), nullptr, da, nullptr, T, H, QS, true, true));
        bf16_t* dqkv = a.take<bf16_t>((size_t)T * QKV);
        float* delta = a.take<float>((size_t)ls.q_heads * T);
        AttnBwdArgs ab{};
        ab.q = s.qkv; ab.k = s.qkv + QS; ab.v = s.qkv + QS + KS; ab.o = s.a; ab.d_o = da; ab.dq = dqkv; ab.dk = dqkv + QS; ab.dv = dqkv + QS + KS;
        ab.q_tok_stride = ab.k_tok_stride = ab.v_tok_stride = QKV; ab.o_tok_stride = QS; ab.do_tok_stride = QS;
        ab.dq_tok_stride = ab.dk_tok_stride = ab.dv_tok_stride = QKV;
        ab.q_head_stride = ab.k_head_stride = ab.v_head_stride = ab.o_head_stride = ab.do_head_stride = ab.dq_head_stride = ab.dk_head_stride = ab.dv_head_stride = hd;
        ab.cu_seqlens = b->cu_seqlens; ab.n_seq = b->n_seq; ab.total_tokens = T; ab.max_seqlen = b->max_seqlen;
        ab.n_q_heads = ls.q_heads; ab.n_kv_heads = ls.kv_heads; ab.head_dim = hd; ab.causal = 1; ab.scale = 1.0f / sqrtf((float)hd); ab.lse = s.lse; ab.delta = delta;
        RUN(launch_attn_bwd(ab, c.s));
        RUN(launch_rope_bwd(dqkv, cs, sn, T, ls.q_heads, ls.kv_heads, hd, c.s));
        bf16_t* dh1 = a.take<bf16_t>((size_t)T * H);
        VILA_TRY(linear_bwd(c, s.h1, B(L.wq), dqkv, B((void*)G.wq), B((void*)G.bq), dh1, nullptr, T, QKV, H, true, true));
        bf16_t* dxi = a.take<bf16_t>((size_t)T * H);
        VILA_TRY(norm_bwd(c, s.x_in, B(L.ln1_w), dh1, dxi, B((void*)G.ln1_w), nullptr, T, H, ls.rms_eps, 1));
        bf16_t* dnext = dx_pp[l & 1] != dx ? dx_pp[l & 1] : dx_pp[(l & 1) ^ 1];
        RUN(launch_add(dx_mid, dxi, dnext, (int64_t)T * H, c.s));
        dx = dnext;
        a.release(layer_mark);
        ready(VILA_BUCKET_LLM_LAYER, l);
    }

    // ================= embedding rows (text + "\n") and media rows =================
    bf16_t* ge = B((void*)lg->embed);
    if (b->n_txt > 0) {
        bf16_t* dtxt = a.take<bf16_t>((size_t)b->n_txt * H);
        RUN(launch_copy_rows(dx, dtxt, b->txt_dst, nullptr, b->n_txt, H, c.s));
        RUN(launch_scatter_add_rows(dtxt, ge, b->txt_src, b->n_txt, H, c.s));
    }
    if (b->n_nl > 0) {
        bf16_t* dnl = a.take<bf16_t>((size_t)b->n_nl * H);
        RUN(launch_copy_rows(dx, dnl, b->nl_dst, nullptr, b->n_nl, H, c.s));
        RUN(launch_scatter_add_rows(dnl, ge, nl_src, b->n_nl, H, c.s));
    }
    ready(VILA_BUCKET_EMBED, 0);
    if (n_img == 0) return 0;

    // ================= projector backward =================
    bf16_t* dproj = a.take<bf16_t>((size_t)Mbuf * H);
    if (!c.dry) VILA_HIP(hipMemsetAsync(dproj, 0, (size_t)Mbuf * H * 2, c.s));           // rows cut off by the truncation keep a zero gradient
    if (b->n_feat > 0) RUN(launch_copy_rows(dx, dproj, b->feat_dst, b->feat_src, b->n_feat, H, c.s));
    for (int i = 0; i < n_pools; ++i) {                                                   // pooled rows' gradients back onto their frames' rows
        const int32_t* q = b->pools + 7 * i;
        RUN(launch_video_pool_bwd(dproj + (size_t)q[5] * H, dproj + (size_t)q[0] * Tm * H, q[1], gd, H, q[2], q[3], q[4], 1, c.s));
    }
    bf16_t* dz1 = nullptr;
    if (kdown == 2) {
        bf16_t* dh1 = a.take<bf16_t>((size_t)Mp_ * H);
        VILA_TRY(linear_bwd(c, p_h1, B(pj->fc2_w), dproj, B((void*)pg->fc2_w), B((void*)pg->fc2_b), dh1, nullptr, Mp_, H, H, vit_cm(), true));
        dz1 = a.take<bf16_t>((size_t)Mp_ * H);
        RUN(launch_act_bwd(p_z1, dh1, dz1, (int64_t)Mp_ * H, 2, c.s));
    } else {
        const int C3 = 3 * Dp;
        bf16_t* dh2 = a.take<bf16_t>((size_t)Mp_ * H);
        VILA_TRY(linear_bwd(c, p_h2, B(pj->fc3_w), dproj, B((void*)pg->fc3_w), B((void*)pg->fc3_b), dh2, nullptr, Mp_, H, H, vit_cm(), true));
        bf16_t* dz2 = a.take<bf16_t>((size_t)Mp_ * H);
        RUN(launch_act_bwd(p_z2, dh2, dz2, (int64_t)Mp_ * H, 2, c.s));
        bf16_t* dh1n = a.take<bf16_t>((size_t)Mp_ * C3);
        VILA_TRY(linear_bwd(c, p_h1n, B(pj->fc2_w), dz2, B((void*)pg->fc2_w), B((void*)pg->fc2_b), dh1n, nullptr, Mp_, H, C3, vit_cm(), true));
        bf16_t* dh1 = a.take<bf16_t>((size_t)Mp_ * C3);
        VILA_TRY(norm_bwd(c, p_h1, B(pj->ln2_w), dh1n, dh1, B((void*)pg->ln2_w), B((void*)pg->ln2_b), Mp_, C3, 1e-5f, 0));
        dz1 = a.take<bf16_t>((size_t)Mp_ * C3);
        RUN(launch_act_bwd(p_z1, dh1, dz1, (int64_t)Mp_ * C3, 2, c.s));
    }
    const int N1 = (kdown == 2) ? H : 3 * Dp;
    bf16_t* dyn = a.take<bf16_t>((size_t)Mp_ * C1);
    VILA_TRY(linear_bwd(c, p_yn, B(pj->fc1_w), dz1, B((void*)pg->fc1_w), B((void*)pg->fc1_b), dyn, nullptr, Mp_, N1, C1, vit_cm(), true));
    bf16_t* dy = a.take<bf16_t>((size_t)Mp_ * C1);
    VILA_TRY(norm_bwd(c, p_y, B(pj->ln1_w), dyn, dy, B((void*)pg->ln1_w), B((void*)pg->ln1_b), Mp_, C1, 1e-5f, 0));
    ready(VILA_BUCKET_PROJECTOR, 0);
    bf16_t* dv = a.take<bf16_t>((size_t)Mv * D);
    if (s2) {
        bf16_t* dmerged = a.take<bf16_t>((size_t)n_pin * Nv * Dp);
        RUN(launch_depth_to_space(dy, dmerged, n_pin, g_, Dp, kdown, c.s));
        RUN(launch_s2_merge_bwd(dmerged, dv, b->s2_tile_desc, n_img, g_, D, b->s2_n_scales, s2_splits, c.s));
    } else {
        RUN(launch_depth_to_space(dy, dv, n_img, g_, D, kdown, c.s));
    }

    // ================= vision tower backward =================
    bf16_t* dv_pp[2] = {a.take<bf16_t>((size_t)Mv * D), a.take<bf16_t>((size_t)Mv * D)};
    for (int l = vs.n_layers_run - 1; l >= 0; --l) {
        const VilaVitLayer& L = vit->layers[l];
        const VilaVitLayer& G = vg->layers[l];
        VitSaved& s = vsv[l];
        const size_t layer_mark = a.mark();
        bf16_t* df = a.take<bf16_t>((size_t)Mv * Fv);
        VILA_TRY(linear_bwd(c, s.f, B(L.fc2_w), dv, B((void*)G.fc2_w), B((void*)G.fc2_b), df, nullptr, Mv, D, Fv, vit_cm(), true));
        bf16_t* dz = a.take<bf16_t>((size_t)Mv * Fv);
        RUN(launch_act_bwd(s.z1, df, dz, (int64_t)Mv * Fv, 1, c.s));
        bf16_t* dh2 = a.take<bf16_t>((size_t)Mv * D);
        VILA_TRY(linear_bwd(c, s.h2, B(L.fc1_w), dz, B((void*)G.fc1_w), B((void*)G.fc1_b), dh2, nullptr, Mv, Fv, D, vit_cm(), true));
        bf16_t* dxm = a.take<bf16_t>((size_t)Mv * D);
        VILA_TRY(norm_bwd(c, s.x_mid, B(L.ln2_w), dh2, dxm, B((void*)G.ln2_w), B((void*)G.ln2_b), Mv, D, vs.ln_eps, 0));
        bf16_t* dx_mid = a.take<bf16_t>((size_t)Mv * D);
        RUN(launch_add(dv, dxm, dx_mid, (int64_t)Mv * D, c.s));
        bf16_t* da = a.take<bf16_t>((size_t)Mv * D);
        VILA_TRY(linear_bwd(c, s.a, B(L.wo), dx_mid, B((void*)G.wo), B((void*)G.bo), da, nullptr, Mv, D, D, vit_cm(), true));
        bf16_t* dqkv = a.take<bf16_t>((size_t)Mv * 3 * D);
        float* delta = a.take<float>((size_t)vs.heads * Mv);
        AttnBwdArgs ab{};
        ab.q = s.qkv; ab.k = s.qkv + D; ab.v = s.qkv + 2 * D; ab.o = s.a; ab.d_o = da; ab.dq = dqkv; ab.dk = dqkv + D; ab.dv = dqkv + 2 * D;
        ab.q_tok_stride = ab.k_tok_stride = ab.v_tok_stride = 3 * D; ab.o_tok_stride = D; ab.do_tok_stride = D;
        ab.dq_tok_stride = ab.dk_tok_stride = ab.dv_tok_stride = 3 * D;
        ab.q_head_stride = ab.k_head_stride = ab.v_head_stride = ab.o_head_stride = ab.do_head_stride = ab.dq_head_stride = ab.dk_head_stride = ab.dv_head_stride = hdv;
        ab.cu_seqlens = nullptr; ab.n_seq = n_img; ab.total_tokens = Mv; ab.max_seqlen = Nv;
        ab.n_q_heads = ab.n_kv_heads = vs.heads; ab.head_dim = hdv; ab.causal = 0; ab.scale = 1.0f / sqrtf((float)hdv); ab.lse = s.lse; ab.delta = delta;
        RUN(launch_attn_bwd(ab, c.s));
        bf16_t* dh1 = a.take<bf16_t>((size_t)Mv * D);
        VILA_TRY(linear_bwd(c, s.h1, B(L.wq), dqkv, B((void*)G.wq), B((void*)G.bq), dh1, nullptr, Mv, 3 * D, D, vit_cm(), true));
        bf16_t* dxi = a.take<bf16_t>((size_t)Mv * D);
        VILA_TRY(norm_bwd(c, s.x_in, B(L.ln1_w), dh1, dxi, B((void*)G.ln1_w), B((void*)G.ln1_b), Mv, D, vs.ln_eps, 0));
        bf16_t* dnext = dv_pp[l & 1] != dv ? dv_pp[l & 1] : dv_pp[(l & 1) ^ 1];
        RUN(launch_add(dx_mid, dxi, dnext, (int64_t)Mv * D, c.s));
        dv = dnext;
        a.release(layer_mark);
        ready(VILA_BUCKET_VIT_LAYER, l);
    }
    // patch embedding: weight (un-padded), bias, position embedding (summed over the images)
    {
        const int Mp = (Mv + 63) / 64 * 64;
        bf16_t* dvt = a.take<bf16_t>((size_t)D * Mp);
        bf16_t* pt = a.take<bf16_t>((size_t)Kp * Mp);
        bf16_t* gwp = a.take<bf16_t>((size_t)D * Kp);
        float* scr = a.take<float>(colsum_scratch_floats(Mv, D));
        RUN(launch_transpose(dv, dvt, Mv, D, D, Mp, c.s));
        RUN(launch_transpose(patches, pt, Mv, Kp, Kp, Mp, c.s));
        VILA_TRY(gemm(c, dvt, Mp, pt, Mp, nullptr, nullptr, 0, gwp, Kp, D, Kp, Mp));
        if (!c.dry) VILA_HIP(hipMemcpy2DAsync((void*)vg->patch_w, (size_t)Kc * 2, gwp, (size_t)Kp * 2, (size_t)Kc * 2, D, hipMemcpyDeviceToDevice, c.s));
        RUN(launch_colsum(dv, B((void*)vg->patch_b), scr, Mv, D, D, 0, 0, c.s));
        RUN(launch_colsum(dv, B((void*)vg->pos_emb), nullptr, Mv, D, D, 0, Nv, c.s));
    }
    ready(VILA_BUCKET_VIT_EMBED, 0);
    return 0;
}
}  // namespace

extern "C" size_t vila_sft_workspace_bytes(const VilaVitWeights* vit, const VilaProjWeights* proj, const VilaLlmWeights* llm, const VilaSftBatch* batch) {
    Ws w{(char*)4096, 0, true};
    Ctx c{nullptr, &w, true, nullptr, 0};
    if (run(vit, vit, proj, proj, llm, llm, batch, nullptr, c, nullptr, nullptr) != 0) return 0;
    return w.peak + kSlabBytes + 4096;                       // + the fp32 slab region of the K-sliced GEMM launches
}

extern "C" int vila_sft_fwd_bwd(const VilaVitWeights* vit, const VilaVitWeights* vit_grad, const VilaProjWeights* proj, const VilaProjWeights* proj_grad,
                                const VilaLlmWeights* llm, const VilaLlmWeights* llm_grad, const VilaSftBatch* batch, float* loss_out,
                                void* workspace, size_t workspace_bytes, VilaGradReadyCb cb, void* cb_arg, vila_stream_t stream) {
    VILA_REQUIRE(vit && vit_grad && proj && proj_grad && llm && llm_grad && batch && loss_out && workspace, "sft: NULL argument");
    const size_t need = vila_sft_workspace_bytes(vit, proj, llm, batch);
    VILA_REQUIRE(need != 0, "sft: %s", vila_last_error());
    VILA_REQUIRE(workspace_bytes >= need, "sft: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    const size_t slab = kSlabBytes;
    if (sft_debug()) fprintf(stderr, "sft: workspace %p bytes %zu need %zu\n", workspace, workspace_bytes, need);
    Ws w{(char*)workspace + slab, 0, false};
    Ctx c{vila_stream_enter((void*)stream), &w, false, (float*)workspace, slab};
    return run(vit, vit_grad, proj, proj_grad, llm, llm_grad, batch, loss_out, c, cb, cb_arg);
}
