// Batch-1 decode kernels (HOT LOOP #3): every weight byte is read exactly once per token, so
// the roofline is HBM (14.14 GB / token for NVILA-8B bf16), not MFMA.  Design rules (cdna guide, "GEMV / M<=16"):
// weights go straight HBM -> VGPR with 16-B non-temporal loads, deep unroll, late wait; the activation vector is
// staged once per block in LDS (with the preceding RMSNorm fused in, HF rounding order kept); all epilogues
// (bias, RoPE, KV-cache write, SiLU*up, residual add) are fused so a decoder layer is 6 launches.
#include "kernels.h"

#define GEMV_U 4

__device__ __forceinline__ u32x4 ldg_nt(const bf16_t* p) { return __builtin_nontemporal_load((const u32x4*)p); }

__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc = fmaf(lo_bf(w[k]), lo_bf(x[k]), acc);
        acc = fmaf(hi_bf(w[k]), hi_bf(x[k]), acc);
    }
    return acc;
}

// stage x (optionally RMS-normalised with gain) as bf16 into LDS; all 256 threads participate
__device__ __forceinline__ void stage_x(const bf16_t* __restrict__ x, const bf16_t* __restrict__ norm_w, float eps, int K,
                                        bf16_t* sx, float* scratch) {
    const int tid = threadIdx.x, nch = K >> 3;
    if (norm_w == nullptr) {
        for (int c = tid; c < nch; c += 256) *(u32x4*)(sx + c * 8) = *(const u32x4*)(x + c * 8);
        __syncthreads();
        return;
    }
    float s = 0.f;
    for (int c = tid; c < nch; c += 256) {
        const u32x4 v = *(const u32x4*)(x + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float a = lo_bf(v[k]), b = hi_bf(v[k]); s += a * a + b * b; }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) scratch[tid >> 6] = s;
    __syncthreads();
    const float rstd = rsqrtf((scratch[0] + scratch[1] + scratch[2] + scratch[3]) / K + eps);
    for (int c = tid; c < nch; c += 256) {
        const u32x4 v = *(const u32x4*)(x + c * 8);
        const u32x4 g = *(const u32x4*)(norm_w + c * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack2bf(lo_bf(g[k]) * bfround(lo_bf(v[k]) * rstd), hi_bf(g[k]) * bfround(hi_bf(v[k]) * rstd));
        *(u32x4*)(sx + c * 8) = o;
    }
    __syncthreads();
}

// R rows x K dot products for one wave; rows given by pointers
template <int R>
__device__ __forceinline__ void wave_rows_dot(const bf16_t* const (&wrow)[R], const bf16_t* sx, int K, int lane, float (&acc)[R]) {
    const int nch = K >> 3;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < nch; c0 += 64 * GEMV_U) {
        u32x4 wv[GEMV_U][R];
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            const int c = c0 + u * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) wv[u][r] = (c < nch) ? ldg_nt(wrow[r] + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < GEMV_U; ++u) {
            const int c = c0 + u * 64 + lane;
            const u32x4 xv = (c < nch) ? *(const u32x4*)(sx + c * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = dot8(wv[u][r], xv, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

// ------------------------------------------------------------------------------------------------
// generic GEMV: y = W x (+bias) (+residual)   |   gate/up: y = silu(Wg x) * (Wu x)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs p, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sx = (bf16_t*)smem;
    float* scratch = (float*)(smem + ((p.K * 2 + 15) & ~15));
    stage_x(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        if (MODE == 1) {
            // two (gate, up) pairs per wave iteration
            const int n = g * 2;
            const int n1 = (n + 1 < p.N) ? n + 1 : n;
            const bf16_t* const rows[4] = {p.W + (int64_t)n * p.K, p.W2 + (int64_t)n * p.K, p.W + (int64_t)n1 * p.K, p.W2 + (int64_t)n1 * p.K};
            float acc[4];
            wave_rows_dot<4>(rows, sx, p.K, lane, acc);
            if (lane == 0) {
                // HF: down(act(gate(x)) * up(x)) with every tensor rounded to bf16
                const float g0 = bfround(acc[0]), u0 = bfround(acc[1]);
                p.y[n] = f2bf(bfround(silu_f(g0)) * u0);
                if (n + 1 < p.N) {
                    const float g1 = bfround(acc[2]), u1 = bfround(acc[3]);
                    p.y[n + 1] = f2bf(bfround(silu_f(g1)) * u1);
                }
            }
        } else {
            const int n = g * 2;
            const int n1 = (n + 1 < p.N) ? n + 1 : n;
            const bf16_t* const rows[2] = {p.W + (int64_t)n * p.K, p.W + (int64_t)n1 * p.K};
            float acc[2];
            wave_rows_dot<2>(rows, sx, p.K, lane, acc);
            if (lane < 2 && n + lane < p.N) {
                const int nn = n + lane;
                float v = lane == 0 ? acc[0] : acc[1];
                if (p.bias != nullptr) v += bf2f(p.bias[nn]);
                if (p.y_f32 != nullptr) p.y_f32[nn] = v;
                if (p.y != nullptr) {
                    if (p.residual != nullptr) v = bfround(v) + bf2f(p.residual[nn]);
                    p.y[nn] = f2bf(v);
                }
            }
        }
    }
}

int launch_gemv(const GemvArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.K % 8 == 0 && a.K > 0 && a.N > 0, "gemv: K=%d must be a positive multiple of 8", a.K);
    VILA_REQUIRE((uintptr_t)a.W % 16 == 0 && (uintptr_t)a.x % 16 == 0, "gemv: pointer alignment");
    const int n_groups = cdiv(a.N, 2);
    int grid = cdiv(n_groups, 4);
    if (grid > 2048) grid = 2048;
    const size_t lds = ((size_t)a.K * 2 + 15) / 16 * 16 + 16;
    if (a.mode == 1) {
        VILA_REQUIRE(a.W2 != nullptr && a.y != nullptr, "gemv: gate/up mode needs W2 and bf16 y");
        hipLaunchKernelGGL(gemv_kernel<1>, dim3(grid), dim3(256), lds, s, a, n_groups);
    } else {
        hipLaunchKernelGGL(gemv_kernel<0>, dim3(grid), dim3(256), lds, s, a, n_groups);
    }
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fused RMSNorm + QKV projection + bias + RoPE + KV-cache append for one new token.
// Group = 4 rows: q/k heads -> rows {d, d+1, d+hd/2, d+hd/2+1} of one head (the two rotate-half pairs),
// v heads -> 4 consecutive rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qkv_decode_kernel(QkvDecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sx = (bf16_t*)smem;
    float* scratch = (float*)(smem + ((p.K * 2 + 15) & ~15));
    stage_x(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = p.hd, half = hd >> 1;
    const int gph = hd >> 2;                                   // groups per head
    const int n_groups = (p.nq + 2 * p.nkv) * gph;
    const int pos = *p.pos_ptr;
    for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        const int head = g / gph, gi = g % gph;
        const bool is_v = head >= p.nq + p.nkv;
        int rows_i[4];
        if (is_v) {
            for (int r = 0; r < 4; ++r) rows_i[r] = head * hd + gi * 4 + r;
        } else {
            const int d = gi * 2;
            rows_i[0] = head * hd + d; rows_i[1] = head * hd + d + 1;
            rows_i[2] = head * hd + d + half; rows_i[3] = head * hd + d + half + 1;
        }
        const bf16_t* const rows[4] = {p.Wqkv + (int64_t)rows_i[0] * p.K, p.Wqkv + (int64_t)rows_i[1] * p.K,
                                       p.Wqkv + (int64_t)rows_i[2] * p.K, p.Wqkv + (int64_t)rows_i[3] * p.K};
        float acc[4];
        wave_rows_dot<4>(rows, sx, p.K, lane, acc);
        if (lane == 0) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = bfround(acc[r] + (p.bqkv != nullptr ? bf2f(p.bqkv[rows_i[r]]) : 0.f));
            if (!is_v) {
                const int d = gi * 2;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float inv = 1.0f / powf(p.theta, (float)(2 * (d + e)) / (float)hd);
                    const float ang = (float)pos * inv;
                    const float c = bfround(cosf(ang)), sn = bfround(sinf(ang));
                    const float lo = v[e], hi = v[2 + e];
                    v[e] = bfround(bfround(lo * c) + bfround(-hi * sn));
                    v[2 + e] = bfround(bfround(hi * c) + bfround(lo * sn));
                }
            }
            if (head < p.nq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.q_out[rows_i[r]] = f2bf(v[r]);
            } else if (pos < p.max_ctx) {
                const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
                bf16_t* dst = (is_v ? p.vcache : p.kcache) + ((int64_t)kvh * p.max_ctx + pos) * hd;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[rows_i[r] - head * hd] = f2bf(v[r]);
            }
        }
    }
}

int launch_qkv_decode(const QkvDecodeArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.K % 8 == 0 && a.hd % 4 == 0, "qkv_decode: K=%d hd=%d", a.K, a.hd);
    const int n_groups = (a.nq + 2 * a.nkv) * (a.hd / 4);
    const size_t lds = ((size_t)a.K * 2 + 15) / 16 * 16 + 16;
    hipLaunchKernelGGL(qkv_decode_kernel, dim3(cdiv(n_groups, 4)), dim3(256), lds, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// split-KV decode attention (q_len = 1, GQA): grid (n_splits, nkv); block handles G = nq/nkv query heads over
// a 64-key slice of the cache, writes un-normalised partial O and (m, l); a second kernel merges the splits.
// hd must be 128 (16 lanes x 8 elements per key row).
// ------------------------------------------------------------------------------------------------
#define DEC_KS 64
#define DEC_MAXG 8
__global__ __launch_bounds__(256) void attn_decode_partial(AttnDecodeArgs p) {
    __shared__ float sq[DEC_MAXG][128];
    __shared__ float sc[DEC_MAXG][DEC_KS];
    __shared__ float so[4][DEC_MAXG][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    const int nkeys = *p.pos_ptr + 1;
    const int k0 = split * DEC_KS;
    if (k0 >= nkeys) return;
    const int kn = (nkeys - k0) < DEC_KS ? (nkeys - k0) : DEC_KS;
    const bf16_t* kb = p.kcache + ((int64_t)kvh * p.max_ctx + k0) * 128;
    const bf16_t* vb = p.vcache + ((int64_t)kvh * p.max_ctx + k0) * 128;

    // issue all K and V loads up front: thread -> (key = tid/16 + 16 i, chunk = tid%16)
    const int ch = tid & 15, kr = tid >> 4;
    u32x4 kv_[4], vv_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = kr + 16 * i;
        kv_[i] = (u32x4){0u, 0u, 0u, 0u}; vv_[i] = (u32x4){0u, 0u, 0u, 0u};
        if (key < kn) { kv_[i] = *(const u32x4*)(kb + key * 128 + ch * 8); vv_[i] = *(const u32x4*)(vb + key * 128 + ch * 8); }
    }
    for (int i = tid; i < G * 128; i += 256) sq[i >> 7][i & 127] = bf2f(p.q[(kvh * G + (i >> 7)) * 128 + (i & 127)]);
    __syncthreads();

    // scores: 16 lanes cooperate on one key
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = kr + 16 * i;
        for (int g = 0; g < G; ++g) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a = fmaf(lo_bf(kv_[i][k]), sq[g][ch * 8 + 2 * k], a);
                a = fmaf(hi_bf(kv_[i][k]), sq[g][ch * 8 + 2 * k + 1], a);
            }
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
            if (ch == 0) sc[g][key] = key < kn ? a * p.scale : -INFINITY;
        }
    }
    __syncthreads();

    // softmax statistics per head: wave w handles heads w, w+4; lane = key
    for (int g = wave; g < G; g += 4) {
        const float s = sc[g][lane];
        const float m = wave_max(s);
        const float e = __expf(s - m);
        const float l = wave_sum(e);
        sc[g][lane] = e;
        if (lane == 0) {
            float* ml = p.part_ml + ((int64_t)split * p.nq + kvh * G + g) * 2;
            ml[0] = m; ml[1] = l;
        }
    }
    __syncthreads();

    // partial O: thread accumulates its 4 keys x 8 d for every head, then reduce over the 16 key rows
    for (int g = 0; g < G; ++g) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pr = sc[g][kr + 16 * i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = fmaf(pr, lo_bf(vv_[i][k]), o[2 * k]);
                o[2 * k + 1] = fmaf(pr, hi_bf(vv_[i][k]), o[2 * k + 1]);
            }
        }
        // lanes with equal ch inside a wave differ in kr by 1,2,3 -> xor 16, 32
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[k] += __shfl_xor(o[k], 16, 64); o[k] += __shfl_xor(o[k], 32, 64); }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) so[wave][g][ch * 8 + k] = o[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < G * 128; i += 256) {
        const int g = i >> 7, d = i & 127;
        p.part_o[((int64_t)split * p.nq + kvh * G + g) * 128 + d] = so[0][g][d] + so[1][g][d] + so[2][g][d] + so[3][g][d];
    }
}

__global__ __launch_bounds__(128) void attn_decode_merge(AttnDecodeArgs p) {
    const int h = blockIdx.x, d = threadIdx.x;
    const int nkeys = *p.pos_ptr + 1;
    const int ns = (nkeys + DEC_KS - 1) / DEC_KS;
    float M = -INFINITY;
    for (int s = 0; s < ns; ++s) M = fmaxf(M, p.part_ml[((int64_t)s * p.nq + h) * 2]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float* ml = p.part_ml + ((int64_t)s * p.nq + h) * 2;
        const float w = __expf(ml[0] - M);
        L += w * ml[1];
        o += w * p.part_o[((int64_t)s * p.nq + h) * 128 + d];
    }
    p.o[h * 128 + d] = f2bf(o / L);
}

int launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.hd == 128, "attn_decode: head_dim must be 128 (got %d)", a.hd);
    VILA_REQUIRE(a.nq % a.nkv == 0 && a.nq / a.nkv <= DEC_MAXG, "attn_decode: GQA group %d/%d unsupported (max %d)", a.nq, a.nkv, DEC_MAXG);
    VILA_REQUIRE(a.n_splits * DEC_KS >= a.max_ctx, "attn_decode: n_splits too small for max_ctx");
    hipLaunchKernelGGL(attn_decode_partial, dim3(a.n_splits, a.nkv), dim3(256), 0, s, a);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_decode_merge, dim3(a.nq), dim3(128), 0, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ void embed_token_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ tok, bf16_t* __restrict__ out, int H, int64_t vocab) {
    int64_t id = *tok;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < (H >> 3); c += gridDim.x * blockDim.x)
        *(u32x4*)(out + c * 8) = *(const u32x4*)(table + id * H + c * 8);
}
int launch_embed_token(const bf16_t* table, const int64_t* tok, bf16_t* out, int H, int64_t vocab, hipStream_t s) {
    hipLaunchKernelGGL(embed_token_kernel, dim3(cdiv(H / 8, 256)), dim3(256), 0, s, table, tok, out, H, vocab);
    VILA_LAUNCH_CHECK();
    return 0;
}
__global__ void decode_advance_kernel(int32_t* pos, const int64_t* tok, int64_t* out_ids, int32_t* n_out, int max_out) {
    const int n = *n_out;
    if (n < max_out) out_ids[n] = *tok;
    *n_out = n + 1;
    *pos = *pos + 1;
}
int launch_decode_advance(int32_t* pos, const int64_t* tok, int64_t* out_ids, int32_t* n_out, int max_out, hipStream_t s) {
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(1), 0, s, pos, tok, out_ids, n_out, max_out);
    VILA_LAUNCH_CHECK();
    return 0;
}
