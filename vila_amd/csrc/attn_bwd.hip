// Flash-attention backward for the same two shapes as attn.hip (hd 72 non-causal, hd 128 causal GQA
// varlen).  Three launches, no atomics:
//   delta[h][t] = sum_d dO*O
//   dQ kernel  : block = 64 query rows (4 waves x 16), loops over KV tiles — mirror image of the forward kernel:
//                S^T = K Q^T, dP^T = V dO^T (A operands K, V row-major from LDS; B operands Q, dO fragments from HBM),
//                dS^T = P^T o (dP^T - delta) * scale stays in the C layout = B layout of  dQ^T = K^T . dS^T  (A = K^T from LDS)
//   dK/dV kernel: block = 64 keys of one kv head (wave = 16 keys), loops over the G query heads of the group and 32-row
//                query tiles: S = Q K^T, dP = dO V^T (A = Q, dO row-major from LDS; B = K, V fragments held in registers),
//                P, dS in the C layout = B layout of  dV^T = dO^T . P,  dK^T = Q^T . dS  (A = transposed tiles from LDS)
// P is recomputed from the saved log-sum-exp (natural log, fp32), as flash-attn does.
#include <stdlib.h>
#include "kernels.h"
#include "train.h"

#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnBwdArgs p) {
    const int cpr = p.head_dim >> 3;                 // 16-B chunks per row (<= 16)
    const int grp = threadIdx.x >> 4, ln = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + grp;    // row = t * Hq + h
    if (row >= (int64_t)p.total_tokens * p.n_q_heads) return;
    const int t = (int)(row / p.n_q_heads), h = (int)(row % p.n_q_heads);
    float acc = 0.f;
    if (ln < cpr) {
        const u32x4 a = *(const u32x4*)(p.o + (int64_t)t * p.o_tok_stride + h * p.o_head_stride + ln * 8);
        const u32x4 b = *(const u32x4*)(p.d_o + (int64_t)t * p.do_tok_stride + h * p.do_head_stride + ln * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += lo_bf(a[k]) * lo_bf(b[k]) + hi_bf(a[k]) * hi_bf(b[k]);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
    if (ln == 0) p.delta[(int64_t)h * p.total_tokens + t] = acc;
}

// ------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    constexpr int KK = (HD + 31) / 32, HDP = KK * 32, DN = (HD + 15) / 16, CH = HD / 8;
    constexpr int KSTR = HDP + 8, VSTR = 72, KT = 64, BQ = 64;
    constexpr int K_ITERS = (KT * CH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sK = (bf16_t*)smem;                 // [64][KSTR]
    bf16_t* sV = sK + KT * KSTR;                // [64][KSTR]
    bf16_t* sKt = sV + KT * KSTR;               // [DN*16][VSTR]  K^T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, seq = blockIdx.z;
    const int kvh = h / (p.n_q_heads / p.n_kv_heads);
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int qb0 = blockIdx.x * BQ;
    if (qb0 >= seqlen) return;

    if constexpr (HDP > HD) {
        for (int i = tid; i < 2 * KT * (HDP - HD); i += 256) {
            const int row = i / (HDP - HD), c = i % (HDP - HD);
            sK[row * KSTR + HD + c] = 0;           // rows 0..127 cover sK and sV (contiguous)
        }
    }
    if constexpr (DN * 16 > HD) {
        for (int i = tid; i < (DN * 16 - HD) * VSTR; i += 256) sKt[HD * VSTR + i] = 0;
    }

    const int qw0 = qb0 + wave * 16;
    const int qrow = qw0 + l15;
    const bool qok = qrow < seqlen;
    bf16x8 qf[KK], dof[KK];
    {
        const int qr = qok ? qrow : seqlen - 1;
        const bf16_t* qp = p.q + (int64_t)(tok0 + qr) * p.q_tok_stride + h * p.q_head_stride;
        const bf16_t* dp = p.d_o + (int64_t)(tok0 + qr) * p.do_tok_stride + h * p.do_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int d = kk * 32 + lg * 8;
            u32x4 a = (u32x4){0u, 0u, 0u, 0u}, b = (u32x4){0u, 0u, 0u, 0u};
            if (d < HD && qok) { a = *(const u32x4*)(qp + d); b = *(const u32x4*)(dp + d); }
            qf[kk] = __builtin_bit_cast(bf16x8, a);
            dof[kk] = __builtin_bit_cast(bf16x8, b);
        }
    }
    const float lse2 = qok ? p.lse[(int64_t)h * p.total_tokens + tok0 + qrow] * LOG2E : 0.f;
    const float dl = qok ? p.delta[(int64_t)h * p.total_tokens + tok0 + qrow] : 0.f;
    const float c = p.scale * LOG2E;

    f32x4 dq[DN];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) dq[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kv_end = seqlen;
    if (CAUSAL) { const int lim = qb0 + BQ; kv_end = lim < seqlen ? lim : seqlen; }
    const int ntiles = (kv_end + KT - 1) / KT;
    const bf16_t* kbase = p.k + (int64_t)tok0 * p.k_tok_stride + kvh * p.k_head_stride;
    const bf16_t* vbase = p.v + (int64_t)tok0 * p.v_tok_stride + kvh * p.v_head_stride;

    u32x4 rk[K_ITERS], rv[K_ITERS], rt[4];
    const int t_dc = tid >> 4, t_kg = tid & 15;
    auto gload = [&](int t) {
        const int key0 = t * KT;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int cidx = tid + 256 * i;
            const int key = cidx / CH, ch = cidx % CH;
            rk[i] = (u32x4){0u, 0u, 0u, 0u}; rv[i] = (u32x4){0u, 0u, 0u, 0u};
            if (cidx < KT * CH && key0 + key < seqlen) {
                rk[i] = *(const u32x4*)(kbase + (int64_t)(key0 + key) * p.k_tok_stride + ch * 8);
                rv[i] = *(const u32x4*)(vbase + (int64_t)(key0 + key) * p.v_tok_stride + ch * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = key0 + t_kg * 4 + i;
            rt[i] = (u32x4){0u, 0u, 0u, 0u};
            if (t_dc < CH && key < seqlen) rt[i] = *(const u32x4*)(kbase + (int64_t)key * p.k_tok_stride + t_dc * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int cidx = tid + 256 * i;
            const int key = cidx / CH, ch = cidx % CH;
            if (cidx < KT * CH) { *(u32x4*)(sK + key * KSTR + ch * 8) = rk[i]; *(u32x4*)(sV + key * KSTR + ch * 8) = rv[i]; }
        }
        if (t_dc < CH) {
            bf16_t* dT = sKt + (t_dc * 8) * VSTR + t_kg * 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int w = e >> 1;
                u32x2 o;
                if (e & 1) { o[0] = (rt[0][w] >> 16) | (rt[1][w] & 0xffff0000u); o[1] = (rt[2][w] >> 16) | (rt[3][w] & 0xffff0000u); }
                else { o[0] = (rt[0][w] & 0xffffu) | (rt[1][w] << 16); o[1] = (rt[2][w] & 0xffffu) | (rt[3][w] << 16); }
                *(u32x2*)(dT + e * VSTR) = o;
            }
        }
    };

    gload(0);
    __syncthreads();
    lstore();
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gload(t + 1);
        const int key0 = t * KT;
        f32x4 sacc[4], pacc[4];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) { sacc[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; pacc[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const bf16x8 kf = *(const bf16x8*)(sK + (jn * 16 + l15) * KSTR + kk * 32 + lg * 8);
                const bf16x8 vf = *(const bf16x8*)(sV + (jn * 16 + l15) * KSTR + kk * 32 + lg * 8);
                sacc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], sacc[jn], 0, 0, 0);
                pacc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[kk], pacc[jn], 0, 0, 0);
            }
        // dS^T = exp(s*scale - lse) * (dP^T - delta) * scale   (masked entries -> 0)
        bf16x8 dsf[2];
        {
            float ds[4][4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kpos = key0 + jn * 16 + lg * 4 + r;
                    const bool ok = (kpos < seqlen) && (!CAUSAL || kpos <= qrow) && qok;
                    const float pr = ok ? __builtin_amdgcn_exp2f(sacc[jn][r] * c - lse2) : 0.f;
                    ds[jn][r] = pr * (pacc[jn][r] - dl) * p.scale;
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w[0] = pack2bf(ds[2 * ks][0], ds[2 * ks][1]); w[1] = pack2bf(ds[2 * ks][2], ds[2 * ks][3]);
                w[2] = pack2bf(ds[2 * ks + 1][0], ds[2 * ks + 1][1]); w[3] = pack2bf(ds[2 * ks + 1][2], ds[2 * ks + 1][3]);
                dsf[ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
#pragma unroll
        for (int dn = 0; dn < DN; ++dn)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16_t* kp = sKt + (dn * 16 + l15) * VSTR + ks * 32 + lg * 4;
                const u32x2 lo = *(const u32x2*)kp, hi = *(const u32x2*)(kp + 16);
                const u32x4 w = (u32x4){lo[0], lo[1], hi[0], hi[1]};
                dq[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), dsf[ks], dq[dn], 0, 0, 0);
            }
        __syncthreads();
        if (t + 1 < ntiles) lstore();
        __syncthreads();
    }
    if (qok) {
        bf16_t* op = p.dq + (int64_t)(tok0 + qrow) * p.dq_tok_stride + h * p.dq_head_stride;
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            const int d = dn * 16 + lg * 4;
            if (d < HD) {
                u32x2 o; o[0] = pack2bf(dq[dn][0], dq[dn][1]); o[1] = pack2bf(dq[dn][2], dq[dn][3]);
                *(u32x2*)(op + d) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV
// ------------------------------------------------------------------------------------------------
// 8 waves = 2 groups x 4: both groups hold the SAME 64 keys (wave w and w+4 the same 16) and split the (query head, query tile)
// sequence by parity, each with its own LDS stages; their dK / dV partial sums meet in LDS at the end.  Two waves per SIMD hide
// the LDS-read latency that one wave per SIMD exposed (the grid is only 208 blocks at 4 x 769 tokens: one block per CU), and the
// longest (causal, first key tile) block walks half as many tiles.  NG = 2 is used when the grid is at most one block per CU
// (LLM: 521 -> 362 us per layer); large grids (ViT: 1024 blocks) keep NG = 1 with two 4-wave blocks per CU (181 us vs 229).
template <int HD, bool CAUSAL, int NG>
__global__ __launch_bounds__(256 * NG, NG == 1 ? 2 : 1) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
    constexpr int KK = (HD + 31) / 32, HDP = KK * 32, DN = (HD + 15) / 16, CH = HD / 8;
    constexpr int KSTR = HDP + 8, QSTR = 40, QT = 32, KT = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two stages of {Q [32][KSTR], dO [32][KSTR], Q^T [DN*16][QSTR], dO^T [DN*16][QSTR]} + the rows' lse / delta: the next query
    // tile is written into the other stage while this one is consumed -> ONE barrier per tile, and no global load (lse, delta
    // used to be fetched right where exp2 needs them: a full memory round trip per tile) sits on the critical path
    constexpr int STAGE = 2 * QT * KSTR + 2 * DN * 16 * QSTR;       // bf16 elements
    const int grp = (NG == 2) ? (threadIdx.x >> 8) : 0;             // group 0 / 1
    bf16_t* const stage0 = (bf16_t*)smem + grp * 2 * STAGE;         // this group's two stages
    float* const s_ld = (float*)((bf16_t*)smem + 2 * NG * STAGE) + grp * 128;   // [2 stages][lse*log2e | delta][32] per group

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // tid / wave: within the group
    const int l15 = lane & 15, lg = lane >> 4;
    const int kvh = blockIdx.y, seq = blockIdx.z;
    const int G = p.n_q_heads / p.n_kv_heads;
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int kb0 = blockIdx.x * KT;
    if (kb0 >= seqlen) return;                                       // block-uniform

#pragma unroll
    for (int b = 0; b < 2; ++b) {                                    // each group pads its own stages
        bf16_t* sQ = stage0 + b * STAGE;
        bf16_t* sQt = sQ + 2 * QT * KSTR;
        bf16_t* sdOt = sQt + DN * 16 * QSTR;
        if constexpr (HDP > HD) {
            for (int i = tid; i < 2 * QT * (HDP - HD); i += 256) {
                const int row = i / (HDP - HD), c = i % (HDP - HD);
                sQ[row * KSTR + HD + c] = 0;           // sQ and sdO are contiguous
            }
        }
        if constexpr (DN * 16 > HD) {
            for (int i = tid; i < (DN * 16 - HD) * QSTR; i += 256) { sQt[HD * QSTR + i] = 0; sdOt[HD * QSTR + i] = 0; }
        }
    }

    // this wave's 16 keys as B operands (held in registers for the whole block)
    const int key = kb0 + wave * 16 + l15;
    const bool kok = key < seqlen;
    bf16x8 kf[KK], vf[KK];
    {
        const int kr = kok ? key : seqlen - 1;
        const bf16_t* kp = p.k + (int64_t)(tok0 + kr) * p.k_tok_stride + kvh * p.k_head_stride;
        const bf16_t* vp = p.v + (int64_t)(tok0 + kr) * p.v_tok_stride + kvh * p.v_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int d = kk * 32 + lg * 8;
            u32x4 a = (u32x4){0u, 0u, 0u, 0u}, b = (u32x4){0u, 0u, 0u, 0u};
            if (d < HD && kok) { a = *(const u32x4*)(kp + d); b = *(const u32x4*)(vp + d); }
            kf[kk] = __builtin_bit_cast(bf16x8, a);
            vf[kk] = __builtin_bit_cast(bf16x8, b);
        }
    }
    f32x4 dk[DN], dv[DN];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) { dk[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float c = p.scale * LOG2E;

    const int q_begin = CAUSAL ? (kb0 / QT) * QT : 0;
    const int ntq = (seqlen - q_begin + QT - 1) / QT;
    const int total_iters = G * ntq;                                 // (head, query tile) pairs; this group takes it = NG j + grp
    const int n_steps = (total_iters + NG - 1) / NG;                 // steps of group 0 (>= group 1's): both groups cross every barrier

    // staging: threads 0..127 -> Q, 128..255 -> dO ; work item = (q group of 4 rows, 16-B d chunk)
    const int op_sel = tid >> 7, wi = tid & 127;
    const int s_qg = wi & 7, s_dc = wi >> 3;
    // two register sets: the tile a set holds was requested TWO iterations before it is written to LDS (one iteration of
    // compute does not cover a memory round trip: with a single set every tile waited ~3 us for its loads)
    struct TileRegs { u32x4 rs[4]; float ldv; };
    TileRegs setA, setB;
    setA.ldv = 0.f; setB.ldv = 0.f;
    auto gload = [&](int it, TileRegs& T) {
        u32x4 (&rs)[4] = T.rs;
        float& ldv = T.ldv;
        const int g = it / ntq, qt = it % ntq;
        const int hq = kvh * G + g;
        const int q0 = q_begin + qt * QT;
        if (tid < 64) {                                 // lanes 0..31: lse * log2(e), lanes 32..63: delta, of the tile's 32 rows
            const int q = q0 + (tid & 31);
            ldv = 0.f;
            if (q < seqlen) {
                const int64_t idx = (int64_t)hq * p.total_tokens + tok0 + q;
                ldv = (tid < 32) ? p.lse[idx] * LOG2E : p.delta[idx];
            }
        }
        const bf16_t* base = op_sel ? (p.d_o + (int64_t)tok0 * p.do_tok_stride + hq * p.do_head_stride)
                                    : (p.q + (int64_t)tok0 * p.q_tok_stride + hq * p.q_head_stride);
        const int64_t ts = op_sel ? p.do_tok_stride : p.q_tok_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = q0 + s_qg * 4 + i;
            rs[i] = (u32x4){0u, 0u, 0u, 0u};
            if (s_dc < CH && q < seqlen) rs[i] = *(const u32x4*)(base + (int64_t)q * ts + s_dc * 8);
        }
    };
    auto lstore = [&](int b, const TileRegs& T) {
        const u32x4 (&rs)[4] = T.rs;
        const float ldv = T.ldv;
        bf16_t* sQ = stage0 + b * STAGE;
        bf16_t* sdO = sQ + QT * KSTR;
        bf16_t* sQt = sdO + QT * KSTR;
        bf16_t* sdOt = sQt + DN * 16 * QSTR;
        if (tid < 64) s_ld[b * 64 + tid] = ldv;
        if (s_dc >= CH) return;
        bf16_t* rm = (op_sel ? sdO : sQ) + (s_qg * 4) * KSTR + s_dc * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(rm + i * KSTR) = rs[i];
        bf16_t* tp = (op_sel ? sdOt : sQt) + (s_dc * 8) * QSTR + s_qg * 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int w = e >> 1;
            u32x2 o;
            if (e & 1) { o[0] = (rs[0][w] >> 16) | (rs[1][w] & 0xffff0000u); o[1] = (rs[2][w] >> 16) | (rs[3][w] & 0xffff0000u); }
            else { o[0] = (rs[0][w] & 0xffffu) | (rs[1][w] << 16); o[1] = (rs[2][w] & 0xffffu) | (rs[3][w] << 16); }
            *(u32x2*)(tp + e * QSTR) = o;
        }
    };

    if (grp < total_iters) gload(grp, setA);
    __syncthreads();                                    // padding zeros written
    if (grp < total_iters) lstore(0, setA);
    if (grp + NG < total_iters) gload(grp + NG, setA);
    if (grp + 2 * NG < total_iters) gload(grp + 2 * NG, setB);
    __syncthreads();
    auto tile_step = [&](int j, TileRegs& T) {
        const int it = NG * j + grp;
        if (it >= total_iters) { __syncthreads(); return; }          // group-uniform: keep the barrier count equal
        const int qt = it % ntq;
        const int q0 = q_begin + qt * QT;
        const int b = j & 1;
        const bf16_t* sQ = stage0 + b * STAGE;
        const bf16_t* sdO = sQ + QT * KSTR;
        const bf16_t* sQt = sdO + QT * KSTR;
        const bf16_t* sdOt = sQt + DN * 16 * QSTR;
        const float* lse2 = s_ld + b * 64;
        const float* dlt = lse2 + 32;
        // S = Q K^T, dP = dO V^T  : C layout col = key (l15), rows = q (lg*4 + r) per 16-row q fragment
        f32x4 sacc[2], pacc[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) { sacc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; pacc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 qa = *(const bf16x8*)(sQ + (f * 16 + l15) * KSTR + kk * 32 + lg * 8);
                const bf16x8 da = *(const bf16x8*)(sdO + (f * 16 + l15) * KSTR + kk * 32 + lg * 8);
                sacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kk], sacc[f], 0, 0, 0);
                pacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kk], pacc[f], 0, 0, 0);
            }
        u32x4 pw, dw;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + f * 16 + lg * 4 + r;
                const bool ok = kok && (q < seqlen) && (!CAUSAL || key <= q);
                const float l2 = lse2[f * 16 + lg * 4 + r], dl = dlt[f * 16 + lg * 4 + r];
                pr[r] = ok ? __builtin_amdgcn_exp2f(sacc[f][r] * c - l2) : 0.f;
                ds[r] = pr[r] * (pacc[f][r] - dl) * p.scale;
            }
            pw[2 * f] = pack2bf(pr[0], pr[1]); pw[2 * f + 1] = pack2bf(pr[2], pr[3]);
            dw[2 * f] = pack2bf(ds[0], ds[1]); dw[2 * f + 1] = pack2bf(ds[2], ds[3]);
        }
        const bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw), dsfrag = __builtin_bit_cast(bf16x8, dw);
        // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]   (k slots: j<4 -> q = lg*4+j, j>=4 -> 16+lg*4+j-4)
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            const bf16_t* ap = sdOt + (dn * 16 + l15) * QSTR + lg * 4;
            const u32x2 lo = *(const u32x2*)ap, hi = *(const u32x2*)(ap + 16);
            const u32x4 wa = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            dv[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa), pfrag, dv[dn], 0, 0, 0);
            const bf16_t* bp = sQt + (dn * 16 + l15) * QSTR + lg * 4;
            const u32x2 lo2 = *(const u32x2*)bp, hi2 = *(const u32x2*)(bp + 16);
            const u32x4 wb = (u32x4){lo2[0], lo2[1], hi2[0], hi2[1]};
            dk[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb), dsfrag, dk[dn], 0, 0, 0);
        }
        if (it + NG < total_iters) lstore(b ^ 1, T);     // this group's next tile (requested two steps ago) -> the other stage
        if (it + 3 * NG < total_iters) gload(it + 3 * NG, T);
        __syncthreads();                                // stage b^1 complete for the next step; everyone is done reading stage b
    };
    for (int j = 0; j < n_steps; j += 2) {
        tile_step(j, setA);                             // even steps hand over set A, odd steps set B
        if (j + 1 < n_steps) tile_step(j + 1, setB);
    }
    // ---- the two groups' partial dK / dV meet in LDS (the stages are free now): group 1 writes, group 0 adds and stores ----
    if constexpr (NG == 2) {
        float* red = (float*)smem + (size_t)wave * DN * 64 * 8;      // per key-wave: [DN][64 lanes][8 floats]
        if (grp == 1) {
#pragma unroll
            for (int dn = 0; dn < DN; ++dn) {
                *(f32x4*)(red + (dn * 64 + lane) * 8) = dk[dn];
                *(f32x4*)(red + (dn * 64 + lane) * 8 + 4) = dv[dn];
            }
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            const f32x4 a = *(const f32x4*)(red + (dn * 64 + lane) * 8), b2 = *(const f32x4*)(red + (dn * 64 + lane) * 8 + 4);
            dk[dn][0] += a[0]; dk[dn][1] += a[1]; dk[dn][2] += a[2]; dk[dn][3] += a[3];
            dv[dn][0] += b2[0]; dv[dn][1] += b2[1]; dv[dn][2] += b2[2]; dv[dn][3] += b2[3];
        }
    }
    if (kok) {
        bf16_t* kp = p.dk + (int64_t)(tok0 + key) * p.dk_tok_stride + kvh * p.dk_head_stride;
        bf16_t* vp = p.dv + (int64_t)(tok0 + key) * p.dv_tok_stride + kvh * p.dv_head_stride;
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            const int d = dn * 16 + lg * 4;
            if (d < HD) {
                u32x2 o; o[0] = pack2bf(dk[dn][0], dk[dn][1]); o[1] = pack2bf(dk[dn][2], dk[dn][3]);
                *(u32x2*)(kp + d) = o;
                u32x2 o2; o2[0] = pack2bf(dv[dn][0], dv[dn][1]); o2[1] = pack2bf(dv[dn][2], dv[dn][3]);
                *(u32x2*)(vp + d) = o2;
            }
        }
    }
}

// parts: 1 = delta (rowsum(dO o O), needed by both others), 2 = dQ, 4 = dK / dV.  The three launches share no output, so a caller may put
// dQ and dK / dV on different streams once delta is done (vila_attn_bwd_bf16_parts)
template <int HD, bool CAUSAL>
static int launch_bwd_t(const AttnBwdArgs& a, hipStream_t s, int parts) {
    constexpr int KK = (HD + 31) / 32, HDP = KK * 32, DN = (HD + 15) / 16;
    const size_t lds_dq = (size_t)(2 * 64 * (HDP + 8) + DN * 16 * 72) * 2;
    const size_t lds_stage = (size_t)(2 * 32 * (HDP + 8) + 2 * DN * 16 * 40) * 2;
    const size_t lds_kv1 = 2 * lds_stage + 2 * 64 * sizeof(float), lds_kv2 = 4 * lds_stage + 4 * 64 * sizeof(float);   // groups x 2 stages
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<HD, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        VILA_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<HD, CAUSAL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv1));
        VILA_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<HD, CAUSAL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv2));
        attr_set = true;
    }
    if (parts & 1) {
        hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(a.total_tokens * a.n_q_heads, 16)), dim3(256), 0, s, a);
        VILA_LAUNCH_CHECK();
    }
    if (parts & 2) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, CAUSAL>), dim3(cdiv(a.max_seqlen, 64), a.n_q_heads, a.n_seq), dim3(256), lds_dq, s, a);
        VILA_LAUNCH_CHECK();
    }
    if (parts & 4) {
        const dim3 grid_kv(cdiv(a.max_seqlen, 64), a.n_kv_heads, a.n_seq);
        if ((int64_t)grid_kv.x * grid_kv.y * grid_kv.z <= 320) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, CAUSAL, 2>), grid_kv, dim3(512), lds_kv2, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, CAUSAL, 1>), grid_kv, dim3(256), lds_kv1, s, a);
        VILA_LAUNCH_CHECK();
    }
    return 0;
}

int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s, int parts) {
    VILA_REQUIRE(parts >= 1 && parts <= 7, "attn_bwd: parts must be a combination of 1 (delta), 2 (dQ), 4 (dK/dV)");
    VILA_REQUIRE(a.n_seq >= 1 && a.total_tokens >= 1 && a.max_seqlen >= 1, "attn_bwd: empty input");
    VILA_REQUIRE(a.n_q_heads % a.n_kv_heads == 0, "attn_bwd: q heads must be a multiple of kv heads");
    VILA_REQUIRE(a.cu_seqlens != nullptr || (int64_t)a.n_seq * a.max_seqlen == a.total_tokens, "attn_bwd: uniform batches need total = n_seq*max_seqlen");
    VILA_REQUIRE(a.lse != nullptr && a.delta != nullptr, "attn_bwd: lse / delta workspace missing");
    // VILA_ATTN_BWD=v1 keeps the round-1/2 dQ and dK / dV kernels of this file (A/B measurements); default: the DMA-ring kernels of
    // attn_bwd_dma.hip for those two passes, delta from here
    static int impl = -1;
    if (impl < 0) { const char* e = getenv("VILA_ATTN_BWD"); impl = (e && e[0] == 'v' && e[1] == '1') ? 1 : 2; }
    if (impl == 2 && (parts & 6)) {
        if (parts & 1) {
            hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(a.total_tokens * a.n_q_heads, 16)), dim3(256), 0, s, a);
            VILA_LAUNCH_CHECK();
        }
        return launch_attn_bwd_dma(a, s, parts & 6);
    }
    if (a.head_dim == 128) return a.causal ? launch_bwd_t<128, true>(a, s, parts) : launch_bwd_t<128, false>(a, s, parts);
    if (a.head_dim == 72) return a.causal ? launch_bwd_t<72, true>(a, s, parts) : launch_bwd_t<72, false>(a, s, parts);
    if (a.head_dim == 64) return a.causal ? launch_bwd_t<64, true>(a, s, parts) : launch_bwd_t<64, false>(a, s, parts);
    VILA_FAIL(-1, "attn_bwd: unsupported head_dim %d", a.head_dim);
}
