// Flash-attention backward, round-3 kernels: the dQ and dK / dV passes of attn_bwd.hip re-staged like the round-3 forward
// (attn.hip): 8 waves per block, operand tiles by LDS-DMA into a ring (NST - 1 tiles in flight, one counted wait + one barrier per
// tile), lane-linear LDS images with the 32-B-pair swizzle — conflict-free for BOTH ways an image is read here, as rows
// (ds_read_b128: A operands of the S / dP products) and transposed (asm ds_read_b64_tr_b16: A operands of the dQ / dK / dV products) —
// so no transposed copies are built, XCD-aware 1-D grids, P / dS packed with v_cvt_pk_bf16_f32, outputs staged through LDS.
// Same math and the same three-launch structure (delta, dQ, dK / dV; no atomics) as attn_bwd.hip; the arithmetic per element is
// unchanged, so the parity tests of the round-1 kernels apply as they are.
//   dQ   : block = 128 query rows (wave = 16), loop over 64-key tiles:  S^T = K Q^T, dP^T = V dO^T (A = K, V rows; B = Q, dO fragments
//          from HBM), dS^T = P^T o (dP^T - delta) * scale in the C layout = B operand of dQ^T += K^T dS^T (A = K^T by transpose reads)
//   dK/dV: block = 64 keys of one kv head; waves = 4 key groups (16 keys, K / V fragments in registers) x 2 halves of the 64-row query
//          tile; loop over the G query heads of the group x query tiles:  S = Q K^T, dP = dO V^T (A = Q, dO rows), P / dS in the C layout
//          = B operands of dV^T += dO^T P, dK^T += Q^T dS (A = transpose reads of the SAME dO / Q images); lse / delta of the tile's rows
//          ride in the ring as two 4-B DMA slices; the two halves' sums meet in LDS at the end.
#include <stdlib.h>
#include "attn_common.h"
#include "train.h"

#define LOG2E 1.4426950408889634f

namespace {

// one A fragment (8 k-slots) of a TRANSPOSED operand: rows of the image are the contraction index.  `a` = LDS byte address of
// (row group base + this lane's row / chunk), OFF = byte offset of the 32-row k-step; second read 16 rows further
template <int OFF, int ROWB> __device__ __forceinline__ void tr_frag_issue(uint32_t a, u32x2& lo, u32x2& hi) {
    lo = ds_read_tr16_b64<OFF>(a);
    hi = ds_read_tr16_b64<OFF + 16 * ROWB>(a);
}
__device__ __forceinline__ bf16x8 tr_frag_join(const u32x2& lo, const u32x2& hi) {
    u32x4 w; w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
    return __builtin_bit_cast(bf16x8, w);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dQ
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_dma_kernel(AttnBwdArgs p, int nqb) {
    using C = AttnDma<HD>;
    constexpr int KK = C::KK, DN = C::DN, CH = C::CH, CHP = C::CHP, ROWB = C::ROWB, KT = C::KT, NST = C::NST, P = C::P, PW = C::PW;
    constexpr int BQ = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int G = p.n_q_heads / p.n_kv_heads;
    int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb_r = vid % nqb; vid /= nqb;
    const int hq = vid % G; vid /= G;
    const int kvh = vid % p.n_kv_heads;
    const int seq = vid / p.n_kv_heads;
    const int h = kvh * G + hq;
    const int qb = CAUSAL ? nqb - 1 - qb_r : qb_r;
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int qb0 = qb * BQ;
    if (qb0 >= seqlen) return;
    int kv_end = seqlen;
    if (CAUSAL) { const int lim = qb0 + BQ; kv_end = lim < seqlen ? lim : seqlen; }
    const int ntiles = (kv_end + KT - 1) / KT;
    const bf16_t* kbase = p.k + (int64_t)tok0 * p.k_tok_stride + kvh * p.k_head_stride;
    const bf16_t* vbase = p.v + (int64_t)tok0 * p.v_tok_stride + kvh * p.v_head_stride;

    // DMA plan: stage image = [K image | V image], both with the pair swizzle
    const bf16_t* dsrc[P]; int64_t dstr[P]; int drow[P]; bool dact[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int sl = i * 64 + lane;
        dact[i] = sl < PW;
        const int s = wave * PW + (dact[i] ? sl : 0);
        const int isv = s >= KT * CHP;
        const int w = s - isv * KT * CHP;
        const int row = w / CHP, pos = w % CHP;
        int c = C::swz_v(row, pos);
        if (HD == 72 && c >= CH) c = CH - 1;
        drow[i] = row;
        dsrc[i] = (isv ? vbase : kbase) + c * 8;
        dstr[i] = isv ? p.v_tok_stride : p.k_tok_stride;
    }
    auto issue_tile = [&](int t) {
        const int st = t % NST;
        const int key0 = t * KT;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int key = key0 + drow[i];
            key = key < seqlen ? key : seqlen - 1;
            char* dst = smem + st * C::STAGE + (wave * PW + i * 64) * 16;
            if (dact[i]) __builtin_amdgcn_global_load_lds((gbl_void_t*)(dsrc[i] + (int64_t)key * dstr[i]), (lds_void_t*)dst, 16, 0, 0);
        }
    };

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
    float lse2 = qok ? p.lse[(int64_t)h * p.total_tokens + tok0 + qrow] * LOG2E : 0.f;
    float dl = qok ? p.delta[(int64_t)h * p.total_tokens + tok0 + qrow] : 0.f;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < ntiles) issue_tile(t);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) { asm volatile("" : "+v"(qf[kk])); asm volatile("" : "+v"(dof[kk])); }   // arrived before the loop (see attn.hip)
    asm volatile("" : "+v"(lse2), "+v"(dl));

    int koff[KK], toff[DN];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        int c = kk * 4 + lg;
        if (HD == 72 && kk == 2) c = 8 + (lg & 1);
        koff[kk] = l15 * ROWB + C::swz_v(l15, c) * 16;
    }
    const int trow = lg * 4 + (l15 >> 2);
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const int c = dn * 2 + ((l15 & 3) >> 1);
        toff[dn] = trow * ROWB + C::swz_v(trow, c) * 16 + (l15 & 1) * 8;
    }
    f32x4 dq[DN];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) dq[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * LOG2E;
    const bool wave_has_rows = qw0 < seqlen;
    const int qw_last = qw0 + 15;

    for (int t = 0; t < ntiles; ++t) {
        wait_tiles_ahead<P, NST - 2>(ntiles - 1 - t);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int key0 = t * KT;
        const bool active = wave_has_rows && !(CAUSAL && key0 > qw_last);
        const char* cK = smem + (t % NST) * C::STAGE;
        const char* cV = cK + C::IMG;
        f32x4 sacc[4], pacc[4];
        if (active) {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) { sacc[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; pacc[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            bf16x8 kf[KK][4];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) kf[kk][jn] = *(const bf16x8*)(cK + jn * 16 * ROWB + koff[kk]);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) sacc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk][jn], qf[kk], sacc[jn], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) kf[kk][jn] = *(const bf16x8*)(cV + jn * 16 * ROWB + koff[kk]);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) pacc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk][jn], dof[kk], pacc[jn], 0, 0, 0);
        }
        if (t + NST - 1 < ntiles) issue_tile(t + NST - 1);       // behind the MFMAs of this tile (see attn.hip)
        if (!active) continue;
        const uint32_t k_lds = lds_addr(cK);
        // K^T fragments of the first k-step travel while dS is formed
        u32x2 tlo[2][DN], thi[2][DN];
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) tr_frag_issue<0, ROWB>(k_lds + toff[dn], tlo[0][dn], thi[0][dn]);
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
                    const float pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[jn][r], c, -lse2)) : 0.f;
                    ds[jn][r] = pr * (pacc[jn][r] - dl) * p.scale;
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w[0] = cvt_pk_bf16(ds[2 * ks][0], ds[2 * ks][1]); w[1] = cvt_pk_bf16(ds[2 * ks][2], ds[2 * ks][3]);
                w[2] = cvt_pk_bf16(ds[2 * ks + 1][0], ds[2 * ks + 1][1]); w[3] = cvt_pk_bf16(ds[2 * ks + 1][2], ds[2 * ks + 1][3]);
                dsf[ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        // dQ^T += K^T dS^T : second k-step requested, first retired behind a counted wait
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) tr_frag_issue<32 * ROWB, ROWB>(k_lds + toff[dn], tlo[1][dn], thi[1][dn]);
        static_assert(2 * DN <= 15 + 1, "lgkmcnt range");
        static_for<0, 2>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            // group ks is complete once at most the (ks == 0 ? 2 * DN : 0) younger reads are outstanding
            if constexpr (ks == 0) { if constexpr (2 * DN <= 15) lds_wait_imm<2 * DN>(); else lds_wait_imm<15>(); }
            else lds_wait_imm<0>();
#pragma unroll
            for (int dn = 0; dn < DN; ++dn) {
                asm volatile("" : "+v"(tlo[ks][dn]), "+v"(thi[ks][dn]));          // consumers stay behind the wait above
                dq[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_join(tlo[ks][dn], thi[ks][dn]), dsf[ks], dq[dn], 0, 0, 0);
            }
        });
    }

    // ---- store dQ[q][d] through a per-wave LDS staging area ----
    __syncthreads();
    bf16_t* so = (bf16_t*)smem + wave * 16 * C::OSTR;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const int d = dn * 16 + lg * 4;
        if (d < HD) {
            u32x2 o;
            o[0] = cvt_pk_bf16(dq[dn][0], dq[dn][1]);
            o[1] = cvt_pk_bf16(dq[dn][2], dq[dn][3]);
            *(u32x2*)(so + l15 * C::OSTR + d) = o;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (16 * CH + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CH, ch = idx % CH;
        const int qr = qw0 + row;
        if (idx < 16 * CH && qr < seqlen) {
            const u32x4 v = *(const u32x4*)(so + row * C::OSTR + ch * 8);
            *(u32x4*)(p.dq + (int64_t)(tok0 + qr) * p.dq_tok_stride + h * p.dq_head_stride + ch * 8) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dK, dV
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_dma_kernel(AttnBwdArgs p, int nkb) {
    using C = AttnDma<HD>;
    constexpr int KK = C::KK, DN = C::DN, CH = C::CH, CHP = C::CHP, ROWB = C::ROWB, NST = C::NST, P = C::P, PW = C::PW;
    constexpr int QT = 64, KT = 64;
    constexpr int STAGE = C::STAGE + 512;                   // Q image | dO image | lse[64] | delta[64] (fp32)
    constexpr int PT = P + 1;                               // DMA instructions per wave and tile (one 4-B slice of lse / delta on top)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave & 3, qh = wave >> 2;
    const int l15 = lane & 15, lg = lane >> 4;
    const int G = p.n_q_heads / p.n_kv_heads;
    int vid = xcd_remap(blockIdx.x, gridDim.x);              // consecutive ids = key tiles of one (sequence, kv head): they share Q / dO
    const int kb = vid % nkb; vid /= nkb;
    const int kvh = vid % p.n_kv_heads;
    const int seq = vid / p.n_kv_heads;
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int kb0 = kb * KT;
    if (kb0 >= seqlen) return;                               // block-uniform

    // this wave's 16 keys as B operands (registers for the whole block)
    const int key = kb0 + kg * 16 + l15;
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
    const int q_begin = CAUSAL ? (kb0 / QT) * QT : 0;
    const int ntq = (seqlen - q_begin + QT - 1) / QT;
    const int total = G * ntq;                               // (query head of the group, query tile) pairs

    // DMA plan: stage image = [Q image | dO image] (pair swizzle) + this wave's 16 floats of [lse | delta]
    int dcoff[P]; int drow[P]; bool dact[P], disdo[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int sl = i * 64 + lane;
        dact[i] = sl < PW;
        const int s = wave * PW + (dact[i] ? sl : 0);
        const int isdo = s >= QT * CHP;
        const int w = s - isdo * QT * CHP;
        const int row = w / CHP, pos = w % CHP;
        int c = C::swz_v(row, pos);
        if (HD == 72 && c >= CH) c = CH - 1;
        drow[i] = row; dcoff[i] = c * 8; disdo[i] = isdo != 0;
    }
    const float* fsrc = (wave < 4 ? p.lse : p.delta);
    const int frow = (wave & 3) * 16 + (lane & 15);
    auto issue_tile = [&](int it) {
        const int st = it % NST;
        const int g = it / ntq, qt = it - g * ntq;
        const int hq = kvh * G + g;
        const int q0 = q_begin + qt * QT;
        const bf16_t* qbase = p.q + (int64_t)tok0 * p.q_tok_stride + hq * p.q_head_stride;
        const bf16_t* dobase = p.d_o + (int64_t)tok0 * p.do_tok_stride + hq * p.do_head_stride;
        char* sbase = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int q = q0 + drow[i];
            q = q < seqlen ? q : seqlen - 1;
            const bf16_t* src = disdo[i] ? dobase + (int64_t)q * p.do_tok_stride + dcoff[i] : qbase + (int64_t)q * p.q_tok_stride + dcoff[i];
            char* dst = sbase + (wave * PW + i * 64) * 16;
            if (dact[i]) __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
        }
        {
            int q = q0 + frow;
            q = q < seqlen ? q : seqlen - 1;
            const float* src = fsrc + (int64_t)hq * p.total_tokens + tok0 + q;
            char* dst = sbase + C::STAGE + wave * 64;                // 16 floats per wave: waves 0..3 -> lse rows, 4..7 -> delta rows
            if (lane < 16) __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 4, 0, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < total) issue_tile(t);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) { asm volatile("" : "+v"(kf[kk])); asm volatile("" : "+v"(vf[kk])); }

    int aoff[KK], toff[DN];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        int c = kk * 4 + lg;
        if (HD == 72 && kk == 2) c = 8 + (lg & 1);
        aoff[kk] = l15 * ROWB + C::swz_v(l15, c) * 16;       // row (qh*32 + f*16 + l15): the swizzle class depends on l15 only
    }
    const int trow = lg * 4 + (l15 >> 2);
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const int c = dn * 2 + ((l15 & 3) >> 1);
        toff[dn] = trow * ROWB + C::swz_v(trow, c) * 16 + (l15 & 1) * 8;
    }
    f32x4 dk[DN], dv[DN];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) { dk[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float c = p.scale * LOG2E;
    const int key_lo = kb0 + kg * 16;                        // first key of this wave

    for (int it = 0; it < total; ++it) {
        wait_tiles_ahead<PT, NST - 2>(total - 1 - it);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int g = it / ntq, qt = it - g * ntq;
        const int q0 = q_begin + qt * QT + qh * 32;          // first query row of this wave's half tile
        const bool active = (q0 < seqlen) && (key_lo < seqlen) && !(CAUSAL && q0 + 31 < key_lo);
        const char* cQ = smem + (it % NST) * STAGE + qh * 32 * ROWB;
        const char* cdO = cQ + C::IMG;
        const float* s_lse = (const float*)(smem + (it % NST) * STAGE + C::STAGE) + qh * 32;
        const float* s_dl = s_lse + 64;
        // S = Q K^T, dP = dO V^T : C layout col = key (l15), rows = q (lg*4 + r) per 16-row fragment f
        f32x4 sacc[2], pacc[2];
        if (active) {
#pragma unroll
            for (int f = 0; f < 2; ++f) { sacc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; pacc[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            bf16x8 qa[KK][2], da[KK][2];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    qa[kk][f] = *(const bf16x8*)(cQ + f * 16 * ROWB + aoff[kk]);
                    da[kk][f] = *(const bf16x8*)(cdO + f * 16 * ROWB + aoff[kk]);
                }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    sacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[kk][f], kf[kk], sacc[f], 0, 0, 0);
                    pacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[kk][f], vf[kk], pacc[f], 0, 0, 0);
                }
        }
        if (it + NST - 1 < total) issue_tile(it + NST - 1);
        if (!active) continue;
        // dO^T fragments (A of dV^T += dO^T P) travel while P / dS are formed; k slots: j < 4 -> q = lg*4 + j, j >= 4 -> 16 + lg*4 + j - 4
        const uint32_t do_lds = lds_addr(cdO), q_lds = lds_addr(cQ);
        u32x2 tlo[2][DN], thi[2][DN];
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) tr_frag_issue<0, ROWB>(do_lds + toff[dn], tlo[0][dn], thi[0][dn]);
        u32x4 pw, dw;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const f32x4 l4 = *(const f32x4*)(s_lse + f * 16 + lg * 4), d4 = *(const f32x4*)(s_dl + f * 16 + lg * 4);
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + f * 16 + lg * 4 + r;
                const bool ok = kok && (q < seqlen) && (!CAUSAL || key <= q);
                pr[r] = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[f][r], c, -l4[r] * LOG2E)) : 0.f;
                ds[r] = pr[r] * (pacc[f][r] - d4[r]) * p.scale;
            }
            pw[2 * f] = cvt_pk_bf16(pr[0], pr[1]); pw[2 * f + 1] = cvt_pk_bf16(pr[2], pr[3]);
            dw[2 * f] = cvt_pk_bf16(ds[0], ds[1]); dw[2 * f + 1] = cvt_pk_bf16(ds[2], ds[3]);
        }
        const bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw), dsfrag = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) tr_frag_issue<0, ROWB>(q_lds + toff[dn], tlo[1][dn], thi[1][dn]);
        if constexpr (2 * DN <= 15) lds_wait_imm<2 * DN>(); else lds_wait_imm<15>();
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            asm volatile("" : "+v"(tlo[0][dn]), "+v"(thi[0][dn]));
            dv[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_join(tlo[0][dn], thi[0][dn]), pfrag, dv[dn], 0, 0, 0);
        }
        lds_wait_imm<0>();
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            asm volatile("" : "+v"(tlo[1][dn]), "+v"(thi[1][dn]));
            dk[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_join(tlo[1][dn], thi[1][dn]), dsfrag, dk[dn], 0, 0, 0);
        }
    }

    // ---- the two query halves' partial sums meet in LDS (the ring is free now): half 1 writes, half 0 adds and stores ----
    __syncthreads();
    float* red = (float*)smem + (size_t)kg * DN * 64 * 8;             // per key group: [DN][64 lanes][dk(4) | dv(4)]
    if (qh == 1) {
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            *(f32x4*)(red + (dn * 64 + lane) * 8) = dk[dn];
            *(f32x4*)(red + (dn * 64 + lane) * 8 + 4) = dv[dn];
        }
    }
    __syncthreads();
    if (qh == 1) return;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const f32x4 a = *(const f32x4*)(red + (dn * 64 + lane) * 8), b2 = *(const f32x4*)(red + (dn * 64 + lane) * 8 + 4);
        dk[dn][0] += a[0]; dk[dn][1] += a[1]; dk[dn][2] += a[2]; dk[dn][3] += a[3];
        dv[dn][0] += b2[0]; dv[dn][1] += b2[1]; dv[dn][2] += b2[2]; dv[dn][3] += b2[3];
    }
    // dK[key][d], dV[key][d]: C layout (rows d, col key) -> per-wave staging [16 keys][HD] -> whole 16-B row chunks
    bf16_t* sk = (bf16_t*)(smem + 4 * DN * 64 * 8 * 4) + kg * 2 * 16 * C::OSTR;
    bf16_t* sv = sk + 16 * C::OSTR;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const int d = dn * 16 + lg * 4;
        if (d < HD) {
            u32x2 o; o[0] = cvt_pk_bf16(dk[dn][0], dk[dn][1]); o[1] = cvt_pk_bf16(dk[dn][2], dk[dn][3]);
            *(u32x2*)(sk + l15 * C::OSTR + d) = o;
            u32x2 o2; o2[0] = cvt_pk_bf16(dv[dn][0], dv[dn][1]); o2[1] = cvt_pk_bf16(dv[dn][2], dv[dn][3]);
            *(u32x2*)(sv + l15 * C::OSTR + d) = o2;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (16 * CH + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CH, ch = idx % CH;
        const int kr = key_lo + row;
        if (idx < 16 * CH && kr < seqlen) {
            const u32x4 a = *(const u32x4*)(sk + row * C::OSTR + ch * 8), b2 = *(const u32x4*)(sv + row * C::OSTR + ch * 8);
            *(u32x4*)(p.dk + (int64_t)(tok0 + kr) * p.dk_tok_stride + kvh * p.dk_head_stride + ch * 8) = a;
            *(u32x4*)(p.dv + (int64_t)(tok0 + kr) * p.dv_tok_stride + kvh * p.dv_head_stride + ch * 8) = b2;
        }
    }
}

template <int HD, bool CAUSAL>
int launch_dma_t(const AttnBwdArgs& a, hipStream_t s, int parts) {
    using C = AttnDma<HD>;
    const size_t lds_dq = (size_t)C::NST * C::STAGE;
    const size_t red = (size_t)4 * C::DN * 64 * 8 * 4 + (size_t)4 * 2 * 16 * C::OSTR * 2;
    const size_t ring_kv = (size_t)C::NST * (C::STAGE + 512);
    const size_t lds_kv = ring_kv > red ? ring_kv : red;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)attn_bwd_dq_dma_kernel<HD, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
        VILA_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_dma_kernel<HD, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
        attr_set = true;
    }
    if (parts & 2) {
        const int nqb = cdiv(a.max_seqlen, 128);
        const int64_t blocks = (int64_t)nqb * a.n_q_heads * a.n_seq;
        VILA_REQUIRE(blocks < (1ll << 31), "attn_bwd: grid too large");
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<HD, CAUSAL>), dim3((unsigned)blocks), dim3(512), lds_dq, s, a, nqb);
        VILA_LAUNCH_CHECK();
    }
    if (parts & 4) {
        const int nkb = cdiv(a.max_seqlen, 64);
        const int64_t blocks = (int64_t)nkb * a.n_kv_heads * a.n_seq;
        VILA_REQUIRE(blocks < (1ll << 31), "attn_bwd: grid too large");
        hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<HD, CAUSAL>), dim3((unsigned)blocks), dim3(512), lds_kv, s, a, nkb);
        VILA_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace

// dQ (parts & 2) and dK / dV (parts & 4) with the round-3 kernels; delta (parts & 1) stays with attn_bwd.hip
int launch_attn_bwd_dma(const AttnBwdArgs& a, hipStream_t s, int parts) {
    VILA_REQUIRE((uintptr_t)a.q % 16 == 0 && (uintptr_t)a.k % 16 == 0 && (uintptr_t)a.v % 16 == 0 && (uintptr_t)a.d_o % 16 == 0 &&
                 (uintptr_t)a.dq % 16 == 0 && (uintptr_t)a.dk % 16 == 0 && (uintptr_t)a.dv % 16 == 0, "attn_bwd: pointers must be 16-B aligned");
    VILA_REQUIRE(a.q_tok_stride % 8 == 0 && a.k_tok_stride % 8 == 0 && a.v_tok_stride % 8 == 0 && a.do_tok_stride % 8 == 0 && a.dq_tok_stride % 8 == 0 &&
                 a.dk_tok_stride % 8 == 0 && a.dv_tok_stride % 8 == 0 && a.q_head_stride % 8 == 0 && a.k_head_stride % 8 == 0 && a.v_head_stride % 8 == 0 &&
                 a.do_head_stride % 8 == 0 && a.dq_head_stride % 8 == 0 && a.dk_head_stride % 8 == 0 && a.dv_head_stride % 8 == 0,
                 "attn_bwd: strides must keep 16-B alignment");
    if (a.head_dim == 128) return a.causal ? launch_dma_t<128, true>(a, s, parts) : launch_dma_t<128, false>(a, s, parts);
    if (a.head_dim == 72) return a.causal ? launch_dma_t<72, true>(a, s, parts) : launch_dma_t<72, false>(a, s, parts);
    if (a.head_dim == 64) return a.causal ? launch_dma_t<64, true>(a, s, parts) : launch_dma_t<64, false>(a, s, parts);
    VILA_FAIL(-1, "attn_bwd: unsupported head_dim %d", a.head_dim);
}
