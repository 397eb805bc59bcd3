// Flash-attention forward for the two shapes on the path:
//   SigLIP  (modeling_siglip.py:389-439 / :461-526): non-causal, 16 heads x hd 72, N = 1024 per tile
//   Qwen2   (HF qwen2 attention; in-tree mirror fp8activationqwen2.py:992-1055): causal GQA, hd 128,
//           varlen over cu_seqlens (packing.py:12-21 semantics: block-diagonal causal per sample)
//
// Layout trick (wave64 / v_mfma_f32_16x16x32_bf16): everything is computed TRANSPOSED so that no operand ever has
// to move between lanes:
//   S^T[key][q] = K . Q^T      A = K tile from LDS (row-major [key][d]),  B = Q fragments straight from HBM
//   O^T[d][q]  = V^T . P^T     A = V^T tile from LDS ([d][key], transposed while staging), B = P^T = the lane's own
//                              S^T accumulator registers (C layout of step 1 == B layout of step 2 under the k-slot
//                              permutation key(lg,j) = 16*(2ks + j/4) + 4*lg + j%4, applied to V^T reads as well)
// so the softmax statistics (per q = lane&15) and the O^T accumulator columns live in the same lane.
// Block = 4 waves x 32 query rows; KV tile = 64 keys, double-buffered in LDS, one barrier per tile.
#include <stdlib.h>
#include "kernels.h"
#include "attn_common.h"


template <int HD, int QF, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_v1_kernel(AttnArgs p) {
    constexpr int KK = (HD + 31) / 32;      // 32-wide contraction chunks for QK^T
    constexpr int HDP = KK * 32;
    constexpr int DN = (HD + 15) / 16;      // 16-row output fragments of O^T
    constexpr int CH = HD / 8;              // 16-B chunks per K/V row
    constexpr int KSTR = HDP + 8;           // sK row stride (elements): +16 B pad => conflict-free ds_read_b128
    constexpr int KT = 64;
    constexpr int BQ = 4 * QF * 16;
    constexpr int K_ITERS = (KT * CH + 255) / 256;
    static_assert(HD % 8 == 0, "head dim must be a multiple of 8");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sK = (bf16_t*)smem;                       // [2][64][KSTR]
    bf16_t* sV = sK + 2 * KT * KSTR;                  // [2][64][KSTR]  row-major like K; V^T fragments come out of ds_read_b64_tr_b16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, seq = blockIdx.z;
    const int kvh = h / (p.n_q_heads / p.n_kv_heads);
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;   // cu_seqlens == NULL: n_seq sequences of max_seqlen tokens
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int qb0 = blockIdx.x * BQ;
    if (qb0 >= seqlen) return;

    // zero the pad columns/rows once (they are never overwritten): K cols [HD,HDP), V^T rows [HD, DN*16)
    if constexpr (HDP > HD) {
        for (int i = tid; i < 4 * KT * (HDP - HD); i += 256) {     // 2 K buffers + 2 V buffers are contiguous rows
            const int row = i / (HDP - HD), c = i % (HDP - HD);
            sK[row * KSTR + HD + c] = 0;
        }
    }

    // ---- Q fragments (B operand of S^T): lane holds Q[q0 + qf*16 + l15][kk*32 + lg*8 .. +8] ----
    const int qw0 = qb0 + wave * QF * 16;   // first query row of this wave (within the sequence)
    bf16x8 qf_[QF][KK];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qrow = qw0 + f * 16 + l15;
        const bf16_t* qp = p.q + (int64_t)(tok0 + (qrow < seqlen ? qrow : seqlen - 1)) * p.q_tok_stride + h * p.q_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int d = kk * 32 + lg * 8;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (d < HD && qrow < seqlen) v = *(const u32x4*)(qp + d);
            qf_[f][kk] = __builtin_bit_cast(bf16x8, v);
        }
    }

    f32x4 oacc[DN][QF];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn)
#pragma unroll
        for (int f = 0; f < QF; ++f) oacc[dn][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) { m_run[f] = NEG_BIG; l_run[f] = 0.f; }

    const float c = p.scale * 1.4426950408889634f;   // exp(x*scale) = exp2(x*c)
    int kv_end = seqlen;
    if (CAUSAL) { const int lim = qb0 + BQ; kv_end = lim < seqlen ? lim : seqlen; }
    const int ntiles = (kv_end + KT - 1) / KT;

    const bf16_t* kbase = p.k + (int64_t)tok0 * p.k_tok_stride + kvh * p.k_head_stride;
    const bf16_t* vbase = p.v + (int64_t)tok0 * p.v_tok_stride + kvh * p.v_head_stride;

    // staging registers
    u32x4 rk[K_ITERS];
    u32x4 rv[K_ITERS];
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
    };
    auto lstore = [&](int buf) {
        bf16_t* dK = sK + buf * KT * KSTR;
        bf16_t* dVv = sV + buf * KT * KSTR;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int cidx = tid + 256 * i;
            const int key = cidx / CH, ch = cidx % CH;
            if (cidx < KT * CH) { *(u32x4*)(dK + key * KSTR + ch * 8) = rk[i]; *(u32x4*)(dVv + key * KSTR + ch * 8) = rv[i]; }
        }
    };

    gload(0);
    __syncthreads();   // pad zero-fill complete before the first tile lands next to it
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const bf16_t* cK = sK + buf * KT * KSTR;
        const bf16_t* cV = sV + buf * KT * KSTR;
        const int key0 = t * KT;

        // ---- S^T = K Q^T ----
        f32x4 sacc[4][QF];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int f = 0; f < QF; ++f) sacc[jn][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const bf16x8 kf = *(const bf16x8*)(cK + (jn * 16 + l15) * KSTR + kk * 32 + lg * 8);
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    sacc[jn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf_[f][kk], sacc[jn][f], 0, 0, 0);
            }
        }

        // ---- mask (sequence end / causal diagonal) ----
        const bool need_mask = (key0 + KT > seqlen) || (CAUSAL && (key0 + KT - 1 > qw0));
        if (need_mask) {
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const int qpos = qw0 + f * 16 + l15;
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kpos = key0 + jn * 16 + lg * 4 + r;
                        const bool ok = (kpos < seqlen) && (!CAUSAL || kpos <= qpos);
                        if (!ok) sacc[jn][f][r] = NEG_BIG;
                    }
            }
        }

        // ---- online softmax (per q = lane&15; reduce over own 16 keys, then over the 4 lane groups) ----
        bf16x8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = sacc[0][f][0];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[jn][f][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[f] - m_new) * c);   // raw v_exp_f32: arguments are <= 0, no range fix-up needed
            const float mc = m_new * c;
            m_run[f] = m_new;
            float ps = 0.f;
            float pv[4][4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[jn][f][r], c, -mc));
                    pv[jn][r] = e;
                    ps += e;
                }
            l_run[f] = l_run[f] * alpha + ps;
            if (!__all(alpha == 1.f)) {          // the running max moved for some row of this wave: rescale O (wave-uniform branch)
#pragma unroll
                for (int dn = 0; dn < DN; ++dn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[dn][f][r] *= alpha;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w[0] = pack2bf(pv[2 * ks][0], pv[2 * ks][1]);
                w[1] = pack2bf(pv[2 * ks][2], pv[2 * ks][3]);
                w[2] = pack2bf(pv[2 * ks + 1][0], pv[2 * ks + 1][1]);
                w[3] = pack2bf(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
                pf[f][ks] = __builtin_bit_cast(bf16x8, w);
            }
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // V^T fragment via the LDS transpose read: lane i of a 16-lane group points at V[key0 + (i>>2)][d0 + 4*(i&3)] and
                // receives V[key0 + 0..3][d0 + i] (semantics pinned by tools/tr16_probe.hip); key0 = ks*32 + lg*4 (+16 for slots 4..7)
                const bf16_t* vp = cV + (ks * 32 + lg * 4 + (l15 >> 2)) * KSTR + dn * 16 + (l15 & 3) * 4;
                const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)vp);
                const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vp + 16 * KSTR));
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    oacc[dn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[f][ks], oacc[dn][f], 0, 0, 0);
            }
        }

        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise and store O[q][d]; lane: q = l15, d = dn*16 + lg*4 + r ----
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        const int qrow = qw0 + f * 16 + l15;
        if (qrow < seqlen) {
            bf16_t* op = p.o + (int64_t)(tok0 + qrow) * p.o_tok_stride + h * p.o_head_stride;
#pragma unroll
            for (int dn = 0; dn < DN; ++dn) {
                const int d = dn * 16 + lg * 4;
                if (d < HD) {
                    u32x2 o;
                    o[0] = pack2bf(oacc[dn][f][0] * inv, oacc[dn][f][1] * inv);
                    o[1] = pack2bf(oacc[dn][f][2] * inv, oacc[dn][f][3] * inv);
                    *(u32x2*)(op + d) = o;
                }
            }
            if (p.lse != nullptr && lg == 0)
                p.lse[(int64_t)h * p.total_tokens + tok0 + qrow] = m_run[f] * p.scale + logf(l);
        }
    }
}

template <int HD, bool CAUSAL, int QF>
static int launch_attn_v1_q(const AttnArgs& a, hipStream_t s) {
    constexpr int KK = (HD + 31) / 32, HDP = KK * 32;
    const size_t lds = (size_t)4 * 64 * (HDP + 8) * 2;   // K and V tiles, double buffered
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)attn_fwd_v1_kernel<HD, QF, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(cdiv(a.max_seqlen, 4 * QF * 16), a.n_q_heads, a.n_seq);
    hipLaunchKernelGGL((attn_fwd_v1_kernel<HD, QF, CAUSAL>), grid, dim3(256), lds, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}

// 32 query rows per wave (QF = 2) halves the LDS traffic per MFMA; with few blocks (one 448^2 tile = 8 x 16, S = 769 prefill
// = 7 x 28) 16 rows per wave (QF = 1) doubles the grid and fills the 256 CUs
template <int HD, bool CAUSAL>
static int launch_attn_v1_t(const AttnArgs& a, hipStream_t s) {
    const int64_t blocks2 = (int64_t)cdiv(a.max_seqlen, 128) * a.n_q_heads * a.n_seq;
    if (blocks2 < 256) return launch_attn_v1_q<HD, CAUSAL, 1>(a, s);
    return launch_attn_v1_q<HD, CAUSAL, 2>(a, s);
}


// =================================================================================================================================
// Round-3 forward kernel: the same transposed MFMA formulation, re-staged.
//   * 8 waves (512 threads) per block, one block per CU, each wave 16 * QF query rows -> block = 128 * QF rows;
//   * K / V tiles of 64 keys arrive by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip, no ds_write pass) into a ring of
//     NST stages, NST - 1 tiles in flight ahead of the one being consumed: the HBM / L2 latency of a tile is hidden behind NST - 2
//     tiles of compute instead of being paid once per tile (round 2: ONE tile of look-ahead through registers, 7 us per tile in the
//     64-frame tower batch).  One counted `s_waitcnt vmcnt` + one barrier per tile;
//   * LDS images are lane-linear for the DMA; bank conflicts are removed by permuting the per-lane SOURCE chunk and applying the same
//     involution on the reads: hd 128 (256-B rows): K chunk ^= row & 15 (ds_read_b128), V 32-B pair ^= row & 7 (ds_read_b64_tr_b16);
//     hd 64: K chunk ^= (row >> 1) & 7, V pair ^= (row >> 1) & 3; hd 72: rows padded to TEN chunks (160 B: chunk 9 = a second copy
//     of chunk 8, never used) — with that stride both read patterns are conflict-free as they lie (tools: /tmp swizzle search, DESIGN);
//   * 1-D grid, XCD-aware: the blocks of one (sequence, kv head) — all query blocks of all G query heads of the group — get
//     consecutive ids inside ONE XCD's contiguous range, so a K / V tile is fetched into one L2, not eight; causal blocks are ordered
//     heaviest (last query block) first;
//   * P is packed with v_cvt_pk_bf16_f32; O is staged through LDS and stored as whole 16-B row chunks.
// =================================================================================================================================
// retire `cnt` fragments (two 8-B halves each) of a P.V group: wait until at most N younger LDS reads are outstanding, registers tied
template <int N, int cnt> __device__ __forceinline__ void pv_retire(u32x2 (&v)[4][2]) {
    if constexpr (cnt == 4) lds_wait<N>(v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1], v[3][0], v[3][1]);
    else if constexpr (cnt == 2) lds_wait<N>(v[0][0], v[0][1], v[1][0], v[1][1]);
    else if constexpr (cnt == 1) lds_wait<N>(v[0][0], v[0][1]);
    else { lds_wait<N>(v[0][0], v[0][1], v[1][0], v[1][1]); lds_wait<N>(v[2][0], v[2][1]); }
}

// KS (round 6): key split inside the block.  KS = 1: 8 waves x QF*16 query rows, every wave walks every K / V tile.  KS = 2: 4 query groups
// x 2 key groups — waves 0-3 take the even tiles, waves 4-7 the odd ones, for the SAME 4 x QF*16 rows — and the two partial (m, l, O) of a
// row meet through LDS at the end.  Why: at the path's sizes (S = 769 causal: 196 blocks; one 448^2 tile: 128 blocks) the kernel is ONE block per
// CU walking <= 13 (16) dependent tiles, and with 16 rows per wave (QF = 1, taken to keep 196 blocks) each of the 8 waves reads the whole
// 32-KB K / V stage per tile: 256 KB of LDS reads = 2048 cycles per tile for 512 cycles of MFMA.  32 rows per wave halve the LDS bytes per
// row, the key split keeps the block at 128 rows: the same grid, half the dependent steps.
template <int HD, int QF, bool CAUSAL, int KS = 1>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnArgs p, int nqb) {
    using C = AttnDma<HD>;
    constexpr int KK = C::KK, DN = C::DN, CH = C::CH, CHP = C::CHP, ROWB = C::ROWB, KT = C::KT, NST = C::NST, P = C::P, PW = C::PW;
    static_assert(KS == 1 || KS == 2, "key split: 1 or 2");
    constexpr int WQ = 8 / KS;                     // query groups (waves per key group)
    constexpr int BQ = WQ * QF * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = KS == 1 ? wave : (wave & (WQ - 1)), ksp = KS == 1 ? 0 : wave / WQ;
    const int l15 = lane & 15, lg = lane >> 4;
    // ---- block -> (sequence, kv head, query head of the group, query block): consecutive ids share K / V and share an XCD ----
    const int G = p.n_q_heads / p.n_kv_heads;
    int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb_r = vid % nqb; vid /= nqb;
    const int hq = vid % G; vid /= G;
    const int kvh = vid % p.n_kv_heads;
    const int seq = vid / p.n_kv_heads;
    const int h = kvh * G + hq;
    const int qb = CAUSAL ? nqb - 1 - qb_r : qb_r;
    int tok0 = seq * p.max_seqlen, seqlen = p.max_seqlen;   // cu_seqlens == NULL: n_seq sequences of max_seqlen tokens
    if (p.cu_seqlens != nullptr) { tok0 = p.cu_seqlens[seq]; seqlen = p.cu_seqlens[seq + 1] - tok0; }
    const int qb0 = qb * BQ;
    if (qb0 >= seqlen) return;

    int kv_end = seqlen;
    if (CAUSAL) { const int lim = qb0 + BQ; kv_end = lim < seqlen ? lim : seqlen; }
    const int ntiles = (kv_end + KT - 1) / KT;
    const bf16_t* kbase = p.k + (int64_t)tok0 * p.k_tok_stride + kvh * p.k_head_stride;
    const bf16_t* vbase = p.v + (int64_t)tok0 * p.v_tok_stride + kvh * p.v_head_stride;

    // ---- DMA plan of this lane: slot s = wave * PW + 64 i + lane of the stage image [K image | V image] ----
    const bf16_t* dsrc[P]; int64_t dstr[P]; int drow[P]; bool dact[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int sl = i * 64 + lane;
        dact[i] = sl < PW;
        const int s = wave * PW + (dact[i] ? sl : 0);
        const int isv = s >= KT * CHP;
        const int w = s - isv * KT * CHP;
        const int row = w / CHP, pos = w % CHP;
        int c = isv ? C::swz_v(row, pos) : C::swz_k(row, pos);
        if (HD == 72 && c >= CH) c = CH - 1;                  // pad chunk: a second copy of the last valid chunk (finite, never used)
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
            key = key < seqlen ? key : seqlen - 1;            // rows past the sequence: any valid row (masked in the softmax)
            char* dst = smem + st * C::STAGE + (wave * PW + i * 64) * 16;
            if (dact[i]) __builtin_amdgcn_global_load_lds((gbl_void_t*)(dsrc[i] + (int64_t)key * dstr[i]), (lds_void_t*)dst, 16, 0, 0);
        }
    };

    // ---- Q fragments (B operand of S^T): lane holds Q[q0 + f*16 + l15][kk*32 + lg*8 .. +8]; requested before the DMA burst ----
    const int qw0 = qb0 + wq * QF * 16;
    bf16x8 qf_[QF][KK];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qrow = qw0 + f * 16 + l15;
        const bf16_t* qp = p.q + (int64_t)(tok0 + (qrow < seqlen ? qrow : seqlen - 1)) * p.q_tok_stride + h * p.q_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int d = kk * 32 + lg * 8;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (d < HD && qrow < seqlen) v = *(const u32x4*)(qp + d);
            qf_[f][kk] = __builtin_bit_cast(bf16x8, v);
        }
    }
#pragma unroll
    for (int t = 0; t < NST - KS; ++t)
        if (t < ntiles) issue_tile(t);
    // pin the Q registers as "arrived" in front of the loop: left pending, the compiler's wait for them lands INSIDE the loop body as
    // a vmcnt(0) in front of the first MFMA of every tile, which would also drain the whole DMA ring
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(qf_[f][kk]));

    // ---- fragment read offsets (bytes inside an image) ----
    int koff[KK], voff[DN];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        int c = kk * 4 + lg;
        if (HD == 72 && kk == 2) c = 8 + (lg & 1);            // d >= 72: Q is zero there; any finite chunk of the row serves
        koff[kk] = l15 * ROWB + C::swz_k(l15, c) * 16;
    }
    const int vrow = lg * 4 + (l15 >> 2);                      // + 32 ks + 16 (second read): neither changes the swizzle class
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
        const int c = dn * 2 + ((l15 & 3) >> 1);
        voff[dn] = vrow * ROWB + C::swz_v(vrow, c) * 16 + (l15 & 1) * 8;
    }

    f32x4 oacc[DN][QF];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn)
#pragma unroll
        for (int f = 0; f < QF; ++f) oacc[dn][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) { m_run[f] = NEG_BIG; l_run[f] = 0.f; }
    const float c = p.scale * 1.4426950408889634f;   // exp(x*scale) = exp2(x*c)
    // P.V groups: GD output fragments (dn) of one k-step (ks) at a time
    // (hd 128 with 32 rows per wave is at the 256-register budget: two fragments per group and no early request there)
    constexpr bool TIGHT = (HD == 128 && QF == 2);
    constexpr int GD = TIGHT ? 2 : 4, NGD = (DN + GD - 1) / GD, NG = 2 * NGD;
    u32x2 vt[2][4][2];
    uint32_t v_lds = 0;                                     // LDS byte address of this tile's V image
    auto pv_issue = [&](auto gi) {
        constexpr int g = decltype(gi)::value;
        constexpr int ks = g / NGD, d0 = (g % NGD) * GD, b = g & 1;
#pragma unroll
        for (int j = 0; j < GD; ++j)
            if (d0 + j < DN) {
                vt[b][j][0] = ds_read_tr16_b64<ks * 32 * ROWB>(v_lds + voff[d0 + j]);
                vt[b][j][1] = ds_read_tr16_b64<ks * 32 * ROWB + 16 * ROWB>(v_lds + voff[d0 + j]);
            }
    };
    const int qw_last = qw0 + QF * 16 - 1;
    const bool wave_has_rows = qw0 < seqlen;

    // step s consumes tiles s*KS .. s*KS + KS-1 (key group ksp takes tile s*KS + ksp) and refills the stages of step s-1 with tiles
    // s*KS + NST-KS .. s*KS + NST-1; at its top `issued` tiles have been requested and the first `need` of them must have landed
    const int nsteps = (ntiles + KS - 1) / KS;
    for (int st = 0; st < nsteps; ++st) {
        {
            const int issued = (st * KS + NST - KS) < ntiles ? (st * KS + NST - KS) : ntiles;
            const int need = (st * KS + KS) < ntiles ? (st * KS + KS) : ntiles;
            wait_tiles_ahead<P, (NST - 2 * KS > 0 ? NST - 2 * KS : 0)>(issued - need);
        }
        __builtin_amdgcn_s_barrier();                       // everyone's pieces of this step's tiles are in; everyone is done with step st-1
        asm volatile("" ::: "memory");
        const int t = st * KS + ksp;
        const int key0 = t * KT;
        const bool active = wave_has_rows && t < ntiles && !(CAUSAL && key0 > qw_last);   // else nothing of this tile is visible to this wave's rows
        const char* cK = smem + ((t < ntiles ? t : 0) % NST) * C::STAGE;
        v_lds = lds_addr(cK + C::IMG);

        // ---- S^T = K Q^T ----
        f32x4 sacc[4][QF];
        if (active) {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int f = 0; f < QF; ++f) sacc[jn][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (TIGHT) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) {
                        const bf16x8 kf = *(const bf16x8*)(cK + jn * 16 * ROWB + koff[kk]);
#pragma unroll
                        for (int f = 0; f < QF; ++f)
                            sacc[jn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf_[f][kk], sacc[jn][f], 0, 0, 0);
                    }
                }
            } else {
                // every K fragment of the tile is requested first (48-64 registers), then the MFMAs retire them in order: ONE LDS
                // round trip on the tile's critical path instead of one per group of reads
                bf16x8 kf[KK][4];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn) kf[kk][jn] = *(const bf16x8*)(cK + jn * 16 * ROWB + koff[kk]);
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                        for (int f = 0; f < QF; ++f)
                            sacc[jn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk][jn], qf_[f][kk], sacc[jn][f], 0, 0, 0);
            }
        }
        // the refill of the stage tile t-1 was read from is issued BEHIND the MFMAs of this tile (the matrix pipe works through them
        // while the wave spends its 100-180 issue cycles per DMA instruction), not in front of them on the tile's critical path
#pragma unroll
        for (int j = 0; j < KS; ++j)
            if (st * KS + NST - KS + j < ntiles) issue_tile(st * KS + NST - KS + j);
        if (!active) continue;

        // ---- mask (sequence end / causal diagonal) ----
        const bool need_mask = (key0 + KT > seqlen) || (CAUSAL && (key0 + KT - 1 > qw0));
        if (need_mask) {
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const int qpos = qw0 + f * 16 + l15;
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kpos = key0 + jn * 16 + lg * 4 + r;
                        const bool ok = (kpos < seqlen) && (!CAUSAL || kpos <= qpos);
                        if (!ok) sacc[jn][f][r] = NEG_BIG;
                    }
            }
        }

        if constexpr (!TIGHT) pv_issue(std::integral_constant<int, 0>{});       // the first V^T group travels while the softmax runs

        // ---- online softmax (per q = lane&15; reduce over own 16 keys, then over the 4 lane groups) ----
        bf16x8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = sacc[0][f][0];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[jn][f][r]);
            mx = xor16_max(mx);                              // v_permlane16_swap / v_permlane32_swap: no LDS round trips
            mx = xor32_max(mx);
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[f] - m_new) * c);   // raw v_exp_f32: arguments are <= 0
            const float mc = m_new * c;
            m_run[f] = m_new;
            float ps = 0.f;
            float pv[4][4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[jn][f][r], c, -mc));
                    pv[jn][r] = e;
                    ps += e;
                }
            l_run[f] = l_run[f] * alpha + ps;
            if (!__all(alpha == 1.f)) {          // the running max moved for some row of this wave: rescale O (wave-uniform branch)
#pragma unroll
                for (int dn = 0; dn < DN; ++dn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[dn][f][r] *= alpha;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w[0] = cvt_pk_bf16(pv[2 * ks][0], pv[2 * ks][1]);
                w[1] = cvt_pk_bf16(pv[2 * ks][2], pv[2 * ks][3]);
                w[2] = cvt_pk_bf16(pv[2 * ks + 1][0], pv[2 * ks + 1][1]);
                w[3] = cvt_pk_bf16(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
                pf[f][ks] = __builtin_bit_cast(bf16x8, w);
            }
        }

        // ---- O^T += V^T P^T ----
        // V^T fragments via the LDS transpose read (asm form, see common.h): lane i of a 16-lane group points at
        // V[key0 + (i>>2)][d0 + 4*(i&3)] and receives V[key0 + 0..3][d0 + i]; key0 = ks*32 + lg*4 (+16 for k-slots 4..7).
        // Groups of up to GD output fragments per k-step, double buffered: group g+1 is requested before group g is retired with a
        // counted lgkmcnt, so the LDS latency of one group hides behind the MFMAs of the previous one.
        if constexpr (TIGHT) pv_issue(std::integral_constant<int, 0>{});
        static_for<0, NG>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            constexpr int ks = g / NGD;
            constexpr int d0 = (g % NGD) * GD;
            constexpr int cnt = (DN - d0) < GD ? (DN - d0) : GD;
            constexpr int b = g & 1;
            if constexpr (g + 1 < NG) {
                constexpr int nd0 = ((g + 1) % NGD) * GD;
                constexpr int ncnt = (DN - nd0) < GD ? (DN - nd0) : GD;
                pv_issue(std::integral_constant<int, g + 1>{});
                pv_retire<2 * ncnt, cnt>(vt[b]);
            } else {
                pv_retire<0, cnt>(vt[b]);
            }
#pragma unroll
            for (int j = 0; j < cnt; ++j) {
                u32x4 w;
                w[0] = vt[b][j][0][0]; w[1] = vt[b][j][0][1]; w[2] = vt[b][j][1][0]; w[3] = vt[b][j][1][1];
                const bf16x8 vf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    oacc[d0 + j][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[f][ks], oacc[d0 + j][f], 0, 0, 0);
            }
        });
    }

    // ---- normalise, stage O[q][d] through LDS (per-wave region), store whole 16-B row chunks ----
    __syncthreads();                                         // every wave is done with the K / V ring
    constexpr int MRG_LANE = DN * QF * 16 + QF * 8;          // bytes per lane of a key group's partial result: O^T accumulators, m, l
    if constexpr (KS == 2) {
        // the odd-tile group hands (m, l, O^T) to its even-tile partner (same rows, same lanes) through LDS, lane-linear: f32x4 slot i of lane
        // l at i * 1024 + l * 16; then  m = max(m0, m1), O = O0 * 2^((m0-m)c) + O1 * 2^((m1-m)c), l likewise (per lane: both partial sums are
        // over the same lane's key slots).  An empty side (m = NEG_BIG, l = 0, O = 0) scales by 2^(-huge) = 0 or, both empty, by 1 on zeros.
        char* mg = smem + (size_t)wq * 64 * MRG_LANE;
        if (ksp == 1) {
#pragma unroll
            for (int dn = 0; dn < DN; ++dn)
#pragma unroll
                for (int f = 0; f < QF; ++f) *(f32x4*)(mg + (dn * QF + f) * 1024 + lane * 16) = oacc[dn][f];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                *(float*)(mg + DN * QF * 1024 + (2 * f) * 256 + lane * 4) = m_run[f];
                *(float*)(mg + DN * QF * 1024 + (2 * f + 1) * 256 + lane * 4) = l_run[f];
            }
        }
        __syncthreads();
        if (ksp == 1) return;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const float m1 = *(const float*)(mg + DN * QF * 1024 + (2 * f) * 256 + lane * 4);
            const float l1 = *(const float*)(mg + DN * QF * 1024 + (2 * f + 1) * 256 + lane * 4);
            const float m = fmaxf(m_run[f], m1);
            const float a0 = __builtin_amdgcn_exp2f((m_run[f] - m) * c), a1 = __builtin_amdgcn_exp2f((m1 - m) * c);
            m_run[f] = m;
            l_run[f] = l_run[f] * a0 + l1 * a1;
#pragma unroll
            for (int dn = 0; dn < DN; ++dn) {
                const f32x4 o1 = *(const f32x4*)(mg + (dn * QF + f) * 1024 + lane * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[dn][f][r] = oacc[dn][f][r] * a0 + o1[r] * a1;
            }
        }
    }
    bf16_t* so = (bf16_t*)(smem + (KS == 2 ? (size_t)WQ * 64 * MRG_LANE : 0)) + wq * (QF * 16) * C::OSTR;
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int qrow = qw0 + f * 16 + l15;
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
            const int d = dn * 16 + lg * 4;
            if (d < HD) {
                u32x2 o;
                o[0] = cvt_pk_bf16(oacc[dn][f][0] * inv, oacc[dn][f][1] * inv);
                o[1] = cvt_pk_bf16(oacc[dn][f][2] * inv, oacc[dn][f][3] * inv);
                *(u32x2*)(so + (f * 16 + l15) * C::OSTR + d) = o;
            }
        }
        if (p.lse != nullptr && lg == 0 && qrow < seqlen)
            p.lse[(int64_t)h * p.total_tokens + tok0 + qrow] = m_run[f] * p.scale + logf(l);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // the wave reads back only what it wrote itself
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < (QF * 16 * CH + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CH, ch = idx % CH;
        const int qrow = qw0 + row;
        if (idx < QF * 16 * CH && qrow < seqlen) {
            const u32x4 v = *(const u32x4*)(so + row * C::OSTR + ch * 8);
            *(u32x4*)(p.o + (int64_t)(tok0 + qrow) * p.o_tok_stride + h * p.o_head_stride + ch * 8) = v;
        }
    }
}

template <int HD, bool CAUSAL, int QF, int KS = 1>
static int launch_attn_q(const AttnArgs& a, hipStream_t s) {
    using C = AttnDma<HD>;
    constexpr int WQ = 8 / KS;
    const size_t ring = (size_t)C::NST * C::STAGE;
    const size_t ost = (size_t)WQ * QF * 16 * C::OSTR * 2 + (KS == 2 ? (size_t)WQ * 64 * (C::DN * QF * 16 + QF * 8) : 0);
    const size_t lds = ring > ost ? ring : ost;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, QF, CAUSAL, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int nqb = cdiv(a.max_seqlen, WQ * QF * 16);
    const int64_t blocks = (int64_t)nqb * a.n_q_heads * a.n_seq;
    VILA_REQUIRE(blocks < (1ll << 31), "attn: grid too large");
    hipLaunchKernelGGL((attn_fwd_kernel<HD, QF, CAUSAL, KS>), dim3((unsigned)blocks), dim3(512), lds, s, a, nqb);
    VILA_LAUNCH_CHECK();
    return 0;
}

// 32 query rows per wave (QF = 2) halve the LDS bytes per MFMA.  While 256-row blocks would give the 256 CUs fewer than two rounds (one
// 448^2 tile, the S = 769 prefill, the 4 x 769 SFT batch) the block stays at 128 rows and its waves split the KEYS two ways instead (KS = 2,
// round 6).  VILA_ATTN_KS=0: rounds 3-5's choice for those grids, 16 rows per wave and every wave on every tile (A/B switch); 1 = automatic.
static int attn_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0; hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
static int attn_ks_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_ATTN_KS"); v = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1; }
    return v;
}
template <int HD, bool CAUSAL>
static int launch_attn_t(const AttnArgs& a, hipStream_t s) {
    const int64_t blocks2 = (int64_t)cdiv(a.max_seqlen, 256) * a.n_q_heads * a.n_seq;
    if (blocks2 < 512) {
        const int ks = attn_ks_env();
        if (ks == 0) return launch_attn_q<HD, CAUSAL, 1>(a, s);
        // 64-row blocks (16 rows per wave, 4 query groups x 2 key groups: half the dependent steps at the old per-step cost) while they all fit the
        // chip at once — one 448^2 tile: 256 blocks, 20.7 -> 16.3 us; S = 769 would be 364 blocks = two rounds, 21.1 -> 22.0 us, and keeps 128 rows
        // (profiles/r06_attn_keysplit_scan.txt).  VILA_ATTN_KS=2 / 3 force the 64- / 128-row form
        const int64_t blocks64 = (int64_t)cdiv(a.max_seqlen, 64) * a.n_q_heads * a.n_seq;
        if (ks == 2 || (ks == 1 && blocks64 <= attn_cu_count())) return launch_attn_q<HD, CAUSAL, 1, 2>(a, s);
        return launch_attn_q<HD, CAUSAL, 2, 2>(a, s);
    }
    return launch_attn_q<HD, CAUSAL, 2>(a, s);
}

static int attn_fwd_impl() {          // VILA_ATTN_FWD=v1 selects the round-2 kernel (A/B measurements); default: the DMA-ring kernel
    static int impl = -1;
    if (impl < 0) { const char* e = getenv("VILA_ATTN_FWD"); impl = (e && e[0] == 'v' && e[1] == '1') ? 1 : 2; }
    return impl;
}

int launch_attn_fwd(const AttnArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.n_seq >= 1 && a.total_tokens >= 1 && a.max_seqlen >= 1, "attn: empty input");
    VILA_REQUIRE(a.cu_seqlens != nullptr || (int64_t)a.n_seq * a.max_seqlen == a.total_tokens,
                 "attn: without cu_seqlens total_tokens (%d) must equal n_seq*max_seqlen (%d*%d)", a.total_tokens, a.n_seq, a.max_seqlen);
    VILA_REQUIRE(a.n_q_heads % a.n_kv_heads == 0, "attn: q heads (%d) must be a multiple of kv heads (%d)", a.n_q_heads, a.n_kv_heads);
    VILA_REQUIRE(a.q_tok_stride % 8 == 0 && a.k_tok_stride % 8 == 0 && a.v_tok_stride % 8 == 0 && a.o_tok_stride % 4 == 0 &&
                 a.q_head_stride % 8 == 0 && a.k_head_stride % 8 == 0 && a.v_head_stride % 8 == 0 && a.o_head_stride % 4 == 0,
                 "attn: strides must keep 16-B (q,k,v) / 8-B (o) alignment");
    VILA_REQUIRE((uintptr_t)a.q % 16 == 0 && (uintptr_t)a.k % 16 == 0 && (uintptr_t)a.v % 16 == 0 && (uintptr_t)a.o % 8 == 0, "attn: pointer alignment");
    if (attn_fwd_impl() == 1) {
        if (a.head_dim == 128) return a.causal ? launch_attn_v1_t<128, true>(a, s) : launch_attn_v1_t<128, false>(a, s);
        if (a.head_dim == 72) return a.causal ? launch_attn_v1_t<72, true>(a, s) : launch_attn_v1_t<72, false>(a, s);
        if (a.head_dim == 64) return a.causal ? launch_attn_v1_t<64, true>(a, s) : launch_attn_v1_t<64, false>(a, s);
    }
    if (a.head_dim == 128) return a.causal ? launch_attn_t<128, true>(a, s) : launch_attn_t<128, false>(a, s);
    if (a.head_dim == 72) return a.causal ? launch_attn_t<72, true>(a, s) : launch_attn_t<72, false>(a, s);
    if (a.head_dim == 64) return a.causal ? launch_attn_t<64, true>(a, s) : launch_attn_t<64, false>(a, s);
    VILA_FAIL(-1, "attn: unsupported head_dim %d (supported: 64, 72, 128)", a.head_dim);
}
