// W4A16 decode GEMVs (SURVEY.md §8f row 3 / BASELINE configs[4]: "AWQ int4 dequant-GEMV").  The reference names TinyChat
// (mit-han-lab/llm-awq, external, un-pinned, zero call sites in-tree: README.md:87) as its W4A16 backend; there is no reference
// code for it, so the format below is ours and parity is against a CPU dequantise-then-fp32 oracle of the SAME quantised weights.
//
// Quantisation: AWQ-style asymmetric uint4, groups of 128 along K:  w = (q - zero) * scale.
// HBM layout (vila_amd/quant.py: tile_w4) is TILE-MAJOR so that one wave instruction reads 1 KB contiguous:
//   Wq  [N/16 tiles][K/128 groups][64 lanes][4] u32      lane = 16*g + n: row n of the tile, 32-wide K slice g of the group;
//        word w of the lane holds the 8 weights k = 128*group + 32*g + 8*w + (0..7); nibble p (p<4) = element 2p, nibble p+4 =
//        element 2p+1, so ((word >> 4p) & 0x000F000F) | 0x43004300 is the bf16 PAIR (128+q[e_2p], 128+q[e_2p+1]): a shift and one
//        v_and_or per two weights (bf16 has 7 mantissa bits: only a nibble in bits 0..3 stays inside the [128, 256) binade)
//   Wsz [N/16 tiles][K/128 groups][16] u32                lo = bf16 scale, hi = bf16 (128 + zero), zero an integer 0..15
//
// Why MFMA for a GEMV: a VALU dot needs ~3.5 lane-ops per int4 weight (unpack, two bf16->f32 converts, FMA) and the first
// version of this kernel ran at 2.3 TB/s, VALU-bound.  Here the unpacked bf16 pairs ARE the B fragment of
// v_mfma_f32_16x16x32_bf16 (lane (n, g) supplies 8 consecutive k of output row n) and the activation is the A fragment,
// broadcast to all 16 A rows from LDS, so D[*][n] = sum_k x[k] * (128 + q[n][k]) over one 128-wide group after 4 MFMAs: 0.9 VALU
// ops per weight, the matrix pipe does the arithmetic (1.7 us of MFMA time for the 136 M weights of gate/up).  The 128-offset
// and the zero point come out per group:   y[n] += scale * (D - (128 + zero) * sum(x))     (group sums precomputed in LDS).
// One workgroup = one 16-row tile; its W waves split K (contiguous group ranges), partial sums meet in LDS.
#include "gemv_common.h"
#include "w4.h"

#define W4_MAX_WAVES 16

// stage x (optionally RMS-normalised, HF rounding order) as bf16 into LDS + the per-group sums of the staged values; any blockDim
// that is a multiple of 64.  Chunk c = 8 consecutive elements; 16 consecutive chunks (= 16 consecutive lanes) form a group.
__device__ __forceinline__ void stage_x_w4(const bf16_t* __restrict__ x, const bf16_t* __restrict__ norm_w, float eps, int K,
                                           bf16_t* sx, float* xg, float* scratch) {
    const int tid = threadIdx.x, nt = blockDim.x, nch = K >> 3, nw = nt >> 6;
    constexpr int MAXC = 3;
    const bool small = nch <= nt * MAXC;
    u32x4 v[MAXC], gw[MAXC];
    float rstd = 1.f;
    if (small) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + nt * i;
            v[i] = (c < nch) ? *(const u32x4*)(x + c * 8) : (u32x4){0u, 0u, 0u, 0u};
            // the gain rides with x: fetching it after the reduction would add a dependent L2 round trip
            gw[i] = (c < nch && norm_w != nullptr) ? *(const u32x4*)(norm_w + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    if (norm_w != nullptr) {
        float s = 0.f;
        if (small) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float a = lo_bf(v[i][k]), b = hi_bf(v[i][k]); s += a * a + b * b; }
        } else {
            for (int c = tid; c < nch; c += nt) {
                const u32x4 t = *(const u32x4*)(x + c * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float a = lo_bf(t[k]), b = hi_bf(t[k]); s += a * a + b * b; }
            }
        }
        s = wave_sum(s);
        if ((tid & 63) == 0) scratch[tid >> 6] = s;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += scratch[i];
        rstd = rsqrtf(tot / K + eps);
    }
    auto emit = [&](int c, const u32x4 t, const u32x4 g) {
        float e[8];
        if (norm_w != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[2 * k] = bfround(lo_bf(g[k]) * bfround(lo_bf(t[k]) * rstd));
                e[2 * k + 1] = bfround(hi_bf(g[k]) * bfround(hi_bf(t[k]) * rstd));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { e[2 * k] = lo_bf(t[k]); e[2 * k + 1] = hi_bf(t[k]); }
        }
        float a = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(e[2 * k], e[2 * k + 1]);
        *(u32x4*)(sx + c * 8) = o;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
        if ((c & 15) == 0) xg[c >> 4] = a;
    };
    // K % 128 == 0 makes nch a multiple of 16, so the 16 lanes of an aligned lane group are in range together
    if (small) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + nt * i;
            if (c < nch) emit(c, v[i], gw[i]);
        }
    } else {
        for (int c = tid; c < nch; c += nt)
            emit(c, *(const u32x4*)(x + c * 8), norm_w != nullptr ? *(const u32x4*)(norm_w + c * 8) : (u32x4){0u, 0u, 0u, 0u});
    }
    __syncthreads();
}

// stage x = the merge of the decode attention's per-slice partials (the flash-decoding combine of gemv.hip's stage_x_attn, here for the
// W4 o_proj): o[h][d] = sum_s exp(m_s - M) part_o[s][h][d] / sum_s exp(m_s - M) l_s, rounded to bf16 like the attention output, plus the
// group sums of the rounded values.  Heads are 128 wide = one quantisation group = 16 chunks of 8.
__device__ __forceinline__ void stage_x_attn_w4(const float* __restrict__ part_o, const float* __restrict__ part_ml, int n_active, int nq,
                                                bf16_t* sx, float* xg, float* wsm /* [n_active * nq] */) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int h = tid; h < nq; h += nt) {
        float M = -INFINITY;
        for (int s = 0; s < n_active; ++s) M = fmaxf(M, part_ml[((int64_t)s * nq + h) * 2]);
        float L = 0.f;
        for (int s = 0; s < n_active; ++s) {
            const float* ml = part_ml + ((int64_t)s * nq + h) * 2;
            L += __expf(ml[0] - M) * ml[1];
        }
        const float invL = 1.f / L;
        for (int s = 0; s < n_active; ++s) wsm[s * nq + h] = __expf(part_ml[((int64_t)s * nq + h) * 2] - M) * invL;
    }
    __syncthreads();
    const int nch = nq * 16;                                    // chunks of 8; a multiple of 16, so aligned 16-lane groups are in range together
    for (int c = tid; c < nch; c += nt) {
        const int h = c >> 4;
        float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < n_active; ++s) {
            const float* src = part_o + ((int64_t)s * nq) * 128 + c * 8;
            const f32x4 p0 = *(const f32x4*)src, p1 = *(const f32x4*)(src + 4);
            const float w = wsm[s * nq + h];
#pragma unroll
            for (int k = 0; k < 4; ++k) { e[k] = fmaf(w, p0[k], e[k]); e[4 + k] = fmaf(w, p1[k], e[4 + k]); }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = pack2bf(e[2 * k], e[2 * k + 1]); e[2 * k] = lo_bf(o[k]); e[2 * k + 1] = hi_bf(o[k]); }
        *(u32x4*)(sx + c * 8) = o;
        float a = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
        if ((c & 15) == 0) xg[h] = a;
    }
    __syncthreads();
}

// MODE 0: y = W x (+bias)(+residual) ; 4: the same with x merged from attention partials ; 1: rows interleaved gate/up, y = silu(g) * u ; 3: fused QKV + bias + RoPE + KV append
// (q/k rows interleaved so RoPE partners i, i + hd/2 are neighbours).  UB = groups (KB) per wave and item; PIPE = the next
// item's weights are issued before the current item is consumed (persistent blocks walking several tiles).
template <int MODE, int UB, bool PIPE>
__global__ __launch_bounds__(1024) void gemv_w4_kernel(GemvW4Args p, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, G = K >> 7;
    bf16_t* sx = (bf16_t*)smem;
    float* xg = (float*)(smem + K * 2);
    float* red = (float*)(xg + G);                 // [2][W4_MAX_WAVES][16]
    float* scratch = red + 2 * W4_MAX_WAVES * 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, W = blockDim.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int g0 = (G * wv) / W, g1 = (G * (wv + 1)) / W;
    const int half = p.hd >> 1;

    // The wave walks its (tile, batch-of-UB-groups) items in order.  The first item goes in flight before the activation is
    // staged (it does not depend on it).
    const u32x4* wq_base = (const u32x4*)p.Wq + lane;
    const uint32_t* wsz_base = p.Wsz + n;
    auto issue = [&](int tile_, int gb_, u32x4 (&wr)[UB], uint32_t (&sr)[UB]) {
        const u32x4* wq = wq_base + (size_t)tile_ * G * 64;
        const uint32_t* wsz = wsz_base + (size_t)tile_ * G * 16;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = gb_ + u;
            wr[u] = (c < g1) ? __builtin_nontemporal_load(wq + (size_t)c * 64) : (u32x4){0u, 0u, 0u, 0u};
            sr[u] = (c < g1) ? __builtin_nontemporal_load(wsz + c * 16) : 0u;
        }
    };
    // operands of the epilogue (bias / residual / RoPE row), fetched by the 16 epilogue lanes when a tile starts
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    int pos = 0;
    auto epi_fetch = [&](int tile_) {
        if (tid >= 16) return;
        const int pr = tile_ * 16 + tid;
        if (MODE == 0 || MODE == 4) {
            if (pr < p.N) {
                e0 = p.bias != nullptr ? bf2f(p.bias[pr]) : 0.f;
                e1 = p.residual != nullptr ? bf2f(p.residual[pr]) : 0.f;
            }
        } else if (MODE == 3) {
            const int head = pr / p.hd, within = pr - head * p.hd;
            if (head < p.nq + 2 * p.nkv) {
                pos = *p.pos_ptr;
                if (head >= p.nq + p.nkv) {
                    e0 = p.bias != nullptr ? bf2f(p.bias[pr]) : 0.f;
                } else {
                    const int i = within >> 1, b = within & 1;
                    e0 = p.bias != nullptr ? bf2f(p.bias[head * p.hd + i + b * half]) : 0.f;
                    e1 = p.bias != nullptr ? bf2f(p.bias[head * p.hd + i + (b ^ 1) * half]) : 0.f;
                    e2 = p.rope_cs[i]; e3 = p.rope_cs[half + i];
                }
            }
        }
    };
    int tile = blockIdx.x, gb = g0, par = 0;
    u32x4 wa[UB], wb[PIPE ? UB : 1];
    uint32_t sa[UB], sb[PIPE ? UB : 1];
    issue(tile, gb, wa, sa);
    epi_fetch(tile);
    if constexpr (MODE == 4) {
        const int n_active = (*p.pos_ptr + p.split_keys) / p.split_keys;     // ceil((pos + 1) / split_keys)
        stage_x_attn_w4(p.part_o, p.part_ml, n_active, K >> 7, sx, xg, scratch + W4_MAX_WAVES);
    } else {
        stage_x_w4(p.x, p.norm_w, p.eps, K, sx, xg, scratch);
    }

    float total = 0.f;
    for (;;) {
        int ntile = tile, ngb = gb + UB;
        if (ngb >= g1) { ngb = g0; ntile = tile + gridDim.x; }
        const bool have_next = ntile < n_tiles;
        if constexpr (PIPE) { if (have_next) issue(ntile, ngb, wb, sb); }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c = gb + u;
            if (c < g1) {                                       // wave-uniform
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // two chains: dependent MFMAs are 8 passes apart
                const bf16_t* xa = sx + c * 128 + g * 32;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t t = wa[u][w];
                    u32x4 b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = ((t >> (4 * j)) & 0x000F000Fu) | 0x43004300u;
                    const u32x4 a = *(const u32x4*)(xa + w * 8);
                    acc[w & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[w & 1], 0, 0, 0);
                }
                total = fmaf(lo_bf(sa[u]), (acc[0][0] + acc[1][0]) - hi_bf(sa[u]) * xg[c], total);
                __builtin_amdgcn_sched_barrier(0);              // keep the LDS reads of later groups from being hoisted (VGPRs)
            }
        }
        if (ngb == g0) {                                        // last batch of this tile: K-split partials meet in LDS
            float* rd = red + par * (W4_MAX_WAVES * 16);        // double-buffered: one barrier per tile is enough
            if (lane < 16) rd[wv * 16 + lane] = total;
            __syncthreads();
            if (tid < 16) {
                float v = 0.f, vp = 0.f;                        // own row and the partner row (n ^ 1)
                for (int i = 0; i < W; ++i) { v += rd[i * 16 + tid]; vp += rd[i * 16 + (tid ^ 1)]; }
                const int pr = tile * 16 + tid;                 // packed row
                if (MODE == 0 || MODE == 4) {
                    if (pr < p.N) {
                        v += e0;
                        if (p.residual != nullptr) v = bfround(v) + e1;
                        p.y[pr] = f2bf(v);
                    }
                } else if (MODE == 1) {
                    if ((tid & 1) == 0 && (pr >> 1) < p.N) p.y[pr >> 1] = f2bf(bfround(silu_f(bfround(v))) * bfround(vp));
                } else {
                    const int head = pr / p.hd, within = pr - head * p.hd;
                    const bool is_v = head >= p.nq + p.nkv;
                    if (head < p.nq + 2 * p.nkv) {
                        int d;                                  // element of the head this lane produces
                        float out;
                        if (is_v) {
                            d = within;
                            out = bfround(v + e0);
                        } else {
                            const int i = within >> 1, b = within & 1;
                            d = i + b * half;
                            const float mine = bfround(v + e0), other = bfround(vp + e1);
                            // rotate-half: lo' = lo*c - hi*s ; hi' = hi*c + lo*s   (bf16 rounding after every op, as the bf16 kernel)
                            out = b ? bfround(bfround(mine * e2) + bfround(other * e3)) : bfround(bfround(mine * e2) + bfround(-other * e3));
                        }
                        if (head < p.nq) {
                            p.q_out[head * p.hd + d] = f2bf(out);
                        } else if (pos < p.max_ctx) {
                            const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
                            bf16_t* dst = (is_v ? p.vcache : p.kcache) + ((int64_t)kvh * p.max_ctx + pos) * p.hd;
                            dst[d] = f2bf(out);
                        }
                    }
                }
            }
            par ^= 1;
            total = 0.f;
            if (have_next) epi_fetch(ntile);
        }
        if (!have_next) break;
        tile = ntile; gb = ngb;
        if constexpr (PIPE) {
#pragma unroll
            for (int u = 0; u < UB; ++u) { wa[u] = wb[u]; sa[u] = sb[u]; }
        } else {
            issue(tile, gb, wa, sa);
        }
    }
}

int launch_gemv_w4(const GemvW4Args& a, hipStream_t s) {
    VILA_REQUIRE(a.K % 128 == 0 && a.K > 0, "gemv_w4: K=%d must be a multiple of the 128-wide quantisation group", a.K);
    VILA_REQUIRE((uintptr_t)a.Wq % 16 == 0 && (uintptr_t)a.x % 16 == 0, "gemv_w4: pointer alignment");
    const int G = a.K / 128;
    int rows = a.N;                                            // packed rows
    if (a.mode == 1) rows = 2 * a.N;
    if (a.mode == 3) {
        VILA_REQUIRE(a.rope_cs != nullptr && a.pos_ptr != nullptr, "gemv_w4: qkv mode needs rope table and position");
        VILA_REQUIRE(a.hd % 16 == 0, "gemv_w4: head_dim=%d must be a multiple of the 16-row tile", a.hd);
        rows = (a.nq + 2 * a.nkv) * a.hd;
    }
    const int n_tiles = cdiv(rows, 16);
    // K-split: W waves of <= UB groups each.  K = 3584 -> 4 waves x 7 groups (7 KB per wave in flight, 14 with PIPE);
    // K = 18944 -> 16 waves x 9-10 groups, everything issued up front (one tile per block: nothing to pipeline against)
    const bool deep = G > 7 * W4_MAX_WAVES;
    const int UB = deep ? 10 : 7;
    int W = cdiv(G, UB);
    if (W > W4_MAX_WAVES) W = W4_MAX_WAVES;
    // persistent blocks: as many as are resident at once (<= 128 VGPRs -> 16 waves per CU), each walks tiles blockIdx, +grid, ...
    const int resident = 256 * (16 / W);
    const bool pipe = n_tiles > resident;
    const int grid = pipe ? resident : n_tiles;
    size_t lds = (size_t)a.K * 2 + (size_t)G * 4 + 2 * W4_MAX_WAVES * 16 * 4 + W4_MAX_WAVES * 4;
    if (a.mode == 4) {
        VILA_REQUIRE(a.part_o != nullptr && a.part_ml != nullptr && a.pos_ptr != nullptr && a.n_splits > 0 && a.split_keys > 0 && !deep,
                     "gemv_w4: attention-merge mode needs the partials, the position and K <= %d", 7 * W4_MAX_WAVES * 128);
        lds += (size_t)a.n_splits * G * 4;                     // the merge weights [n_splits][heads]
    }
    VILA_REQUIRE(lds <= 160 * 1024, "gemv_w4: K=%d does not fit the 160 KB LDS", a.K);
#define W4_LAUNCH(MODE, UB_, PIPE_) hipLaunchKernelGGL((gemv_w4_kernel<MODE, UB_, PIPE_>), dim3(grid), dim3(W * 64), lds, s, a, n_tiles)
    if (a.mode == 1) { if (pipe) W4_LAUNCH(1, 7, true); else W4_LAUNCH(1, 7, false); }
    else if (a.mode == 3) W4_LAUNCH(3, 7, false);
    else if (a.mode == 4) { if (pipe) W4_LAUNCH(4, 7, true); else W4_LAUNCH(4, 7, false); }
    else if (deep) W4_LAUNCH(0, 10, false);
    else if (pipe) W4_LAUNCH(0, 7, true);
    else W4_LAUNCH(0, 7, false);
#undef W4_LAUNCH
    VILA_LAUNCH_CHECK();
    return 0;
}
