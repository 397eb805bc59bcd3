// Batch-1 decode kernels (HOT LOOP #3): every weight byte is read exactly once per token, so
// the roofline is HBM (14.14 GB / token for NVILA-8B bf16), not MFMA.  Design rules (cdna guide, "GEMV / M<=16"):
// weights go straight HBM -> VGPR with 16-B non-temporal loads, deep unroll, late wait; the activation vector is
// staged once per block in LDS (with the preceding RMSNorm fused in, HF rounding order kept); all epilogues
// (bias, RoPE, KV-cache write, SiLU*up, residual add) are fused so a decoder layer is 6 launches.
#include "kernels.h"
#include "gemv_common.h"

#define DEC_KS 64      // keys per split of the decode attention
#define DEC_MAXG 8     // max query heads per kv head

typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned long long pack_f2(float a, float b) {
    return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
__device__ __forceinline__ float lo_f2(unsigned long long v) { return __uint_as_float((uint32_t)v); }
__device__ __forceinline__ float hi_f2(unsigned long long v) { return __uint_as_float((uint32_t)(v >> 32)); }

__device__ __forceinline__ u32x4 ldg_nt(const bf16_t* p) { return __builtin_nontemporal_load((const u32x4*)p); }

__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc = fmaf(lo_bf(w[k]), lo_bf(x[k]), acc);
        acc = fmaf(hi_bf(w[k]), hi_bf(x[k]), acc);
    }
    return acc;
}

// stage x = merged split-KV attention output (flash-decoding combine fused into the o_proj GEMV):
//   o[h][d] = sum_s exp(m_s - M) part_o[s][h][d] / sum_s exp(m_s - M) l_s ,  rounded to bf16 like the reference's attn output
__device__ __forceinline__ void stage_x_attn(const float* __restrict__ part_o, const float* __restrict__ part_ml, int n_active,
                                             int nq, bf16_t* sx, float* wsm /* [n_active*nq] */) {
    const int tid = threadIdx.x;
    for (int h = tid; h < nq; h += 256) {
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
    const int n4 = nq * 32;   // float4 chunks
    for (int i = tid; i < n4; i += 256) {
        const int h = i >> 5;
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < n_active; ++s) {
            const f32x4 pv = *(const f32x4*)(part_o + ((int64_t)s * nq) * 128 + i * 4);
            const float w = wsm[s * nq + h];
            o[0] = fmaf(w, pv[0], o[0]); o[1] = fmaf(w, pv[1], o[1]); o[2] = fmaf(w, pv[2], o[2]); o[3] = fmaf(w, pv[3], o[3]);
        }
        u32x2 r; r[0] = pack2bf(o[0], o[1]); r[1] = pack2bf(o[2], o[3]);
        *(u32x2*)(sx + i * 4) = r;
    }
    __syncthreads();
}

// ---- weight streaming -----------------------------------------------------------------------------
template <int R, int U> struct Batch { u32x4 v[U][R]; };

template <int R, int U>
__device__ __forceinline__ void load_batch(const bf16_t* const (&wrow)[R], int c0, int lane, int nch, Batch<R, U>& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
#pragma unroll
        for (int r = 0; r < R; ++r) b.v[u][r] = (c < nch) ? ldg_nt(wrow[r] + c * 8) : (u32x4){0u, 0u, 0u, 0u};
    }
}
template <int R, int U>
__device__ __forceinline__ void fma_batch(const Batch<R, U>& b, const bf16_t* sx, int c0, int lane, int nch, float (&acc)[R]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
        const u32x4 xv = (c < nch) ? *(const u32x4*)(sx + c * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot8(b.v[u][r], xv, acc[r]);
    }
}
// dot products of R rows with x (LDS) for chunks [c_start, nch); accumulates into acc, then reduces across the wave
template <int R, int U>
__device__ __forceinline__ void wave_rows_dot(const bf16_t* const (&wrow)[R], const bf16_t* sx, int K, int lane, float (&acc)[R], int c_start) {
    const int nch = K >> 3;
    for (int c0 = c_start; c0 < nch; c0 += 64 * U) {
        Batch<R, U> b;
        load_batch<R, U>(wrow, c0, lane, nch, b);
        fma_batch<R, U>(b, sx, c0, lane, nch, acc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

// ------------------------------------------------------------------------------------------------
// generic GEMV: y = W x (+bias) (+residual)   |   gate/up: y = silu(Wg x) * (Wu x)   |   x from attention partials
// The first weight batch of every wave is issued BEFORE the activation is staged, so the HBM latency of the first
// loads overlaps the norm / merge prologue.
// ------------------------------------------------------------------------------------------------
// U = 16-B loads in flight per row and lane: 7 covers a whole K = 3584 row in ONE round trip (the short K=hidden GEMVs are
// latency-bound), 4 is enough for the long rows (K = 18944) where many iterations pipeline anyway.
template <int MODE, int U>   // MODE 0 plain, 1 gate/up, 2 plain with x = merged attention partials
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs p, int n_groups, int ncu, int skew) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sx = (bf16_t*)smem;
    float* scratch = (float*)(smem + ((p.K * 2 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = p.K >> 3;
    constexpr int R = (MODE == 1) ? 4 : 2;
    // ncu > 0: the CU-balanced map (gemv_common.h CuMap; the grid is ncu x blocks-per-CU); ncu == 0: small grids, the plain grid-stride walk
    const CuMap cm(n_groups, ncu > 0 ? ncu : 8, ncu > 0 ? skew : 0);
    const int cu = ncu > 0 ? (int)(blockIdx.x % ncu) : 0;
    const int wpc = ncu > 0 ? (int)(gridDim.x / ncu) * 4 : 0;                 // waves dealing the CU's groups
    const int cnt = ncu > 0 ? cm.count(cu) : 0;
    int j = ncu > 0 ? (int)(blockIdx.x / ncu) * 4 + wave : 0;
    const int stride = gridDim.x * 4;
    int g = ncu > 0 ? (j < cnt ? cm.gid(cu, j) : n_groups) : blockIdx.x * 4 + wave;
    auto next_group = [&]() { if (ncu > 0) { j += wpc; g = j < cnt ? cm.gid(cu, j) : n_groups; } else g += stride; };

    auto rows_of = [&](int gg, const bf16_t* (&rows)[R]) {
        const int n = gg * 2;
        const int n1 = (n + 1 < p.N) ? n + 1 : n;
        if constexpr (MODE == 1) {
            rows[0] = p.W + (int64_t)n * p.K; rows[1] = p.W2 + (int64_t)n * p.K;
            rows[2] = p.W + (int64_t)n1 * p.K; rows[3] = p.W2 + (int64_t)n1 * p.K;
        } else {
            rows[0] = p.W + (int64_t)n * p.K; rows[1] = p.W + (int64_t)n1 * p.K;
        }
    };
    // Epilogue: after the wave reduction every lane holds the sums; lanes 0 and 1 each produce one of the wave's two adjacent outputs (rows n, n + 1).
    // Plain launch: each stores its own bf16.  Chained launch (a successor reads the output while this kernel's neighbours still run): lane 0
    // stores both as ONE write-through 4-byte word (gemv_common.h store_bf16_pair).
    const bool coh_out = p.chain.ctr != nullptr && p.chain.done_idx >= 0;
    float e_bias = 0.f, e_res = 0.f;
    auto finish = [&](int gg, float (&acc)[R]) {
        const int n = gg * 2;
        bf16_t o = 0;
        if constexpr (MODE == 1) {
            // HF: down(act(gate(x)) * up(x)) with every tensor rounded to bf16
            const float gv = bfround(lane == 0 ? acc[0] : acc[2]), uv = bfround(lane == 0 ? acc[1] : acc[3]);
            o = f2bf(bfround(silu_f(gv)) * uv);
        } else {
            float v = lane == 0 ? acc[0] : acc[1];
            v += e_bias;
            if (lane < 2 && n + lane < p.N && p.y_f32 != nullptr) p.y_f32[n + lane] = v;
            if (p.residual != nullptr) v = bfround(v) + e_res;
            o = f2bf(v);
        }
        if (p.y == nullptr) return;
        if (coh_out) {
            const bf16_t o1 = (bf16_t)__shfl((int)o, 1, 64);
            if (lane == 0 && n < p.N) store_bf16_pair(p.y, n, n + 1 < p.N, o, o1, true);
        } else if (lane < 2 && n + lane < p.N) {
            p.y[n + lane] = o;
        }
    };
    // the epilogue's operands are requested BEFORE the dot product: fetched after the reduction they add a dependent memory round
    // trip (~1 us) to the tail of every wave, i.e. to the kernel
    auto epi_fetch = [&](int gg) {
        if constexpr (MODE != 1) {
            const int nn = gg * 2 + lane;
            e_bias = 0.f; e_res = 0.f;
            if (lane < 2 && nn < p.N) {
                if (p.bias != nullptr) e_bias = bf2f(p.bias[nn]);
                if (p.residual != nullptr && p.y != nullptr) e_res = bf2f(p.residual[nn]);
            }
        }
    };

    const bf16_t* rows[R];
    Batch<R, U> b0;
    const bool has = g < n_groups;
    // issue the first weight batch before staging x only for short rows (K <= 4096: the staging latency is comparable to the
    // stream); for K = 18944 the x staging is long and queuing the weight loads in front of it measured 13 % slower
    // (a chained kernel always prefetches: what it loads before the wait streams in under its predecessor's tail)
    const bool early = has && (p.K <= 4096 || p.chain.ctr != nullptr);
    if (early) { rows_of(g, rows); load_batch<R, U>(rows, 0, lane, nch, b0); }
    if (has) epi_fetch(g);
    chain_wait(p.chain);                        // everything above reads weights / operands of kernels <= i-2 only
    if constexpr (MODE == 2) {
        const int ks = p.split_keys > 0 ? p.split_keys : DEC_KS;
        const int n_active = (*p.pos_ptr + ks) / ks;             // ceil((pos+1)/ks)
        stage_x_attn(p.part_o, p.part_ml, n_active, p.K >> 7, sx, scratch);
    } else if (p.chain.ctr != nullptr && p.chain.wait_idx >= 0) {
        stage_x<true>(p.x, p.norm_w, p.eps, p.K, sx, scratch);   // x comes from the kernel just waited for: sc1 loads
    } else {
        stage_x<false>(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    }
    if (early) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        fma_batch<R, U>(b0, sx, 0, lane, nch, acc);
        wave_rows_dot<R, U>(rows, sx, p.K, lane, acc, 64 * U);
        finish(g, acc);
        next_group();
    }
    for (; g < n_groups; next_group()) {
        rows_of(g, rows);
        epi_fetch(g);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        wave_rows_dot<R, U>(rows, sx, p.K, lane, acc, 0);
        finish(g, acc);
    }
    chain_done(p.chain);
}

// Grid sizing for the HBM-bound GEMVs: a multiple of the 256 CUs (the dispatcher deals blocks round-robin, so 448 blocks
// would leave 64 CUs with half the work of the others) and at most 4 blocks (16 waves) per CU = everything resident at once;
// waves then walk the row groups with a grid stride.
// bpc = blocks per CU cap: a chained kernel that leaves half of every CU's registers to its neighbour lets that neighbour's blocks be
// resident (and prefetching) while it runs (api.hip "chained decode step").
static int gemv_default_bpc() {           // VILA_GEMV_BPC (1..4): tuning / A-B switch for the unchained launches
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_GEMV_BPC"); v = (e && e[0] >= '1' && e[0] <= '4') ? e[0] - '0' : 4; }
    return v;
}
static int gemv_cu_count() {
    static thread_local int per_dev[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n = per_dev[dev & 15];
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
// VILA_GEMV_CU_MAP=0: the grid-stride walk of rounds 1-5 (A/B switch); VILA_GEMV_SKEW=n: row groups per CU moved from odd to even XCDs (gate/up)
static int gemv_cu_map_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_GEMV_CU_MAP"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}
static int gemv_skew() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_GEMV_SKEW"); v = (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : 1; }
    return v;
}
// grid of an HBM-bound GEMV: (CUs) x (blocks per CU, 1..4) so that blocks b, b + CUs, ... share a CU and deal that CU's row groups among their
// waves (*ncu_out = CUs); small problems (fewer groups than one block per CU would hold) keep one block per 4 groups and the plain walk (*ncu_out = 0)
static inline int balanced_grid(int n_groups, int bpc, int* ncu_out) {
    const int ncu = gemv_cu_count();
    int want = cdiv(n_groups, 4);
    const int b = (bpc >= 1 && bpc <= 4) ? bpc : gemv_default_bpc();
    if (want <= ncu || !gemv_cu_map_on()) {
        if (ncu_out) *ncu_out = 0;
        const int cap = ncu * b;
        if (want > cap) want = cap;
        return want <= ncu ? want : cdiv(want, ncu) * ncu;
    }
    int per = cdiv(want, ncu); if (per > b) per = b;
    if (ncu_out) *ncu_out = ncu;
    return ncu * per;
}

int launch_gemv(const GemvArgs& a, hipStream_t s, int* grid_out) {
    VILA_REQUIRE(a.K % 8 == 0 && a.K > 0 && a.N > 0, "gemv: K=%d must be a positive multiple of 8", a.K);
    VILA_REQUIRE((uintptr_t)a.W % 16 == 0, "gemv: weight pointer alignment");
    const int n_groups = cdiv(a.N, 2);
    int ncu = 0;
    int grid = balanced_grid(n_groups, a.max_bpc, &ncu);
    size_t lds = ((size_t)a.K * 2 + 15) / 16 * 16 + 16;
    const bool short_k = a.K <= 3584;
    GemvArgs b = a;                                              // (the chain link learns the grid it is launched with)
    if (a.mode == 1) {
        VILA_REQUIRE(a.W2 != nullptr && a.y != nullptr && (uintptr_t)a.x % 16 == 0, "gemv: gate/up mode needs W2, bf16 y, aligned x");
        b.chain.done_blocks = (uint32_t)grid;
        hipLaunchKernelGGL((gemv_kernel<1, 4>), dim3(grid), dim3(256), lds, s, b, n_groups, ncu, gemv_skew());
    } else if (a.mode == 2) {
        VILA_REQUIRE(a.part_o != nullptr && a.part_ml != nullptr && a.pos_ptr != nullptr && a.K % 128 == 0, "gemv: attention-merge mode needs partials");
        lds += (size_t)a.n_splits * (a.K / 128) * 4;
        const int cap = a.grid_cap > 0 ? a.grid_cap : 256;           // the merge prologue is paid per block: default ~1 block per CU
        if (grid > cap) grid = ncu > 0 ? (cap / ncu >= 1 ? (cap / ncu) * ncu : ncu) : cap;
        b.chain.done_blocks = (uint32_t)grid;
        if (short_k) hipLaunchKernelGGL((gemv_kernel<2, 7>), dim3(grid), dim3(256), lds, s, b, n_groups, ncu, 0);
        else hipLaunchKernelGGL((gemv_kernel<2, 4>), dim3(grid), dim3(256), lds, s, b, n_groups, ncu, 0);
    } else {
        VILA_REQUIRE((uintptr_t)a.x % 16 == 0, "gemv: x alignment");
        b.chain.done_blocks = (uint32_t)grid;
        if (short_k) hipLaunchKernelGGL((gemv_kernel<0, 7>), dim3(grid), dim3(256), lds, s, b, n_groups, ncu, 0);
        else hipLaunchKernelGGL((gemv_kernel<0, 4>), dim3(grid), dim3(256), lds, s, b, n_groups, ncu, 0);
    }
    VILA_LAUNCH_CHECK();
    if (grid_out != nullptr) *grid_out = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fused RMSNorm + QKV projection + bias + RoPE + KV-cache append for one new token.
// Group = 2 rows per wave: q/k heads -> the rotate-half pair {d, d+hd/2} of one head, v heads -> 2 consecutive rows.  cos/sin of the token's position come from the per-token table written by
// decode_prologue_kernel (already rounded to bf16 like HF's cast of cos/sin to the activation dtype).
// ------------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void qkv_decode_kernel(QkvDecodeArgs p, int ncu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sx = (bf16_t*)smem;
    float* scratch = (float*)(smem + ((p.K * 2 + 15) & ~15));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = p.hd, half = hd >> 1;
    const int gph = half;                                      // groups (row pairs) per head
    const int n_groups = (p.nq + 2 * p.nkv) * gph;
    const int nch = p.K >> 3;
    const CuMap cm(n_groups, ncu > 0 ? ncu : 8, 0);
    const int cu = ncu > 0 ? (int)(blockIdx.x % ncu) : 0;
    const int wpc = ncu > 0 ? (int)(gridDim.x / ncu) * 4 : 0;
    const int cnt = ncu > 0 ? cm.count(cu) : 0;
    int j = ncu > 0 ? (int)(blockIdx.x / ncu) * 4 + wave : 0;
    const int stride = gridDim.x * 4;
    int g = ncu > 0 ? (j < cnt ? cm.gid(cu, j) : n_groups) : blockIdx.x * 4 + wave;
    auto next_group = [&]() { if (ncu > 0) { j += wpc; g = j < cnt ? cm.gid(cu, j) : n_groups; } else g += stride; };

    int rows_i[2];
    const bf16_t* rows[2];
    auto rows_of = [&](int gg) {
        const int head = gg / gph, gi = gg % gph;
        if (head >= p.nq + p.nkv) { rows_i[0] = head * hd + gi * 2; rows_i[1] = rows_i[0] + 1; }   // v: two consecutive rows
        else { rows_i[0] = head * hd + gi; rows_i[1] = rows_i[0] + half; }                        // q/k: the rotate-half pair (d, d + hd/2)
        rows[0] = p.Wqkv + (int64_t)rows_i[0] * p.K;
        rows[1] = p.Wqkv + (int64_t)rows_i[1] * p.K;
    };
    // epilogue operands (position, the pair's biases, its RoPE row) are requested before the dot product, not after the reduction
    int e_pos = 0;
    float e_b0 = 0.f, e_b1 = 0.f, e_c = 1.f, e_s = 0.f;
    auto epi_fetch = [&](int gg) {
        if (lane >= 2) return;
        const int head = gg / gph, gi = gg % gph;
        e_pos = *p.pos_ptr;
        e_b0 = p.bqkv != nullptr ? bf2f(p.bqkv[rows_i[0]]) : 0.f;
        e_b1 = p.bqkv != nullptr ? bf2f(p.bqkv[rows_i[1]]) : 0.f;
        if (head < p.nq + p.nkv) { e_c = p.rope_cs[gi]; e_s = p.rope_cs[half + gi]; }
    };
    auto finish = [&](int gg, float (&acc)[2]) {
        if (lane >= 2) return;
        const int head = gg / gph;
        const bool is_v = head >= p.nq + p.nkv;
        const int pos = e_pos;
        const float lo = bfround(acc[0] + e_b0);
        const float hi = bfround(acc[1] + e_b1);
        float out = lane ? hi : lo;
        if (!is_v) {
            const float c = e_c, sn = e_s;
            out = lane ? bfround(bfround(hi * c) + bfround(lo * sn)) : bfround(bfround(lo * c) + bfround(-hi * sn));
        }
        const int row = rows_i[lane];
        if (head < p.nq) {
            p.q_out[row] = f2bf(out);
        } else if (pos < p.max_ctx) {
            const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
            bf16_t* dst = (is_v ? p.vcache : p.kcache) + ((int64_t)kvh * p.max_ctx + pos) * hd;
            dst[row - head * hd] = f2bf(out);
        }
    };

    Batch<2, U> b0;
    const bool has = g < n_groups;
    if (has) { rows_of(g); load_batch<2, U>(rows, 0, lane, nch, b0); epi_fetch(g); }
    chain_wait(p.chain);
    if (p.chain.ctr != nullptr && p.chain.wait_idx >= 0) stage_x<true>(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    else stage_x<false>(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    if (has) {
        float acc[2] = {0.f, 0.f};
        fma_batch<2, U>(b0, sx, 0, lane, nch, acc);
        wave_rows_dot<2, U>(rows, sx, p.K, lane, acc, 64 * U);
        finish(g, acc);
        next_group();
    }
    for (; g < n_groups; next_group()) {
        rows_of(g);
        epi_fetch(g);
        float acc[2] = {0.f, 0.f};
        wave_rows_dot<2, U>(rows, sx, p.K, lane, acc, 0);
        finish(g, acc);
    }
    chain_done(p.chain);
}

int launch_qkv_decode(const QkvDecodeArgs& a, hipStream_t s, int* grid_out) {
    VILA_REQUIRE(a.K % 8 == 0 && a.hd % 4 == 0 && a.rope_cs != nullptr, "qkv_decode: K=%d hd=%d", a.K, a.hd);
    const int n_groups = (a.nq + 2 * a.nkv) * (a.hd / 2);
    const size_t lds = ((size_t)a.K * 2 + 15) / 16 * 16 + 16;
    // one rotate-half pair per wave; K <= 3584: the whole row pair (14 x 16 B per lane) is in flight in ONE round trip
    int ncu = 0;
    const int grid = balanced_grid(n_groups, a.max_bpc, &ncu);
    QkvDecodeArgs b = a;
    b.chain.done_blocks = (uint32_t)grid;
    if (a.K <= 3584) hipLaunchKernelGGL(qkv_decode_kernel<7>, dim3(grid), dim3(256), lds, s, b, ncu);
    else hipLaunchKernelGGL(qkv_decode_kernel<4>, dim3(grid), dim3(256), lds, s, b, ncu);
    VILA_LAUNCH_CHECK();
    if (grid_out != nullptr) *grid_out = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// split-KV decode attention (q_len = 1, GQA, hd = 128): grid (n_splits, nkv); a block handles the G = nq/nkv query
// heads of one kv head over a 64-key slice of the cache and writes the un-normalised partial O and (m, l).
// The merge over splits is fused into the o_proj GEMV (gemv_kernel<2>).
//   scores : thread = (key, quarter of d): 32 FMAs per head, 2 cross-lane adds
//   P.V    : thread = (4 keys, 8-wide d chunk): 56 accumulators, reduced over the 16 key groups through LDS
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_decode_partial(AttnDecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sq = (float*)smem;                       // [G][128]  (pre-scaled)
    float* sc = sq + DEC_MAXG * 128;                // [G][64]
    float* red = sc + DEC_MAXG * DEC_KS;            // [16][G][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    const int nkeys = *p.pos_ptr + 1;
    const int k0 = split * DEC_KS;
    if (k0 >= nkeys) return;
    const int kn = (nkeys - k0) < DEC_KS ? (nkeys - k0) : DEC_KS;
    const bf16_t* kb = p.kcache + ((int64_t)kvh * p.max_ctx + k0) * 128;
    const bf16_t* vb = p.vcache + ((int64_t)kvh * p.max_ctx + k0) * 128;

    const int kq = tid >> 2, qd = tid & 3;          // scores: key, d quarter
    const int vkg = tid >> 4, vch = tid & 15;       // P.V: key group (4 keys), d chunk
    u32x4 kv_[4], vv_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        kv_[j] = (kq < kn) ? *(const u32x4*)(kb + kq * 128 + qd * 32 + j * 8) : (u32x4){0u, 0u, 0u, 0u};
        const int key = vkg * 4 + j;
        vv_[j] = (key < kn) ? *(const u32x4*)(vb + key * 128 + vch * 8) : (u32x4){0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < G * 128; i += 256) sq[i] = bf2f(p.q[kvh * G * 128 + i]) * p.scale;
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 q0 = *(const f32x4*)(sq + g * 128 + qd * 32 + j * 8);
            const f32x4 q1 = *(const f32x4*)(sq + g * 128 + qd * 32 + j * 8 + 4);
            a = fmaf(lo_bf(kv_[j][0]), q0[0], a); a = fmaf(hi_bf(kv_[j][0]), q0[1], a);
            a = fmaf(lo_bf(kv_[j][1]), q0[2], a); a = fmaf(hi_bf(kv_[j][1]), q0[3], a);
            a = fmaf(lo_bf(kv_[j][2]), q1[0], a); a = fmaf(hi_bf(kv_[j][2]), q1[1], a);
            a = fmaf(lo_bf(kv_[j][3]), q1[2], a); a = fmaf(hi_bf(kv_[j][3]), q1[3], a);
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (qd == 0) sc[g * DEC_KS + kq] = kq < kn ? a : -INFINITY;
    }
    __syncthreads();

    for (int g = wave; g < G; g += 4) {             // softmax statistics: one wave per head, lane = key
        const float s = sc[g * DEC_KS + lane];
        const float m = wave_max(s);
        const float e = __expf(s - m);
        const float l = wave_sum(e);
        sc[g * DEC_KS + lane] = e;
        if (lane == 0) {
            // (m, l) as ONE 8-byte write-through (sc1) store: the combine below reads it with sc1 loads, no fences needed
            gu64* ml = (gu64*)(p.part_ml + ((int64_t)split * p.nq + kvh * G + g) * 2);
            __hip_atomic_store(ml, pack_f2(m, l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pr = sc[g * DEC_KS + vkg * 4 + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = fmaf(pr, lo_bf(vv_[j][k]), o[2 * k]);
                o[2 * k + 1] = fmaf(pr, hi_bf(vv_[j][k]), o[2 * k + 1]);
            }
        }
        float* r = red + ((vkg * G + g) * 128 + vch * 8);
        *(f32x4*)r = (f32x4){o[0], o[1], o[2], o[3]};
        *(f32x4*)(r + 4) = (f32x4){o[4], o[5], o[6], o[7]};
    }
    __syncthreads();
    // partial O as 8-byte write-through (sc1) stores (guide G16 R1: payload sc1 -> every storing wave drains -> flag)
    for (int i = tid; i < G * 64; i += 256) {
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) { o0 += red[kg * G * 128 + 2 * i]; o1 += red[kg * G * 128 + 2 * i + 1]; }
        gu64* dst = (gu64*)(p.part_o + ((int64_t)split * p.nq + kvh * G) * 128 + 2 * i);
        __hip_atomic_store(dst, pack_f2(o0, o1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// merge of the split partials.  Measured alternatives (profiles/r01 notes in DESIGN.md §4.3): combine by the last-arriving
// block inside the partial kernel (fences: 14.6 us, sc1 stores+loads: 16.3 us) and combine inside the o_proj GEMV prologue
// (18.0 us for o_proj) are all no faster than this separate 7 us launch, so the simplest form is kept.
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

// ------------------------------------------------------------------------------------------------
// single-launch decode attention for short contexts (cache capacity <= 2048): one block of 16 waves per QUERY head walks the
// whole context (wave w takes 16-key chunks w, w+16, ...), online softmax per wave, cross-wave merge in LDS.  No split-KV
// partials and no merge launch: 2 launches -> 1 per layer (each launch costs ~3-4 us of floor inside the decode graph); the
// price is that the G = 7 query heads of a kv head each read that head's K/V (served by L2 / MALL, 2.5 % of a token's bytes).
//   scores: lane = (key = lane/4, d quarter = lane%4): 32 FMAs + 2 cross-lane adds;  P.V: lane = (4-key subgroup, 8-wide d chunk)
// ------------------------------------------------------------------------------------------------
// SPLIT: grid (nq, ceil(max_ctx / 256)); block (h, s) covers keys [256 s, 256 s + 256) — ONE 16-key chunk per wave, no loop — and writes the
// un-normalised partial (o, m, l) of that slice; the merge over the <= 8 slices happens in the o_proj GEMV's prologue (gemv_kernel<2>).
template <bool SPLIT>
__global__ __launch_bounds__(1024) void attn_decode_head(AttnDecodeArgs p) {
    __shared__ float sq[128];
    __shared__ float so[16][128];
    __shared__ float sml[16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, kvh = h / (p.nq / p.nkv);
    const int key_lo = SPLIT ? blockIdx.y * 256 : 0;
    const int row = blockIdx.z;                                  // batched decode: sequence = cache slot (0 for the batch-1 step)
    p.q += row * p.q_row_stride; p.kcache += row * p.slot_stride; p.vcache += row * p.slot_stride;
    if (!SPLIT) p.o += row * p.o_row_stride;
    const int nkeys_all = p.pos_ptr[row] + 1;
    if (SPLIT && key_lo >= nkeys_all) return;                    // block-uniform: slices beyond the context write nothing (the merge skips them)
    const int nkeys = SPLIT ? (nkeys_all < key_lo + 256 ? nkeys_all : key_lo + 256) : nkeys_all;
    const bf16_t* kb = p.kcache + (int64_t)kvh * p.max_ctx * 128;
    const bf16_t* vb = p.vcache + (int64_t)kvh * p.max_ctx * 128;
    if (tid < 128) sq[tid] = bf2f(p.q[h * 128 + tid]) * p.scale;
    __syncthreads();
    const int kq = lane >> 2, qd = lane & 3;        // scores: key within the chunk, d quarter
    const int sg = lane >> 4, dc = lane & 15;       // P.V: 4-key subgroup, d chunk
    float qr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) qr[i] = sq[qd * 32 + i];
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;

    u32x4 kc[4], vc[4], kn_[4], vn_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { kn_[j] = (u32x4){0u, 0u, 0u, 0u}; vn_[j] = (u32x4){0u, 0u, 0u, 0u}; }
    auto load_chunk = [&](int k0, u32x4 (&kk)[4], u32x4 (&vv)[4]) {
        const int key = k0 + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kk[j] = (key < nkeys) ? *(const u32x4*)(kb + (int64_t)key * 128 + qd * 32 + j * 8) : (u32x4){0u, 0u, 0u, 0u};
            const int vk = k0 + sg * 4 + j;
            vv[j] = (vk < nkeys) ? *(const u32x4*)(vb + (int64_t)vk * 128 + dc * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    int k0 = key_lo + wave * 16;
    if (k0 < nkeys) load_chunk(k0, kc, vc);
    for (; k0 < nkeys; k0 += 256) {
        const int kn = k0 + 256;
        if (kn < nkeys) load_chunk(kn, kn_, vn_);            // prefetch the wave's next chunk under this chunk's math
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a = fmaf(lo_bf(kc[j][e]), qr[j * 8 + 2 * e], a);
                a = fmaf(hi_bf(kc[j][e]), qr[j * 8 + 2 * e + 1], a);
            }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        const float s = (k0 + kq < nkeys) ? a : -INFINITY;
        float cm = s;
        cm = fmaxf(cm, __shfl_xor(cm, 4, 64)); cm = fmaxf(cm, __shfl_xor(cm, 8, 64));
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float m_new = fmaxf(m, cm);
        const float alpha = __expf(m - m_new);
        const float pr = __expf(s - m_new);
        float ps = pr;
        ps += __shfl_xor(ps, 4, 64); ps += __shfl_xor(ps, 8, 64); ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = m_new;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pj = __shfl(pr, (sg * 4 + j) * 4, 64);   // probability of key k0 + sg*4 + j (held by lanes 4*key .. 4*key+3)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] = fmaf(pj, lo_bf(vc[j][e]), o[2 * e]);
                o[2 * e + 1] = fmaf(pj, hi_bf(vc[j][e]), o[2 * e + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { kc[j] = kn_[j]; vc[j] = vn_[j]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) so[wave][dc * 8 + e] = o[e];
    }
    if (lane == 0) { sml[wave][0] = m; sml[wave][1] = l; }
    __syncthreads();
    if (tid < 128) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 16; ++w) M = fmaxf(M, sml[w][0]);
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const float wgt = __expf(sml[w][0] - M);
            L += wgt * sml[w][1];
            acc += wgt * so[w][tid];
        }
        if (SPLIT) {
            const int64_t slot = (int64_t)blockIdx.y * p.nq + h;
            p.part_o[slot * 128 + tid] = acc;
            if (tid == 0) { p.part_ml[slot * 2] = M; p.part_ml[slot * 2 + 1] = L; }
        } else {
            p.o[h * 128 + tid] = f2bf(acc / L);
        }
    }
}


// batched decode: one block per (query head, sequence) over the sequence's whole context (caches up to 2048 positions)
int launch_attn_decode_rows(const AttnDecodeArgs& a0, int n_rows, int64_t q_row_stride, int64_t o_row_stride, int64_t slot_stride, hipStream_t s) {
    AttnDecodeArgs a = a0;
    VILA_REQUIRE(a.hd == 128 && a.o != nullptr && a.max_ctx <= 2048 && n_rows >= 1, "attn_decode_rows: head_dim 128, caches up to 2048 positions");
    VILA_REQUIRE(a.nq % a.nkv == 0, "attn_decode_rows: q heads must be a multiple of kv heads");
    a.q_row_stride = q_row_stride; a.o_row_stride = o_row_stride; a.slot_stride = slot_stride;
    hipLaunchKernelGGL(attn_decode_head<false>, dim3(a.nq, 1, n_rows), dim3(1024), 0, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}

int launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s, int* grid_out) {
    VILA_REQUIRE(a.hd == 128, "attn_decode: head_dim must be 128 (got %d)", a.hd);
    VILA_REQUIRE(a.nq % a.nkv == 0 && a.nq / a.nkv <= DEC_MAXG, "attn_decode: GQA group %d/%d unsupported (max %d)", a.nq, a.nkv, DEC_MAXG);
    VILA_REQUIRE(a.n_splits * DEC_KS >= a.max_ctx, "attn_decode: n_splits too small for max_ctx");
    if (a.split256) {                                            // partials per 256-key slice; merged by the o_proj GEMV (mode 2, split_keys 256)
        VILA_REQUIRE(a.max_ctx <= 2048 && a.n_splits * DEC_KS >= a.max_ctx, "attn_decode: 256-key slices need max_ctx <= 2048");
        hipLaunchKernelGGL(attn_decode_head<true>, dim3(a.nq, cdiv(a.max_ctx, 256)), dim3(1024), 0, s, a);
        VILA_LAUNCH_CHECK();
        if (grid_out != nullptr) *grid_out = a.nq * cdiv(a.max_ctx, 256);
        return 0;
    }
    if (a.o != nullptr && a.max_ctx <= 2048 && !a.force_split) {
        hipLaunchKernelGGL(attn_decode_head<false>, dim3(a.nq), dim3(1024), 0, s, a);
        VILA_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = (size_t)(DEC_MAXG * 128 + DEC_MAXG * DEC_KS + 16 * DEC_MAXG * 128) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)attn_decode_partial, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_decode_partial, dim3(a.n_splits, a.nkv), dim3(256), lds, s, a);
    VILA_LAUNCH_CHECK();
    if (a.o != nullptr) {
        hipLaunchKernelGGL(attn_decode_merge, dim3(a.nq), dim3(128), 0, s, a);
        VILA_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-token prologue: x = embed[token]; rope table for the token's position: cs[0:hd/2] = cos, cs[hd/2:hd] = sin,
// both rounded to bf16 (HF casts cos/sin to the activation dtype before use)
__global__ void decode_prologue_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ tok, bf16_t* __restrict__ out, int H,
                                       int64_t vocab, const int32_t* __restrict__ pos, float* __restrict__ rope_cs, int hd, float theta,
                                       uint32_t* __restrict__ chain_ctr, int n_chain) {
    // chained step: this token's done counters start at zero (every chained kernel is launched behind this one)
    // (n_chain kernels x (1 count + CHAIN_FLAGS flag) words, CHAIN_STRIDE words apart)
    if (chain_ctr != nullptr)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_chain * (CHAIN_FLAGS + 1); i += gridDim.x * blockDim.x)
            chain_ctr[(size_t)i * CHAIN_STRIDE] = 0u;
    int64_t id = *tok;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < (H >> 3); c += gridDim.x * blockDim.x)
        *(u32x4*)(out + c * 8) = *(const u32x4*)(table + id * H + c * 8);
    if (blockIdx.x == 0 && (int)threadIdx.x < (hd >> 1)) {
        const int d = threadIdx.x;
        const float inv = 1.0f / powf(theta, (float)(2 * d) / (float)hd);
        const float ang = (float)(*pos) * inv;
        rope_cs[d] = bfround(cosf(ang));
        rope_cs[(hd >> 1) + d] = bfround(sinf(ang));
    }
}
int launch_decode_prologue(const bf16_t* table, const int64_t* tok, bf16_t* out, int H, int64_t vocab, const int32_t* pos, float* rope_cs,
                           int hd, float theta, hipStream_t s, uint32_t* chain_ctr, int n_chain) {
    VILA_REQUIRE(hd / 2 <= 256, "decode_prologue: head_dim too large");
    hipLaunchKernelGGL(decode_prologue_kernel, dim3(cdiv(H / 8, 256)), dim3(256), 0, s, table, tok, out, H, vocab, pos, rope_cs, hd, theta,
                       chain_ctr, n_chain);
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
