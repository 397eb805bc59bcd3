// 256x256x64 bf16 MFMA GEMM for the large contractions (SFT step: T = 3076 packed tokens; wgrad / dgrad; gate/up and
// down projections of the prefill).
//
// Why a second kernel: on gfx950 the 128x128 tile moves 32 KB per K-tile through the 64 B/clk vector-memory path and the
// LDS write port for only 512 MFMA cycles per SIMD — it is L1/LDS-bound by construction.  The 256x256 tile halves the bytes
// per flop, and LDS-DMA (`global_load_lds`, 16 B per lane) removes the VGPR round trip and the ds_write pass of the staging.
//
// Structure (cdna guide §5 "glds, 2 LDS buffers, BK = 64"):
//   8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 accumulator fragments (128 VGPRs), 64 MFMAs per wave and K-tile
//   LDS = 2 K-tile buffers x {A_lo, A_hi, B_lo, B_hi} half-tiles of [128][64] bf16 (16 KB each) = 128 KB
//   one barrier per K-tile:  wait own DMA (vmcnt 0) -> barrier -> issue the DMA of tile t+1 into the other buffer ->
//   4 quadrant phases on tile t (ds_read_b128 fragments, 16 MFMAs each); the two waves of a SIMD run free between
//   barriers, so one wave's LDS reads overlap the other's MFMAs.
//   LDS image is lane-linear for the DMA; the 16-B slot swizzle (slot ^= (row>>1)&7, conflict-free for ds_read_b128 and the
//   same involution as gemm.hip) is applied on the per-lane SOURCE address and on the fragment reads (guide rule 21).
// Variants: fused gate/up epilogue (each wave's 64 B rows = 32 gate + 32 up rows of the same output columns -> silu(g)*u),
// and split-K (grid.y slices of K write fp32 slabs that splitk_reduce_kernel sums with bias / residual) for the
// M = 769 x N = 3584 prefill shapes whose 56 tiles cannot fill 256 CUs.
// K tail (K % 64 != 0, K % 8 == 0): the 16-B chunks of the last K-tile that lie beyond K are DMA'd from a zero chunk in global
// memory (LDS-DMA cannot predicate a lane's LDS write, but it can read a different address).  M / N tails by row clamping +
// masked stores.  (Skipping the MFMAs of half-tiles beyond M was measured and does not pay: with one tile of prefetch the DMA
// round trip of a K-tile costs as much as its 64 MFMAs, so an "empty" tile is not cheaper than a full one.)
#include "gemm256_kernel.h"

// Sum of `splits` fp32 slices of 4 consecutive columns, slice 0 first: the order every reduce of this file keeps.
// (Round 6 measured a variant with the slice count as a template parameter and every slice's load requested before the first add: the reduce
// launches of the prefill stayed at 15.4-16.0 us — they move 57-66 MB of slabs at ~4.3 TB/s, memory-bound, not latency-chained — removed.)
__device__ __forceinline__ f32x4 sum_slices4(const float* __restrict__ p, int64_t stride, int splits) {
    f32x4 v = *(const f32x4*)p;
    for (int s = 1; s < splits; ++s) {
        const f32x4 w = *(const f32x4*)(p + s * stride);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    return v;
}

// tail round of a gate/up GEMM: out[m][col0 + c] = bf16(silu(sum_s gate_s[m][c]) * sum_s up_s[m][c]); slab = [splits][2][M][tc] fp32
__global__ void splitk_gu_reduce_kernel(const float* __restrict__ slab, int splits, bf16_t* __restrict__ out, int64_t ldc, int M, int tc, int col0) {
    const int n4 = tc >> 2;
    const int64_t total = (int64_t)M * n4, plane = (int64_t)M * tc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), c = (int)(i % n4) * 4;
        const float* b = slab + (int64_t)m * tc + c;
        const f32x4 g = sum_slices4(b, 2 * plane, splits), u = sum_slices4(b + plane, 2 * plane, splits);
        u32x2 o;
        o[0] = pack2bf(silu_f(g[0]) * u[0], silu_f(g[1]) * u[1]);
        o[1] = pack2bf(silu_f(g[2]) * u[2], silu_f(g[3]) * u[3]);
        *(u32x2*)(out + (int64_t)m * ldc + col0 + c) = o;
    }
}

// out[m][n] = bf16(sum_s slab[s][m][n] + bias[n] + residual[m][n])
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t slab_stride, const bf16_t* __restrict__ bias,
                                     const bf16_t* __restrict__ residual, int64_t ldr, bf16_t* __restrict__ out, int64_t ldc, int M, int N, int res_mod) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), c = (int)(i % n4) * 4;
        f32x4 v = sum_slices4(slab + (int64_t)m * N + c, slab_stride, splits);
        if (bias != nullptr) {
            const u32x2 b = *(const u32x2*)(bias + c);
            v[0] += lo_bf(b[0]); v[1] += hi_bf(b[0]); v[2] += lo_bf(b[1]); v[3] += hi_bf(b[1]);
        }
        if (residual != nullptr) {
            const u32x2 r = *(const u32x2*)(residual + (int64_t)(res_mod > 0 ? m % res_mod : m) * ldr + c);
            v[0] += lo_bf(r[0]); v[1] += hi_bf(r[0]); v[2] += lo_bf(r[1]); v[3] += hi_bf(r[1]);
        }
        u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(out + (int64_t)m * ldc + c) = o;
    }
}

// The reduce of a K-sliced q/k/v projection with rope_kv_kernel's work fused in (round 6): one thread per (token, head, 8-wide chunk of the low half
// of the head) sums the slices of its 8 + 8 columns in slice order, adds the bias and rounds to bf16 exactly as splitk_reduce_kernel does, then
// applies HF's rotation with HF's roundings (bf16(x * cos) + bf16(rot * sin), elementwise.hip rope_kv_kernel: same expressions) to the q and k
// heads, stores the row into the fused qkv buffer and the k / v heads into the cache.  One launch and one round trip of the qkv buffer less.
__global__ void splitk_reduce_rope_kernel(const float* __restrict__ slab, int splits, int64_t slab_stride, const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                          int64_t ldc, int S, int N, const float* __restrict__ cs, const float* __restrict__ sn, const int32_t* __restrict__ pos,
                                          const int32_t* __restrict__ seq_of_tok, bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache, int nq, int nkv, int hd,
                                          int max_ctx) {
    const int half = hd >> 1, cpr = half >> 3, heads = nq + 2 * nkv;
    const int64_t total = (int64_t)S * heads * cpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int ch = (int)(r % cpr); r /= cpr;
        const int hh = (int)(r % heads); r /= heads;
        const int s = (int)r;
        const int col = hh * hd + ch * 8;
        u32x4 x1, x2;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c0 = col + hf * half;
            const float* p0 = slab + (int64_t)s * N + c0;
            f32x4 a = sum_slices4(p0, slab_stride, splits), b = sum_slices4(p0 + 4, slab_stride, splits);
            if (bias != nullptr) {
                const u32x4 bv = *(const u32x4*)(bias + c0);
                a[0] += lo_bf(bv[0]); a[1] += hi_bf(bv[0]); a[2] += lo_bf(bv[1]); a[3] += hi_bf(bv[1]);
                b[0] += lo_bf(bv[2]); b[1] += hi_bf(bv[2]); b[2] += lo_bf(bv[3]); b[3] += hi_bf(bv[3]);
            }
            u32x4 o; o[0] = pack2bf(a[0], a[1]); o[1] = pack2bf(a[2], a[3]); o[2] = pack2bf(b[0], b[1]); o[3] = pack2bf(b[2], b[3]);
            if (hf == 0) x1 = o; else x2 = o;
        }
        if (hh < nq + nkv) {
            const float* c = cs + (int64_t)s * half + ch * 8;
            const float* sv = sn + (int64_t)s * half + ch * 8;
            u32x4 o1, o2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a0 = lo_bf(x1[k]), a1 = hi_bf(x1[k]), b0 = lo_bf(x2[k]), b1 = hi_bf(x2[k]);
                const float c0 = c[2 * k], c1 = c[2 * k + 1], s0 = sv[2 * k], s1 = sv[2 * k + 1];
                o1[k] = pack2bf(bfround(a0 * c0) + bfround(-b0 * s0), bfround(a1 * c1) + bfround(-b1 * s1));
                o2[k] = pack2bf(bfround(b0 * c0) + bfround(a0 * s0), bfround(b1 * c1) + bfround(a1 * s1));
            }
            x1 = o1; x2 = o2;
        }
        bf16_t* base = out + (int64_t)s * ldc + col;
        *(u32x4*)base = x1;
        *(u32x4*)(base + half) = x2;
        const int p = pos[s];
        if (hh >= nq && kcache != nullptr && p >= 0 && p < max_ctx) {
            const int sq = seq_of_tok != nullptr ? seq_of_tok[s] : 0;
            const bool isv = hh >= nq + nkv;
            const int kvh = isv ? hh - nq - nkv : hh - nq;
            bf16_t* dst = (isv ? vcache : kcache) + (((int64_t)sq * nkv + kvh) * max_ctx + p) * hd + ch * 8;
            *(u32x4*)dst = x1;
            *(u32x4*)(dst + half) = x2;
        }
    }
}

// The same reduce with the NEXT block's normalisation fused in (round 4): one block per output row — sum the slices (+ bias + residual), store
// the row (the residual stream, rounded to bf16 as the separate kernel does), then normalise those ROUNDED values (what a separate norm launch would
// read back) and store norm_out.  Saves the norm launch and its read of the row: LayerNorm (RMS = false: (x - mean) * rstd * w + b, two-pass
// variance) or Qwen2RMSNorm (w * bf16(x * rstd)); same per-element arithmetic as elementwise.hip norm_kernel, N % 8 == 0, N <= 16384.
template <bool RMS>
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* __restrict__ slab, int splits, int64_t slab_stride, const bf16_t* __restrict__ bias,
                                                        const bf16_t* __restrict__ residual, int64_t ldr, bf16_t* __restrict__ out, int64_t ldc, int N, int res_mod,
                                                        const bf16_t* __restrict__ nw, const bf16_t* __restrict__ nb, float eps, bf16_t* __restrict__ nout) {
    __shared__ float scratch[4];
    const int m = blockIdx.x, tid = threadIdx.x, nch = N >> 3;
    constexpr int MAXC = 8;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            const float* p = slab + (int64_t)m * N + c * 8;
            const f32x4 a = sum_slices4(p, slab_stride, splits), b = sum_slices4(p + 4, slab_stride, splits);
            float e[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if (bias != nullptr) {
                const u32x4 bv = *(const u32x4*)(bias + c * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[2 * k] += lo_bf(bv[k]); e[2 * k + 1] += hi_bf(bv[k]); }
            }
            if (residual != nullptr) {
                const u32x4 rv = *(const u32x4*)(residual + (int64_t)(res_mod > 0 ? m % res_mod : m) * ldr + c * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[2 * k] += lo_bf(rv[k]); e[2 * k + 1] += hi_bf(rv[k]); }
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = pack2bf(e[2 * k], e[2 * k + 1]);
                v[i][2 * k] = lo_bf(o[k]); v[i][2 * k + 1] = hi_bf(o[k]);          // the stored (rounded) values are what gets normalised
                s += RMS ? (v[i][2 * k] * v[i][2 * k] + v[i][2 * k + 1] * v[i][2 * k + 1]) : (v[i][2 * k] + v[i][2 * k + 1]);
            }
            *(u32x4*)(out + (int64_t)m * ldc + c * 8) = o;
        }
    }
    s = block_sum_256(s, scratch);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / N + eps);
    } else {
        mean = s / N;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (tid + 256 * i < nch) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
            }
        q = block_sum_256(q, scratch);
        rstd = rsqrtf(q / N + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            const u32x4 wv = *(const u32x4*)(nw + c * 8);
            u32x4 bv = (u32x4){0u, 0u, 0u, 0u};
            if (!RMS && nb != nullptr) bv = *(const u32x4*)(nb + c * 8);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = v[i][2 * k], bb = v[i][2 * k + 1];
                if (RMS) { a = lo_bf(wv[k]) * bfround(a * rstd); bb = hi_bf(wv[k]) * bfround(bb * rstd); }
                else { a = (a - mean) * rstd * lo_bf(wv[k]) + lo_bf(bv[k]); bb = (bb - mean) * rstd * hi_bf(wv[k]) + hi_bf(bv[k]); }
                o[k] = pack2bf(a, bb);
            }
            *(u32x4*)(nout + (int64_t)m * N + c * 8) = o;
        }
    }
}

// tail tiles behind whole rounds: out tile (tm, tn) = bf16(sum_s slab[s][t][256][256] + bias + residual); slab slot t holds tile id tile0 + t
// of the GEMM's tile order (the kernel indexes its slab by id - tile0, after its own XCD remap)
__global__ __launch_bounds__(256) void splitk_tail_reduce_kernel(const float* __restrict__ slab, int splits, int n_tail, int tile0, int tiles_m, int tiles_n, int grp,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual, int64_t ldr,
                                                                 bf16_t* __restrict__ out, int64_t ldc, int M, int N, int res_mod) {
    const int t = blockIdx.x >> 4, part = blockIdx.x & 15;                   // 16 blocks of 16 rows per tile
    const int id = t + tile0;
    int tm, tn;
    gemm256_tile_of(id, tiles_m, tiles_n, grp, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int c = (threadIdx.x & 63) * 4;
    for (int r = part * 16 + (threadIdx.x >> 6); r < part * 16 + 16; r += 4) {
        const int gm = m0 + r, gc = n0 + c;
        if (gm >= M || gc >= N) continue;
        f32x4 v = sum_slices4(slab + (int64_t)t * 65536 + r * 256 + c, (int64_t)n_tail * 65536, splits);
        if (bias != nullptr) {
            const u32x2 b = *(const u32x2*)(bias + gc);
            v[0] += lo_bf(b[0]); v[1] += hi_bf(b[0]); v[2] += lo_bf(b[1]); v[3] += hi_bf(b[1]);
        }
        if (residual != nullptr) {
            const u32x2 rv = *(const u32x2*)(residual + (int64_t)(res_mod > 0 ? gm % res_mod : gm) * ldr + gc);
            v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
        }
        u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(out + (int64_t)gm * ldc + gc) = o;
    }
}

int g_gemm256_sched = 0;      // tuning hook (vila_gemm_force_sched): 0 = default schedule of each layout; 1 / 2 / 9 = gemm256_kernel.h SCHED, 10 = SCHED 0
extern "C" void vila_gemm_force_sched(int sched) { g_gemm256_sched = sched; }
int launch_gemm256_cm(const GemmArgs& a, hipStream_t s);                       // gemm256_cm.hip
int launch_gemm256_cm_splitk(const GemmArgs& a, int splits, float* slab, int per, hipStream_t s);
int launch_gemm256_sched(const GemmArgs& a, int sched, hipStream_t s);
int launch_gemm256_cm_range(const GemmArgs& a, int mode, int splits, int tile0, int n_tiles, int per, hipStream_t s);
int g_gemm256_group = -1;           // tuning hook (vila_gemm_force_group): tile order, see gemm256_kernel.h
extern "C" void vila_gemm_force_group(int grp) { g_gemm256_group = grp; }
int g_gemm256_ex = -1;             // tuning hook (vila_gemm_force_ex): see gemm256_kernel.h
extern "C" void vila_gemm_force_ex(int mode) { g_gemm256_ex = mode; }
int g_gemm256_bm = 0;              // tuning hook (vila_gemm_force_bm): 0 = prefer_bm192's rule, 192 / 256 = force that tile height
extern "C" void vila_gemm_force_bm(int bm) { g_gemm256_bm = bm; }
static int g_gemm256_hybrid = 1;   // tuning hook: 0 = never cut a GEMM into whole rounds + K-sliced tail
extern "C" void vila_gemm_force_hybrid(int on) { g_gemm256_hybrid = on; }

// Tile quantisation: T tiles on 256 CUs cost ceil(T / 256) rounds.  When the last round is short (wgrad of gate/up/down: 1036 tiles =
// 4 rounds + 12 tiles, i.e. a fifth round for 1 % of the work) the whole rounds run as usual and the tail tiles are sliced over K so that
// they fill the chip for a fraction of a tile time; their raw sums go to compact per-tile fp32 slabs and meet in a small reduce kernel
// (bias / residual applied there).  Returns 1 when the GEMM was issued this way, 0 when the caller should launch it whole.
static int try_hybrid(const GemmArgs& a, hipStream_t s) {
    if (!g_gemm256_hybrid || a.ws == nullptr || a.epi != EPI_NONE || a.out_f32 || a.N % 4 != 0) return 0;
    const int tiles_m = cdiv(a.M, 256), tiles = tiles_m * cdiv(a.N, 256), kt = cdiv(a.K, T256_BK);
    const int full = (tiles / 256) * 256, tail = tiles - full;
    if (full == 0 || tail == 0 || tail > 96 || kt < 16) return 0;      // a tail above ~1/3 of a round is cheaper left alone
    int splits = 256 / tail;
    if (splits > 8) splits = 8;
    while (splits >= 2 && (cdiv(kt, splits) < 6 || (size_t)splits * tail * 65536 * 4 > a.ws_bytes)) --splits;
    if (splits < 2) return 0;
    const int per = cdiv(kt, splits);
    splits = cdiv(kt, per);
    const bool cm = a.a_cm || a.b_cm;
    GemmArgs b = a;
    b.C = a.ws; b.bias = nullptr; b.residual = nullptr;
    if (cm) {
        VILA_TRY(launch_gemm256_cm_range(a, 0, 1, 0, full, 0, s));
        VILA_TRY(launch_gemm256_cm_range(b, 5, splits, full, tail, per, s));
    } else {
        VILA_TRY((launch256_t<0, EPI_NONE, false, false, T256_CC_SCHED>(a, s, 1, 0, full)));
        VILA_TRY((launch256_t<5, EPI_NONE, false, false, T256_CC_SCHED>(b, s, splits, full, tail, 0, per)));
    }
    hipLaunchKernelGGL(splitk_tail_reduce_kernel, dim3(tail * 16), dim3(256), 0, s, a.ws, splits, tail, full, tiles_m, cdiv(a.N, 256),
                       gemm256_group(tiles_m, cdiv(a.N, 256), false), a.bias, a.residual, a.ldr,
                       (bf16_t*)a.C, a.ldc, a.M, a.N, a.res_mod);
    VILA_LAUNCH_CHECK();
    return 1;
}

int gemm256_tiles_m_of(int M) { return gemm256_tiles_m(M); }       // for the dispatcher in gemm.hip

bool gemm256_supported(const GemmArgs& a) {
    if (a.a_cm && (a.M % 8 != 0 || a.lda % 8 != 0)) return false;
    if (a.b_cm && (a.N % 8 != 0 || a.ldw % 8 != 0)) return false;
    const int64_t ea = a.a_cm ? (int64_t)64 * a.lda + a.M : (int64_t)a.M * a.lda;      // largest 32-bit element offset the DMA lanes form
    const int64_t eb = a.b_cm ? (int64_t)64 * a.ldw + a.N : (int64_t)a.N * a.ldw;
    // a CC operand is read in 16-B chunks along K (K % 8 == 0); a CM operand in whole k-rows (any K)
    if ((!a.a_cm || !a.b_cm) && a.K % 8 != 0) return false;
    return a.K >= 2 * T256_BK && ea < (1ll << 31) && eb < (1ll << 31);
}

// Gate/up with an under-filled LAST round (S = 769: 592 tiles = 2 full rounds of 256 + 80): the full rounds run fused as usual, the
// tail tiles are sliced over K so the last round costs 1/splits of a tile time; raw gate / up sums meet in a small reduce kernel.
static int launch_gateup(const GemmArgs& a, hipStream_t s) {
    const int tiles_n = cdiv(a.N, 128), kt = cdiv(a.K, T256_BK);
    const bool ex = gemm256_ex_saves_round(a.M, tiles_n);   // 1..16 leftover rows ride in the last row tile (EX kernels)
    const int tiles_m = ex ? a.M / 256 : cdiv(a.M, 256);
    const int slots = 256;                                   // one 512-thread block per CU
    const int full_tn = ((tiles_m * tiles_n) / slots) * slots / tiles_m;     // tile columns covered by whole rounds
    const int tail_tn = tiles_n - full_tn, tail_tiles = tail_tn * tiles_m;
    if (a.ws != nullptr && full_tn > 0 && tail_tiles > 0 && tail_tiles <= slots / 2 && kt >= 16) {
        int splits = slots / tail_tiles;
        if (splits > 4) splits = 4;
        const int per = cdiv(kt, splits);
        splits = cdiv(kt, per);
        const int tc = tail_tn * 128 < a.N - full_tn * 128 ? tail_tn * 128 : a.N - full_tn * 128;      // output columns of the tail
        if (splits >= 2 && (size_t)splits * 2 * a.M * tc * 4 <= a.ws_bytes && tc % 4 == 0) {
            VILA_TRY((launch256_fwd<2, EPI_NONE>(a, s, ex, 1, 0, full_tn * tiles_m)));
            GemmArgs b = a;
            b.C = a.ws; b.ldc = tc;
            VILA_TRY((launch256_fwd<4, EPI_NONE>(b, s, ex, splits, full_tn * tiles_m, tail_tiles, full_tn * 128, per)));
            const int64_t total = (int64_t)a.M * (tc / 4);
            const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_gu_reduce_kernel, dim3(grid), dim3(256), 0, s, a.ws, splits, (bf16_t*)a.C, a.ldc, a.M, tc, full_tn * 128);
            VILA_LAUNCH_CHECK();
            return 0;
        }
    }
    return launch256_fwd<2, EPI_NONE>(a, s, ex);
}

int launch_gemm256(const GemmArgs& a, hipStream_t s) {
    if (g_gemm256_sched == 0 && a.epi == EPI_NONE && !a.out_f32) {
        const int h = try_hybrid(a, s);
        if (h != 0) return h < 0 ? h : 0;
    }
    if (a.a_cm || a.b_cm) return launch_gemm256_cm(a, s);
    if (g_gemm256_sched != 0 && a.epi == EPI_NONE && !a.out_f32) return launch_gemm256_sched(a, g_gemm256_sched, s);
    if (a.epi == EPI_GATEUP) return launch_gateup(a, s);
    const bool ex = gemm256_ex_saves_round(a.M, cdiv(a.N, 256));
    if (a.out_f32) return launch256_fwd<1, EPI_NONE>(a, s, ex);
    switch (a.epi) {
        case EPI_NONE:
            if (prefer_bm192(a.M, a.N, g_gemm256_bm)) return launch256_t<0, EPI_NONE, false, false, T256_CC_SCHED, 192>(a, s);
            return launch256_fwd<0, EPI_NONE>(a, s, ex);
        case EPI_GELU_TANH: return launch256_fwd<0, EPI_GELU_TANH>(a, s, ex);
        case EPI_GELU_ERF: return launch256_fwd<0, EPI_GELU_ERF>(a, s, ex);
    }
    VILA_FAIL(-1, "gemm256: unsupported epilogue %d", a.epi);
}

// VILA_FUSE_NORM=0: the reduce never takes the next block's normalisation along (A/B switch)
static int g_fuse_norm = -1;
extern "C" void vila_gemm_force_fuse_norm(int on) { g_fuse_norm = on ? 1 : 0; }
static bool fused_norm_enabled() {
    if (g_fuse_norm < 0) { const char* e = getenv("VILA_FUSE_NORM"); g_fuse_norm = (e && e[0] == '0') ? 0 : 1; }
    return g_fuse_norm != 0;
}
// split-K: C = sum over `splits` K-slices; `slab` = splits * M * N fp32 workspace owned by the caller
int launch_gemm256_splitk(const GemmArgs& a, int splits, float* slab, hipStream_t s) {
    const int kt = cdiv(a.K, T256_BK), per = cdiv(kt, splits);
    VILA_REQUIRE(a.epi == EPI_NONE && !a.out_f32 && a.N % 4 == 0 && splits >= 1 && (splits - 1) * per < kt,
                 "gemm256 split-K: %d K tiles cannot be cut into %d non-empty slices", kt, splits);
    GemmArgs b = a;
    b.C = slab; b.ldc = a.N; b.bias = nullptr; b.residual = nullptr;
    if (a.a_cm || a.b_cm) VILA_TRY(launch_gemm256_cm_splitk(b, splits, slab, per, s));
    else VILA_TRY((launch256_fwd<3, EPI_NONE>(b, s, gemm256_ex_rows(a.M) != 0, splits, 0, -1, 0, per)));      // the last slice takes the remainder
    if (gemm_rope_offer(a)) {
        const int64_t total = (int64_t)a.M * (a.rope_nq + 2 * a.rope_nkv) * (a.rope_hd / 16);
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(splitk_reduce_rope_kernel, dim3(grid), dim3(256), 0, s, slab, splits, (int64_t)a.M * a.N, a.bias, (bf16_t*)a.C, a.ldc, a.M, a.N,
                           a.rope_cs, a.rope_sn, a.rope_pos, a.rope_seq, a.rope_kc, a.rope_vc, a.rope_nq, a.rope_nkv, a.rope_hd, a.rope_max_ctx);
        VILA_LAUNCH_CHECK();
        *a.rope_done = 1;
        return 0;
    }
    if (a.norm_out != nullptr && a.norm_w != nullptr && a.N % 8 == 0 && a.N <= 16384 && fused_norm_enabled()) {
        // the reduce holds whole rows: the next block's LayerNorm / RMSNorm rides along (one launch and one read of the row less)
        if (a.norm_rms) hipLaunchKernelGGL(splitk_reduce_norm_kernel<true>, dim3(a.M), dim3(256), 0, s, slab, splits, (int64_t)a.M * a.N, a.bias, a.residual, a.ldr,
                                      (bf16_t*)a.C, a.ldc, a.N, a.res_mod, a.norm_w, a.norm_b, a.norm_eps, a.norm_out);
        else hipLaunchKernelGGL(splitk_reduce_norm_kernel<false>, dim3(a.M), dim3(256), 0, s, slab, splits, (int64_t)a.M * a.N, a.bias, a.residual, a.ldr,
                           (bf16_t*)a.C, a.ldc, a.N, a.res_mod, a.norm_w, a.norm_b, a.norm_eps, a.norm_out);
        VILA_LAUNCH_CHECK();
        if (a.norm_done != nullptr) *a.norm_done = 1;
        return 0;
    }
    const int64_t total = (int64_t)a.M * (a.N / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, slab, splits, (int64_t)a.M * a.N, a.bias, a.residual, a.ldr,
                       (bf16_t*)a.C, a.ldc, a.M, a.N, a.res_mod);
    VILA_LAUNCH_CHECK();
    return 0;
}
