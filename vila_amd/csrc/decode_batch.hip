// Batched decode (the serving side of generate — llava/model/llava_arch.py:823-833 with a batch, server.py:171-290
// serving concurrent requests): ONE pass over the weights serves up to 16 sequences.  The batch-1 GEMVs (gemv.hip) become skinny GEMMs on
// the matrix cores: C[batch 16][n 16] += X[batch][k] . W[n][k] with v_mfma_f32_16x16x32_bf16, A = the activations (rows = sequences, zero
// rows beyond the batch), B = 16 weight rows straight from HBM.  The roofline stays HBM (the weight bytes of a token step are read once
// for the whole batch); what changes is the FMA engine: a VALU dot product costs one lane-op per weight element AND sequence.
//
//   * a block = 8 waves = one work item (16 output rows, or a pair of 16-row tiles: gate + up, the rotate-half partners of RoPE) with the
//     64-wide k-blocks dealt round-robin to the waves (consecutive waves read consecutive 128-B lines of a row); partial sums meet in LDS,
//     wave 0 runs the epilogue while the others already stream the next item (one raw barrier per item, double-buffered exchange area);
//   * weights AND activations reach the matrix core through per-wave LDS-DMA rings in whole 128-B lines (bgemm_dma_kernel's header has the
//     why and the numbers); the RMSNorm in front of qkv / gate-up / lm_head is its own launch (elementwise.hip, HF rounding order);
//   * epilogues mirror gemv.hip's rounding exactly: bias, residual, silu(gate) * up, RoPE with the row's own position + KV-cache append into
//     the row's own cache slot, fp32 logits;
//   * attention: one block per (kv head, 256-key slice, sequence) serving the whole GQA group from one K/V read (bdec_attn_kernel).
// History (round 3, batch 8, NVILA-8B, ms per step on one MI355X): register kernel with operand-shaped loads 5.17 -> software-pipelined
// 5.14 -> row-contiguous loads re-dealt by ds_bpermute 6.30 (rejected) -> LDS-DMA rings 3.74 -> GQA-sliced attention 3.59.
#include "kernels.h"
#include "gemv_common.h"
#include "attn_common.h"
#include <cstdlib>
#include <cstring>


struct BGemmArgs {
    const bf16_t* x; int64_t ldx;            // [n][K] activations
    const bf16_t* W; const bf16_t* W2;       // [N][K]; W2: up_proj rows (mode 1)
    const bf16_t* bias;                      // [N] optional
    const bf16_t* residual; int64_t ldr;     // [n][N] optional
    bf16_t* y; int64_t ldy;                  // [n][N] bf16 out
    float* y_f32; int64_t ldf;               // [n][N] fp32 out (logits)
    int n, N, K, mode;                       // mode 0 plain, 1 gate/up, 2 qkv (+bias, RoPE, cache append)
    // mode 2
    bf16_t* q_out; int64_t ldq;              // [n][nq*hd]
    bf16_t* kcache; bf16_t* vcache;          // this layer's [slots][nkv][max_ctx][hd]; row i uses slot i
    int64_t slot_stride;
    const int32_t* pos;                      // [n]
    const float* rope_cs;                    // [n][hd]: cos | sin of each row's position
    int nq, nkv, hd, max_ctx;
};

// The finishing step of one work item, run by one wave on the block-reduced sums (C layout of v_mfma_f32_16x16x32: sequence m = lg*4 + r,
// output feature n = r0 + l15).  Rounding mirrors gemv.hip / HF: every tensor rounded to bf16.
template <int MODE, int NT>
__device__ __forceinline__ void bgemm_epilogue(const BGemmArgs& p, int item, const int (&r0)[NT], const f32x4 (&sum)[NT], int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    const int half = p.hd >> 1, gph = half >> 4;
    (void)half; (void)gph; (void)item;
    // C layout: sequence m = lg*4 + r, output feature n = r0 + l15
    if constexpr (MODE == 1) {
        const int n = r0[0] + l15;
        if (n < p.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = lg * 4 + r;
                if (m < p.n) {
                    const float gv = bfround(sum[0][r]), uv = bfround(sum[1][r]);            // HF: every tensor rounded to bf16
                    p.y[(int64_t)m * p.ldy + n] = f2bf(bfround(silu_f(gv)) * uv);
                }
            }
        }
    } else if constexpr (MODE == 2) {
        const int head = item / gph, j = item % gph;
        const bool is_v = head >= p.nq + p.nkv, is_q = head < p.nq;
        const int gi = j * 16 + l15;                                                         // index inside the half head
        const int na = r0[0] + l15, nb = r0[1] + l15;
        const float ba = p.bias != nullptr ? bf2f(p.bias[na]) : 0.f, bb = p.bias != nullptr ? bf2f(p.bias[nb]) : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lg * 4 + r;
            if (m >= p.n) continue;
            const float lo = bfround(sum[0][r] + ba), hi = bfround(sum[1][r] + bb);
            float oa = lo, ob = hi;
            if (!is_v) {
                const float c = p.rope_cs[m * p.hd + gi], sn = p.rope_cs[m * p.hd + half + gi];
                oa = bfround(bfround(lo * c) + bfround(-hi * sn));
                ob = bfround(bfround(hi * c) + bfround(lo * sn));
            }
            if (is_q) {
                p.q_out[(int64_t)m * p.ldq + na] = f2bf(oa);
                p.q_out[(int64_t)m * p.ldq + nb] = f2bf(ob);
            } else {
                const int ps = p.pos[m];
                if (ps < p.max_ctx) {
                    const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
                    bf16_t* dst = (is_v ? p.vcache : p.kcache) + (int64_t)m * p.slot_stride + ((int64_t)kvh * p.max_ctx + ps) * p.hd;
                    dst[gi] = f2bf(oa);
                    dst[half + gi] = f2bf(ob);
                }
            }
        }
    } else {
        const int n = r0[0] + l15;
        if (n < p.N) {
            const float bv = p.bias != nullptr ? bf2f(p.bias[n]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = lg * 4 + r;
                if (m >= p.n) continue;
                float v = sum[0][r] + bv;
                if (p.y_f32 != nullptr) p.y_f32[(int64_t)m * p.ldf + n] = v;
                if (p.y != nullptr) {
                    if (p.residual != nullptr) v = bfround(v) + bf2f(p.residual[(int64_t)m * p.ldr + n]);
                    p.y[(int64_t)m * p.ldy + n] = f2bf(v);
                }
            }
        }
    }
}

// ---- the weight stream through LDS (the kernel the step uses) ------------------------------------------------------------------------
// The MFMA operand layout (lane = row & 15, 16 B per lane) is the wrong shape to LOAD in: one instruction touches 16 rows x 64 B, every
// 128-B line is fetched by two instructions, and with 8 waves x 16+ loads in flight the second one no longer finds it in the 32-KB vector
// cache (tools/exp/stream_bench.hip: 4.5 TB/s, 5.5 without the streaming hint; the round's first kernel, which loaded that way, reached 3.5-3.9).  Whole lines —
// 8 adjacent lanes x 16 B = one 128-B line, 8 rows per instruction — stream at 6.2 TB/s, but put a row's chunks in 8 different lanes.
// So the lines go through LDS: `global_load_lds` (no VGPRs, lane-linear 1-KB image per instruction) with the XOR swizzle applied to the
// SOURCE chunk (slot s of row r holds chunk s ^ ((r >> 1) & 7)), and ds_read_b128 hands each lane its operand (conflict-free under
// gfx950's b128 lane groups {0-3,12-15,20-27},...: 16 distinct rows per group, 8 of them one k-group further).  The activations take the
// same road (an [XR][128 B] slice per k-block, L2-resident), so the kernel has no staging phase and no K limit; the RMSNorm in front
// becomes its own small launch.  Each wave owns a private ring of NS slots (one slot = one 64-wide k-block: NT x 2 KB of weights +
// XR x 128 B of activations), refilled as soon as a slot is consumed, counted s_waitcnt vmcnt; the only block-wide event is the
// exchange of the 8 partial sums at an item's end (raw s_barrier: __syncthreads() would drain the rings).
// Measured and rejected: carrying the RMSNorm inside this kernel (raw slices in the rings, w * bf16(x * rstd) applied to every A fragment
// between its ds_read and its MFMA, rstd per block in a prologue).  The fragment is re-normalised for every work item and wave, and with
// software rounding that is ~240 VALU ops per k-block: gate/up 46.8 -> 67 us, lm_head 180 -> 414 us; with v_cvt_pk_bf16_f32 the step is
// 3.64 ms against 3.60 with the two 5-us norm launches per layer — not worth a second code path.
template <int MODE, int XR>
__global__ __launch_bounds__(512) void bgemm_dma_kernel(BGemmArgs p, int n_items) {
    constexpr int NT = (MODE == 1 || MODE == 2) ? 2 : 1;
    constexpr int XI = XR / 8;                               // DMA instructions per activation slice
    constexpr int DPS = NT * 2 + XI;                         // DMA instructions per slot
    constexpr int SLOT = DPS * 1024;
    constexpr int RING = 16384;                              // per wave
    constexpr int NS = RING / SLOT;                          // 3 (two tiles, 8 rows) .. 5 (one tile, 8 rows); 2 for two tiles at 16 rows
    static_assert(NS >= 2 && NS <= 5, "ring depth");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int K = p.K, nkb = K >> 6;
    char* ring = smem + wave * RING;
    float* red = (float*)(smem + 8 * RING);                  // [2 parities][8 waves][NT][64 lanes] f32x4
    const int nkw = nkb > wave ? (nkb - wave + 7) >> 3 : 0;  // this wave's k-blocks: wave, wave + 8, ...
    const int half = p.hd >> 1, gph = half >> 4;
    auto tile_rows = [&](int item, int (&r0)[NT]) {
        if constexpr (MODE == 2) {
            const int head = item / gph, j = item % gph;
            r0[0] = head * p.hd + j * 16; r0[1] = r0[0] + half;
        } else if constexpr (MODE == 1) {
            r0[0] = item * 16; r0[1] = item * 16;
        } else {
            r0[0] = item * 16;
        }
    };
    // DMA side: lane = (row8 = lane >> 3, slot = lane & 7) of an 8-row x 128-B piece; the swizzle picks the source chunk
    const int d_row = lane >> 3, d_slot = lane & 7;
    const bf16_t* xsrc[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int m = 8 * j + d_row, mc = m < p.n ? m : p.n - 1;                   // (rows beyond the batch: a copy of the last one, never stored)
        xsrc[j] = p.x + (int64_t)mc * p.ldx + ((d_slot ^ ((m >> 1) & 7)) << 3);
    }
    struct Cur { int item, i; };
    auto issue = [&](Cur c, int slot) {
        const int kb = wave + 8 * c.i;
        int r0[NT];
        tile_rows(c.item, r0);
        char* dst = ring + slot * SLOT;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16_t* wbase = (MODE == 1 && t == 1) ? p.W2 : p.W;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rl = 8 * j + d_row;
                int n = r0[t] + rl; n = n < p.N ? n : p.N - 1;
                const bf16_t* src = wbase + (int64_t)n * K + kb * 64 + ((d_slot ^ ((rl >> 1) & 7)) << 3);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + (t * 2 + j) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < XI; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(xsrc[j] + kb * 64), (lds_void_t*)(dst + (NT * 2 + j) * 1024), 16, 0, 0);
    };
    auto advance = [&](Cur& c) { if (++c.i >= nkw) { c.i = 0; c.item += gridDim.x; } };
    // read side: lane (n = l15, k-group lg) wants chunk h*4 + lg of row n for the h-th MFMA of the k-block
    const int offB = (l15 >> 3) * 1024 + (l15 & 7) * 128 + ((lg ^ ((l15 >> 1) & 7)) << 4);
    const int mA = l15 & (XR - 1);
    const int offA = NT * 2048 + (mA >> 3) * 1024 + (mA & 7) * 128 + ((lg ^ ((mA >> 1) & 7)) << 4);

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Cur ci{(int)blockIdx.x, 0};
    int in_flight = 0, islot = 0, cslot = 0, parity = 0;
    if (nkw > 0) {
#pragma unroll 1
        for (int k = 0; k < NS - 1 && ci.item < n_items; ++k) {
            issue(ci, islot); advance(ci); ++in_flight; if (++islot == NS) islot = 0;
        }
    }
#pragma unroll 1
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
#pragma unroll 1
        for (int i = 0; i < nkw; ++i) {
            if (ci.item < n_items) { issue(ci, islot); advance(ci); ++in_flight; if (++islot == NS) islot = 0; }
            wait_tiles_ahead<DPS, NS - 1>(in_flight - 1);
            --in_flight;
            const char* sp = ring + cslot * SLOT;
            if (++cslot == NS) cslot = 0;
            const u32x4 xa = *(const u32x4*)(sp + offA), xb = *(const u32x4*)(sp + (offA ^ 64));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4 wa = *(const u32x4*)(sp + t * 2048 + offB), wb = *(const u32x4*)(sp + t * 2048 + (offB ^ 64));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, wa), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xb), __builtin_bit_cast(bf16x8, wb), acc[t], 0, 0, 0);
            }
        }
        // ---- the 8 partial sums meet in LDS; wave 0 finishes the item while the others stream the next one ----
        float* rp = red + (size_t)parity * 8 * NT * 256;
        parity ^= 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) { *(f32x4*)(rp + ((wave * NT + t) * 64 + lane) * 4) = acc[t]; acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0) {
            int r0[NT];
            tile_rows(item, r0);
            f32x4 sum[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                sum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const f32x4 a = *(const f32x4*)(rp + ((w * NT + t) * 64 + lane) * 4);
                    sum[t][0] += a[0]; sum[t][1] += a[1]; sum[t][2] += a[2]; sum[t][3] += a[3];
                }
            }
            bgemm_epilogue<MODE, NT>(p, item, r0, sum, lane);
        }
    }
}
template <int MODE, int XR>
static int launch_bgemm_dma_t(const BGemmArgs& a, int n_items, hipStream_t s) {
    constexpr int NT = (MODE == 1 || MODE == 2) ? 2 : 1;
    const size_t lds = (size_t)8 * 16384 + (size_t)2 * 8 * NT * 256 * 4;
    static bool attr = false;
    if (!attr) {
        VILA_HIP(hipFuncSetAttribute((const void*)bgemm_dma_kernel<MODE, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int grid = n_items < 256 ? n_items : 256;
    hipLaunchKernelGGL((bgemm_dma_kernel<MODE, XR>), dim3(grid), dim3(512), lds, s, a, n_items);
    VILA_LAUNCH_CHECK();
    return 0;
}

static int launch_bgemm(const BGemmArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.n >= 1 && a.n <= 16, "batched decode: 1..16 sequences (got %d)", a.n);
    VILA_REQUIRE(a.K % 64 == 0 && a.K > 0 && a.N > 0, "batched decode GEMM: K (%d) must be a positive multiple of 64", a.K);
    VILA_REQUIRE((uintptr_t)a.W % 16 == 0 && (uintptr_t)a.x % 16 == 0 && a.ldx % 8 == 0, "batched decode GEMM: operand alignment");
    const bool x8 = a.n <= 8;
    if (a.mode == 1) {
        VILA_REQUIRE(a.W2 != nullptr && a.y != nullptr, "batched decode GEMM: gate/up needs W2 and a bf16 output");
        return x8 ? launch_bgemm_dma_t<1, 8>(a, cdiv(a.N, 16), s) : launch_bgemm_dma_t<1, 16>(a, cdiv(a.N, 16), s);
    }
    if (a.mode == 2) {
        VILA_REQUIRE(a.hd % 32 == 0 && a.N == (a.nq + 2 * a.nkv) * a.hd && a.q_out && a.kcache && a.vcache && a.pos && a.rope_cs,
                     "batched decode GEMM: qkv mode needs head_dim %% 32 == 0 and its outputs");
        const int items = (a.nq + 2 * a.nkv) * (a.hd / 32);
        return x8 ? launch_bgemm_dma_t<2, 8>(a, items, s) : launch_bgemm_dma_t<2, 16>(a, items, s);
    }
    return x8 ? launch_bgemm_dma_t<0, 8>(a, cdiv(a.N, 16), s) : launch_bgemm_dma_t<0, 16>(a, cdiv(a.N, 16), s);
}

// ---- per-row prologue / pick / advance -------------------------------------------------------------------------------------------
// x[row] = embed[token[row]]; rope table of the row's position: cs[row][0:hd/2] = cos, [hd/2:hd] = sin, rounded to bf16 like HF
__global__ void bdec_prologue_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ tok, bf16_t* __restrict__ out, int H, int64_t vocab,
                                     const int32_t* __restrict__ pos, float* __restrict__ rope_cs, int hd, float theta) {
    const int row = blockIdx.y;
    int64_t id = tok[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < (H >> 3); c += gridDim.x * blockDim.x)
        *(u32x4*)(out + (int64_t)row * H + c * 8) = *(const u32x4*)(table + id * H + c * 8);
    if (blockIdx.x == 0 && (int)threadIdx.x < (hd >> 1)) {
        const int d = threadIdx.x;
        const float inv = 1.0f / powf(theta, (float)(2 * d) / (float)hd);
        const float ang = (float)pos[row] * inv;
        rope_cs[row * hd + d] = bfround(cosf(ang));
        rope_cs[row * hd + (hd >> 1) + d] = bfround(sinf(ang));
    }
}
// greedy pick per row (first index of the maximum, like argmax_stage1/2) and the state advance.  Two launches: PICK_SLICES blocks per row each
// scan a slice of the vocabulary (one block per row took 72 us for 8 x 152064 logits on 8 CUs), then one small block per row merges and advances.
#define PICK_SLICES 32
__global__ __launch_bounds__(256) void bdec_pick1_kernel(const float* __restrict__ logits, int V, float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.y, sl = blockIdx.x;
    const int per = (V + PICK_SLICES - 1) / PICK_SLICES, lo = sl * per, hi = lo + per < V ? lo + per : V;
    const float* lr = logits + (int64_t)row * V;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) { const float v = lr[i]; if (v > best || (v == best && i < bi)) { best = v; bi = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        pv[row * PICK_SLICES + sl] = best; pi[row * PICK_SLICES + sl] = bi;
    }
}
__global__ __launch_bounds__(64) void bdec_pick2_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int64_t* __restrict__ token,
                                                        int32_t* __restrict__ pos, int64_t* __restrict__ out_ids, int32_t* __restrict__ n_out, int max_out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float best = lane < PICK_SLICES ? pv[row * PICK_SLICES + lane] : -INFINITY;
    int bi = lane < PICK_SLICES ? pi[row * PICK_SLICES + lane] : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) {
        token[row] = (int64_t)bi;
        const int n = n_out[row];
        if (n < max_out) out_ids[(int64_t)row * max_out + n] = (int64_t)bi;
        n_out[row] = n + 1;
        pos[row] = pos[row] + 1;
    }
}

// ---- attention of the batch: one block per (kv head, 256-key slice, sequence) -------------------------------------------------------------
// The batch-1 kernel (gemv.hip attn_decode_head) gives every QUERY head its own block, so the 7 heads of a GQA group each pull the same
// K/V rows through L2, and a row's whole context is one block's serial loop (19.6 us per layer at 8 x ~800 keys).  Here the K/V chunk a
// wave loads (16 keys) serves all G query heads of its kv head (scores on the matrix core: S[16 keys][16 heads] = K . Q^T in 4 MFMAs; P.V on
// the VALU with the probabilities fetched by DPP row broadcasts, reductions by row swaps — the __shfl_xor formulation, 30 ds_bpermute per
// head, took 69 us), a slice is one chunk per wave (no loop), and the slices are merged by a second small launch.  Merging inside the launch
// (last-arriving block, agent-scope release/acquire by one lane) measured the same 16-17 us total and needs counters; with EVERY wave
// fencing it was 59 us (an agent fence is an L2 write-back + invalidate on this multi-XCD part).
// sum over lane l and lane l ^ 16 (resp. l ^ 32) with gfx950's row swaps (see attn_common.h xor16_max): no LDS crossbar, no lgkmcnt
__device__ __forceinline__ float bd_xor16_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float bd_xor32_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
struct BAttnArgs {
    const bf16_t* q; bf16_t* o;                 // [rows][nq*128]
    const bf16_t* kcache; const bf16_t* vcache; // this layer's [slots][nkv][max_ctx][128]
    const int32_t* pos;                         // [rows]: keys 0 .. pos inclusive
    float* part_o; float* part_ml;              // [rows][nkv][NSL][G][128], [rows][nkv][NSL][G][2]
    int nq, nkv, max_ctx, nsl; int64_t row_stride, slot_stride; float scale;
};
template <int G>
__global__ __launch_bounds__(1024) void bdec_attn_kernel(BAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* so = (float*)smem;                   // [16 waves][G][128]
    float* sml = so + 16 * G * 128;             // [16][G][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kvh = blockIdx.x, slice = blockIdx.y, row = blockIdx.z;
    const int nkeys_all = p.pos[row] + 1;
    const int active = (nkeys_all + 255) >> 8;
    if (slice >= active) return;                                 // block-uniform
    const int key_lo = slice * 256;
    const int nkeys = nkeys_all < key_lo + 256 ? nkeys_all : key_lo + 256;
    const bf16_t* kb = p.kcache + (int64_t)row * p.slot_stride + (int64_t)kvh * p.max_ctx * 128;
    const bf16_t* vb = p.vcache + (int64_t)row * p.slot_stride + (int64_t)kvh * p.max_ctx * 128;
    const bf16_t* qrow = p.q + (int64_t)row * p.row_stride + kvh * G * 128;
    const int l15 = lane & 15, lg = lane >> 4;      // scores: MFMA A rows = keys, B rows = query heads; C: key lg*4 + r, head l15
    const int sg = lane >> 4, dc = lane & 15;       // P.V: 4-key subgroup (= the C layout's row group), d chunk
    // the wave's 16 keys: everything is issued before anything is used
    const int k0 = key_lo + wave * 16;
    u32x4 kc[4], qc[4], vc[4];
    {
        const int key = k0 + l15;
        const bool kok = key < nkeys, qok = l15 < G;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kc[ks] = kok ? *(const u32x4*)(kb + (int64_t)key * 128 + ks * 32 + lg * 8) : (u32x4){0u, 0u, 0u, 0u};
            qc[ks] = qok ? *(const u32x4*)(qrow + l15 * 128 + ks * 32 + lg * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int vk = k0 + sg * 4 + j;
            vc[j] = (vk < nkeys) ? *(const u32x4*)(vb + (int64_t)vk * 128 + dc * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    // S[key][head] on the matrix core: 4 MFMAs replace G x (64 FMAs + a 4-lane reduction)
    f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kc[ks]), __builtin_bit_cast(bf16x8, qc[ks]), sc, 0, 0, 0);
    float pr[4], m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[r] = (k0 + lg * 4 + r < nkeys) ? sc[r] * p.scale : -INFINITY; m = fmaxf(m, pr[r]); }
    m = xor32_max(xor16_max(m));                                 // over the 16 keys of the chunk (lanes l15, l15 + 16, + 32, + 48)
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[r] = (m == -INFINITY) ? 0.f : __expf(pr[r] - m); l += pr[r]; }
    l = bd_xor32_sum(bd_xor16_sum(l));
    if (lg == 0 && l15 < G) { sml[(wave * G + l15) * 2] = m; sml[(wave * G + l15) * 2 + 1] = l; }
    // P.V on the VALU: the probability of (key sg*4 + j, head g) sits in lane 16*sg + g, register j — one DPP row broadcast away
    static_for<0, G>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pj = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(pr[j]), 0x150 + g, 0xf, 0xf, false));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] = fmaf(pj, lo_bf(vc[j][e]), o[2 * e]);
                o[2 * e + 1] = fmaf(pj, hi_bf(vc[j][e]), o[2 * e + 1]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bd_xor32_sum(bd_xor16_sum(o[e]));
        if (sg == 0) {
            float* dst = so + (wave * G + g) * 128 + dc * 8;
            *(f32x4*)dst = (f32x4){o[0], o[1], o[2], o[3]};
            *(f32x4*)(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
        }
    });
    __syncthreads();
    // ---- the 16 waves' partials -> the slice's partial ----
    const int64_t pbase = ((int64_t)(row * p.nkv + kvh) * p.nsl + slice) * G;
    const int g = tid >> 7, d = tid & 127;
    if (g < G) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 16; ++w) M = fmaxf(M, sml[(w * G + g) * 2]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const float mw = sml[(w * G + g) * 2];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L = fmaf(sml[(w * G + g) * 2 + 1], f, L);
            O = fmaf(so[(w * G + g) * 128 + d], f, O);
        }
        p.part_o[(pbase + g) * 128 + d] = O;
        if (d == 0) { p.part_ml[(pbase + g) * 2] = M; p.part_ml[(pbase + g) * 2 + 1] = L; }
    }
}
// the slices of a (sequence, query head) -> the bf16 attention output; the launch boundary is the publish (no fences)
__global__ __launch_bounds__(128) void bdec_attn_merge_kernel(BAttnArgs p, int G) {
    const int h = blockIdx.x, row = blockIdx.y, d = threadIdx.x;
    const int kvh = h / G, g = h % G;
    const int active = (p.pos[row] + 1 + 255) >> 8;
    const int64_t b0 = (int64_t)(row * p.nkv + kvh) * p.nsl * G;
    float M = -INFINITY;
    for (int sl = 0; sl < active; ++sl) M = fmaxf(M, p.part_ml[(b0 + sl * G + g) * 2]);
    float L = 0.f, O = 0.f;
    for (int sl = 0; sl < active; ++sl) {
        const float f = __expf(p.part_ml[(b0 + sl * G + g) * 2] - M);
        L = fmaf(p.part_ml[(b0 + sl * G + g) * 2 + 1], f, L);
        O = fmaf(p.part_o[(b0 + sl * G + g) * 128 + d], f, O);
    }
    p.o[(int64_t)row * p.row_stride + h * 128 + d] = f2bf(O / L);
}
template <int G>
static int launch_bdec_attn_t(const BAttnArgs& a, int rows, hipStream_t s) {
    const size_t lds = (size_t)(16 * G * 128 + 16 * G * 2) * 4;
    static bool attr = false;
    if (!attr) {
        VILA_HIP(hipFuncSetAttribute((const void*)bdec_attn_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((bdec_attn_kernel<G>), dim3(a.nkv, a.nsl, rows), dim3(1024), lds, s, a);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(bdec_attn_merge_kernel, dim3(a.nq, rows), dim3(128), 0, s, a, G);
    VILA_LAUNCH_CHECK();
    return 0;
}
// returns 1 when the GQA group size has no instantiation (the caller falls back to the per-head kernel)
static int launch_bdec_attn(const BAttnArgs& a, int rows, hipStream_t s) {
    switch (a.nq / a.nkv) {
        case 1: return launch_bdec_attn_t<1>(a, rows, s);
        case 2: return launch_bdec_attn_t<2>(a, rows, s);
        case 4: return launch_bdec_attn_t<4>(a, rows, s);
        case 7: return launch_bdec_attn_t<7>(a, rows, s);
        case 8: return launch_bdec_attn_t<8>(a, rows, s);
        default: return 1;
    }
}

// ---- the step ---------------------------------------------------------------------------------------------------------------------
size_t bdecode_workspace_bytes(int H, int F, int QS, int hd, int n) {
    size_t b = 0;
    b += 3 * align_up((size_t)n * H * 2, 256) + 2 * align_up((size_t)n * QS * 2, 256) + align_up((size_t)n * F * 2, 256);
    b += align_up((size_t)n * hd * 4, 256);
    b += align_up((size_t)n * (QS / hd) * 8 * (hd + 2) * 4, 256) + 256 + 2 * align_up((size_t)n * 32 * 4, 256);   // attention slice partials, argmax partials
    return b + 4096;
}

int bdecode_step(const BDecodeArgs& m, const BLayer* layers, bf16_t* kcache, bf16_t* vcache, int max_ctx, int n_slots, int n, int32_t* pos, int64_t* token,
                 int64_t* out_ids, int32_t* n_out, int max_out, float* logits, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const int H = m.hidden, F = m.inter, hd = m.head_dim, QS = m.q_heads * hd, KS = m.kv_heads * hd;
    VILA_REQUIRE(n >= 1 && n <= 16 && n <= n_slots, "batched decode: %d sequences need 1..16 KV-cache slots (cache has %d)", n, n_slots);
    VILA_REQUIRE(QS == H, "batched decode: q_heads*head_dim (%d) must equal hidden (%d)", QS, H);
    VILA_REQUIRE(hd == 128 && max_ctx <= 2048, "batched decode: head_dim 128 and caches up to 2048 positions (got %d, %d)", hd, max_ctx);
    VILA_REQUIRE(workspace_bytes >= bdecode_workspace_bytes(H, F, QS, hd, n), "batched decode: workspace too small");
    char* wp = (char*)workspace; size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); void* r = wp + off; off += bytes; return r; };
    bf16_t* x = (bf16_t*)take((size_t)n * H * 2);
    bf16_t* x2 = (bf16_t*)take((size_t)n * H * 2);
    bf16_t* xn = (bf16_t*)take((size_t)n * H * 2);           // the normalised activations in front of qkv / gate-up / lm_head
    auto normed = [&](BGemmArgs& g, const bf16_t* src, const void* w) -> int {
        VILA_TRY(launch_rmsnorm(src, (const bf16_t*)w, xn, n, H, m.rms_eps, s));
        g.x = xn;
        return 0;
    };
    bf16_t* q = (bf16_t*)take((size_t)n * QS * 2);
    bf16_t* ao = (bf16_t*)take((size_t)n * QS * 2);
    bf16_t* act = (bf16_t*)take((size_t)n * F * 2);
    float* rope_cs = (float*)take((size_t)n * hd * 4);
    const int nsl = cdiv(max_ctx, 256);
    float* part_o = (float*)take((size_t)n * m.q_heads * nsl * hd * 4);
    float* part_ml = (float*)take((size_t)n * m.q_heads * nsl * 2 * 4);
    float* pick_v = (float*)take((size_t)n * PICK_SLICES * 4);
    int* pick_i = (int*)take((size_t)n * PICK_SLICES * 4);
    VILA_REQUIRE(off <= workspace_bytes, "batched decode: workspace layout");
    hipLaunchKernelGGL(bdec_prologue_kernel, dim3(cdiv(H / 8, 256), n), dim3(256), 0, s, (const bf16_t*)m.embed, token, x, H, (int64_t)m.vocab, pos, rope_cs, hd, m.rope_theta);
    VILA_LAUNCH_CHECK();
    const int64_t per_layer = (int64_t)n_slots * m.kv_heads * max_ctx * hd, slot_stride = (int64_t)m.kv_heads * max_ctx * hd;
    bf16_t* cur = x; bf16_t* nxt = x2;
    for (int l = 0; l < m.n_layers; ++l) {
        const BLayer& L = layers[l];
        bf16_t* kc = kcache + l * per_layer; bf16_t* vc = vcache + l * per_layer;
        BGemmArgs qa{};
        VILA_TRY(normed(qa, cur, L.ln1_w));
        qa.ldx = H; qa.W = (const bf16_t*)L.wqkv; qa.bias = (const bf16_t*)L.bqkv;
        qa.n = n; qa.N = QS + 2 * KS; qa.K = H; qa.mode = 2; qa.q_out = q; qa.ldq = QS; qa.kcache = kc; qa.vcache = vc; qa.slot_stride = slot_stride;
        qa.pos = pos; qa.rope_cs = rope_cs; qa.nq = m.q_heads; qa.nkv = m.kv_heads; qa.hd = hd; qa.max_ctx = max_ctx;
        VILA_TRY(launch_bgemm(qa, s));
        AttnDecodeArgs ad{};
        ad.q = q; ad.kcache = kc; ad.vcache = vc; ad.o = ao; ad.pos_ptr = pos; ad.nq = m.q_heads; ad.nkv = m.kv_heads; ad.hd = hd; ad.max_ctx = max_ctx;
        ad.n_splits = cdiv(max_ctx, 64); ad.scale = 1.0f / sqrtf((float)hd);
        BAttnArgs ba{};
        ba.q = q; ba.o = ao; ba.kcache = kc; ba.vcache = vc; ba.pos = pos; ba.part_o = part_o; ba.part_ml = part_ml;
        ba.nq = m.q_heads; ba.nkv = m.kv_heads; ba.max_ctx = max_ctx; ba.nsl = nsl; ba.row_stride = QS; ba.slot_stride = slot_stride; ba.scale = ad.scale;
        const int rc = (m.q_heads % m.kv_heads == 0) ? launch_bdec_attn(ba, n, s) : 1;
        if (rc < 0) return rc;
        if (rc == 1) VILA_TRY(launch_attn_decode_rows(ad, n, QS, QS, slot_stride, s));       // group size without an instantiation: one block per query head
        BGemmArgs o{};
        o.x = ao; o.ldx = QS; o.W = (const bf16_t*)L.wo; o.residual = cur; o.ldr = H; o.y = nxt; o.ldy = H; o.n = n; o.N = H; o.K = QS; o.mode = 0;
        VILA_TRY(launch_bgemm(o, s));
        BGemmArgs gu{};
        VILA_TRY(normed(gu, nxt, L.ln2_w));
        gu.ldx = H; gu.W = (const bf16_t*)L.w_gate; gu.W2 = (const bf16_t*)L.w_up;
        gu.y = act; gu.ldy = F; gu.n = n; gu.N = F; gu.K = H; gu.mode = 1;
        VILA_TRY(launch_bgemm(gu, s));
        BGemmArgs dn{};
        dn.x = act; dn.ldx = F; dn.W = (const bf16_t*)L.w_down; dn.residual = nxt; dn.ldr = H; dn.y = cur; dn.ldy = H; dn.n = n; dn.N = H; dn.K = F; dn.mode = 0;
        VILA_TRY(launch_bgemm(dn, s));
    }
    BGemmArgs lm{};
    VILA_TRY(normed(lm, cur, m.norm_w));
    lm.ldx = H; lm.W = (const bf16_t*)m.lm_head; lm.y_f32 = logits; lm.ldf = m.vocab;
    lm.n = n; lm.N = m.vocab; lm.K = H; lm.mode = 0;
    VILA_TRY(launch_bgemm(lm, s));
    hipLaunchKernelGGL(bdec_pick1_kernel, dim3(PICK_SLICES, n), dim3(256), 0, s, logits, m.vocab, pick_v, pick_i);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(bdec_pick2_kernel, dim3(n), dim3(64), 0, s, pick_v, pick_i, token, pos, out_ids, n_out, max_out);
    VILA_LAUNCH_CHECK();
    return 0;
}
